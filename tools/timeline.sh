#!/bin/bash
# Per-launch timeline of ONE proof alone on the GPU (run on the GPU box through gpurun): rocprofv3 kernel trace of a short single-lane bench, reduced by tools/trace_gaps.py --dump.
#   bash tools/timeline.sh OUTDIR NAME [ENV=V ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; N=$2; shift 2
mkdir -p $O
for kv in "$@"; do export "$kv"; done
rm -rf $O/trace_$N
rocprofv3 --kernel-trace --output-format csv -d $O/trace_$N -- python3 bench.py --inflight 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/trace_$N.log 2>&1 || { echo "rocprofv3 failed"; tail -5 $O/trace_$N.log; exit 1; }
python3 tools/trace_gaps.py $O/trace_$N --dump > $O/timeline_$N.txt 2>&1
head -32 $O/timeline_$N.txt
rm -rf $O/trace_$N      # (the raw trace is tens of MiB; the reduced timeline is what is kept)
