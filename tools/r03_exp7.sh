set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp7
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -- python3 bench.py --inflight 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/trace1.log 2>&1
python3 tools/trace_gaps.py $O/trace1 | tee $O/gaps.txt
