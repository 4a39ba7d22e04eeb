#!/bin/bash
# Same-box A/B of benchmark variants (run on the GPU box through gpurun): the variants are environment settings or alternative libraries, alternated
# over several passes inside ONE call - box-to-box spread is +-6 %, so only same-call comparisons mean anything.
#   bash tools/ab.sh OUTDIR PASSES "name1|ENV=V ..|bench args" "name2|..|.." ...
# e.g. bash tools/ab.sh gpurun_out/ab_share 3 "share|MS_NTT_SHARE=1|--steps 40" "noshare|MS_NTT_SHARE=0|--steps 40"
# A variant's third field is appended to `python3 bench.py --no-cpu-baseline --no-extras`; a field starting with "tool:" runs that command line instead.
# (This replaces the one-off tools/r03_exp*.sh drivers of round 3; profiles/HISTORY.md is the record of what they measured.)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; P=$2; shift 2
mkdir -p $O
for pass in $(seq 1 $P); do
  for v in "$@"; do
    IFS='|' read -r name envs args <<< "$v"
    if [[ "$args" == tool:* ]]; then cmd="${args#tool:}"; else cmd="python3 bench.py --no-cpu-baseline --no-extras $args"; fi
    env $envs timeout -k 10 600 $cmd > $O/${name}_$pass.json 2> $O/${name}_$pass.err
    rc=$?
    echo "$name pass $pass rc=$rc $(grep -o '"value": [0-9.]*' $O/${name}_$pass.json | head -1)"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ]; then echo "stopping: $name timed out or aborted"; exit $rc; fi
  done
done
