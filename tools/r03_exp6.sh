set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp6
mkdir -p $O
timeout -k 10 200 python3 tools/ntt_bench.py --field 0 --log-rows 16 17 18 19 20 21 22 --reps 40 --tag sizes > $O/ntt_sizes.log 2>&1
grep tag $O/ntt_sizes.log
