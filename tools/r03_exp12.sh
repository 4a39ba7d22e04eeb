set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp12
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ntt or coset or test_prove" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for i in 1 2 3; do
timeout -k 10 120 python3 tools/ntt_bench.py --field 0 --log-rows 20 24 --reps 60 --tag share >> $O/ab.log 2>&1
MS_NTT_SHARE=0 timeout -k 10 120 python3 tools/ntt_bench.py --field 0 --log-rows 20 24 --reps 60 --tag noshare >> $O/ab.log 2>&1
done
grep tag $O/ab.log | python3 -c "
import sys, json, collections
d=collections.defaultdict(list)
for l in sys.stdin: j=json.loads(l); d[(j['tag'],j['log_rows'])].append(j['lde_ms'])
for k,v in sorted(d.items()): print(k, v)"
