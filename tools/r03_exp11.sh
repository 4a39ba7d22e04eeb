set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp11
mkdir -p $O
for i in 1 2 3; do
for f in 0 1; do
timeout -k 10 120 python3 tools/ntt_bench.py --field $f --log-rows 20 --reps 60 --tag share_f$f >> $O/ab.log 2>&1
timeout -k 10 120 python3 tools/ntt_bench.py --field $f --log-rows 20 --reps 60 --tag noshare_f$f --lib tools/libs/libministark_noshare.so >> $O/ab.log 2>&1
done; done
grep tag $O/ab.log | python3 -c "
import sys, json, collections
d=collections.defaultdict(list)
for l in sys.stdin: j=json.loads(l); d[j['tag']].append(j['lde_ms'])
for k,v in sorted(d.items()): print(k, v)"
