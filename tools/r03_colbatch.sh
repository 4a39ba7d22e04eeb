set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_colbatch
mkdir -p $O
rm -f $O/ab.log
for i in 1 2 3; do
for cb in 0 1 2 3; do
MS_NTT_COLBATCH=$cb timeout -k 10 120 python3 tools/ntt_bench.py --field 0 --log-rows 20 --reps 40 --tag colbatch$cb --passes >> $O/ab.log 2>> $O/ab.err
done
done
grep lde_ms $O/ab.log | python3 -c "
import sys, json, collections
d=collections.defaultdict(list)
for l in sys.stdin: j=json.loads(l); d[(j['field'],j['log_rows'],j['tag'])].append(j['lde_ms'])
for k,v in sorted(d.items()): print(k, v)"
grep pass_us $O/ab.log | head -4
