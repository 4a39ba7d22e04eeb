set -e
O=gpurun_out/r03_stagger
mkdir -p $O
rm -f $O/ab2.log
for i in 1 2; do
for fl in 6 8 10 12; do
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --inflight $fl --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('inflight $fl steps 20', round(d['value'],1))" >> $O/ab2.log
done
for x in 0 1; do
MS_BENCH_STAGGER_MS=$x timeout -k 10 300 python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('stagger $x steps 200', round(d['value'],1))" >> $O/ab2.log
done
done
sort $O/ab2.log
