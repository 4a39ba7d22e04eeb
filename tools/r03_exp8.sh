set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp8
mkdir -p $O
for i in 1 2; do
timeout -k 10 200 python3 bench.py --inflight 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/lat_default_$i.json 2> $O/lat_default.err
HSA_ENABLE_INTERRUPT=0 timeout -k 10 200 python3 bench.py --inflight 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/lat_nointr_$i.json 2> $O/lat_nointr.err
done
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras > $O/thr_default.json 2> $O/thr_default.err
HSA_ENABLE_INTERRUPT=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras > $O/thr_nointr.json 2> $O/thr_nointr.err
python3 -c "
import json,glob
for f in sorted(glob.glob('$O/*.json')): print(f.split('/')[-1], round(json.load(open(f))['value'],1))"
