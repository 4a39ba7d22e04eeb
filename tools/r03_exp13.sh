set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp13
mkdir -p $O
for i in 1 2; do for v in cur 86cd29d 50f5d8f; do
cp tools/libs/libministark_$v.so mini-stark_amd/libministark.so
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras > $O/thr_${v}_$i.json 2> $O/thr.err
timeout -k 10 200 python3 bench.py --inflight 1 --steps 30 --no-cpu-baseline --no-extras > $O/lat_${v}_$i.json 2> $O/lat.err
done; done
cp tools/libs/libministark_cur.so mini-stark_amd/libministark.so
python3 -c "
import json,glob
for f in sorted(glob.glob('$O/*.json')): d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), {k: round(v,3) for k,v in d['kernel_ms_per_proof'].items() if v > 0.3})"
