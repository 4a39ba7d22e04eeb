set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp9
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_prove or lincomb or wide or merkle or closure or many" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python3 tools/wide_bench.py > $O/wide_virtual.log 2>&1; grep workload $O/wide_virtual.log
MS_LDE_VIRTUAL=0 timeout -k 10 300 python3 tools/wide_bench.py > $O/wide_material.log 2>&1; grep workload $O/wide_material.log
for i in 1 2; do
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras > $O/thr_virtual_$i.json 2> $O/thr.err
MS_LDE_VIRTUAL=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras > $O/thr_material_$i.json 2> $O/thr.err
done
python3 -c "
import json,glob
for f in sorted(glob.glob('$O/thr*.json')): d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), d['kernel_ms_per_proof'].get('leaf_hash'), d['kernel_ms_per_proof'].get('lincomb'))"
