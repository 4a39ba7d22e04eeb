# round-3 batch 4: SHIFT1 (second sub-round boundary as wave-uniform shifts): parity, NTT rates; then the full default bench, timed
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp4
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "arith or ntt or babybear or coset or test_prove or roots" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 120 python3 tools/ntt_bench.py --field 0 --log-rows 20 22 24 --tag gl_shift1 > $O/ntt_gl.log 2>&1
grep tag $O/ntt_gl.log
T0=$(date +%s)
timeout -k 10 1000 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
echo "default bench wall seconds: $(( $(date +%s) - T0 ))"
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['roofline']['frac'], d['kernel_ms_per_proof']['ntt_pass']); e=d['extra']
for k in e: print(k, {kk: vv for kk, vv in e[k].items() if kk != 'workload'} if isinstance(e[k], dict) else e[k])
print(d['cpu_baseline'])"
