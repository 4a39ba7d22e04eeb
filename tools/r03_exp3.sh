# round-3 batch 3: FRI proof written into pinned memory (parity + I/O-inclusive rate), the new bench legs, total time of the default bench
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp3
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pinned or async or wide or lincomb or arith" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
/usr/bin/time -v -o $O/bench.time timeout -k 10 900 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
grep -E "Elapsed|Maximum resident" $O/bench.time
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value']); e=d['extra']
for k in e: print(k, {kk: vv for kk, vv in e[k].items() if kk != 'workload'} if isinstance(e[k], dict) else e[k])
print(d['cpu_baseline'])"
