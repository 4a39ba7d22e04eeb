set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp19
mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config3 or config4 or full_size or 2p22 or test_prove or cubic or degree3" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python3 bench.py --log-rows 24 --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d[\"value\"],2), d[\"ms_per_step\"]); print(json.dumps(d[\"kernel_ms_per_proof\"]))" | tee $O/kernel_ms_2p24.log
python3 bench.py --log-rows 24 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2 in flight', round(d[\"value\"],2))" | tee -a $O/kernel_ms_2p24.log
python3 bench.py --log-rows 22 --steps 5 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2^22', round(d[\"value\"],2))" | tee -a $O/kernel_ms_2p24.log
