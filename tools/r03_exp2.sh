# round-3 batch 2: pruned PassKernel2 + GLM sub3 + BabyBear on the 3-sub-round tiles: parity subset, NTT rates, bench
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp2
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "arith or ntt or babybear or coset or test_prove or roots" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 120 python3 tools/ntt_bench.py --field 0 --log-rows 20 24 --tag gl > $O/ntt_gl.log 2>&1
timeout -k 10 120 python3 tools/ntt_bench.py --field 1 --log-rows 20 22 --tag bb_sub3 > $O/ntt_bb.log 2>&1
MS_NTT_V2_SUB3=0 timeout -k 10 120 python3 tools/ntt_bench.py --field 1 --log-rows 20 22 --tag bb_sub2 >> $O/ntt_bb.log 2>&1
cat $O/ntt_gl.log $O/ntt_bb.log | grep tag
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['roofline']['frac'], d['kernel_ms_per_proof']['ntt_pass'], d['extra']['babybear_fp4_2p20_rows']['value'], d['extra']['ntt_only'])"
