set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp10
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "arith or ntt or babybear or coset or test_prove or roots or virtual" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 120 python3 tools/ntt_bench.py --field 0 --log-rows 20 22 24 --reps 40 --tag gl_share > $O/ntt_gl.log 2>&1; grep tag $O/ntt_gl.log
timeout -k 10 120 python3 tools/ntt_bench.py --field 0 --log-rows 20 --reps 40 --tag gl_noshare --lib tools/libs/libministark_noshift1.so >> $O/ntt_gl.log 2>&1; grep noshare $O/ntt_gl.log
timeout -k 10 120 python3 tools/ntt_bench.py --field 1 --log-rows 20 22 --reps 40 --tag bb_share > $O/ntt_bb.log 2>&1; grep tag $O/ntt_bb.log
