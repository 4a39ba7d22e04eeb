# round-3 experiment batch 1 (GPU box): arithmetic lab with the r03 variants, this box's baseline, BabyBear / wide-AIR starting points
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp1
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/ntt_lab.hip -o /tmp/ntt_lab && timeout -k 5 200 /tmp/ntt_lab > $O/ntt_lab.log 2>&1
echo lab done
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/bench0.json 2> $O/bench0.err
echo bench done
timeout -k 10 120 python3 tools/ntt_bench.py --field 0 --log-rows 20 24 --tag gl > $O/ntt_gl.log 2>&1
timeout -k 10 120 python3 tools/ntt_bench.py --field 1 --log-rows 20 22 --tag bb_default > $O/ntt_bb.log 2>&1
MS_NTT_V2=2 timeout -k 10 120 python3 tools/ntt_bench.py --field 1 --log-rows 20 22 --tag bb_v2 >> $O/ntt_bb.log 2>&1
echo ntt done
timeout -k 10 300 python3 tools/wide_bench.py > $O/wide_default.log 2>&1
MS_LDE_MULTI=1 timeout -k 10 300 python3 tools/wide_bench.py > $O/wide_multi.log 2>&1
echo wide done
