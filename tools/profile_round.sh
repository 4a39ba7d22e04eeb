# Collects the judged profile artefacts for one round (run on the GPU box through gpurun):
#   bash tools/profile_round.sh r01
set -e
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$R
rm -rf $O          # (nothing of an earlier run may be taken for this one's)
mkdir -p $O
date +%s > $O/started_at
trap 'echo "{\"rc\": $?, \"finished_at\": $(date +%s)}" > $O/passes_done.json' EXIT   # rc 0 only if EVERY pass below succeeded (set -e): tools/process_profiles.py refuses anything else
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err   # the driver's command (includes the CPU baselines: ~3.5 minutes)
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ntt_lab.hip -o /tmp/ntt_lab 2>/dev/null && /tmp/ntt_lab > $O/ntt_lab.log 2>&1 || true
# (no 8-lane profiler pass any more: rocprofv3 7.2's queue intercept overruns an AQL ring when several host threads submit to one intercepted queue -
#  profiles/r04_sigsegv_analysis.md; every profiler pass runs ONE proving thread, which is also the pass the roofline figures use)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 bench.py --inflight 1 --no-cpu-baseline --no-extras > $O/stats1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-extras > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-extras > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-extras > $O/pmc_sq.log 2>&1
hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate > $O/valu_rate.txt 2>&1 || true   # built from source every time (the binary is not tracked)
hipcc --offload-arch=gfx950 -O2 tools/latency_probe.hip -o /tmp/latency_probe 2>/dev/null && timeout -k 5 120 /tmp/latency_probe > $O/latency_probe.txt 2>&1 || true
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mini-stark_amd/csrc tools/sha_lab.hip -o /tmp/sha_lab 2>/dev/null && timeout -k 5 100 /tmp/sha_lab > $O/sha_lab.log 2>&1 || true          # SHA-256 on registers only: the hash kernels' ceiling
hipcc --offload-arch=gfx950 -O3 tools/stride_probe.hip -o /tmp/stride_probe 2>/dev/null && timeout -k 5 120 /tmp/stride_probe > $O/stride_probe.log 2>&1 || true            # access pattern of the NTT later pass, no arithmetic
timeout -k 10 600 python3 tools/io_probe3.py --steps 20 --rounds 3 2>/dev/null | grep '^{' > $O/io_probe.log || true                                                          # the I/O leg in halves, HIP copies vs SDMA engines, alternated in one process
# BabyBear NTT: six-column LDE 2^20 -> 2^23 on the three-sub-round tiles: rates without the profiler, then FETCH / WRITE / SQ passes
timeout -k 10 200 python3 tools/ntt_bench.py --field 1 --log-rows 20 22 --reps 40 --tag babybear > $O/ntt_bb.log 2>&1 || true
timeout -k 10 200 python3 tools/ntt_bench.py --field 0 --log-rows 20 24 --reps 40 --tag goldilocks > $O/ntt_gl.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/bb_fetch -- python3 tools/ntt_bench.py --field 1 --log-rows 20 --reps 3 > $O/bb_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/bb_write -- python3 tools/ntt_bench.py --field 1 --log-rows 20 --reps 3 > $O/bb_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/bb_sq -- python3 tools/ntt_bench.py --field 1 --log-rows 20 --reps 3 > $O/bb_sq.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/gl_sq -- python3 tools/ntt_bench.py --field 0 --log-rows 20 --reps 3 > $O/gl_sq.log 2>&1
python3 tools/pmc_summary.py $O/gl_sq $O/bb_sq > $O/sq_counters_ntt_passes.txt 2>&1 || true
rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -- python3 bench.py --inflight 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/trace1.log 2>&1 && python3 tools/trace_gaps.py $O/trace1 --dump > $O/single_proof_timeline.txt 2>&1 || true
ls $O
