# r04 GPU call 2: I/O leg split into halves (upload / read-back, HIP copy / SDMA engine), then ONE single-threaded profiled run long enough to wrap the AQL ring
# (> 16384 packets on one queue) - the experiment that separates "8 threads" from "ring wrap" for the r03 SIGSEGV
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_gpu2
mkdir -p $O
timeout -k 10 600 python3 tools/io_probe3.py --steps 20 --rounds 3 > $O/io_probe3.log 2> $O/io_probe3.err
rc=$?; echo "io_probe3 rc=$rc"; tail -1 $O/io_probe3.log
[ $rc -eq 0 ] || exit $rc
MS_BENCH_DUMP_MAPS=$O/maps_1lane.txt timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1_long -- python3 bench.py --inflight 1 --steps 90 --warmup 4 --no-cpu-baseline --no-extras > $O/stats1_long.log 2>&1
echo "rocprof 1-lane 94 proofs rc=$?"
tail -c 400 $O/stats1_long.log
