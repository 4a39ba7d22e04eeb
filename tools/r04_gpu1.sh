# r04 GPU call 1: the new parity cases, the read-back engines, the I/O probe, and ONE profiled 8-lane run with the process's mappings dumped (the r03 SIGSEGV's frames)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_gpu1
mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reference_script or other_blowups or padding_rows or readback_engines or async_proof_readback or arith_selftest" > $O/pytest_new.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_new.log
tail -3 $O/pytest_new.log
timeout -k 10 400 python3 tools/io_probe3.py --steps 20 --rounds 3 > $O/io_probe3.log 2> $O/io_probe3.err; echo "io_probe3 rc=$?"
cat $O/io_probe3.log
MS_BENCH_DUMP_MAPS=$O/maps_8lane.txt timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats8 -- python3 bench.py --no-cpu-baseline --no-extras > $O/stats8.log 2>&1; echo "rocprof 8-lane rc=$?"
tail -c 600 $O/stats8.log
