# r04 GPU call 3: whole GPU suite on the final-ish library, the driver's bench line, the I/O probe with the SDMA defaults, and kernel traces of the I/O leg with one
# prover (hip copies vs SDMA: are the blit kernels gone?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_gpu3
mkdir -p $O
step() { name=$1; shift; echo "== $name"; "$@"; rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then echo "stopping after $name"; exit $rc; fi; }
step pytest_all timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_all.log 2>&1
tail -3 $O/pytest_all.log
step bench timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
cut -c1-400 $O/bench_default.json
step io_probe3 timeout -k 10 600 python3 tools/io_probe3.py --steps 20 --rounds 3 > $O/io_probe3.log 2> $O/io_probe3.err
tail -1 $O/io_probe3.log
step trace_io_sdma timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_io_sdma -- python3 tools/io_probe3.py --inflight 1 --steps 20 --rounds 1 --only async_sdma_upload_sdma > $O/trace_io_sdma.log 2>&1
step trace_io_hip timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_io_hip -- python3 tools/io_probe3.py --inflight 1 --steps 20 --rounds 1 --only async_hip > $O/trace_io_hip.log 2>&1
grep -h "copyBuffer\|fillBuffer" $O/trace_io_sdma/*/*kernel_stats.csv $O/trace_io_hip/*/*kernel_stats.csv
