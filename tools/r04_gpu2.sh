# r04 GPU call 2 (second attempt; the first ran a library whose host and device halves were compiled from different header states - an edit during the build):
# the I/O leg split into halves, the sharded proofs with distributed coefficient work on the real kernels, ONE single-threaded profiled run long enough to wrap
# the AQL ring (> 16384 packets on one queue), and what one rank of a W-rank proof computes (stub exchange)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_gpu2
mkdir -p $O
step() { name=$1; shift; echo "== $name"; "$@"; rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ]; then echo "stopping after $name"; exit $rc; fi; }
step pytest_shard timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sharded or test_prove or error_codes or device_trace" -p no:cacheprovider > $O/pytest_shard.log 2>&1
tail -3 $O/pytest_shard.log
step io_probe3 timeout -k 10 600 python3 tools/io_probe3.py --steps 20 --rounds 3 > $O/io_probe3.log 2> $O/io_probe3.err
tail -1 $O/io_probe3.log
step rank_probe_20 timeout -k 10 300 python3 tools/shard_rank_probe.py --log-rows 20 --worlds 2 8 > $O/rank_probe_2p20.log 2>&1
step rank_probe_24 timeout -k 10 600 python3 tools/shard_rank_probe.py --log-rows 24 --worlds 2 4 8 > $O/rank_probe_2p24.log 2>&1
cat $O/rank_probe_2p24.log | cut -c1-600
MS_BENCH_DUMP_MAPS=$O/maps_1lane.txt timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1_long -- python3 bench.py --inflight 1 --steps 90 --warmup 4 --no-cpu-baseline --no-extras > $O/stats1_long.log 2>&1
echo "rocprof 1-lane 94 proofs rc=$?"
tail -c 400 $O/stats1_long.log
