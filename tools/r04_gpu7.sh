# r04 GPU call 7: the rank probe again (lazy lincomb in), and the whole `bench.py --gpus 4` flow on ONE GPU through gloo (4 ranks share the GPU; payloads staged through host
# memory): replicas leg + sharded leg with distributed round polynomials, proof on rank 0, matches_unsharded
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_gpu7
mkdir -p $O
step() { name=$1; shift; echo "== $name"; "$@"; rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then echo "stopping after $name"; exit $rc; fi; }
step rank_probe_24 timeout -k 10 600 python3 tools/shard_rank_probe.py --log-rows 24 --worlds 8 --dist 1 > $O/rank_probe_2p24.log 2>&1
cut -c1-1500 $O/rank_probe_2p24.log
step bench_gloo4 timeout -k 10 900 python3 bench.py --gpus 4 --backend gloo --steps 5 --warmup 1 --inflight 2 --shard-log-rows 22 --shard-steps 2 --no-cpu-baseline --no-extras > $O/bench_gloo4.json 2> $O/bench_gloo4.err
cut -c1-3000 $O/bench_gloo4.json; tail -5 $O/bench_gloo4.err
