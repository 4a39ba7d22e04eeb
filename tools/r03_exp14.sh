set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp14
mkdir -p $O
# 4 ranks sharing the one GPU (gloo; the process guard allows 6): replicas leg at 2^20 rows, then ONE 2^22-row proof sharded over the 4 ranks (sliced exchange through the callback)
timeout -k 10 600 python3 bench.py --gpus 4 --backend gloo --steps 5 --warmup 1 --inflight 2 --shard-log-rows 22 --shard-steps 2 --no-cpu-baseline --no-extras > $O/bench_gloo4.log 2> $O/bench_gloo4.err || { tail -20 $O/bench_gloo4.err; exit 1; }
python3 -c "
import json; d=json.loads([l for l in open('$O/bench_gloo4.log') if l.startswith('{')][-1]); print(d['n_gpus'], round(d['value'],1), d.get('launcher'), {k:v for k,v in d['sharded'].items() if k not in ('workload','parallelism')})"
