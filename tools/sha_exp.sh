set -e
for v in "" sha8 sha6 sha5; do
  lib=""; [ -n "$v" ] && lib="mini-stark_amd/libms_$v.so"
  MS_LIB=$lib timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --inflight 1 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_proof']; print('$v', round(d['value'],1), 'leaf', k['leaf_hash'], 'inner', k['inner_hash'])"
done
