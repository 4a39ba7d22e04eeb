set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp15
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cubic or degree3 or test_prove" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 600 python3 - > $O/cubic.log 2>&1 <<'PY'
import json, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import mini_stark_amd as ms
from mini_stark_amd.host import cubic_rows_native
from mini_stark_amd.synthetic import SplitMix64
P, lr, w = 2**64 - 2**32 + 1, 22, 64
N = 1 << lr
tr, sc = cubic_rows_native(P, N, w, 9)
d_tr = torch.from_numpy(tr.view(np.int64)).cuda()
spec = [(j, j, (j + 1) % w, (j + 2) % w, (j + 3) % w) for j in range(w)]
c2 = ms.Context(0)
omega = c2.root_of_unity(N)
rounds = lr + 4
times = {}
def st(name, fn):
    t = time.perf_counter(); r = fn(); c2.synchronize(); times[name] = times.get(name, 0) + time.perf_counter() - t; return r
for it in range(3):
    if it == 1: times.clear(); t0 = time.perf_counter()
    r2 = SplitMix64(6)
    c2.check(st('trace_commit', lambda: c2.trace_commit_device(d_tr.data_ptr(), N, w, w))[0]); c2.check(st('interpolate', c2.interpolate))
    c2.check(st('lde_commit', lambda: c2.lde_commit(8, r2.next() % P or 3, w))[0])
    c2.check(st('mix_cubic', lambda: c2.mix_cubic(r2.next() % P, spec, sc)))
    z = [r2.next() % P, r2.next() % P]; wz = [z[0] * omega % P, z[1] * omega % P]
    c2.check(st('eval_ext', lambda: c2.eval_ext(np.array([z, wz], dtype=np.uint64)))[0])
    c2.check(st('fri_begin', lambda: c2.fri_begin(8, rounds))[0])
    for _ in range(1, rounds):
        c2.check(st('fri_deep', lambda: c2.fri_deep([r2.next() % P, r2.next() % P]))[0]); c2.check(st('fri_fold_commit', lambda: c2.fri_fold_commit([r2.next() % P, r2.next() % P]))[0])
    c2.check(st('fri_query', lambda: c2.fri_query([r2.next()], read=False))[0])
el = (time.perf_counter() - t0) / 2
print(json.dumps({"proofs_per_s": 1 / el, "ms_per_proof": el * 1e3, "stage_ms": {k: round(v / 2 * 1e3, 2) for k, v in times.items()}}))
PY
cat $O/cubic.log | tail -3
