"""N>1 path on CPU: world_size 2, gloo, 127.0.0.1 — barrier, MAX-reduced timing and the all-gather of
per-rank commitments used by bench.py (mini-stark_amd/dist.py)."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_gloo():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "emu")], stdout=subprocess.DEVNULL)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(HERE, "dist_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["world"] == 2 and len(res["roots"]) == 2
    assert res["roots"][0] != res["roots"][1]  # different witnesses -> different commitments
    assert res["elapsed"] > 0
