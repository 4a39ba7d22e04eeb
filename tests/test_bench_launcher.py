"""bench.py --gpus N must run N ranks however it is started: as a plain `python bench.py --gpus N` the script launches
its own ranks (fresh child processes; the parent never touches a GPU) and relays ONE JSON line; under torch.distributed.run it is one
rank.  Here on the CPU: --emu = the kernel-emulation build + gloo (a rehearsal of launcher, rank plumbing and the sharded leg's exchange;
its timings mean nothing).  The sharded leg also proves the same trace unsharded on rank 0 and reports `matches_unsharded`."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def run_bench(argv, launcher=None, timeout=900):
    subprocess.check_call(["make", "-C", os.path.join(HERE, "emu")], stdout=subprocess.DEVNULL)
    env = dict(os.environ, MS_SHARD_MIN_LEAVES="16")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MS_BENCH_LAUNCHED"):
        env.pop(k, None)
    cmd = [sys.executable] + (launcher or []) + [os.path.join(ROOT, "bench.py")] + argv
    cp = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert cp.returncode == 0, (cp.stdout[-2000:], cp.stderr[-4000:])
    lines = [l for l in cp.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    return json.loads(lines[0])


def check(res, world):
    assert res["n_gpus"] == world and res["scaling"] == "weak" and res["value"] > 0
    sh = res["sharded"]
    assert "error" not in sh, sh
    assert sh["ranks_in_communicator"] == world and sh["all_ranks_same_final_root"]
    assert sh["matches_unsharded"]["all"], sh["matches_unsharded"]
    cc = sh["collective_calls_per_rank"]
    assert cc["all_to_all"] >= 3 and cc["all_gather"] > cc["all_to_all"]   # the LDE and the large FRI rounds went through the digest exchange; r04: the coefficient-domain work adds small all-gathers
    assert sh["distributed_rounds"] >= 2 and sh["replicated_ms_estimate"] is not None and sh["partitioned_ms_estimate"] is not None


ARGV = ["--emu", "--log-rows", "7", "--steps", "1", "--warmup", "1", "--inflight", "1", "--shard-log-rows", "9", "--shard-steps", "1"]


def test_self_launched_two_ranks():
    res = run_bench(["--gpus", "2"] + ARGV)
    assert "self-launched" in res["launcher"]
    check(res, 2)


def test_under_torchrun_two_ranks():
    res = run_bench(["--gpus", "2"] + ARGV, launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29871"])
    assert "launcher" not in res
    check(res, 2)


def test_self_launched_eight_ranks():
    """world 8 - the target configuration - end to end: 8 replica ranks, then ONE 2^10-row proof sharded over 8 ranks vs the unsharded proof."""
    res = run_bench(["--gpus", "8", "--emu", "--log-rows", "6", "--steps", "1", "--warmup", "0", "--inflight", "1", "--shard-log-rows", "10", "--shard-steps", "1"], timeout=1200)
    check(res, 8)


def test_failing_rank_fails_the_launcher():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--emu", "--log-rows", "40"], env=env, capture_output=True, text=True, timeout=300)
    assert cp.returncode != 0 and not [l for l in cp.stdout.splitlines() if l.startswith("{")]


def test_a_failing_lane_fails_the_run():
    """A lane thread that dies (e.g. MS_ERR_NOMEM with too many proofs in flight) must fail the timed run - r04: eight 2^24-row provers ran out of HBM, their threads
    died, and the run still printed a (doubled) proofs/s figure.  Lanes.run re-raises the first lane error."""
    subprocess.check_call(["make", "-C", os.path.join(HERE, "emu")], stdout=subprocess.DEVNULL)
    sys.path.insert(0, ROOT)
    import torch
    import bench
    ln = bench.Lanes(0, 5, 8, 3, 0, torch.device("cpu"), lib=bench.EMU_LIB)
    ln.run(1)                                   # healthy: no error
    orig = ln._prove_n

    def flaky(i, n):
        if i == 1:
            raise RuntimeError("ministark error -7: out of memory")
        return orig(i, n)
    ln._prove_n = flaky
    with pytest.raises(RuntimeError, match="lane 1 of 3 failed"):
        ln.run(1)
    ln.close()
