"""One proof sharded over several ranks (ms_set_shard) on CPU: kernel-emulation library, gloo,
127.0.0.1.  Each rank evaluates and hashes the leaf groups j = rank (mod world), digests go through an all-to-all,
subtree roots through an all-gather; the query phase finds leaves by value across ranks (MIN) and assembles the
Merkle paths from their owners (SUM).  The workers compare every stage output and the FRI proof with the oracle."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def run_world(world, field, log_n, blowup, min_leaves, port, env=None):
    subprocess.check_call(["make", "-C", os.path.join(HERE, "emu")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(HERE, "..", "oracle")], stdout=subprocess.DEVNULL)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "shard_worker.py"), str(field), str(log_n), str(blowup), str(min_leaves)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("world,field,log_n,blowup", [(2, 0, 8, 8), (4, 0, 9, 8), (2, 1, 7, 8), (4, 1, 8, 4), (2, 0, 6, 2), (8, 0, 10, 8), (8, 1, 9, 8)])
def test_sharded_proof_matches_oracle(world, field, log_n, blowup):
    res = run_world(world, field, log_n, blowup, 16, 29800 + world * 10 + field * 3 + log_n)
    assert res["world"] == world
    calls = {int(k): v for k, v in res["calls"].items()}
    # the LDE commitment and the large FRI rounds went through the digest all-to-all + root all-gather,
    # the query phase through one MIN and one SUM all-reduce
    assert calls[0] >= 3 and calls[1] == calls[0] and calls[2] == 1 and calls[3] == 1


def test_small_proof_stays_replicated():
    res = run_world(2, 0, 5, 8, 1 << 20, 29877)
    calls = {int(k): v for k, v in res["calls"].items()}
    assert calls[0] == 0 and calls[1] == 0  # nothing reaches MS_SHARD_MIN_LEAVES: no commitment exchange


@pytest.mark.parametrize("world,field,log_n,slices", [(2, 0, 8, 4), (4, 0, 9, 2), (8, 0, 10, 4), (2, 1, 8, 8)])
def test_sliced_digest_exchange_matches_oracle(world, field, log_n, slices):
    """r03: large commitments hash their leaf groups in MS_SHARD_SLICES slices (slice s = the s-th part of every peer's chunk) so that the digests of a slice
    travel while the next slice is hashed (RCCL: on the context's communication stream; here: the callback, op ALL_TO_ALL_SLICE with the strided layout of
    ms_shard_slice_layout).  MS_SHARD_SLICE_MIN=1 forces the slicing at test sizes; every stage output must still equal the oracle's."""
    res = run_world(world, field, log_n, 8, 16, 29700 + world * 10 + field * 3 + log_n, env={"MS_SHARD_SLICES": str(slices), "MS_SHARD_SLICE_MIN": "1"})
    calls = {int(k): v for k, v in res["calls"].items()}
    assert res["slices"] >= slices and calls[0] >= 1 and calls[1] >= calls[0]
