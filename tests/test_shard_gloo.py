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


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run_world(world, field, log_n, blowup, min_leaves, port, env=None, mode=""):
    port = _free_port()   # (the callers' fixed numbers collided when xdist ran two parametrisations of one test side by side: a rendezvous failure once in a few runs)
    subprocess.check_call(["make", "-C", os.path.join(HERE, "emu")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(HERE, "..", "oracle")], stdout=subprocess.DEVNULL)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "shard_worker.py"), str(field), str(log_n), str(blowup), str(min_leaves)] + ([mode] if mode else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("world,field,log_n,blowup", [(2, 0, 8, 8), (4, 0, 9, 8), (2, 1, 7, 8), (8, 0, 10, 8), (8, 1, 9, 8), (4, 1, 9, 16), (4, 0, 10, 2)])
def test_sharded_proof_matches_oracle(world, field, log_n, blowup):
    res = run_world(world, field, log_n, blowup, 16, 29800 + world * 10 + field * 3 + log_n)
    assert res["world"] == world
    calls = {int(k): v for k, v in res["calls"].items()}
    # the LDE commitment and the large FRI rounds went through the digest all-to-all + root all-gather, the query phase through one MIN and one SUM
    # all-reduce; r04: the coefficient-domain work is partitioned too - per distributed round two more small all-gathers (DEEP partial sums, the suffix
    # scan's carries), one for the raw-trace tree's roots, one for the DEEP-ALI partial sums, one for the query jobs' aggregates, and the proof slices
    assert calls[0] >= 3 and calls[1] >= calls[0] + 2 * (res["dist_rounds"] - 1) + 3 and calls[2] == 1 and calls[3] == 1
    assert res["dist_rounds"] >= 2


@pytest.mark.parametrize("world,field,log_n", [(4, 1, 8)])
def test_sharded_proof_with_replicated_coefficient_work(world, field, log_n):
    """MS_SHARD_DIST=0: the r03 scheme (only the evaluation-domain work of the commitments is partitioned) still gives the oracle's bytes."""
    res = run_world(world, field, log_n, 8, 16, 29600 + world * 10 + field * 3 + log_n, env={"MS_SHARD_DIST": "0"})
    calls = {int(k): v for k, v in res["calls"].items()}
    assert calls[0] >= 3 and calls[1] == calls[0] and calls[2] == 1 and calls[3] == 1 and res["dist_rounds"] == 0


@pytest.mark.parametrize("world,field,log_n", [(4, 0, 9), (8, 1, 9)])
def test_sharded_proof_assembled_on_rank0_only(world, field, log_n):
    """ms_shard_proof_on_root: the ranks' slices of the quotient polynomials are GATHERED to rank 0 (MS_XCHG_GATHER) instead of all-gathered; rank 0's proof
    equals the oracle's, the other ranks hold none; every other stage output is still identical on every rank."""
    res = run_world(world, field, log_n, 8, 16, 29500 + world * 10 + field * 3 + log_n, mode="root-only")
    assert res["world"] == world and res["root_only"] is True


@pytest.mark.parametrize("world,field,log_n", [(2, 0, 8), (4, 1, 8)][1:])
def test_sharded_proof_with_base_field_deep_points(world, field, log_n):
    """A DEEP point in the base field makes the evaluation-domain fold impossible (x^2 - z can vanish): the codeword of that round comes from a transform of
    the round polynomial, which for a DISTRIBUTED polynomial means gathering its parts first (round_commit's fallback)."""
    res = run_world(world, field, log_n, 8, 16, 29400 + world * 10 + field * 3 + log_n, mode="base-z")
    assert res["world"] == world


def test_small_proof_stays_replicated():
    res = run_world(2, 0, 5, 8, 1 << 20, 29877)
    calls = {int(k): v for k, v in res["calls"].items()}
    assert calls[0] == 0 and calls[1] == 0  # nothing reaches MS_SHARD_MIN_LEAVES: no commitment exchange


@pytest.mark.parametrize("world,field,log_n,slices", [(8, 0, 10, 4), (2, 1, 8, 8)])
def test_sliced_digest_exchange_matches_oracle(world, field, log_n, slices):
    """r03: large commitments hash their leaf groups in MS_SHARD_SLICES slices (slice s = the s-th part of every peer's chunk) so that the digests of a slice
    travel while the next slice is hashed (RCCL: on the context's communication stream; here: the callback, op ALL_TO_ALL_SLICE with the strided layout of
    ms_shard_slice_layout).  MS_SHARD_SLICE_MIN=1 forces the slicing at test sizes; every stage output must still equal the oracle's."""
    res = run_world(world, field, log_n, 8, 16, 29700 + world * 10 + field * 3 + log_n, env={"MS_SHARD_SLICES": str(slices), "MS_SHARD_SLICE_MIN": "1"})
    calls = {int(k): v for k, v in res["calls"].items()}
    assert res["slices"] >= slices and calls[0] >= 1 and calls[1] >= calls[0]


@pytest.mark.parametrize("world,field,log_n,env,mode", [(2, 1, 14, {}, ""), (4, 0, 14, {"MS_SHARD_GATHER_CHUNK": "4096"}, ""),
                                                       (2, 0, 12, {"MS_SHARD_GATHER_CHUNK": "1024"}, "root-only")])
def test_sharded_multilevel_scans_and_chunked_proof_gather(world, field, log_n, env, mode):
    """Ranks whose coefficient ranges exceed one block of the suffix scan (2048 elements: the carry-in from the higher ranks then travels down the scan's levels), and
    proof slices that cross the exchange buffer in several pieces (MS_SHARD_GATHER_CHUNK forces small pieces), all-gathered and gathered to rank 0."""
    res = run_world(world, field, log_n, 8, 16, 29300 + world * 10 + field * 3 + log_n + (5 if mode else 0), env=env, mode=mode)
    assert res["world"] == world and res["dist_rounds"] >= 8


@pytest.mark.parametrize("world,field,log_n", [(8, 0, 11), (2, 1, 8)])
def test_sharded_proof_with_half_empty_coefficient_ranges(world, field, log_n):
    """Trace polynomials of degree N/2 + 2: FRI's round-0 domain is sized for N coefficients but the polynomial has N/2 + 3, so the upper ranks' coefficient ranges are
    empty and one rank holds the ragged top - empty scan jobs, zero aggregates in the carry chain, slices of length zero in the proof gather."""
    res = run_world(world, field, log_n, 8, 16, 29200 + world * 10 + field * 3 + log_n, mode="low-degree")
    assert res["world"] == world and res["dist_rounds"] >= 2


@pytest.mark.parametrize("world,field,log_n", [(4, 0, 10), (2, 1, 9)])
def test_sharded_proof_with_launch_per_step_tail_rounds(world, field, log_n):
    """r05: the replicated rounds of a sharded proof run as fused rounds by default (csrc/fri_tail.hpp); MS_FRI_TAIL_MAX=0 keeps the launch-per-step path they replaced
    covered in sharded mode (replicated fold / scan / degree / tree launches between distributed rounds and the query phase)."""
    res = run_world(world, field, log_n, 8, 16, 29840 + world + field, env={"MS_FRI_TAIL_MAX": "0"})
    assert res["world"] == world and res["dist_rounds"] >= 2


@pytest.mark.parametrize("world,field,log_n", [(4, 0, 10), (2, 1, 8)])
def test_sharded_proof_with_latency_flag_settings(world, field, log_n):
    """r05: what MS_FLAG_LATENCY sets (side stream for a replicated round's coefficient side, polled results) inside a sharded proof: the top launch of a sharded tree
    carries the polled flag, stages that end in a collective keep the stream synchronisation - the oracle's bytes on every rank."""
    res = run_world(world, field, log_n, 8, 16, 0, env={"MS_FRI_OVERLAP": "1", "MS_SYNC_POLL": "1"})
    assert res["world"] == world and res["dist_rounds"] >= 2
