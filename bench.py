#!/usr/bin/env python3
"""bench.py — STARK proofs/s on the BASELINE.json workload, one process per GPU.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one complete proof of the hot path (src/starks.rs:59-169 between
"trace filled" and "StarkProof returned"): trace commit, INTT, coset LDE,
LDE commit, mix, DEEP-ALI evaluations, FRI commit phase (all rounds) and FRI
query phase, with the Fibonacci-AIR trace already resident in HBM and the FRI
proof left resident in HBM (PCIe-inclusive rate: see DESIGN.md).  Workload at
every N: BASELINE.json configs[1] — Fibonacci AIR, Goldilocks, 2^20 trace rows,
blowup 8, 20 security bits — one independent proof per step per rank (weak
scaling: the path partitions over proofs; no data-path collective).
`--mode shard` instead computes ONE proof per step with all ranks together
(ms_set_shard: RCCL digest all-to-all + root all-gather per large commitment;
strong scaling) — the latency configuration for single large proofs
(`--log-rows 24` = BASELINE.json configs[3]).

Rank 0 prints ONE JSON line.  Extra legs (rank 0, outside the timed region):
  roofline     — per-kernel HIP-event timings on the launching stream for the
                 NTT pass kernel vs its algorithmic bytes
  cpu_baseline — the CPU oracle ("port") on a bounded sample, N == 1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--log-rows", type=int, default=20, help="log2 of trace rows (BASELINE configs[1]: 20)")
    ap.add_argument("--blowup", type=int, default=8)
    ap.add_argument("--field", type=int, default=0, help="0 Goldilocks, 1 BabyBear")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log-rows", type=int, default=18)
    ap.add_argument("--inflight", type=int, default=None, help="independent proofs in flight per GPU (one ms_ctx + HIP stream each); a step = this many proofs (default: 8 up to 2^20 rows, fewer above)")
    ap.add_argument("--mode", choices=["replicas", "shard"], default="replicas",
                    help="replicas (default): every rank proves its own traces, no data-path collective (weak scaling).  shard: ONE proof per step computed by all "
                         "ranks together (ms_set_shard: coset-partitioned LDE/FRI + leaf hashing, RCCL digest all-to-all and root all-gather; strong scaling)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = RCCL, one GPU per rank (the measured configuration).  gloo: rehearsal of the N>1 paths on a box with fewer GPUs than ranks "
                         "(ranks share GPUs, exchange payloads are staged through host memory)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import mini_stark_amd as ms
    from mini_stark_amd.stark import StarkConfig, fibonacci_air
    from mini_stark_amd.host import HostStark  # C++ mirror of StarkConfig::new / Stark::prove above the C ABI

    from mini_stark_amd.dist import Group
    grp = Group(args.backend)
    world, rank, local_rank, dev = grp.world, grp.rank, grp.local_rank, grp.device
    if args.backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())
        dev = torch.device("cuda", local_rank)

    N = 1 << args.log_rows
    steps = N - 1  # "2^k trace rows" => steps = 2^k - 1 (quirk Q3)
    import threading
    if args.inflight is None:  # 8 fills the 4 hardware queues twice over at the benchmark size; larger proofs need the HBM (25 GiB each at 2^24 rows)
        args.inflight = 8 if args.log_rows <= 20 else (4 if args.log_rows == 21 else (3 if args.log_rows == 22 else 2))
    shard = args.mode == "shard" and world > 1
    C_IN = 1 if shard else max(1, args.inflight)
    # one context (own HIP stream, own HBM buffers) per in-flight proof; raises if libministark.so / the GPU is missing
    ctxs = [ms.Context(args.field, device=local_rank) for _ in range(C_IN)]
    ctx = ctxs[0]
    tts = [fibonacci_air(c, steps, secret_b=2 + (0 if shard else rank * C_IN) + i) for i, c in enumerate(ctxs)]  # shard: every rank holds the same trace
    cfg = StarkConfig(ctx, 20, args.blowup, steps, tts[0].constrain_number())
    starks = [HostStark(c, 20, args.blowup, steps, tts[0].constrain_number()) for c in ctxs]
    d_traces = [torch.from_numpy(t.data.view(np.int64)).to(dev) for t in tts]  # resident in HBM before the timed region
    torch.cuda.synchronize()
    last = [None] * C_IN
    xchg = None
    if shard:  # exchange buffers: the leaf digests of the largest commitment (32 B x L / world) + the query phase's Merkle paths
        from mini_stark_amd.dist import ShardExchange
        xchg = ShardExchange(grp, ctx, 32 * N * args.blowup // world + (4 << 20), staged=args.backend == "gloo", buffer_device=dev)

    def prove_n(i, n):
        for _ in range(n):
            ctxs[i].check(starks[i].prove_raw(tts[i], trace_device_ptr=d_traces[i].data_ptr(), read_fri_proof=False))
        last[i] = starks[i].last_proof(read_fri_proof=False)

    def run_steps(n):  # n steps = n proofs on each of the C_IN in-flight lanes (ctypes releases the GIL inside the library)
        if C_IN == 1:
            prove_n(0, n)
            return
        th = [threading.Thread(target=prove_n, args=(i, n)) for i in range(C_IN)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    def step():
        prove_n(0, 1)
        return last[0]

    run_steps(args.warmup)
    grp.barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    grp.barrier()
    elapsed = grp.max_over_ranks(time.perf_counter() - t0)
    proof = last[0]
    # every rank's final FRI root, gathered over RCCL (outside the timed region): all ranks finished a proof
    final_roots = grp.all_gather_bytes(proof.fri_roots[-1])
    assert len(final_roots) == world
    ms_per_step = elapsed / args.steps * 1e3
    value = (1 if shard else world) * args.steps * C_IN / elapsed

    out = {
        "metric": "stark_proofs_per_s", "value": value, "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None,
        "dtype": "u64" if args.field == 0 else "u32", "data": "synthetic",
        "config": {"workload": f"Fibonacci AIR, {'Goldilocks' if args.field == 0 else 'BabyBear+Fp4'}, 2^{args.log_rows} trace rows, blowup {args.blowup}, 20 security bits "
                               f"(w=3, c=6, rounds={cfg.rounds}, ood_queries={cfg.constrain_queries}, fri_queries={cfg.fri_queries}); a step = {C_IN} independent proofs in flight per GPU",
                   "proofs_per_step_per_gpu": C_IN, "parallelism": f"replicas x{world} GPUs x {C_IN} in-flight proofs (no data-path collective)"},
    }
    if shard:
        out["config"]["parallelism"] = f"one proof sharded over {world} GPUs (ms_set_shard): digest all-to-all + root all-gather per large commitment over RCCL"
        out["config"]["workload"] = out["config"]["workload"].split("; a step")[0] + f"; a step = 1 proof computed by {world} ranks together"
        out["config"]["proofs_per_step_per_gpu"] = 1.0 / world
        out["exchange"] = {"collective_calls_per_rank": {n: xchg.calls[i] for i, n in enumerate(["all_to_all", "all_gather", "all_reduce_min", "all_reduce_sum"])},
                           "bytes_through_callback_per_rank": xchg.bytes, "proofs": args.steps + args.warmup}

    if shard and rank != 0:
        step()  # the roofline leg's extra proof is collective in shard mode
    if rank == 0:
        # ---- roofline leg: per-kernel HIP events on the launching stream, one extra (untimed) proof
        buf = C.create_string_buffer(1 << 14)
        ctx.check(ctx.L.ms_profile_begin(ctx.h))
        step()
        ctx.check(ctx.L.ms_profile_end(ctx.h, buf, C.c_size_t(len(buf))))
        prof = json.loads(buf.value.decode())
        variants = prof.pop("ntt_pass_variants")
        # the NTT pass template instance with the largest total time in one proof = "the dominant HBM-class kernel"
        kname, k = max(variants.items(), key=lambda kv: kv[1]["ms"])
        avg_ms = k["ms"] / max(1, k["launches"])
        achieved = (k["alg_bytes"] / max(1, k["launches"])) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_ntt_pass.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                if pj.get("kernel") == kname:
                    traffic = pj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        allp = prof["ntt_pass"]
        out["roofline"] = {"kernel": kname, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": traffic, "launches_per_proof": k["launches"], "avg_launch_ms": avg_ms, "alg_bytes_per_launch": k["alg_bytes"] / max(1, k["launches"]),
                           "all_ntt_pass_kernels": {"launches_per_proof": allp["launches"], "ms_per_proof": allp["ms"],
                                                    "alg_GBps": allp["alg_bytes"] / (allp["ms"] * 1e-3) / 1e9 if allp["ms"] else 0.0},
                           "note": "pass kernels are integer-VALU-issue bound on MI355X (DESIGN.md 6.2): ~4.4 cycles per wave-instruction, 36.7 T int-op/s"}
        tot = sum(v["ms"] for v in prof.values()) or 1.0
        out["kernel_ms_per_proof"] = {n: round(v["ms"], 4) for n, v in prof.items() if v["launches"]}
        out["kernel_ms_total_single_proof"] = tot
        sha = prof["leaf_hash"]["ms"] + prof["inner_hash"]["ms"]
        out["sha256_share_of_kernel_time"] = sha / tot

        # ---- CPU baseline leg (N == 1): oracle "port", single thread, bounded sample
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import parity_cases as pc
            from common import fibonacci_trace_fast
            from oracle import oracle as orc
            cl = min(args.cpu_log_rows, args.log_rows)
            tr = fibonacci_trace_fast(args.field, 1 << cl)
            c0 = time.perf_counter()
            pc.drive(orc.Session(args.field), args.field, tr, args.blowup, max(0, cfg.fri_queries - 2), seed=1, q_ood=cfg.constrain_queries, read_big=False)
            ct = time.perf_counter() - c0
            scale = float(1 << (args.log_rows - cl))
            try:
                cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
            except Exception:
                cpu_model = "unknown"
            out["cpu_baseline"] = {"value": 1.0 / (ct * scale), "unit": "proofs/s", "cores": 1, "kind": "port", "cpu_model": cpu_model,
                                   "sample": f"one 2^{cl}-row proof of the same AIR on the oracle (oracle/ministark_oracle.cpp, 1 thread) took {ct:.2f} s; "
                                             f"scaled linearly x{int(scale)} to 2^{args.log_rows} rows (optimistic for the CPU: ignores the log factor)"}
            # the same sample with OpenMP over the oracle's independent loops (columns, leaf groups, tree levels):
            # what a rayon-enabled reference could reach on this host; the reference itself is single-threaded (README.md:33)
            ncpu = max(1, min(orc.max_threads(), len(os.sched_getaffinity(0))))
            best = None
            for nthr in sorted({t for t in (16, 32, 64, ncpu) if 1 < t <= ncpu}):  # the port's fork-join loops stop scaling well before 128 threads: keep the best
                orc.set_threads(nthr)
                c0 = time.perf_counter()
                pc.drive(orc.Session(args.field), args.field, tr, args.blowup, max(0, cfg.fri_queries - 2), seed=1, q_ood=cfg.constrain_queries, read_big=False)
                ctm = time.perf_counter() - c0
                if best is None or ctm < best[0]:
                    best = (ctm, nthr)
            orc.set_threads(1)
            if best is not None:
                out["cpu_baseline"]["all_cores"] = {"value": 1.0 / (best[0] * scale), "unit": "proofs/s", "cores": best[1], "kind": "port", "host_cpus": ncpu,
                                                    "sample": f"same 2^{cl}-row proof with OpenMP x{best[1]} (best of 16/32/64/{ncpu} threads) took {best[0]:.2f} s"}
        print(json.dumps(out), flush=True)
    grp.close()


if __name__ == "__main__":
    main()
