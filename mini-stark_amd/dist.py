"""One-process-per-GPU helpers (torch.distributed; backend "nccl" is RCCL on ROCm).

Two ways to use N GPUs:
 * replicas (bench.py default): the path partitions over independent proofs, ranks never exchange data
   inside a proof; the only collectives are the timing barrier, a MAX-reduce of the elapsed time and an
   all-gather of each rank's final commitment (outside the timed region).
 * one proof sharded over the ranks (`ShardExchange`, ms_set_shard): the library partitions the
   evaluation-domain work of every large commitment and calls back here for the digest all-to-all, the
   subtree-root all-gather and the two small all-reduces of the query phase (include/ministark.h).
Covered on CPU by tests/test_dist_gloo.py and tests/test_shard_gloo.py (gloo, world_size 2, 4 and 8) and tests/test_bench_launcher.py (bench.py self-launched, world 2 and 8)."""
import os

import torch


class Group:
    def __init__(self, backend=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.backend = backend or "nccl"
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if self.backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend=self.backend)
            self.dist = dist
        elif self.backend == "nccl":
            torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank) if self.backend == "nccl" else torch.device("cpu")

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if self.backend == "nccl":
            torch.cuda.synchronize()

    def max_over_ranks(self, seconds: float) -> float:
        if self.dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_gather_bytes(self, data: bytes):
        """Every rank's `data` (equal lengths), rank order."""
        if self.dist is None:
            return [data]
        mine = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(self.device)
        out = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [bytes(t.cpu().numpy().tobytes()) for t in out]

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


class RcclShard:
    """One proof sharded over the ranks with the collectives INSIDE the library (ms_set_shard_rccl): rank 0's ncclUniqueId is
    broadcast with torch.distributed, every rank joins the library's own RCCL communicator; after that no Python runs in the loop."""

    def __init__(self, group: Group, ctx, cap_bytes: int):
        self.g, self.ctx = group, ctx
        uid = ctx.rccl_unique_id() if group.rank == 0 else bytes(128)
        if group.dist is not None:
            t = torch.frombuffer(bytearray(uid), dtype=torch.uint8).to(group.device)
            group.dist.broadcast(t, src=0)
            uid = bytes(t.cpu().numpy().tobytes())
        ctx.set_shard_rccl(group.rank, group.world, uid, cap_bytes)

    @property
    def calls(self):
        st = self.ctx.shard_stats()
        return {i: st[i] for i in range(4)}

    @property
    def bytes(self):
        return sum(self.ctx.shard_stats()[4:8])

    def close(self):
        self.ctx.set_shard_rccl(0, 1, bytes(128), 0)


class ShardExchange:
    """The exchange callback of ms_set_shard over torch.distributed.

    Owns the two exchange buffers (torch tensors on the group's device: HBM with "nccl" = RCCL, host memory with
    "gloo" on the kernel-emulation library) and runs the collective the library asks for on views of them.
    `staged=True` keeps the buffers on the GPU but moves the payload through host tensors for the collective
    (gloo with GPU contexts: used by the single-GPU rehearsal of the sharded path in tests/test_gpu_parity.py)."""

    def __init__(self, group: Group, ctx, cap_bytes: int, staged: bool = False, buffer_device=None):
        import torch.distributed as dist
        self.g, self.dist, self.ctx, self.staged = group, dist, ctx, staged
        dev = buffer_device if buffer_device is not None else group.device
        self.send = torch.zeros(cap_bytes, dtype=torch.uint8, device=dev)
        self.recv = torch.zeros(cap_bytes, dtype=torch.uint8, device=dev)
        self.calls = {0: 0, 1: 0, 2: 0, 3: 0}
        self.bytes = 0
        self.slices = 0
        ctx.set_shard(group.rank, group.world, self.send.data_ptr(), self.recv.data_ptr(), cap_bytes, self._exchange if group.world > 1 else None)

    def _exchange(self, op, nbytes):
        try:
            W, d = self.g.world, self.dist
            if op == 4:      # one slice of a sliced all-to-all: peer p's piece at offset + p * stride of both buffers
                import ctypes as C
                off, stride = C.c_size_t(0), C.c_size_t(0)
                self.ctx.check(self.ctx.L.ms_shard_slice_layout(self.ctx.h, C.byref(off), C.byref(stride)))
                off, stride = off.value, stride.value
                self.slices += 1
                self.bytes += nbytes * W
                a = torch.stack([self.send[off + p * stride: off + p * stride + nbytes] for p in range(W)]).cpu().contiguous().view(-1)
                b = torch.empty_like(a)
                d.all_to_all_single(b, a)
                b = b.view(W, nbytes).to(self.recv.device)
                for p in range(W):
                    self.recv[off + p * stride: off + p * stride + nbytes].copy_(b[p])
                if off == 0:
                    self.calls[0] += 1
                if self.send.is_cuda:
                    torch.cuda.synchronize()
                return 0
            if op == 5:      # gather to rank 0: chunk r of rank 0's receive buffer from rank r (ms_shard_proof_on_root)
                self.calls[1] += 1
                a = self.send[:nbytes].cpu() if self.staged else self.send[:nbytes]
                if self.g.rank == 0:
                    parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(W)]
                    d.gather(a.cpu(), parts, dst=0)
                    self.recv[: nbytes * W].copy_(torch.cat(parts))
                else:
                    d.gather(a.cpu(), None, dst=0)
                    self.bytes += nbytes
                if self.send.is_cuda:
                    torch.cuda.synchronize()
                return 0
            self.calls[op] += 1
            self.bytes += nbytes * (W if op == 0 else 1)
            if op == 0:      # all-to-all of W chunks
                a, b = self.send[: nbytes * W], self.recv[: nbytes * W]
            elif op == 1:    # all-gather
                a, b = self.send[:nbytes], self.recv[: nbytes * W]
            else:            # in-place all-reduce on send
                a, b = self.send[:nbytes], None
            if self.staged:
                ha = a.cpu()
                hb = torch.empty(b.numel(), dtype=torch.uint8) if b is not None else None
            else:
                ha, hb = a, b
            if op == 0:
                d.all_to_all_single(hb, ha)
            elif op == 1:
                d.all_gather_into_tensor(hb, ha) if not self.staged and self.g.backend == "nccl" else self._all_gather(hb, ha, W)
            elif op == 2:
                v = ha.view(torch.int64)  # u64 indices < 2^63 or ~0 (= -1): order them as unsigned through a sign flip
                v ^= torch.iinfo(torch.int64).min
                d.all_reduce(v, op=d.ReduceOp.MIN)
                v ^= torch.iinfo(torch.int64).min
            else:
                d.all_reduce(ha, op=d.ReduceOp.SUM)
            if self.staged:
                if b is not None:
                    b.copy_(hb)
                else:
                    a.copy_(ha)
            if self.send.is_cuda:
                torch.cuda.synchronize()
            return 0
        except Exception as e:  # the library turns a non-zero return into MS_ERR_HIP
            import traceback
            traceback.print_exc()
            return 1

    def _all_gather(self, out, mine, W):
        parts = list(out.view(W, -1).unbind(0))
        self.dist.all_gather(parts, mine)

    def close(self):
        self.ctx.set_shard(0, 1, 0, 0, 0, None)


class LocalShard:
    """The exchange callback of a ONE-rank "world" (env MS_SHARD_WORLD1=1 at ms_create; tests): every collective is a transfer to itself, done here with tensor copies
    - no torch.distributed, no second process - so that every kernel and buffer layout of the sharded prover runs inside one process, against the oracle.
    `rccl=True` instead joins a one-rank RCCL communicator inside the library (ms_set_shard_rccl): the RCCL calls themselves - grouped send / recv, all-gathers,
    all-reduces, the sliced exchange on its own stream - then execute through whole proofs on one GPU."""

    def __init__(self, ctx, cap_bytes: int, device=None, rccl: bool = False):
        self.ctx, self.rccl = ctx, rccl
        self.calls = {i: 0 for i in range(6)}
        if rccl:
            ctx.set_shard_rccl(0, 1, ctx.rccl_unique_id(), cap_bytes)
            return
        dev = device if device is not None else torch.device("cpu")
        self.send = torch.zeros(cap_bytes, dtype=torch.uint8, device=dev)
        self.recv = torch.zeros(cap_bytes, dtype=torch.uint8, device=dev)
        ctx.set_shard(0, 1, self.send.data_ptr(), self.recv.data_ptr(), cap_bytes, self._exchange)

    def _exchange(self, op, nbytes):
        import ctypes as C
        self.calls[op] += 1
        if op in (0, 1, 5):          # all-to-all / all-gather / gather of one rank: its own payload comes back
            self.recv[:nbytes].copy_(self.send[:nbytes])
        elif op == 4:                # one slice of the sliced all-to-all
            off, stride = C.c_size_t(0), C.c_size_t(0)
            self.ctx.check(self.ctx.L.ms_shard_slice_layout(self.ctx.h, C.byref(off), C.byref(stride)))
            self.recv[off.value: off.value + nbytes].copy_(self.send[off.value: off.value + nbytes])
        # 2, 3: all-reduces over one rank leave the buffer as it is
        if self.send.is_cuda:
            torch.cuda.synchronize()
        return 0

    def close(self):
        if self.rccl:
            self.ctx.set_shard_rccl(0, 1, bytes(128), 0)
        else:
            self.ctx.set_shard(0, 1, 0, 0, 0, None)
