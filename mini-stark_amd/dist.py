"""One-process-per-GPU helpers (torch.distributed; backend "nccl" is RCCL on ROCm).

The path partitions over independent proofs, so ranks never exchange data inside
a proof: the only collectives are the timing barrier, a MAX-reduce of the elapsed
time and an all-gather of each rank's final commitment (outside the timed region).
Covered on CPU by tests/test_dist_gloo.py (world_size 2, gloo)."""
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
