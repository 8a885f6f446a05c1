"""Frame-parallel encoder across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" IS
RCCL on ROCm, over xGMI).  The reference has no collectives at all: frames are an independent batch
dim through the ViT (videollama2_arch.py:130-132), so rank r encodes frames [r*T/R, (r+1)*T/R) and ONE all-gather of
visual tokens in front of the connector reassembles [T, 576, 1024] on every rank (BASELINE.json north_star).
world_size 1 takes the same code path (the collective is skipped, nothing else changes)."""
import torch
import torch.distributed as dist


class FrameSharder:
    def __init__(self, group=None):
        self.group = group

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    @property
    def rank(self):
        return dist.get_rank(self.group) if dist.is_available() and dist.is_initialized() else 0

    @staticmethod
    def split(T, world):
        """Contiguous, balanced split: first T % world ranks take one extra frame.  Returns [(start, count)]."""
        base, extra = divmod(T, world)
        out, s = [], 0
        for r in range(world):
            c = base + (1 if r < extra else 0)
            out.append((s, c))
            s += c
        return out

    def encode(self, tower, frames):
        """frames [T,3,H,W] present on EVERY rank (each rank only reads its slice) -> features [T, n, h] on every rank."""
        T = frames.shape[0]
        world, rank = self.world, self.rank
        if world == 1:
            return tower(frames)
        parts = self.split(T, world)
        s, c = parts[rank]
        maxc = max(p[1] for p in parts)
        local = tower(frames[s:s + c]) if c > 0 else None
        n, h = (local.shape[1], local.shape[2]) if local is not None else (tower.num_patches, tower.hidden_size)
        dtype = local.dtype if local is not None else frames.dtype
        dev = local.device if local is not None else tower.device
        send = torch.zeros((maxc, n, h), dtype=dtype, device=dev)          # equal-sized shards (ragged T padded)
        if c > 0:
            send[:c].copy_(local)
        recv = torch.empty((world * maxc, n, h), dtype=dtype, device=dev)
        dist.all_gather_into_tensor(recv, send, group=self.group)          # ONE collective: ncclAllGather over xGMI
        if all(p[1] == maxc for p in parts):
            return recv
        return torch.cat([recv[r * maxc:r * maxc + p[1]] for r, p in enumerate(parts)], 0)
