"""N>1 path on CPU: world_size-2 `gloo` run of the frame sharder (videollama2_amd/dist.py): each rank encodes only its
slice, one all-gather reassembles the full [T, n, h] on every rank, ragged T included."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class FakeTower:
    num_patches, hidden_size, device = 5, 8, torch.device("cpu")

    def __init__(self):
        self.seen = []

    def __call__(self, frames):
        self.seen.append(frames.shape[0])
        # deterministic per-frame feature: depends only on the frame content
        return frames.reshape(frames.shape[0], -1)[:, :1].view(-1, 1, 1) * torch.ones(1, 5, 8) + torch.arange(8.0)


def _worker(rank, world, port, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from videollama2_amd.dist import FrameSharder
    frames = torch.arange(T, dtype=torch.float32).view(T, 1, 1, 1).expand(T, 3, 2, 2).contiguous()
    tower = FakeTower()
    out = FrameSharder().encode(tower, frames)
    ref = FakeTower()(frames)
    q.put((rank, torch.equal(out, ref), tower.seen, tuple(out.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T", [8, 5])
def test_frame_shard_allgather_world2(T):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, T, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    parts = {0: T - T // 2, 1: T // 2}
    for rank, ok, seen, shape in res:
        assert ok and shape == (T, 5, 8)
        assert seen == [parts[rank]]          # every rank ran the tower on its own slice only
