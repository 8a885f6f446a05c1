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


def _worker_sharded(rank, world, port, q, fixture):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import vl2_oracle as O
    from tests.emu.backend import emulated_backend
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture), weights_only=False)
    cfg = g["cfg"]
    sd = O.seeded_state_dict(cfg, g["seed"])
    with emulated_backend():
        from videollama2_amd.connector import HipSTCConnector
        from videollama2_amd.dist import FrameSharder
        from videollama2_amd.tower import HipCLIPVisionTower, HipSiglipVisionTower
        siglip = O.vision_family(cfg) == "siglip"
        tower = (HipSiglipVisionTower if siglip else HipCLIPVisionTower)(cfg, sd, "cpu")
        conn = HipSTCConnector(sd, "cpu", padding=O.conv3d_padding(cfg))
        sh = FrameSharder(cut="sharded_connector")
        assert sh.can_shard_connector(4)
        out = sh.encode_video(tower, conn, g["frames"])                     # sharded ViT + s1 + halo + conv3d/s2/readout
        ns = FrameSharder()                                                 # default cut = BASELINE.json's north_star
        assert ns.cut == "north_star"
        out_ns = ns.encode_video(tower, conn, g["frames"])                  # ViT sharded, all-gather of tower tokens, connector replicated
        assert out_ns.dtype == out.dtype and torch.equal(out_ns, out), (out_ns.dtype, out.dtype, (out_ns.float() - out.float()).abs().max().item())
        feats = tower(g["frames"])
        ref = conn(feats.view(1, *feats.shape))                             # single-process path, same kernels
    q.put((rank, torch.equal(out, ref), tuple(out.shape), ((out - g["mm_features"]).norm() / g["mm_features"].norm()).item()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fixture,shape", [("small_T4.pt", (1, 27, 256)), ("small_v21_T4.pt", (1, 8, 384))])
def test_sharded_connector_world2_bit_identical(fixture, shape):
    """The 'better cut' (SURVEY 8e): ViT + STC s1 sharded by frames, one-frame halo (none for the unpadded v35 sampler),
    Conv3d/s2/readout on the rank's output frames, all-gather of final tokens -- must equal the unsharded connector bit for
    bit (real kernel sources, emulated)."""
    from tests.emu.build_emu import build
    build()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_sharded, args=(r, 2, port, q, fixture)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(60)
    for rank, same, shp, err in res:
        assert same and shp == shape and err < 2.5e-2


def _worker_tp(rank, world, port, q, family):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import vl2_oracle as O
    from tests.emu.backend import emulated_backend
    from tests.util import rel
    cfg = O.config_small(4) if family == "mistral" else O.config_small_v21(4)
    cfg["llm"].update(hidden_size=512, num_attention_heads=4, num_key_value_heads=2, intermediate_size=512)   # 2 kv heads: TP=2
    keep = lambda n: n.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))
    sd = O.seeded_state_dict(cfg, 21, only=keep)
    x = (torch.randn(37, 512, generator=torch.Generator().manual_seed(3)) * 0.5).bfloat16().float()
    with torch.no_grad():
        toks, lg = O.greedy_generate(sd, cfg, x, 4)
    with emulated_backend():
        from videollama2_amd.decoder import HipMistralDecoder
        tp = HipMistralDecoder(cfg, sd, "cpu", max_seq_len=64, tp_group=dist.group.WORLD)
        assert tp.tp == world and tp.nh == 2 and tp.nkv == 1 and tp.w["layers"][0]["wo"].shape == (512, 256)
        out, logits = tp.generate(x, max_new_tokens=4, return_logits=True)
        one = HipMistralDecoder(cfg, sd, "cpu", max_seq_len=64)
        out1, logits1 = one.generate(x, max_new_tokens=4, return_logits=True)
    q.put((rank, out[0].tolist(), out1[0].tolist(), toks, rel(logits, lg), rel(logits1, lg), rel(logits, logits1.float())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("family", ["mistral", "qwen2"])
def test_tensor_parallel_decoder_world2(family):
    """SURVEY 8f row 3 (capability, not in the reference): the decoder sharded over 2 ranks by heads / MLP width, partial sums
    all-reduced -- prefill + 4 greedy steps must match the single-rank decoder and the fp32 oracle to the bf16 floor, with
    identical tokens on every rank."""
    from tests.emu.build_emu import build
    build()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_tp, args=(r, 2, port, q, family)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(60)
    assert res[0][1] == res[1][1]                               # every rank produced the same tokens
    for rank, tp_toks, one_toks, oracle_toks, e_tp, e_one, e_rel in res:
        assert e_tp < 2.5e-2 and e_one < 2.5e-2 and e_rel < 2.5e-2, (e_tp, e_one, e_rel)
        assert tp_toks == oracle_toks or tp_toks == one_toks
