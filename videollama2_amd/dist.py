"""Frame-parallel encoder across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" IS
RCCL on ROCm, over xGMI).  The reference has no collectives at all: frames are an independent batch
dim through the ViT (videollama2_arch.py:130-132), so rank r encodes frames [r*T/R, (r+1)*T/R) and ONE all-gather of
visual tokens in front of the connector reassembles [T, 576, 1024] on every rank (BASELINE.json north_star).
world_size 1 takes the same code path (the collective is skipped, nothing else changes)."""
import torch
import torch.distributed as dist

from . import _lib


class RankEncoderGraph:
    """The rank-local pieces of the sharded-connector cut as two captured hipGraphs (one per side of the halo exchange):
        A: local frames (static buffer) -> ViT -> STC stage s1            -> `s1` rows (static)
        B: s1 + halo (static) -> Conv3d + s2 + readout on this rank's output frames -> `tok` (static)
    At 2 frames per rank the ~330 launches of a rank's share take 3.2 ms on the GPU and 2.4 ms to enqueue one by one
    (profiles/r01_shard_model.jsonl): replayed from a graph the rank is GPU-bound and a step is 2 replays + 2 collectives.
    The collectives stay outside the graphs (eager RCCL calls on the same stream).  Same kernels, same order: bit-identical
    to the eager path (tests/test_gpu_rccl.py)."""

    def __init__(self, tower, connector, local_frames, T, rank, world):
        self.key = (tuple(local_frames.shape), local_frames.dtype, T, rank, world)
        self.frames = torch.empty_like(local_frames)
        self.frames.copy_(local_frames)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # eager warm-up: first-launch set-up, gather tables, allocator
            s1, n, self.in_dtype = FrameSharder.local_s1_of(tower, connector, self.frames)
            halo = torch.zeros((n, s1.shape[1]), dtype=s1.dtype, device=s1.device)
            FrameSharder.local_tokens(connector, s1, halo if rank > 0 else None, T, rank, world)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.n, self.halo = n, halo
        self.graph_a = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_a, capture_error_mode="thread_local"):
            self.s1 = FrameSharder.local_s1_of(tower, connector, self.frames)[0]
        self.graph_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_b, capture_error_mode="thread_local"):
            self.tok = FrameSharder.local_tokens(connector, self.s1, self.halo if rank > 0 else None, T, rank, world)

    def run_s1(self, local_frames):
        self.frames.copy_(local_frames)
        self.graph_a.replay()
        return self.s1

    def run_tokens(self):
        self.graph_b.replay()
        return self.tok


class LocalTensorParallel:
    """Every rank of a tensor-parallel decoder in ONE process, for boxes that reach one GPU (the decoder-side counterpart of
    `FrameSharder.encode_video_all_ranks_locally`): rank r's shard is a `HipMistralDecoder(..., tp_shard=(r, R), tp_local=self)` that
    runs in its own host thread, all of them on the same device and stream.  `reduce(rank, t)` is the all-reduce of the row-parallel
    projections (o_proj / down_proj partial sums): the ranks rendezvous, the partials are summed in RANK ORDER in fp32 and rounded
    once to the tensor's dtype, and every rank receives the same bits (what an RCCL all-reduce guarantees, though its ring adds in
    another order).  Validation only: the collectives of a real run are RCCL calls (`tp_group`)."""

    def __init__(self, size):
        import threading
        self.size = size
        self._bar = threading.Barrier(size)
        self._turn = threading.Lock()         # ONE rank enqueues at a time (the ranks hand the turn over at every reduction): the
        self._slots = [None] * size           # launch order on the shared stream is deterministic, and a host-side backend that is
        self._sum = None                      # not thread-safe (the CPU emulator of the tests) can sit underneath
        self._holds = [False] * size          # which rank's thread owns the turn right now
        self.reductions = 0

    def reduce(self, rank, t):
        self._slots[rank] = t
        self._holds[rank] = False
        self._turn.release()
        self._bar.wait()                      # every rank's partial is enqueued; nobody holds the turn
        if rank == 0:
            with self._turn:
                acc = self._slots[0].float().clone()
                for r in range(1, self.size):
                    acc += self._slots[r].float()
                self._sum = acc.to(t.dtype)
                self.reductions += 1
        self._bar.wait()
        self._turn.acquire()                  # keeps the turn until its next reduction (or the end of its program)
        self._holds[rank] = True
        t.copy_(self._sum)
        return t

    def run(self, fns):
        """fns[r]() is rank r's program (every rank must reach the same reductions); returns [fns[r]() for r], re-raises the first error."""
        import threading
        out, err = [None] * self.size, [None] * self.size

        def body(r):
            self._turn.acquire()
            self._holds[r] = True
            try:
                out[r] = fns[r]()
            except BaseException as exc:          # noqa: BLE001 -- a failed rank must not leave the others waiting forever
                err[r] = exc
                self._bar.abort()
            finally:
                if self._holds[r]:
                    self._holds[r] = False
                    self._turn.release()

        th = [threading.Thread(target=body, args=(r,)) for r in range(self.size)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        errs = [e for e in err if e is not None]
        if errs:
            real = [e for e in errs if type(e).__name__ != "BrokenBarrierError"]
            raise (real or errs)[0]
        return out


class FrameSharder:
    """cut: which of the two frame-parallel cuts `encode_video` takes ("keep the north_star cut as the default /
    baseline and report the alternative"):
        "north_star"        (default) BASELINE.json's own: ViT on the rank's frames, ONE all-gather of [T/R, 576, 1024] visual
                            tokens in front of the connector, connector replicated on every rank;
        "sharded_connector" everything per-frame stays sharded (ViT + STC s1, one-frame halo, Conv3d / s2 / readout on the rank's
                            output frames, all-gather of the final tokens); falls back to north_star when T does not split into an
                            even number of frames per rank."""
    CUTS = ("north_star", "sharded_connector")

    def __init__(self, group=None, use_graph=False, cut="north_star"):
        if cut not in self.CUTS:
            raise ValueError(f"cut {cut!r}: expected one of {self.CUTS}")
        self.group = group
        self.cut = cut
        self.use_graph = use_graph            # hipGraph replay of the rank-local encoder pieces (CUDA tensors only)
        self._graphs = {}
        self.graph_error = None               # why the graphs were abandoned for eager launches, if they were

    def rank_graph(self, tower, connector, local_frames, T, rank, world):
        key = (id(tower), id(connector), tuple(local_frames.shape), local_frames.dtype, T, rank, world)
        if key not in self._graphs:
            try:
                self._graphs[key] = RankEncoderGraph(tower, connector, local_frames, T, rank, world)
            except Exception as exc:          # a failed capture must not take the step down: same kernels, launched one by one
                self.use_graph, self.graph_error = False, repr(exc)[:300]
                torch.cuda.synchronize()
                return None
        return self._graphs[key]

    # ---- collectives.  RCCL takes device tensors directly; with the gloo backend (CPU tests, or the debug mode that lets
    #      several ranks share one GPU) device tensors are staged through host memory.
    def _host_staged(self, t):
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _all_gather(self, recv, send):
        if self._host_staged(send):
            r = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_gather_into_tensor(r, send.cpu(), group=self.group)
            recv.copy_(r)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)

    def _halo_exchange(self, send_to_next, recv_from_prev, rank, world):
        """Send `send_to_next` to rank+1 (if any) and fill `recv_from_prev` from rank-1 (if any)."""
        staged = self._host_staged(recv_from_prev)
        out_t = send_to_next.cpu() if staged else send_to_next
        in_t = torch.empty(recv_from_prev.shape, dtype=recv_from_prev.dtype) if staged else recv_from_prev
        reqs = []
        if rank + 1 < world:
            reqs.append(dist.P2POp(dist.isend, out_t, self._peer(rank + 1), self.group))
        if rank > 0:
            reqs.append(dist.P2POp(dist.irecv, in_t, self._peer(rank - 1), self.group))
        for r in (dist.batch_isend_irecv(reqs) if reqs else []):
            r.wait()
        if staged and rank > 0:
            recv_from_prev.copy_(in_t)

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
        # a rank without frames must send the dtype the other ranks' towers produce (uint8 ingest -> bf16 features)
        dtype = local.dtype if local is not None else (_lib.elem_dtype() if frames.dtype == torch.uint8 else frames.dtype)
        dev = local.device if local is not None else tower.device
        send = torch.zeros((maxc, n, h), dtype=dtype, device=dev)          # equal-sized shards (ragged T padded)
        if c > 0:
            send[:c].copy_(local)
        recv = torch.empty((world * maxc, n, h), dtype=dtype, device=dev)
        self._all_gather(recv, send)                                       # ONE collective: ncclAllGather over xGMI
        if all(p[1] == maxc for p in parts):
            return recv
        return torch.cat([recv[r * maxc:r * maxc + p[1]] for r, p in enumerate(parts)], 0)


    # ---------------------------------------------------------------------------------------------------------------
    def can_shard_connector(self, T):
        """The sharded-connector cut needs an even, equal number of frames per rank (Conv3d pairs frames 2k-1, 2k)."""
        w = self.world
        return w > 1 and T % w == 0 and (T // w) % 2 == 0

    def encode_video(self, tower, connector, frames):
        """frames [T,3,H,W] -> visual tokens [1, N_vis, D] on every rank.

        cut "north_star" (default), world 1, or a frame count that does not split evenly: the north-star cut -- ViT on the local
        frames, ONE all-gather of [T/R, 576, 1024] tokens, connector replicated (`encode` above + connector).
        cut "sharded_connector": the "better cut" of SURVEY.md 8(e): everything per-frame stays sharded -- ViT and STC stage s1 on the
        local frames, a one-frame halo (the last s1 frame goes to the next rank: Conv3d output `to` reads frames
        2to-1 and 2to), Conv3d + s2 + readout on this rank's output frames, then ONE all-gather of the final visual
        tokens (1.4 MB per rank at T=16).  Same arithmetic, same row order, bit-identical output."""
        T = frames.shape[0]
        if self.cut != "sharded_connector" or not self.can_shard_connector(T):
            feats = self.encode(tower, frames)
            return connector(feats.view(1, *feats.shape))
        world, rank = self.world, self.rank
        fpr = T // world
        rg = None
        if self.use_graph and frames.is_cuda:
            rg = self.rank_graph(tower, connector, frames[rank * fpr:(rank + 1) * fpr], T, rank, world)
        if rg is not None:
            s1, n, in_dtype = rg.run_s1(frames[rank * fpr:(rank + 1) * fpr]), rg.n, rg.in_dtype
        else:
            s1, n, in_dtype = self.local_s1(tower, connector, frames, rank, world)
        # halo: frame f0-1 comes from rank-1; our last frame goes to rank+1 (padding 1 only: the unpadded v35 sampler pairs
        # frames 2to, 2to+1, which an even frames-per-rank split never separates)
        halo = rg.halo if rg is not None else torch.empty((n, s1.shape[1]), dtype=s1.dtype, device=s1.device)
        if connector.padding == 1:
            self._halo_exchange(s1[s1.shape[0] - n:].contiguous(), halo, rank, world)
        tok = rg.run_tokens() if rg is not None else self.local_tokens(connector, s1, halo if rank > 0 else None, T, rank, world)
        extra = connector.padding                                  # padding 1: the last rank also owns output frame T/2
        per = tok.shape[0] // (T // world // 2 + (extra if rank == world - 1 else 0))
        max_rows = (T // world // 2 + extra) * per
        send = torch.zeros((max_rows, tok.shape[1]), dtype=tok.dtype, device=tok.device)
        send[:tok.shape[0]].copy_(tok)
        recv = torch.empty((world * max_rows, tok.shape[1]), dtype=tok.dtype, device=tok.device)
        self._all_gather(recv, send)
        return self._assemble(recv, T, world, per, max_rows, extra).to(in_dtype)

    # ---- the rank-local pieces of the sharded-connector cut (no communication inside: also driven rank by rank in ONE
    #      process by `encode_video_all_ranks_locally`, which is how the cut is validated and timed on a 1-GPU box)
    @staticmethod
    def local_s1(tower, connector, frames, rank, world):
        """ViT + STC stage s1 on this rank's frames -> (s1 rows [fpr*n, C], n tokens per frame, dtype of the tower output)."""
        fpr = frames.shape[0] // world
        return FrameSharder.local_s1_of(tower, connector, frames[rank * fpr:(rank + 1) * fpr])

    @staticmethod
    def local_s1_of(tower, connector, local_frames):
        fpr = local_frames.shape[0]
        local = tower(local_frames)                                          # [fpr, n, 1024]
        n = local.shape[1]
        rows = local.to(_lib.elem_dtype()).reshape(fpr * n, -1).contiguous()
        return connector.run_s1(rows, fpr, int(n ** 0.5)), n, local.dtype

    @staticmethod
    def local_tokens(connector, s1, halo, T, rank, world):
        """Conv3d + s2 + readout on this rank's output frames.  `halo` = s1 rows of frame f0-1 (None on rank 0: that
        frame is the Conv3d's zero padding and is never gathered).  Output frames: to in [f0/2, f0/2 + fpr/2); the
        last rank also owns to = T/2."""
        fpr = T // world
        n = s1.shape[0] // fpr
        hw = int(n ** 0.5)
        f0 = rank * fpr
        to0 = f0 // 2
        if connector.padding == 0:                                  # v35: output `to` reads frames 2to, 2to+1 -- all local
            samp, (nto, Ho, Wo) = connector.run_sampler(s1, T, hw, to_range=(to0, to0 + fpr // 2), frame_lo=f0, n_local=fpr)
            return connector.run_s2_readout(samp, nto, Ho, Wo)
        pool = torch.empty(((fpr + 1) * n, s1.shape[1]), dtype=s1.dtype, device=s1.device)
        pool[n:].copy_(s1)
        if halo is None:
            pool[:n].zero_()
        else:
            pool[:n].copy_(halo)
        to1 = to0 + fpr // 2 + (1 if rank == world - 1 else 0)
        samp, (nto, Ho, Wo) = connector.run_sampler(pool, T, hw, to_range=(to0, to1), frame_lo=f0 - 1, n_local=fpr + 1)
        return connector.run_s2_readout(samp, nto, Ho, Wo)                   # [nto*Ho*Wo, D]

    @staticmethod
    def _assemble(recv, T, world, per, max_rows, extra=1):
        fpr = T // world
        parts = [recv[r * max_rows:r * max_rows + (fpr // 2 + (extra if r == world - 1 else 0)) * per] for r in range(world)]
        return torch.cat(parts, 0).unsqueeze(0)

    @classmethod
    def encode_video_all_ranks_locally(cls, tower, connector, frames, world, timer=None, graphs=None):
        """Every rank's share of the sharded-connector cut executed one after the other in THIS process (halos passed by
        reference, concatenation instead of the all-gather).  Returns what `encode_video` returns on every rank; with
        `timer` (a callable returning a timestamp object after recording on the current stream) also the per-rank
        (vit+s1, conv3d+s2+readout) stamps."""
        T = frames.shape[0]
        assert T % world == 0 and (T // world) % 2 == 0
        s1s, stamps = [], []
        for r in range(world):
            t0 = timer() if timer else None
            fpr = T // world
            rg = graphs.rank_graph(tower, connector, frames[r * fpr:(r + 1) * fpr], T, r, world) if graphs is not None else None
            if rg is not None:                           # a FrameSharder(use_graph=True): every rank's pieces replayed from its graphs
                s1, n, in_dtype = rg.run_s1(frames[r * fpr:(r + 1) * fpr]), rg.n, rg.in_dtype
            else:
                s1, n, in_dtype = cls.local_s1(tower, connector, frames, r, world)
            s1s.append(s1)
            stamps.append([t0, timer() if timer else None])
        toks = []
        for r in range(world):
            t0 = timer() if timer else None
            halo = s1s[r - 1][s1s[r - 1].shape[0] - n:] if r > 0 else None
            rg = graphs.rank_graph(tower, connector, frames[r * fpr:(r + 1) * fpr], T, r, world) if graphs is not None else None
            if rg is not None:
                if halo is not None:
                    rg.halo.copy_(halo)
                toks.append(rg.run_tokens().clone())
            else:
                toks.append(cls.local_tokens(connector, s1s[r], halo, T, r, world))
            stamps[r] += [t0, timer() if timer else None]
        out = torch.cat(toks, 0).unsqueeze(0).to(in_dtype)
        return (out, stamps) if timer else out

    def _peer(self, group_rank):
        return dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank
