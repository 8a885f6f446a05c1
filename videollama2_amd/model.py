"""VideoLLaMA2Hip -- the reference's inference surface for the two families its released checkpoints use (VideoLLaMA2:
CLIP-ViT + stc_connector + Mistral; VideoLLaMA2.1: SigLIP + stc_connector_v35 + Qwen2 -- picked from the config dict),
method for method (videollama2/model/videollama2_arch.py:99-263, videollama2_mistral.py:110-144), on the HIP path:
    encode_images_or_videos(images)                   arch.py:114-134   (+ temporal_aggregator :136-159)
    prepare_inputs_labels_for_multimodal(...)         arch.py:161-263   (inference subset: batch 1, no labels)
    generate(inputs, images=..., **hf_generate_kwargs) videollama2_mistral.py:110-144
Frames may be sharded over ranks (dist.py): each rank encodes its slice with the ViT and the visual tokens are
all-gathered (RCCL) in front of the connector."""
import torch
import torch.nn as nn

from . import _lib, ops
from .config import check_supported
from .connector import HipSTCConnector
from .constants import MODAL_INDEX_MAP, NUM_FRAMES
from .decoder import HipMistralDecoder
from .dist import FrameSharder
from .tower import HipCLIPVisionTower, HipSiglipVisionTower


class VideoLLaMA2Hip(nn.Module):
    def __init__(self, cfg, state_dict, device="cuda", max_seq_len=4096, image_processor=None, n_llm_layers=None,
                 mm_projector_type=None, sharder=None, tp_group=None):
        super().__init__()
        check_supported(cfg)
        mm_projector_type = mm_projector_type or cfg.get("projector", "stc_connector")
        if mm_projector_type not in ("stc_connector", "stc_connector_v35"):
            raise Exception(f"Unsupported projector type {mm_projector_type}!!!")     # arch.py:157 / projector.py:122
        self.cfg = cfg
        self.mm_projector_type = mm_projector_type
        self._dev = torch.device(device)
        tower_cls = HipSiglipVisionTower if cfg["vision"].get("family", "clip") == "siglip" else HipCLIPVisionTower   # encoder.py:157-160
        self.vision_tower = tower_cls(cfg, state_dict, device, image_processor=image_processor)
        self.mm_projector = HipSTCConnector(state_dict, device, padding=0 if mm_projector_type == "stc_connector_v35" else 1)
        self.decoder = HipMistralDecoder(cfg, state_dict, device, max_seq_len, n_llm_layers, tp_group=tp_group)
        self.sharder = sharder or FrameSharder()

    def get_vision_tower(self):
        return self.vision_tower

    @property
    def device(self):
        return self._dev

    def num_frames(self):
        return self.cfg.get("num_frames", NUM_FRAMES)

    # ---------------------------------------------------------------------------------- arch.py:114-134
    @torch.no_grad()
    def encode_images_or_videos(self, images):
        num_frames = self.num_frames()
        batch = []
        for data, modal in images:
            batch.append(data.expand(num_frames, -1, -1, -1) if modal == "image" else data)
        batch = batch[0].unsqueeze(0) if len(batch) == 1 else torch.stack(batch, dim=0)   # one video: a view, not a copy
        assert len(batch.size()) == 5                                                   # arch.py:127
        b, t = batch.shape[:2]
        frames = batch.reshape(b * t, *batch.shape[2:])                                  # 'b t c h w -> (b t) c h w'
        if b == 1 and self.sharder.world > 1:                                            # frame-parallel over the ranks
            return self.sharder.encode_video(self.vision_tower, self.mm_projector, frames)
        feats = self.vision_tower(frames)                                                # [(b t), n, h]
        feats = feats.view(b, t, *feats.shape[1:])                                       # '(b t) n h -> b t n h'
        return self.temporal_aggregator(feats)

    def temporal_aggregator(self, frames_features):                                      # arch.py:136-159
        if "tc_connector" in self.mm_projector_type:
            return self.mm_projector(frames_features)
        raise Exception(f"Unsupported projector type {self.mm_projector_type}!!!")

    # ---------------------------------------------------------------------------------- arch.py:161-263
    def _embed_ids(self, ids, out):
        self._check_ids(ids)
        if ids.numel():
            ops.embed_rows(ids.to(torch.int32).contiguous(), self.decoder.w["embed"], out)

    @torch.no_grad()
    def prepare_inputs_labels_for_multimodal(self, input_ids, attention_mask, past_key_values, labels, images, mm_features=None):
        """arch.py:161-263, inference subset (labels must be None).  Any batch size: every sequence's modal sentinels are replaced,
        in order, by the next block of `mm_features` (a sequence WITHOUT a sentinel still consumes one block, arch.py:178-191);
        ragged results are right-padded with zero rows (arch.py:227-236).  attention_mask: equal lengths -> ones for the inserted
        visual positions in FRONT of the given mask (arch.py:256-259); ragged -> True for a sequence's real rows, False for its
        padding (what arch.py:245-253 builds; the reference itself only reaches that code with labels).
        Returns (None, attention_mask, past_key_values, inputs_embeds [B, S, D] bf16, labels)."""
        if images is None or input_ids.shape[1] == 1:                                    # arch.py:166-169
            return input_ids, attention_mask, past_key_values, None, labels
        if labels is not None:
            raise NotImplementedError("HIP path: inference only (labels / loss belong to training, out of scope)")
        # Everything that needs the token ids on the HOST (sentinel positions, range check, media count) happens first, on one small
        # D2H copy, while the device queue is still empty: once the encoder is enqueued nothing below synchronises, so the
        # encoder, the embedding gathers and the prefill run back to back on the device.
        D = self.decoder.D
        B, L = input_ids.shape
        ids_host = input_ids.detach().cpu()
        sent = set(MODAL_INDEX_MAP.values())
        plans, need = [], 0
        n_media = len(mm_features) if mm_features is not None else len(images)
        for bi in range(B):
            row = ids_host[bi].tolist()
            mm_pos = [i for i, t in enumerate(row) if t in sent]
            text = [t for t in row if t not in sent]
            if need + max(len(mm_pos), 1) > n_media:
                raise ValueError(f"prompt {bi} holds {len(mm_pos)} modal tag(s) but only {n_media - need} media input(s) are left")
            if text and (min(text) < 0 or max(text) >= self.decoder.V):                    # what torch's embedding lookup raises in the reference
                raise IndexError(f"token id out of range: [{min(text)}, {max(text)}] vs vocab_size {self.decoder.V}")
            plans.append(mm_pos)
            need += max(len(mm_pos), 1)
        if mm_features is None:                       # (generate_batch hands in features it encoded for several requests at once)
            mm_features = self.encode_images_or_videos(images)
        embeds, cur_mm = [], 0
        for bi in range(B):
            mm_pos = plans[bi]
            ids32 = input_ids[bi].to(self._dev).clamp(min=0).to(torch.int32)
            if not mm_pos:                                                               # pure text: consumes one (unused) block
                emb = torch.empty((L, D), dtype=_lib.elem_dtype(), device=self._dev)
                if L:
                    ops.embed_rows(ids32.contiguous(), self.decoder.w["embed"], emb)
                embeds.append(emb)
                cur_mm += 1
                continue
            n_vis = sum(mm_features[cur_mm + k].shape[0] for k in range(len(mm_pos)))
            S = L - len(mm_pos) + n_vis
            emb = torch.empty((S, D), dtype=_lib.elem_dtype(), device=self._dev)
            cur, prev = 0, 0
            for k, p in enumerate(mm_pos + [L]):
                if p > prev:                                                             # text piece -> embed_tokens
                    ops.embed_rows(ids32[prev:p].contiguous(), self.decoder.w["embed"], emb[cur:cur + (p - prev)])
                    cur += p - prev
                if k < len(mm_pos):                                                      # visual block in place of the sentinel
                    f = mm_features[cur_mm].to(_lib.elem_dtype())
                    emb[cur:cur + f.shape[0]].copy_(f)
                    cur += f.shape[0]
                    cur_mm += 1
                prev = p + 1
            embeds.append(emb)
        lens = [e.shape[0] for e in embeds]
        self._splice_lens = lens                     # unpadded spliced length of every row (host-side; see _inputs_embeds)
        max_len = max(lens)
        if any(n != max_len for n in lens):                                              # arch.py:227-253
            out = torch.zeros((B, max_len, D), dtype=_lib.elem_dtype(), device=self._dev)
            for bi, e in enumerate(embeds):
                out[bi, :e.shape[0]].copy_(e)
            if attention_mask is not None:
                rows = []
                for bi, n in enumerate(lens):
                    left = torch.ones((n - L,), dtype=attention_mask.dtype, device=attention_mask.device)
                    right = torch.zeros((max_len - n,), dtype=attention_mask.dtype, device=attention_mask.device)
                    rows.append(torch.cat((left, attention_mask[bi], right), 0))
                attention_mask = torch.stack(rows, 0)
        else:
            out = torch.stack(embeds, 0) if B > 1 else embeds[0].unsqueeze(0)
            if attention_mask is not None:                                               # arch.py:256-259
                pad = torch.ones((B, max_len - L), dtype=attention_mask.dtype, device=attention_mask.device)
                attention_mask = torch.cat((pad, attention_mask), dim=1)
        return None, attention_mask, past_key_values, out, labels

    def _check_ids(self, ids):
        """embed_rows_kernel indexes the table with the ids unchecked: an id outside [0, vocab) (a leftover modal sentinel, a
        tokenizer / checkpoint mismatch) must fail here the way torch's embedding lookup does in the reference."""
        if ids.numel():
            lo, hi = int(ids.min()), int(ids.max())
            if lo < 0 or hi >= self.decoder.V:
                raise IndexError(f"token id out of range: [{lo}, {hi}] vs vocab_size {self.decoder.V}")

    @staticmethod
    def _valid_lengths(attention_mask, S, B):
        """Rows of the (spliced) attention mask -> real length per sequence.  HF masks padded KEYS and derives positions from the
        mask's cumulative sum, so a right-padded prompt decodes exactly like the unpadded one; that is how it runs here (the
        sequences go through the ragged batched decoder, each with its own cache and position).  Interior or left padding is not
        built."""
        if attention_mask is None:
            return [S] * B
        m = attention_mask.bool()
        lens = m.sum(1).tolist()
        for bi, n in enumerate(lens):
            if n == 0 or not bool(m[bi, :n].all()):
                raise NotImplementedError("HIP path: only right-padded prompts (attention_mask = 1...1 0...0) are supported")
        return lens

    def _inputs_embeds(self, inputs, attention_mask, images):
        """(inputs_embeds [B, S, D], real length of every row).  The attention mask is validated on the host BEFORE anything is
        enqueued (one small D2H while the queue is empty), so that encoder, splice and prefill run back to back on the device."""
        B, L = inputs.shape
        msum = None
        if attention_mask is not None:
            m = attention_mask.detach().cpu().bool()
            msum = m.sum(1).tolist()
            for bi, n in enumerate(msum):
                if n == 0 or not bool(m[bi, :n].all()):
                    raise NotImplementedError("HIP path: only right-padded prompts (attention_mask = 1...1 0...0) are supported")
        emb = None
        if images is not None:
            _, _, _, emb, _ = self.prepare_inputs_labels_for_multimodal(inputs, attention_mask, None, None, images)
        if emb is None:                            # no media, or input_ids.shape[1] == 1 (arch.py:166-169 hands the ids back)
            emb = torch.empty((B, L, self.decoder.D), dtype=_lib.elem_dtype(), device=self._dev)
            for bi in range(B):
                self._embed_ids(inputs[bi].to(self._dev), emb[bi])
            rows = [L] * B
        else:
            rows = self._splice_lens
        if msum is None:                           # no mask = every text token is real: the row's length is its own splice length
            return emb, list(rows)                 # (ragged media token counts pad the shorter rows with zeros: not prompt tokens)
        return emb, [rows[bi] - L + msum[bi] for bi in range(B)]        # the inserted visual rows + the row's real text tokens

    # ---------------------------------------------------------------------------------- videollama2_mistral.py:63-108
    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, images=None, return_dict=None, **kwargs):
        """One full forward (no cache reuse): logits of EVERY position, fp32 [B, S, V] (rows of a sequence's padding are zero).
        The inference half of the reference's `forward(..., images=)`: labels / loss, past_key_values, attentions and hidden
        states are the training / HF-generation plumbing and are not built (asked for -> NotImplementedError)."""
        if labels is not None or past_key_values is not None or output_attentions or output_hidden_states:
            raise NotImplementedError("HIP path forward(): logits only (no labels / past_key_values / attentions / hidden states)")
        if inputs_embeds is None:
            emb, lens = self._inputs_embeds(input_ids, attention_mask, images)
        else:
            emb = inputs_embeds.to(self._dev)
            lens = self._valid_lengths(attention_mask, emb.shape[1], emb.shape[0])
        B, S = emb.shape[:2]
        logits = torch.zeros((B, S, self.decoder.V), dtype=torch.float32, device=self._dev)
        for bi in range(B):
            logits[bi, :lens[bi]].copy_(self.decoder.prefill(emb[bi, :lens[bi]], return_all_logits=True))
        import types
        return types.SimpleNamespace(logits=logits, loss=None, labels=None, past_key_values=None)

    # ---------------------------------------------------------------------------------- videollama2_mistral.py:110-144
    @torch.no_grad()
    def generate(self, inputs=None, images=None, **kwargs):
        kwargs.pop("position_ids", None)
        attention_mask = kwargs.pop("attention_mask", None)
        if "inputs_embeds" in kwargs:
            raise NotImplementedError("`inputs_embeds` is not supported")                # videollama2_mistral.py:119-120
        sampler = None
        if kwargs.get("do_sample", False):
            # HF GenerationMixin.generate -> _get_logits_processor: temperature (default 1.0), top_k (generation_config default 50), top_p (default
            # 1.0) as warpers in that order, then one multinomial draw per step (videollama2/__init__.py:93-106 passes temperature and top_p)
            # An unspecified value comes from the checkpoint's generation_config when the model carries one (`self.generation_config`, set by the loader
            # from generation_config.json like HF's from_pretrained does), else from HF's GenerationConfig defaults.  Warpers this path does not build are
            # refused instead of silently skipped (a checkpoint that sets them would be sampled from another kept set than the reference's).
            gc = getattr(self, "generation_config", None)
            gcv = lambda k, dflt: getattr(gc, k, None) if (gc is not None and getattr(gc, k, None) is not None) else dflt
            for k, neutral in (("repetition_penalty", 1.0), ("min_p", None), ("typical_p", 1.0), ("epsilon_cutoff", 0.0), ("eta_cutoff", 0.0),
                               ("no_repeat_ngram_size", 0), ("num_beams", 1)):
                val = kwargs.get(k, gcv(k, neutral))
                if val is not None and val != neutral:
                    raise NotImplementedError(f"HIP path: do_sample with {k}={val} is not built (temperature / top_k / top_p are)")
            temperature, top_k, top_p = kwargs.get("temperature", gcv("temperature", 1.0)), kwargs.get("top_k", gcv("top_k", 50)), kwargs.get("top_p", gcv("top_p", 1.0))
            temperature = 1.0 if temperature is None else float(temperature)
            top_k, top_p = (0 if top_k is None else int(top_k)), (1.0 if top_p is None else float(top_p))
            if not temperature > 0.0:
                raise ValueError(f"`temperature` (={temperature}) has to be a strictly positive float")       # HF TemperatureLogitsWarper.__init__
            if top_k < 0 or not 0.0 < top_p <= 1.0:
                raise ValueError(f"`top_k` has to be >= 0 and `top_p` a float > 0 and <= 1, but are {top_k} / {top_p}")
            sampler = (temperature, top_k, top_p, kwargs.get("generator", None))
        if inputs.dim() == 1:
            inputs = inputs[None]
        emb, lens = self._inputs_embeds(inputs, attention_mask, images)
        max_new, eos = kwargs.get("max_new_tokens", 2048), kwargs.get("eos_token_id", None)
        if emb.shape[0] == 1:
            return self.decoder.generate(emb[0, :lens[0]], max_new_tokens=max_new, eos_token_id=eos,
                                         stopping_criteria=kwargs.get("stopping_criteria", None),
                                         return_logits=kwargs.get("return_logits", False), streamer=kwargs.get("streamer", None), sampler=sampler,
                                         # one captured hipGraph per token by default (what bench.py measures); `use_graph=False`
                                         # keeps the eager launch loop
                                         use_graph=kwargs.get("use_graph", self._dev.type == "cuda" and self.decoder.tp == 1))
        # batch > 1 (right-padded, arch.py:227-261): the sequences decode together, each on its own cache / position; finished rows
        # are filled with pad_token_id like HF's generate does
        if kwargs.get("stopping_criteria") is not None or kwargs.get("streamer") is not None or kwargs.get("return_logits"):
            raise NotImplementedError("HIP path: stopping_criteria / streamer / return_logits are built for batch 1")
        outs = self.decoder.generate_batch([emb[bi, :lens[bi]] for bi in range(emb.shape[0])], max_new_tokens=max_new, eos_token_id=eos, sampler=sampler)
        pad = kwargs.get("pad_token_id", None)
        pad = 0 if pad is None else int(pad)
        width = max(o.numel() for o in outs)
        res = torch.full((len(outs), width), pad, dtype=torch.long, device=self._dev)
        for bi, o in enumerate(outs):
            res[bi, :o.numel()] = o
        return res

    def batcher(self, max_slots=4, eos_token_id=None, use_graph=None):
        """Continuous batching over this model (serving.ModelBatcher): submit (input_ids, images) requests at any time, `step()`
        decodes one token for everything in flight, admission / retirement happen between steps."""
        from .serving import ModelBatcher
        return ModelBatcher(self, max_slots, eos_token_id, use_graph)

    @torch.no_grad()
    def generate_batch(self, requests, **kwargs):
        """Several requests decoded together (the reference serialises requests): `requests` is a list of
        (input_ids [1, L] or [L], images) pairs exactly as `generate(inputs, images=...)` takes them one at a time.  Every
        request is encoded, spliced and prefilled on its own; the decode steps then run for all of them at once
        (`HipMistralDecoder.generate_batch`).  Returns a list of LongTensor [n_new] with the NEW tokens of each request."""
        if kwargs.get("do_sample", False):
            raise NotImplementedError("HIP path implements greedy decoding (do_sample=False, the reference default)")
        # video requests with the same frame count are encoded in ONE tower call (frames are an independent batch dim), the
        # connector then runs per video; every kernel involved is row-independent, so a request's tokens are the same as alone
        feats = {}
        groups = {}
        for i, (ids, images) in enumerate(requests):
            if images is not None and len(images) == 1 and images[0][1] == "video":
                groups.setdefault(tuple(images[0][0].shape), []).append(i)
        for idxs in groups.values():
            if len(idxs) > 1 and self.sharder.world == 1:
                frames = torch.cat([requests[i][1][0][0] for i in idxs], 0)
                tower = self.vision_tower(frames.to(self._dev))
                t = tower.shape[0] // len(idxs)
                for j, i in enumerate(idxs):
                    feats[i] = self.mm_projector(tower[j * t:(j + 1) * t].unsqueeze(0))
        embeds = []
        for i, (ids, images) in enumerate(requests):
            ids = ids if ids.dim() == 2 else ids[None]
            if images is not None:
                _, _, _, emb, _ = self.prepare_inputs_labels_for_multimodal(ids, torch.ones_like(ids), None, None, images,
                                                                            mm_features=feats.get(i))
                embeds.append(emb[0])
            else:
                self._check_ids(ids[0])
                ids32 = ids[0].to(self._dev).to(torch.int32).contiguous()
                emb = torch.empty((ids32.numel(), self.decoder.D), dtype=_lib.elem_dtype(), device=self._dev)
                ops.embed_rows(ids32, self.decoder.w["embed"], emb)
                embeds.append(emb)
        return self.decoder.generate_batch(embeds, max_new_tokens=kwargs.get("max_new_tokens", 2048),
                                           eos_token_id=kwargs.get("eos_token_id", None),
                                           return_logits=kwargs.get("return_logits", False))

