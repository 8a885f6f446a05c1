"""Continuous batching: new requests are admitted BETWEEN decode steps, finished ones leave at once.

The reference's worker (videollama2/serve/model_worker.py:263-300 `generate_stream`, :350-352 the `limit_model_concurrency`
semaphore) lets several requests be in flight, each in its own `model.generate` thread with a TextIteratorStreamer; on one GPU
those generate loops serialise kernel by kernel.  Here the in-flight requests share ONE decode step per token: every request
owns a slot (its KV-cache slice, position and current token live on the device), a step streams the weights once for all
occupied slots (decoder._decode_kernels_batched), and between two steps
  * requests that hit EOS / a stop criterion / max_new_tokens / the cache end are retired (their streamer gets `end()`),
  * waiting requests are prefilled straight into a free slot's cache and join the next step.
With at most 4 slots the step runs the multi-row GEMV whose rows are bit-identical to the single-sequence step, so a request's
tokens do not depend on what else is in flight or when it was admitted (asserted in the tests)."""
import collections
import itertools

import torch


class _Request:
    __slots__ = ("rid", "x", "max_new", "streamer", "criteria", "tokens", "slot", "done")

    def __init__(self, rid, x, max_new, streamer, criteria):
        self.rid, self.x, self.max_new, self.streamer, self.criteria = rid, x, max_new, streamer, criteria
        self.tokens, self.slot, self.done = [], None, False


class ContinuousBatcher:
    """decoder: HipMistralDecoder.  max_slots: requests decoding together (<= 4 keeps every request bit-identical to its
    solo greedy decode; more switches the step to the skinny-M MFMA path, equal to bf16 rounding).
    use_graph (default: on a GPU): one captured hipGraph per occupancy (highest occupied slot + 1)."""

    def __init__(self, decoder, max_slots=4, eos_token_id=None, use_graph=None):
        if decoder.tp > 1:
            raise NotImplementedError("continuous batching is built for the single-GPU decoder")
        self.dec = decoder
        self.max_slots = int(max_slots)
        eos = eos_token_id
        self.eos = set() if eos is None else (set(eos) if isinstance(eos, (list, tuple, set)) else {int(eos)})
        self.use_graph = (decoder._dev.type == "cuda") if use_graph is None else bool(use_graph)
        self.waiting = collections.deque()
        self.slots = [None] * self.max_slots
        self.finished = {}
        self._ids = itertools.count()
        self.steps = 0
        self.bb = decoder._ensure_batch(self.max_slots, owner=self)

    # ---- admission
    def submit(self, inputs_embeds, max_new_tokens=2048, streamer=None, stopping_criteria=None):
        """Queue one request (inputs_embeds [S, D]: the spliced prompt of `prepare_inputs_labels_for_multimodal`).  It is prefilled
        and joins the batch at the next `step()` with a free slot.  Returns the request id."""
        S = inputs_embeds.shape[0]
        if S > self.dec.max_seq_len:
            raise ValueError(f"sequence length {S} exceeds the KV cache ({self.dec.max_seq_len})")
        crit = stopping_criteria
        if crit is not None and not isinstance(crit, (list, tuple)):
            crit = [crit]
        r = _Request(next(self._ids), inputs_embeds, min(int(max_new_tokens), self.dec.max_seq_len - S + 1), streamer, crit)
        self.waiting.append(r)
        return r.rid

    def _admit(self):
        bb, dec = self.bb, self.dec
        for slot in range(self.max_slots):
            if not self.waiting:
                return
            if self.slots[slot] is not None:
                continue
            r = self.waiting.popleft()
            cache = ([k[slot] for k in bb["k"]], [v[slot] for v in bb["v"]])
            own_pos = dec.pos                                                     # the decoder's own single-sequence state is not ours
            dec.prefill(r.x, cache=cache, logits_out=bb["logits"][slot])          # fills rows [0, S) of the slot's cache
            dec.pos = own_pos
            bb["pos"][slot:slot + 1].fill_(r.x.shape[0])
            r.slot, r.x = slot, None
            self.slots[slot] = r

    # ---- one decode step for everything in flight
    @torch.no_grad()
    def step(self):
        """Admit, run one step, retire.  Returns {request id: new token} for the requests that produced a token."""
        self.dec._ensure_batch(self.max_slots, owner=self)     # raises if ANOTHER batcher owns the slots with requests in flight
        self.dec._bb_busy = self                               # from here on generate_batch / a second batcher are refused while we hold requests
        if self.dec._bb is not self.bb:       # a larger generate_batch() on the same decoder reallocated the slot buffers (and caches)
            raise RuntimeError("the decoder's batch buffers were reallocated while requests were in flight: use one batcher per decoder "
                               "and do not call generate_batch() with more sequences than max_slots on it")
        self._admit()
        occupied = [s for s, r in enumerate(self.slots) if r is not None]
        if not occupied:
            return {}
        bb, dec = self.bb, self.dec
        nb = occupied[-1] + 1
        holes = [s for s in range(nb) if self.slots[s] is None]
        if holes:            # an empty slot below the highest occupied one still computes (and is ignored): park it at position 0
            bb["pos"][torch.tensor(holes, device=bb["pos"].device)] = 0
        if self.use_graph:
            dec.capture_batch_graph(nb).replay()          # argmax of every slot's logits + the forward of the new tokens
        else:
            dec._batched_step(nb)
        self.steps += 1
        toks = bb["tok"][:nb].tolist()                    # one small D2H per step for the stop checks (as HF's loop does)
        out = {}
        for s in occupied:
            r = self.slots[s]
            t = int(toks[s])
            r.tokens.append(t)
            out[r.rid] = t
            if r.streamer is not None:
                r.streamer.put(torch.tensor([[t]], dtype=torch.long))
            stop = t in self.eos or len(r.tokens) >= r.max_new
            if not stop and r.criteria is not None:
                ids = torch.tensor([r.tokens], dtype=torch.long)
                stop = any(bool(c(ids, None)) for c in r.criteria)
            if stop:
                self._retire(s)
        return out

    def _retire(self, slot):
        r = self.slots[slot]
        r.done = True
        if r.streamer is not None:
            r.streamer.end()
        self.finished[r.rid] = torch.tensor(r.tokens, dtype=torch.long, device=self.dec._dev)
        self.slots[slot] = None

    def in_flight(self):
        return sum(r is not None for r in self.slots) + len(self.waiting)

    def run(self):
        """Step until nothing is waiting or in flight.  Returns {request id: LongTensor [n_new]} of everything finished so far."""
        while self.in_flight():
            self.step()
        return self.finished


class ModelBatcher:
    """The same loop one level up: requests are (input_ids, images) pairs exactly as `VideoLLaMA2Hip.generate(inputs, images=)`
    takes them; the video is encoded and spliced at submission, the decode steps are shared."""

    def __init__(self, model, max_slots=4, eos_token_id=None, use_graph=None):
        self.model = model
        self.inner = ContinuousBatcher(model.decoder, max_slots, eos_token_id, use_graph)

    @torch.no_grad()
    def submit(self, input_ids, images=None, attention_mask=None, **kw):
        ids = input_ids if input_ids.dim() == 2 else input_ids[None]
        emb, lens = self.model._inputs_embeds(ids, attention_mask, images)
        return self.inner.submit(emb[0, :lens[0]], **kw)

    def step(self):
        return self.inner.step()

    def run(self):
        return self.inner.run()

    def in_flight(self):
        return self.inner.in_flight()

    @property
    def finished(self):
        return self.inner.finished
