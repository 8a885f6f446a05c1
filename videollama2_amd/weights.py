"""HF state-dict -> kernel layouts, done once at load time (bf16 matrices in nn.Linear [N,K] layout, fp32 vectors).

Accepts both transformers-5.x key names (`...vision_tower.vision_tower.embeddings...`) and the 4.40-era names real
VideoLLaMA2 checkpoints carry (`...vision_tower.vision_tower.vision_model.embeddings...`); SURVEY.md 7.3-7."""
import torch

from . import _lib


def normalise_keys(sd):
    if getattr(sd, "lazy", False):            # generated on demand, already in 5.x naming: never materialise it
        return sd
    return {k.replace("vision_tower.vision_tower.vision_model.", "vision_tower.vision_tower."): v for k, v in sd.items()}


def _aligned(t):
    """The C ABI wants 16-byte aligned pointers; a tensor that came straight out of a memory-mapped checkpoint file (and
    needed no dtype / device conversion) may not be."""
    return t if t.data_ptr() % 16 == 0 else t.clone()


def _bf(t, dev):
    return _aligned(t.detach().to(device=dev, dtype=_lib.elem_dtype()).contiguous())


def _f32(t, dev):
    # parameters are stored in bf16 by the reference's bf16 path; bf16 -> fp32 is exact
    return _aligned(t.detach().to(dtype=_lib.elem_dtype()).to(device=dev, dtype=torch.float32).contiguous())


def fold_norm(w, g, b=None, c=None, dev=None):
    """The affine part of a LayerNorm / RMSNorm folded into the linear layer that follows it (csrc/k_gemm.h, "norm-carrying
    GEMMs"): Norm(x) W^T + c = rstd * (x W'^T - mean * s) + t with
        W' = bf16(W * g)  (per input column),   s[n] = sum_k W'[n][k]  (fp32, of the ROUNDED W': what the MFMA accumulates),
        t  = W b + c      (fp32; LayerNorm only -- RMSNorm has no shift).
    w [N, K], g / b [K], c [N] or None.  Returns (W' bf16 on dev, s fp32, t fp32 or None).  Parameters are taken as the bf16
    path stores them (rounded to bf16 first), so folding a real checkpoint and folding the oracle's seeded weights agree."""
    dev = w.device if dev is None else dev
    wf = w.detach().to(device=dev, dtype=_lib.elem_dtype()).float()
    wp = (wf * g.detach().to(device=dev, dtype=_lib.elem_dtype()).float()[None, :]).to(_lib.elem_dtype()).contiguous()
    s = wp.float().sum(1).contiguous()
    t = None
    if b is not None:
        t = wf @ b.detach().to(device=dev, dtype=_lib.elem_dtype()).float()
        if c is not None:
            t = t + c.detach().to(device=dev, dtype=_lib.elem_dtype()).float()
        t = t.contiguous()
    return _aligned(wp), _aligned(s), (None if t is None else _aligned(t))


def pack_tower(sd, cfg, dev, prefix="model.vision_tower.vision_tower."):
    """CLIPVisionModel weights (HF:models/clip/modeling_clip.py).  Only the layers feeding hidden_states[select_layer]."""
    sd = normalise_keys(sd)
    v = cfg["vision"]
    D, P = v["hidden_size"], v["patch_size"]
    L, sel = v["num_hidden_layers"], v["select_layer"]
    n_run = (L + 1 + sel) if sel < 0 else sel
    kreal = 3 * P * P
    kp = (kreal + 63) // 64 * 64
    pw = torch.zeros((D, kp), dtype=_lib.elem_dtype(), device=dev)
    pw[:, :kreal] = _bf(sd[prefix + "embeddings.patch_embedding.weight"].reshape(D, kreal), dev)
    pos = _bf(sd[prefix + "embeddings.position_embedding.weight"], dev)
    cls = _bf(sd[prefix + "embeddings.class_embedding"], dev)
    out = dict(kp=kp, n_run=n_run, patch_w=pw, pos=pos,
               cls_pos=(cls.float() + pos[0].float()).to(_lib.elem_dtype()).contiguous(),
               pre_w=_f32(sd[prefix + "pre_layrnorm.weight"], dev), pre_b=_f32(sd[prefix + "pre_layrnorm.bias"], dev),
               layers=[])
    for i in range(n_run):
        p = f"{prefix}encoder.layers.{i}."
        a = p + "self_attn."
        # layer_norm1 rides in the q/k/v GEMM, layer_norm2 in the fc1 GEMM (fold_norm): no standalone LayerNorm per layer
        wqkv, sqkv, bqkv = fold_norm(torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]], 0),
                                     sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"],
                                     torch.cat([sd[a + "q_proj.bias"], sd[a + "k_proj.bias"], sd[a + "v_proj.bias"]], 0), dev)
        w1, s1, b1 = fold_norm(sd[p + "mlp.fc1.weight"], sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], sd[p + "mlp.fc1.bias"], dev)
        out["layers"].append(dict(
            wqkv=wqkv, sqkv=sqkv, bqkv=bqkv, wo=_bf(sd[a + "out_proj.weight"], dev), bo=_f32(sd[a + "out_proj.bias"], dev),
            w1=w1, s1=s1, b1=b1, w2=_bf(sd[p + "mlp.fc2.weight"], dev), b2=_f32(sd[p + "mlp.fc2.bias"], dev)))
    return out


def padded_head_dim(hd):
    """The attention kernel is built for head_dim 64, 96 and 128: other head dims (SigLIP-so400m: 72 -> 96) run zero-padded."""
    if hd > 128:
        raise ValueError(f"vision head_dim {hd} > 128 is not supported")
    return 64 if hd <= 64 else 96 if hd <= 96 else 128


def pack_siglip_tower(sd, cfg, dev, prefix="model.vision_tower.vision_tower."):
    """SiglipVisionModel weights (HF:models/siglip/modeling_siglip.py) for the layers feeding hidden_states[select_layer].
    Two exact paddings make the so400m shapes fit the kernels (zeros in, zeros out):
      * heads: head_dim 72 -> 96.  q/k/v rows of every head are followed by zero rows (and zero bias), so q.k is unchanged
        and the padded v columns come out 0; out_proj gets zero COLUMNS at those positions.
      * MLP: 4304 -> 4352 (next multiple of 128).  fc1 gets zero rows + zero bias -> gelu_tanh(0) = 0 -> fc2's zero columns."""
    sd = normalise_keys(sd)
    v = cfg["vision"]
    D, P, H, I = v["hidden_size"], v["patch_size"], v["num_attention_heads"], v["intermediate_size"]
    L, sel = v["num_hidden_layers"], v["select_layer"]
    n_run = (L + 1 + sel) if sel < 0 else sel
    hd = D // H
    hdp = padded_head_dim(hd)
    Ip = (I + 127) // 128 * 128
    kreal = 3 * P * P
    kp = (kreal + 63) // 64 * 64
    pw = torch.zeros((D, kp), dtype=_lib.elem_dtype(), device=dev)
    pw[:, :kreal] = _bf(sd[prefix + "embeddings.patch_embedding.weight"].reshape(D, kreal), dev)

    def pad_rows(w):                                # [H*hd, ...] -> [H*hdp, ...], zero rows after every head
        w = w.detach().to(_lib.elem_dtype())
        out = torch.zeros((H, hdp) + tuple(w.shape[1:]), dtype=_lib.elem_dtype())
        out[:, :hd] = w.reshape((H, hd) + tuple(w.shape[1:]))
        return out.reshape((H * hdp,) + tuple(w.shape[1:]))

    out = dict(kp=kp, n_run=n_run, hd=hd, hdp=hdp, patch_w=pw, patch_b=_f32(sd[prefix + "embeddings.patch_embedding.bias"], dev),
               pos=_bf(sd[prefix + "embeddings.position_embedding.weight"], dev), layers=[])
    for i in range(n_run):
        p = f"{prefix}encoder.layers.{i}."
        a = p + "self_attn."
        wo = torch.zeros((D, H, hdp), dtype=_lib.elem_dtype())
        wo[:, :, :hd] = sd[a + "out_proj.weight"].detach().to(_lib.elem_dtype()).reshape(D, H, hd)
        w1 = torch.zeros((Ip, D), dtype=_lib.elem_dtype()); w1[:I] = sd[p + "mlp.fc1.weight"].detach().to(_lib.elem_dtype())
        b1 = torch.zeros((Ip,), dtype=_lib.elem_dtype()); b1[:I] = sd[p + "mlp.fc1.bias"].detach().to(_lib.elem_dtype())
        w2 = torch.zeros((D, Ip), dtype=_lib.elem_dtype()); w2[:, :I] = sd[p + "mlp.fc2.weight"].detach().to(_lib.elem_dtype())
        # layer norms folded into the q/k/v and fc1 GEMMs (fold_norm); the zero padding rows stay zero rows with zero shift
        wqkv, sqkv, bqkv = fold_norm(torch.cat([pad_rows(sd[a + n + "_proj.weight"]) for n in "qkv"], 0), sd[p + "layer_norm1.weight"],
                                     sd[p + "layer_norm1.bias"], torch.cat([pad_rows(sd[a + n + "_proj.bias"]) for n in "qkv"], 0), dev)
        w1f, s1, b1f = fold_norm(w1, sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], b1, dev)
        out["layers"].append(dict(
            wqkv=wqkv, sqkv=sqkv, bqkv=bqkv, wo=_bf(wo.reshape(D, H * hdp), dev), bo=_f32(sd[a + "out_proj.bias"], dev),
            w1=w1f, s1=s1, b1=b1f, w2=_bf(w2, dev), b2=_f32(sd[p + "mlp.fc2.bias"], dev)))
    return out


def _pack_bottleneck(sd, p, dev):
    C = sd[p + "conv1.conv.weight"].shape[0]
    blk = dict(
        conv1_w=_bf(sd[p + "conv1.conv.weight"].reshape(C, -1), dev),
        bn1_w=_f32(sd[p + "conv1.bn.weight"], dev), bn1_b=_f32(sd[p + "conv1.bn.bias"], dev),
        dw_w=_f32(sd[p + "conv2.conv.weight"].reshape(C, 9).t(), dev),             # [9][C] tap-major
        bn2_w=_f32(sd[p + "conv2.bn.weight"], dev), bn2_b=_f32(sd[p + "conv2.bn.bias"], dev),
        fc1_w=_bf(sd[p + "se.fc1.weight"].reshape(-1, C), dev), fc1_b=_f32(sd[p + "se.fc1.bias"], dev),
        fc2_w=_bf(sd[p + "se.fc2.weight"].reshape(C, -1), dev), fc2_b=_f32(sd[p + "se.fc2.bias"], dev),
        conv3_w=_bf(sd[p + "conv3.conv.weight"].reshape(C, C), dev),
        bn3_w=_f32(sd[p + "conv3.bn.weight"], dev), bn3_b=_f32(sd[p + "conv3.bn.bias"], dev))
    if (p + "downsample.conv.weight") in sd:
        blk.update(ds_w=_bf(sd[p + "downsample.conv.weight"].reshape(C, -1), dev),
                   dsbn_w=_f32(sd[p + "downsample.bn.weight"], dev), dsbn_b=_f32(sd[p + "downsample.bn.bias"], dev))
    return blk


def pack_connector(sd, dev, prefix="model.mm_projector."):
    """STCConnector weights (videollama2/model/projector.py:135-187); key names of timm RegStage."""
    w3 = sd[prefix + "sampler.0.weight"]                       # [Cout, Cin, 2, 2, 2]
    cout, cin = w3.shape[:2]
    return dict(
        s1=[_pack_bottleneck(sd, f"{prefix}s1.b{b}.", dev) for b in range(1, 5)],
        s2=[_pack_bottleneck(sd, f"{prefix}s2.b{b}.", dev) for b in range(1, 5)],
        samp_w=_bf(w3.permute(0, 2, 3, 4, 1).reshape(cout, 8 * cin), dev),   # K order = (kt, kh, kw, cin)
        samp_b=_f32(sd[prefix + "sampler.0.bias"], dev), cin=cin,
        ro0_w=_bf(sd[prefix + "readout.0.weight"], dev), ro0_b=_f32(sd[prefix + "readout.0.bias"], dev),
        ro2_w=_bf(sd[prefix + "readout.2.weight"], dev), ro2_b=_f32(sd[prefix + "readout.2.bias"], dev),
        zero_row=torch.zeros(cin, dtype=_lib.elem_dtype(), device=dev))


def pack_gate_up(gate, up):
    """[I,D] x2 -> [2I, D] in blocks of 64 rows {32 gate rows, 32 up rows} (VL2_GEMM_SWIGLU layout)."""
    I, D = gate.shape
    return torch.stack([gate.reshape(I // 32, 32, D), up.reshape(I // 32, 32, D)], 1).reshape(2 * I, D).contiguous()


def pack_decoder(sd, cfg, dev, n_layers=None, tp_rank=0, tp_size=1):
    """MistralForCausalLM / Qwen2ForCausalLM weights (HF:models/mistral/modeling_mistral.py, models/qwen2/modeling_qwen2.py:
    the same decoder, Qwen2 with a bias on q/k/v).
    tp_size > 1: this rank's Megatron-style shard -- q/k/v and gate/up split by output rows (whole heads; whole 32-row
    SwiGLU blocks), o_proj and down_proj split by input columns (their outputs are partial sums that the caller all-reduces);
    norms, embedding and lm_head replicated."""
    l = cfg["llm"]
    n_layers = l["num_hidden_layers"] if n_layers is None else n_layers
    nh, nkv, hd, I = l["num_attention_heads"], l["num_key_value_heads"], l["head_dim"], l["intermediate_size"]
    if nh % tp_size or nkv % tp_size or I % tp_size:
        raise ValueError(f"tensor-parallel degree {tp_size} does not divide heads {nh}/{nkv} or the MLP width {I}")
    q0, q1 = tp_rank * (nh // tp_size) * hd, (tp_rank + 1) * (nh // tp_size) * hd
    k0, k1 = tp_rank * (nkv // tp_size) * hd, (tp_rank + 1) * (nkv // tp_size) * hd
    i0, i1 = tp_rank * (I // tp_size), (tp_rank + 1) * (I // tp_size)
    Ip = (i1 - i0 + 63) // 64 * 64            # this rank's MLP slice, zero-padded to the GEMM's K granularity (Qwen2-72B at TP=8:
                                              # 29568 / 8 = 3696 -> 3712; silu(0) * 0 = 0 meets down_proj's zero columns: exact)

    def rows_padded(w):
        w = w[i0:i1]
        if Ip == i1 - i0:
            return w
        out = torch.zeros((Ip, w.shape[1]), dtype=w.dtype, device=w.device)
        out[:i1 - i0] = w
        return out

    def cols_padded(w):
        w = w[:, i0:i1]
        if Ip == i1 - i0:
            return w
        out = torch.zeros((w.shape[0], Ip), dtype=w.dtype, device=w.device)
        out[:, :i1 - i0] = w
        return out

    out = dict(embed=_bf(sd["model.embed_tokens.weight"], dev), norm_w=_f32(sd["model.norm.weight"], dev),
               lm_head=_bf(sd["lm_head.weight"], dev), layers=[],
               ones=torch.ones((l["hidden_size"],), dtype=torch.float32, device=dev))   # unit RMSNorm weight: the real ones are folded below
    for i in range(n_layers):
        p = f"model.layers.{i}."
        a = p + "self_attn."
        bqkv = None                                    # Qwen2Attention: bias on q/k/v (HF:models/qwen2/modeling_qwen2.py)
        if (a + "q_proj.bias") in sd:
            bqkv = _f32(torch.cat([sd[a + "q_proj.bias"][q0:q1], sd[a + "k_proj.bias"][k0:k1], sd[a + "v_proj.bias"][k0:k1]], 0), dev)
        # input_layernorm / post_attention_layernorm weights folded into the columns of q/k/v and gate/up (fold_norm): the
        # prefill GEMMs normalise in their epilogue from the statistics the previous GEMM emitted, the decode GEMVs normalise
        # x with a unit weight in their prologue
        wqkv, _, _ = fold_norm(torch.cat([sd[a + "q_proj.weight"][q0:q1], sd[a + "k_proj.weight"][k0:k1], sd[a + "v_proj.weight"][k0:k1]], 0),
                               sd[p + "input_layernorm.weight"], dev=dev)
        wgu, _, _ = fold_norm(pack_gate_up(rows_padded(sd[p + "mlp.gate_proj.weight"]), rows_padded(sd[p + "mlp.up_proj.weight"])),
                              sd[p + "post_attention_layernorm.weight"], dev=dev)
        out["layers"].append(dict(bqkv=bqkv, wqkv=wqkv, wo=_bf(sd[a + "o_proj.weight"][:, q0:q1], dev), wgu=wgu,
                                  wd=_bf(cols_padded(sd[p + "mlp.down_proj.weight"]), dev)))
    return out


def state_dict_names(cfg):
    """(name, shape) of every parameter the hot path reads, transformers-5.x key naming (what
    `Videollama2MistralForCausalLM(config).state_dict()` yields for CLIP + stc_connector + Mistral, or
    `Videollama2Qwen2ForCausalLM` for SigLIP + stc_connector_v35 + Qwen2, minus the SigLIP pooling head)."""
    v, l = cfg["vision"], cfg["llm"]
    Dv, Iv, P = v["hidden_size"], v["intermediate_size"], v["patch_size"]
    npos = (v["image_size"] // P) ** 2 + 1
    D, I = l["hidden_size"], l["intermediate_size"]
    hd, nh, nkv = l["head_dim"], l["num_attention_heads"], l["num_key_value_heads"]
    vt, mp = "model.vision_tower.vision_tower.", "model.mm_projector."
    if v.get("family", "clip") == "siglip":
        out = [(vt + "embeddings.patch_embedding.weight", (Dv, 3, P, P)), (vt + "embeddings.patch_embedding.bias", (Dv,)),
               (vt + "embeddings.position_embedding.weight", (npos - 1, Dv))]
    else:
        out = [(vt + "embeddings.class_embedding", (Dv,)), (vt + "embeddings.patch_embedding.weight", (Dv, 3, P, P)),
               (vt + "embeddings.position_embedding.weight", (npos, Dv)),
               (vt + "pre_layrnorm.weight", (Dv,)), (vt + "pre_layrnorm.bias", (Dv,))]
    for i in range(v["num_hidden_layers"]):
        p = f"{vt}encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            out += [(p + f"self_attn.{n}.weight", (Dv, Dv)), (p + f"self_attn.{n}.bias", (Dv,))]
        out += [(p + "layer_norm1.weight", (Dv,)), (p + "layer_norm1.bias", (Dv,)),
                (p + "mlp.fc1.weight", (Iv, Dv)), (p + "mlp.fc1.bias", (Iv,)),
                (p + "mlp.fc2.weight", (Dv, Iv)), (p + "mlp.fc2.bias", (Dv,)),
                (p + "layer_norm2.weight", (Dv,)), (p + "layer_norm2.bias", (Dv,))]
    for stage, cin in (("s1", Dv), ("s2", D)):
        for b in range(1, 5):
            ci = cin if b == 1 else D
            rd = int(round(ci * 0.25))
            p = f"{mp}{stage}.b{b}."
            out += [(p + "conv1.conv.weight", (D, ci, 1, 1)), (p + "conv1.bn.weight", (D,)), (p + "conv1.bn.bias", (D,)),
                    (p + "conv2.conv.weight", (D, 1, 3, 3)), (p + "conv2.bn.weight", (D,)), (p + "conv2.bn.bias", (D,)),
                    (p + "se.fc1.weight", (rd, D, 1, 1)), (p + "se.fc1.bias", (rd,)),
                    (p + "se.fc2.weight", (D, rd, 1, 1)), (p + "se.fc2.bias", (D,)),
                    (p + "conv3.conv.weight", (D, D, 1, 1)), (p + "conv3.bn.weight", (D,)), (p + "conv3.bn.bias", (D,))]
            if ci != D:
                out += [(p + "downsample.conv.weight", (D, ci, 1, 1)), (p + "downsample.bn.weight", (D,)),
                        (p + "downsample.bn.bias", (D,))]
    out += [(mp + "sampler.0.weight", (D, D, 2, 2, 2)), (mp + "sampler.0.bias", (D,)),
            (mp + "readout.0.weight", (D, D)), (mp + "readout.0.bias", (D,)),
            (mp + "readout.2.weight", (D, D)), (mp + "readout.2.bias", (D,)),
            ("model.embed_tokens.weight", (l["vocab_size"], D))]
    for i in range(l["num_hidden_layers"]):
        p = f"model.layers.{i}."
        out += [(p + "self_attn.q_proj.weight", (nh * hd, D)), (p + "self_attn.k_proj.weight", (nkv * hd, D)),
                (p + "self_attn.v_proj.weight", (nkv * hd, D)), (p + "self_attn.o_proj.weight", (D, nh * hd)),
                (p + "mlp.gate_proj.weight", (I, D)), (p + "mlp.up_proj.weight", (I, D)),
                (p + "mlp.down_proj.weight", (D, I)),
                (p + "input_layernorm.weight", (D,)), (p + "post_attention_layernorm.weight", (D,))]
        if l.get("family", "mistral") == "qwen2":
            out += [(p + "self_attn.q_proj.bias", (nh * hd,)), (p + "self_attn.k_proj.bias", (nkv * hd,)),
                    (p + "self_attn.v_proj.bias", (nkv * hd,))]
    out += [("model.norm.weight", (D,)), ("lm_head.weight", (l["vocab_size"], D))]
    return out


def random_state_dict(cfg, device, seed=1234, n_llm_layers=None):
    """Synthetic random-init weights created ON THE DEVICE (no checkpoints / hub on the target box): matrices
    ~ N(0, 1/fan_in), norm weights ~ 1 + 0.1 N, biases ~ 0.02 N.  bf16.  Used by bench.py and smoke()."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in state_dict_names(cfg):
        if n_llm_layers is not None and name.startswith("model.layers."):
            if int(name.split(".")[2]) >= n_llm_layers:
                continue
        x = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        leaf = name.split(".")[-1]
        is_norm = any(t in name for t in ("layernorm", "layer_norm", "layrnorm", ".bn.")) or name == "model.norm.weight"
        if leaf == "weight" and is_norm:
            x = 1.0 + 0.1 * x
        elif leaf == "bias":
            x = 0.02 * x
        elif len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            x = x * (fan_in ** -0.5)
        else:
            x = 0.5 * x
        sd[name] = x.to(_lib.elem_dtype())
    return sd


class LazyRandomStateDict:
    """`random_state_dict` without ever holding the model twice: every parameter is generated ON the device when it is indexed
    (its own generator, seeded by the parameter name) and dropped by the caller after packing.  What bench.py needs for the
    72B decoder: 145 GB of bf16 weights fit one MI355X (288 GB), the raw copy plus the packed copy would not."""
    lazy = True

    def __init__(self, cfg, device, seed=1234, n_llm_layers=None):
        self.device, self.seed = device, seed
        self.shapes = {}
        for name, shape in state_dict_names(cfg):
            if n_llm_layers is not None and name.startswith("model.layers.") and int(name.split(".")[2]) >= n_llm_layers:
                continue
            self.shapes[name] = shape

    def __contains__(self, name):
        return name in self.shapes

    def __len__(self):
        return len(self.shapes)

    def keys(self):
        return self.shapes.keys()

    def get(self, name, default=None):
        return self[name] if name in self.shapes else default

    def __getitem__(self, name):
        import hashlib
        shape = self.shapes[name]
        h = int.from_bytes(hashlib.sha256(f"{self.seed}:{name}".encode()).digest()[:7], "little")
        g = torch.Generator(device=self.device).manual_seed(h)
        x = torch.randn(shape, generator=g, device=self.device, dtype=torch.float32)
        leaf = name.split(".")[-1]
        is_norm = any(t in name for t in ("layernorm", "layer_norm", "layrnorm", ".bn.")) or name == "model.norm.weight"
        if leaf == "weight" and is_norm:
            x = 1.0 + 0.1 * x
        elif leaf == "bias":
            x = 0.02 * x
        elif len(shape) >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            x = x * (fan_in ** -0.5)
        else:
            x = 0.5 * x
        return x.to(_lib.elem_dtype())
