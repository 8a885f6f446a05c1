"""TEST INFRASTRUCTURE ONLY.  CPU restatement (plain PyTorch, fp32 by default) of the reference's
video-inference hot path, written as functions over a flat state-dict.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this file; the product path (videollama2_amd/) never does and
fails loudly when its HIP library is missing.

Pinning: this restatement is checked (tests/test_oracle_pin.py) against
  (a) the reference package itself, imported in place from /root/reference through oracle/ref_harness.py
      (only possible in the build container), and
  (b) golden tensors minted from that same reference run, committed under tests/golden/ by
      oracle/make_golden.py (travels to the GPU box).
The reference holds no tests / golden vectors of its own for this path, and the timm
part of the connector is a from-memory restatement of timm 1.0.3 => "parity unpinned" for that part with
respect to real timm; everything HF-side is pinned to transformers 5.15 eager kernels.

Every function cites the reference (or HF / timm) code it follows.  HF: = site-packages/transformers.
"""
import hashlib
import os
import math

import numpy as np
import torch
import torch.nn.functional as F

# ------------------------------------------------------------------------------------------ conv lowering switch
# CONV_AS_GEMM = True computes every convolution of the path through its GEMM form (unfold + F.linear; depthwise 3x3 as nine
# shifted multiply-adds accumulated in fp32 and rounded once): the same arithmetic a convolution library performs (fp32
# accumulate, one output rounding), without going through a convolution library.  Only used by the parity tests when they
# run this restatement in bf16 ON THE GPU (torch-ROCm ops) to measure the reference's own bf16 noise floor: MIOpen's
# bf16 Conv3d / depthwise paths may JIT-search for minutes on a fresh box.  The fp32 truth runs with the plain F.conv*.
CONV_AS_GEMM = False


def _conv1x1(x_nchw, w):
    """F.conv2d(x, w[Cout,Cin,1,1]) (no bias)."""
    if not CONV_AS_GEMM:
        return F.conv2d(x_nchw, w)
    return F.linear(x_nchw.permute(0, 2, 3, 1), w.flatten(1)).permute(0, 3, 1, 2)


def _patch_conv(x, w, b, stride):
    """F.conv2d(x, w[Cout,3,P,P], b, stride=P) with P == kernel size (non-overlapping patches)."""
    if not CONV_AS_GEMM:
        return F.conv2d(x, w, b, stride=stride)
    T, C, H, W = x.shape
    P = stride
    gh, gw = H // P, W // P
    cols = x[:, :, :gh * P, :gw * P].reshape(T, C, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(T, gh * gw, C * P * P)
    return F.linear(cols, w.flatten(1), b).transpose(1, 2).reshape(T, -1, gh, gw)


def _dwconv3x3(x_nchw, w):
    """F.conv2d(x, w[C,1,3,3], padding=1, groups=C)."""
    if not CONV_AS_GEMM:
        return F.conv2d(x_nchw, w, padding=1, groups=x_nchw.shape[1])
    xp = F.pad(x_nchw.float(), (1, 1, 1, 1))
    H, W = x_nchw.shape[2:]
    acc = torch.zeros_like(x_nchw, dtype=torch.float32)
    for ky in range(3):
        for kx in range(3):
            acc += xp[:, :, ky:ky + H, kx:kx + W] * w[:, 0, ky, kx].float().view(1, -1, 1, 1)
    return acc.to(x_nchw.dtype)


def _conv3d_k2s2(x_ncthw, w, b, padding):
    """F.conv3d(x, w[Cout,Cin,2,2,2], b, stride=2, padding=p)."""
    if not CONV_AS_GEMM:
        return F.conv3d(x_ncthw, w, b, stride=2, padding=padding)
    p = padding
    xp = F.pad(x_ncthw, (p, p, p, p, p, p))
    B, C, T, H, W = xp.shape
    To, Ho, Wo = (T - 2) // 2 + 1, (H - 2) // 2 + 1, (W - 2) // 2 + 1
    xp = xp[:, :, :2 * To, :2 * Ho, :2 * Wo].reshape(B, C, To, 2, Ho, 2, Wo, 2)
    cols = xp.permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, To * Ho * Wo, C * 8)          # K order (cin, kt, kh, kw) = w.flatten(1)
    return F.linear(cols, w.flatten(1), b).transpose(1, 2).reshape(B, -1, To, Ho, Wo)


# ------------------------------------------------------------------------------------------------ configs


def config_videollama2_7b(num_frames=16):
    """VideoLLaMA2-7B public hyper-parameters (SURVEY.md section 8 header)."""
    return dict(
        vision=dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                    image_size=336, patch_size=14, layer_norm_eps=1e-5, select_layer=-2),
        llm=dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                 num_key_value_heads=8, head_dim=128, vocab_size=32000, rms_norm_eps=1e-5, rope_theta=1e6),
        num_frames=num_frames)


def config_small(num_frames=4):
    """Smallest config every HIP kernel accepts (ViT head_dim 64, LLM head_dim 128, GEMM N%128, K%64).
    56x56 frames -> 4x4 patches (+CLS = 17 tokens); STC: 4x4 -> 3x3, t -> t/2+1."""
    return dict(
        vision=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                    image_size=56, patch_size=14, layer_norm_eps=1e-5, select_layer=-2),
        llm=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                 num_key_value_heads=1, head_dim=128, vocab_size=512, rms_norm_eps=1e-5, rope_theta=1e6),
        num_frames=num_frames)


def config_videollama2_1_7b_16f(num_frames=16):
    """VideoLLaMA2.1-7B-16F public hyper-parameters: SigLIP-so400m-patch14-384 tower
    (encoder.py:84-151), stc_connector_v35 (projector.py:225-238), Qwen2-7B decoder (videollama2_qwen2.py)."""
    return dict(
        vision=dict(family="siglip", hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16,
                    image_size=384, patch_size=14, layer_norm_eps=1e-6, select_layer=-2),
        llm=dict(family="qwen2", hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28,
                 num_key_value_heads=4, head_dim=128, vocab_size=152064, rms_norm_eps=1e-6, rope_theta=1e6),
        projector="stc_connector_v35", num_frames=num_frames)


def config_small_v21(num_frames=4):
    """Small VideoLLaMA2.1-shaped config: SigLIP tower whose head_dim (32) and MLP width (200) are NOT multiples the HIP
    kernels take directly (exercises the head/MLP padding the real 72 / 4304 need), v35 connector (Conv3d padding 0:
    4x4 -> 2x2, t -> t/2), Qwen2 decoder with QKV bias and an odd GQA group (3 q heads per kv head)."""
    return dict(
        vision=dict(family="siglip", hidden_size=128, intermediate_size=200, num_hidden_layers=3, num_attention_heads=4,
                    image_size=56, patch_size=14, layer_norm_eps=1e-6, select_layer=-2),
        llm=dict(family="qwen2", hidden_size=384, intermediate_size=512, num_hidden_layers=2, num_attention_heads=3,
                 num_key_value_heads=1, head_dim=128, vocab_size=512, rms_norm_eps=1e-6, rope_theta=1e6),
        projector="stc_connector_v35", num_frames=num_frames)


def vision_family(cfg):
    return cfg["vision"].get("family", "clip")


def llm_family(cfg):
    return cfg["llm"].get("family", "mistral")


def conv3d_padding(cfg):
    """STCConnector pads the Conv3d sampler by 1 (projector.py:164-174), STCConnectorV35 by 0 (projector.py:225-238)."""
    return 0 if cfg.get("projector", "stc_connector") == "stc_connector_v35" else 1


# ------------------------------------------------------------------------------- seeded synthetic weights


def seeded_tensor(name, shape, seed=1234, lm_head_scale=1.0):
    """Deterministic tensor keyed by parameter NAME (sha256 -> numpy Generator), independent of creation
    order, so oracle, reference model and HIP path can all rebuild identical weights without sharing files."""
    h = int.from_bytes(hashlib.sha256(f"{seed}:{name}".encode()).digest()[:8], "little")
    rng = np.random.default_rng(h)
    n = int(np.prod(shape)) if len(shape) else 1
    x = torch.from_numpy(rng.standard_normal(n, dtype=np.float32)).reshape(shape)
    leaf = name.split(".")[-1]
    is_norm = any(t in name for t in ("layernorm", "layer_norm", "layrnorm", ".bn.", ".norm.")) or name.endswith("model.norm.weight")
    if leaf == "weight" and is_norm:
        return 1.0 + 0.1 * x
    if leaf == "bias":
        return 0.02 * x
    if "class_embedding" in name:
        return 0.5 * x
    if "position_embedding" in name:
        return 0.1 * x
    if "embed_tokens" in name:
        return 0.5 * x
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        s = 1.0 / math.sqrt(fan_in)
        if name.startswith("lm_head"):
            s *= lm_head_scale
        return s * x
    return x


def state_dict_names(cfg):
    """(name, shape) of every floating-point parameter of Videollama2MistralForCausalLM for `cfg`, in the key
    naming transformers 5.x produces (no `vision_model.` level).  Verified against the real
    reference model in tests/test_oracle_pin.py."""
    v, l = cfg["vision"], cfg["llm"]
    Dv, Iv, P = v["hidden_size"], v["intermediate_size"], v["patch_size"]
    npos = (v["image_size"] // P) ** 2 + 1
    D, I = l["hidden_size"], l["intermediate_size"]
    hd, nh, nkv = l["head_dim"], l["num_attention_heads"], l["num_key_value_heads"]
    out = []
    vt = "model.vision_tower.vision_tower."
    siglip = vision_family(cfg) == "siglip"
    if siglip:      # HF:models/siglip/modeling_siglip.py SiglipVisionEmbeddings: biased patch conv, no CLS, no pre-LN
        out += [(vt + "embeddings.patch_embedding.weight", (Dv, 3, P, P)), (vt + "embeddings.patch_embedding.bias", (Dv,)),
                (vt + "embeddings.position_embedding.weight", (npos - 1, Dv))]
    else:
        out += [(vt + "embeddings.class_embedding", (Dv,)),
                (vt + "embeddings.patch_embedding.weight", (Dv, 3, P, P)),
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
    out += [(vt + "post_layernorm.weight", (Dv,)), (vt + "post_layernorm.bias", (Dv,))]
    if siglip:      # SiglipMultiheadAttentionPoolingHead: exists in the module, never reached by hidden_states[-2]
        out += [(vt + "head.probe", (1, 1, Dv)), (vt + "head.attention.in_proj_weight", (3 * Dv, Dv)),
                (vt + "head.attention.in_proj_bias", (3 * Dv,)), (vt + "head.attention.out_proj.weight", (Dv, Dv)),
                (vt + "head.attention.out_proj.bias", (Dv,)), (vt + "head.layernorm.weight", (Dv,)),
                (vt + "head.layernorm.bias", (Dv,)), (vt + "head.mlp.fc1.weight", (Iv, Dv)), (vt + "head.mlp.fc1.bias", (Iv,)),
                (vt + "head.mlp.fc2.weight", (Dv, Iv)), (vt + "head.mlp.fc2.bias", (Dv,))]
    mp = "model.mm_projector."
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
            (mp + "readout.2.weight", (D, D)), (mp + "readout.2.bias", (D,))]
    out += [("model.embed_tokens.weight", (l["vocab_size"], D))]
    for i in range(l["num_hidden_layers"]):
        p = f"model.layers.{i}."
        out += [(p + "self_attn.q_proj.weight", (nh * hd, D)), (p + "self_attn.k_proj.weight", (nkv * hd, D)),
                (p + "self_attn.v_proj.weight", (nkv * hd, D)), (p + "self_attn.o_proj.weight", (D, nh * hd)),
                (p + "mlp.gate_proj.weight", (I, D)), (p + "mlp.up_proj.weight", (I, D)),
                (p + "mlp.down_proj.weight", (D, I)),
                (p + "input_layernorm.weight", (D,)), (p + "post_attention_layernorm.weight", (D,))]
        if llm_family(cfg) == "qwen2":      # HF:models/qwen2/modeling_qwen2.py Qwen2Attention: bias on q/k/v only
            out += [(p + "self_attn.q_proj.bias", (nh * hd,)), (p + "self_attn.k_proj.bias", (nkv * hd,)),
                    (p + "self_attn.v_proj.bias", (nkv * hd,))]
    out += [("model.norm.weight", (D,)), ("lm_head.weight", (l["vocab_size"], D))]
    return out


def seeded_state_dict(cfg, seed=1234, lm_head_scale=1.0, only=None, round_bf16=True):
    """Build the synthetic weights.  `only`: optional predicate(name) to build a subset (full 7B is 29 GB fp32).
    round_bf16: round once to bf16 ('round to bf16 once; the same tensors feed oracle and kernels')."""
    todo = [(name, shape) for name, shape in state_dict_names(cfg) if only is None or only(name)]

    def make(item):
        t = seeded_tensor(item[0], item[1], seed, lm_head_scale)
        return t.bfloat16().float() if round_bf16 else t

    # every tensor has its OWN name-keyed generator, so the tensors can be drawn in parallel without changing a bit (numpy's Generator
    # releases the GIL): the 7.2 B numbers of the full-size case take ~65 s on one thread
    if sum(int(np.prod(sh)) if len(sh) else 1 for _, sh in todo) > 50_000_000:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            vals = list(ex.map(make, todo))
    else:
        vals = [make(it) for it in todo]
    return {name: v for (name, _), v in zip(todo, vals)}


def normalise_keys(sd):
    """Accept transformers-4.40-era checkpoints (`...vision_tower.vision_model.embeddings...`) as well as 5.x."""
    return {k.replace("vision_tower.vision_tower.vision_model.", "vision_tower.vision_tower."): v for k, v in sd.items()}


# -------------------------------------------------------------------------------------- a1: preprocessing

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def frame_sample_uniform(duration, num_frames):
    """videollama2/mm_utils.py:106-121 ('uniform' branch): centre of num_frames equal segments, round(x+1e-6)."""
    seg = float(duration - 1) / num_frames
    ids = [(seg * i + seg * (i + 1)) / 2 for i in range(num_frames)]
    return np.round(np.array(ids) + 1e-6).astype(int)


def normalise_frames_u8(frames_u8_thwc):
    """The arithmetic tail of HF CLIPImageProcessor.preprocess for frames that are already image_size^2 RGB
    (resize/crop are identities then): x/255 -> (x-mean)/std -> NCHW fp32.  mm_utils.py:199-201;
    HF:models/clip/image_processing_*: rescale_factor 1/255, image_mean/std above."""
    x = torch.from_numpy(np.ascontiguousarray(frames_u8_thwc)).float() * (1.0 / 255.0)
    mean = torch.tensor(CLIP_MEAN).view(1, 1, 1, 3)
    std = torch.tensor(CLIP_STD).view(1, 1, 1, 3)
    return ((x - mean) / std).permute(0, 3, 1, 2).contiguous()


# ---------------------------------------------------------------------------------------- a6: CLIP tower

_VT = "model.vision_tower.vision_tower."


def clip_embeddings(sd, cfg, pixel_values):
    """HF:models/clip/modeling_clip.py CLIPVisionEmbeddings.forward (patch conv, CLS, + position_embedding)
    followed by `pre_layrnorm` (CLIPVisionTransformer.forward).  Returns hidden_states[0]  [T, N+1, D]."""
    v = cfg["vision"]
    w = sd[_VT + "embeddings.patch_embedding.weight"]
    x = _patch_conv(pixel_values.to(w.dtype), w, None, v["patch_size"])
    x = x.flatten(2).transpose(1, 2)
    cls = sd[_VT + "embeddings.class_embedding"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1) + sd[_VT + "embeddings.position_embedding.weight"].unsqueeze(0)
    return F.layer_norm(x, (x.shape[-1],), sd[_VT + "pre_layrnorm.weight"], sd[_VT + "pre_layrnorm.bias"],
                        v["layer_norm_eps"])


def quick_gelu(x):
    """HF:activations.py QuickGELUActivation: x * sigmoid(1.702 x)."""
    return x * torch.sigmoid(1.702 * x)


def clip_attention(sd, cfg, p, x):
    """HF:modeling_clip.py CLIPAttention.forward + eager_attention_forward (softmax in fp32, scale d^-0.5)."""
    v = cfg["vision"]
    B, N, D = x.shape
    H = v["num_attention_heads"]
    hd = D // H
    q = F.linear(x, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]).view(B, N, H, hd).transpose(1, 2)
    k = F.linear(x, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"]).view(B, N, H, hd).transpose(1, 2)
    vv = F.linear(x, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"]).view(B, N, H, hd).transpose(1, 2)
    a = torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5)
    a = F.softmax(a, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(a, vv).transpose(1, 2).reshape(B, N, D)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def clip_layer(sd, cfg, i, x):
    """HF:modeling_clip.py CLIPEncoderLayer.forward (pre-LN; attn; +res; LN; fc1; quick_gelu; fc2; +res)."""
    eps = cfg["vision"]["layer_norm_eps"]
    p = f"{_VT}encoder.layers.{i}."
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], eps)
    x = x + clip_attention(sd, cfg, p + "self_attn.", h)
    h = F.layer_norm(x, (D,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], eps)
    h = F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    h = F.linear(quick_gelu(h), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + h


def clip_tower(sd, cfg, frames, return_hidden=False):
    """videollama2/model/encoder.py:41-53 CLIPVisionTower.forward + feature_select (:31-39):
    hidden_states[select_layer][:, 1:] cast back to the input dtype.  hidden_states[0] is the post-pre_layrnorm
    embedding, hidden_states[i] the output of layer i; select_layer=-2 -> output of layer L-1 (layer L and
    post_layernorm are computed by HF and discarded; not computed here)."""
    v = cfg["vision"]
    L = v["num_hidden_layers"]
    sel = v["select_layer"]
    n_run = (L + 1 + sel) if sel < 0 else sel  # hidden_states has L+1 entries
    x = clip_embeddings(sd, cfg, frames)
    hs = [x]
    for i in range(n_run):
        x = clip_layer(sd, cfg, i, x)
        hs.append(x)
    out = x[:, 1:].to(frames.dtype)
    return (out, hs) if return_hidden else out


# ------------------------------------------------------------------------------------- 8f-1: SigLIP tower

SIGLIP_MEAN = (0.5, 0.5, 0.5)
SIGLIP_STD = (0.5, 0.5, 0.5)


def normalise_frames_u8_siglip(frames_u8_thwc):
    """Arithmetic tail of SiglipImageProcessor for frames already image_size^2: x/255 -> (x-0.5)/0.5 -> NCHW fp32."""
    x = torch.from_numpy(np.ascontiguousarray(frames_u8_thwc)).float() * (1.0 / 255.0)
    return ((x - 0.5) / 0.5).permute(0, 3, 1, 2).contiguous()


def gelu_tanh(x):
    """HF:activations.py `gelu_pytorch_tanh` = nn.GELU(approximate='tanh')."""
    return F.gelu(x, approximate="tanh")


def siglip_embeddings(sd, cfg, pixel_values):
    """HF:models/siglip/modeling_siglip.py SiglipVisionEmbeddings.forward: biased patch conv (valid padding), flatten,
    + position_embedding (one row per patch, no CLS).  = hidden_states[0]."""
    w = sd[_VT + "embeddings.patch_embedding.weight"]
    x = _patch_conv(pixel_values.to(w.dtype), w, sd[_VT + "embeddings.patch_embedding.bias"], cfg["vision"]["patch_size"])
    return x.flatten(2).transpose(1, 2) + sd[_VT + "embeddings.position_embedding.weight"].unsqueeze(0)


def siglip_layer(sd, cfg, i, x):
    """SiglipEncoderLayer.forward: pre-LN; SiglipAttention (same arithmetic as CLIP's: scale d^-0.5, fp32 softmax);
    +res; LN; fc1; gelu_pytorch_tanh; fc2; +res."""
    eps = cfg["vision"]["layer_norm_eps"]
    p = f"{_VT}encoder.layers.{i}."
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], eps)
    x = x + clip_attention(sd, cfg, p + "self_attn.", h)
    h = F.layer_norm(x, (D,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], eps)
    h = F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    h = F.linear(gelu_tanh(h), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + h


def siglip_tower(sd, cfg, frames, return_hidden=False):
    """videollama2/model/encoder.py:111-123 SiglipVisionTower.forward + feature_select (:103-109): hidden_states[
    select_layer] with NO token dropped ('patch' is the identity here), cast back to the input dtype."""
    v = cfg["vision"]
    L, sel = v["num_hidden_layers"], v["select_layer"]
    n_run = (L + 1 + sel) if sel < 0 else sel
    x = siglip_embeddings(sd, cfg, frames)
    hs = [x]
    for i in range(n_run):
        x = siglip_layer(sd, cfg, i, x)
        hs.append(x)
    out = x.to(frames.dtype)
    return (out, hs) if return_hidden else out


def vision_tower(sd, cfg, frames, return_hidden=False):
    """build_vision_tower's dispatch (encoder.py:154-164) on the tower family."""
    return (siglip_tower if vision_family(cfg) == "siglip" else clip_tower)(sd, cfg, frames, return_hidden)


# ------------------------------------------------------------------------------------ a7: STC connector

_MP = "model.mm_projector."


def layernorm2d_nchw(x, w, b, eps=1e-5):
    """timm LayerNormAct2d: LN over the channel axis of an NCHW tensor (eps 1e-5; oracle/shims/timm)."""
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, eps).permute(0, 3, 1, 2)


def stc_bottleneck(sd, p, x):
    """timm 1.0.3 regnet.Bottleneck as RegStage(act SiLU, norm LayerNorm2d) builds it (projector.py:153-161):
    conv1 1x1 -> LN -> SiLU; conv2 depthwise 3x3 -> LN -> SiLU; SE; conv3 1x1 -> LN; (+1x1 conv+LN shortcut when
    in!=out); add; SiLU.  x is NCHW."""
    C = sd[p + "conv1.conv.weight"].shape[0]
    sc = x
    h = _conv1x1(x, sd[p + "conv1.conv.weight"])
    h = F.silu(layernorm2d_nchw(h, sd[p + "conv1.bn.weight"], sd[p + "conv1.bn.bias"]))
    h = _dwconv3x3(h, sd[p + "conv2.conv.weight"])
    h = F.silu(layernorm2d_nchw(h, sd[p + "conv2.bn.weight"], sd[p + "conv2.bn.bias"]))
    s = h.mean((2, 3), keepdim=True)
    s = F.conv2d(F.silu(F.conv2d(s, sd[p + "se.fc1.weight"], sd[p + "se.fc1.bias"])),
                 sd[p + "se.fc2.weight"], sd[p + "se.fc2.bias"])
    h = h * torch.sigmoid(s)
    h = layernorm2d_nchw(_conv1x1(h, sd[p + "conv3.conv.weight"]), sd[p + "conv3.bn.weight"], sd[p + "conv3.bn.bias"])
    if (p + "downsample.conv.weight") in sd:
        sc = layernorm2d_nchw(_conv1x1(sc, sd[p + "downsample.conv.weight"]), sd[p + "downsample.bn.weight"],
                              sd[p + "downsample.bn.bias"])
    return F.silu(h + sc)


def stc_stage(sd, stage, x):
    for b in range(1, 5):
        x = stc_bottleneck(sd, f"{_MP}{stage}.b{b}.", x)
    return x


def stc_connector(sd, x, return_stages=False, padding=1):
    """videollama2/model/projector.py:189-215 STCConnector.forward.  x [b, t, l, d] -> [b, (t' h' w'), D].
    padding = 1: STCConnector; padding = 0: STCConnectorV35 (projector.py:225-238) -- the only difference."""
    b, t, l, d = x.shape
    hw = int(l ** 0.5)
    x = x.view(b, t, hw, hw, d).permute(0, 4, 1, 2, 3)                 # b d t h w      (:199)
    x = x.permute(0, 2, 1, 3, 4).reshape(b * t, d, hw, hw)             # (b t) d h w    (:202)
    s1 = stc_stage(sd, "s1", x)                                        # (:205)
    x = s1.view(b, t, -1, hw, hw).permute(0, 2, 1, 3, 4)               # b d t h w      (:206)
    samp = F.silu(_conv3d_k2s2(x, sd[_MP + "sampler.0.weight"], sd[_MP + "sampler.0.bias"], padding))               # (:208)
    nt, nh, nw = samp.shape[2:]
    x = samp.permute(0, 2, 1, 3, 4).reshape(b * nt, -1, nh, nw)        # (:211)
    s2 = stc_stage(sd, "s2", x)                                        # (:212)
    x = s2.view(b, nt, -1, nh * nw).permute(0, 1, 3, 2).reshape(b, nt * nh * nw, -1)   # b (t h w) d  (:213)
    h = F.gelu(F.linear(x, sd[_MP + "readout.0.weight"], sd[_MP + "readout.0.bias"]))  # nn.GELU() = exact erf
    out = F.linear(h, sd[_MP + "readout.2.weight"], sd[_MP + "readout.2.bias"])        # (:214)
    return (out, dict(s1=s1, sampler=samp, s2=s2)) if return_stages else out


def encode_images_or_videos(sd, cfg, images):
    """videollama2/model/videollama2_arch.py:114-134 + temporal_aggregator (:136-159, 'tc_connector' branch)."""
    nf = cfg["num_frames"]
    batch = []
    for data, modal in images:
        batch.append(data.expand(nf, -1, -1, -1) if modal == "image" else data)
    batch = torch.stack(batch, 0)
    assert batch.dim() == 5
    b, t = batch.shape[:2]
    feats = vision_tower(sd, cfg, batch.reshape(b * t, *batch.shape[2:]))
    return stc_connector(sd, feats.view(b, t, *feats.shape[1:]), padding=conv3d_padding(cfg))


# -------------------------------------------------------------------------------- a4: multimodal splice

MODAL_INDEX_MAP = {"<image>": -200, "<video>": -201, "<audio>": -202}   # videollama2/constants.py:28-32


def splice_inputs_embeds(sd, input_ids_1d, mm_features_list):
    """videollama2_arch.py:177-222 for ONE sequence: split the ids at the (negative) modal sentinels, embed the text
    pieces with embed_tokens, put the k-th visual feature block in place of the k-th sentinel."""
    emb = sd["model.embed_tokens.weight"]
    ids = input_ids_1d
    pieces, k = [], 0
    while True:
        pos = [i for i, t in enumerate(ids.tolist()) if t in MODAL_INDEX_MAP.values()]
        if not pos:
            break
        pieces.append(F.embedding(ids[:pos[0]], emb))
        pieces.append(mm_features_list[k].to(emb.dtype))
        k += 1
        ids = ids[pos[0] + 1:]
    if ids.numel() > 0:
        pieces.append(F.embedding(ids, emb))
    return torch.cat(pieces, 0)


# ------------------------------------------------------------------------------------------ a8: Mistral


def rope_cos_sin(cfg, positions, dtype=torch.float32):
    """HF:modeling_mistral.py MistralRotaryEmbedding.forward (default rope): inv_freq = theta^(-2i/d), fp32;
    emb = cat(freqs, freqs); cos/sin cast to the activation dtype."""
    l = cfg["llm"]
    hd = l["head_dim"]
    inv = 1.0 / (l["rope_theta"] ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
    fr = positions.float()[:, None] * inv[None, :]
    emb = torch.cat([fr, fr], -1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], -1)


def rmsnorm(x, w, eps):
    """HF:modeling_mistral.py MistralRMSNorm.forward: fp32 upcast, x*rsqrt(mean(x^2)+eps), cast back, * weight."""
    dt = x.dtype
    xf = x.float()
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * xf.to(dt)


def mistral_layer(sd, cfg, i, x, cos, sin, kv=None):
    """HF:modeling_mistral.py MistralDecoderLayer.forward / MistralAttention.forward / eager_attention_forward /
    MistralMLP.forward for ONE sequence.  x [S_new, D]; kv: optional (K[nkv,S_past,hd], V) cache, returns updated."""
    l = cfg["llm"]
    nh, nkv, hd, eps = l["num_attention_heads"], l["num_key_value_heads"], l["head_dim"], l["rms_norm_eps"]
    p = f"model.layers.{i}."
    S = x.shape[0]
    h = rmsnorm(x, sd[p + "input_layernorm.weight"], eps)
    # Qwen2Attention (HF:models/qwen2/modeling_qwen2.py) is the same arithmetic with a bias on q/k/v
    q = F.linear(h, sd[p + "self_attn.q_proj.weight"], sd.get(p + "self_attn.q_proj.bias")).view(S, nh, hd).transpose(0, 1)
    k = F.linear(h, sd[p + "self_attn.k_proj.weight"], sd.get(p + "self_attn.k_proj.bias")).view(S, nkv, hd).transpose(0, 1)
    v = F.linear(h, sd[p + "self_attn.v_proj.weight"], sd.get(p + "self_attn.v_proj.bias")).view(S, nkv, hd).transpose(0, 1)
    q = q * cos + rotate_half(q) * sin
    k = k * cos + rotate_half(k) * sin
    if kv is not None:
        k = torch.cat([kv[0], k], 1)
        v = torch.cat([kv[1], v], 1)
    Sk = k.shape[1]
    rep = nh // nkv
    kk = k[:, None].expand(nkv, rep, Sk, hd).reshape(nh, Sk, hd)
    vv = v[:, None].expand(nkv, rep, Sk, hd).reshape(nh, Sk, hd)
    a = torch.matmul(q, kk.transpose(1, 2)) * (hd ** -0.5)
    qpos = torch.arange(Sk - S, Sk, device=x.device)[:, None]
    mask = torch.arange(Sk, device=x.device)[None, :] > qpos
    a = a.masked_fill(mask[None], torch.finfo(a.dtype).min)
    a = F.softmax(a, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(a, vv).transpose(0, 1).reshape(S, nh * hd)
    x = x + F.linear(o, sd[p + "self_attn.o_proj.weight"])
    h = rmsnorm(x, sd[p + "post_attention_layernorm.weight"], eps)
    h = F.linear(F.silu(F.linear(h, sd[p + "mlp.gate_proj.weight"])) * F.linear(h, sd[p + "mlp.up_proj.weight"]),
                 sd[p + "mlp.down_proj.weight"])
    return x + h, (k, v)


def mistral_forward(sd, cfg, x, start_pos=0, caches=None, n_layers=None, last_only=True):
    """MistralModel.forward + lm_head for one sequence of new embeddings x [S_new, D] at positions start_pos...
    Returns (logits [1 or S_new, V] fp32-or-dtype, caches)."""
    l = cfg["llm"]
    n_layers = l["num_hidden_layers"] if n_layers is None else n_layers
    S = x.shape[0]
    cos, sin = rope_cos_sin(cfg, torch.arange(start_pos, start_pos + S), x.dtype)
    cos, sin = cos.to(x.device), sin.to(x.device)
    new = []
    for i in range(n_layers):
        x, kv = mistral_layer(sd, cfg, i, x, cos, sin, None if caches is None else caches[i])
        new.append(kv)
    x = rmsnorm(x, sd["model.norm.weight"], l["rms_norm_eps"])
    if last_only:
        x = x[-1:]
    return F.linear(x, sd["lm_head.weight"]), new


def greedy_generate(sd, cfg, inputs_embeds, max_new_tokens, eos_token_id=None, n_layers=None, return_prefill_caches=False):
    """HF GenerationMixin._sample with do_sample=False as mm_infer drives it (videollama2/__init__.py:99-110):
    prefill on inputs_embeds, argmax, feed embed_tokens(next) one token at a time; stop at EOS (the semantic
    content of KeywordsStoppingCriteria, mm_utils.py:329-339).  Returns (new token ids list, per-step logits)."""
    logits, caches = mistral_forward(sd, cfg, inputs_embeds, 0, None, n_layers)
    prefill_caches = caches            # (a layer's cache grows by concatenation into NEW tensors: this list keeps the prefill state)
    pos = inputs_embeds.shape[0]
    toks, all_logits = [], []
    for _ in range(max_new_tokens):
        all_logits.append(logits[0].float())
        nxt = int(torch.argmax(logits[0].float()))
        toks.append(nxt)
        if eos_token_id is not None and nxt == eos_token_id:
            break
        x = F.embedding(torch.tensor([nxt], device=inputs_embeds.device), sd["model.embed_tokens.weight"]).to(inputs_embeds.dtype)
        logits, caches = mistral_forward(sd, cfg, x, pos, caches, n_layers)
        pos += 1
    if return_prefill_caches:          # test harness: teacher-forced decode of ANOTHER weight set (fp8-dequantised) from the same prefill
        return toks, torch.stack(all_logits), prefill_caches
    return toks, torch.stack(all_logits)


def generate(sd, cfg, input_ids_1d, frames, max_new_tokens, eos_token_id=None):
    """videollama2_mistral.py:110-144 generate(inputs, images=[(frames,'video')]) for batch 1."""
    feats = encode_images_or_videos(sd, cfg, [(frames, "video")])
    emb = splice_inputs_embeds(sd, input_ids_1d, [feats[0]])
    return greedy_generate(sd, cfg, emb, max_new_tokens, eos_token_id)


def n_visual_tokens(T, grid=24, padding=1):
    """(T/2+1) * (grid/2+1)^2 for Conv3d(k=2,s=2,p=1): 845 / 1521 / 2873 for T=8/16/32; with padding 0
    (v35, 27x27 SigLIP grid): (T/2) * 13 * 13 = 1352 at T=16."""
    o = lambda n: (n + 2 * padding - 2) // 2 + 1
    return o(T) * o(grid) * o(grid)
