"""Model hyper-parameters of the hot path (no hub access on the target box, so the public VideoLLaMA2-7B values are
hard-coded; SURVEY.md section 8 header).  The dict layout {vision, llm, num_frames} is shared with the test oracle."""


def videollama2_7b(num_frames=16):
    """CLIP-ViT-L/14-336 + stc_connector + Mistral-7B-Instruct-v0.2 (README.md:117-120 of the reference)."""
    return dict(
        vision=dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                    image_size=336, patch_size=14, layer_norm_eps=1e-5, select_layer=-2),
        llm=dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                 num_key_value_heads=8, head_dim=128, vocab_size=32000, rms_norm_eps=1e-5, rope_theta=1e6),
        num_frames=num_frames)


def videollama2_1_7b_16f(num_frames=16):
    """VideoLLaMA2.1-7B-16F (the reference's default checkpoint, README.md:327): SigLIP-so400m-patch14-384 tower
    (encoder.py:84-151), stc_connector_v35 (projector.py:225-238), Qwen2-7B decoder (videollama2_qwen2.py)."""
    return dict(
        vision=dict(family="siglip", hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16,
                    image_size=384, patch_size=14, layer_norm_eps=1e-6, select_layer=-2),
        llm=dict(family="qwen2", hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28,
                 num_key_value_heads=4, head_dim=128, vocab_size=152064, rms_norm_eps=1e-6, rope_theta=1e6),
        projector="stc_connector_v35", num_frames=num_frames)


def videollama2_72b(num_frames=16):
    """VideoLLaMA2-72B (BASELINE.json configs[3]): CLIP-ViT-L/14-336 + stc_connector + Qwen2-72B-Instruct (public config:
    hidden 8192, 80 layers, 64 q / 8 kv heads x 128, MLP 29568, vocab 152064).  145 GB of bf16 weights: fits ONE MI355X (288 GB
    HBM3E); `HipMistralDecoder(tp_group=...)` shards it (29568 / 8 = 3696 per rank is zero-padded to 3712)."""
    return dict(
        vision=dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                    image_size=336, patch_size=14, layer_norm_eps=1e-5, select_layer=-2),
        llm=dict(family="qwen2", hidden_size=8192, intermediate_size=29568, num_hidden_layers=80, num_attention_heads=64,
                 num_key_value_heads=8, head_dim=128, vocab_size=152064, rms_norm_eps=1e-6, rope_theta=1e6),
        projector="stc_connector", num_frames=num_frames)


def from_hf_config(hf_cfg, vision_cfg):
    """Build the dict from a Videollama2MistralConfig + CLIPVisionConfig (videollama2_arch.py:49-68 keys)."""
    g = lambda o, k, d=None: getattr(o, k, d)
    rope_theta = g(hf_cfg, "rope_theta", None)
    if rope_theta is None and g(hf_cfg, "rope_parameters", None):
        rope_theta = hf_cfg.rope_parameters.get("rope_theta", 1e6)
    vfam = "siglip" if "siglip" in str(g(vision_cfg, "model_type", "")) else "clip"
    lfam = "qwen2" if "qwen2" in str(g(hf_cfg, "model_type", "")) else "mistral"
    return dict(
        projector=g(hf_cfg, "mm_projector_type", "stc_connector"),
        vision=dict(family=vfam, hidden_size=vision_cfg.hidden_size, intermediate_size=vision_cfg.intermediate_size,
                    num_hidden_layers=vision_cfg.num_hidden_layers, num_attention_heads=vision_cfg.num_attention_heads,
                    image_size=vision_cfg.image_size, patch_size=vision_cfg.patch_size,
                    layer_norm_eps=vision_cfg.layer_norm_eps, select_layer=g(hf_cfg, "mm_vision_select_layer", -2)),
        llm=dict(family=lfam, hidden_size=hf_cfg.hidden_size, intermediate_size=hf_cfg.intermediate_size,
                 num_hidden_layers=hf_cfg.num_hidden_layers, num_attention_heads=hf_cfg.num_attention_heads,
                 num_key_value_heads=hf_cfg.num_key_value_heads,
                 head_dim=g(hf_cfg, "head_dim", None) or hf_cfg.hidden_size // hf_cfg.num_attention_heads,
                 vocab_size=hf_cfg.vocab_size, rms_norm_eps=hf_cfg.rms_norm_eps, rope_theta=float(rope_theta or 1e6)),
        num_frames=g(hf_cfg, "num_frames", 8))


def check_supported(cfg):
    """The gfx950 kernels are built for head_dim 64 / 128 (the SigLIP tower pads other head dims <= 128 and its MLP width
    with zeros at load time), LLM head_dim 128, GEMM N%128==0, K%64==0."""
    v, l = cfg["vision"], cfg["llm"]
    hd_v = v["hidden_size"] // v["num_attention_heads"]
    siglip = v.get("family", "clip") == "siglip"
    errs = []
    if (hd_v != 64 and not siglip) or hd_v > 128:
        errs.append(f"vision head_dim {hd_v} not supported")
    if l["head_dim"] != 128:
        errs.append(f"llm head_dim {l['head_dim']} != 128")
    if l["num_attention_heads"] % l["num_key_value_heads"]:
        errs.append("q heads not a multiple of kv heads")
    if cfg.get("projector", "stc_connector") not in ("stc_connector", "stc_connector_v35"):
        errs.append(f"projector {cfg.get('projector')} not built (stc_connector, stc_connector_v35)")
    dims = [("vision hidden", v["hidden_size"]), ("llm hidden", l["hidden_size"]), ("llm mlp", l["intermediate_size"]),
            ("vocab", l["vocab_size"])] + ([] if siglip else [("vision mlp", v["intermediate_size"])])
    for name, n in dims:
        if n % 128:
            errs.append(f"{name} {n} % 128 != 0")
    if l["hidden_size"] > 8192 or v["hidden_size"] > 8192:
        errs.append("hidden size > 8192 (row-norm / depthwise kernels)")
    if errs:
        raise ValueError("config not supported by the gfx950 kernels: " + "; ".join(errs))
