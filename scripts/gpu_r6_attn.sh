#!/bin/bash
# round 6: two key streams per query block (vl2_attn_fwd variant 4) against the shipped one-stream form (3)
mkdir -p gpurun_out
python - <<'PY' 2>&1 | tee gpurun_out/r06_attn_ns2_check.txt
import torch
from videollama2_amd import ops
torch.manual_seed(0)
def rel(a, b): return float((a.float() - b.float()).norm() / b.float().norm())
D, smax = 128, 4096
for S, nh, nkv in ((945, 32, 8), (1621, 32, 8), (2973, 32, 8), (130, 4, 2), (64, 4, 2)):
    q = torch.randn(S, nh * D, device="cuda").bfloat16(); kc = torch.randn(nkv, smax, D, device="cuda").bfloat16(); vc = torch.randn(nkv, smax, D, device="cuda").bfloat16()
    outs = {}
    for v in (3, 4):
        ops.set_attn_kv_groups(v)
        o = torch.zeros(S, nh * D, dtype=torch.bfloat16, device="cuda")
        ops.attn_fwd(q, kc, vc, o, (0, D, nh * D), (0, smax * D, D), (0, smax * D, D), (0, D, nh * D), 1, nh, S, S, nh // nkv, D ** -0.5, True, 0, D)
        torch.cuda.synchronize(); outs[v] = o
    qf = q.view(S, nh, D).transpose(0, 1).float(); g = nh // nkv
    kf, vf = kc[:, :S].float().repeat_interleave(g, 0), vc[:, :S].float().repeat_interleave(g, 0)
    sc = (qf @ kf.transpose(1, 2) * D ** -0.5).masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device="cuda"), 1), float("-inf"))
    ref = (torch.softmax(sc, -1) @ vf).transpose(0, 1).reshape(S, nh * D)
    print(f"causal S={S}: v3 vs torch {rel(outs[3], ref):.3e}  v4 vs torch {rel(outs[4], ref):.3e}  v4 vs v3 {rel(outs[4], outs[3]):.3e}")
for B in (16, 3):
    H, N, D = 16, 577, 64
    qkv = torch.randn(B * N, 3 * H * D, device="cuda").bfloat16(); st = (N * 3 * H * D, D, 3 * H * D)
    outs = {}
    for v in (0, 3, 4):
        ops.set_attn_kv_groups(v)
        o = torch.zeros(B * N, H * D, dtype=torch.bfloat16, device="cuda")
        ops.attn_fwd(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], o, st, st, st, (N * H * D, D, H * D), B, H, N, N, 1, D ** -0.5, False, 0, D)
        torch.cuda.synchronize(); outs[v] = o
    q, k, v_ = [t.view(B, N, H, D).transpose(1, 2).float() for t in qkv.view(B * N, 3, H * D).unbind(1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * D ** -0.5, -1) @ v_).transpose(1, 2).reshape(B * N, H * D)
    print(f"vit T={B}: default vs torch {rel(outs[0], ref):.3e}  v4 vs torch {rel(outs[4], ref):.3e}  v4 vs default {rel(outs[4], outs[0]):.3e}")
ops.set_attn_kv_groups(0)
PY
python scripts/attn_bench2.py 2>&1 | tee gpurun_out/r06_attn_ns2_ab.jsonl
