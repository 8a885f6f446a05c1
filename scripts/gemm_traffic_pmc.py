#!/usr/bin/env python
"""Driver for the HBM-traffic PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes): one launch of every
distinct GEMM shape of the T=16 workload (after a warm-up launch), with the library's own per-shape kernel choice.
scripts/gemm_traffic_post.py turns the two counter CSVs into profiles/r01_gemm_traffic.json."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops
from videollama2_amd.connector import conv3d_k2s2p1_index
dev = "cuda"
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
SHAPES = [  # M, N, K, count per step, kwargs
    (9216, 1024, 640, 1, {}), (9232, 3072, 1024, 23, dict(bias=1)), (9232, 1024, 1024, 23, dict(bias=1, res=1)),
    (9232, 4096, 1024, 23, dict(bias=1, act=1)), (9232, 1024, 4096, 23, dict(bias=1, res=1)),
    (9216, 4096, 1024, 2, {}), (9216, 4096, 4096, 7, {}), (1521, 4096, 4096, 10, {}),
    (1621, 6144, 4096, 32, {}), (1621, 4096, 4096, 32, dict(res=1)), (1621, 28672, 4096, 32, dict(swiglu=1)),
    (1621, 4096, 14336, 32, dict(res=1)),
]
SEP = torch.zeros(64, device=dev)
def run(M, N, K, kw):
    a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    ncol = N // 2 if kw.get("swiglu") else N
    bias = torch.randn(N, device=dev) if kw.get("bias") else None
    res = rnd(M, ncol) if kw.get("res") else None
    c = torch.empty(M, ncol, dtype=torch.bfloat16, device=dev)
    ops.attach_workspace(dev)
    # the kernels of the DEFAULT path (round 6): the tower's q/k/v and fc1 take the persistent form (vl2_vit_forward: VL2_GEMM_PERSISTENT + a tile
    # counter block -- round 5's driver set the flag without the counters and so counted the non-persistent kernels), the decoder's gate/up runs
    # on the 16 x 16 x 32 instruction (vl2_llm_prefill: VL2_GEMM_MFMA16, the mixed launch of k_gemm9.h)
    persistent = M == 9232 and K == 1024 and N >= 3072
    ops.set_stage_flags(ops.STAGE_PERSISTENT_GEMM if persistent else 0)
    ctr = torch.zeros(16, dtype=torch.int32, device=dev) if persistent else None
    for _ in range(2):      # launch 1 = warm-up (L2/MALL state), launch 2 = the one post-processing reads
        ops.gemm(a, w, bias=bias, res=res, act=kw.get("act", 0), swiglu=bool(kw.get("swiglu")), out=c, mfma16=bool(kw.get("swiglu")), tile_ctr=ctr)
        SEP.fill_(1.0)      # separator kernel: a vl2_gemm call may be two kernels back to back (row split), the post-processor groups by it
    torch.cuda.synchronize()
    ops.set_stage_flags(0)
for M, N, K, cnt, kw in SHAPES:
    run(M, N, K, kw)
# Conv3d as the gathered GEMM
T, H, C = 16, 24, 4096
pool = rnd(T * H * H, C); w3 = rnd(C, 8 * C, scale=(8 * C) ** -0.5); b3 = torch.randn(C, device=dev)
idx, _ = conv3d_k2s2p1_index(T, H, H, dev)
for _ in range(2):
    ops.gemm(pool, w3, bias=b3, act=3, gather=(idx, torch.zeros(C, dtype=torch.bfloat16, device=dev), C))
    SEP.fill_(1.0)
torch.cuda.synchronize()
