#!/usr/bin/env python
"""Generates videollama2_amd/csrc/k_gemm8_loop.inc: the main loop of gemm8 (k_gemm8.h) as ONE inline-asm statement with hand-allocated registers.

Why assembly: the schedule this kernel needs -- one wave per SIMD, 256 accumulator registers, global loads staged through two register sets into
ds_write_b128, ONE memory instruction in every MFMA gap -- is what the vendor's GEMM kernels do and what hipcc cannot keep: with more than one
register set live across the loop its allocator shuffles accumulators between the VGPR and AGPR files (profiles/r05_experiments.md section 7).
Here the allocation is fixed by hand (arch VGPRs 128..255; the accumulators are the compiler's sixteen AGPR tuples, passed as "+a" operands) and
the compiler only sees one opaque statement.

Register map (arch VGPRs):   FA0 v[128:143]  FB0 v[144:159]   fragments of k-step 0 (4 x 16-B A fragments, 4 x B)
                             FA1 v[160:175]  FB1 v[176:191]   fragments of k-step 1
                             G0  v[192:223]  G1  v[224:255]   the two staging sets (8 pieces of 16 B per lane each)
Schedule, slab t (P = t & 1), every memory instruction behind one MFMA:
   region 1 (16 MFMAs on FA0 / FB0):  gaps 0-7  ds_read FA1 / FB1 <- slab t, k-step 1, LDS stage P
                                      gaps 8-15 [vmcnt(8)] ds_write G[P] (= slab t+1, loaded in iteration t-2) -> LDS stage P^1
   lgkmcnt(0) ; s_barrier
   region 2 (16 MFMAs on FA1 / FB1):  gaps 0-7  ds_read FA0 / FB0 <- slab t+1, k-step 0, LDS stage P^1
                                      gaps 8-15 buffer_load G[P] <- slab t+3 (1.5 slab times before its ds_write)
Two slabs per loop trip (K % 64 == 0 makes the slab count even).  Past the end of K the loads re-read the last slab (offset clamped) and the
last iteration writes / reads a stray slab nobody consumes: no tail code.  Same MFMA, same k order as every kernel of the family -> same bits.
MEASURED (round 5, calls of scripts/gpu_r5_i.sh; profiles/r05_experiments.md section 7): correct at the first run (bit-identical to the family on the device),
and NOT faster: 8192^3 912 us with the reads / writes / loads bunched, 857 us alternating (gemm4 844 on that box), 890 vs 879 with the odd waves on the
swapped stream -- the same speed as the compiler-scheduled LDS-DMA loop that ships as variant 9.  The schedule was never the bound: scripts/ubench/mfma_power16.hip
shows the instruction is (v_mfma_f32_16x16x32_bf16, which the vendor kernels use, sustains 2.12-2.15 PF on random operands where 32x32x16 sustains 1.79-1.88).
LAB MATERIAL: the product does not include the generated file.  To try it: python scripts/ubench/gen_gemm8_loop.py (writes csrc/k_gemm8_loop.inc), then
`git apply scripts/ubench/gemm8_asm_loop.patch` (the integration into k_gemm8.h: operands of the statement, `#ifndef VL2_GEMM8_CXX_LOOP` for the emulator,
which needs `#define VL2_GEMM8_CXX_LOOP 1` in tests/emu/hip_emu.h) and rebuild."""
import os

STAGE = 32768


def frag(base, i):
    return f"v[{base + 4 * i}:{base + 4 * i + 3}]"


FA = [128, 160]
FB = [144, 176]
G = [192, 224]


def mfma(tr, i, j, s):
    a, b = frag(FA[s], i), frag(FB[s], j)
    src = f"{b}, {a}" if tr else f"{a}, {b}"
    return f"v_mfma_f32_32x32x16_bf16 %[c{i * 4 + j}], {src}, %[c{i * 4 + j}]"


def read_frag(kind, s, idx, stage):
    # fragment set s = k-step s: base address operand a_rd{s} / b_rd{s}; tile idx is +2048 B; W lives 16 KiB into the stage
    if kind == "a":
        return f"ds_read_b128 {frag(FA[s], idx)}, %[ard{s}] offset:{stage * STAGE + idx * 2048}"
    return f"ds_read_b128 {frag(FB[s], idx)}, %[brd{s}] offset:{stage * STAGE + 16384 + idx * 2048}"


def write_piece(gset, pc, stage):
    off = stage * STAGE + (pc % 4) * 4096 + (16384 if pc >= 4 else 0)
    return f"ds_write_b128 %[ldswr], v[{G[gset] + 4 * pc}:{G[gset] + 4 * pc + 3}] offset:{off}"


def load_piece(gset, pc):
    dst = f"v[{G[gset] + 4 * pc}:{G[gset] + 4 * pc + 3}]"
    if pc < 4:
        return [f"buffer_load_dwordx4 {dst}, %[avo{pc}], %[rsa], %[kb] offen"]
    i = pc - 4
    if i == 0:
        return [f"buffer_load_dwordx4 {dst}, %[wvo], %[rsw], %[kb] offen"]
    return [f"s_add_u32 %[tmp], %[kb], %[wst{i}]", f"buffer_load_dwordx4 {dst}, %[wvo], %[rsw], %[tmp] offen"]


def slab(tr, P, swap=False):
    """One iteration (slab parity P).  swap: the odd waves' stream -- write / load FIRST in every pair of gaps, so that at any moment two waves of the
    workgroup read fragments while the other two write (region 1) or load (region 2): the per-CU LDS-write and vector-memory paths see half the burst."""
    out = []
    # ---- region 1: MFMAs on fragment set 0
    out.append("s_waitcnt lgkmcnt(0)")                     # FA0 / FB0 (read in the previous region 2) have landed
    order = [(i, j) for i in range(4) for j in range(4)]
    # reads and writes ALTERNATE (the four waves of the workgroup run this stream in lockstep: eight ds_write_b128 in consecutive gaps ask the LDS
    # write path for 4 KiB per 32 cycles, 128 B/clk against the ~79 it has -- measured: 912 us on 8192^3 bunched)
    rd = [read_frag("a", 1, i, P) for i in range(4)] + [read_frag("b", 1, j, P) for j in range(4)]
    wr = [write_piece(P, pc, P ^ 1) for pc in range(8)]
    mem1 = [x for pair in (zip(wr, rd) if swap else zip(rd, wr)) for x in pair]
    for g, (i, j) in enumerate(order):
        out.append(mfma(tr, i, j, 0))
        if g == (0 if swap else 1):
            out.append("s_waitcnt vmcnt(8)")               # the set's loads (two iterations old) have landed; the other set's eight stay in flight
        out.append(mem1[g])
    out.append("s_waitcnt lgkmcnt(0)")
    out.append("s_barrier")
    # ---- region 2: MFMAs on fragment set 1
    rd2 = [[read_frag("a", 0, i, P ^ 1)] for i in range(4)] + [[read_frag("b", 0, j, P ^ 1)] for j in range(4)]
    ld2 = [load_piece(P, pc) for pc in range(8)]
    mem2 = [x for pair in (zip(ld2, rd2) if swap else zip(rd2, ld2)) for x in pair]
    for g, (i, j) in enumerate(order):
        out.append(mfma(tr, i, j, 1))
        out.extend(mem2[g])
    out.append("s_add_u32 %[kb], %[kb], 64")
    out.append("s_min_u32 %[kb], %[kb], %[kblast]")
    return out


def body(tr):
    lines = []
    # ---- prologue: slab 0 -> G1 -> LDS stage 0; slab 1 -> G0 (iteration 0 writes it); slab 2 -> G1; fragments (slab 0, k-step 0)
    lines.append("s_mov_b32 %[kb], 0")
    for pc in range(8):
        lines.extend(load_piece(1, pc))
    lines.append("s_min_u32 %[kb], 64, %[kblast]")
    for pc in range(8):
        lines.extend(load_piece(0, pc))
    lines.append("s_waitcnt vmcnt(8)")
    for pc in range(8):
        lines.append(write_piece(1, pc, 0))
    lines.append("s_min_u32 %[kb], 128, %[kblast]")
    for pc in range(8):
        lines.extend(load_piece(1, pc))
    lines.append("s_min_u32 %[kb], 192, %[kblast]")
    lines.append("s_waitcnt lgkmcnt(0)")
    lines.append("s_barrier")
    for i in range(4):
        lines.append(read_frag("a", 0, i, 0))
    for j in range(4):
        lines.append(read_frag("b", 0, j, 0))
    lines.append("s_cmp_lg_u32 %[odd], 0")                 # odd waves take the swapped stream (wave-uniform branch)
    lines.append("s_cbranch_scc1 L_gemm8_odd_%=")
    lines.append("L_gemm8_loop_%=:")
    lines.extend(slab(tr, 0))
    lines.extend(slab(tr, 1))
    lines.append("s_sub_u32 %[nloop], %[nloop], 1")
    lines.append("s_cmp_lg_u32 %[nloop], 0")
    lines.append("s_cbranch_scc1 L_gemm8_loop_%=")
    lines.append("s_branch L_gemm8_done_%=")
    lines.append("L_gemm8_odd_%=:")
    lines.extend(slab(tr, 0, True))
    lines.extend(slab(tr, 1, True))
    lines.append("s_sub_u32 %[nloop], %[nloop], 1")
    lines.append("s_cmp_lg_u32 %[nloop], 0")
    lines.append("s_cbranch_scc1 L_gemm8_odd_%=")
    lines.append("L_gemm8_done_%=:")
    lines.append("s_waitcnt vmcnt(0) lgkmcnt(0)")          # the stray loads / reads of the last iterations
    lines.append("s_barrier")                               # every wave is out of the ring before the epilogue reuses it
    return lines


def emit(name, tr):
    ls = body(tr)
    s = f"#define {name}() \\\n    asm volatile( \\\n"
    for l in ls:
        s += f'        "{l}\\n\\t" \\\n'
    outs = ", ".join(f'[c{k}] "+a"(acc[{k // 4}][{k % 4}])' for k in range(16))
    s += f"        : {outs}, \\\n          [kb] \"=&s\"(g8_kb), [nloop] \"+s\"(g8_nloop), [tmp] \"=&s\"(g8_tmp) \\\n"
    s += ("        : [avo0] \"v\"(a_vo[0]), [avo1] \"v\"(a_vo[1]), [avo2] \"v\"(a_vo[2]), [avo3] \"v\"(a_vo[3]), [wvo] \"v\"(w_vo0), \\\n"
          "          [ard0] \"v\"(g8_ard[0]), [ard1] \"v\"(g8_ard[1]), [brd0] \"v\"(g8_brd[0]), [brd1] \"v\"(g8_brd[1]), [ldswr] \"v\"(g8_ldswr), \\\n"
          "          [rsa] \"s\"(g8_rsa), [rsw] \"s\"(g8_rsw), [kblast] \"s\"(g8_kblast), [wst1] \"s\"(g8_wst[0]), [wst2] \"s\"(g8_wst[1]), [wst3] \"s\"(g8_wst[2]), [odd] \"s\"(g8_odd) \\\n")
    clob = ", ".join(f'"v{r}"' for r in range(128, 256))
    s += f"        : {clob}, \"memory\", \"scc\")\n"
    return s


def main():
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = os.path.join(here, "videollama2_amd", "csrc", "k_gemm8_loop.inc")
    txt = ("// GENERATED by scripts/gen_gemm8_loop.py -- do not edit.  The main loop of gemm8 (k_gemm8.h) as one inline-asm statement with hand-allocated\n"
           "// registers; see the generator's docstring for the register map and the schedule.  VL2_GEMM8_LOOP() = acc += A.W^T, VL2_GEMM8_LOOP_TR() = the\n"
           "// operand-swapped form (accumulators hold C^T, for the register-resident epilogue).\n")
    txt += emit("VL2_GEMM8_LOOP", False) + "\n" + emit("VL2_GEMM8_LOOP_TR", True)
    open(out, "w").write(txt)
    print(out, len(txt.splitlines()), "lines")


if __name__ == "__main__":
    main()
