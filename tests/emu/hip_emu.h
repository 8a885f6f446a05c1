// TEST INFRASTRUCTURE ONLY -- host-side functional emulator of the gfx950 execution model, used by
// tests/test_emu_kernels.py to check kernel INDEXING LOGIC (LDS layouts, swizzles, MFMA fragment maps,
// edge handling) on the CPU-only build container, where no GPU is present.  It is not a backend: the product
// library is compiled by hipcc for gfx950 only, and nothing under videollama2_amd/ includes this file.
//
// Model: one workgroup at a time; every work-item is a ucontext fiber; wave = 64 consecutive threads;
// __syncthreads and wave-collective builtins (shuffles, MFMA, LDS-DMA) are rendezvous points.
// MFMA lane<->element maps follow /opt/skills/guides/cdna_hip_programming.md section 3.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
struct Wave {
    int arrived = 0;
    unsigned gen = 0;
    alignas(16) unsigned char scratch[64][128];   // per-lane operand deposit area
    alignas(16) unsigned char shfl[2][64][16];    // double-buffered deposit area of the shuffles (one sync per shuffle)
};
struct Thread {
    dim3 tid;
    int lane = 0, wave = 0, flat = 0, shfl_phase = 0;
    ucontext_t ctx;          // portable switch (non-x86-64 hosts)
    void* sp = nullptr;      // x86-64: saved stack pointer of the parked fiber (emu_ctx_switch)
    char* stack = nullptr;
    bool done = false;
};
inline Thread* cur = nullptr;
inline dim3 g_block, g_bdim, g_gdim;
inline std::vector<Thread> threads;
inline std::vector<Wave> waves;
inline ucontext_t main_ctx;
inline int blk_arrived = 0;
inline unsigned blk_gen = 0;
inline std::function<void()> body;
inline int nthreads = 0;
alignas(64) inline unsigned char dyn_smem[160 * 1024];

#if defined(__x86_64__)
// Fiber switch = push the SysV callee-saved registers, swap stack pointers, pop, ret (~5 ns).  glibc's swapcontext makes two
// rt_sigprocmask system calls per switch, and the emulated kernels switch at every barrier, shuffle and MFMA: with it the
// system calls were most of the emulator's run time.
extern "C" void emu_ctx_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .weak emu_ctx_switch
    .type emu_ctx_switch,@function
emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_ctx_switch, .-emu_ctx_switch
)");
inline void* sched_sp = nullptr;
inline void yield() { emu_ctx_switch(&cur->sp, sched_sp); }
inline void resume(Thread& t) { cur = &t; emu_ctx_switch(&sched_sp, t.sp); }
inline void leave() { emu_ctx_switch(&cur->sp, sched_sp); __builtin_trap(); }
inline void trampoline();
inline void prepare(Thread& t, size_t stack_bytes) {
    // stack image the first resume pops: six zeroed callee-saved registers, then `ret` into trampoline with rsp = 8 mod 16
    uintptr_t top = ((uintptr_t)t.stack + stack_bytes) & ~(uintptr_t)15;
    void** sp = (void**)(top - 8 * sizeof(void*));
    for (int i = 0; i < 6; ++i) sp[i] = nullptr;
    sp[6] = (void*)&trampoline;
    sp[7] = nullptr;
    t.sp = sp;
}
#else
inline void yield() { swapcontext(&cur->ctx, &main_ctx); }
inline void resume(Thread& t) { cur = &t; swapcontext(&main_ctx, &t.ctx); }
inline void leave() { swapcontext(&cur->ctx, &main_ctx); }
inline void trampoline();
inline void prepare(Thread& t, size_t stack_bytes) {
    getcontext(&t.ctx);
    t.ctx.uc_stack.ss_sp = t.stack; t.ctx.uc_stack.ss_size = stack_bytes; t.ctx.uc_link = &main_ctx;
    makecontext(&t.ctx, (void (*)())trampoline, 0);
}
#endif

inline void block_sync() {
    unsigned g = blk_gen;
    if (++blk_arrived == nthreads) { blk_arrived = 0; ++blk_gen; }
    else while (blk_gen == g) yield();
}
inline void wave_sync() {
    Wave& w = waves[cur->wave];
    unsigned g = w.gen;
    int n = 64;
    int wave_threads = nthreads - cur->wave * 64;
    if (wave_threads < n) n = wave_threads;
    if (++w.arrived == n) { w.arrived = 0; ++w.gen; }
    else while (w.gen == g) yield();
}
inline void trampoline() {
    body();
    cur->done = true;
    leave();
}
// optional explicit workgroup order (1-D grids): used to run stream-K contributors before the workgroups that wait on them
inline std::vector<unsigned> block_order;
inline void launch(dim3 grid, dim3 block, std::function<void()> fn, size_t stack_bytes = 256 * 1024) {
    g_gdim = grid; g_bdim = block;
    nthreads = block.x * block.y * block.z;
    body = fn;
    threads.assign(nthreads, Thread());
    for (auto& t : threads) t.stack = (char*)malloc(stack_bytes);
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bi = 0; bi < grid.x; ++bi) {
        const unsigned bx = block_order.size() == grid.x ? block_order[bi] : bi;
        g_block = dim3(bx, by, bz);
        waves.assign((nthreads + 63) / 64, Wave());
        blk_arrived = 0; blk_gen = 0;
        int f = 0;
        for (unsigned z = 0; z < block.z; ++z) for (unsigned y = 0; y < block.y; ++y) for (unsigned x = 0; x < block.x; ++x, ++f) {
            Thread& t = threads[f];
            t.tid = dim3(x, y, z); t.flat = f; t.lane = f & 63; t.wave = f >> 6; t.done = false; t.shfl_phase = 0;
            prepare(t, stack_bytes);
        }
        int remaining = nthreads;
        while (remaining) {
            for (auto& t : threads) if (!t.done) {
                resume(t);
                if (t.done) --remaining;
            }
        }
    }
    for (auto& t : threads) free(t.stack);
    threads.clear();
    block_order.clear();
}

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
inline float bf2f(short s) { uint32_t u = ((uint32_t)(uint16_t)s) << 16; float f; memcpy(&f, &u, 4); return f; }

// v_mfma_f32_32x32x16_bf16: A lane l = A[i=l&31][k=(l>>5)*8+j]; B lane l = B[k=(l>>5)*8+j][n=l&31];
// D reg r lane l = D[(r&3)+8*(r>>2)+4*(l>>5)][l&31].
inline f32x16 mfma_32x32x16_bf16(s16x8 a, s16x8 b, f32x16 c) {
    Wave& w = waves[cur->wave];
    memcpy(w.scratch[cur->lane], &a, 16); memcpy(w.scratch[cur->lane] + 16, &b, 16);
    wave_sync();
    int l = cur->lane;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            short av, bv;
            memcpy(&av, w.scratch[row + 32 * (k >> 3)] + 2 * (k & 7), 2);
            memcpy(&bv, w.scratch[col + 32 * (k >> 3)] + 16 + 2 * (k & 7), 2);
            acc = fmaf(bf2f(av), bf2f(bv), acc);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}
// v_mfma_f32_16x16x32_bf16: A lane l = A[i=l&15][k=(l>>4)*8+j]; B = B[k=(l>>4)*8+j][n=l&15]; D reg r = D[(l>>4)*4+r][l&15].
inline f32x4 mfma_16x16x32_bf16(s16x8 a, s16x8 b, f32x4 c) {
    Wave& w = waves[cur->wave];
    memcpy(w.scratch[cur->lane], &a, 16); memcpy(w.scratch[cur->lane] + 16, &b, 16);
    wave_sync();
    int l = cur->lane;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            short av, bv;
            memcpy(&av, w.scratch[row + 16 * (k >> 3)] + 2 * (k & 7), 2);
            memcpy(&bv, w.scratch[col + 16 * (k >> 3)] + 16 + 2 * (k & 7), 2);
            acc = fmaf(bf2f(av), bf2f(bv), acc);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}
// v_mfma_f32_32x32x64_f8f6f4 with e4m3fn operands (the scaled builtin with zero scale operands = the unscaled instruction): a lane holds 32
// bytes of A row (l & 31) and 32 bytes of B column (l & 31) for k half (l >> 5); D[i][j] = sum over both halves and the 32 bytes.  Which k a
// byte is does not matter to the sum (the same index function for A and B); the products are exact in fp32, the sum order is this loop's.
float emu_mx_e4m3(unsigned b);
typedef int emu_i32x8 __attribute__((ext_vector_type(8)));
inline f32x16 mfma_32x32x64_f8(emu_i32x8 a, emu_i32x8 b, f32x16 c) {
    Wave& w = waves[cur->wave];
    memcpy(w.scratch[cur->lane], &a, 32); memcpy(w.scratch[cur->lane] + 32, &b, 32);
    wave_sync();
    int l = cur->lane;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int h = 0; h < 2; ++h)
            for (int k = 0; k < 32; ++k)
                acc = fmaf(emu_mx_e4m3(w.scratch[row + 32 * h][k]), emu_mx_e4m3(w.scratch[col + 32 * h][32 + k]), acc);
        c[r] = acc;
    }
    wave_sync();
    return c;
}
// ds_read_b64_tr_b16 (gfx950 LDS transpose read), per 16-lane group: lane m supplies the 8-byte-aligned address of 4 consecutive
// 16-bit elements; lane i receives element (i & 3) of lanes 4e + (i >> 2), e = 0..3 (guide: with linear addresses lane l, element j
// = lds[(l & 15) + 16 j + 64 (l >> 4)]).  A misaligned address returns the 8-aligned address's data on hardware: refused here.
typedef short emu_s16x4 __attribute__((ext_vector_type(4)));
inline emu_s16x4 ds_read_tr16(const void* p) {
    if (((uintptr_t)p) & 7) { fprintf(stderr, "EMU: ds_read_b64_tr_b16 address not 8-byte aligned\n"); abort(); }
    Wave& w = waves[cur->wave];
    memcpy(w.scratch[cur->lane], p, 8);
    wave_sync();
    const int l = cur->lane, g = l >> 4, i = l & 15;
    emu_s16x4 r;
    for (int e = 0; e < 4; ++e) {
        short v;
        memcpy(&v, w.scratch[16 * g + 4 * e + (i >> 2)] + 2 * (i & 3), 2);
        r[e] = v;
    }
    wave_sync();
    return r;
}
struct Rsrc { const char* p; unsigned bytes; };   // bytes: NUM_RECORDS of a raw buffer (stride 0): loads past it return 0
}  // namespace emu
typedef emu::Rsrc __amdgpu_buffer_rsrc_t;
namespace emu {   // bytes: NUM_RECORDS of a raw buffer (stride 0): loads past it return 0
typedef unsigned int emu_u32x4 __attribute__((ext_vector_type(4)));
// buffer_load_dwordx4 of a raw buffer: the range check covers voffset only (the SGPR offset is outside it, gfx9 ISA)
inline emu_u32x4 buffer_load_b128(Rsrc r, unsigned voff, unsigned soff) {
    emu_u32x4 v = {0u, 0u, 0u, 0u};
    if ((unsigned long long)voff + 16 <= r.bytes) memcpy(&v, r.p + voff + soff, 16);
    return v;
}
template <class T> inline float dot2(T a, T b, float c) {   // v_dot2c_f32_bf16
    uint32_t x, y; memcpy(&x, &a, 4); memcpy(&y, &b, 4);
    float r = c;
    r = fmaf(bf2f((short)(x & 0xffff)), bf2f((short)(y & 0xffff)), r);
    r = fmaf(bf2f((short)(x >> 16)), bf2f((short)(y >> 16)), r);
    return r;
}
inline bool wave_all(bool pred) {
    Wave& w = waves[cur->wave];
    w.scratch[cur->lane][0] = pred ? 1 : 0;
    wave_sync();
    bool r = true;
    for (int i = 0; i < 64; ++i) r = r && w.scratch[i][0];
    wave_sync();
    return r;
}
// v_mov_b32_dpp with row_mask = bank_mask = 0xf and bound_ctrl: lane l reads `src` of the lane the control selects inside its row of 16
// lanes.  Controls built: quad_perm (0x00-0xFF), row_half_mirror (0x141), row_mirror (0x140).
template <class T> inline T shfl_idx(T v, int src);
inline int update_dpp(int old, int src, int ctrl) {
    const int l = cur->lane, row = l & ~15, i = l & 15;
    int from;
    if (ctrl <= 0xFF) from = row + (i & ~3) + ((ctrl >> (2 * (i & 3))) & 3);
    else if (ctrl == 0x141) from = row + (i & 8) + (7 - (i & 7));
    else if (ctrl == 0x140) from = row + (15 - i);
    else { fprintf(stderr, "EMU: dpp control 0x%x not modelled\n", ctrl); abort(); }
    (void)old;
    return shfl_idx(src, from);
}
template <class T> inline T shfl_idx(T v, int src) {
    static_assert(sizeof(T) <= 16, "shuffle operand");
    Wave& w = waves[cur->wave];
    const int ph = cur->shfl_phase;           // lanes shuffle in lockstep, so their phases agree; a lane that races ahead
    cur->shfl_phase ^= 1;                     // deposits into the OTHER buffer, and cannot reach this one again before the
    memcpy(w.shfl[ph][cur->lane], &v, sizeof(T));   // next shuffle's sync, which every lane passes only after this read
    wave_sync();
    T r; memcpy(&r, w.shfl[ph][src & 63], sizeof(T));
    return r;
}
// LDS-DMA: destination = wave-uniform base (first lane's) + lane*size; source address is per lane.
inline void global_load_lds(const void* g, void* lds, unsigned size, int offset, unsigned) {
    Wave& w = waves[cur->wave];
    memcpy(w.scratch[cur->lane], &lds, sizeof(void*));
    wave_sync();
    void* base; memcpy(&base, w.scratch[0], sizeof(void*));
    if (base != lds) { fprintf(stderr, "EMU: global_load_lds with non-uniform LDS base\n"); abort(); }
    wave_sync();
    memcpy((char*)base + offset + cur->lane * size, (const char*)g + offset, size);
}
// raw buffer load to LDS: an offset at or beyond num_records (0x7fffffff here) reads zeros (hardware range check)
inline void buffer_load_lds(Rsrc r, void* lds, unsigned size, unsigned voff, unsigned soff) {
    static const char zeros[16] = {0};
    const bool oob = (unsigned long long)voff + size > r.bytes;      // raw buffer, stride 0: the range check covers voffset only
    global_load_lds(oob ? zeros : r.p + voff + soff, lds, size, 0, 0);
}
}  // namespace emu

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define threadIdx (emu::cur->tid)
#define blockIdx (emu::g_block)
#define blockDim (emu::g_bdim)
#define gridDim (emu::g_gdim)
#define __syncthreads() emu::block_sync()
#define __builtin_amdgcn_s_barrier() emu::block_sync()
#define __builtin_amdgcn_wave_barrier() emu::wave_sync()     /* hardware: no instruction (lanes of a wave run in lockstep); here the wave's fibers meet */
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu::mfma_32x32x16_bf16(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu::mfma_16x16x32_bf16(a, b, c)
#define __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, cbsz, blgp, oa, sa, ob, sb) emu::mfma_32x32x64_f8(a, b, c)
#define __builtin_amdgcn_global_load_lds(g, l, sz, off, aux) emu::global_load_lds((const void*)(g), (void*)(l), sz, off, aux)
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, bytes, flags) emu::Rsrc{(const char*)(p), (unsigned)(bytes)}
#define __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, aux) emu::buffer_load_b128((r), (unsigned)(voff), (unsigned)(soff))
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(r, l, sz, voff, soff, off, aux) emu::buffer_load_lds((r), (void*)(l), sz, (unsigned)(voff), (unsigned)(soff))
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) emu::ds_read_tr16((const void*)(p))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rmask, bmask, bc) emu::update_dpp((old), (src), (ctrl))
#define __builtin_amdgcn_readfirstlane(x) emu::shfl_idx((x), 0)
#define __builtin_amdgcn_fdot2_f32_bf16(a, b, c, cl) emu::dot2((a), (b), (c))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_fetch_add(p, v, order, scope) emu_fetch_add((p), (v))
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_readcyclecounter() 0LL
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(m, n, id) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define __all(p) emu::wave_all(p)
#define __shfl_xor(v, m) emu::shfl_idx((v), emu::cur->lane ^ (m))
#define __shfl_down(v, d) emu::shfl_idx((v), (emu::cur->lane + (d)) > 63 ? emu::cur->lane : emu::cur->lane + (d))
#define __shfl(v, s) emu::shfl_idx((v), (s))
#define __expf(x) expf(x)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __fdividef(a, b) ((a) / (b))
#define __frcp_rn(x) (1.0f / (x))
#define VL2_EMU 1
// v_permlane32_swap_b32 vdst, src (k_gemm.h gemm_store_tr): lanes 32-63 of vdst trade places with lanes 0-31 of src
template <class T> static inline void emu_permlane32_swap(T& vdst, T& src) {
    const int lane = emu::cur->lane;
    const T from_src = emu::shfl_idx(src, lane & 31), from_dst = emu::shfl_idx(vdst, (lane & 31) + 32);
    if (lane >= 32) vdst = from_src; else src = from_dst;
}
#define VL2_PERMLANE32_SWAP_2(a, b) emu_permlane32_swap(a, b)
#define VL2_PERMLANE32_SWAP_4(pk) do { emu_permlane32_swap(pk[0], pk[2]); emu_permlane32_swap(pk[1], pk[3]); } while (0)
#define VL2_LANE_ID_FRESH(ln) do { ln = (int)emu::cur->lane; } while (0)
#define VL2_LDS_I32(off) (*(int*)(vl2_smem + (off)))
#define VL2_TAIL_STORE_BF16(ptr, val) (*(ptr) = (val))
#define VL2_ATOMIC_INC_ASYNC(dst, ptr) do { dst = *(ptr); *(ptr) = dst + 1; } while (0)
#define VL2_PIN3(a, b, c) ((void)0)
#define VL2_PIN2(a, b) ((void)0)
#define VL2_PERMLANE32_SWAP_8(pk) do { emu_permlane32_swap(pk[0], pk[2]); emu_permlane32_swap(pk[1], pk[3]); \
                                       emu_permlane32_swap(pk[4], pk[6]); emu_permlane32_swap(pk[5], pk[7]); } while (0)
// OCP e4m3fn (gfx950's fp8) in software, for csrc/k_fp8.h: v_cvt_pk_f32_fp8 / v_cvt_pk_fp8_f32 (round to nearest even, saturating)
static inline float emu_e4m3fn_to_f32(unsigned b) {
    const unsigned e = (b >> 3) & 15u, m = b & 7u;
    float v = e == 0 ? ldexpf((float)m, -9) : (e == 15 && m == 7 ? NAN : ldexpf(1.0f + (float)m * 0.125f, (int)e - 7));
    return (b & 0x80u) ? -v : v;
}
namespace emu { inline float emu_mx_e4m3(unsigned b) { return emu_e4m3fn_to_f32(b & 0xffu); } }
static inline unsigned emu_f32_to_e4m3fn(float f) {
    if (f != f) return 0x7fu;
    const unsigned sgn = std::signbit(f) ? 0x80u : 0u;
    const float a = fabsf(f);
    if (a >= 464.0f) return sgn | 0x7eu;                              // beyond the last rounding boundary: saturate to 448
    if (a < 0.015625f) return sgn | (unsigned)nearbyintf(ldexpf(a, 9));   // subnormal range: units of 2^-9 (8 units = the first normal, 0x08)
    int ex;
    const float mant = frexpf(a, &ex) * 2.0f;                         // [1, 2)
    ex -= 1;
    unsigned q = (unsigned)nearbyintf((mant - 1.0f) * 8.0f);          // nearbyint: round-half-even in the default rounding mode
    if (q == 8u) { q = 0u; ex += 1; }
    return sgn | ((unsigned)(ex + 7) << 3) | q;
}
typedef float emu_f32x2 __attribute__((ext_vector_type(2)));
static inline emu_f32x2 emu_cvt_pk_f32_fp8(unsigned w, bool hi) {
    const unsigned h = hi ? (w >> 16) : w;
    return emu_f32x2{emu_e4m3fn_to_f32(h & 0xffu), emu_e4m3fn_to_f32((h >> 8) & 0xffu)};
}
static inline int emu_cvt_pk_fp8_f32(float a, float b, int old, bool hi) {
    const unsigned pk = emu_f32_to_e4m3fn(a) | (emu_f32_to_e4m3fn(b) << 8);
    return hi ? (int)(((unsigned)old & 0x0000ffffu) | (pk << 16)) : (int)(((unsigned)old & 0xffff0000u) | pk);
}
#define VL2_CVT_PK_F32_FP8(w, hi) emu_cvt_pk_f32_fp8((unsigned)(w), (hi))
#define VL2_CVT_PK_FP8_F32(a, b, old, hi) emu_cvt_pk_fp8_f32((a), (b), (old), (hi))
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline int emu_fetch_add(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned emu_fetch_add(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
// the harness defines the dynamic LDS array that kernels declare `extern __shared__ ... vl2_smem[]`
// (the build step rewrites `extern __shared__` -> `extern`)
