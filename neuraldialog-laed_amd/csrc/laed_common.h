// Internal helpers shared by the kernels of liblaed_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/laed_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// acc += a (x) b with v_mfma_f32_32x32x2_f32 and the accumulator tile held in AccVGPRs.  Left to itself the compiler
// keeps MFMA accumulators in ArchVGPRs whenever the register budget allows ("VGPR form"); there the MFMA's C/D traffic
// shares the register-file ports with the returning global / LDS loads of the staging pipeline.  With the tiles in
// AccVGPRs the fp32 GEMM gained 2 % over all products and 3.7 % on the dW product (S-fixed step 439.5 -> 431.2 ms, same
// bits).  The constraint "a" is the only way to ask for that form, hence inline asm; the asm is
// opaque to the compiler's MFMA hazard recognizer, so every reader of `acc` that is not another such MFMA on the same
// tile must be preceded by laed_mfma_drain(acc) (XDL write -> VALU / v_accvgpr_read needs 18 wait states after a 16-pass op).
__device__ __forceinline__ void laed_mfma_32x32x2(f32x16& acc, float a, float b) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// What the drain cannot protect: a copy the register allocator itself places behind an MFMA -- e.g. v_accvgpr_mov of a tile on
// the exit edge of a K loop, in FRONT of a drain that follows the loop (seen twice in round 3: gru_step_fwd_kernel with a
// two-entry K loop, and the unaligned-operand sgemm instantiations).  Loops that are not hot therefore drain inside their
// body; tools/check_mfma_hazards.py scans every built library for the pattern (tests/test_isa_cpu.py).
// The drain carries a DEPENDENCY on every tile it protects: each tile is an in/out operand ("+a") of the drain's asm
// statements, so any later read of the tile (v_accvgpr_read, extractelement) uses the value the drain "produced" and
// cannot be scheduled in front of the s_nops -- a memory clobber or `volatile` alone orders memory operations and other
// volatile asm, not register-only instructions.  The first statement holds the wait states; the statements for the other
// tiles are empty and ordered behind it (volatile asm statements keep their program order).
__device__ __forceinline__ void laed_mfma_drain_first(f32x16& acc) {
    asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc));
}
__device__ __forceinline__ void laed_mfma_drain_next(f32x16& acc) {
    asm volatile("" : "+a"(acc));
}
template <int N>
__device__ __forceinline__ void laed_mfma_drain(f32x16 (&acc)[N]) {
    laed_mfma_drain_first(acc[0]);
#pragma unroll
    for (int i = 1; i < N; ++i) laed_mfma_drain_next(acc[i]);
}
template <int M, int N>
__device__ __forceinline__ void laed_mfma_drain(f32x16 (&acc)[M][N]) {
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (i == 0 && j == 0) laed_mfma_drain_first(acc[0][0]);
            else laed_mfma_drain_next(acc[i][j]);
        }
}

char* laed_err_buf();   // thread-local, 256 bytes (laed_api.hip)

static inline int laed_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(laed_err_buf(), 256, fmt, ap);
    va_end(ap);
    return code;
}

#define LAED_CHECK_ARG(cond, ...) \
    do { if (!(cond)) return laed_fail(LAED_E_BADARG, __VA_ARGS__); } while (0)

#define LAED_LAUNCH_CHECK(name) \
    do { hipError_t e_ = hipGetLastError(); \
         if (e_ != hipSuccess) return laed_fail((int)e_, "%s launch failed: %s", name, hipGetErrorString(e_)); } while (0)

#define LAED_TRY(expr) do { int s_ = (expr); if (s_ != 0) return s_; } while (0)

#define LAED_HIP_TRY(expr) \
    do { hipError_t e_ = (expr); \
         if (e_ != hipSuccess) return laed_fail((int)e_, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

// "done once per device" flag for hipFuncSetAttribute (function attributes are per device): true the first time it is
// asked about the current device.  A benign race repeats the idempotent call at worst.
static inline bool laed_first_use_on_device(uint64_t* mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if ((*mask >> dev) & 1ull) return false;
    *mask |= (1ull << dev);
    return true;
}

// EXPERIMENT hooks (tile / split overrides, kernel-variant switches used by tools/*.py sweeps) are read from the
// environment only in a hooks build (`make HOOKS=1`, -DLAED_ENV_HOOKS -> liblaed_hip_hooks.so, selected with
// LAED_HIP_LIB); the product library compiles them to their defaults.  The four switches the product does read are
// listed in INTEGRATION.md: LAED_GEMM_MODE, LAED_GRU_SMALL, LAED_GRU_PERSIST (operational) and LAED_GRU_WM (test hook).
#ifdef LAED_ENV_HOOKS
static inline const char* laed_exp_hook(const char* name) { return getenv(name); }
#else
static inline const char* laed_exp_hook(const char*) { return nullptr; }
#endif

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// internal entry shared by gru / proj_nll / api
int laed_sgemm_impl(bool transA, bool transB, int M, int N, int K,
                    const float* A, int64_t lda, const float* B, int64_t ldb,
                    float* C, int64_t ldc, const float* bias, bool accumulate,
                    float* colsum_a, bool colsum_accumulate, hipStream_t stream);

// split-K variant: `ws` (ws_floats floats) lets short-M/N, long-K problems fill the chip.  raw_partials: leave the
// S partial products in ws[S][M][N] (no bias / accumulate / C write) and report S in *splits_out (S == 1: the
// product was written to C as usual).
int laed_sgemm_splitk_impl(bool transA, bool transB, int M, int N, int K,
                           const float* A, int64_t lda, const float* B, int64_t ldb,
                           float* C, int64_t ldc, const float* bias, bool accumulate,
                           float* colsum_a, bool colsum_accumulate,
                           float* ws, int64_t ws_floats, bool raw_partials, int* splits_out, hipStream_t stream);

int laed_sgemm_softmax_stats_impl(int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                                  const float* bias, const int64_t* labels, float* part, float* lab_logit, hipStream_t stream);
int laed_sgemm_argmax_impl(int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                           const float* bias, float* part, int64_t* ids, hipStream_t stream);

extern "C" int laed_get_gemm_mode(void);
int laed_xgemm_nt_launch(int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                         float* C, int64_t ldc, const float* bias, bool accumulate,
                         int splitk, int kper, float* ws, bool vec, float* rowsum_a, bool rowsum_acc, hipStream_t stream);

int laed_transpose_impl(const float* src, int64_t ld, int R, int C, float* dst, int64_t ldd, hipStream_t stream);
// C[M,N] (+)= A[M,K] B[N,K]^T and rowsum[m] (+)= sum_k A[m][k], in the current GEMM math mode
int laed_sgemm_rowsum_impl(int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                           float* C, int64_t ldc, bool accumulate, float* rowsum, bool rowsum_acc,
                           float* ws, int64_t ws_floats, hipStream_t stream);

int laed_gru_step_impl(const float* hprev, const float* w_hh, const float* b_hh, float* gates, float* hout, float* hn,
                       int B, int H, hipStream_t stream);

// small-batch GRU recurrences (gru_small.hip): B <= 64, H % 16 == 0; ws 16-byte aligned, laed_gru_small_*_ws floats
bool laed_gru_small_ok(int B, int H);
int64_t laed_gru_small_fwd_ws(int B, int H);
int64_t laed_gru_small_bwd_ws(int B, int H);
// cu_budget: CUs a persistent (all-workgroups-resident) grid may count on; 0 = the whole device
int laed_persist_cu_limit(int cu_budget);
// h0 / dhT: initial state / incoming gradient of the last state (nullptr = zeros); hbuf[0] / dh0 are initialised inside
int laed_gru_small_fwd(const float* w_hh, const float* b_hh, const float* h0, float* hbuf, float* gates, float* hn, int T, int B,
                       int H, float* ws, int cu_budget, hipStream_t stream);
int laed_gru_small_bwd(const float* w_hh, const float* hbuf, const float* hn, const float* dout, const float* dhT, float* gates,
                       float* dgh, float* dh0, int T, int B, int H, float* ws, int cu_budget, hipStream_t stream);

// Gate functions of the GRU kernels: branch-free and short (a recurrence step applies 3 of them per (row, unit), in the
// epilogue of an MFMA kernel where every VALU instruction is exposed).  v_exp_f32 / v_rcp_f32 are 1-ulp instructions;
// checked against float64 over [-30, 30] (tests/test_kernels_gpu.py::test_gate_functions): absolute error < 1.2e-7 for
// both, relative < 4e-7 for tanh everywhere and for sigmoid on |x| <= 4 (its far negative tail carries the rounding of
// x log2 e: relative |x| 1e-7 on values below 2e-2).  libm's expf / tanhf / IEEE division cost ~140 instructions per (row, unit) with two
// data-dependent branches; these cost ~35.
// Streaming (non-temporal, `nt`) 16-byte accesses for data that is touched once per pass and is far larger than any cache
// (the softmax-CE pass over an [rows][V] logits chunk: 8 GB): 3.5 % off that kernel (3083 -> 2975 us per 40960-row chunk,
// tools/exp/rows_ab.sh), neutral where the chunk is small enough to stay in the Infinity Cache (V = 10 k, B = 30).
__device__ __forceinline__ float4 load_stream(const float4* p) {
    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ void store_stream(float4* p, float4 v) {
    f32x4 t;
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
}

__device__ __forceinline__ float sigmoidf_acc(float x) {
    // 1 / (1 + 2^(-x log2 e)): 2^(+big) = inf -> 0, 2^(-big) = 0 -> 1, NaN propagates
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
__device__ __forceinline__ float tanhf_gate(float x) {
    const float ax = fabsf(x);
    // |x| >= 0.625: 1 - 2 / (e^{2|x|} + 1)
    const float t = __builtin_amdgcn_exp2f(ax * 2.88539008177792681f);
    const float big = fmaf(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);
    // |x| < 0.625: x + x^3 P(x^2), P fitted to (tanh(x) / x - 1) / x^2 on [0, 0.625] (relative error 8e-8 in fp32)
    const float s = ax * ax;
    float p = fmaf(s, -0.005704042501747608f, 0.020637862384319305f);
    p = fmaf(s, p, -0.05373915657401085f);
    p = fmaf(s, p, 0.133314311504364f);
    p = fmaf(s, p, -0.3333328068256378f);
    const float small = fmaf(ax * s, p, ax);
    return copysignf(ax < 0.625f ? small : big, x);
}

// Lane exchange for a butterfly reduction over the 32 lanes that share lane>>5, stage `o` = 1, 2, 4, 8, 16: the partner of
// __shfl_xor(v, o), but by DPP (a VALU modifier: no LDS crossbar round trip, no lgkmcnt wait) for the four stages inside a
// row of 16 lanes.  Stages 4 and 8 use the mirror permutations, which pair the same GROUPS as xor 4 / xor 8; since every
// lane of a group already holds the group's value after the earlier stages, the reduction tree -- and for sums the
// association order -- is exactly the xor butterfly's.  Only the 16-stage crosses rows and stays a ds_bpermute.
template <int O>
__device__ __forceinline__ int laed_bfly32_i(int v) {
    if (O == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
    if (O == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
    if (O == 4) return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);     // row_half_mirror
    if (O == 8) return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);     // row_mirror
    return __shfl_xor(v, 16, 64);
}
template <int O>
__device__ __forceinline__ float laed_bfly32(float v) { return __int_as_float(laed_bfly32_i<O>(__float_as_int(v))); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// counter-based RNG: splitmix64-style hash of (seed, index) -> uniform in [0,1)
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t idx) {
    uint64_t r = mix64(mix64(seed) ^ (idx * 0xD6E8FEB86659FD93ull));
    return (float)(r >> 40) * (1.0f / 16777216.0f);   // 24 random bits -> [0,1)
}
