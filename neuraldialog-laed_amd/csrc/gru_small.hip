// GRU timestep kernels for SMALL batches (B <= 64: the reference's own script shapes, B = 30 / 60) and, as row groups of
// 64, for mid-size ones (the 300 utterances of an AE-ED / ST-ED batch).
//
// At these sizes a timestep is ~50 MFLOP against 3-12 MB of recurrent weights: what matters is how many CUs pull
// W_hh out of L2 at once and how short the dependent chain inside the kernel is, not MFMA throughput.  The tiled
// kernel of gru.hip gives such a step 16 workgroups that each stream 196 KB through LDS (24.5 us at B = 30, H = 512).
// Here a workgroup owns 4 (forward) / 8 (backward) hidden units, so the grid is H/4 or H/8 workgroups, and no
// operand is staged in LDS: both MFMA operands are kept in memory in the order the lanes consume them.
//   * v_mfma_f32_16x16x4_f32: lane (i = lane % 16, kq = lane / 16) supplies A[i][kq] and B[kq][i].  The contraction
//     is order-free, so lane kq takes a whole QUAD of consecutive k-values (one float4) and feeds its four elements
//     to four successive MFMAs -- both operands use the same permutation.
//   * activations travel between the steps in "quad-major" form X[K/4][BP][4] (BP = batch rows padded to 16): the
//     16 lanes of a row tile read 256 contiguous bytes.  Each step's epilogue writes its units' slice of the next
//     step's operand in that form (besides the plain [B][H] tensors the rest of the library reads).
//   * W_hh is packed once per call into per-workgroup P[K/4][NC][4] slabs (NC = 12 forward: r,z,n of 4 units;
//     8 backward), again one float4 per lane per quad.
//   * 8 waves split K; the partial 16x16 tiles meet in LDS, one thread per (batch row, hidden unit) sums them in a
//     fixed order and applies the gate math.
// forward : one launch per step, W_hh[3H][H] read once per step by H/4 workgroups.
// backward: ONE launch per step instead of two: workgroup j-slice forms dh_t[:, slice] = carry + dgh_{t+1} W_hh[:, slice]
//           (needs all of dgh_{t+1}: the kernel boundary) and goes straight on to the gate gradients of step t for
//           its slice.
#include "laed_common.h"
#include <math.h>
#include <stdlib.h>
#include <atomic>

namespace {

constexpr int NTH = 512;        // 8 waves
constexpr int NW = NTH / 64;
constexpr int UF = 4;           // hidden units per workgroup, forward  (12 of 16 MFMA columns)
constexpr int UB = 8;           // hidden units per workgroup, backward ( 8 of 16 MFMA columns)
constexpr int CH = 4;           // quad groups (16 k-values each) per load chunk

// acc[rt] += X[rows 16 rt + i][k] * P[k][col i] over this wave's share of K (ngrp groups of 4 quads).
template <int RT, int NC>
__device__ __forceinline__ void small_core(const float* __restrict__ x, const float* __restrict__ pw, int ngrp, int wave, int lane,
                                           f32x4 (&acc)[RT]) {
    constexpr int BP = 16 * RT;
    const int li = lane & 15, kq = lane >> 4;
    const bool bval = li < NC;
    const int q0 = wave * 4 * ngrp + kq;
    const float4* ap = reinterpret_cast<const float4*>(x) + (int64_t)q0 * BP + li;
    const float4* bp = reinterpret_cast<const float4*>(pw) + (int64_t)q0 * NC + (bval ? li : 0);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a0[CH][RT], b0[CH], a1[CH][RT], b1[CH];
    auto ld = [&](float4 (&a)[CH][RT], float4 (&b)[CH], int g0) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int g = g0 + c;
            if (g < ngrp) {                       // wave-uniform
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a[c][rt] = ap[(int64_t)4 * g * BP + rt * 16];
                b[c] = bval ? bp[(int64_t)4 * g * NC] : z4;
            }
        }
    };
    auto mm = [&](const float4 (&a)[CH][RT], const float4 (&b)[CH], int g0) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (g0 + c < ngrp) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][rt].x, b[c].x, acc[rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][rt].y, b[c].y, acc[rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][rt].z, b[c].z, acc[rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][rt].w, b[c].w, acc[rt], 0, 0, 0);
            }
        }
    };
    ld(a0, b0, 0);
    for (int g0 = 0; g0 < ngrp; g0 += 2 * CH) {       // chunk g0+CH is in flight while chunk g0 multiplies
        if (g0 + CH < ngrp) ld(a1, b1, g0 + CH);
        mm(a0, b0, g0);
        if (g0 + 2 * CH < ngrp) ld(a0, b0, g0 + 2 * CH);
        if (g0 + CH < ngrp) mm(a1, b1, g0 + CH);
    }
}

// partial tiles -> LDS red[wave][row BP][NC]   (C/D map of 16x16x4: col = lane % 16, row = 4 * (lane / 16) + e)
template <int RT, int NC>
__device__ __forceinline__ void spill_partials(float* red, const f32x4 (&acc)[RT], int wave, int lane) {
    constexpr int BP = 16 * RT;
    const int col = lane & 15, kq = lane >> 4;
    if (col < NC) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(wave * BP + rt * 16 + 4 * kq + e) * NC + col] = acc[rt][e];
    }
}

template <int RT, int NC>
__device__ __forceinline__ float sum_partials(const float* red, int nkw, int row, int col) {
    constexpr int BP = 16 * RT;
    float s = 0.f;
    for (int w = 0; w < nkw; ++w) s += red[(w * BP + row) * NC + col];   // fixed order
    return s;
}

struct SmallFwdArgs {
    const float* hx_in;     // [H/4][BP][4] h_{t-1}, quad-major
    const float* wp;        // [H/UF][H/4][3 UF][4] packed W_hh
    const float* b_hh;
    const float* hprev;     // [B][H]
    float* gates;           // [B][3H] in: gi (+b_ih), out: r,z,n
    float* hout;            // [B][H]
    float* hn;              // [B][H] or NULL
    float* hx_out;          // [H/4][BP][4] h_t, quad-major
    int B, H, nkw;          // nkw waves split K (H % (16 nkw) == 0)
};

template <int RT>
__global__ __launch_bounds__(NTH) void gru_small_fwd_kernel(SmallFwdArgs p) {
    constexpr int NC = 3 * UF, BP = 16 * RT;
    __shared__ float red[NW * BP * NC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, B = p.B, j0 = blockIdx.x * UF;
    // batches of more than 64 rows run as row groups of BP rows (blockIdx.y), each with its own quad-major operand
    const int64_t xg = (int64_t)blockIdx.y * H * BP;
    // epilogue operands of this thread's (row, unit) pair: fetched now, needed after the reduction
    const int row = tid / UF, u = tid % UF, j = j0 + u;
    const int grow = blockIdx.y * BP + row;
    const bool mine = row < BP && grow < B;
    float gi_r = 0.f, gi_z = 0.f, gi_n = 0.f, hp = 0.f, bhr = 0.f, bhz = 0.f, bhn = 0.f;
    float* g = p.gates + (int64_t)grow * 3 * H;
    if (mine) {
        gi_r = g[j]; gi_z = g[H + j]; gi_n = g[2 * H + j];
        hp = p.hprev[(int64_t)grow * H + j];
        bhr = p.b_hh[j]; bhz = p.b_hh[H + j]; bhn = p.b_hh[2 * H + j];
    }
    if (wave < p.nkw) {
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[rt][e] = 0.f;
        small_core<RT, NC>(p.hx_in + xg, p.wp + (int64_t)blockIdx.x * H * NC, H / 16 / p.nkw, wave, lane, acc);
        spill_partials<RT, NC>(red, acc, wave, lane);
    }
    __syncthreads();
    if (row >= BP) return;
    float h = 0.f;
    if (mine) {
        const float ghr = sum_partials<RT, NC>(red, p.nkw, row, u), ghz = sum_partials<RT, NC>(red, p.nkw, row, UF + u);
        const float ghn = sum_partials<RT, NC>(red, p.nkw, row, 2 * UF + u) + bhn;
        const float r = sigmoidf_acc(gi_r + ghr + bhr);
        const float z = sigmoidf_acc(gi_z + ghz + bhz);
        const float n = tanhf_gate(gi_n + r * ghn);
        h = (1.0f - z) * n + z * hp;
        p.hout[(int64_t)grow * H + j] = h;
        if (p.hn != nullptr) {
            g[j] = r; g[H + j] = z; g[2 * H + j] = n;
            p.hn[(int64_t)grow * H + j] = ghn;
        }
    }
    p.hx_out[xg + ((int64_t)(j / 4) * BP + row) * 4 + (j & 3)] = h;      // padding rows carry zeros
}

struct SmallBwdArgs {
    const float* dx_in;      // [3H/4][BP][4] dgh of step t+1, quad-major, or NULL at the first (last-time) step
    const float* wpT;        // [H/UB][3H/4][UB][4] packed W_hh^T
    const float* dout;       // [B][H] or NULL
    const float* hprev;      // [B][H]
    const float* hn;         // [B][H]
    float* gates;            // [B][3H] in: r,z,n  out: d(gi)
    float* dgh;              // [B][3H] out (step t)
    float* dx_out;           // [3H/4][BP][4] out (step t), quad-major
    float* dh;               // [B][H] carry: in dh_{t+1}*z_{t+1} (without the W_hh term), out dh_t*z_t; final: out dh_0
    int B, H, nkw, final_step;
};

template <int RT>
__global__ __launch_bounds__(NTH) void gru_small_bwd_kernel(SmallBwdArgs p) {
    constexpr int NC = UB, BP = 16 * RT;
    __shared__ float red[NW * BP * NC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, B = p.B, j0 = blockIdx.x * UB, K = 3 * H;
    const int64_t xg = (int64_t)blockIdx.y * K * BP;          // row group (see the forward kernel)
    const int row = tid / UB, u = tid % UB, j = j0 + u;
    const int grow = blockIdx.y * BP + row;
    const bool mine = row < BP && grow < B;
    const int64_t i = (int64_t)grow * H + j;
    float dh = 0.f, r = 0.f, z = 0.f, nn = 0.f, hnv = 0.f, hp = 0.f;
    float* g = p.gates + (int64_t)grow * 3 * H;
    if (mine) {
        dh = p.dh[i];
        if (!p.final_step) {
            if (p.dout != nullptr) dh += p.dout[i];
            r = g[j]; z = g[H + j]; nn = g[2 * H + j];
            hnv = p.hn[i]; hp = p.hprev[i];
        }
    }
    if (p.dx_in != nullptr) {        // block-uniform
        if (wave < p.nkw) {
            f32x4 acc[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[rt][e] = 0.f;
            small_core<RT, NC>(p.dx_in + xg, p.wpT + (int64_t)blockIdx.x * K * NC, K / 16 / p.nkw, wave, lane, acc);
            spill_partials<RT, NC>(red, acc, wave, lane);
        }
        __syncthreads();
        if (mine) dh += sum_partials<RT, NC>(red, p.nkw, row, u);
    }
    if (row >= BP) return;
    if (p.final_step) { if (mine) p.dh[i] = dh; return; }
    float drp = 0.f, dzp = 0.f, dnr = 0.f;
    if (mine) {
        float* d = p.dgh + (int64_t)grow * 3 * H;
        const float dn = dh * (1.0f - z);
        const float dz = dh * (hp - nn);
        const float dnp = dn * (1.0f - nn * nn);
        dzp = dz * z * (1.0f - z);
        drp = dnp * hnv * r * (1.0f - r);
        dnr = dnp * r;
        g[j] = drp; g[H + j] = dzp; g[2 * H + j] = dnp;
        d[j] = drp; d[H + j] = dzp; d[2 * H + j] = dnr;
        p.dh[i] = dh * z;
    }
    const int e4 = j & 3;
    p.dx_out[xg + ((int64_t)(j / 4) * BP + row) * 4 + e4] = drp;           // padding rows carry zeros
    p.dx_out[xg + ((int64_t)((H + j) / 4) * BP + row) * 4 + e4] = dzp;
    p.dx_out[xg + ((int64_t)((2 * H + j) / 4) * BP + row) * 4 + e4] = dnr;
}

// ---- persistent forward: ALL T steps in one launch, the workgroup's W_hh slab resident in LDS across timesteps.
// Workgroups exchange h_t through the quad-major ping-pong buffers inside the launch:
//   producer: write-through (sc1) stores of its slice -> every wave s_waitcnt vmcnt(0) -> workgroup barrier -> one lane
//             adds 1 to an agent-scope arrival counter;
//   consumer: one lane polls the counter (sc1 loads, s_sleep, BOUNDED: on timeout it raises sync[1] and the workgroup
//             stops waiting, so the grid always drains) -> workgroup barrier -> sc1 buffer loads of h_{t-1}.
// (MI355X_MICROARCH.md, inter-workgroup visibility, "valid forms" table row 3: each 128-byte line of the payload is
// written whole by one store instruction of one wave; one workgroup per CU, forced by the LDS request.)
// h_{t-1} of a workgroup's own units stays in registers.  Needs all H/UF workgroups resident at once (H/UF <= 256).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int WS4 = 16;             // float4 slots per quad in the LDS slab (12 used: keeps the kq groups on disjoint banks)
// Arrival counters of the persistent launches.  A group's workgroups used to add to ONE counter: agent-scope atomics on one
// word are served one after the other (11-13 ns each, MI355X_MICROARCH.md "fanin"), i.e. ~1.5 us of a 4-5 us step for the 128
// workgroups of H = 512.  The counter is therefore kept in up to MAX_SHARDS shards, each on a 128-byte line of its own:
// workgroup x adds to shard x % nsh, and the polling wave reads every shard with one load instruction (lane s polls shard s)
// and goes on when all of them have reached their own targets ("valid forms" row 1: a sharded counter, every shard polled).
// Layout of `sync` (SYNC_WORDS unsigned, zeroed by the preparation launch): word 1 = timeout flag; shard s of group g at
// word 32 * (1 + g * MAX_SHARDS + s).
constexpr int MAX_SHARDS = 8, MAX_PGROUPS = 4;
constexpr int SYNC_WORDS = 32 * (1 + MAX_PGROUPS * MAX_SHARDS);
constexpr unsigned SPIN_LIMIT = 1000000;    // ~1-2 s of polling (s_sleep + one L2 round trip each): only a grid that can never become resident gets there

// Sticky per-device count of timed-out waits (besides the per-call sync[1] word): the host reads it with
// laed_gru_persist_timeouts() whenever it next synchronises anyway (e.g. when it sees a NaN loss), so a poisoned output
// can be told from an ordinary numerical blow-up.  Written with a vector atomic (agent scope); never reset by kernels.
__device__ unsigned g_persist_timeouts = 0;

// Publishing a workgroup's slice of the quad-major operand: the 4 lanes of a quad hold the 4 floats of one (quad, row) slot;
// lane 0 of the quad gathers them (DPP quad broadcasts) and writes ONE 16-byte write-through (sc1) store instead of four
// 4-byte ones -- an sc1 store is one fabric write whatever its size (MI355X_MICROARCH.md: a dword sc1 store costs ~6x the
// dwordx4 time per byte), and the step waits for every one of them to be acknowledged before it may signal.
__device__ __forceinline__ void publish_quad(__amdgpu_buffer_rsrc_t out, int byte_off, float v, bool leader) {
    const int iv = __float_as_int(v);
    u32x4 d;
    d.x = (unsigned)__builtin_amdgcn_mov_dpp(iv, 0x00, 0xF, 0xF, true);      // quad_perm [0,0,0,0]
    d.y = (unsigned)__builtin_amdgcn_mov_dpp(iv, 0x55, 0xF, 0xF, true);      // [1,1,1,1]
    d.z = (unsigned)__builtin_amdgcn_mov_dpp(iv, 0xAA, 0xF, 0xF, true);      // [2,2,2,2]
    d.w = (unsigned)__builtin_amdgcn_mov_dpp(iv, 0xFF, 0xF, 0xF, true);      // [3,3,3,3]
    if (leader) __builtin_amdgcn_raw_buffer_store_b128(d, out, byte_off, 0, 16);                          // aux 16 = sc1
}

// Wave 0 of a workgroup waits until every shard of its group's counter has reached (workgroups of that shard) x step.
// BOUNDED: on timeout the flag word and the sticky device counter are raised and *s_dead set, so the grid always drains.
__device__ __forceinline__ void persist_wait(unsigned* sync, unsigned* cnt, int nsh, unsigned G, unsigned step, int lane,
                                             unsigned* s_dead) {
    const bool poller = lane < nsh;
    const unsigned mine = G / (unsigned)nsh + ((unsigned)lane < G % (unsigned)nsh ? 1u : 0u);   // workgroups x with x % nsh == lane
    const unsigned target = mine * step;
    unsigned* const c = cnt + 32 * (poller ? lane : 0);
    unsigned spins = 0;
    for (;;) {
        const unsigned v = poller ? __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (__all(!poller || v >= target)) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > SPIN_LIMIT) {
            if (lane == 0) {
                __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&g_persist_timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_dead = 1;
            }
            break;
        }
    }
}

// rows: the group's rows that carry data (<= BP).  Lanes of rows beyond it load nothing (their operand is zero): a group of 8
// rows pulls half the bytes of a 16-row one across the fabric, and that pull is what a step of these kernels waits for.
template <int RT, int NC, int WS>
__device__ __forceinline__ void persist_core(__amdgpu_buffer_rsrc_t x, const float* __restrict__ slab, int ngrp, int wave, int lane,
                                             int rows, f32x4 (&acc)[RT]) {
    constexpr int BP = 16 * RT;
    const int li = lane & 15, kq = lane >> 4;
    const bool bval = li < NC;
    const int q0 = wave * 4 * ngrp + kq;
    const int aoff = (q0 * BP + li) * 16;                                        // bytes
    const float4* bp = reinterpret_cast<const float4*>(slab) + q0 * WS + (bval ? li : 0);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g0 = 0; g0 < ngrp; g0 += CH) {
        u32x4 a[CH][RT];
        float4 b[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int g = g0 + c;
            if (g < ngrp) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    a[c][rt] = u32x4{0u, 0u, 0u, 0u};
                    if (rt * 16 + li < rows) a[c][rt] = __builtin_amdgcn_raw_buffer_load_b128(x, aoff + (4 * g * BP + rt * 16) * 16, 0, 16);
                }
                b[c] = bval ? bp[4 * g * WS] : z4;
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (g0 + c < ngrp) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[c][rt].x), b[c].x, acc[rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[c][rt].y), b[c].y, acc[rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[c][rt].z), b[c].z, acc[rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[c][rt].w), b[c].w, acc[rt], 0, 0, 0);
            }
        }
    }
}

struct PersistFwdArgs {
    float* hx0;             // [H/4][BP][4]: h0 on entry (quad-major); ping
    float* hx1;             //               pong
    const float* wp;        // [H/UF][H/4][3 UF][4] packed W_hh
    const float* b_hh;
    float* hbuf;            // [(T+1)][B][H], hbuf[0] = h0
    float* gates;           // [T][B][3H]
    float* hn;              // [T][B][H] or NULL
    unsigned* sync;         // SYNC_WORDS words: timeout flag + sharded arrival counters; zeroed before the launch
    int T, B, H, nkw, nsh;
};

template <int RT>
__global__ __launch_bounds__(NTH) void gru_small_persist_fwd_kernel(PersistFwdArgs p) {
    constexpr int NC = 3 * UF, BP = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, B = p.B, j0 = blockIdx.x * UF;
    float* slab = lds;                               // [H/4][WS4][4]
    float* red = lds + (int64_t)H * WS4;             // [NW][BP][NC]
    {   // the workgroup's W_hh slab: HBM/L2 -> LDS once for all T steps
        const float4* src = reinterpret_cast<const float4*>(p.wp + (int64_t)blockIdx.x * H * NC);
        for (int i = tid; i < (H / 4) * NC; i += NTH) reinterpret_cast<float4*>(slab)[(i / NC) * WS4 + i % NC] = src[i];
    }
    // row groups (blockIdx.y): the sequences of a batch are independent, so each group of BP rows is a recurrence of its
    // own -- own ping-pong operand, own arrival counter -- and a workgroup reads only its group's rows of h_{t-1}
    const int grp = blockIdx.y;
    const int row = tid / UF, u = tid % UF, j = j0 + u;
    const int grow = grp * BP + row;
    const bool mine = row < BP && grow < B;
    float hp = 0.f, bhr = 0.f, bhz = 0.f, bhn = 0.f;
    if (mine) {
        hp = p.hbuf[(int64_t)grow * H + j];
        bhr = p.b_hh[j]; bhz = p.b_hh[H + j]; bhn = p.b_hh[2 * H + j];
    }
    float* const hx0 = p.hx0 + (int64_t)grp * H * BP;
    float* const hx1 = p.hx1 + (int64_t)grp * H * BP;
    unsigned* const cnt = p.sync + 32 * (1 + grp * MAX_SHARDS);       // this group's shards, a 128-byte line each
    const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc(hx0, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(hx1, 0, 0x7fffffff, 0x00020000);
    const unsigned G = gridDim.x;
    bool dead = false;                               // set (per workgroup, via LDS) once a wait has timed out
    __shared__ unsigned s_dead;
    if (tid == 0) s_dead = 0;
    __syncthreads();
    for (int t = 0; t < p.T; ++t) {
        float* g = p.gates + ((int64_t)t * B + grow) * 3 * H;
        float gi_r = 0.f, gi_z = 0.f, gi_n = 0.f;
        if (mine) { gi_r = g[j]; gi_z = g[H + j]; gi_n = g[2 * H + j]; }
        if (t > 0) {                                 // every workgroup's slice of h_{t-1} must have been published
            if (wave == 0 && !dead) persist_wait(p.sync, cnt, p.nsh, G, (unsigned)t, lane, &s_dead);
            __syncthreads();
            dead = s_dead != 0;
        }
        if (wave < p.nkw) {
            f32x4 acc[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[rt][e] = 0.f;
            persist_core<RT, NC, WS4>((t & 1) ? rx1 : rx0, slab, H / 16 / p.nkw, wave, lane, BP, acc);
            spill_partials<RT, NC>(red, acc, wave, lane);
        }
        __syncthreads();
        if (row < BP) {
            float h = 0.f;
            if (mine) {
                const float ghr = sum_partials<RT, NC>(red, p.nkw, row, u), ghz = sum_partials<RT, NC>(red, p.nkw, row, UF + u);
                const float ghn = sum_partials<RT, NC>(red, p.nkw, row, 2 * UF + u) + bhn;
                const float r = sigmoidf_acc(gi_r + ghr + bhr);
                const float z = sigmoidf_acc(gi_z + ghz + bhz);
                const float n = tanhf_gate(gi_n + r * ghn);
                h = (1.0f - z) * n + z * hp;
                if (dead) h = __builtin_nanf("");         // a timed-out wait poisons the output: loud, not silently stale
                hp = h;
                p.hbuf[((int64_t)(t + 1) * B + grow) * H + j] = h;
                if (p.hn != nullptr) {
                    g[j] = r; g[H + j] = z; g[2 * H + j] = n;
                    p.hn[((int64_t)t * B + grow) * H + j] = ghn;
                }
            }
            static_assert(UF == 4, "a workgroup's units are one quad of the operand");
            publish_quad((t & 1) ? rx0 : rx1, (int)(((int64_t)blockIdx.x * BP + row) * 16), h, u == 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores ...
        __syncthreads();                                           // ... before one lane signals for the workgroup
        if (tid == 0) __hip_atomic_fetch_add(cnt + 32 * (blockIdx.x % p.nsh), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- persistent backward: the whole reverse recurrence in one launch, the workgroup's W_hh^T slab resident in LDS.
// Same hand-off as the forward kernel; step s = T-1-t publishes dgh_t (quad-major, sc1) and step s+1 consumes it.
// The dh carry of a workgroup's own (row, unit) pairs stays in registers.  Step s = T has no gate math: it only
// forms dh_0 = carry + dgh_0 W_hh.
struct PersistBwdArgs {
    float* dx0;              // [3H/4][BP][4] ping
    float* dx1;              //               pong
    const float* wpT;        // [H/UB][3H/4][UB][4] packed W_hh^T
    const float* dout;       // [T][B][H] or NULL
    const float* hbuf;       // [(T+1)][B][H]
    const float* hn;         // [T][B][H]
    float* gates;            // [T][B][3H] in: r,z,n  out: d(gi)
    float* dgh;              // [T][B][3H] out
    float* dh;               // [B][H] in: dh_T, out: dh_0
    unsigned* sync;          // SYNC_WORDS words: timeout flag + sharded arrival counters; zeroed before the launch
    int T, B, H, nkw, nsh;
    int rpg;                 // batch rows per row group (<= BP = 16 RT; 8 = half-filled 16-row tiles, half the operand pull)
};

template <int RT>
__global__ __launch_bounds__(NTH) void gru_small_persist_bwd_kernel(PersistBwdArgs p) {
    constexpr int NC = UB, BP = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, B = p.B, j0 = blockIdx.x * UB, K = 3 * H;
    float* slab = lds;                               // [3H/4][UB][4]
    float* red = lds + (int64_t)K * NC;              // [NW][BP][NC]
    {
        const float4* src = reinterpret_cast<const float4*>(p.wpT + (int64_t)blockIdx.x * K * NC);
        for (int i = tid; i < (K / 4) * NC; i += NTH) reinterpret_cast<float4*>(slab)[i] = src[i];
    }
    const int grp = blockIdx.y;                       // row group: see the forward kernel
    const int row = tid / UB, u = tid % UB, j = j0 + u;
    const int grow = grp * p.rpg + row;
    const bool mine = row < p.rpg && grow < B;
    const int64_t i = (int64_t)grow * H + j;
    float carry = mine ? p.dh[i] : 0.f;
    float* const dx0 = p.dx0 + (int64_t)grp * K * BP;
    float* const dx1 = p.dx1 + (int64_t)grp * K * BP;
    unsigned* const cnt = p.sync + 32 * (1 + grp * MAX_SHARDS);
    const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc(dx0, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(dx1, 0, 0x7fffffff, 0x00020000);
    const unsigned G = gridDim.x;
    bool dead = false;
    __shared__ unsigned s_dead;
    if (tid == 0) s_dead = 0;
    __syncthreads();
    for (int s = 0; s <= p.T; ++s) {
        const int t = p.T - 1 - s;                   // t = -1: the closing step
        const bool fin = t < 0;
        float dh = carry, r = 0.f, z = 0.f, nn = 0.f, hnv = 0.f, hp = 0.f;
        float* g = fin ? nullptr : p.gates + ((int64_t)t * B + grow) * 3 * H;
        if (mine && !fin) {
            const int64_t it = (int64_t)t * B * H + i;
            if (p.dout != nullptr) dh += p.dout[it];
            r = g[j]; z = g[H + j]; nn = g[2 * H + j];
            hnv = p.hn[it]; hp = p.hbuf[it];
        }
        if (s > 0) {                                 // every workgroup's slice of dgh_{t+1} must have been published
            if (wave == 0 && !dead) persist_wait(p.sync, cnt, p.nsh, G, (unsigned)s, lane, &s_dead);
            __syncthreads();
            dead = s_dead != 0;
            if (wave < p.nkw) {
                f32x4 acc[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[rt][e] = 0.f;
                persist_core<RT, NC, UB>((s & 1) ? rx0 : rx1, slab, K / 16 / p.nkw, wave, lane, p.rpg, acc);     // step s-1 wrote dx[(s-1)&1]
                spill_partials<RT, NC>(red, acc, wave, lane);
            }
            __syncthreads();
            if (mine) dh += sum_partials<RT, NC>(red, p.nkw, row, u);
        }
        if (dead) dh = __builtin_nanf("");
        if (fin) {
            if (mine) p.dh[i] = dh;
            break;
        }
        if (row < BP) {
            float drp = 0.f, dzp = 0.f, dnr = 0.f;
            if (mine) {
                float* d = p.dgh + ((int64_t)t * B + grow) * 3 * H;
                const float dn = dh * (1.0f - z);
                const float dz = dh * (hp - nn);
                const float dnp = dn * (1.0f - nn * nn);
                dzp = dz * z * (1.0f - z);
                drp = dnp * hnv * r * (1.0f - r);
                dnr = dnp * r;
                g[j] = drp; g[H + j] = dzp; g[2 * H + j] = dnp;
                d[j] = drp; d[H + j] = dzp; d[2 * H + j] = dnr;
                carry = dh * z;
            }
            static_assert(UB % 4 == 0, "a workgroup's units are whole quads of the operand");
            const __amdgpu_buffer_rsrc_t out = (s & 1) ? rx1 : rx0;
            const bool lead = (u & 3) == 0 && row < p.rpg;      // image rows beyond the group's rows are never read
            const int q = j / 4;                         // the same for the 4 lanes of a quad (j0 and H are multiples of 4)
            publish_quad(out, (int)(((int64_t)q * BP + row) * 16), drp, lead);
            publish_quad(out, (int)(((int64_t)(H / 4 + q) * BP + row) * 16), dzp, lead);
            publish_quad(out, (int)(((int64_t)(2 * H / 4 + q) * BP + row) * 16), dnr, lead);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(cnt + 32 * (blockIdx.x % p.nsh), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// wp[wg][q][c][e] = W_hh[(c / UF) H + wg UF + c % UF][4q + e]      (one float4 per thread)
// The same launch also prepares the recurrence's start (all of it independent of the packing): hbuf0 = h0 (zeros for
// h0 == nullptr) with its quad-major image xq[grp][q][row][e] (rows_to_quads_kernel's layout, padded rows zero), and
// the persistent kernel's arrival counters cleared -- one launch instead of memset/copy + pack + rows_to_quads + memset.
__global__ __launch_bounds__(256) void pack_w_fwd_kernel(const float* __restrict__ w_hh, int H, float* __restrict__ wp,
                                                         const float* __restrict__ h0, float* __restrict__ hbuf0, int B, int BP,
                                                         int groups, float* __restrict__ xq, unsigned* __restrict__ sync) {
    if (xq != nullptr) {
        const int64_t per = (int64_t)(H / 4) * BP, nq = per * groups;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
            const int grp = (int)(i / per);
            const int64_t r = i - (int64_t)grp * per;
            const int row = grp * BP + (int)(r % BP);
            const int q = (int)(r / BP);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < B) {
                if (h0 != nullptr) { const float* s = h0 + (int64_t)row * H + 4 * q; v = make_float4(s[0], s[1], s[2], s[3]); }
                float* hb = hbuf0 + (int64_t)row * H + 4 * q;
                hb[0] = v.x; hb[1] = v.y; hb[2] = v.z; hb[3] = v.w;
            }
            float* d = xq + i * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    if (sync != nullptr && blockIdx.x == 0)
        for (int i = threadIdx.x; i < SYNC_WORDS; i += blockDim.x) sync[i] = 0u;
    constexpr int NC = 3 * UF;
    const int64_t n = (int64_t)(H / UF) * (H / 4) * NC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % NC);
        const int q = (int)((i / NC) % (H / 4));
        const int wg = (int)(i / NC / (H / 4));
        const float* s = w_hh + ((int64_t)(c / UF) * H + wg * UF + c % UF) * H + 4 * q;
        float* d = wp + i * 4;
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
    }
}

// wpT[wg][q][c][e] = W_hh[4q + e][wg UB + c],  q < 3H/4
// also: dh0 = dhT (zeros for dhT == nullptr) and the persistent kernel's arrival counters cleared
__global__ __launch_bounds__(256) void pack_w_bwd_kernel(const float* __restrict__ w_hh, int H, float* __restrict__ wpT,
                                                         const float* __restrict__ dhT, float* __restrict__ dh0, int64_t BH,
                                                         unsigned* __restrict__ sync) {
    if (dh0 != nullptr && dh0 != dhT)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < BH; i += (int64_t)gridDim.x * blockDim.x)
            dh0[i] = dhT != nullptr ? dhT[i] : 0.f;
    if (sync != nullptr && blockIdx.x == 0)
        for (int i = threadIdx.x; i < SYNC_WORDS; i += blockDim.x) sync[i] = 0u;
    const int64_t n = (int64_t)(H / UB) * (3 * H / 4) * UB;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % UB);
        const int q = (int)((i / UB) % (3 * H / 4));
        const int wg = (int)(i / UB / (3 * H / 4));
        const float* s = w_hh + (int64_t)(4 * q) * H + wg * UB + c;
        float* d = wpT + i * 4;
        d[0] = s[0]; d[1] = s[H]; d[2] = s[2 * (int64_t)H]; d[3] = s[3 * (int64_t)H];
    }
}

// xq[grp][q][row][e] = grp BP + row < B ? m[grp BP + row][4q + e] : 0      (K = row length of m, groups of BP padded rows)
__global__ __launch_bounds__(256) void rows_to_quads_kernel(const float* __restrict__ m, int B, int K, int BP, int groups,
                                                            float* __restrict__ xq) {
    const int64_t per = (int64_t)(K / 4) * BP, n = per * groups;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int grp = (int)(i / per);
        const int64_t r = i - (int64_t)grp * per;
        const int row = grp * BP + (int)(r % BP);
        const int q = (int)(r / BP);
        float* d = xq + i * 4;
        if (row < B) {
            const float* s = m + (int64_t)row * K + 4 * q;
            d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
        } else {
            d[0] = 0.f; d[1] = 0.f; d[2] = 0.f; d[3] = 0.f;
        }
    }
}

bool small_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("LAED_GRU_SMALL"); v = (e != nullptr && atoi(e) == 0) ? 0 : 1; }
    return v == 1;
}

// largest K split (waves) with K % (16 nkw) == 0
int k_waves(int K) {
    for (int w = NW; w > 1; w >>= 1)
        if (K % (16 * w) == 0) return w;
    return 1;
}

template <typename A, typename K1, typename K2, typename K3, typename K4>
void launch_rt(int B, dim3 grid, hipStream_t stream, const A& a, K1 k1, K2 k2, K3 k3, K4 k4) {
    if (B <= 16) hipLaunchKernelGGL(k1, grid, dim3(NTH), 0, stream, a);
    else if (B <= 32) hipLaunchKernelGGL(k2, grid, dim3(NTH), 0, stream, a);
    else if (B <= 48) hipLaunchKernelGGL(k3, grid, dim3(NTH), 0, stream, a);
    else hipLaunchKernelGGL(k4, grid, dim3(NTH), 0, stream, a);
}

}   // namespace

// CUs a persistent grid may fill: the device's multiprocessor count (queried once per device; 256 on MI355X, fewer on
// a partitioned or CU-masked device), capped by the caller's budget (cu_budget > 0).
int laed_persist_cu_limit(int cu_budget) {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 0;
        cus[dev] = n;
    }
    return (cu_budget > 0 && cu_budget < cus[dev]) ? cu_budget : cus[dev];
}

// Timed-out waits of the persistent kernels on the current device since the process started (synchronous read).
extern "C" int laed_gru_persist_timeouts(unsigned* out_host) {
    LAED_CHECK_ARG(out_host != nullptr, "gru_persist_timeouts: null pointer");
    LAED_HIP_TRY(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_persist_timeouts), sizeof(unsigned), 0, hipMemcpyDeviceToHost));
    return 0;
}

// Persistent launches issued by this process (host-side counters: [0] forward, [1] backward).  Tests and the engine
// read them to know WHICH recurrence path a call took (a CU budget, a sharer count or a grid that does not fit sends
// a call to the per-step launches without any other visible sign).
static std::atomic<unsigned> g_persist_launches[2];
extern "C" int laed_gru_persist_launches(unsigned* fwd_host, unsigned* bwd_host) {
    LAED_CHECK_ARG(fwd_host != nullptr && bwd_host != nullptr, "gru_persist_launches: null pointer");
    *fwd_host = g_persist_launches[0].load();
    *bwd_host = g_persist_launches[1].load();
    return 0;
}

// Row groups: up to 64 rows one group (padded to whole 16-row MFMA tiles); larger batches run as groups of 64 rows in
// blockIdx.y of the per-step kernels, each group re-reading the packed W_hh out of L2.
static int mid_rows() {      // rows per group of a mid-size batch (experiment hook LAED_GRU_MID_ROWS = 16 / 32 / 64)
    static const int v = []() { const char* e = laed_exp_hook("LAED_GRU_MID_ROWS"); const int r = e ? atoi(e) : 64;
                                return (r == 16 || r == 32 || r == 64) ? r : 64; }();
    return v;
}
static int small_groups(int B) { return B <= 64 ? 1 : (B + mid_rows() - 1) / mid_rows(); }

// H a multiple of 16 (quads, whole MFMA k-groups, whole unit slices) and
//   B <= 64: the reference's own batch sizes (persistent launches), or
//   64 < B <= 512 while the groups' W_hh reads stay under 32 MB per step (AE-ED / ST-ED utterance encoders: B x turns
//   = 300 sequences of H = 256): per-step launches of the packed core, ~2x faster there than the tiled step kernel,
//   which needs >= 256 of its 64-row x 32-unit tiles to fill the chip.  LAED_GRU_MID=0 keeps the tiled kernels.
bool laed_gru_small_ok(int B, int H) {
    if (!(small_enabled() && getenv("LAED_GRU_WM") == nullptr && B >= 1 && H >= 16 && (H % 16) == 0)) return false;
    if (B <= 64) return true;
    static const bool mid = []() { const char* e = laed_exp_hook("LAED_GRU_MID"); return !(e != nullptr && atoi(e) == 0); }();
    return mid && B <= 512 && (int64_t)small_groups(B) * 3 * H * H * 4 <= ((int64_t)32 << 20);
}

int laed_gru_small_bp(int B) { return B <= 64 ? (B + 15) / 16 * 16 : mid_rows(); }

// Row groups of the PERSISTENT launches (B <= 64): the batch is cut into groups of 16 or 32 rows while all their
// workgroups (grid_x per group, one per CU) still fit the CUs this call may count on -- every workgroup then pulls only
// its group's rows of the previous step's operand across the fabric, which is what a step of these kernels waits for
// (one group: 4.9 / 6.75 us per forward / backward step at B = 30, 8.3 / 11.1 us at B = 60, H = 512 -- the time follows the
// rows; grouped: DI-VAE step at B = 30 1.99 -> 1.78 ms, DI-VST at B = 60 3.82 -> 3.33 ms, losses bit-identical).
// LAED_GRU_PGROUPS=0 (hooks build) keeps one group.
struct PersistGroups { int rows, groups; };
// shards of a group's arrival counter: 8 (one per XCD's share of the workgroups); LAED_GRU_SHARDS=1..8 (hooks build) for A/B
static int persist_shards() {
    static const int n = []() { const char* e = laed_exp_hook("LAED_GRU_SHARDS"); const int v = e ? atoi(e) : MAX_SHARDS;
                                return v < 1 ? 1 : (v > MAX_SHARDS ? MAX_SHARDS : v); }();
    return n;
}
static PersistGroups persist_groups(int B, int grid_x, int cu_limit) {
    static const bool on = []() { const char* e = laed_exp_hook("LAED_GRU_PGROUPS"); return !(e != nullptr && atoi(e) == 0); }();
    const int bp = laed_gru_small_bp(B);
    if (on && B <= 64) {
        for (int rows = 16; rows < bp; rows += 16) {
            const int groups = (B + rows - 1) / rows;
            if (rows != 48 && (int64_t)groups * grid_x <= cu_limit) return {rows, groups};
        }
    }
    return {bp, 1};
}
// padded rows of the quad-major operands: the persistent grouping above may pad B up to 4 x 16 or 2 x 32 rows
static int small_ws_rows(int B) { return B <= 64 ? 64 : laed_gru_small_bp(B) * small_groups(B); }

int64_t laed_gru_small_fwd_ws(int B, int H) {
    return (int64_t)3 * H * H + (int64_t)2 * H * small_ws_rows(B) + SYNC_WORDS;
}
int64_t laed_gru_small_bwd_ws(int B, int H) {
    return (int64_t)3 * H * H + (int64_t)2 * 3 * H * small_ws_rows(B) + SYNC_WORDS;
}

// forward over all T steps; gates already holds x W_ih^T + b_ih, hbuf[0] = h0.  ws: laed_gru_small_fwd_ws floats, 16-byte aligned.
// h0: initial state [B][H] (nullptr = zeros); hbuf[0] is written here (h0 == hbuf is allowed).
int laed_gru_small_fwd(const float* w_hh, const float* b_hh, const float* h0, float* hbuf, float* gates, float* hn, int T, int B,
                       int H, float* ws, int cu_budget, hipStream_t stream) {
    int BP = laed_gru_small_bp(B), NG = small_groups(B);
    const int64_t BH = (int64_t)B * H;
    float* wp = ws;
    // default: one persistent launch with LDS-resident W_hh (4.9 vs 6.3 us per step + launch gaps at B = 30, H = 512);
    // LAED_GRU_PERSIST=0 selects the per-step launches below
    static const bool persist = []() { const char* e = getenv("LAED_GRU_PERSIST"); return !(e != nullptr && atoi(e) == 0); }();
    // every workgroup of the persistent grid must be resident at once, one per CU: take that path only when the grid
    // fits the CUs this call may count on (the device's CU count, or the caller's share of it when a second stream
    // may hold a persistent grid of its own at the same time)
    const int cu_limit = laed_persist_cu_limit(cu_budget);
    const bool persistent = persist && NG == 1 && H / UF <= cu_limit;
    if (persistent) {
        const PersistGroups pg = persist_groups(B, H / UF, cu_limit);
        BP = pg.rows;
        NG = pg.groups;
        if (NG > MAX_PGROUPS) return laed_fail(LAED_E_SIZE, "gru_small: %d row groups (at most %d)", NG, MAX_PGROUPS);
    }
    float* hx[2] = {ws + (int64_t)3 * H * H, ws + (int64_t)3 * H * H + (int64_t)H * BP * NG};
    // SYNC_WORDS words (flag + sharded arrival counters of <= 4 groups), cleared by the preparation launch
    unsigned* sync = persistent ? reinterpret_cast<unsigned*>(ws + (int64_t)3 * H * H + (int64_t)2 * H * BP * NG) : nullptr;
    // one preparation launch: pack W_hh, hbuf[0] = h0 (or zeros) + its quad-major image, clear the counters
    hipLaunchKernelGGL(pack_w_fwd_kernel, dim3(256), dim3(256), 0, stream, w_hh, H, wp, h0, hbuf, B, BP, NG, hx[0], sync);
    LAED_LAUNCH_CHECK("gru_small prepare_fwd");
    if (persistent) {
        PersistFwdArgs a;
        a.hx0 = hx[0]; a.hx1 = hx[1]; a.wp = wp; a.b_hh = b_hh; a.hbuf = hbuf; a.gates = gates; a.hn = hn; a.sync = sync;
        a.T = T; a.B = B; a.H = H; a.nkw = k_waves(H); a.nsh = persist_shards();
        size_t lds = ((size_t)H * WS4 + (size_t)NW * BP * 3 * UF) * sizeof(float);
        if (lds < 82 * 1024) lds = 82 * 1024;          // one workgroup per CU (160 KB LDS per CU)
        if (lds > 160 * 1024) return laed_fail(LAED_E_SIZE, "gru_small: LDS slab too large (H = %d)", H);
        auto go = [&](auto kern) -> int {
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e2 != hipSuccess) return laed_fail((int)e2, "gru_small: hipFuncSetAttribute failed");
            hipLaunchKernelGGL(kern, dim3(H / UF, NG), dim3(NTH), lds, stream, a);
            return 0;
        };
        int st;
        if (BP <= 16) st = go(gru_small_persist_fwd_kernel<1>);
        else if (BP <= 32) st = go(gru_small_persist_fwd_kernel<2>);
        else if (BP <= 48) st = go(gru_small_persist_fwd_kernel<3>);
        else st = go(gru_small_persist_fwd_kernel<4>);
        if (st != 0) return st;
        LAED_LAUNCH_CHECK("gru_small_persist_fwd");
        g_persist_launches[0].fetch_add(1);
        return 0;
    }
    for (int t = 0; t < T; ++t) {
        SmallFwdArgs a;
        a.hx_in = hx[t & 1]; a.wp = wp; a.b_hh = b_hh; a.hprev = hbuf + (int64_t)t * BH;
        a.gates = gates + (int64_t)t * B * 3 * H; a.hout = hbuf + (int64_t)(t + 1) * BH;
        a.hn = hn ? hn + (int64_t)t * BH : nullptr; a.hx_out = hx[(t + 1) & 1];
        a.B = B; a.H = H; a.nkw = k_waves(H);
        launch_rt(BP, dim3(H / UF, NG), stream, a, gru_small_fwd_kernel<1>, gru_small_fwd_kernel<2>, gru_small_fwd_kernel<3>,
                  gru_small_fwd_kernel<4>);
        LAED_LAUNCH_CHECK("gru_small_fwd");
    }
    return 0;
}

// backward recurrence over all T steps (gate gradients + dh chain); dh0 holds dh_T on entry and dh_0 on return.
// ws: laed_gru_small_bwd_ws floats, 16-byte aligned.
int laed_gru_small_bwd(const float* w_hh, const float* hbuf, const float* hn, const float* dout, const float* dhT, float* gates,
                       float* dgh, float* dh0, int T, int B, int H, float* ws, int cu_budget, hipStream_t stream) {
    const int BP = laed_gru_small_bp(B), NG = small_groups(B);
    const int64_t BH = (int64_t)B * H;
    float* wpT = ws;
    static const bool persist = []() { const char* e = getenv("LAED_GRU_PERSIST"); return !(e != nullptr && atoi(e) == 0); }();
    const int cu_limit = laed_persist_cu_limit(cu_budget);
    if (persist && NG == 1 && H / UB <= cu_limit) {
        PersistGroups pg = persist_groups(B, H / UB, cu_limit);
        int rpg = pg.rows;
        // half-filled tiles: groups of 8 batch rows in 16-row images while 4 such groups cover the batch and still fit the CUs --
        // a workgroup then pulls 8 rows x 3H floats per step instead of 16 (B = 30, H = 512: 49 KB instead of 98 KB)
        static const bool half_ok = []() { const char* e = laed_exp_hook("LAED_GRU_HALF_GROUPS"); return !(e != nullptr && atoi(e) == 0); }();
        if (half_ok && B <= 8 * MAX_PGROUPS && (int64_t)((B + 7) / 8) * (H / UB) <= cu_limit && pg.rows <= 16) {
            pg.rows = 16;
            pg.groups = (B + 7) / 8;
            rpg = 8;
        }
        const int PBP = pg.rows, PG = pg.groups;
        if (PG > MAX_PGROUPS) return laed_fail(LAED_E_SIZE, "gru_small: %d row groups (at most %d)", PG, MAX_PGROUPS);
        float* pdx[2] = {ws + (int64_t)3 * H * H, ws + (int64_t)3 * H * H + (int64_t)3 * H * PBP * PG};
        unsigned* sync = reinterpret_cast<unsigned*>(ws + (int64_t)3 * H * H + (int64_t)2 * 3 * H * PBP * PG);
        PersistBwdArgs a;
        a.dx0 = pdx[0]; a.dx1 = pdx[1]; a.wpT = wpT; a.dout = dout; a.hbuf = hbuf; a.hn = hn; a.gates = gates; a.dgh = dgh;
        a.dh = dh0; a.sync = sync; a.T = T; a.B = B; a.H = H; a.nkw = k_waves(3 * H); a.nsh = persist_shards(); a.rpg = rpg;
        size_t lds = ((size_t)3 * H * UB + (size_t)NW * PBP * UB) * sizeof(float);
        if (lds < 82 * 1024) lds = 82 * 1024;          // one workgroup per CU
        if (lds <= 160 * 1024) {
            // one preparation launch: pack W_hh^T, dh0 = dhT (or zeros), clear the arrival counters (a line per group + flag)
            hipLaunchKernelGGL(pack_w_bwd_kernel, dim3(256), dim3(256), 0, stream, w_hh, H, wpT, dhT, dh0, BH, sync);
            LAED_LAUNCH_CHECK("gru_small prepare_bwd");
            auto go = [&](auto kern) -> int {
                hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e2 != hipSuccess) return laed_fail((int)e2, "gru_small: hipFuncSetAttribute failed");
                hipLaunchKernelGGL(kern, dim3(H / UB, PG), dim3(NTH), lds, stream, a);
                return 0;
            };
            int st;
            if (PBP <= 16) st = go(gru_small_persist_bwd_kernel<1>);
            else if (PBP <= 32) st = go(gru_small_persist_bwd_kernel<2>);
            else if (PBP <= 48) st = go(gru_small_persist_bwd_kernel<3>);
            else st = go(gru_small_persist_bwd_kernel<4>);
            if (st != 0) return st;
            LAED_LAUNCH_CHECK("gru_small_persist_bwd");
            g_persist_launches[1].fetch_add(1);
            return 0;
        }
    }
    hipLaunchKernelGGL(pack_w_bwd_kernel, dim3(256), dim3(256), 0, stream, w_hh, H, wpT, dhT, dh0, BH, (unsigned*)nullptr);
    LAED_LAUNCH_CHECK("gru_small prepare_bwd");
    float* dx[2] = {ws + (int64_t)3 * H * H, ws + (int64_t)3 * H * H + (int64_t)3 * H * BP * NG};
    for (int t = T - 1; t >= -1; --t) {
        SmallBwdArgs a;
        const bool fin = t < 0;
        a.dx_in = t == T - 1 ? nullptr : dx[(t + 1) & 1];
        a.wpT = wpT;
        a.dout = (!fin && dout) ? dout + (int64_t)t * BH : nullptr;
        a.hprev = fin ? nullptr : hbuf + (int64_t)t * BH;
        a.hn = fin ? nullptr : hn + (int64_t)t * BH;
        a.gates = fin ? nullptr : gates + (int64_t)t * B * 3 * H;
        a.dgh = fin ? nullptr : dgh + (int64_t)t * B * 3 * H;
        a.dx_out = fin ? nullptr : dx[t & 1];
        a.dh = dh0; a.B = B; a.H = H; a.nkw = k_waves(3 * H); a.final_step = fin ? 1 : 0;
        launch_rt(BP, dim3(H / UB, NG), stream, a, gru_small_bwd_kernel<1>, gru_small_bwd_kernel<2>, gru_small_bwd_kernel<3>,
                  gru_small_bwd_kernel<4>);
        LAED_LAUNCH_CHECK("gru_small_bwd");
    }
    return 0;
}
