// GRU (1 layer, uni-directional, gates r,z,n), time-major, fp32.
//
// forward : gates = x W_ih^T + b_ih for all T at once (sgemm), then per timestep ONE fused kernel:
//             gh = h_{t-1} W_hh^T (fp32 MFMA)  ->  gate epilogue in registers  ->  h_t, (r,z,n), hn
//           The block tile is (32*WM batch rows) x (32 hidden units x 3 gates): a wave's three 32x32 MFMA
//           accumulators are the r, z and n pre-activations of the SAME (row, unit) pairs, lane for lane, so the
//           gate math needs no cross-lane traffic.  WK waves may split K (small batches) and are summed through LDS.
// backward: per timestep (reverse): elementwise gate-gradient kernel, then dh_{t-1} += dgh_t W_hh (sgemm);
//           after the loop three big GEMMs: dW_hh = dgh^T hprev, dW_ih = dgi^T x, dx = dgi W_ih
//           (bias gradients fall out of the transposed-A GEMMs as column sums).
// Time-major storage makes h_{t-1} for all (t,b) one plain matrix: hbuf[0 : T*B] (hbuf[0] = h0).
#include "laed_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {


// uniform base pointer + 32-bit BYTE offset per lane (the form the hardware addresses with a scalar base register)
__device__ __forceinline__ float ldb(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void stb(float* base, unsigned byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
// the same for data that is touched ONCE per pass -- the input projection's gi rows on their way in, the gates and hn kept for the
// backward pass on their way out: streaming (nt) accesses, so that W_hh and h_{t-1}, which every block re-reads, keep the L2
// (214 -> 211 us per step at B = 4096, alternated on one box; bit-identical)
__device__ __forceinline__ float ldb_s(const float* base, unsigned byte_off) {
    return __builtin_nontemporal_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off));
}
__device__ __forceinline__ void stb_s(float* base, unsigned byte_off, float v) {
    __builtin_nontemporal_store(v, reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off));
}

struct GruStepArgs {
    const float* hprev;   // [B][H]
    const float* w_hh;    // [3H][H]
    const float* b_hh;    // [3H]
    float* gates;         // [B][3H]  in: gi (+b_ih), out: r,z,n
    float* hout;          // [B][H]
    float* hn;            // [B][H] or NULL
    int B, H;
    int units_in_x;       // grid = (unit tiles, row tiles) instead of (row tiles, unit tiles)
    int narrow;           // B * 3H < 2^30: the gate epilogue may address with 32-bit element offsets
};

// WM row tiles x WK K-splits = NT/64 waves (4 waves with K tiles of 32, or 8 waves with K tiles of 64: in both a load
// pass covers 32 rows, i.e. the h tile in WM passes and one gate block of the W tile per pass)
// WU unit tiles per block (WU = 2: 64 hidden units x 3 gates = 192 columns): the W_hh tile is then shared by more MFMAs
// per staged byte -- (BR + 96 WU) * 128 bytes per K tile for 3 * WM * WU * 16 MFMAs: 213 B/MFMA for 64 x 96,
// 107 B/MFMA for 128 x 192 (the 128 x 128 GEMM tile: 128) -- which is what the GEMM ablations say the loss scales with.
template <int WM, int WK, bool VEC, int NT = 256, int BK = 32, int WU = 1>
__global__ __launch_bounds__(NT) void gru_step_fwd_kernel(GruStepArgs p) {
    static_assert(WM * WU * WK * 64 == NT && NT / (BK / 4) == 32 * WU, "geometry");
    constexpr int BR = 32 * WM;        // batch rows per block
    constexpr int UB = 32 * WU;        // hidden units per block
    constexpr int SH = BR + 1;         // K-contiguous operands: transposing b32 writes, pad 1
    constexpr int SW = 3 * UB + 1;
    constexpr int RPP = NT / (BK / 4); // tile rows covered by one load pass (= UB: one gate block of the W tile)
    constexpr int NLH = BR * BK / 4 / NT > 0 ? BR * BK / 4 / NT : 1;   // float4 per thread (h tile)
    constexpr int NLW = 3 * UB * BK / 4 / NT;                            // = 3
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sH = smem;                  // [2][BK][SH]
    float* sW = smem + 2 * BK * SH;    // [2][BK][SW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wu = (wave / WM) % WU, wq = wave / (WM * WU);
    // blocks are dealt round-robin over the 8 XCDs by their linear id: with the unit tile in blockIdx.x the blocks of one
    // XCD share 1/8 of the W_hh slices (L2-resident) and differ in their batch rows; with the row tile in blockIdx.x
    // every XCD streams ALL of W_hh each step (12.6 MB at H = 1024: more than its 4 MB L2)
    const int m0 = (p.units_in_x ? blockIdx.y : blockIdx.x) * BR, j0 = (p.units_in_x ? blockIdx.x : blockIdx.y) * UB;
    const int B = p.B, H = p.H;

    f32x16 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;

    // Gate epilogue: with a 2-way K split the two waves share it -- wave wq applies the gates to rows e in
    // [wq * EPW, (wq + 1) * EPW) of its 32x32 tiles after the partial sums have met in LDS (B = 4096: -2.7 %); with a
    // 4-way split (small batches, one block per CU) the wq == 0 wave keeps all 16 rows (sharing measured +10 % there).
    // Its operands (gi_r, gi_z, gi_n, h_prev of this lane's EPW (row, unit) pairs) are fetched NOW and consumed after the
    // K loop: loaded there they were dependent round trips (the stores of one row alias the loads of the next).
    constexpr int EW = (WK == 2) ? 2 : 1;      // waves per row tile that apply gates
    constexpr int EPW = 16 / EW;
    float pre[EPW][4];
    float bhr = 0.f, bhz = 0.f, bhn = 0.f;
    // interior blocks (whole tile inside [B] x [H], 16-byte aligned operands): no guards anywhere, 32-bit element offsets
    // from the uniform base pointers (one VALU add per access instead of a 64-bit add pair)
    const bool interior = VEC && (m0 + BR <= B) && (j0 + UB <= H);
    const bool lean = interior && p.narrow;
    const int jj = j0 + wu * 32 + (tid & 31);
    // lane's first row: accumulator element e of a 32x32 tile is row (e & 3) + 8 (e >> 2) + 4 (lane >> 5); with two
    // epilogue waves wave wq owns e in [8 wq, 8 wq + 8), i.e. 16 wq rows further down
    const int row0 = m0 + wm * 32 + 4 * ((tid & 63) >> 5) + (EW == 2 ? 16 * wq : 0);
    // lean blocks address the gate operands as uniform base + 32-bit byte offset
    const unsigned og = 4u * ((unsigned)row0 * 3u * (unsigned)H + (unsigned)jj), oh = 4u * ((unsigned)row0 * (unsigned)H + (unsigned)jj);
    const unsigned hb = 4u * (unsigned)H;              // bytes per row of [.][H]
    auto pre_load = [&](int i) {                       // i = 4 ee + k, a compile-time constant wherever it is called
        const int ee = i >> 2, k = i & 3;
        const unsigned dr = (ee & 3) + 8 * (ee >> 2);
#if defined(GRU_EXP) && (GRU_EXP & 16)
        pre[ee][k] = bhr;
#else
        pre[ee][k] = k < 3 ? ldb_s(p.gates, og + dr * 3u * hb + k * hb) : ldb(p.hprev, oh + dr * hb);
#endif
    };
    // The 4 EPW gate operands per lane are 67 MB per step at B = 4096 -- a third of what the step moves.  Issued first
    // thing (as they were until round 3) they sit in front of the first K tiles in every wave's in-order load queue and
    // the K loop starts late in each round of blocks (ablation: 224 -> 212 us per step without them); blocks that run the
    // scheduled loop therefore issue them BEHIND the loads of K tiles 0 and 1, so that tile 0 multiplies while they land.
    const int nfull = H / BK;
    const bool late = lean && nfull >= 2;
    if (wq < EW) {
        if (lean) {
            bhr = p.b_hh[jj]; bhz = p.b_hh[H + jj]; bhn = p.b_hh[2 * H + jj];
            if (!late) {
#pragma unroll
                for (int i = 0; i < 4 * EPW; ++i) pre_load(i);
            }
        } else if (jj < H) {
            bhr = p.b_hh[jj]; bhz = p.b_hh[H + jj]; bhn = p.b_hh[2 * H + jj];
#pragma unroll
            for (int ee = 0; ee < EPW; ++ee) {
                const int row = row0 + (ee & 3) + 8 * (ee >> 2);
                if (row < B) {
                    const float* g = p.gates + (int64_t)row * 3 * H;
                    pre[ee][0] = g[jj]; pre[ee][1] = g[H + jj]; pre[ee][2] = g[2 * H + jj];
                    pre[ee][3] = p.hprev[(int64_t)row * H + jj];
                }
            }
        }
    }

    float4 rh[NLH], rw[NLW];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NLH; ++i) {
            int idx = tid + i * NT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < BR * BK / 4) {
                int row = idx / (BK / 4), kq = idx % (BK / 4);
                int gr = m0 + row, gk = k0 + 4 * kq;
                if (gr < B && gk < H) {
                    const float* src = p.hprev + (int64_t)gr * H + gk;
                    if (VEC && gk + 3 < H) v = *reinterpret_cast<const float4*>(src);
                    else { v.x = src[0]; if (gk + 1 < H) v.y = src[1]; if (gk + 2 < H) v.z = src[2]; if (gk + 3 < H) v.w = src[3]; }
                }
            }
            rh[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NLW; ++i) {
            int idx = tid + i * NT;
            int row = idx / (BK / 4), kq = idx % (BK / 4);   // row in [0, 3 UB): gate = row / UB, unit = row % UB
            int unit = j0 + (row % UB), gk = k0 + 4 * kq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (unit < H && gk < H) {
                const float* src = p.w_hh + ((int64_t)(row / UB) * H + unit) * H + gk;
                if (VEC && gk + 3 < H) v = *reinterpret_cast<const float4*>(src);
                else { v.x = src[0]; if (gk + 1 < H) v.y = src[1]; if (gk + 2 < H) v.z = src[2]; if (gk + 3 < H) v.w = src[3]; }
            }
            rw[i] = v;
        }
    };
    auto store = [&](int buf) {
        float* dh = sH + buf * BK * SH;
        float* dw = sW + buf * BK * SW;
#pragma unroll
        for (int i = 0; i < NLH; ++i) {
            int idx = tid + i * NT;
            if (idx < BR * BK / 4) {
                int row = idx / (BK / 4), kq = idx % (BK / 4);
                float* d = dh + (4 * kq) * SH + row;
                d[0] = rh[i].x; d[SH] = rh[i].y; d[2 * SH] = rh[i].z; d[3 * SH] = rh[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < NLW; ++i) {
            int idx = tid + i * NT;
            int row = idx / (BK / 4), kq = idx % (BK / 4);
            float* d = dw + (4 * kq) * SW + row;
            d[0] = rw[i].x; d[SW] = rw[i].y; d[2 * SW] = rw[i].z; d[3 * SW] = rw[i].w;
        }
    };

    const int nkt = (H + BK - 1) / BK;
    const int lrow = lane & 31, lk = lane >> 5;
    // interior blocks: scheduled loop (see sgemm.hip) -- loads / LDS reads / LDS stores sit in the MFMA issue gaps
    const bool fast = interior && nfull > 0;
    constexpr int KS = BK / 2 / WK;            // k pairs per wave per K tile
    constexpr int Q = 3 * KS;                  // MFMA issue gaps per K tile
    constexpr int NL = NLH + NLW;
    const float* gh = p.hprev + (int64_t)(m0 + tid / (BK / 4)) * H + 4 * (tid % (BK / 4));
    const float* gw = p.w_hh + (int64_t)(j0 + tid / (BK / 4)) * H + 4 * (tid % (BK / 4));
    const int64_t h_ist = (int64_t)RPP * H, w_ist = (int64_t)H * H;   // RPP rows per pass; one gate block per pass
    const int s_off_h = (4 * (tid % (BK / 4))) * SH + tid / (BK / 4);
    const int s_off_w = (4 * (tid % (BK / 4))) * SW + tid / (BK / 4);
    auto load_unit = [&](int u) {
        if (u < NLH) rh[u] = *reinterpret_cast<const float4*>(gh + u * h_ist);
        else rw[u - NLH] = *reinterpret_cast<const float4*>(gw + (u - NLH) * w_ist);
    };
    auto store_unit = [&](int u, int buf) {
        if (u < NLH) {
            float* d = sH + buf * BK * SH + s_off_h + u * RPP;
            d[0] = rh[u].x; d[SH] = rh[u].y; d[2 * SH] = rh[u].z; d[3 * SH] = rh[u].w;
        } else {
            const int v = u - NLH;
            float* d = sW + buf * BK * SW + s_off_w + v * RPP;
            d[0] = rw[v].x; d[SW] = rw[v].y; d[2 * SW] = rw[v].z; d[3 * SW] = rw[v].w;
        }
    };
    if (fast) {
#pragma unroll
        for (int u = 0; u < NL; ++u) load_unit(u);
#pragma unroll
        for (int u = 0; u < NL; ++u) store_unit(u, 0);
    } else {
        load(0);
        store(0);
    }
    __syncthreads();
    int cur = 0, kt = 0;
    if (fast && nfull >= 2) {
        // prefetch distance = one whole K tile (as in sgemm.hip): while tile kt multiplies out of LDS[cur], the registers
        // (tile kt+1, fetched during tile kt-1) go to LDS[cur^1] and are refilled with tile kt+2 unit by unit, so a
        // load has a full tile of MFMA time to land -- with one block per CU (B = 512) the half-tile distance used
        // before stalled every K tile on vmcnt.  The last pass re-reads tile kt+1 (cache hit, unused) instead of branching.
        gh += BK;
        gw += BK;
#pragma unroll
        for (int u = 0; u < NL; ++u) load_unit(u);
        if (late && wq < EW) {
#pragma unroll
            for (int i = 0; i < 4 * EPW; ++i) pre_load(i);
        }
        for (; kt + 1 < nfull; ++kt) {
            const float* h_s = sH + cur * BK * SH + wm * 32 + lrow + (2 * wq + lk) * SH;
            const float* w_s = sW + cur * BK * SW + wu * 32 + lrow + (2 * wq + lk) * SW;
            if (kt + 2 < nfull) { gh += BK; gw += BK; }
            float af[2], bf[2][3];
            af[0] = h_s[0];
            bf[0][0] = w_s[0]; bf[0][1] = w_s[UB]; bf[0][2] = w_s[2 * UB];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const int q_ = ks * 3 + g;
                    laed_mfma_32x32x2(acc[g], af[ks & 1], bf[ks & 1][g]);          // accumulators in AccVGPRs (laed_common.h)
                    if (g == 0 && ks + 1 < KS) {
#if !defined(GRU_EXP) || !(GRU_EXP & 4)
                        const int o = 2 * (ks + 1) * WK;
                        af[(ks + 1) & 1] = h_s[o * SH];
                        bf[(ks + 1) & 1][0] = w_s[o * SW]; bf[(ks + 1) & 1][1] = w_s[o * SW + UB]; bf[(ks + 1) & 1][2] = w_s[o * SW + 2 * UB];
#else
                        af[(ks + 1) & 1] = af[ks & 1];
                        bf[(ks + 1) & 1][0] = bf[ks & 1][0]; bf[(ks + 1) & 1][1] = bf[ks & 1][1]; bf[(ks + 1) & 1][2] = bf[ks & 1][2];
#endif
                    }
#if !defined(GRU_EXP) || !(GRU_EXP & 2)
#pragma unroll
                    for (int u = 0; u < NL; ++u) {
                        const int gs = Q / 8 + (u * (3 * Q / 4)) / NL;      // store gap of unit u
                        if (q_ == gs) store_unit(u, cur ^ 1);               // tile kt+1 -> LDS
                        if (q_ == gs + 1) load_unit(u);                     // tile kt+2 -> registers
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#if !defined(GRU_EXP) || !(GRU_EXP & 1)
            __syncthreads();
#endif
            cur ^= 1;
        }
    }
    for (; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) load((kt + 1) * BK);
        const float* h_s = sH + cur * BK * SH + wm * 32 + lrow;
        const float* w_s = sW + cur * BK * SW + wu * 32 + lrow;
#pragma unroll
        for (int ks = 0; ks < BK / 2 / WK; ++ks) {
            const int krow = 2 * (wq + ks * WK) + lk;
            float a = h_s[krow * SH];
            float b0 = w_s[krow * SW], b1 = w_s[krow * SW + UB], b2 = w_s[krow * SW + 2 * UB];
            laed_mfma_32x32x2(acc[0], a, b0);
            laed_mfma_32x32x2(acc[1], a, b1);
            laed_mfma_32x32x2(acc[2], a, b2);
        }
        laed_mfma_drain(acc);                       // inside the body, every pass: see the generic loop of sgemm.hip
        if (more) store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    laed_mfma_drain(acc);
    if (WK > 1) {   // every wave hands the rows it does not own to their owners (fixed summation order: wq ascending)
        float* red = smem;   // all waves are past the last barrier: tiles are dead
        float* d = red + ((wq * WM * WU + wu * WM + wm) * 48) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (e / EPW != wq) d[(g * 16 + e) * 64] = acc[g][e];
        __syncthreads();
        if (wq >= EW) return;
#pragma unroll
        for (int s = 0; s < WK; ++s) {
            if (s == wq) continue;
            const float* sp = red + ((s * WM * WU + wu * WM + wm) * 48) * 64 + lane;
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (e / EPW == wq) acc[g][e] += sp[(g * 16 + e) * 64];
        }
    }

    // gate epilogue on the wave's EPW accumulator rows starting at element E0 (compile-time: the accumulators are registers)
    auto gates_of = [&](auto e0c, auto keepc) {
        constexpr int E0 = decltype(e0c)::value;
        constexpr bool KEEP = decltype(keepc)::value;      // training: r, z, n and gh_n are kept for backward
        if (lean) {
#if defined(GRU_EXP) && (GRU_EXP & 8)
            float sum = 0.f;
#pragma unroll
            for (int ee = 0; ee < EPW; ++ee) sum += acc[0][E0 + ee] + acc[1][E0 + ee] + acc[2][E0 + ee] + pre[ee][0] + pre[ee][1] + pre[ee][2] + pre[ee][3];
            if (sum == 1234.5f) stb(p.hout, oh, sum + og + hb);
            return;
#endif
#pragma unroll
            for (int ee = 0; ee < EPW; ++ee) {
                const int e = E0 + ee;
                const unsigned dr = (ee & 3) + 8 * (ee >> 2);
                const float ghn = acc[2][e] + bhn;
                const float r = sigmoidf_acc(pre[ee][0] + acc[0][e] + bhr);
                const float z = sigmoidf_acc(pre[ee][1] + acc[1][e] + bhz);
                const float n = tanhf_gate(pre[ee][2] + r * ghn);
                const unsigned a = og + dr * 3u * hb, b = oh + dr * hb;
                stb(p.hout, b, (1.0f - z) * n + z * pre[ee][3]);
                if (KEEP) {
                    stb_s(p.gates, a, r); stb_s(p.gates, a + hb, z); stb_s(p.gates, a + 2u * hb, n);
                    stb_s(p.hn, b, ghn);
                }
            }
            return;
        }
        if (jj >= H) return;
#pragma unroll
        for (int ee = 0; ee < EPW; ++ee) {
            const int e = E0 + ee;
            const int row = row0 + (ee & 3) + 8 * (ee >> 2);
            if (row < B) {
                float* g = p.gates + (int64_t)row * 3 * H;
                const float ghn = acc[2][e] + bhn;
                const float r = sigmoidf_acc(pre[ee][0] + acc[0][e] + bhr);
                const float z = sigmoidf_acc(pre[ee][1] + acc[1][e] + bhz);
                const float n = tanhf_gate(pre[ee][2] + r * ghn);
                p.hout[(int64_t)row * H + jj] = (1.0f - z) * n + z * pre[ee][3];
                if (KEEP) {
                    g[jj] = r; g[H + jj] = z; g[2 * H + jj] = n;
                    p.hn[(int64_t)row * H + jj] = ghn;
                }
            }
        }
    };
    auto gates_at = [&](auto e0c) {
        if (p.hn != nullptr) gates_of(e0c, std::true_type());
        else gates_of(e0c, std::false_type());
    };
    if (EW == 2 && wq == 1) gates_at(std::integral_constant<int, 16 - EPW>());   // = 8 when EW == 2 (never taken otherwise)
    else gates_at(std::integral_constant<int, 0>());
}

template <int WM, int WK, int NT = 256, int BKT = 32, int WU = 1>
int launch_step(const GruStepArgs& a, bool vec, hipStream_t stream) {
    GruStepArgs b = a;
    const unsigned mt = (a.B + 32 * WM - 1) / (32 * WM), jt = (a.H + 32 * WU - 1) / (32 * WU);
    constexpr int TILE = BKT * (32 * WM + 1 + 96 * WU + 1);
    constexpr int RED = (WK > 1) ? WK * WM * WU * 64 * 48 : 0;
    constexpr size_t lds = sizeof(float) * ((2 * TILE > RED) ? 2 * TILE : RED);
    auto kern_t = gru_step_fwd_kernel<WM, WK, true, NT, BKT, WU>;
    auto kern_f = gru_step_fwd_kernel<WM, WK, false, NT, BKT, WU>;
    static uint64_t attr_set = 0;       // per device (and per instantiation: the static lives in the template)
    if (lds > 48 * 1024 && laed_first_use_on_device(&attr_set)) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern_t), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern_f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e1 != hipSuccess || e2 != hipSuccess) return laed_fail((int)(e1 != hipSuccess ? e1 : e2), "gru_step: hipFuncSetAttribute failed");
    }
    static const int force_x = []() { const char* e = laed_exp_hook("LAED_GRU_UNITS_IN_X"); return e ? atoi(e) : -1; }();
    b.units_in_x = force_x >= 0 ? force_x : 1;
    b.narrow = ((int64_t)a.B * 3 * a.H < ((int64_t)1 << 30)) ? 1 : 0;
    if (mt > 65535) b.units_in_x = 1;
    dim3 grid = b.units_in_x ? dim3(jt, mt) : dim3(mt, jt);
    if (vec) hipLaunchKernelGGL(kern_t, grid, dim3(NT), lds, stream, b);
    else hipLaunchKernelGGL(kern_f, grid, dim3(NT), lds, stream, b);
    LAED_LAUNCH_CHECK("gru_step_fwd");
    return 0;
}

int gru_step(const GruStepArgs& a, hipStream_t stream) {
    const bool vec = aligned16(a.hprev) && aligned16(a.w_hh) && (a.H % 4 == 0);
    const int64_t jt = (a.H + 31) / 32;
    const char* env_wm = getenv("LAED_GRU_WM");   // test hook (4, 2 or 1): forces the block geometry
    const int forced_wm = env_wm ? atoi(env_wm) : 0;
    if (forced_wm == 4) return launch_step<4, 1>(a, vec, stream);
    if (forced_wm == 2) return launch_step<2, 2>(a, vec, stream);
    if (forced_wm == 1) return launch_step<1, 4>(a, vec, stream);
    if (forced_wm == 8) return launch_step<2, 4, 512, 64>(a, vec, stream);
    // two-unit-tile geometries (half the staged bytes per MFMA), measured in round 2 and not selected: the step kernel's
    // loss sits outside its 32-tile K loop (gate operands / results, block start-up), not in the staging -- us per step
    // at H = 1024, B = 4096 incl. the input GEMM's share: 64x96 299.6, 128x96 293.1, 128x192 298.5, 64x192 (K split) 306.9
    if (forced_wm == 42) return launch_step<4, 1, 512, 32, 2>(a, vec, stream);     // 128 rows x 64 units, one block per CU
    if (forced_wm == 22) return launch_step<2, 2, 512, 32, 2>(a, vec, stream);     //  64 rows x 64 units, K split 2
    // measured at H = 1024 (tools/gru_geom.py, us/step incl. the input GEMM share, B = 256 / 512 / 2048 / 4096):
    //   128-row tiles 92 / 98 / 173 / 333,  64-row tiles 51 / 56 / 161 / 309,  32-row tiles (K split 4 ways) 39 / 70 / 256 / -
    // 64-row tiles (41 KB of LDS: three blocks = 12 waves per CU) win from one block per CU upwards; smaller batches
    // split K over all four waves of a 32-row tile instead.
    // from two full rounds of 128-row tiles upwards (2 blocks per CU: 512 slots) those win by 2-3 % (round 2, same
    // script: B = 2048 148.5 vs 153.3, B = 4096 293.1 vs 299.6): 1024 blocks are exactly two rounds, where the 2048
    // 64-row blocks run as 2.67 rounds of 3 per CU
    if (cdiv64(a.B, 128) * jt >= 512) return launch_step<4, 1>(a, vec, stream);
    // Below that the launch is one to three rounds of blocks and the rounds decide (round 3, tools/gru_exp.py with
    // GRU_EXP_B / GRU_EXP_H over B = 128..4096 at H = 256 / 512 / 1024, us per step):
    //   * at most 256 blocks of 64 rows (one per CU): the 32-row tiles with K split 4 ways run as one or two rounds and beat
    //     the 64-row / 2-way tiles (H = 1024, B = 512: 36.7 vs 37.9; H = 512, B = 1024: 23.2 vs 24.3; H = 256, B = 2048: 16.7
    //     vs 17.7); with a long K (H >= 1024) the 8-wave block -- 64 rows, K split 4 ways, 64-deep K tiles, two waves per
    //     SIMD -- is better still once the 32-row tiles need a second round (B = 384 / 512: 34.1 / 34.5);
    //   * between 256 and 512 blocks of 64 rows, three rounds of 32-row tiles beat the 64-row tiles, which put two blocks on
    //     some CUs and one on the rest (H = 1024, B = 640 / 768: 53.3 / 53.9 vs 61.4 / 62.1; H = 512, B = 1536: 33.7 vs 37.6);
    //   * from 512 blocks of 64 rows (two per CU) those tiles win again (H = 1024, B = 1024: 63.6 vs 68.3).
    const int64_t n64 = cdiv64(a.B, 64) * jt, n32 = cdiv64(a.B, 32) * jt;
    if (n64 <= 256) {
        if (a.H >= 1024 && n32 > 256) return launch_step<2, 4, 512, 64>(a, vec, stream);
        return launch_step<1, 4>(a, vec, stream);
    }
    if (n64 < 512 && n32 <= 768) return launch_step<1, 4>(a, vec, stream);
    return launch_step<2, 2>(a, vec, stream);
}

// ---- backward, one timestep: gate gradients (elementwise over [B][H])
struct GruBwdStepArgs {
    const float* dout;    // [B][H] slice of dOut at t, or NULL
    const float* hprev;   // [B][H]
    const float* hn;      // [B][H]
    float* gates;         // [B][3H] in: r,z,n ; out: d(gi)
    float* dgh;           // [B][3H] out
    float* dh;            // [B][H] in: carried dh_t (without dout) ; out: dh_t * z
    const float* parts;   // [nparts][B][H] split-K partials of the previous step's dgh W_hh product (or NULL)
    int nparts;
    int B, H;
};

__global__ __launch_bounds__(256) void gru_bwd_gates_kernel(GruBwdStepArgs p) {
    const int64_t n = (int64_t)p.B * p.H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / p.H;
        const int j = (int)(i - b * p.H);
        float dh = p.dh[i];
        for (int s_ = 0; s_ < p.nparts; ++s_) dh += p.parts[(int64_t)s_ * n + i];   // fixed order: reproducible
        if (p.dout != nullptr) dh += p.dout[i];
        float* g = p.gates + b * 3 * p.H;
        float* d = p.dgh + b * 3 * p.H;
        // r, z, n and hn were kept by the forward pass long ago and are read once; d(gi) written below is read by the post-loop
        // products only: streaming accesses (42.0 -> 40.9 us per step at B = 4096, H = 1024).  d(gh) feeds the very next GEMM.
        const float r = __builtin_nontemporal_load(g + j), z = __builtin_nontemporal_load(g + p.H + j), nn = __builtin_nontemporal_load(g + 2 * p.H + j);
        const float hn = __builtin_nontemporal_load(p.hn + i), hp = p.hprev[i];
        const float dn = dh * (1.0f - z);
        const float dz = dh * (hp - nn);
        const float dnp = dn * (1.0f - nn * nn);
        const float dzp = dz * z * (1.0f - z);
        const float drp = dnp * hn * r * (1.0f - r);
        __builtin_nontemporal_store(drp, g + j); __builtin_nontemporal_store(dzp, g + p.H + j); __builtin_nontemporal_store(dnp, g + 2 * p.H + j);
        d[j] = drp; d[p.H + j] = dzp; d[2 * p.H + j] = dnp * r;
        p.dh[i] = dh * z;
    }
}

__global__ __launch_bounds__(256) void add_parts_kernel(float* __restrict__ y, const float* __restrict__ parts, int nparts, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = y[i];
        for (int s_ = 0; s_ < nparts; ++s_) v += parts[(int64_t)s_ * n + i];
        y[i] = v;
    }
}

}   // namespace

extern "C" int64_t laed_gru_fwd_workspace(int T, int B, int E, int H) {
    (void)T; (void)E;
    return laed_gru_small_ok(B, H) ? laed_gru_small_fwd_ws(B, H) : 0;       // packed W_hh + quad-major h ping-pong
}

extern "C" int64_t laed_gru_bwd_workspace(int T, int B, int E, int H) {
    const int64_t TBp = ((int64_t)T * B + 31) / 32 * 32;
    int64_t n = (int64_t)16 * B * H + (int64_t)8 * 3 * H * ((H > E ? H : E) + 1);          // split-K scratch
    if (laed_gru_small_ok(B, H) && n < laed_gru_small_bwd_ws(B, H)) n = laed_gru_small_bwd_ws(B, H);
    if (laed_get_gemm_mode() == 1)
        n += (int64_t)3 * H * H + (int64_t)3 * H * E + (int64_t)3 * H * TBp + (int64_t)H * TBp;   // transposed operands
    return n;
}

int laed_gru_step_impl(const float* hprev, const float* w_hh, const float* b_hh, float* gates, float* hout, float* hn,
                       int B, int H, hipStream_t stream) {
    GruStepArgs a;
    a.hprev = hprev; a.w_hh = w_hh; a.b_hh = b_hh; a.gates = gates; a.hout = hout; a.hn = hn; a.B = B; a.H = H;
    return gru_step(a, stream);
}

extern "C" int laed_gru_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                            const float* h0, int T, int B, int E, int H,
                            float* hbuf, float* gates, float* hn, float* ws, int64_t ws_floats, void* stream_) {
    return laed_gru_fwd_cus(x, w_ih, w_hh, b_ih, b_hh, h0, T, B, E, H, hbuf, gates, hn, ws, ws_floats, 0, stream_);
}

extern "C" int laed_gru_fwd_cus(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                                const float* h0, int T, int B, int E, int H,
                                float* hbuf, float* gates, float* hn, float* ws, int64_t ws_floats, int cu_budget,
                                void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LAED_CHECK_ARG(T > 0 && B > 0 && E > 0 && H > 0, "gru_fwd: non-positive dimension");
    LAED_CHECK_ARG(x && w_ih && w_hh && b_ih && b_hh && hbuf && gates, "gru_fwd: null pointer");
    const int64_t BH = (int64_t)B * H;
    LAED_CHECK_ARG((int64_t)T * B < 0x7fffffff, "gru_fwd: T*B too large");
    // small batches: packed-operand step kernels (gru_small.hip); their preparation launch also writes hbuf[0]
    const bool small = ws != nullptr && aligned16(ws) && laed_gru_small_ok(B, H) && ws_floats >= laed_gru_small_fwd_ws(B, H);
    if (small) {
    } else if (h0 != nullptr) {
        if (h0 != hbuf) {
            hipError_t e = hipMemcpyAsync(hbuf, h0, BH * sizeof(float), hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) return laed_fail((int)e, "gru_fwd: h0 copy failed");
        }
    } else {
        hipError_t e = hipMemsetAsync(hbuf, 0, BH * sizeof(float), stream);
        if (e != hipSuccess) return laed_fail((int)e, "gru_fwd: h0 memset failed");
    }
    // hoisted input projection for every timestep: gates[T*B][3H] = x[T*B][E] W_ih^T + b_ih
    LAED_TRY(laed_sgemm_impl(false, true, T * B, 3 * H, E, x, E, w_ih, E, gates, 3 * H, b_ih, false, nullptr, false, stream));
    if (small) return laed_gru_small_fwd(w_hh, b_hh, h0, hbuf, gates, hn, T, B, H, ws, cu_budget, stream);
    for (int t = 0; t < T; ++t) {
        GruStepArgs a;
        a.hprev = hbuf + (int64_t)t * BH;
        a.w_hh = w_hh;
        a.b_hh = b_hh;
        a.gates = gates + (int64_t)t * B * 3 * H;
        a.hout = hbuf + (int64_t)(t + 1) * BH;
        a.hn = hn ? hn + (int64_t)t * BH : nullptr;
        a.B = B;
        a.H = H;
        LAED_TRY(gru_step(a, stream));
    }
    return 0;
}

extern "C" int laed_gru_bwd(const float* x, const float* w_ih, const float* w_hh,
                            const float* hbuf, float* gates, const float* hn,
                            const float* dout, const float* dhT, int T, int B, int E, int H,
                            float* dgh, float* dx, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh,
                            float* dh0, int accumulate_w, float* ws, int64_t ws_floats, void* stream_) {
    return laed_gru_bwd_cus(x, w_ih, w_hh, hbuf, gates, hn, dout, dhT, T, B, E, H, dgh, dx, dw_ih, dw_hh, db_ih, db_hh, dh0,
                            accumulate_w, ws, ws_floats, 0, stream_);
}

extern "C" int laed_gru_bwd_cus(const float* x, const float* w_ih, const float* w_hh,
                                const float* hbuf, float* gates, const float* hn,
                                const float* dout, const float* dhT, int T, int B, int E, int H,
                                float* dgh, float* dx, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh,
                                float* dh0, int accumulate_w, float* ws, int64_t ws_floats, int cu_budget, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LAED_CHECK_ARG(T > 0 && B > 0 && E > 0 && H > 0, "gru_bwd: non-positive dimension");
    LAED_CHECK_ARG(x && w_ih && w_hh && hbuf && gates && hn && dgh && dw_ih && dw_hh && db_ih && db_hh && dh0,
                   "gru_bwd: null pointer");
    LAED_CHECK_ARG((int64_t)T * B < 0x7fffffff, "gru_bwd: T*B too large");
    const int64_t BH = (int64_t)B * H;
    const int nblk = (int)((BH + 255) / 256 < 2048 ? (BH + 255) / 256 : 2048);
    if (ws == nullptr) ws_floats = 0;
    // split-bf16 mode: bring the big products to the K-contiguous (x W^T) form with transposed fp32 copies carved
    // from the END of the workspace (the front stays split-K scratch)
    const int64_t TBp = ((int64_t)T * B + 31) / 32 * 32;
    const int64_t xneed = (int64_t)3 * H * H + (int64_t)3 * H * E + (int64_t)3 * H * TBp + (int64_t)H * TBp;
    const bool xmode = laed_get_gemm_mode() == 1 && ws_floats >= xneed + (int64_t)4 * B * H;
    float *WhhT = nullptr, *WihT = nullptr, *dghT = nullptr, *hT = nullptr;
    if (xmode) {
        ws_floats -= xneed;
        WhhT = ws + ws_floats;              // [H][3H]
        WihT = WhhT + (int64_t)3 * H * H;   // [E][3H]
        dghT = WihT + (int64_t)3 * H * E;   // [3H][TBp]
        hT = dghT + (int64_t)3 * H * TBp;   // [H][TBp]
        LAED_TRY(laed_transpose_impl(w_hh, H, 3 * H, H, WhhT, 3 * H, stream));
    }
    int nparts = 0;
    // small batches: one fused launch per step (gru_small.hip) over a packed copy of W_hh^T at the front of ws
    const bool small = ws != nullptr && aligned16(ws) && laed_gru_small_ok(B, H) && ws_floats >= laed_gru_small_bwd_ws(B, H);
    if (small) {      // its preparation launch also sets dh0 = dhT (or zeros)
        LAED_TRY(laed_gru_small_bwd(w_hh, hbuf, hn, dout, dhT, gates, dgh, dh0, T, B, H, ws, cu_budget, stream));
    } else {
        hipError_t e;
        if (dhT != nullptr) e = hipMemcpyAsync(dh0, dhT, BH * sizeof(float), hipMemcpyDeviceToDevice, stream);
        else e = hipMemsetAsync(dh0, 0, BH * sizeof(float), stream);
        if (e != hipSuccess) return laed_fail((int)e, "gru_bwd: dh init failed");
    }
    for (int t = small ? -1 : T - 1; t >= 0; --t) {
        GruBwdStepArgs a;
        a.parts = ws;
        a.nparts = nparts;
        a.dout = dout ? dout + (int64_t)t * BH : nullptr;
        a.hprev = hbuf + (int64_t)t * BH;
        a.hn = hn + (int64_t)t * BH;
        a.gates = gates + (int64_t)t * B * 3 * H;
        a.dgh = dgh + (int64_t)t * B * 3 * H;
        a.dh = dh0;
        a.B = B;
        a.H = H;
        hipLaunchKernelGGL(gru_bwd_gates_kernel, dim3(nblk), dim3(256), 0, stream, a);
        LAED_LAUNCH_CHECK("gru_bwd_gates");
        // dh_{t-1} = dh_t * z + dgh_t W_hh      ([B][3H] x [3H][H])
        //   small batches split K over blocks; the partials are summed by the next step's gate kernel
        int S = 1;
        if (xmode)
            LAED_TRY(laed_sgemm_splitk_impl(false, true, B, H, 3 * H, a.dgh, 3 * H, WhhT, 3 * H, dh0, H, nullptr, true, nullptr,
                                            false, ws, ws_floats, true, &S, stream));
        else
            LAED_TRY(laed_sgemm_splitk_impl(false, false, B, H, 3 * H, a.dgh, 3 * H, w_hh, H, dh0, H, nullptr, true, nullptr,
                                            false, ws, ws_floats, true, &S, stream));
        nparts = S > 1 ? S : 0;
    }
    if (nparts > 0) {
        hipLaunchKernelGGL(add_parts_kernel, dim3(nblk), dim3(256), 0, stream, dh0, ws, nparts, BH);
        LAED_LAUNCH_CHECK("add_parts");
    }
    const bool acc = accumulate_w != 0;
    const int TB = T * B;
    // dW_hh[3H][H] = dgh^T[3H][TB] hprev[TB][H] ; db_hh = colsum(dgh)
    if (xmode) {
        LAED_TRY(laed_transpose_impl(dgh, 3 * H, TB, 3 * H, dghT, TBp, stream));
        LAED_TRY(laed_transpose_impl(hbuf, H, TB, H, hT, TBp, stream));
        LAED_TRY(laed_sgemm_rowsum_impl(3 * H, H, TB, dghT, TBp, hT, TBp, dw_hh, H, acc, db_hh, acc, ws, ws_floats, stream));
    } else {
        LAED_TRY(laed_sgemm_splitk_impl(true, false, 3 * H, H, TB, dgh, 3 * H, hbuf, H, dw_hh, H, nullptr, acc, db_hh, acc,
                                        ws, ws_floats, false, nullptr, stream));
    }
    // dW_ih[3H][E] = dgi^T[3H][TB] x[TB][E] ; db_ih = colsum(dgi)
    LAED_TRY(laed_sgemm_splitk_impl(true, false, 3 * H, E, TB, gates, 3 * H, x, E, dw_ih, E, nullptr, acc, db_ih, acc,
                                    ws, ws_floats, false, nullptr, stream));
    // dx[TB][E] = dgi[TB][3H] W_ih[3H][E]   (E = 200 wide: 76 blocks of 64x64 at the reference's batch of 30 -- K may split)
    if (dx != nullptr) {
        if (xmode) {
            LAED_TRY(laed_transpose_impl(w_ih, E, 3 * H, E, WihT, 3 * H, stream));
            LAED_TRY(laed_sgemm_impl(false, true, TB, E, 3 * H, gates, 3 * H, WihT, 3 * H, dx, E, nullptr, false, nullptr, false, stream));
        } else {
            LAED_TRY(laed_sgemm_splitk_impl(false, false, TB, E, 3 * H, gates, 3 * H, w_ih, E, dx, E, nullptr, false, nullptr, false,
                                            ws, ws_floats, false, nullptr, stream));
        }
    }
    return 0;
}
