// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 157 TF peak).
//
//   C[M,N] = op(A) * op(B) (+ bias[N]) (+ C)          all row-major, arbitrary leading dimensions
//
// Block = 256 threads = 4 waves (2 x 2); block tile BM x BN (128x128 or 64x64), K step 32.
// Both operands are staged in LDS as [k][row] (row contiguous, +pad) whatever their layout in HBM, so the
// MFMA fragment read is one conflict-free ds_read_b32 per operand per instruction:
//   lane l holds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]   (guide section 3, f32 32x32x2 map)
// HBM -> registers -> LDS double buffering: the next K tile's global loads are issued before the MFMA block
// of the current tile and written to the other LDS buffer after it; one barrier per K tile.
// Edge tiles are zero-filled on load and masked on store, so any M, N, K works; 16-byte vector loads are
// used when base pointers and leading dimensions allow (VEC), scalar loads otherwise.
// blockIdx -> tile: XCD-aware remap (blocks b, b+8, ... share an XCD/L2) then grouped (8 row tiles) ordering,
// so the blocks resident on one XCD at a time share A row panels and walk the same B column panels.
#include "laed_common.h"

#include <atomic>
#include <math.h>
#include <mutex>
#include <type_traits>
#include <vector>
#include <stdlib.h>
#include <string.h>

namespace {

// ---- optional live profiler (bench.py's roofline leg): HIP events around every GEMM launch, on the launch stream.
// Off by default; when off the launch path touches none of this.  Variant id = layout*2 + (big tile ? 0 : 1).
struct ProfRec { hipEvent_t start, stop; double flops; int variant; };
std::atomic<int64_t> g_xgemm_min_tiles{192};   // bf16x3 mode: NT problems with at least this many 128x128 tiles use xgemm
std::atomic<int> g_gemm_mode{-1};   // process-wide; -1 = not initialised (reads LAED_GEMM_MODE), 0 = exact fp32 MFMA, 1 = split-bf16 (bf16x3) for NT
std::mutex g_prof_mu;
std::atomic<bool> g_prof_on{false};
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;

// 128x128 tile with 4 waves (2x2, wave tile 64x64) or 8 waves (2x4, wave tile 64x32; 4 waves per SIMD at 2 blocks/CU)
int big_tile_threads() {
    static int v = 0;
    if (v == 0) {
        const char* e = laed_exp_hook("LAED_GEMM_THREADS");
        v = (e != nullptr && atoi(e) == 256) ? 256 : 512;
    }
    return v;
}

bool prof_event(hipEvent_t* out) {       // g_prof_mu held
    if (!g_prof_pool.empty()) { *out = g_prof_pool.back(); g_prof_pool.pop_back(); return true; }
    return hipEventCreate(out) == hipSuccess;
}

// start a record: two events, the first recorded on `stream`.  false (and profiling switched off) if an event cannot
// be created, so that no record with an uninitialised handle ever reaches laed_prof_collect
bool prof_begin(ProfRec& r, double flops, int variant, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!prof_event(&r.start)) { g_prof_on = false; return false; }
    if (!prof_event(&r.stop)) { g_prof_pool.push_back(r.start); g_prof_on = false; return false; }
    r.flops = flops;
    r.variant = variant;
    (void)hipEventRecord(r.start, stream);
    return true;
}
void prof_end(ProfRec& r, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(r.stop, stream);
    g_prof_recs.push_back(r);
}

struct SgemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    float* colsum;
    int64_t lda, ldb, ldc;
    int M, N, K;
    int accumulate, colsum_acc;
    int tiles_m, tiles_n;
    int group_m;           // row tiles per group of the grouped block order
    int splitk, kper;      // splitk > 1: block `split` covers K range [split*kper, min(K, (split+1)*kper)) and writes
    float* ws;             //   its raw partial tile to ws[split][M][N] (colsum partials to ws_colsum[split][M])
    float* ws_colsum;
    // EPI == 1 (online-softmax epilogue, forward-only projection + NLL): C is never written; instead
    const int64_t* labels; //   labels[M]
    float* part;           //   part[tile_n][M][2] = (max, sum exp(x - max)) over the tile's columns of row m
    float* lab_logit;      //   lab_logit[m] = logit[m][labels[m]] (written by the one lane that holds it)
    // phase lock (sync != nullptr): the blocks that are co-resident on one XCD (64 consecutive blocks of its range) meet
    // every `sync_every` K tiles, so that they keep walking their shared operand strips together (see phase_lock below)
    unsigned* sync;        //   one zeroed counter per (XCD, group of 64 blocks)
    int sync_every;        //   K tiles between meetings (a multiple of 2)
    int sync_shift;        //   log2 of the blocks per meeting group (6: the XCD's 64 resident blocks, 5: one block per CU)
};

#ifndef SGEMM_BK16_WAVES
#define SGEMM_BK16_WAVES 2      // minimum waves per SIMD asked of the compiler for the 16-deep instantiations
#endif
#ifndef SGEMM_BK
#define SGEMM_BK 32
#endif
constexpr int BK = SGEMM_BK;   // K tile depth (16 measured in round 2: more blocks per CU)

// Load a (BR rows) x (BK k) tile into registers as float4s.
//  KCONTIG: element (r,k) at P[r*ld + k]   -> each thread takes 4 consecutive k of one row
// !KCONTIG: element (r,k) at P[k*ld + r]   -> each thread takes 4 consecutive rows of one k
template <bool KCONTIG, int BR, bool VEC, int NT, int BKT = BK>
__device__ __forceinline__ void tile_load(const float* __restrict__ P, int64_t ld, int r0, int k0, int R, int K,
                                          int tid, float4 (&reg)[BR * BKT / 4 / NT]) {
    constexpr int BK = BKT;
    constexpr int NL = BR * BK / 4 / NT;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        int idx = tid + i * NT;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KCONTIG) {
            int row = idx / (BK / 4), kq = idx % (BK / 4);
            int gr = r0 + row, gk = k0 + 4 * kq;
            if (gr < R && gk < K) {
                const float* src = P + (int64_t)gr * ld + gk;
                if (VEC && gk + 3 < K) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0];
                    if (gk + 1 < K) v.y = src[1];
                    if (gk + 2 < K) v.z = src[2];
                    if (gk + 3 < K) v.w = src[3];
                }
            }
        } else {
            int k = idx / (BR / 4), rq = idx % (BR / 4);
            int gk = k0 + k, gr = r0 + 4 * rq;
            if (gk < K && gr < R) {
                const float* src = P + (int64_t)gk * ld + gr;
                if (VEC && gr + 3 < R) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0];
                    if (gr + 1 < R) v.y = src[1];
                    if (gr + 2 < R) v.z = src[2];
                    if (gr + 3 < R) v.w = src[3];
                }
            }
        }
        reg[i] = v;
    }
}

// Write the registers to the LDS image s[BK][S], S = BR + pad (pad 1 for KCONTIG: conflict-free transposing
// b32 writes; pad 4 otherwise: 16-byte aligned b128 writes).
template <bool KCONTIG, int BR, int NT, int BKT = BK>
__device__ __forceinline__ void tile_store(float* __restrict__ s, int tid, const float4 (&reg)[BR * BKT / 4 / NT]) {
    constexpr int BK = BKT;
    constexpr int NL = BR * BK / 4 / NT;
    constexpr int S = BR + (KCONTIG ? 1 : 4);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        int idx = tid + i * NT;
        if (KCONTIG) {
            int row = idx / (BK / 4), kq = idx % (BK / 4);
            float* d = s + (4 * kq) * S + row;
            d[0] = reg[i].x;
            d[S] = reg[i].y;
            d[2 * S] = reg[i].z;
            d[3 * S] = reg[i].w;
        } else {
            int k = idx / (BR / 4), rq = idx % (BR / 4);
            *reinterpret_cast<float4*>(s + k * S + 4 * rq) = reg[i];
        }
    }
}

// C/D map of v_mfma_f32_32x32x2_f32:  col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
template <int BM, int BN, int WGM, int WGN, int TM, int TN>
__device__ __forceinline__ void store_acc(const SgemmArgs& p, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                          int lrow, int lk, int split) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / WGN) + j * 32 + lrow;
            const float bv = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * (BM / WGM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                if (row < p.M && col < p.N) {
                    if (p.splitk > 1) {
                        p.ws[((int64_t)split * p.M + row) * p.N + col] = acc[i][j][e];
                    } else {
                        float* dst = p.C + (int64_t)row * p.ldc + col;
                        float v = acc[i][j][e] + bv;
                        if (p.accumulate) v += *dst;
                        *dst = v;
                    }
                }
            }
        }
    }
}

template <int BM, int BN, bool TA, bool TB, bool VEC, int NT, int EPI = 0, int BKT = BK, int PF = 1>
__device__ __forceinline__ void sgemm_block(const SgemmArgs& p) {
    constexpr int SGEMM_PF2 = PF == 2 ? 1 : 0;   // PF = 2: the scheduled loop keeps TWO K tiles in flight in registers
    constexpr int BK = BKT;    // K tile depth of THIS instantiation (the file-scope BK stays the host side's K granule)
    constexpr bool AK = !TA;   // A is K-contiguous when not transposed ([M][K])
    constexpr bool BKC = TB;   // B is K-contiguous when transposed   ([N][K])
    constexpr int SA = BM + (AK ? 1 : 4);
    constexpr int SB = BN + (BKC ? 1 : 4);
    // waves form a WGM x WGN grid over the block tile: 2 x 2 (256 threads), 2 x 4 (512); wave tile (BM/WGM) x (BN/WGN).
    // (4 x 4 = 1024 threads on a 256x256 tile, one block per CU, was measured: 137.6 vs 135.4 TFLOP/s at 8192^3 but
    // slower on every projection shape once their tile counts are counted in rounds of 256 slots -- not kept.
    // LDS-DMA staging (global_load_lds_dwordx4 into lane-linear images, XOR-swizzled quads read with ds_read_b128 and
    // a permuted k order; 12 DS + 4 VMEM instead of 40 DS + 27 VALU + 4 VMEM per K tile) was built and measured too:
    // 132.4 vs 135.3 TFLOP/s at 8192^3 -- two LDS buffers force the DMA of tile t+1 into tile t, and the one-tile
    // prefetch distance of the register path below is worth more than the instructions saved -- not kept.)
    constexpr int WGM = NT >= 1024 ? 4 : 2;
    constexpr int WGN = NT / 64 / WGM;
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
    constexpr int NLA = BM * BK / 4 / NT, NLB = BN * BK / 4 / NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                 // [2][BK][SA]
    float* sB = smem + 2 * BK * SA;   // [2][BK][SB]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: keep it (and wm, wn) in SGPRs
    const int wm = wave / WGN, wn = wave % WGN;

    // XCD-aware remap (bijective for any grid size), then grouped ordering over row tiles
    int nwg = gridDim.x, pid = blockIdx.x;
    int xcd = pid & 7, q = nwg >> 3, r = nwg & 7;
    int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (pid >> 3);
#ifdef LAED_ENV_HOOKS
    // phase-lock identity: XCD `xcd` runs the q (+1) blocks of its contiguous wg range in order, 64 at a time (32 CUs x 2)
    const int pl_local = pid >> 3, pl_nx = q + (xcd < r ? 1 : 0);
    const int pl_g = 1 << p.sync_shift;
    unsigned* pl_cnt = p.sync ? p.sync + xcd * ((q + pl_g) >> p.sync_shift) + (pl_local >> p.sync_shift) : nullptr;
    const unsigned pl_size = (unsigned)min(pl_g, pl_nx - (pl_local & ~(pl_g - 1)));
    unsigned pl_epoch = 0;
    bool pl_wait = true;
#endif
    int split = 0;
    if (p.splitk > 1) { split = wg % p.splitk; wg /= p.splitk; }
    const int kbeg = split * p.kper;
    const int kend = (p.splitk > 1) ? min(p.K, kbeg + p.kper) : p.K;
    const int GROUP_M = p.group_m;
    int in_group = GROUP_M * p.tiles_n;
    int gid = wg / in_group, first_m = gid * GROUP_M;
    int gsz = min(p.tiles_m - first_m, GROUP_M);
    int tile_m = first_m + (wg % in_group) % gsz;
    int tile_n = (wg % in_group) / gsz;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[NLA], rb[NLB];
    float4 ra2[NLA], rb2[NLB];       // second staging set (PF = 2): the scheduled loop prefetches TWO K tiles ahead
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_colsum = TA && (p.colsum != nullptr) && (tile_n == 0);
    const int lrow = lane & 31, lk = lane >> 5;
    const int nkt = (kend - kbeg + BK - 1) / BK;
    const float* a_base = sA + wm * (BM / WGM) + lrow + lk * SA;   // fragment row (2*kk + lk) folded into the base
    const float* b_base = sB + wn * (BN / WGN) + lrow + lk * SB;

    auto add_colsum = [&](int set = 0) {
        if (TA && do_colsum) {
            asm volatile("; colsum" ::: "memory");     // keeps this a (scalar) branch: 7 of 8 blocks skip the adds
            if (SGEMM_PF2 && set) {
#pragma unroll
                for (int i = 0; i < NLA; ++i) { csum.x += ra2[i].x; csum.y += ra2[i].y; csum.z += ra2[i].z; csum.w += ra2[i].w; }
                return;
            }
#pragma unroll
            for (int i = 0; i < NLA; ++i) { csum.x += ra[i].x; csum.y += ra[i].y; csum.z += ra[i].z; csum.w += ra[i].w; }
        }
    };

    // Interior blocks with vector-loadable operands run their full K tiles on the scheduled loop: a straight-line
    // body whose global loads (tile t+1), LDS fragment reads (next k pair) and LDS stores (tile t+1) are placed one
    // by one in the issue gaps between consecutive MFMAs (a 32x32x2 f32 MFMA holds the matrix pipe for 64 cycles
    // and a wave issues in order, so whatever sits between two MFMAs is hidden).  Edge blocks and the K tail use
    // the generic loop with the guarded loader.
    const bool interior = VEC && (m0 + BM <= p.M) && (n0 + BN <= p.N);
    const int nfull = (kend - kbeg) / BK;   // K tiles that need no K-tail guard
    const bool fast = interior && nfull > 0;
    constexpr int NMF = TM * TN;         // MFMAs per k pair
    constexpr int Q = (BK / 2) * NMF;    // MFMA issue gaps per K tile
    constexpr int NL = NLA + NLB;

    // per-thread source pointers of the first float4 of K tile 0; the NL loads per operand are a fixed stride apart
    const float* ga;
    const float* gb;
    int64_t a_step, b_step, a_ist, b_ist;
    if (AK) { ga = p.A + (int64_t)(m0 + tid / (BK / 4)) * p.lda + kbeg + 4 * (tid % (BK / 4)); a_step = BK; a_ist = (int64_t)(NT / (BK / 4)) * p.lda; }
    else    { ga = p.A + (int64_t)(kbeg + tid / (BM / 4)) * p.lda + m0 + 4 * (tid % (BM / 4)); a_step = (int64_t)BK * p.lda; a_ist = (int64_t)(NT / (BM / 4)) * p.lda; }
    if (BKC) { gb = p.B + (int64_t)(n0 + tid / (BK / 4)) * p.ldb + kbeg + 4 * (tid % (BK / 4)); b_step = BK; b_ist = (int64_t)(NT / (BK / 4)) * p.ldb; }
    else     { gb = p.B + (int64_t)(kbeg + tid / (BN / 4)) * p.ldb + n0 + 4 * (tid % (BN / 4)); b_step = (int64_t)BK * p.ldb; b_ist = (int64_t)(NT / (BN / 4)) * p.ldb; }

    // one store "unit" = one float4 register; per-thread LDS offset + compile-time unit offsets
    const int sa_off = AK ? (4 * (tid % (BK / 4))) * SA + tid / (BK / 4) : (tid / (BM / 4)) * SA + 4 * (tid % (BM / 4));
    const int sb_off = BKC ? (4 * (tid % (BK / 4))) * SB + tid / (BK / 4) : (tid / (BN / 4)) * SB + 4 * (tid % (BN / 4));
    auto store_unit = [&](int u, int buf, int set = 0) {
        if (u < NLA) {
            const float4 x = (SGEMM_PF2 && set) ? ra2[u] : ra[u];
            float* d = sA + buf * BK * SA + sa_off + (AK ? u * (NT / (BK / 4)) : u * (NT / (BM / 4)) * SA);
            if (AK) { d[0] = x.x; d[SA] = x.y; d[2 * SA] = x.z; d[3 * SA] = x.w; }
            else *reinterpret_cast<float4*>(d) = x;
        } else {
            const int v = u - NLA;
            const float4 x = (SGEMM_PF2 && set) ? rb2[v] : rb[v];
            float* d = sB + buf * BK * SB + sb_off + (BKC ? v * (NT / (BK / 4)) : v * (NT / (BN / 4)) * SB);
            if (BKC) { d[0] = x.x; d[SB] = x.y; d[2 * SB] = x.z; d[3 * SB] = x.w; }
            else *reinterpret_cast<float4*>(d) = x;
        }
    };
    auto load_unit = [&](int u, int set = 0) {
        if (SGEMM_PF2 && set) {
            if (u < NLA) ra2[u] = *reinterpret_cast<const float4*>(ga + u * a_ist);
            else rb2[u - NLA] = *reinterpret_cast<const float4*>(gb + (u - NLA) * b_ist);
            return;
        }
        if (u < NLA) ra[u] = *reinterpret_cast<const float4*>(ga + u * a_ist);
        else rb[u - NLA] = *reinterpret_cast<const float4*>(gb + (u - NLA) * b_ist);
    };

    // ---- prologue: tile 0
    if (fast) {
#pragma unroll
        for (int u = 0; u < NL; ++u) load_unit(u);
    } else {
        tile_load<AK, BM, VEC, NT, BK>(p.A, p.lda, m0, kbeg, p.M, kend, tid, ra);
        tile_load<BKC, BN, VEC, NT, BK>(p.B, p.ldb, n0, kbeg, p.N, kend, tid, rb);
    }
    add_colsum();
#pragma unroll
    for (int u = 0; u < NL; ++u) store_unit(u, 0);
    __syncthreads();

    int cur = 0, kt = 0;
    if (fast && nfull >= 2) {
        // ---- scheduled loop, prefetch distance = one whole K tile: while tile kt computes out of LDS[cur], the
        // registers (tile kt+1, fetched during tile kt-1) are written to LDS[cur^1] and immediately refilled with
        // tile kt+2, unit by unit.  A global load therefore has a full tile of MFMA time (>= 4096 cycles per wave)
        // to land before its s_waitcnt, which covers an HBM miss under load; with the half-tile distance this loop
        // used before, every K tile stalled on vmcnt in its middle.  The last pass has no tile kt+2 to fetch and
        // re-reads tile kt+1 (cache hit, result unused) instead of branching.
        ga += a_step;
        gb += b_step;
        if (SGEMM_PF2) {
            // prefetch distance = TWO K tiles: tile kt+1 sits in register set (kt+1)&1, tile kt+2 in the other one; while
            // tile kt computes, set (kt+1)&1 goes to LDS[cur^1] and is refilled with tile kt+3
#pragma unroll
            for (int u = 0; u < NL; ++u) load_unit(u, 1);
            if (2 < nfull) { ga += a_step; gb += b_step; }
#pragma unroll
            for (int u = 0; u < NL; ++u) load_unit(u, 0);
        } else {
#pragma unroll
            for (int u = 0; u < NL; ++u) load_unit(u);
        }
        // one K tile out of LDS buffer CUR (compile-time, so that every LDS offset of the body is an immediate)
        auto tile_body = [&](auto cur_c, bool advance) {
            constexpr int CUR = decltype(cur_c)::value;
            const float* a_s = a_base + CUR * BK * SA;
            const float* b_s = b_base + CUR * BK * SB;
#if !defined(SGEMM_EXP) || !(SGEMM_EXP & 32)
            {   // uniform (SGPR) select of the step, then one 64-bit add per pointer: a select between two POINTERS made the
                // compiler split the following dwordx4 loads into dwordx2 halves with separate address arithmetic
                const int64_t sa_ = advance ? a_step : 0, sb_ = advance ? b_step : 0;
                ga += sa_;
                gb += sb_;
            }
#endif
            // fragments are read TWO k pairs ahead (three register sets): the wait before a k pair's first MFMA then
            // covers reads issued a whole k pair (>= 256 cycles) earlier and leaves the newest reads in flight
            // (counted lgkmcnt), which is what one or two waves per SIMD need to keep the LDS latency off the MFMA chain
            float af[3][TM], bf[3][TN];
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[s_][i] = a_s[(2 * s_) * SA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[s_][j] = b_s[(2 * s_) * SB + j * 32];
            }
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int q_ = kk * NMF + i * TN + j;
                        laed_mfma_32x32x2(acc[i][j], af[kk % 3][i], bf[kk % 3][j]);
#if !defined(SGEMM_EXP) || !(SGEMM_EXP & 4)
                        if (i * TN + j == 0 && kk + 2 < BK / 2) {        // fragments of the k pair after the next
#pragma unroll
                            for (int ii = 0; ii < TM; ++ii) af[(kk + 2) % 3][ii] = a_s[(2 * (kk + 2)) * SA + ii * 32];
#pragma unroll
                            for (int jj = 0; jj < TN; ++jj) bf[(kk + 2) % 3][jj] = b_s[(2 * (kk + 2)) * SB + jj * 32];
                        }
#else
                        if (i * TN + j == 0 && kk + 2 < BK / 2) {
                            for (int ii = 0; ii < TM; ++ii) af[(kk + 2) % 3][ii] = af[kk % 3][ii];
                            for (int jj = 0; jj < TN; ++jj) bf[(kk + 2) % 3][jj] = bf[kk % 3][jj];
                        }
#endif
#if !defined(SGEMM_EXP) || !(SGEMM_EXP & 2)
#pragma unroll
                        for (int u = 0; u < NL; ++u) {
                            const int gs = Q / 8 + (u * (3 * Q / 4)) / NL;               // store gap of unit u
                            if (q_ == gs) {
                                if (u == 0) add_colsum(SGEMM_PF2 ? (CUR ^ 1) : 0);
#if defined(SGEMM_EXP) && (SGEMM_EXP & 8)
                                if (u < NLA) asm volatile("" :: "v"(ra[u].x), "v"(ra[u].y), "v"(ra[u].z), "v"(ra[u].w));
                                else asm volatile("" :: "v"(rb[u - NLA].x), "v"(rb[u - NLA].y), "v"(rb[u - NLA].z), "v"(rb[u - NLA].w));
#else
                                store_unit(u, CUR ^ 1, SGEMM_PF2 ? (CUR ^ 1) : 0);       // tile kt+1 -> LDS
#endif
                            }
#if !defined(SGEMM_EXP) || !(SGEMM_EXP & 16)
                            if (q_ == gs + 1) load_unit(u, SGEMM_PF2 ? (CUR ^ 1) : 0);   // tile kt+2 (kt+3 with two sets) -> registers
#endif
                        }
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
#if !defined(SGEMM_EXP) || !(SGEMM_EXP & 1)
            __syncthreads();
#endif
        };
#ifdef LAED_ENV_HOOKS
        // Phase lock (EXPERIMENT, hooks build only, LAED_GEMM_PHASE=<K tiles>; measured and NOT adopted: DESIGN.md section 4
        // "L2 traffic of the projection's backward products").
        // The 64 blocks that are resident on an XCD share operand strips through its 4 MB L2 (8 row tiles x 8 column tiles: every strip is read by 8 blocks), which only works while they read the same K range at about
        // the same time: the L2 holds ~16 K tiles of the group's stream.  Nothing keeps them together -- a block's loads
        // never stall it (they are issued a whole tile ahead), so a block that falls behind stays behind and re-fetches
        // what its group already evicted (PMC: 3.2x the fabric traffic the tiling needs on the dW product).  Every
        // `sync_every` K tiles one thread per block adds itself to the group's counter and waits, BOUNDED, for the
        // others: lockstep groups also retire together, so the next 64 blocks start together.  A wait that runs out
        // (the group is not co-resident: odd occupancy, another stream's blocks on the XCD) switches waiting off for
        // that block -- it keeps counting, so nobody waits for it -- and the product never depends on the counters.
        auto phase_lock = [&]() {
            if (tid == 0) {
                ++pl_epoch;
                __hip_atomic_fetch_add(pl_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (pl_wait) {
                    const unsigned target = pl_size * pl_epoch;
                    int spins = 0;
                    while (__hip_atomic_load(pl_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                        if (++spins > 64) { pl_wait = false; break; }           // ~30 us
                        __builtin_amdgcn_s_sleep(8);
                    }
                }
            }
            __syncthreads();
        };
#endif
        for (; kt + 2 < nfull; kt += 2) {           // cur == 0 here
            tile_body(std::integral_constant<int, 0>(), kt + 2 + SGEMM_PF2 < nfull);
            tile_body(std::integral_constant<int, 1>(), kt + 3 + SGEMM_PF2 < nfull);
#ifdef LAED_ENV_HOOKS
            if (pl_cnt != nullptr && ((kt + 2) % p.sync_every) == 0) phase_lock();
#endif
        }
        if (kt + 1 < nfull) {
            tile_body(std::integral_constant<int, 0>(), false);
            ++kt;
            cur = 1;
        }
    }
    // ---- generic loop: edge blocks from tile 0; interior blocks only their last full tile and the K tail
    for (; kt < nkt; ++kt) {
        const bool more = (kt + 1 < nkt);
        if (more) {
            tile_load<AK, BM, VEC, NT, BK>(p.A, p.lda, m0, kbeg + (kt + 1) * BK, p.M, kend, tid, ra);
            tile_load<BKC, BN, VEC, NT, BK>(p.B, p.ldb, n0, kbeg + (kt + 1) * BK, p.N, kend, tid, rb);
        }
        const float* a_s = a_base + cur * BK * SA;
        const float* b_s = b_base + cur * BK * SB;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = a_s[2 * kk * SA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = b_s[2 * kk * SB + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    laed_mfma_32x32x2(acc[i][j], a[i], b[j]);
        }
        // Drained INSIDE the body, every pass (this loop is not the hot one): behind the loop's exit the register allocator may
        // move a tile (v_accvgpr_mov / read on the exit edge, straight behind an MFMA it cannot see -- the MFMA is inline asm,
        // and a drain placed after the loop comes too late for such a copy).  tools/check_mfma_hazards.py found exactly that in
        // the unaligned-operand instantiations (round 3) and checks every build for it.
        laed_mfma_drain(acc);
        if (more) {
            add_colsum();
            tile_store<AK, BM, NT, BK>(sA + (cur ^ 1) * BK * SA, tid, ra);
            tile_store<BKC, BN, NT, BK>(sB + (cur ^ 1) * BK * SB, tid, rb);
        }
        __syncthreads();
        cur ^= 1;
    }
    laed_mfma_drain(acc);        // the accumulators are read (v_accvgpr_read) from here on

    if (EPI == 1) {
        // ---- online-softmax epilogue: per row of the block tile, (max, sum exp) over its BN columns and the label's
        // logit; nothing else leaves the block.  A row's 32 columns of one MFMA tile sit in the 32 lanes that share
        // lane>>5, so xor-shuffles up to 16 reduce them; the WGN waves that cover the tile's columns meet in LDS and are
        // combined in wave order (fixed order: reproducible).
        float* red = smem;                      // [WGN][BM][2]; every wave is past the main loop's last barrier
        // the labels of this lane's 32 rows, fetched in one batch: read one by one inside the loop below they were 32
        // dependent round trips per wave (each behind the previous row's lab_logit store, which may alias them) -- the
        // epilogue then took about as long as the tile's K loop (validation forward 387 ms at the scaled shape)
        int labs[TM][16];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * (BM / WGM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                const int64_t l64 = (row < p.M) ? p.labels[row] : -1;
                labs[i][e] = (l64 >= 0 && l64 < p.N) ? (int)l64 : -1;
            }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = wm * (BM / WGM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                const int row = m0 + rl;
                const int lab = labs[i][e];
                float v[TN];
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + wn * (BN / WGN) + j * 32 + lrow;
                    v[j] = -INFINITY;
                    if (col < p.N) {
                        v[j] = acc[i][j][e] + (p.bias != nullptr ? p.bias[col] : 0.f);
                        if (col == lab) p.lab_logit[row] = v[j];
                    }
                    m = fmaxf(m, v[j]);
                }
                m = fmaxf(m, laed_bfly32<1>(m)); m = fmaxf(m, laed_bfly32<2>(m)); m = fmaxf(m, laed_bfly32<4>(m));
                m = fmaxf(m, laed_bfly32<8>(m)); m = fmaxf(m, laed_bfly32<16>(m));
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) sum += (v[j] == -INFINITY) ? 0.f : expf(v[j] - m);
                sum += laed_bfly32<1>(sum); sum += laed_bfly32<2>(sum); sum += laed_bfly32<4>(sum);
                sum += laed_bfly32<8>(sum); sum += laed_bfly32<16>(sum);
                if (lrow == 0) { red[(wn * BM + rl) * 2] = m; red[(wn * BM + rl) * 2 + 1] = sum; }
            }
        }
        __syncthreads();
        if (tid < BM && m0 + tid < p.M) {
            float m = red[tid * 2], sum = red[tid * 2 + 1];
#pragma unroll
            for (int w = 1; w < WGN; ++w) {
                const float m2 = red[(w * BM + tid) * 2], s2 = red[(w * BM + tid) * 2 + 1];
                const float mm = fmaxf(m, m2);
                sum = (m == -INFINITY ? 0.f : sum * expf(m - mm)) + (m2 == -INFINITY ? 0.f : s2 * expf(m2 - mm));
                m = mm;
            }
            float* dst = p.part + ((int64_t)tile_n * p.M + (m0 + tid)) * 2;
            dst[0] = m;
            dst[1] = sum;
        }
        return;
    }

    if (EPI == 2) {
        // ---- arg-max epilogue (greedy decode): per row of the block tile the largest logit of its BN columns and that
        // column's index -- lowest index on ties, as laed_argmax_rows -- and nothing else leaves the block.  Same
        // lane / wave order as the softmax epilogue above; the index travels as the bits of a float.
        float* red = smem;                      // [WGN][BM][2]
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = wm * (BM / WGM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                float bv = -INFINITY;
                int bi = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + wn * (BN / WGN) + j * 32 + lrow;
                    if (col < p.N) {
                        const float v = acc[i][j][e] + (p.bias != nullptr ? p.bias[col] : 0.f);
                        if (v > bv) { bv = v; bi = col; }
                    }
                }
#define LAED_ARGMAX_STAGE(O)                                                        \
                {                                                                   \
                    const float ov = laed_bfly32<O>(bv);                            \
                    const int oi = laed_bfly32_i<O>(bi);                            \
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }     \
                }
                LAED_ARGMAX_STAGE(1) LAED_ARGMAX_STAGE(2) LAED_ARGMAX_STAGE(4) LAED_ARGMAX_STAGE(8) LAED_ARGMAX_STAGE(16)
#undef LAED_ARGMAX_STAGE
                if (lrow == 0) { red[(wn * BM + rl) * 2] = bv; red[(wn * BM + rl) * 2 + 1] = __int_as_float(bi); }
            }
        }
        __syncthreads();
        if (tid < BM && m0 + tid < p.M) {
            float bv = red[tid * 2];
            int bi = __float_as_int(red[tid * 2 + 1]);
#pragma unroll
            for (int w = 1; w < WGN; ++w) {
                const float ov = red[(w * BM + tid) * 2];
                const int oi = __float_as_int(red[(w * BM + tid) * 2 + 1]);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            float* dst = p.part + ((int64_t)tile_n * p.M + (m0 + tid)) * 2;
            dst[0] = bv;
            dst[1] = __int_as_float(bi);
        }
        return;
    }

    // ---- epilogue
    store_acc<BM, BN, WGM, WGN, TM, TN>(p, acc, m0, n0, wm, wn, lrow, lk, split);

    // ---- optional column sums of A (TA only): colsum[m] = sum_k A[k][m]
    if (TA) {
        if (do_colsum) {   // block-uniform
            // every load of this thread has the same rq = tid % (BM/4); NT/(BM/4) threads share an rq
            constexpr int RQ = BM / 4, G = NT / RQ;
            float* red = smem;   // reuse (all waves passed the final barrier of the main loop)
            const int rq = tid % RQ, g = tid / RQ;
            red[(g * RQ + rq) * 4 + 0] = csum.x;
            red[(g * RQ + rq) * 4 + 1] = csum.y;
            red[(g * RQ + rq) * 4 + 2] = csum.z;
            red[(g * RQ + rq) * 4 + 3] = csum.w;
            __syncthreads();
            if (tid < BM) {
                float s = 0.f;
#pragma unroll
                for (int gg = 0; gg < G; ++gg) s += red[(gg * RQ + (tid >> 2)) * 4 + (tid & 3)];
                const int m = m0 + tid;
                if (m < p.M) {
                    if (p.splitk > 1) p.ws_colsum[(int64_t)split * p.M + m] = s;
                    else p.colsum[m] = p.colsum_acc ? p.colsum[m] + s : s;
                }
            }
        }
    }
}

template <int BM, int BN, bool TA, bool TB, bool VEC, int NT, int EPI = 0, int BKT = BK, int PF = 1>
__global__ __launch_bounds__(NT, (NT >= 1024 ? 4 : (BKT == 16 && NT == 512 ? SGEMM_BK16_WAVES : 2))) void sgemm_kernel(SgemmArgs p) {
    sgemm_block<BM, BN, TA, TB, VEC, NT, EPI, BKT, PF>(p);
}

#ifdef LAED_ENV_HOOKS
// ---------------------------------------------------------------------------------------------------------------
// EXPERIMENT (hooks build only, LAED_GEMM_GLDS=1; measured and NOT adopted -- DESIGN.md section 4, "LDS-DMA, round 2").
// LDS-DMA variant of the 128x128 / 8-wave kernel (interior blocks, 16-byte-loadable operands, K % 16 == 0).
// The ablations of the register-staged loop above (tools/gemm_exp.py, dW shape with the tail split) price its staging at
// 17 % of the kernel: 128 TF as built, 135 without the LDS stores, 141 without the global loads, 150 with neither --
// while the fragment reads cost nothing.  Here neither operand ever touches a VGPR on its way in:
//   * global_load_lds_dwordx4: each wave-instruction moves 1 KiB straight into LDS (lane-linear destination, per-lane
//     source address), 2 instructions per wave per 16-deep K sub-tile (A piece + B piece);
//   * a ring of NST = 4 sub-tile stages (64 KiB per block: still 2 blocks per CU); the DMA of sub-tile t+3 is issued at
//     the top of sub-tile t and only has to land by the END of sub-tile t+1: counted s_waitcnt vmcnt(4) + raw s_barrier
//     (a __syncthreads() would drain the DMA queue: vmcnt(0));
//   * K-contiguous operands ([row][k] in memory) keep that order in LDS: image [128 rows][16 k], the four 16-byte quads
//     of a row XOR-swizzled by (row >> 2) & 3 on the SOURCE address, so that one conflict-free ds_read_b128 hands a lane
//     four k-values; the contraction is order-free, so MFMA step (g, j) takes k = 8g + 4*(lane>>5) + j from BOTH
//     operands.  Other operands ([k][row] in memory) keep [16 k][128 rows] and are read with ds_read_b32 as before.
// Edge blocks and K ranges that do not qualify run the register-staged block body above (same launch, same LDS).
#define LAED_GPTR(x) ((const __attribute__((address_space(1))) void*)(x))
#define LAED_LPTR(x) ((__attribute__((address_space(3))) void*)(x))

template <bool TA, bool TB>
__global__ __launch_bounds__(512, 2) void sgemm_glds_kernel(SgemmArgs p) {
    constexpr int BM = 128, BN = 128, NT = 512, WGM = 2, WGN = 4, TM = 2, TN = 1;
    constexpr int KS = 16, NST = 4;
    constexpr bool AK = !TA, BKC = TB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int lrow = lane & 31, lk = lane >> 5;

    int nwg = gridDim.x, pid = blockIdx.x;
    int xcd = pid & 7, q = nwg >> 3, r = nwg & 7;
    int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (pid >> 3);
    int split = 0;
    if (p.splitk > 1) { split = wg % p.splitk; wg /= p.splitk; }
    const int kbeg = split * p.kper;
    const int kend = (p.splitk > 1) ? min(p.K, kbeg + p.kper) : p.K;
    const int GROUP_M = p.group_m;
    int in_group = GROUP_M * p.tiles_n;
    int gid = wg / in_group, first_m = gid * GROUP_M;
    int gsz = min(p.tiles_m - first_m, GROUP_M);
    int tile_m = first_m + (wg % in_group) % gsz;
    int tile_n = (wg % in_group) / gsz;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int klen = kend - kbeg;
    if (!((m0 + BM <= p.M) && (n0 + BN <= p.N) && klen >= KS * NST && (klen % KS) == 0)) {     // block-uniform
        sgemm_block<BM, BN, TA, TB, true, NT, 0>(p);
        return;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][0][e] = 0.f;

    float* imgA = smem;                         // [NST][BM * KS]
    float* imgB = smem + NST * BM * KS;         // [NST][BN * KS]
    const int nsub = klen / KS;

    // per-lane source of this wave's 1-KiB piece of each operand's sub-tile image
    const float* pa;
    const float* pb;
    int64_t pa_t, pb_t;
    if (AK) {
        const int row = 16 * wave + (lane >> 2), qs = (lane & 3) ^ ((row >> 2) & 3);
        pa = p.A + (int64_t)(m0 + row) * p.lda + kbeg + 4 * qs; pa_t = KS;
    } else {
        pa = p.A + (int64_t)(kbeg + 2 * wave + (lane >> 5)) * p.lda + m0 + 4 * (lane & 31); pa_t = (int64_t)KS * p.lda;
    }
    if (BKC) {
        const int row = 16 * wave + (lane >> 2), qs = (lane & 3) ^ ((row >> 2) & 3);
        pb = p.B + (int64_t)(n0 + row) * p.ldb + kbeg + 4 * qs; pb_t = KS;
    } else {
        pb = p.B + (int64_t)(kbeg + 2 * wave + (lane >> 5)) * p.ldb + n0 + 4 * (lane & 31); pb_t = (int64_t)KS * p.ldb;
    }
    auto issue = [&](int stage, bool advance) {
        __builtin_amdgcn_global_load_lds(LAED_GPTR(pa), LAED_LPTR(imgA + stage * BM * KS + wave * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(LAED_GPTR(pb), LAED_LPTR(imgB + stage * BN * KS + wave * 256), 16, 0, 0);
        if (advance) { pa += pa_t; pb += pb_t; }
    };

    // fragment read offsets (floats) inside a stage image
    const int arow = wm * (BM / WGM) + lrow, brow = wn * (BN / WGN) + lrow;
    const int fsw = (lrow >> 2) & 3;            // (row >> 2) & 3: wave / MFMA-tile row offsets are multiples of 32
    int aoff[2], boff[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        aoff[g] = AK ? arow * KS + 4 * ((2 * g + lk) ^ fsw) : (4 * (2 * g + lk)) * BM + arow;
        boff[g] = BKC ? brow * KS + 4 * ((2 * g + lk) ^ fsw) : (4 * (2 * g + lk)) * BN + brow;
    }
    const bool do_colsum = TA && (p.colsum != nullptr) && (tile_n == 0);
    const bool csum_wave = do_colsum && wn == 0;
    float fsum[TM] = {0.f, 0.f};

    // fragments of one sub-tile: fa[g][i][j], fb[g][j] for MFMA step (g, j)
    struct Frag { float a[2][TM][4]; float b[2][4]; };
    auto read_frags = [&](int tile, Frag& f) {
        const float* a_s = imgA + (tile & (NST - 1)) * BM * KS;
        const float* b_s = imgB + (tile & (NST - 1)) * BN * KS;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (AK) {
                    const float4 v = *reinterpret_cast<const float4*>(a_s + aoff[g] + i * 32 * KS);
                    f.a[g][i][0] = v.x; f.a[g][i][1] = v.y; f.a[g][i][2] = v.z; f.a[g][i][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) f.a[g][i][j] = a_s[aoff[g] + j * BM + i * 32];
                }
            }
            if (BKC) {
                const float4 v = *reinterpret_cast<const float4*>(b_s + boff[g]);
                f.b[g][0] = v.x; f.b[g][1] = v.y; f.b[g][2] = v.z; f.b[g][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) f.b[g][j] = b_s[boff[g] + j * BN];
            }
        }
    };
    auto mma_half = [&](const Frag& f, int g) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                laed_mfma_32x32x2(acc[i][0], f.a[g][i][j], f.b[g][j]);
        if (TA && csum_wave) {                      // wave-uniform: column sums of A from this wave's own fragments
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) fsum[i] += f.a[g][i][j];
        }
    };

    // ---- prologue: sub-tiles 0..2 in flight, sub-tile 0 landed, its fragments in registers
#pragma unroll
    for (int s_ = 0; s_ < NST - 1; ++s_) issue(s_, s_ + 1 < nsub);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    Frag f0, f1;
    read_frags(0, f0);

    // One sub-tile: first half of its MFMAs; then the ONE barrier of the step -- by then every wave holds all of this
    // sub-tile's fragments in registers (they were read during the previous step) and, after the counted vmcnt, its
    // pieces of sub-tile t+1 have landed -- so behind it sub-tile t+1 is readable and the stage of sub-tile t-1 is
    // free: issue the DMA of sub-tile t+3 into it, read sub-tile t+1's fragments into the other register set, and run
    // the second half of the MFMAs over those reads' latency.
    auto step = [&](int t, const Frag& cur, Frag& nxt) {
        mma_half(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#if !defined(GLDS_EXP) || !(GLDS_EXP & 4)
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
#if !defined(GLDS_EXP) || !(GLDS_EXP & 1)
        issue((t + NST - 1) & (NST - 1), t + NST < nsub);        // sub-tile t+3 (or a harmless re-load of the last one)
#endif
#if !defined(GLDS_EXP) || !(GLDS_EXP & 2)
        if (t + 1 < nsub) read_frags(t + 1, nxt);
#endif
        __builtin_amdgcn_sched_barrier(0);
        mma_half(cur, 1);
    };
    int t = 0;
    for (; t + 1 < nsub; t += 2) {
        step(t, f0, f1);
        step(t + 1, f1, f0);
    }
    if (t < nsub) step(t, f0, f1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the trailing re-loads must not outlive the block

    laed_mfma_drain(acc);
    store_acc<BM, BN, WGM, WGN, TM, TN>(p, acc, m0, n0, wm, wn, lrow, lk, split);
    if (TA && csum_wave) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float s_ = fsum[i] + __shfl_xor(fsum[i], 32, 64);          // the two k halves of the wave
            if (lk == 0) {
                const int m = m0 + wm * (BM / WGM) + i * 32 + lrow;
                if (p.splitk > 1) p.ws_colsum[(int64_t)split * p.M + m] = s_;
                else p.colsum[m] = p.colsum_acc ? p.colsum[m] + s_ : s_;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// EXPERIMENT (hooks build only, LAED_GEMM_GLDS=2): the LDS-DMA kernel with a LOADER WAVE.  Eight compute waves execute
// nothing but LDS fragment reads, MFMAs and the barrier -- no VMEM instruction and no vmcnt wait ever enters their
// instruction streams -- and a ninth wave issues every DMA of the block (16 per 16-deep sub-tile: 8 one-KiB pieces per
// operand) three sub-tiles ahead, waits on the counted vmcnt and joins the same barrier.  This separates "a wave that
// issues loads loses MFMA issue" from "the loads' latency reaches the barrier".
template <bool TA, bool TB>
__global__ __launch_bounds__(576, 2) void sgemm_gldsw_kernel(SgemmArgs p) {
    constexpr int BM = 128, BN = 128, NTC = 512, WGM = 2, WGN = 4, TM = 2, TN = 1;
    constexpr int KS = 16, NST = 4;
    constexpr bool AK = !TA, BKC = TB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lk = lane >> 5;

    int nwg = gridDim.x, pid = blockIdx.x;
    int xcd = pid & 7, q = nwg >> 3, r = nwg & 7;
    int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (pid >> 3);
    int split = 0;
    if (p.splitk > 1) { split = wg % p.splitk; wg /= p.splitk; }
    const int kbeg = split * p.kper;
    const int kend = (p.splitk > 1) ? min(p.K, kbeg + p.kper) : p.K;
    const int GROUP_M = p.group_m;
    int in_group = GROUP_M * p.tiles_n;
    int gid = wg / in_group, first_m = gid * GROUP_M;
    int gsz = min(p.tiles_m - first_m, GROUP_M);
    int tile_m = first_m + (wg % in_group) % gsz;
    int tile_n = (wg % in_group) / gsz;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int klen = kend - kbeg;
    if (!((m0 + BM <= p.M) && (n0 + BN <= p.N) && klen >= KS * NST && (klen % KS) == 0)) {     // block-uniform
        if (wave >= NTC / 64) return;              // the loader wave has no part in the register-staged body
        sgemm_block<BM, BN, TA, TB, true, NTC, 0>(p);
        return;
    }
    float* imgA = smem;                         // [NST][BM * KS]
    float* imgB = smem + NST * BM * KS;         // [NST][BN * KS]
    const int nsub = klen / KS;

    if (wave == NTC / 64) {
        // ---- loader wave: piece w of an operand's sub-tile image = rows 16w.. (K-contiguous) or k-rows 2w, 2w+1
        const float* pa;
        const float* pb;
        int64_t pa_t, pb_t, pa_w, pb_w;
        if (AK) {
            const int row = lane >> 2, qs = (lane & 3) ^ ((row >> 2) & 3);        // (16w + row) >> 2 & 3 == row >> 2 & 3
            pa = p.A + (int64_t)(m0 + row) * p.lda + kbeg + 4 * qs; pa_t = KS; pa_w = (int64_t)16 * p.lda;
        } else {
            pa = p.A + (int64_t)(kbeg + (lane >> 5)) * p.lda + m0 + 4 * (lane & 31); pa_t = (int64_t)KS * p.lda; pa_w = (int64_t)2 * p.lda;
        }
        if (BKC) {
            const int row = lane >> 2, qs = (lane & 3) ^ ((row >> 2) & 3);
            pb = p.B + (int64_t)(n0 + row) * p.ldb + kbeg + 4 * qs; pb_t = KS; pb_w = (int64_t)16 * p.ldb;
        } else {
            pb = p.B + (int64_t)(kbeg + (lane >> 5)) * p.ldb + n0 + 4 * (lane & 31); pb_t = (int64_t)KS * p.ldb; pb_w = (int64_t)2 * p.ldb;
        }
        auto issue = [&](int stage, bool advance) {
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                __builtin_amdgcn_global_load_lds(LAED_GPTR(pa + w * pa_w), LAED_LPTR(imgA + stage * BM * KS + w * 256), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(LAED_GPTR(pb + w * pb_w), LAED_LPTR(imgB + stage * BN * KS + w * 256), 16, 0, 0);
            }
            if (advance) { pa += pa_t; pb += pb_t; }
        };
#pragma unroll
        for (int s_ = 0; s_ < NST - 1; ++s_) issue(s_, s_ + 1 < nsub);
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");           // sub-tile 0 landed (1, 2 in flight)
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nsub; ++t) {
            issue((t + NST - 1) & (NST - 1), t + NST < nsub);        // sub-tile t+3 (or a harmless re-load of the last one)
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");       // sub-tile t+1 landed (t+2, t+3 in flight)
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---- compute waves
    const int wm = wave / WGN, wn = wave % WGN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][0][e] = 0.f;
    const int arow = wm * (BM / WGM) + lrow, brow = wn * (BN / WGN) + lrow;
    const int fsw = (lrow >> 2) & 3;
    int aoff[2], boff[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        aoff[g] = AK ? arow * KS + 4 * ((2 * g + lk) ^ fsw) : (4 * (2 * g + lk)) * BM + arow;
        boff[g] = BKC ? brow * KS + 4 * ((2 * g + lk) ^ fsw) : (4 * (2 * g + lk)) * BN + brow;
    }
    const bool do_colsum = TA && (p.colsum != nullptr) && (tile_n == 0);
    const bool csum_wave = do_colsum && wn == 0;
    float fsum[TM] = {0.f, 0.f};
    __builtin_amdgcn_s_barrier();                                    // sub-tile 0 is in LDS
    asm volatile("" ::: "memory");
    for (int t = 0; t < nsub; ++t) {
        const float* a_s = imgA + (t & (NST - 1)) * BM * KS;
        const float* b_s = imgB + (t & (NST - 1)) * BN * KS;
        float fa[2][TM][4], fb[2][4];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (AK) {
                    const float4 v = *reinterpret_cast<const float4*>(a_s + aoff[g] + i * 32 * KS);
                    fa[g][i][0] = v.x; fa[g][i][1] = v.y; fa[g][i][2] = v.z; fa[g][i][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fa[g][i][j] = a_s[aoff[g] + j * BM + i * 32];
                }
            }
            if (BKC) {
                const float4 v = *reinterpret_cast<const float4*>(b_s + boff[g]);
                fb[g][0] = v.x; fb[g][1] = v.y; fb[g][2] = v.z; fb[g][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[g][j] = b_s[boff[g] + j * BN];
            }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    laed_mfma_32x32x2(acc[i][0], fa[g][i][j], fb[g][j]);
        if (TA && csum_wave) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i) fsum[i] += fa[g][i][j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    laed_mfma_drain(acc);
    store_acc<BM, BN, WGM, WGN, TM, TN>(p, acc, m0, n0, wm, wn, lrow, lk, split);
    if (TA && csum_wave) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float s_ = fsum[i] + __shfl_xor(fsum[i], 32, 64);
            if (lk == 0) {
                const int m = m0 + wm * (BM / WGM) + i * 32 + lrow;
                if (p.splitk > 1) p.ws_colsum[(int64_t)split * p.M + m] = s_;
                else p.colsum[m] = p.colsum_acc ? p.colsum[m] + s_ : s_;
            }
        }
    }
}

#endif   // LAED_ENV_HOOKS

// C = sum_s ws[s] (+ bias) (+ C); colsum = sum_s ws_colsum[s] (+ colsum).  Fixed summation order: reproducible.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(SgemmArgs p) {
    const int64_t mn = (int64_t)p.M * p.N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < mn; i += stride) {
        const int64_t row = i / p.N;
        const int col = (int)(i - row * p.N);
        float v = 0.f;
        for (int sidx = 0; sidx < p.splitk; ++sidx) v += p.ws[(int64_t)sidx * mn + i];
        if (p.bias != nullptr) v += p.bias[col];
        float* dst = p.C + row * p.ldc + col;
        if (p.accumulate) v += *dst;
        *dst = v;
    }
    if (p.colsum != nullptr) {
        for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < p.M; m += stride) {
            float v = 0.f;
            for (int sidx = 0; sidx < p.splitk; ++sidx) v += p.ws_colsum[(int64_t)sidx * p.M + m];
            p.colsum[m] = p.colsum_acc ? p.colsum[m] + v : v;
        }
    }
}

// Row tiles per group of the grouped block order (blocks of a group run together on one XCD and share operand strips
// through its L2).  LAED_GEMM_GROUP_M overrides it in the hooks build.
int group_m_for(int tiles_m, int tiles_n) {
    (void)tiles_m; (void)tiles_n;
    static const int forced = []() { const char* e = laed_exp_hook("LAED_GEMM_GROUP_M"); return e ? atoi(e) : 0; }();
    return forced > 0 ? forced : 8;
}

// Phase-lock counters: a ring of slices in a file-scope device array (one slice per launch, zeroed on the launch's stream
// right before it; a slice comes round again after PL_SLICES further phase-locked launches, and every wait on it is
// bounded, so a clash between two streams costs time, never correctness).
constexpr int PL_SLICES = 8, PL_SLICE = 4096;
__device__ __attribute__((aligned(256))) unsigned g_phase_counters[PL_SLICES * PL_SLICE];
std::atomic<unsigned> g_phase_next{0};

int phase_every() {      // K tiles between meetings; 0 = off.  LAED_GEMM_PHASE (hooks build) overrides the default
    static const int v = []() { const char* e = laed_exp_hook("LAED_GEMM_PHASE"); return e ? atoi(e) : 0; }();
    return v;
}

unsigned* phase_slice(hipStream_t stream) {
    static unsigned* base[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (base[dev] == nullptr) {
        unsigned* ptr = nullptr;
        if (hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(g_phase_counters)) != hipSuccess) return nullptr;
        base[dev] = ptr;
    }
    unsigned* sl = base[dev] + (size_t)(g_phase_next.fetch_add(1) % PL_SLICES) * PL_SLICE;
    if (hipMemsetAsync(sl, 0, PL_SLICE * sizeof(unsigned), stream) != hipSuccess) return nullptr;
    return sl;
}

template <int BM, int BN, bool TA, bool TB, bool VEC, int NT = 256, int EPI = 0, int BKT = BK, int PF = 1>
int launch_cfg(const SgemmArgs& a, hipStream_t stream) {
    constexpr int SA = BM + (!TA ? 1 : 4), SB = BN + (TB ? 1 : 4);
    constexpr size_t lds_tiles = (size_t)2 * BKT * (SA + SB) * sizeof(float);
    // the epilogues reuse the tile memory: online-softmax partials [NT/128][BM][2], column sums [NT/(BM/4)][BM/4][4]
    constexpr size_t lds_epi = (size_t)(EPI != 0 ? (NT / 128) * BM * 2 : (TA ? NT * 4 : 0)) * sizeof(float);
    constexpr size_t lds = lds_tiles > lds_epi ? lds_tiles : lds_epi;
    auto kern = sgemm_kernel<BM, BN, TA, TB, VEC, NT, EPI, BKT, PF>;
    static uint64_t attr_set = 0;   // per device
    if (laed_first_use_on_device(&attr_set)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return laed_fail((int)e, "sgemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    SgemmArgs p = a;
    p.tiles_m = (a.M + BM - 1) / BM;
    p.tiles_n = (a.N + BN - 1) / BN;
    p.group_m = group_m_for(p.tiles_m, p.tiles_n);
    int64_t grid = (int64_t)p.tiles_m * p.tiles_n * (p.splitk > 1 ? p.splitk : 1);
    if (grid <= 0 || grid > 0x7fffffff) return laed_fail(LAED_E_SIZE, "sgemm: grid %lld out of range", (long long)grid);
    p.sync = nullptr;
    p.sync_every = 0;
    p.sync_shift = 6;
    if (VEC && EPI == 0 && NT == 512 && BM == 128 && BN == 128 && phase_every() >= 2 && p.splitk <= 1 && a.M % BM == 0 &&
        a.N % BN == 0 && grid >= 1024 && a.K % BKT == 0 && a.K / BKT >= 4 * phase_every() &&
        ((grid >> 3) + 16) / 16 * 8 <= PL_SLICE) {
        static const int shift = []() { const char* e = laed_exp_hook("LAED_GEMM_PHASE_GROUP"); return (e && atoi(e) == 32) ? 5 : (e && atoi(e) == 16) ? 4 : 6; }();
        p.sync_shift = shift;
        p.sync_every = phase_every() & ~1;
        p.sync = phase_slice(stream);          // nullptr (a failed query / memset): the launch simply runs unlocked
    }
    ProfRec r;
    if (g_prof_on.load(std::memory_order_relaxed) && prof_begin(r, 2.0 * a.M * a.N * a.K,
                                                               (TA ? 2 : (TB ? 0 : 1)) * 2 + (BM == 128 ? 0 : 1), stream)) {
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, stream, p);
        prof_end(r, stream);
    } else {
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, stream, p);
    }
    LAED_LAUNCH_CHECK("sgemm");
    return 0;
}

#ifdef LAED_ENV_HOOKS
template <bool TA, bool TB, bool LW = false>
int launch_glds(const SgemmArgs& a, hipStream_t stream) {
    constexpr size_t lds_glds = (size_t)4 * 2 * 128 * 16 * sizeof(float);
    constexpr int SA = 128 + (!TA ? 1 : 4), SB = 128 + (TB ? 1 : 4);
    constexpr size_t lds_reg = (size_t)2 * BK * (SA + SB) * sizeof(float);      // edge blocks: the register-staged body
    constexpr size_t lds = lds_glds > lds_reg ? lds_glds : lds_reg;
    auto kern = LW ? sgemm_gldsw_kernel<TA, TB> : sgemm_glds_kernel<TA, TB>;
    constexpr int nthreads = LW ? 576 : 512;
    static uint64_t attr_set = 0;
    if (laed_first_use_on_device(&attr_set)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return laed_fail((int)e, "sgemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    SgemmArgs p = a;
    p.tiles_m = (a.M + 127) / 128;
    p.tiles_n = (a.N + 127) / 128;
    p.group_m = group_m_for(p.tiles_m, p.tiles_n);
    int64_t grid = (int64_t)p.tiles_m * p.tiles_n * (p.splitk > 1 ? p.splitk : 1);
    if (grid <= 0 || grid > 0x7fffffff) return laed_fail(LAED_E_SIZE, "sgemm: grid %lld out of range", (long long)grid);
    ProfRec r;
    if (g_prof_on.load(std::memory_order_relaxed) && prof_begin(r, 2.0 * a.M * a.N * a.K, (TA ? 2 : (TB ? 0 : 1)) * 2, stream)) {
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(nthreads), lds, stream, p);
        prof_end(r, stream);
    } else {
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(nthreads), lds, stream, p);
    }
    LAED_LAUNCH_CHECK("sgemm_glds");
    return 0;
}

int glds_mode() {       // 0 off, 1 LDS-DMA kernel, 2 LDS-DMA kernel with a loader wave
    static const int v = []() { const char* e = laed_exp_hook("LAED_GEMM_GLDS"); return e != nullptr ? atoi(e) : 0; }();
    return v;
}
#endif

bool bk16_on() {        // LAED_GEMM_BK16=0 (hooks build) keeps the 32-deep K tile everywhere
    static const bool v = []() { const char* e = laed_exp_hook("LAED_GEMM_BK16"); return !(e != nullptr && atoi(e) == 0); }();
    return v;
}

template <bool TA, bool TB>
int launch_layout(SgemmArgs a, bool vec, int64_t ws_floats, bool raw_partials, int* splits_out, int force_split,
                  hipStream_t stream) {
    const int64_t t128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128);
    const int64_t t64 = (int64_t)((a.M + 63) / 64) * ((a.N + 63) / 64);
    const int nkt = (a.K + BK - 1) / BK;
    const int gemm_mode = laed_get_gemm_mode();
    // opt-in split-bf16 math for the K-contiguous (NT) layout, 128x128 tiles (xgemm.hip).  An NT call that asks for
    // sums over K of A (colsum) can only be served by that kernel and forces it.
    const bool force_x = !TA && TB && a.colsum != nullptr;
    const bool use_x = gemm_mode == 1 && !TA && TB && (force_x || t128 >= g_xgemm_min_tiles);
    // Tile and split-K choice.  Problems of two rounds or more of the chip's 512 resident 128x128 blocks run as they
    // are (128x128, no split).  Below that a small cost model, fitted to tools/gemm_sweep.py on the mid-size products
    // of the reference's own batch sizes, prices every (tile, S):
    //   time = R(n) * u * (K tiles per split + 4) + reduce(S),  n = blocks on the busiest of the 256 CUs,
    //   a CU runs 2 blocks of 128x128 (u = 4.13 us per K tile) or 4 of 64x64 (u = 2.66 us per K tile of the four) at
    //   once; a partly filled CU runs its blocks faster (R128(1) = 0.62, R64(1..3) = 0.27, 0.5, 0.75 of a full round);
    //   a block pays ~4 K tiles of prologue/epilogue; S > 1 adds the fixed-order reduce ((S+1) M N floats at ~3 TB/s).
    // Splitting needs caller workspace; the partial tiles are summed in a fixed order (reproducible).
    bool big = true;
    int S = 1;
    if (!use_x && t128 < 1024) {
        const int64_t per = (int64_t)a.M * a.N + (a.colsum ? a.M : 0);
        const int64_t cap64 = a.ws != nullptr ? ws_floats / (per > 0 ? per : 1) : 1;
        int smax = (int)(cap64 < 32 ? cap64 : 32);
        if (smax > nkt / 4) smax = nkt / 4;               // keep >= 4 K tiles per split
        if (smax < 1) smax = 1;
        const float mn_mb = (float)per * 4.0e-6f;         // MB per partial product
        float best = 1e30f;
        for (int tile = 128; tile >= 64; tile -= 64) {
            const int64_t tiles = tile == 128 ? t128 : t64;
            for (int s = 1; s <= smax; ++s) {
                const int64_t n = (tiles * s + 255) / 256;
                float rounds;
                if (tile == 128) rounds = (float)(n / 2) + ((n & 1) ? 0.62f : 0.f);
                else { static const float part[4] = {0.f, 0.27f, 0.5f, 0.75f}; rounds = (float)(n / 4) + part[n & 3]; }
                const int kt = (nkt + s - 1) / s;
                float t = rounds * (tile == 128 ? 4.13f : 2.66f) * (float)(kt + 4);
                if (s > 1) t += 4.0f + (float)(s + 1) * mn_mb / 3.0f;
                if (t < best - 0.01f) { best = t; big = tile == 128; S = s; }
            }
        }
    } else if (use_x && a.ws != nullptr && t128 < 512) {
        // split-bf16 kernel (128x128 only): the split whose grid fills whole rounds of the 512 block slots
        const int64_t per = (int64_t)a.M * a.N + (a.colsum ? a.M : 0);
        const int64_t cap64 = ws_floats / (per > 0 ? per : 1);
        int smax = (int)(cap64 < 32 ? cap64 : 32);
        if (smax > nkt / 8) smax = nkt / 8;
        float best = -1.f;
        for (int s = 1; s <= smax; ++s) {
            const int64_t blocks = t128 * s;
            const int64_t rounds = (blocks + 511) / 512;
            const float eff = (float)blocks / (float)(rounds * 512) - 0.004f * (float)s;
            if (eff > best + 1e-6f) { best = eff; S = s; }
        }
    }
    if (force_split > 1 && a.ws != nullptr) {       // tail slab of a row-split problem: 128x128 tiles, K split as told
        big = true;
        S = force_split;
    }
    {   // experiment hook (tools/gemm_sweep.py): LAED_GEMM_FORCE="<tile>,<S>" overrides the choice above
        static const char* fe = laed_exp_hook("LAED_GEMM_FORCE");
        if (fe != nullptr && !use_x) {
            int ft = 0, fs = 0;
            if (sscanf(fe, "%d,%d", &ft, &fs) == 2 && (ft == 64 || ft == 128) && fs >= 1) {
                const int64_t per = (int64_t)a.M * a.N + (a.colsum ? a.M : 0);
                if (fs > 1 && (a.ws == nullptr || ws_floats / (per > 0 ? per : 1) < fs)) fs = 1;
                if (fs > nkt) fs = nkt;
                big = ft == 128;
                S = fs;
            }
        }
    }
    a.splitk = S;
    a.kper = ((nkt + S - 1) / S) * BK;
    if (S > 1) {
        // every split must own at least one K element
        while (S > 1 && (int64_t)(S - 1) * a.kper >= a.K) { --S; }
        a.splitk = S;
        a.ws_colsum = a.ws + (int64_t)S * a.M * a.N;
    }
    if (splits_out) *splits_out = S;
    int st;
    if (use_x) {
        ProfRec r;
        // read the switch ONCE: a laed_prof_enable() between the two tests must not push a record without events
        const bool prof = g_prof_on.load(std::memory_order_relaxed) && prof_begin(r, 2.0 * a.M * a.N * a.K, 6, stream);
        st = laed_xgemm_nt_launch(a.M, a.N, a.K, a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.bias, a.accumulate != 0,
                                  a.splitk, a.kper, a.ws, vec, S > 1 ? a.ws_colsum : a.colsum, a.colsum_acc != 0, stream);
        if (prof) prof_end(r, stream);
    }
#ifdef LAED_ENV_HOOKS
    else if (big && vec && S == 1 && t128 >= 2048 && laed_exp_hook("LAED_GEMM_TILE256") != nullptr)
        st = launch_cfg<256, 256, TA, TB, true, 1024>(a, stream);      // experiment: 16 waves, 256x256 tile, one block per CU
#endif
#ifdef LAED_ENV_HOOKS
    else if (big && vec && glds_mode() == 1 && a.K % 16 == 0 && big_tile_threads() == 512) st = launch_glds<TA, TB>(a, stream);
    else if (big && vec && glds_mode() == 2 && a.K % 16 == 0 && big_tile_threads() == 512) st = launch_glds<TA, TB, true>(a, stream);
#endif
    // K tile depth of the 8-wave 128x128 kernel: 16 for the x W^T and x W layouts (logits +2 %, dh +1.3 % at the
    // projection shapes: a shorter prologue and a finer interleave of staging and MFMAs -- still two blocks per CU, and
    // forcing three with __launch_bounds__(512, 6) is slower), 32 for A^T B (dW: -1.3 % with 16 in the compiler's VGPR
    // form, +-0 with AccVGPR accumulators) -- tools/gemm_exp.py, DESIGN.md section 4
    // x W (dh) additionally keeps two K tiles in flight in registers (+2.8 %; logits +0.4 %, dW +-0: not worth their registers)
    else if (big && vec && big_tile_threads() == 512 && !TA && !TB && bk16_on()) st = launch_cfg<128, 128, TA, TB, true, 512, 0, 16, 2>(a, stream);
    else if (big && vec && big_tile_threads() == 512 && !TA && bk16_on()) st = launch_cfg<128, 128, TA, TB, true, 512, 0, 16>(a, stream);
    else if (big && vec && big_tile_threads() == 512) st = launch_cfg<128, 128, TA, TB, true, 512>(a, stream);
    else if (big) st = vec ? launch_cfg<128, 128, TA, TB, true>(a, stream) : launch_cfg<128, 128, TA, TB, false>(a, stream);
    else st = vec ? launch_cfg<64, 64, TA, TB, true>(a, stream) : launch_cfg<64, 64, TA, TB, false>(a, stream);
    if (st != 0 || S == 1 || raw_partials) return st;
    const int64_t mn = (int64_t)a.M * a.N;
    const int nb = (int)((mn + 255) / 256 < 2048 ? (mn + 255) / 256 : 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb), dim3(256), 0, stream, a);
    LAED_LAUNCH_CHECK("splitk_reduce");
    return 0;
}

}   // namespace

static int sgemm_dispatch(bool transA, bool transB, int M, int N, int K,
                          const float* A, int64_t lda, const float* B, int64_t ldb,
                          float* C, int64_t ldc, const float* bias, bool accumulate,
                          float* colsum_a, bool colsum_accumulate,
                          float* ws, int64_t ws_floats, bool raw_partials, int* splits_out, int force_split, hipStream_t stream);

int laed_sgemm_splitk_impl(bool transA, bool transB, int M, int N, int K,
                           const float* A, int64_t lda, const float* B, int64_t ldb,
                           float* C, int64_t ldc, const float* bias, bool accumulate,
                           float* colsum_a, bool colsum_accumulate,
                           float* ws, int64_t ws_floats, bool raw_partials, int* splits_out, hipStream_t stream) {
    // Tail split.  A launch of t 128x128 tiles runs in ceil(t / 512) rounds of the chip's 512 resident block slots
    // (2 per CU), and a short last round costs a whole one: the projection's dW product (391 x 8 = 3128 tiles = 6.1
    // rounds) spent 1/7 of its time on 56 tiles.  With workspace at hand the rows are cut so that the first launch
    // fills whole rounds and the remaining row tiles run as ONE more round with K split S ways (1/S as long),
    // summed in a fixed order like every other split-K product here.
    static const bool tail_on = []() { const char* e = laed_exp_hook("LAED_GEMM_TAILSPLIT"); return !(e != nullptr && atoi(e) == 0); }();
    if (tail_on && ws != nullptr && !raw_partials && M > 0 && N > 0 && K > 0 && A && B && C && !(transA && transB) &&
        laed_get_gemm_mode() == 0) {
        const int64_t tm = (M + 127) / 128, tn = (N + 127) / 128, t = tm * tn;
        const int nkt = (K + BK - 1) / BK;
        if (t > 512 && nkt >= 16) {
            const int64_t full = t / 512;
            int64_t tm1 = full * 512 / tn;
            if (tm1 > tm) tm1 = tm;
            const int64_t tm2 = tm - tm1, t2 = tm2 * tn;
            if (tm1 > 0 && tm2 > 0 && t2 <= 256) {
                const int M1 = (int)(tm1 * 128), M2 = M - M1;
                int64_t S = 512 / t2;
                if (S > nkt / 8) S = nkt / 8;
                const int64_t cap = ws_floats / ((int64_t)M2 * N + M2);
                if (S > cap) S = cap;
                if (S > 16) S = 16;
                if (S >= 2) {
                    if (splits_out) *splits_out = 1;
                    LAED_TRY(sgemm_dispatch(transA, transB, M1, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, colsum_a,
                                            colsum_accumulate, nullptr, 0, false, nullptr, 0, stream));
                    const float* A2 = transA ? A + M1 : A + (int64_t)M1 * lda;
                    return sgemm_dispatch(transA, transB, M2, N, K, A2, lda, B, ldb, C + (int64_t)M1 * ldc, ldc, bias, accumulate,
                                          colsum_a ? colsum_a + M1 : nullptr, colsum_accumulate, ws, ws_floats, false, nullptr,
                                          (int)S, stream);
                }
            }
        }
    }
    return sgemm_dispatch(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, colsum_a, colsum_accumulate,
                          ws, ws_floats, raw_partials, splits_out, 0, stream);
}

static int sgemm_dispatch(bool transA, bool transB, int M, int N, int K,
                          const float* A, int64_t lda, const float* B, int64_t ldb,
                          float* C, int64_t ldc, const float* bias, bool accumulate,
                          float* colsum_a, bool colsum_accumulate,
                          float* ws, int64_t ws_floats, bool raw_partials, int* splits_out, int force_split, hipStream_t stream) {
    LAED_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "sgemm: negative dimension");
    if (splits_out) *splits_out = 1;
    if (M == 0 || N == 0) return 0;
    LAED_CHECK_ARG(A && B && C, "sgemm: null matrix pointer");
    LAED_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "sgemm: leading dimension too small");
    LAED_CHECK_ARG(!(transA && transB), "sgemm: transA && transB not supported");
    LAED_CHECK_ARG(colsum_a == nullptr || transA || (transB && laed_get_gemm_mode() == 1),
                   "sgemm: sums over K of A need transA (or the x W^T layout in the bf16x3 mode)");
    SgemmArgs a;
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.colsum = colsum_a;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K;
    a.accumulate = accumulate ? 1 : 0;
    a.colsum_acc = colsum_accumulate ? 1 : 0;
    a.tiles_m = a.tiles_n = 0; a.group_m = 8;
    a.splitk = 1; a.kper = 0; a.ws = ws; a.ws_colsum = nullptr;
    a.labels = nullptr; a.part = nullptr; a.lab_logit = nullptr; a.sync = nullptr; a.sync_every = 0; a.sync_shift = 6;
    const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0);
    if (!transA && transB) return launch_layout<false, true>(a, vec, ws_floats, raw_partials, splits_out, force_split, stream);
    if (!transA && !transB) return launch_layout<false, false>(a, vec, ws_floats, raw_partials, splits_out, force_split, stream);
    return launch_layout<true, false>(a, vec, ws_floats, raw_partials, splits_out, force_split, stream);
}

// Forward-only projection statistics: for x[M][K] W[N][K]^T + bias, per 128-column tile the (max, sum exp) of every row
// and the logit at labels[m]; the [M][N] product itself is never written.  part: [ceil(N/128)][M][2] floats.
int laed_sgemm_softmax_stats_impl(int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                                  const float* bias, const int64_t* labels, float* part, float* lab_logit, hipStream_t stream) {
    LAED_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && labels && part && lab_logit, "sgemm_softmax_stats: bad argument");
    LAED_CHECK_ARG(lda >= K && ldb >= K, "sgemm_softmax_stats: leading dimension too small");
    SgemmArgs a;
    a.A = A; a.B = B; a.C = nullptr; a.bias = bias; a.colsum = nullptr;
    a.lda = lda; a.ldb = ldb; a.ldc = 0;
    a.M = M; a.N = N; a.K = K;
    a.accumulate = 0; a.colsum_acc = 0;
    a.tiles_m = a.tiles_n = 0; a.group_m = 8;
    a.splitk = 1; a.kper = ((K + BK - 1) / BK) * BK; a.ws = nullptr; a.ws_colsum = nullptr;
    a.labels = labels; a.part = part; a.lab_logit = lab_logit; a.sync = nullptr; a.sync_every = 0; a.sync_shift = 6;
    const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0);
    return vec ? launch_cfg<128, 128, false, true, true, 512, 1, 16>(a, stream)
               : launch_cfg<128, 128, false, true, false, 512, 1>(a, stream);
}

// Greedy symbol selection without the logits: for x[M][K] W[N][K]^T + bias, per 128-column tile the largest value of
// every row and its column (lowest column on ties); part: [ceil(N/128)][M][2] floats (value, column index as bits),
// reduced over the tiles by argmax_tiles_kernel.  The [M][N] product is never written.
namespace {
__global__ __launch_bounds__(256) void argmax_tiles_kernel(const float* __restrict__ part, int M, int tiles, int64_t* __restrict__ ids) {
    // one wave per row: the lanes stride over the column tiles, then meet in a shuffle tree (largest value, lowest column)
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int t = lane; t < tiles; t += 64) {    // ascending columns per lane: strict > keeps the lane's first maximum
        const float2 r = *reinterpret_cast<const float2*>(part + ((int64_t)t * M + m) * 2);
        if (r.x > bv) { bv = r.x; bi = __float_as_int(r.y); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) ids[m] = bi == 0x7fffffff ? 0 : bi;        // a row of NaN / -inf only: 0, as laed_argmax_rows
}
}   // namespace

int laed_sgemm_argmax_impl(int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                           const float* bias, float* part, int64_t* ids, hipStream_t stream) {
    LAED_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && part && ids, "sgemm_argmax: bad argument");
    LAED_CHECK_ARG(lda >= K && ldb >= K, "sgemm_argmax: leading dimension too small");
    SgemmArgs a;
    a.A = A; a.B = B; a.C = nullptr; a.bias = bias; a.colsum = nullptr;
    a.lda = lda; a.ldb = ldb; a.ldc = 0;
    a.M = M; a.N = N; a.K = K;
    a.accumulate = 0; a.colsum_acc = 0;
    a.tiles_m = a.tiles_n = 0; a.group_m = 8;
    a.splitk = 1; a.kper = ((K + BK - 1) / BK) * BK; a.ws = nullptr; a.ws_colsum = nullptr;
    a.labels = nullptr; a.part = part; a.lab_logit = nullptr; a.sync = nullptr; a.sync_every = 0; a.sync_shift = 6;
    const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0);
    LAED_TRY((vec ? launch_cfg<128, 128, false, true, true, 512, 2, 16>(a, stream)
                  : launch_cfg<128, 128, false, true, false, 512, 2>(a, stream)));
    hipLaunchKernelGGL(argmax_tiles_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, part, M, (N + 127) / 128, ids);
    LAED_LAUNCH_CHECK("argmax_tiles");
    return 0;
}

// NT product with row sums of A (bf16x3 mode only: the split-bf16 kernel sums its own A-operand registers).
int laed_sgemm_rowsum_impl(int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                           float* C, int64_t ldc, bool accumulate, float* rowsum, bool rowsum_acc,
                           float* ws, int64_t ws_floats, hipStream_t stream) {
    LAED_CHECK_ARG(rowsum != nullptr && laed_get_gemm_mode() == 1, "sgemm_rowsum: only available in the bf16x3 mode");
    return laed_sgemm_splitk_impl(false, true, M, N, K, A, lda, B, ldb, C, ldc, nullptr, accumulate, rowsum, rowsum_acc,
                                  ws, ws_floats, false, nullptr, stream);
}

int laed_sgemm_impl(bool transA, bool transB, int M, int N, int K,
                    const float* A, int64_t lda, const float* B, int64_t ldb,
                    float* C, int64_t ldc, const float* bias, bool accumulate,
                    float* colsum_a, bool colsum_accumulate, hipStream_t stream) {
    return laed_sgemm_splitk_impl(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, colsum_a,
                                  colsum_accumulate, nullptr, 0, false, nullptr, stream);
}

extern "C" int laed_sgemm(int transA, int transB, int M, int N, int K,
                          const float* A, int lda, const float* B, int ldb,
                          float* C, int ldc, const float* bias, int accumulate,
                          float* colsum_a, int colsum_accumulate, void* stream) {
    return laed_sgemm_impl(transA != 0, transB != 0, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate != 0,
                           colsum_a, colsum_accumulate != 0, (hipStream_t)stream);
}

extern "C" int laed_sgemm_ws(int transA, int transB, int M, int N, int K,
                             const float* A, int lda, const float* B, int ldb,
                             float* C, int ldc, const float* bias, int accumulate,
                             float* colsum_a, int colsum_accumulate, float* ws, int64_t ws_floats, void* stream) {
    return laed_sgemm_splitk_impl(transA != 0, transB != 0, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate != 0,
                                  colsum_a, colsum_accumulate != 0, ws, ws ? ws_floats : 0, false, nullptr,
                                  (hipStream_t)stream);
}

extern "C" int laed_set_xgemm_min_tiles(int n) {   // test hook: let small problems take the split-bf16 kernel too
    LAED_CHECK_ARG(n >= 1, "set_xgemm_min_tiles: n >= 1");
    g_xgemm_min_tiles = n;
    return 0;
}

extern "C" int laed_set_gemm_mode(int mode) {
    LAED_CHECK_ARG(mode == 0 || mode == 1, "set_gemm_mode: 0 = exact fp32 MFMA, 1 = split-bf16 (bf16x3)");
    g_gemm_mode = mode;
    return 0;
}
extern "C" int laed_get_gemm_mode(void) {
    int m = g_gemm_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("LAED_GEMM_MODE");
        m = (e != nullptr && (strcmp(e, "bf16x3") == 0 || strcmp(e, "1") == 0)) ? 1 : 0;
        int expected = -1;
        if (!g_gemm_mode.compare_exchange_strong(expected, m)) m = expected;     // a concurrent laed_set_gemm_mode wins
    }
    return m;
}

// ---- profiler C ABI (measurement aid; not part of the compute path)
extern "C" int laed_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return 0;
}

// Pre-creates events so that a timed region does not pay hipEventCreate (2 per GEMM launch) inside it.
extern "C" int laed_prof_reserve(int n_events) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    while ((int)g_prof_pool.size() < n_events) {
        hipEvent_t e;
        hipError_t st = hipEventCreate(&e);
        if (st != hipSuccess) return laed_fail((int)st, "prof_reserve: %s", hipGetErrorString(st));
        g_prof_pool.push_back(e);
    }
    return 0;
}

// Waits for every recorded launch, then fills out[variant*3 + {0,1,2}] = {launches, total ms, total flops} for the
// 6 fp32 GEMM variants (NT, NN, TN) x (128x128, 64x64) + variant 6 = split-bf16 NT 128x128, and clears the records.
// `out` is a HOST array of 21 doubles.
extern "C" int laed_prof_collect(double* out) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < 21; ++i) out[i] = 0.0;
    for (auto& r : g_prof_recs) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(r.stop);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.start, r.stop);
        if (e != hipSuccess) return laed_fail((int)e, "prof_collect: %s", hipGetErrorString(e));
        out[r.variant * 3 + 0] += 1.0;
        out[r.variant * 3 + 1] += ms;
        out[r.variant * 3 + 2] += r.flops;
        g_prof_pool.push_back(r.start);
        g_prof_pool.push_back(r.stop);
    }
    g_prof_recs.clear();
    return 0;
}
