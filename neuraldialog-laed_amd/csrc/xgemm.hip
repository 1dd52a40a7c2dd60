// Split-bf16 ("bf16x3") GEMM on the CDNA4 bf16 matrix cores, fp32 in / fp32 out:
//
//   C[M,N] = A[M,K] * B[N,K]^T (+ bias[N]) (+ C)        both operands K-contiguous fp32 ("NT")
//
// Each fp32 operand element x is split while its tile is staged into LDS:  x = hi + lo,  hi = bf16(x) (RNE),
// lo = bf16(x - hi), and the product is accumulated in fp32 as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  on
// v_mfma_f32_32x32x16_bf16 (16x the rate of the fp32 MFMA, 3 instructions per k16 step instead of 8).  The dropped
// a_lo*b_lo term is 2^-16 relative per product, so a K=1024 dot product is accurate to ~7e-6 absolute at O(1)
// magnitude against ~1e-6 for the exact-fp32 kernel (measured in tests/test_kernels_gpu.py::test_xgemm*).
// This is an OPT-IN math mode (laed_set_gemm_mode / LAED_GEMM_MODE=bf16x3): the default everywhere is the
// exact-fp32 kernel in sgemm.hip.
//
// Geometry: block 256 threads = 4 waves (2x2), block tile 128x128, wave tile 64x64 (2x2 MFMA tiles), K step 32.
// LDS image per operand and buffer: [plane hi|lo][k/8][SR = 130 row slots][8 bf16 = 16 B]: a lane's MFMA fragment
// (row r, 8 consecutive k) is one 16-byte slot and the 32 rows of an MFMA tile are 32 consecutive slots, so every
// fragment read is a conflict-free ds_read_b128 with no swizzle; SR*16 = 32 (mod 128) makes the 8-byte staging
// writes of 16 consecutive lanes (2 rows x 4 k-blocks x 2 halves) tile one 128-byte bank row exactly.
// Main loop as in sgemm.hip: straight-line K tiles, global loads / fragment reads / split+store placed one by one
// in the issue gaps between MFMAs, one barrier per K tile, LDS double buffered; guarded loader for edge tiles
// and the K tail; XCD-aware grouped tile order; optional split-K partials.
#include "laed_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
constexpr int KB = BK / 8;            // k blocks of 8 per K tile
constexpr int SR = 130;               // row slots per k block (128 + 2 pad): SR*16 % 128 == 32
constexpr int PLANE = KB * SR;        // 16-byte slots per plane
constexpr int OPER = 2 * PLANE;       // hi + lo planes
constexpr int NL = 4;                 // float4 per thread per operand per K tile

__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
    const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
    const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
    const f32x2 fa = __builtin_convertvector(ha, f32x2), fb = __builtin_convertvector(hb, f32x2);
    const bf16x2 la = __builtin_convertvector(a - fa, bf16x2), lb = __builtin_convertvector(b - fb, bf16x2);
    hi.x = __builtin_bit_cast(unsigned, ha);
    hi.y = __builtin_bit_cast(unsigned, hb);
    lo.x = __builtin_bit_cast(unsigned, la);
    lo.y = __builtin_bit_cast(unsigned, lb);
}

// guarded load of a (128 rows) x (32 k) fp32 tile: thread (row = idx/8, kq = idx%8) takes 4 consecutive k
template <bool VEC>
__device__ __forceinline__ void tile_load(const float* __restrict__ P, int64_t ld, int r0, int k0, int R, int K,
                                          int tid, float4 (&reg)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = tid + i * NT;
        const int gr = r0 + idx / 8, gk = k0 + 4 * (idx % 8);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
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
        reg[i] = v;
    }
}

struct XArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int64_t lda, ldb, ldc;
    int M, N, K;
    int accumulate;
    int tiles_m, tiles_n;
    int splitk, kper;
    float* ws;
    float* rowsum;      // optional: rowsum[m] (+)= sum_k A[m][k]  (tile_n == 0 blocks; split-K: partials [S][M])
    int rowsum_acc;
};

template <bool VEC>
__global__ __launch_bounds__(NT, 2) void xgemm_kernel(XArgs p) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];   // [2 bufs][A | B][hi | lo][KB][SR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 31, lk = lane >> 5;

    int nwg = gridDim.x, pid = blockIdx.x;
    int xcd = pid & 7, q = nwg >> 3, r = nwg & 7;
    int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (pid >> 3);
    int split = 0;
    if (p.splitk > 1) { split = wg % p.splitk; wg /= p.splitk; }
    const int kbeg = split * p.kper;
    const int kend = (p.splitk > 1) ? min(p.K, kbeg + p.kper) : p.K;
    constexpr int GROUP_M = 8;
    const int in_group = GROUP_M * p.tiles_n;
    const int gid = wg / in_group, first_m = gid * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int tile_m = first_m + (wg % in_group) % gsz;
    const int tile_n = (wg % in_group) / gsz;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[NL], rb[NL];
    float rsum[NL] = {0.f, 0.f, 0.f, 0.f};
    const bool do_rowsum = (p.rowsum != nullptr) && (tile_n == 0);
    auto add_rowsum = [&]() {
        if (do_rowsum) {
#pragma unroll
            for (int i = 0; i < NL; ++i) rsum[i] += (ra[i].x + ra[i].y) + (ra[i].z + ra[i].w);
        }
    };
    const int nkt = (kend - kbeg + BK - 1) / BK;
    const int nfull = (kend - kbeg) / BK;
    const bool interior = VEC && (m0 + BM <= p.M) && (n0 + BN <= p.N);
    const bool fast = interior && nfull > 0;

    // staging: thread (row = tid/8 + 32*i, kq = tid%8): 8-byte slot half (kq&1) of k block (kq>>1)
    const int st_off = ((tid % 8) >> 1) * SR * 16 + (tid / 8) * 16 + ((tid % 8) & 1) * 8;   // bytes within a plane
    char* const lds = reinterpret_cast<char*>(smem);
    auto store_unit = [&](int u, int buf) {   // u in [0, 2*NL): A units then B units
        const bool isb = u >= NL;
        const float4 v = isb ? rb[u - NL] : ra[u];
        uint2 hi, lo;
        split4(v, hi, lo);
        char* d = lds + ((buf * 2 + (isb ? 1 : 0)) * OPER) * 16 + st_off + (isb ? u - NL : u) * 32 * 16;
        *reinterpret_cast<uint2*>(d) = hi;
        *reinterpret_cast<uint2*>(d + PLANE * 16) = lo;
    };
    const float* ga = p.A + (int64_t)(m0 + tid / 8) * p.lda + kbeg + 4 * (tid % 8);
    const float* gb = p.B + (int64_t)(n0 + tid / 8) * p.ldb + kbeg + 4 * (tid % 8);
    const int64_t a_ist = 32 * p.lda, b_ist = 32 * p.ldb;
    auto load_unit = [&](int u) {
        if (u < NL) ra[u] = *reinterpret_cast<const float4*>(ga + u * a_ist);
        else rb[u - NL] = *reinterpret_cast<const float4*>(gb + (u - NL) * b_ist);
    };

    if (fast) {
#pragma unroll
        for (int u = 0; u < 2 * NL; ++u) load_unit(u);
    } else {
        tile_load<VEC>(p.A, p.lda, m0, kbeg, p.M, kend, tid, ra);
        tile_load<VEC>(p.B, p.ldb, n0, kbeg, p.N, kend, tid, rb);
    }
    add_rowsum();
#pragma unroll
    for (int u = 0; u < 2 * NL; ++u) store_unit(u, 0);
    __syncthreads();

    // fragment slot of this lane: k block (2*s + lk), row slot (w*64 + t*32 + lrow)
    const int fa_off = (lk * SR + wm * 64 + lrow) * 16;
    const int fb_off = (lk * SR + wn * 64 + lrow) * 16;
    auto read_frags = [&](int buf, int s, uint4 (&fa)[2][2], uint4 (&fb)[2][2]) {
        const char* a_s = lds + (buf * 2 * OPER) * 16 + fa_off + 2 * s * SR * 16;
        const char* b_s = lds + ((buf * 2 + 1) * OPER) * 16 + fb_off + 2 * s * SR * 16;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                fa[t][pl] = *reinterpret_cast<const uint4*>(a_s + (pl * PLANE + t * 32) * 16);
                fb[t][pl] = *reinterpret_cast<const uint4*>(b_s + (pl * PLANE + t * 32) * 16);
            }
        }
    };
#define XMFMA(AF, BF, ACC) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, AF), __builtin_bit_cast(bf16x8, BF), ACC, 0, 0, 0)

    int cur = 0, kt = 0;
    if (fast) {
        constexpr int Q = 24;   // MFMAs per K tile per wave
        for (; kt + 1 < nfull; ++kt) {
            ga += BK;
            gb += BK;
            uint4 fa[2][2][2], fb[2][2][2];   // [k16 step parity][tile][plane]
            read_frags(cur, 0, fa[0], fb[0]);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const int q_ = ((s * 2 + i) * 2 + j) * 3 + c;
                            if (c == 0) XMFMA(fa[s][i][1], fb[s][j][0], acc[i][j]);        // lo * hi
                            else if (c == 1) XMFMA(fa[s][i][0], fb[s][j][1], acc[i][j]);   // hi * lo
                            else XMFMA(fa[s][i][0], fb[s][j][0], acc[i][j]);               // hi * hi
                            if (s == 0 && q_ == 1) read_frags(cur, 1, fa[1], fb[1]);
                            if (q_ < 2 * NL) load_unit(q_);                                 // gaps 0..7: HBM loads
                            if (q_ == Q - 2 * NL - 2) add_rowsum();
                            if (q_ >= Q - 2 * NL - 2 && q_ < Q - 2) store_unit(q_ - (Q - 2 * NL - 2), cur ^ 1);   // gaps 14..21
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    for (; kt < nkt; ++kt) {
        const bool more = (kt + 1 < nkt);
        if (more) {
            tile_load<VEC>(p.A, p.lda, m0, kbeg + (kt + 1) * BK, p.M, kend, tid, ra);
            tile_load<VEC>(p.B, p.ldb, n0, kbeg + (kt + 1) * BK, p.N, kend, tid, rb);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 fa[2][2], fb[2][2];
            read_frags(cur, s, fa, fb);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    XMFMA(fa[i][1], fb[j][0], acc[i][j]);
                    XMFMA(fa[i][0], fb[j][1], acc[i][j]);
                    XMFMA(fa[i][0], fb[j][0], acc[i][j]);
                }
        }
        if (more) {
            add_rowsum();
#pragma unroll
            for (int u = 0; u < 2 * NL; ++u) store_unit(u, cur ^ 1);
        }
        __syncthreads();
        cur ^= 1;
    }
#undef XMFMA

    // epilogue: C/D map  col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + lrow;
            const float bv = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
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
    if (do_rowsum) {   // block-uniform; the 8 lanes tid%8 = 0..7 of a row are adjacent lanes of one wave
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            float v = rsum[i];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            const int row = m0 + tid / 8 + 32 * i;
            if ((tid & 7) == 0 && row < p.M) {
                if (p.splitk > 1) p.rowsum[(int64_t)split * p.M + row] = v;          // partial: summed by the reduce kernel
                else p.rowsum[row] = p.rowsum_acc ? p.rowsum[row] + v : v;
            }
        }
    }
}

}   // namespace

// Launches the split-bf16 kernel for an NT problem.  Split-K partials (splitk > 1) are written to ws[S][M][N]; the
// caller (sgemm.hip) runs the reduce kernel.
int laed_xgemm_nt_launch(int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                         float* C, int64_t ldc, const float* bias, bool accumulate,
                         int splitk, int kper, float* ws, bool vec, float* rowsum_a, bool rowsum_acc, hipStream_t stream) {
    constexpr size_t lds = (size_t)2 * 2 * OPER * 16;
    static uint64_t attr_set[2] = {0, 0};   // per device
    const void* kern = vec ? reinterpret_cast<const void*>(xgemm_kernel<true>) : reinterpret_cast<const void*>(xgemm_kernel<false>);
    if (laed_first_use_on_device(&attr_set[vec ? 1 : 0])) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return laed_fail((int)e, "xgemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    XArgs p;
    p.A = A; p.B = B; p.C = C; p.bias = bias;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.accumulate = accumulate ? 1 : 0;
    p.tiles_m = (M + BM - 1) / BM;
    p.tiles_n = (N + BN - 1) / BN;
    p.splitk = splitk; p.kper = kper; p.ws = ws;
    p.rowsum = rowsum_a; p.rowsum_acc = rowsum_acc ? 1 : 0;

    const int64_t grid = (int64_t)p.tiles_m * p.tiles_n * (splitk > 1 ? splitk : 1);
    if (grid <= 0 || grid > 0x7fffffff) return laed_fail(LAED_E_SIZE, "xgemm: grid %lld out of range", (long long)grid);
    if (vec) hipLaunchKernelGGL(xgemm_kernel<true>, dim3((unsigned)grid), dim3(NT), lds, stream, p);
    else hipLaunchKernelGGL(xgemm_kernel<false>, dim3((unsigned)grid), dim3(NT), lds, stream, p);
    LAED_LAUNCH_CHECK("xgemm");
    return 0;
}
