// Embedding row gather / scatter-add with optional input dropout, and the counter-based RNG fills.
// HBM-bound: one float4 per lane, rows contiguous (E floats).
// scatter-add (the dense embedding gradient), with caller workspace: DETERMINISTIC, no float atomics --
//   rows <= 4096 : one block per row; the block of an id's FIRST occurrence sums all occurrences in position order;
//   larger       : stable radix sort of (id, position) (rocPRIM), then chunks of 64 sorted entries: runs of one id
//                  that lie inside a chunk are summed and written by that chunk alone, the first / last run of
//                  every chunk go to a partial list that a second pass sums per id in chunk order.
//   Under Zipf ids (the hot tokens occur ~10^4 times per step) this is also ~4x faster than the atomics.
// Without workspace: no-return fp32 atomics (global_atomic_add_f32), order-dependent rounding.
#include "laed_common.h"
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <stdlib.h>

namespace {

template <bool VEC>
__global__ __launch_bounds__(256) void embed_gather_kernel(const int64_t* __restrict__ ids, int64_t rows,
                                                           const float* __restrict__ table, int V, int E,
                                                           const uint8_t* __restrict__ keep, float scale,
                                                           float* __restrict__ out) {
    constexpr int VW = VEC ? 4 : 1;
    const int epr = E / VW;   // work items per row
    const int64_t n = rows * epr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / epr;
        const int c = (int)(i - r * epr) * VW;
        int64_t id = ids[r];
        if (id < 0 || id >= V) id = 0;   // defensive: out-of-range ids read the PAD row
        if (VEC) {
            float4 v = *reinterpret_cast<const float4*>(table + id * E + c);
            if (keep != nullptr) {
                const uint8_t* kp = keep + r * E + c;
                v.x = kp[0] ? v.x * scale : 0.f;
                v.y = kp[1] ? v.y * scale : 0.f;
                v.z = kp[2] ? v.z * scale : 0.f;
                v.w = kp[3] ? v.w * scale : 0.f;
            }
            *reinterpret_cast<float4*>(out + r * E + c) = v;
        } else {
            float v = table[id * E + c];
            if (keep != nullptr) v = keep[r * E + c] ? v * scale : 0.f;
            out[r * E + c] = v;
        }
    }
}

__global__ __launch_bounds__(256) void embed_scatter_kernel(const int64_t* __restrict__ ids, int64_t rows,
                                                            const float* __restrict__ dout, int V, int E,
                                                            const uint8_t* __restrict__ keep, float scale,
                                                            int64_t pad_id, float* __restrict__ dtable) {
    const int64_t n = rows * E;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / E;
        const int c = (int)(i - r * E);
        const int64_t id = ids[r];
        if (id == pad_id || id < 0 || id >= V) continue;   // padding_idx row receives no gradient
        float v = dout[i];
        if (keep != nullptr) v = keep[i] ? v * scale : 0.f;
        atomicAdd(dtable + id * E + c, v);
    }
}

// ---- deterministic scatter-add --------------------------------------------------------------------------------
constexpr int SC_CHUNK = 64;          // sorted entries per chunk
constexpr int SC_SMALL = 4096;        // rows up to which the sort-free kernel is used

__device__ __forceinline__ float sc_val(const float* __restrict__ dout, const uint8_t* __restrict__ keep, float scale,
                                        int64_t pos, int E, int e) {
    const float v = dout[pos * E + e];
    return keep != nullptr ? (keep[pos * E + e] ? v * scale : 0.f) : v;
}

// one block per row r: if no earlier row has the same id, gather the positions of the later rows with that id (ordered
// compaction: ballot + popcount per wave, waves in order), sum rows r, r', ... ascending and own dtable[id].
__global__ __launch_bounds__(256) void embed_scatter_small_kernel(const int64_t* __restrict__ ids, int rows,
                                                                  const float* __restrict__ dout, int V, int E,
                                                                  const uint8_t* __restrict__ keep, float scale,
                                                                  int64_t pad_id, float* __restrict__ dtable) {
    __shared__ unsigned short list[SC_SMALL];
    __shared__ int wcnt[4];
    __shared__ int s_first, s_total;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t id = ids[r];
    if (id == pad_id || id < 0 || id >= V) return;          // block-uniform
    if (tid == 0) { s_first = 1; s_total = 0; }
    __syncthreads();
    for (int i = tid; i < r; i += 256)
        if (ids[i] == id) s_first = 0;                      // benign race: every writer stores 0
    __syncthreads();
    if (!s_first) return;
    for (int base = r + 1; base < rows; base += 256) {      // block-uniform trip count
        const int i = base + tid;
        const bool match = i < rows && ids[i] == id;
        const unsigned long long bal = __ballot(match);
        if (lane == 0) wcnt[wave] = __popcll(bal);
        __syncthreads();
        int off = s_total;
        for (int w = 0; w < wave; ++w) off += wcnt[w];
        if (match) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)i;
        __syncthreads();
        if (tid == 0) s_total += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
    const int total = s_total;
    for (int e = tid; e < E; e += 256) {
        float acc = sc_val(dout, keep, scale, r, E, e);
        // a frequent id (<s>, the head of a Zipf vocabulary) owns a hundred rows of a 1200-row batch: fetch them eight at a time
        // -- the additions stay in list order (bit-identical), only the loads stop waiting for each other
        int k = 0;
        for (; k + 8 <= total; k += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = sc_val(dout, keep, scale, list[k + j], E, e);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
        for (; k < total; ++k) acc += sc_val(dout, keep, scale, list[k], E, e);
        dtable[id * E + e] += acc;
    }
}

__global__ __launch_bounds__(256) void sc_keys_kernel(const int64_t* __restrict__ ids, int n, int V, int64_t pad_id,
                                                      unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int64_t id = ids[i];
        keys[i] = (id == pad_id || id < 0 || id >= V) ? (unsigned)V : (unsigned)id;     // V = "skip", sorts last
        vals[i] = (unsigned)i;
    }
}

// chunk c of a list of rows sorted by id: runs of one id that lie inside the chunk -> dtable (sole owner), its first / last
// run -> partial rows  part[(2c + s)][E], part_id[2c + s] (s = 0 first run, 1 last run if different; V = none).
//   GATHER: row i of the list is dout[pos[i]] (with the dropout factor);  else row i is src[i] (a partial list: second level).
// Entries with id V ("skip" / "none") are transparent.
template <bool GATHER>
__global__ __launch_bounds__(256) void sc_chunks_kernel(const unsigned* __restrict__ keys, const unsigned* __restrict__ pos, int n,
                                                        const float* __restrict__ src, int V, int E,
                                                        const uint8_t* __restrict__ keep, float scale,
                                                        float* __restrict__ dtable, float* __restrict__ part,
                                                        unsigned* __restrict__ part_id) {
    const int c = blockIdx.x, i0 = c * SC_CHUNK, i1 = min(n, i0 + SC_CHUNK);
    __shared__ unsigned sk[SC_CHUNK], sp[SC_CHUNK];
    __shared__ unsigned s_first;
    const int m = i1 - i0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) { sk[i] = keys[i0 + i]; sp[i] = GATHER ? pos[i0 + i] : (unsigned)(i0 + i); }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned f = (unsigned)V, l = (unsigned)V;
        for (int i = 0; i < m; ++i)
            if (sk[i] != (unsigned)V) { if (f == (unsigned)V) f = sk[i]; l = sk[i]; }
        s_first = f;
        part_id[2 * c] = f;
        part_id[2 * c + 1] = (l != f) ? l : (unsigned)V;
    }
    __syncthreads();
    const unsigned first = s_first;
    if (first == (unsigned)V) return;                        // nothing but skipped rows
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        float acc = 0.f;
        unsigned cur = first;
        for (int i = 0; i < m; ++i) {
            const unsigned k = sk[i];
            if (k == (unsigned)V) continue;
            if (k != cur) {                                  // flush the run that just ended
                if (cur == first) part[(int64_t)(2 * c) * E + e] = acc;
                else dtable[(int64_t)cur * E + e] += acc;    // interior run: this block is its sole owner
                cur = k;
                acc = 0.f;
            }
            acc += GATHER ? sc_val(src, keep, scale, sp[i], E, e) : src[(int64_t)sp[i] * E + e];
        }
        if (cur == first) part[(int64_t)(2 * c) * E + e] = acc;                          // one run spans the chunk
        else part[(int64_t)(2 * c + 1) * E + e] = acc;                                    // cur == last
    }
}

// second pass: block b owns the ids in [b V / nb, (b+1) V / nb); the partial rows of an id are contiguous (sorted) and are
// summed in chunk order
__global__ __launch_bounds__(256) void sc_partials_kernel(const float* __restrict__ part, const unsigned* __restrict__ part_id,
                                                          int nparts, int V, int E, float* __restrict__ dtable) {
    const unsigned lo_id = (unsigned)(((int64_t)blockIdx.x * V) / gridDim.x);
    const unsigned hi_id = (unsigned)(((int64_t)(blockIdx.x + 1) * V) / gridDim.x);
    // "none" slots (id V) sit among real ids: skip them while walking; the real ids are non-decreasing
    // first partial row whose (real) id >= lo_id: linear scan by one thread would be O(nparts); use a lower bound over
    // the EVEN slots (first-run ids, always real except in an all-skip chunk, and non-decreasing)
    int lo = 0, hi = nparts / 2;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (part_id[2 * mid] < lo_id) lo = mid + 1; else hi = mid;
    }
    int start = 2 * lo;
    if (lo > 0 && part_id[2 * lo - 1] != (unsigned)V && part_id[2 * lo - 1] >= lo_id) start = 2 * lo - 1;   // last run of the chunk before
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        float acc = 0.f;
        unsigned cur = (unsigned)V;
        for (int r = start; r < nparts; ++r) {
            const unsigned k = part_id[r];
            if (k == (unsigned)V) continue;
            if (k >= hi_id) break;
            if (k != cur) {
                if (cur != (unsigned)V) dtable[(int64_t)cur * E + e] += acc;
                cur = k;
                acc = 0.f;
            }
            acc += part[(int64_t)r * E + e];
        }
        if (cur != (unsigned)V) dtable[(int64_t)cur * E + e] += acc;
    }
}

inline int sc_bits(int V) { int b = 1; while ((1ll << b) <= V) ++b; return b; }
inline int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }

__global__ __launch_bounds__(256) void keep_mask_kernel(uint8_t* __restrict__ keep, int64_t n, float p, uint64_t seed, uint64_t offset) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        keep[i] = uniform01(seed, offset + (uint64_t)i) >= p ? 1 : 0;
}

__global__ __launch_bounds__(256) void uniform_kernel(float* __restrict__ u, int64_t n, uint64_t seed, uint64_t offset) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        u[i] = uniform01(seed, offset + (uint64_t)i);
}

// Batch-first ids in, time-major everything out, in one pass: out row r = t*B + b takes table[ids[b*ld + t]]; the
// transposed ids (what the backward scatter-add walks) and -- with dropout -- the keep mask are written as by-products.
// Dropout decisions come from an injected time-major mask (keep_in) or from the counter RNG: element (b, t, c) is draw
// rng_offset + (b*T + t)*E + c of stream `seed`, the index laed_dropout_keep_mask gives a batch-first [B,T,E] mask.
template <bool VEC>
__global__ __launch_bounds__(256) void embed_gather_bt_kernel(const int64_t* __restrict__ ids, int64_t ld, int B, int T,
                                                              const float* __restrict__ table, int V, int E,
                                                              const uint8_t* __restrict__ keep_in, int use_rng, float p,
                                                              uint64_t seed, uint64_t rng_offset, float scale,
                                                              float* __restrict__ out, int64_t* __restrict__ ids_tm,
                                                              uint8_t* __restrict__ keep_out) {
    constexpr int VW = VEC ? 4 : 1;
    const int epr = E / VW;
    const int64_t n = (int64_t)B * T * epr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / epr;                     // time-major row t*B + b
        const int c = (int)(i - r * epr) * VW;
        const int t = (int)(r / B), b = (int)(r - (int64_t)t * B);
        int64_t id = ids[(int64_t)b * ld + t];
        if (c == 0 && ids_tm != nullptr) ids_tm[r] = id;
        if (id < 0 || id >= V) id = 0;
        float v[VW];
        if (VEC) {
            const float4 q = *reinterpret_cast<const float4*>(table + id * E + c);
            v[0] = q.x; v[1 % VW] = q.y; v[2 % VW] = q.z; v[3 % VW] = q.w;
        } else {
            v[0] = table[id * E + c];
        }
        if (keep_in != nullptr || use_rng) {
            const uint64_t e0 = rng_offset + ((uint64_t)b * T + t) * E + c;
#pragma unroll
            for (int j = 0; j < VW; ++j) {
                const uint8_t kk = keep_in != nullptr ? keep_in[r * E + c + j] : (uniform01(seed, e0 + j) >= p ? 1 : 0);
                if (keep_out != nullptr) keep_out[r * E + c + j] = kk;
                v[j] = kk ? v[j] * scale : 0.f;
            }
        }
        if (VEC) *reinterpret_cast<float4*>(out + r * E + c) = make_float4(v[0], v[1 % VW], v[2 % VW], v[3 % VW]);
        else out[r * E + c] = v[0];
    }
}

inline int grid_for(int64_t n) {
    int64_t b = cdiv64(n, 256);
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}   // namespace

extern "C" int laed_embed_gather(const int64_t* ids, int64_t rows, const float* table, int V, int E,
                                 const uint8_t* keep, float drop_p, float* out, void* stream) {
    LAED_CHECK_ARG(rows >= 0 && V > 0 && E > 0, "embed_gather: bad size");
    LAED_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "embed_gather: dropout p must be in [0,1)");
    if (rows == 0) return 0;
    LAED_CHECK_ARG(ids && table && out, "embed_gather: null pointer");
    const float scale = 1.0f / (1.0f - drop_p);
    const bool vec = (E % 4 == 0) && aligned16(table) && aligned16(out);
    if (vec) hipLaunchKernelGGL((embed_gather_kernel<true>), dim3(grid_for(rows * (E / 4))), dim3(256), 0, (hipStream_t)stream,
                                ids, rows, table, V, E, keep, scale, out);
    else hipLaunchKernelGGL((embed_gather_kernel<false>), dim3(grid_for(rows * E)), dim3(256), 0, (hipStream_t)stream,
                            ids, rows, table, V, E, keep, scale, out);
    LAED_LAUNCH_CHECK("embed_gather");
    return 0;
}

extern "C" int laed_embed_gather_bt(const int64_t* ids, int64_t ld_ids, int B, int T, const float* table, int V, int E,
                                    const uint8_t* keep_in, int use_rng, float drop_p, uint64_t seed, uint64_t rng_offset,
                                    float* out_tm, int64_t* ids_tm, uint8_t* keep_out, void* stream) {
    LAED_CHECK_ARG(B >= 0 && T >= 0 && V > 0 && E > 0 && ld_ids >= T, "embed_gather_bt: bad size");
    LAED_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "embed_gather_bt: dropout p must be in [0,1)");
    if (B == 0 || T == 0) return 0;
    LAED_CHECK_ARG(ids && table && out_tm, "embed_gather_bt: null pointer");
    LAED_CHECK_ARG(!(use_rng && keep_in), "embed_gather_bt: an injected mask and the RNG exclude each other");
    const float scale = 1.0f / (1.0f - drop_p);
    const int64_t rows = (int64_t)B * T;
    const bool vec = (E % 4 == 0) && aligned16(table) && aligned16(out_tm);
    if (vec) hipLaunchKernelGGL((embed_gather_bt_kernel<true>), dim3(grid_for(rows * (E / 4))), dim3(256), 0, (hipStream_t)stream,
                                ids, ld_ids, B, T, table, V, E, keep_in, use_rng, drop_p, seed, rng_offset, scale, out_tm, ids_tm, keep_out);
    else hipLaunchKernelGGL((embed_gather_bt_kernel<false>), dim3(grid_for(rows * E)), dim3(256), 0, (hipStream_t)stream,
                            ids, ld_ids, B, T, table, V, E, keep_in, use_rng, drop_p, seed, rng_offset, scale, out_tm, ids_tm, keep_out);
    LAED_LAUNCH_CHECK("embed_gather_bt");
    return 0;
}

// bytes of workspace the deterministic scatter-add needs for `rows` ids (0 for the sort-free small case)
extern "C" int64_t laed_embed_scatter_workspace(int64_t rows, int V, int E) {
    if (rows <= SC_SMALL || rows >= 0x7fffffff) return 0;
    size_t temp = 0;
    unsigned* nul = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, temp, nul, nul, nul, nul, (size_t)rows, 0, sc_bits(V), (hipStream_t)0) != hipSuccess)
        return -1;
    const int64_t nc1 = cdiv64(rows, SC_CHUNK), nc2 = cdiv64(2 * nc1, SC_CHUNK);
    return 4 * align256(rows * 4) + align256((int64_t)temp) + align256(2 * nc1 * E * 4) + align256(2 * nc1 * 4) +
           align256(2 * nc2 * E * 4) + align256(2 * nc2 * 4);
}

extern "C" int laed_embed_scatter_add(const int64_t* ids, int64_t rows, const float* dout, int V, int E,
                                      const uint8_t* keep, float drop_p, int64_t pad_id, float* dtable,
                                      void* ws, int64_t ws_bytes, void* stream) {
    LAED_CHECK_ARG(rows >= 0 && V > 0 && E > 0, "embed_scatter_add: bad size");
    LAED_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "embed_scatter_add: dropout p must be in [0,1)");
    if (rows == 0) return 0;
    LAED_CHECK_ARG(ids && dout && dtable, "embed_scatter_add: null pointer");
    static const bool det_on = []() { const char* e = laed_exp_hook("LAED_SCATTER_DET"); return !(e != nullptr && atoi(e) == 0); }();
    const float sc = 1.0f / (1.0f - drop_p);
    if (det_on && rows <= SC_SMALL) {
        hipLaunchKernelGGL(embed_scatter_small_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                           ids, (int)rows, dout, V, E, keep, sc, pad_id, dtable);
        LAED_LAUNCH_CHECK("embed_scatter_small");
        return 0;
    }
    const int64_t need = laed_embed_scatter_workspace(rows, V, E);
    if (det_on && ws != nullptr && need > 0 && ws_bytes >= need && (((uintptr_t)ws) & 255) == 0) {
        hipStream_t st = (hipStream_t)stream;
        const int n = (int)rows;
        const int nchunks = (int)cdiv64(rows, SC_CHUNK);
        char* w = (char*)ws;
        unsigned* k_in = (unsigned*)w;  w += align256(rows * 4);
        unsigned* k_out = (unsigned*)w; w += align256(rows * 4);
        unsigned* v_in = (unsigned*)w;  w += align256(rows * 4);
        unsigned* v_out = (unsigned*)w; w += align256(rows * 4);
        size_t temp = 0;
        unsigned* nul = nullptr;
        (void)rocprim::radix_sort_pairs(nullptr, temp, nul, nul, nul, nul, (size_t)rows, 0, sc_bits(V), st);
        void* tmp = w; w += align256((int64_t)temp);
        const int nc2 = (int)cdiv64(2 * (int64_t)nchunks, SC_CHUNK);
        float* part = (float*)w; w += align256((int64_t)2 * nchunks * E * 4);
        unsigned* part_id = (unsigned*)w; w += align256((int64_t)2 * nchunks * 4);
        float* part2 = (float*)w; w += align256((int64_t)2 * nc2 * E * 4);
        unsigned* part_id2 = (unsigned*)w;
        hipLaunchKernelGGL(sc_keys_kernel, dim3(grid_for(rows)), dim3(256), 0, st, ids, n, V, pad_id, k_in, v_in);
        LAED_LAUNCH_CHECK("embed_scatter keys");
        hipError_t e = rocprim::radix_sort_pairs(tmp, temp, k_in, k_out, v_in, v_out, (size_t)rows, 0, sc_bits(V), st);
        if (e != hipSuccess) return laed_fail((int)e, "embed_scatter_add: radix sort failed: %s", hipGetErrorString(e));
        hipLaunchKernelGGL((sc_chunks_kernel<true>), dim3(nchunks), dim3(256), 0, st, k_out, v_out, n, dout, V, E, keep, sc, dtable,
                           part, part_id);
        LAED_LAUNCH_CHECK("embed_scatter chunks");
        const float* fin = part;
        const unsigned* fin_id = part_id;
        int nfin = 2 * nchunks;
        if (nfin > 8 * SC_CHUNK) {      // a hot id owns hundreds of partial rows: one more chunk level before the serial pass
            hipLaunchKernelGGL((sc_chunks_kernel<false>), dim3(nc2), dim3(256), 0, st, part_id, (const unsigned*)nullptr, nfin, part,
                               V, E, (const uint8_t*)nullptr, 1.0f, dtable, part2, part_id2);
            LAED_LAUNCH_CHECK("embed_scatter chunks2");
            fin = part2; fin_id = part_id2; nfin = 2 * nc2;
        }
        const int nb = nfin / 2 < 256 ? nfin / 2 : 256;
        hipLaunchKernelGGL(sc_partials_kernel, dim3(nb), dim3(256), 0, st, fin, fin_id, nfin, V, E, dtable);
        LAED_LAUNCH_CHECK("embed_scatter partials");
        return 0;
    }
    hipLaunchKernelGGL(embed_scatter_kernel, dim3(grid_for(rows * E)), dim3(256), 0, (hipStream_t)stream,
                       ids, rows, dout, V, E, keep, sc, pad_id, dtable);
    LAED_LAUNCH_CHECK("embed_scatter_add");
    return 0;
}

extern "C" int laed_dropout_keep_mask(uint8_t* keep, int64_t n, float drop_p, uint64_t seed, uint64_t offset, void* stream) {
    LAED_CHECK_ARG(n >= 0 && drop_p >= 0.f && drop_p < 1.f, "dropout_keep_mask: bad argument");
    if (n == 0) return 0;
    LAED_CHECK_ARG(keep, "dropout_keep_mask: null pointer");
    hipLaunchKernelGGL(keep_mask_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, keep, n, drop_p, seed, offset);
    LAED_LAUNCH_CHECK("dropout_keep_mask");
    return 0;
}

extern "C" int laed_uniform(float* u, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
    LAED_CHECK_ARG(n >= 0, "uniform: bad size");
    if (n == 0) return 0;
    LAED_CHECK_ARG(u, "uniform: null pointer");
    hipLaunchKernelGGL(uniform_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, u, n, seed, offset);
    LAED_LAUNCH_CHECK("uniform");
    return 0;
}
