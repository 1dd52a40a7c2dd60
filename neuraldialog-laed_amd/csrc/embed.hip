// Embedding row gather / scatter-add with optional input dropout, and the counter-based RNG fills.
// HBM-bound: one float4 per lane, rows contiguous (E floats); scatter uses no-return fp32 atomics
// (global_atomic_add_f32), 256 contiguous bytes per wave-instruction as the atomics guide asks.
#include "laed_common.h"

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

__global__ __launch_bounds__(256) void keep_mask_kernel(uint8_t* __restrict__ keep, int64_t n, float p, uint64_t seed, uint64_t offset) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        keep[i] = uniform01(seed, offset + (uint64_t)i) >= p ? 1 : 0;
}

__global__ __launch_bounds__(256) void uniform_kernel(float* __restrict__ u, int64_t n, uint64_t seed, uint64_t offset) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        u[i] = uniform01(seed, offset + (uint64_t)i);
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

extern "C" int laed_embed_scatter_add(const int64_t* ids, int64_t rows, const float* dout, int V, int E,
                                      const uint8_t* keep, float drop_p, int64_t pad_id, float* dtable, void* stream) {
    LAED_CHECK_ARG(rows >= 0 && V > 0 && E > 0, "embed_scatter_add: bad size");
    LAED_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "embed_scatter_add: dropout p must be in [0,1)");
    if (rows == 0) return 0;
    LAED_CHECK_ARG(ids && dout && dtable, "embed_scatter_add: null pointer");
    hipLaunchKernelGGL(embed_scatter_kernel, dim3(grid_for(rows * E)), dim3(256), 0, (hipStream_t)stream,
                       ids, rows, dout, V, E, keep, 1.0f / (1.0f - drop_p), pad_id, dtable);
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
