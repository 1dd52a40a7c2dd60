// Adam, small elementwise helpers, error string, version.
#include "laed_common.h"
#include <math.h>

char* laed_err_buf() {
    static thread_local char buf[256] = "";
    return buf;
}

extern "C" int laed_version(void) { return 100; }
extern "C" const char* laed_last_error(void) { return laed_err_buf(); }

namespace {

// torch.optim.Adam (no weight decay / amsgrad): 28 B of HBM traffic per parameter (read p,g,m,v; write p,m,v)
struct AdamConsts { float step_size, omb1, b2, omb2, sqrt_bc2, eps, gs; };

__device__ __forceinline__ void adam_range(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                           float* __restrict__ v, int64_t n4, int64_t n, int64_t tid, int64_t stride,
                                           const AdamConsts& c) {
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
#define ADAM1(P, G, M, V)                                         \
    {                                                             \
        const float gg = (G) * c.gs;                              \
        (M) = (M) + c.omb1 * (gg - (M));        /* lerp_ */       \
        (V) = (V) * c.b2 + (c.omb2 * gg) * gg;  /* addcmul_ */    \
        const float den = sqrtf(V) / c.sqrt_bc2 + c.eps;          \
        (P) = (P) - c.step_size * ((M) / den);  /* addcdiv_ */    \
    }
    for (int64_t i = tid; i < n4; i += stride) {
        float4 pp = p4[i], gg4 = g4[i], mm = m4[i], vv = v4[i];
        ADAM1(pp.x, gg4.x, mm.x, vv.x)
        ADAM1(pp.y, gg4.y, mm.y, vv.y)
        ADAM1(pp.z, gg4.z, mm.z, vv.z)
        ADAM1(pp.w, gg4.w, mm.w, vv.w)
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    for (int64_t i = n4 * 4 + tid; i < n; i += stride) {
        float pp = p[i], mm = m[i], vv = v[i];
        ADAM1(pp, g[i], mm, vv)
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
#undef ADAM1
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n4, int64_t n, AdamConsts c) {
    adam_range(p, g, m, v, n4, n, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, c);
}

// Several tensors in one launch (the descriptors travel in the kernel arguments: no table in device memory).  Tensor t
// owns blocks [blk0[t], blk0[t+1]) and is swept by them exactly as adam_kernel sweeps one tensor: same arithmetic per
// element, so the results do not depend on the grouping.
constexpr int ADAM_MAXT = 16;
struct AdamMulti {
    float* p[ADAM_MAXT];
    const float* g[ADAM_MAXT];
    float* m[ADAM_MAXT];
    float* v[ADAM_MAXT];
    int64_t n4[ADAM_MAXT], n[ADAM_MAXT];
    int blk0[ADAM_MAXT + 1];
    int nt;
};

// The descriptor is indexed with a block-uniform runtime t: the compiler reads it with scalar loads from the kernarg
// segment (ISA checked: private_segment_fixed_size 0, no scratch_ instruction), not through a per-thread copy.
__global__ __launch_bounds__(256) void adam_multi_kernel(AdamMulti a, AdamConsts c) {
    int t = 0;
    while (t + 1 < a.nt && (int)blockIdx.x >= a.blk0[t + 1]) ++t;        // block-uniform
    const int64_t nb = a.blk0[t + 1] - a.blk0[t];
    adam_range(a.p[t], a.g[t], a.m[t], a.v[t], a.n4[t], a.n[t],
               (int64_t)(blockIdx.x - a.blk0[t]) * blockDim.x + threadIdx.x, nb * blockDim.x, c);
}

__global__ __launch_bounds__(256) void axpby_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n, float a, float b) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = (b == 0.f) ? a * x[i] : a * x[i] + b * y[i];
}

__global__ __launch_bounds__(256) void scale_dev_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ s, int64_t n) {
    const float sc = s[0];
    if (sc == 1.0f && x == y) return;      // in place by exactly 1 (the usual d loss of a train step): nothing to move
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = sc * x[i];
}

// y_t = s[0] * y_t for up to 8 tensors in one launch (the fused projection's dh / dW / db scaled by the incoming loss
// gradient); a scale of exactly 1 -- the usual d loss of a train step -- moves nothing.
struct ScaleMulti { float* y[8]; int64_t n[8]; int blk0[9]; int nt; };
__global__ __launch_bounds__(256) void scale_dev_multi_kernel(ScaleMulti a, const float* __restrict__ s) {
    const float sc = s[0];
    if (sc == 1.0f) return;
    int t = 0;
    while (t + 1 < a.nt && (int)blockIdx.x >= a.blk0[t + 1]) ++t;
    float* y = a.y[t];
    const int64_t n = a.n[t], nb = a.blk0[t + 1] - a.blk0[t];
    for (int64_t i = (int64_t)(blockIdx.x - a.blk0[t]) * blockDim.x + threadIdx.x; i < n; i += nb * blockDim.x) y[i] = sc * y[i];
}

// torch.optim.SGD (momentum, dampening 0, no nesterov / weight decay) and torch.optim.RMSprop (alpha, eps, momentum,
// not centred) as model_bases.py:126-133 builds them; same op order as torch's single-tensor path.  buf starts at zero
// (mu*0 + g equals torch's first-step clone of g).
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  int64_t n, float lr, float mu, float gs) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float d = g[i] * gs;
        if (buf != nullptr) { d = buf[i] * mu + d; buf[i] = d; }
        p[i] = p[i] - lr * d;
    }
}

__global__ __launch_bounds__(256) void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq,
                                                      float* __restrict__ buf, int64_t n, float lr, float alpha, float oma,
                                                      float eps, float mu, float gs) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gg = g[i] * gs;
        const float s = sq[i] * alpha + (oma * gg) * gg;      /* mul_(alpha).addcmul_(g, g, value=1-alpha) */
        sq[i] = s;
        const float avg = sqrtf(s) + eps;
        float d = gg / avg;
        if (buf != nullptr) { d = buf[i] * mu + d; buf[i] = d; }
        p[i] = p[i] - lr * d;
    }
}

inline int grid_for(int64_t n) {
    int64_t b = cdiv64(n, 256);
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}   // namespace

static AdamConsts adam_consts(int step, double lr, double beta1, double beta2, double eps, float grad_scale) {
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    AdamConsts c;
    c.step_size = (float)(lr / bc1); c.omb1 = (float)(1.0 - beta1); c.b2 = (float)beta2; c.omb2 = (float)(1.0 - beta2);
    c.sqrt_bc2 = (float)sqrt(bc2); c.eps = (float)eps; c.gs = grad_scale;
    return c;
}

extern "C" int laed_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step,
                              double lr, double beta1, double beta2, double eps, float grad_scale, void* stream) {
    LAED_CHECK_ARG(n >= 0 && step >= 1, "adam_step: bad argument");
    if (n == 0) return 0;
    LAED_CHECK_ARG(p && g && m && v, "adam_step: null pointer");
    const bool vec = aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v);
    const int64_t n4 = vec ? n / 4 : 0;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(vec ? n4 + 1 : n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, n,
                       adam_consts(step, lr, beta1, beta2, eps, grad_scale));
    LAED_LAUNCH_CHECK("adam_step");
    return 0;
}

extern "C" int laed_adam_step_multi(int ntensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                                    const int64_t* n, int step, double lr, double beta1, double beta2, double eps,
                                    float grad_scale, void* stream) {
    LAED_CHECK_ARG(ntensors >= 0 && step >= 1, "adam_step_multi: bad argument");
    LAED_CHECK_ARG(ntensors == 0 || (p && g && m && v && n), "adam_step_multi: null pointer table");
    const AdamConsts c = adam_consts(step, lr, beta1, beta2, eps, grad_scale);
    int t = 0;
    while (t < ntensors) {
        AdamMulti a;
        a.nt = 0;
        a.blk0[0] = 0;
        for (; t < ntensors && a.nt < ADAM_MAXT; ++t) {
            LAED_CHECK_ARG(n[t] >= 0, "adam_step_multi: negative size");
            if (n[t] == 0) continue;
            LAED_CHECK_ARG(p[t] && g[t] && m[t] && v[t], "adam_step_multi: null pointer");
            const bool vec = aligned16(p[t]) && aligned16(g[t]) && aligned16(m[t]) && aligned16(v[t]);
            const int k = a.nt++;
            a.p[k] = p[t]; a.g[k] = g[t]; a.m[k] = m[t]; a.v[k] = v[t];
            a.n[k] = n[t]; a.n4[k] = vec ? n[t] / 4 : 0;
            a.blk0[k + 1] = a.blk0[k] + grid_for(vec ? a.n4[k] + 1 : n[t]);
        }
        if (a.nt == 0) break;
        // grid_for() caps each tensor at 2048 blocks, so the grid is at most ADAM_MAXT x 2048; checked, not assumed
        LAED_CHECK_ARG(a.blk0[a.nt] > 0 && a.blk0[a.nt] <= ADAM_MAXT * 2048, "adam_step_multi: block table out of range");
        hipLaunchKernelGGL(adam_multi_kernel, dim3(a.blk0[a.nt]), dim3(256), 0, (hipStream_t)stream, a, c);
        LAED_LAUNCH_CHECK("adam_step_multi");
    }
    return 0;
}

extern "C" int laed_scale_by_dev_multi(int ntensors, float* const* y, const int64_t* n, const float* s, void* stream) {
    LAED_CHECK_ARG(ntensors >= 0 && ntensors <= 8, "scale_by_dev_multi: at most 8 tensors per call");
    LAED_CHECK_ARG(ntensors == 0 || (y && n && s), "scale_by_dev_multi: null pointer");
    ScaleMulti a;
    a.nt = 0;
    a.blk0[0] = 0;
    for (int t = 0; t < ntensors; ++t) {
        LAED_CHECK_ARG(n[t] >= 0, "scale_by_dev_multi: negative size");
        if (n[t] == 0) continue;
        LAED_CHECK_ARG(y[t], "scale_by_dev_multi: null pointer");
        const int k = a.nt++;
        a.y[k] = y[t]; a.n[k] = n[t];
        a.blk0[k + 1] = a.blk0[k] + grid_for(n[t]);
    }
    if (a.nt == 0) return 0;
    hipLaunchKernelGGL(scale_dev_multi_kernel, dim3(a.blk0[a.nt]), dim3(256), 0, (hipStream_t)stream, a, s);
    LAED_LAUNCH_CHECK("scale_by_dev_multi");
    return 0;
}

__global__ __launch_bounds__(256) void gate_functions_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ sig,
                                                             float* __restrict__ tnh) {
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        sig[i] = sigmoidf_acc(v);
        tnh[i] = tanhf_gate(v);
    }
}

extern "C" int laed_gate_functions(const float* x, int64_t n, float* sig, float* tnh, void* stream) {
    LAED_CHECK_ARG(n >= 0, "gate_functions: bad size");
    if (n == 0) return 0;
    LAED_CHECK_ARG(x && sig && tnh, "gate_functions: null pointer");
    hipLaunchKernelGGL(gate_functions_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, sig, tnh);
    LAED_LAUNCH_CHECK("gate_functions");
    return 0;
}

extern "C" int laed_axpby(float* y, const float* x, int64_t n, float a, float b, void* stream) {
    LAED_CHECK_ARG(n >= 0, "axpby: bad size");
    if (n == 0) return 0;
    LAED_CHECK_ARG(x && y, "axpby: null pointer");
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, x, n, a, b);
    LAED_LAUNCH_CHECK("axpby");
    return 0;
}

extern "C" int laed_scale_by_dev(float* y, const float* x, const float* s, int64_t n, void* stream) {
    LAED_CHECK_ARG(n >= 0, "scale_by_dev: bad size");
    if (n == 0) return 0;
    LAED_CHECK_ARG(x && y && s, "scale_by_dev: null pointer");
    hipLaunchKernelGGL(scale_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, x, s, n);
    LAED_LAUNCH_CHECK("scale_by_dev");
    return 0;
}

extern "C" int laed_sgd_step(float* p, const float* g, float* momentum_buf, int64_t n, double lr, double momentum,
                             float grad_scale, void* stream) {
    LAED_CHECK_ARG(n >= 0, "sgd_step: bad size");
    if (n == 0) return 0;
    LAED_CHECK_ARG(p && g && (momentum == 0.0 || momentum_buf), "sgd_step: null pointer");
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, momentum != 0.0 ? momentum_buf : nullptr,
                       n, (float)lr, (float)momentum, grad_scale);
    LAED_LAUNCH_CHECK("sgd_step");
    return 0;
}

extern "C" int laed_rmsprop_step(float* p, const float* g, float* square_avg, float* momentum_buf, int64_t n, double lr,
                                 double alpha, double eps, double momentum, float grad_scale, void* stream) {
    LAED_CHECK_ARG(n >= 0, "rmsprop_step: bad size");
    if (n == 0) return 0;
    LAED_CHECK_ARG(p && g && square_avg && (momentum <= 0.0 || momentum_buf), "rmsprop_step: null pointer");
    hipLaunchKernelGGL(rmsprop_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, square_avg,
                       momentum > 0.0 ? momentum_buf : nullptr, n, (float)lr, (float)alpha, (float)(1.0 - alpha), (float)eps,
                       (float)momentum, grad_scale);
    LAED_LAUNCH_CHECK("rmsprop_step");
    return 0;
}
