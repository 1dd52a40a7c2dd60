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
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n4, int64_t n, float step_size, float omb1, float b2,
                                                   float omb2, float sqrt_bc2, float eps, float gs) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
#define ADAM1(P, G, M, V)                                   \
    {                                                       \
        const float gg = (G) * gs;                          \
        (M) = (M) + omb1 * (gg - (M));      /* lerp_ */     \
        (V) = (V) * b2 + (omb2 * gg) * gg;  /* addcmul_ */  \
        const float den = sqrtf(V) / sqrt_bc2 + eps;        \
        (P) = (P) - step_size * ((M) / den); /* addcdiv_ */ \
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

extern "C" int laed_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step,
                              double lr, double beta1, double beta2, double eps, float grad_scale, void* stream) {
    LAED_CHECK_ARG(n >= 0 && step >= 1, "adam_step: bad argument");
    if (n == 0) return 0;
    LAED_CHECK_ARG(p && g && m && v, "adam_step: null pointer");
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    const bool vec = aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v);
    const int64_t n4 = vec ? n / 4 : 0;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(vec ? n4 + 1 : n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, n,
                       (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps,
                       grad_scale);
    LAED_LAUNCH_CHECK("adam_step");
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
