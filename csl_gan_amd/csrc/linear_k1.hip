// Linear layers with ONE output unit (the critic's head, DCResNet_models.py:145: 8192 -> 1, no bias): forward and data gradient as
// plain streams.  As an implicit GEMM with N = 1 they ran on a 128x32 MFMA tile that is 31/32 padding (14-29 us per launch for
// 4-13 MB); here a workgroup takes one row: y[n] = act(<x[n,:], w> + b) and gx[n,:] = gy[n] * w (* lrelu'(mask)).  gfx950 only.
#include "common.h"
#include "igemm.h"

namespace cslgan {

__global__ __launch_bounds__(256) void linear_k1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            const float* __restrict__ res, long long C, int act, float* __restrict__ y) {
    __shared__ float s_red[4];
    const long long n = blockIdx.x;
    const float4* xr = reinterpret_cast<const float4*>(x + n * C);
    const float4* wr = reinterpret_cast<const float4*>(w);
    float acc = 0.f;
    for (long long i = threadIdx.x; i < (C >> 2); i += 256) {
        const float4 a = xr[i], b = wr[i];
        acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
    }
    const float tot = block_sum_256(acc, s_red);
    if (threadIdx.x == 0) {
        float val = tot + (bias ? bias[0] : 0.f) + (res ? res[n] : 0.f);
        if (act == CSLGAN_ACT_LRELU02) val = val > 0.f ? val : 0.2f * val;
        else if (act == CSLGAN_ACT_RELU) val = val > 0.f ? val : 0.f;
        else if (act == CSLGAN_ACT_TANH) val = tanhf(val);
        y[n] = val;
    }
}

__global__ __launch_bounds__(256) void linear_k1_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w, const float* __restrict__ mask,
                                                              long long C4, float* __restrict__ gx) {
    const long long n = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= C4) return;
    const float g = gy[n];
    const float4 b = reinterpret_cast<const float4*>(w)[i];
    float4 o = make_float4(g * b.x, g * b.y, g * b.z, g * b.w);
    if (mask) {
        const float4 m = reinterpret_cast<const float4*>(mask + n * C4 * 4)[i];
        o.x *= m.x > 0.f ? 1.f : 0.2f; o.y *= m.y > 0.f ? 1.f : 0.2f; o.z *= m.z > 0.f ? 1.f : 0.2f; o.w *= m.w > 0.f ? 1.f : 0.2f;
    }
    reinterpret_cast<float4*>(gx + n * C4 * 4)[i] = o;
}

// a linear layer with one output and a long input: H = W = P = Q = R = S = 1, K = 1, C % 4 == 0
bool linear_k1_shape(const cslgan_conv_t* c) {
    static const int env = [] { const char* e = getenv("CSLGAN_LINEAR_K1"); return e ? atoi(e) : 1; }();
    return env && c->K == 1 && c->H == 1 && c->W == 1 && c->R == 1 && c->S == 1 && c->P == 1 && c->Q == 1 && c->stride == 1 && c->pad == 0 &&
           (c->C & 3) == 0 && c->C >= 256 && c->N <= 65535;
}

int launch_linear_k1_fwd(const cslgan_conv_t* c, const float* x, const float* w, const float* bias, const float* residual, int act,
                         float* y, hipStream_t st) {
    note_kernel("linear_k1_fwd_kernel");
    hipLaunchKernelGGL(linear_k1_fwd_kernel, dim3((unsigned)c->N), dim3(256), 0, st, x, w, bias, residual, (long long)c->C, act, y);
    return check_launch("linear_k1_fwd_kernel");
}

int launch_linear_k1_dgrad(const cslgan_conv_t* c, const float* gy, const float* w, const float* mask, float* gx, hipStream_t st) {
    const long long C4 = c->C >> 2;
    note_kernel("linear_k1_dgrad_kernel");
    hipLaunchKernelGGL(linear_k1_dgrad_kernel, dim3((unsigned)((C4 + 255) / 256), (unsigned)c->N), dim3(256), 0, st, gy, w, mask, C4, gx);
    return check_launch("linear_k1_dgrad_kernel");
}

}  // namespace cslgan
