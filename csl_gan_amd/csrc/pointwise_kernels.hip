// Pointwise / normalisation kernels (HBM-bound, 16-byte accesses where alignment allows).  gfx950.
//
// Replaces (reference file:line): F.leaky_relu backward DCResNet_models.py:132 (autograd);
// nn.GroupNorm + F.relu DCResNet_models.py:55-57,63-67,101-102; torch.optim.Adam.step train.py:76,484.
#include "common.h"

namespace cslgan {

__global__ void act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, long long n, float slope,
                               float* __restrict__ out, int vec) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    if (vec) {
        const long long n4 = n >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(g);
        const float4* y4 = reinterpret_cast<const float4*>(y);
        float4* o4 = reinterpret_cast<float4*>(out);
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            const float4 a = g4[i], b = y4[i];
            o4[i] = make_float4(b.x > 0.f ? a.x : slope * a.x, b.y > 0.f ? a.y : slope * a.y,
                                b.z > 0.f ? a.z : slope * a.z, b.w > 0.f ? a.w : slope * a.w);
        }
        for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
            out[i] = y[i] > 0.f ? g[i] : slope * g[i];
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
            out[i] = y[i] > 0.f ? g[i] : slope * g[i];
    }
}

// one block per (n, group): two-pass mean / variance over HW x cpg elements
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const float* __restrict__ x, int HW, int C, int groups,
                                                              float eps, float* __restrict__ stats) {
    __shared__ float red[4];
    __shared__ float s_mean;
    const int n = blockIdx.x / groups, g = blockIdx.x - n * groups;
    const int cpg = C / groups;
    const long long tot = (long long)HW * cpg;
    const float* base = x + (long long)n * HW * C + g * cpg;
    float acc = 0.f;
    for (long long i = threadIdx.x; i < tot; i += 256) {
        const long long px = i / cpg;
        const int j = (int)(i - px * cpg);
        acc += base[px * C + j];
    }
    float t = block_sum_256(acc, red);
    if (threadIdx.x == 0) s_mean = t / (float)tot;
    __syncthreads();
    const float mean = s_mean;
    acc = 0.f;
    for (long long i = threadIdx.x; i < tot; i += 256) {
        const long long px = i / cpg;
        const int j = (int)(i - px * cpg);
        const float d = base[px * C + j] - mean;
        acc = fmaf(d, d, acc);
    }
    __syncthreads();
    t = block_sum_256(acc, red);
    if (threadIdx.x == 0) {
        stats[2 * blockIdx.x] = mean;
        stats[2 * blockIdx.x + 1] = rsqrtf(t / (float)tot + eps);
    }
}

__global__ void groupnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, const float* __restrict__ stats, long long total,
                                       int HW, int C, int groups, int relu, float* __restrict__ y) {
    const int cpg = C / groups;
    const long long per_img = (long long)HW * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long n = i / per_img;
        const int g = c / cpg;
        const float mean = stats[2 * (n * groups + g)], rstd = stats[2 * (n * groups + g) + 1];
        float v = (x[i] - mean) * rstd * gamma[c] + beta[c];
        if (relu) v = v > 0.f ? v : 0.f;
        y[i] = v;
    }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

static unsigned grid_for(long long n, int per_thread = 1) {
    long long b = (n / per_thread + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace cslgan

using namespace cslgan;

extern "C" {

int cslgan_act_bwd_f32(const float* g, const float* y, int64_t n, float slope, float* out, void* stream) {
    CSLGAN_REQUIRE(g && y && out, "act_bwd: null argument");
    CSLGAN_REQUIRE(n >= 0, "act_bwd: n < 0");
    if (n == 0) return CSLGAN_OK;
    const int vec = aligned16(g) && aligned16(y) && aligned16(out);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, g, y, (long long)n, slope, out, vec);
    return check_launch("act_bwd_kernel");
}

int cslgan_groupnorm_act_f32(const float* x, const float* gamma, const float* beta, int N, int HW, int C, int groups,
                             float eps, int relu, float* stats_ws, float* y, void* stream) {
    CSLGAN_REQUIRE(x && gamma && beta && stats_ws && y, "groupnorm: null argument");
    CSLGAN_REQUIRE(N > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0, "groupnorm: C=%d not divisible by groups=%d", C, groups);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(groupnorm_stats_kernel, dim3((unsigned)(N * groups)), dim3(256), 0, st, x, HW, C, groups, eps, stats_ws);
    int rc = check_launch("groupnorm_stats_kernel");
    if (rc) return rc;
    const long long total = (long long)N * HW * C;
    hipLaunchKernelGGL(groupnorm_apply_kernel, dim3(grid_for(total)), dim3(256), 0, st, x, gamma, beta, stats_ws, total, HW, C,
                       groups, relu, y);
    return check_launch("groupnorm_apply_kernel");
}

int cslgan_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                         float weight_decay, int step, void* stream) {
    CSLGAN_REQUIRE(p && g && m && v, "adam: null argument");
    CSLGAN_REQUIRE(n >= 0 && step >= 1, "adam: bad n/step");
    if (n == 0) return CSLGAN_OK;
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, lr, b1, b2,
                       eps, weight_decay, (float)bc1, (float)sqrt(bc2));
    return check_launch("adam_kernel");
}

}  // extern "C"
