// Small fused kernels of the D-step's glue: what ran as dozens of 4-microsecond elementwise / reduction launches (each one a
// full dispatch on a 256-CU device) becomes one launch per logical operation.  gfx950 only.
//
//   DCResNet_models.py:149-153   real_loss / fake_loss of the critic (+-mean)      -> segment_means (+ backward)
//   train.py:488-496             logger.stats[...] += ... of train_D                -> dstep_stats
//   train.py:310-329             update_grad_logging                                -> grad_log_stats
//   gradient_penalty.py:36       interpolates = alpha * real + (1 - alpha) * fake   -> lerp_rows
//   gradient_penalty.py:52-54    norms, (norms - 1)^2 [clamped], mean, * weight     -> lipschitz_term (+ backward)
//   train.py:76,484              torch.optim.Adam.step over all parameter tensors   -> adam_multi
#include "common.h"

namespace cslgan {

// ------------------------------------------------------------------------------------------------------------------
// segment means:  vec[s] = scale[s] * sum(x[off_s : off_s + n_s]),  total = sum_s vec[s]
// ------------------------------------------------------------------------------------------------------------------
struct SegMeanArgs {
    int n_seg;
    int off[CSLGAN_MAX_ROLES + 1];
    float scale[CSLGAN_MAX_ROLES];
};

__global__ void segment_means_kernel(const float* __restrict__ x, SegMeanArgs a, float* __restrict__ vec, float* __restrict__ total) {
    __shared__ float red[4];
    __shared__ float tot;
    if (threadIdx.x == 0) tot = 0.f;
    for (int s = 0; s < a.n_seg; ++s) {
        float acc = 0.f;
        for (int i = a.off[s] + threadIdx.x; i < a.off[s + 1]; i += blockDim.x) acc += x[i];
        __syncthreads();
        const float r = block_sum_256(acc, red);
        if (threadIdx.x == 0) {
            const float v = r * a.scale[s];
            vec[s] = v;
            tot += v;
        }
    }
    if (threadIdx.x == 0 && total) *total = tot;
}

// gx[i] = (g_total + g_vec[s]) * scale[s] for i in segment s (either gradient may be absent)
__global__ void segment_means_bwd_kernel(const float* __restrict__ g_total, const float* __restrict__ g_vec, SegMeanArgs a,
                                         float* __restrict__ gx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.off[a.n_seg]) return;
    int s = 0;
    while (s + 1 < a.n_seg && i >= a.off[s + 1]) ++s;
    float g = g_total ? *g_total : 0.f;
    if (g_vec) g += g_vec[s];
    gx[i] = g * a.scale[s];
}

// ------------------------------------------------------------------------------------------------------------------
// train.py:488-496: acc[0] gate (adv loss), [1] D Adv Loss, [2] D Real Loss, [3] D Fake Loss, [4] D Real Acc, [5] D Fake Acc,
// [6] D Penalty — all "+=" into persistent accumulators
// ------------------------------------------------------------------------------------------------------------------
__global__ void dstep_stats_kernel(const float* __restrict__ d_real, int n_real, const float* __restrict__ d_fake, int n_fake,
                                   const float* __restrict__ real_loss, const float* __restrict__ fake_loss,
                                   const float* __restrict__ penalty, float* __restrict__ acc) {
    __shared__ float red[4];
    float pos = 0.f, neg = 0.f;
    for (int i = threadIdx.x; i < n_real; i += blockDim.x) pos += d_real[i] > 0.f ? 1.f : 0.f;
    for (int i = threadIdx.x; i < n_fake; i += blockDim.x) neg += d_fake[i] < 0.f ? 1.f : 0.f;
    const float p = block_sum_256(pos, red);
    __syncthreads();
    const float q = block_sum_256(neg, red);
    if (threadIdx.x == 0) {
        const float rl = *real_loss, fl = *fake_loss;
        acc[0] += rl + fl;
        acc[1] += rl + fl;
        acc[2] += rl;
        acc[3] += fl;
        acc[4] += 100.f * p / (float)n_real;
        acc[5] += 100.f * q / (float)n_fake;
        if (penalty) acc[6] += *penalty;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// train.py:310-329: per layer (or for the one flat norm) mean / population std / max of the B logged per-sample norms, the
// clip norms, and the fraction of samples with clip factor < 0.999 — accumulated into the logger's device-side sums.
// One workgroup per logged row; flat mode sums the layers' squared norms first.
// ------------------------------------------------------------------------------------------------------------------
__global__ void grad_log_stats_kernel(const float* __restrict__ sq, int n_layers, long long ld, long long col0, int B,
                                      const float* __restrict__ C, int per_layer, float eps, float* __restrict__ acc_mean,
                                      float* __restrict__ acc_std, float* __restrict__ acc_max, float* __restrict__ acc_c,
                                      float* __restrict__ acc_clipped) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const float c = C[per_layer ? row : 0];
    auto norm_of = [&](int b) {
        float q;
        if (per_layer) q = sq[row * ld + col0 + b];
        else {
            q = 0.f;
            for (int l = 0; l < n_layers; ++l) q += sq[l * ld + col0 + b];
        }
        return sqrtf(q);
    };
    float s1 = 0.f, mx = 0.f, cl = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float n = norm_of(b);
        s1 += n;
        mx = fmaxf(mx, n);
        cl += fminf(1.f, c / (n + eps)) < 0.999f ? 1.f : 0.f;
    }
    __shared__ float s_mean;
    const float t1 = block_sum_256(s1, red);
    if (threadIdx.x == 0) s_mean = t1 / (float)B;
    __syncthreads();
    const float mean = s_mean;
    // second pass over the B norms: sum (n - mean)^2 (E[n^2] - mean^2 cancels catastrophically when the norms are nearly equal)
    float s2 = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float d = norm_of(b) - mean;
        s2 = fmaf(d, d, s2);
    }
    const float t2 = block_sum_256(s2, red);
    __syncthreads();
    const float t3 = block_sum_256(cl, red);
    __syncthreads();
    // max: same 4-float scratch
    float m = mx;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        acc_mean[row] += mean;
        acc_std[row] += sqrtf(t2 / (float)B);
        acc_max[row] += fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        acc_c[row] += c;
        acc_clipped[row] += t3 / (float)B;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// gradient_penalty.py:36: out[b, :] = alpha[b] * real[b, :] + (1 - alpha[b]) * fake[b, :]
// ------------------------------------------------------------------------------------------------------------------
__global__ void lerp_rows_kernel(const float* __restrict__ real, const float* __restrict__ fake, const float* __restrict__ alpha,
                                 long long len, float* __restrict__ out) {
    const long long row = blockIdx.y;
    const float a = alpha[row];
    const float* r = real + row * len;
    const float* f = fake + row * len;
    float* o = out + row * len;
    if ((len & 3) == 0) {
        for (long long j = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; j < len; j += (long long)gridDim.x * blockDim.x * 4) {
            const float4 rv = *reinterpret_cast<const float4*>(r + j), fv = *reinterpret_cast<const float4*>(f + j);
            float4 ov;
            ov.x = a * rv.x + (1.f - a) * fv.x;
            ov.y = a * rv.y + (1.f - a) * fv.y;
            ov.z = a * rv.z + (1.f - a) * fv.z;
            ov.w = a * rv.w + (1.f - a) * fv.w;
            *reinterpret_cast<float4*>(o + j) = ov;
        }
    } else {
        for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < len; j += (long long)gridDim.x * blockDim.x)
            o[j] = a * r[j] + (1.f - a) * f[j];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// gradient_penalty.py:52-54 (+ the mean and the weight of :65 / :41): one workgroup per row.
//   norm[b] = ||t[b, :]||_2,  per[b] = coef * phi(norm[b]),  phi(n) = (n - 1)^2  or  max(n - 1, 0)^2 (one-sided)
//   total   = sum_b per[b]   (the caller folds 1/B into coef for the batch mean)
// The last workgroup to finish adds the rows up (ticket counter in `ticket`, which it resets to 0 for the next launch).
// ------------------------------------------------------------------------------------------------------------------
__global__ void lipschitz_term_kernel(const float* __restrict__ t, long long len, int one_sided, float coef, float* __restrict__ norm,
                                      float* __restrict__ per, float* __restrict__ total, unsigned* __restrict__ ticket) {
    __shared__ float red[4];
    __shared__ bool last;
    const long long row = blockIdx.x;
    const float* p = t + row * len;
    float acc = 0.f;
    if ((len & 3) == 0 && aligned16_dev(p)) {
        for (long long j = (long long)threadIdx.x * 4; j < len; j += (long long)blockDim.x * 4) {
            const float4 v = *reinterpret_cast<const float4*>(p + j);
            acc = fmaf(v.x, v.x, acc);
            acc = fmaf(v.y, v.y, acc);
            acc = fmaf(v.z, v.z, acc);
            acc = fmaf(v.w, v.w, acc);
        }
    } else {
        for (long long j = threadIdx.x; j < len; j += blockDim.x) acc = fmaf(p[j], p[j], acc);
    }
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) {
        const float n = sqrtf(s);
        float d = n - 1.f;
        if (one_sided) d = fmaxf(d, 0.f);
        norm[row] = n;
        per[row] = coef * d * d;
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last || !total) {
        if (last && threadIdx.x == 0) *ticket = 0;
        return;
    }
    __threadfence();
    float a2 = 0.f;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x) a2 += __hip_atomic_load(per + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float tot = block_sum_256(a2, red);
    if (threadIdx.x == 0) {
        *total = tot;
        *ticket = 0;
    }
}

// gt[b, :] = (g_total + g_per[b]) * coef * 2 (n_b - 1)[n_b > 1 if one-sided] * t[b, :] / n_b
__global__ void lipschitz_term_bwd_kernel(const float* __restrict__ t, const float* __restrict__ norm, const float* __restrict__ g_total,
                                          const float* __restrict__ g_per, long long len, int one_sided, float coef, float* __restrict__ gt) {
    const long long row = blockIdx.y;
    const float n = norm[row];
    float d = n - 1.f;
    if (one_sided) d = fmaxf(d, 0.f);
    float g = g_total ? *g_total : 0.f;
    if (g_per) g += g_per[row];
    const float f = g * coef * 2.f * d / n;
    const float* p = t + row * len;
    float* o = gt + row * len;
    if ((len & 3) == 0 && aligned16_dev(p) && aligned16_dev(o)) {
        for (long long j = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; j < len; j += (long long)gridDim.x * blockDim.x * 4) {
            float4 v = *reinterpret_cast<const float4*>(p + j);
            v.x *= f; v.y *= f; v.z *= f; v.w *= f;
            *reinterpret_cast<float4*>(o + j) = v;
        }
    } else {
        for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < len; j += (long long)gridDim.x * blockDim.x) o[j] = p[j] * f;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Adam over up to CSLGAN_MAX_SEGS tensors in one launch (torch.optim.Adam semantics, train.py:76).  A workgroup owns a
// 4096-element chunk of one tensor; the bias corrections come by value or from a device step counter (graph replay).
// ------------------------------------------------------------------------------------------------------------------
constexpr int ADAM_CHUNK = 4096;
struct AdamArgs {
    int n_seg;
    float* p[CSLGAN_MAX_SEGS];
    const float* g[CSLGAN_MAX_SEGS];
    float* m[CSLGAN_MAX_SEGS];
    float* v[CSLGAN_MAX_SEGS];
    long long n[CSLGAN_MAX_SEGS];
    int vec_ok[CSLGAN_MAX_SEGS];
    int chunk_prefix[CSLGAN_MAX_SEGS + 1];
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr_bc1, float b1, float b2, float eps, float wd,
                                         float bc2_sqrt) {
    if (wd != 0.f) g = fmaf(wd, p, g);
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - lr_bc1 * (m / denom);
}

__global__ void adam_multi_kernel(AdamArgs a, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                  const int* __restrict__ step_dev) {
    if (step_dev) {
        const float st = (float)(*step_dev);
        bc1 = 1.f - powf(b1, st);
        bc2_sqrt = sqrtf(1.f - powf(b2, st));
    }
    const float lr_bc1 = lr / bc1;
    int s = 0;
    while (s + 1 < a.n_seg && (int)blockIdx.x >= a.chunk_prefix[s + 1]) ++s;
    const long long base = (long long)((int)blockIdx.x - a.chunk_prefix[s]) * ADAM_CHUNK;
    const long long end = base + ADAM_CHUNK < a.n[s] ? base + ADAM_CHUNK : a.n[s];
    float* p = a.p[s];
    const float* g = a.g[s];
    float* m = a.m[s];
    float* v = a.v[s];
    if (a.vec_ok[s]) {
        for (long long i = base + (long long)threadIdx.x * 4; i + 3 < end; i += (long long)blockDim.x * 4) {
            float4 pv = *reinterpret_cast<float4*>(p + i), mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
            const float4 gv = *reinterpret_cast<const float4*>(g + i);
            adam_one(pv.x, gv.x, mv.x, vv.x, lr_bc1, b1, b2, eps, wd, bc2_sqrt);
            adam_one(pv.y, gv.y, mv.y, vv.y, lr_bc1, b1, b2, eps, wd, bc2_sqrt);
            adam_one(pv.z, gv.z, mv.z, vv.z, lr_bc1, b1, b2, eps, wd, bc2_sqrt);
            adam_one(pv.w, gv.w, mv.w, vv.w, lr_bc1, b1, b2, eps, wd, bc2_sqrt);
            *reinterpret_cast<float4*>(p + i) = pv;
            *reinterpret_cast<float4*>(m + i) = mv;
            *reinterpret_cast<float4*>(v + i) = vv;
        }
        const long long tail = base + ((end - base) & ~3LL);
        for (long long i = tail + threadIdx.x; i < end; i += blockDim.x) adam_one(p[i], g[i], m[i], v[i], lr_bc1, b1, b2, eps, wd, bc2_sqrt);
    } else {
        for (long long i = base + threadIdx.x; i < end; i += blockDim.x) adam_one(p[i], g[i], m[i], v[i], lr_bc1, b1, b2, eps, wd, bc2_sqrt);
    }
}

static int fill_seg_mean(SegMeanArgs& a, int n_seg, const int32_t* sizes, const float* scale) {
    a.n_seg = n_seg;
    int off = 0;
    for (int s = 0; s < n_seg; ++s) {
        a.off[s] = off;
        a.scale[s] = scale[s];
        off += sizes[s];
    }
    for (int s = n_seg; s <= CSLGAN_MAX_ROLES; ++s) a.off[s] = off;
    for (int s = n_seg; s < CSLGAN_MAX_ROLES; ++s) a.scale[s] = 0.f;
    return off;
}

// One workgroup: the adaptive statistic of every layer (tree reduction in a fixed order), the clip norm(s), then the stacked
// squared norms, the clip factors, their gathered rows and the row-weight jobs (include/cslgan.h: cslgan_adaptive_clip_f32).
__global__ __launch_bounds__(256) void adaptive_clip_kernel(const cslgan_adaptive_clip_t a, long long n_adapt, long long n_rows, int stat_max,
                                                            float scalar, int per_layer, float eps, long long first_private_row,
                                                            float* __restrict__ r_out, float* __restrict__ c_out,
                                                            float* __restrict__ sq_out, float* __restrict__ f_out, float* __restrict__ f_mat) {
    __shared__ float s_red[256];
    __shared__ float s_r[CSLGAN_MAX_CLIP_LAYERS], s_c[CSLGAN_MAX_CLIP_LAYERS];
    const int tid = threadIdx.x, L = a.n_layers;
    for (int l = 0; l < L; ++l) {
        float v = 0.f;
        for (long long i = tid; i < n_adapt; i += 256) {
            const float nrm = sqrtf(a.sq_adapt[l][i]);
            v = stat_max ? fmaxf(v, nrm) : v + nrm;
        }
        s_red[tid] = v;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w) s_red[tid] = stat_max ? fmaxf(s_red[tid], s_red[tid + w]) : s_red[tid] + s_red[tid + w];
            __syncthreads();
        }
        if (tid == 0) s_r[l] = stat_max ? s_red[0] : s_red[0] / (float)n_adapt;
        __syncthreads();
    }
    if (tid == 0) {
        float tot = 0.f;
        for (int l = 0; l < L; ++l) { r_out[l] = s_r[l]; s_c[l] = s_r[l] * scalar; tot += s_r[l] * s_r[l]; }
        if (per_layer) for (int l = 0; l < L; ++l) c_out[l] = s_c[l];
        else { s_c[0] = sqrtf(tot) * scalar; c_out[0] = s_c[0]; }
    }
    __syncthreads();
    for (long long r = tid; r < n_rows; r += 256) {           // the arithmetic of clip_factors_kernel
        if (per_layer) {
            for (int l = 0; l < L; ++l) {
                const float q = a.sq_rows[l][r];
                sq_out[(long long)l * n_rows + r] = q;
                float f = s_c[l] / (sqrtf(q) + eps);
                f = f > 1.f ? 1.f : f;
                f_out[(long long)l * n_rows + r] = r < first_private_row ? 1.f : f;
            }
        } else {
            float tot = 0.f;
            for (int l = 0; l < L; ++l) { const float q = a.sq_rows[l][r]; sq_out[(long long)l * n_rows + r] = q; tot += q; }
            float f = s_c[0] / (sqrtf(tot) + eps);
            f = f > 1.f ? 1.f : f;
            f_out[r] = r < first_private_row ? 1.f : f;
        }
    }
    __syncthreads();                                          // f_out written by this workgroup is visible to it from here on
    if (f_mat && per_layer)
        for (int m = 0; m < a.n_mat; ++m)
            for (long long r = tid; r < n_rows; r += 256) f_mat[(long long)m * n_rows + r] = f_out[(long long)a.mat_layer[m] * n_rows + r];
    for (int j = 0; j < a.n_jobs; ++j) {
        const float* src = f_out + (per_layer ? (long long)a.job_layer[j] * n_rows : 0) + a.job_first[j];
        for (int i = tid; i < a.job_count[j]; i += 256) a.job_dst[j][i] = a.job_scale[j] * src[i];
    }
}

}  // namespace cslgan

using namespace cslgan;

extern "C" {

int cslgan_segment_means_f32(const float* x, int n_seg, const int32_t* sizes, const float* scale, float* vec, float* total,
                             void* stream) {
    CSLGAN_REQUIRE(x && sizes && scale && vec, "segment_means: null argument");
    CSLGAN_REQUIRE(n_seg >= 1 && n_seg <= CSLGAN_MAX_ROLES, "segment_means: n_seg=%d outside [1, %d]", n_seg, CSLGAN_MAX_ROLES);
    for (int s = 0; s < n_seg; ++s) CSLGAN_REQUIRE(sizes[s] >= 1, "segment_means: empty segment %d", s);
    SegMeanArgs a;
    fill_seg_mean(a, n_seg, sizes, scale);
    hipLaunchKernelGGL(segment_means_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, a, vec, total);
    return check_launch("segment_means_kernel");
}

int cslgan_segment_means_bwd_f32(const float* g_total, const float* g_vec, int n_seg, const int32_t* sizes, const float* scale,
                                 float* gx, void* stream) {
    CSLGAN_REQUIRE(sizes && scale && gx, "segment_means_bwd: null argument");
    CSLGAN_REQUIRE(g_total || g_vec, "segment_means_bwd: no incoming gradient");
    CSLGAN_REQUIRE(n_seg >= 1 && n_seg <= CSLGAN_MAX_ROLES, "segment_means_bwd: n_seg=%d outside [1, %d]", n_seg, CSLGAN_MAX_ROLES);
    SegMeanArgs a;
    const int n = fill_seg_mean(a, n_seg, sizes, scale);
    if (n == 0) return CSLGAN_OK;
    hipLaunchKernelGGL(segment_means_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, g_total, g_vec, a, gx);
    return check_launch("segment_means_bwd_kernel");
}

int cslgan_dstep_stats_f32(const float* d_real, int n_real, const float* d_fake, int n_fake, const float* real_loss,
                           const float* fake_loss, const float* penalty, float* acc7, void* stream) {
    CSLGAN_REQUIRE(d_real && d_fake && real_loss && fake_loss && acc7, "dstep_stats: null argument");
    CSLGAN_REQUIRE(n_real >= 1 && n_fake >= 1, "dstep_stats: empty batch");
    hipLaunchKernelGGL(dstep_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d_real, n_real, d_fake, n_fake, real_loss,
                       fake_loss, penalty, acc7);
    return check_launch("dstep_stats_kernel");
}

int cslgan_grad_log_stats_f32(const float* sq, int n_layers, int64_t ld, int64_t col0, int B, const float* max_norm, int per_layer,
                              float eps, float* acc_mean, float* acc_std, float* acc_max, float* acc_c, float* acc_clipped,
                              void* stream) {
    CSLGAN_REQUIRE(sq && max_norm && acc_mean && acc_std && acc_max && acc_c && acc_clipped, "grad_log_stats: null argument");
    CSLGAN_REQUIRE(n_layers >= 1 && B >= 1 && col0 >= 0 && col0 + B <= ld, "grad_log_stats: columns [%lld, %lld) outside a row of %lld",
                   (long long)col0, (long long)(col0 + B), (long long)ld);
    hipLaunchKernelGGL(grad_log_stats_kernel, dim3(per_layer ? n_layers : 1), dim3(256), 0, (hipStream_t)stream, sq, n_layers,
                       (long long)ld, (long long)col0, B, max_norm, per_layer, eps, acc_mean, acc_std, acc_max, acc_c, acc_clipped);
    return check_launch("grad_log_stats_kernel");
}

int cslgan_lerp_rows_f32(const float* real, const float* fake, const float* alpha, int64_t n_rows, int64_t len, float* out,
                         void* stream) {
    CSLGAN_REQUIRE(real && fake && alpha && out, "lerp_rows: null argument");
    CSLGAN_REQUIRE(n_rows >= 0 && n_rows <= 65535 && len >= 0, "lerp_rows: bad sizes");
    if (n_rows == 0 || len == 0) return CSLGAN_OK;
    CSLGAN_REQUIRE((len & 3) != 0 || (aligned16(real) && aligned16(fake) && aligned16(out)), "lerp_rows: unaligned buffers");
    unsigned gx = (unsigned)((len + 1023) / 1024);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(lerp_rows_kernel, dim3(gx, (unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, real, fake, alpha,
                       (long long)len, out);
    return check_launch("lerp_rows_kernel");
}

int cslgan_lipschitz_term_f32(const float* t, int64_t n_rows, int64_t len, int one_sided, float coef, float* norm, float* per,
                              float* total, uint32_t* ticket, void* stream) {
    CSLGAN_REQUIRE(t && norm && per && ticket, "lipschitz_term: null argument");
    CSLGAN_REQUIRE(n_rows >= 1 && n_rows <= 65535 && len >= 1, "lipschitz_term: bad sizes");
    hipLaunchKernelGGL(lipschitz_term_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, t, (long long)len, one_sided,
                       coef, norm, per, total, ticket);
    return check_launch("lipschitz_term_kernel");
}

int cslgan_lipschitz_term_bwd_f32(const float* t, const float* norm, const float* g_total, const float* g_per, int64_t n_rows,
                                  int64_t len, int one_sided, float coef, float* gt, void* stream) {
    CSLGAN_REQUIRE(t && norm && gt, "lipschitz_term_bwd: null argument");
    CSLGAN_REQUIRE(g_total || g_per, "lipschitz_term_bwd: no incoming gradient");
    CSLGAN_REQUIRE(n_rows >= 1 && n_rows <= 65535 && len >= 1, "lipschitz_term_bwd: bad sizes");
    unsigned gx = (unsigned)((len + 1023) / 1024);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(lipschitz_term_bwd_kernel, dim3(gx, (unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, t, norm, g_total,
                       g_per, (long long)len, one_sided, coef, gt);
    return check_launch("lipschitz_term_bwd_kernel");
}

int cslgan_adaptive_clip_f32(const cslgan_adaptive_clip_t* a, int64_t n_adapt, int64_t n_rows, int stat_max, float scalar, int per_layer,
                             float eps, int64_t first_private_row, float* r_out, float* c_out, float* sq_out, float* f_out, float* f_mat,
                             void* stream) {
    CSLGAN_REQUIRE(a && r_out && c_out && sq_out && f_out, "adaptive_clip: null argument");
    CSLGAN_REQUIRE(a->n_layers >= 1 && a->n_layers <= CSLGAN_MAX_CLIP_LAYERS && a->n_mat >= 0 && a->n_mat <= CSLGAN_MAX_CLIP_LAYERS &&
                   a->n_jobs >= 0 && a->n_jobs <= CSLGAN_MAX_CLIP_JOBS && n_adapt >= 1 && n_rows >= 1,
                   "adaptive_clip: sizes out of range (%d layers, %d gathered, %d jobs)", a->n_layers, a->n_mat, a->n_jobs);
    for (int l = 0; l < a->n_layers; ++l) CSLGAN_REQUIRE(a->sq_adapt[l] && a->sq_rows[l], "adaptive_clip: layer %d has no norms", l);
    for (int m = 0; m < a->n_mat; ++m) CSLGAN_REQUIRE(a->mat_layer[m] >= 0 && a->mat_layer[m] < a->n_layers, "adaptive_clip: bad gathered layer");
    for (int j = 0; j < a->n_jobs; ++j)
        CSLGAN_REQUIRE(a->job_dst[j] && a->job_count[j] >= 0 && a->job_first[j] >= 0 && a->job_first[j] + a->job_count[j] <= n_rows &&
                       a->job_layer[j] >= 0 && a->job_layer[j] < a->n_layers, "adaptive_clip: bad row-weight job %d", j);
    CSLGAN_REQUIRE(a->n_mat == 0 || (f_mat && per_layer), "adaptive_clip: gathered factor rows need f_mat and per-layer clipping");
    hipLaunchKernelGGL(adaptive_clip_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *a, (long long)n_adapt, (long long)n_rows, stat_max,
                       scalar, per_layer, eps, (long long)first_private_row, r_out, c_out, sq_out, f_out, f_mat);
    return check_launch("adaptive_clip_kernel");
}

int cslgan_adam_multi_f32(int n_seg, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                          float lr, float b1, float b2, float eps, float weight_decay, int step, const int32_t* step_dev,
                          void* stream) {
    CSLGAN_REQUIRE(p && g && m && v && n, "adam_multi: null argument");
    CSLGAN_REQUIRE(n_seg >= 1 && n_seg <= CSLGAN_MAX_SEGS, "adam_multi: n_seg=%d outside [1, %d]", n_seg, CSLGAN_MAX_SEGS);
    CSLGAN_REQUIRE(step_dev || step >= 1, "adam_multi: step must be >= 1");
    AdamArgs a;
    a.n_seg = n_seg;
    int tot = 0;
    for (int s = 0; s < n_seg; ++s) {
        CSLGAN_REQUIRE(p[s] && g[s] && m[s] && v[s] && n[s] >= 0, "adam_multi: bad tensor %d", s);
        a.p[s] = p[s]; a.g[s] = g[s]; a.m[s] = m[s]; a.v[s] = v[s]; a.n[s] = n[s];
        a.vec_ok[s] = aligned16(p[s]) && aligned16(g[s]) && aligned16(m[s]) && aligned16(v[s]);
        a.chunk_prefix[s] = tot;
        tot += (int)((n[s] + ADAM_CHUNK - 1) / ADAM_CHUNK);
    }
    for (int s = n_seg; s < CSLGAN_MAX_SEGS; ++s) { a.p[s] = nullptr; a.g[s] = nullptr; a.m[s] = nullptr; a.v[s] = nullptr; a.n[s] = 0; a.vec_ok[s] = 0; }
    for (int s = n_seg; s <= CSLGAN_MAX_SEGS; ++s) a.chunk_prefix[s] = tot;
    if (tot == 0) return CSLGAN_OK;
    double bc1 = 1.0, bc2 = 1.0;
    if (!step_dev) {
        bc1 = 1.0 - pow((double)b1, (double)step);
        bc2 = 1.0 - pow((double)b2, (double)step);
    }
    hipLaunchKernelGGL(adam_multi_kernel, dim3(tot), dim3(256), 0, (hipStream_t)stream, a, lr, b1, b2, eps, weight_decay, (float)bc1,
                       (float)sqrt(bc2), step_dev);
    return check_launch("adam_multi_kernel");
}

}  // extern "C"
