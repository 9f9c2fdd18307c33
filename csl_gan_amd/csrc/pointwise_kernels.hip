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

// ---- normalisation statistics over NHWC data ------------------------------------------------
// x is [R][C] (R = N*H*W rows).  A statistic s covers `rows_per_stat` consecutive rows and `cpg`
// consecutive channels:  s = (row / rows_per_stat) * n_groups + c / cpg.
//   GroupNorm(G): rows_per_stat = H*W, cpg = C/G, n_groups = G        (per image, per group)
//   BatchNorm   : rows_per_stat = R,   cpg = 1,   n_groups = C        (per channel over the batch)
// ONE pass over x accumulates shifted moments  S1 = sum(x - k_s),  S2 = sum((x - k_s)^2)  with the shift k_s = the
// statistic's first element (x[first row of s][first channel of s]): mean = k + S1/n, var = S2/n - (S1/n)^2.
// The shift is a sample of the same distribution, so the subtraction loses about one bit instead of the
// catastrophic cancellation of raw E[x^2] - mean^2, and x is read once instead of twice.  A tiny finalize
// kernel turns (S1, S2) into the (sum, centred sum of squares) pair the apply / backward kernels consume.
// Rows are read with 16-byte loads, coalesced along C; partial sums go thread -> LDS (per channel) -> one
// global atomic per channel per block.
constexpr int NS_ROWS_MIN = 64;   // rows per block, lower bound (the launch picks a multiple: ~1024 blocks in all)

// PART: `final_stats` is the partials array [row group][workgroup of the row group][group][2] (see launch_norm): every workgroup
// leaves its own (S1, S2) pairs there and the apply kernel adds them — no zeroed accumulator, no atomics, no finalize launch.
// (Round 3's attempt to finish the statistics in the LAST workgroup of a row group behind a ticket needed a device-scope fence in
// every workgroup and made this kernel 5x slower; it is gone.)
// Element access for the normalisation kernels: fp32 tensors, or bfloat16-stored activations (csrc/igemm_bf16s.hip's storage mode;
// statistics and arithmetic are fp32 either way, a bf16 output is rounded to nearest even).
typedef unsigned short bf16_t;
__device__ __forceinline__ float ldf(const float* p, long long i) { return p[i]; }
__device__ __forceinline__ float ldf(const bf16_t* p, long long i) { return __uint_as_float((unsigned)p[i] << 16); }
__device__ __forceinline__ float4 ld4(const float* p, long long i) { return *reinterpret_cast<const float4*>(p + i); }
__device__ __forceinline__ float4 ld4(const bf16_t* p, long long i) {
    const uint2 v = *reinterpret_cast<const uint2*>(p + i);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}
__device__ __forceinline__ unsigned rne_bf16(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ void st1(float* p, long long i, float v) { p[i] = v; }
__device__ __forceinline__ void st1(bf16_t* p, long long i, float v) { p[i] = (bf16_t)rne_bf16(v); }
__device__ __forceinline__ void st4(float* p, long long i, float a, float b, float c, float d) { *reinterpret_cast<float4*>(p + i) = make_float4(a, b, c, d); }
__device__ __forceinline__ void st4(bf16_t* p, long long i, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p + i) = make_uint2(rne_bf16(a) | (rne_bf16(b) << 16), rne_bf16(c) | (rne_bf16(d) << 16));
}

template <bool VEC, bool PART, typename TX>
__global__ __launch_bounds__(256) void norm_stats_kernel(const TX* __restrict__ x, long long rows_per_stat, int C, int cpg,
                                                         int n_groups, int rows_per_block, float* __restrict__ stats,
                                                         float* __restrict__ final_stats, long long n_stats) {
    extern __shared__ float s_acc[];   // [2*C]: S1, S2 per channel
    const int tid = threadIdx.x;
    for (int c = tid; c < 2 * C; c += 256) s_acc[c] = 0.f;
    __syncthreads();
    const long long sr = blockIdx.y;                      // which row-group of statistics
    const long long row_first = sr * rows_per_stat;
    const long long r0 = row_first + (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    const long long rend = row_first + rows_per_stat;
    if (r1 > rend) r1 = rend;
    if (VEC) {
        const int c4n = C >> 2;
        const int c4 = tid % c4n, rl = tid / c4n, rstep = 256 / c4n;
        float k[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) k[e] = ldf(x, row_first * C + ((4 * c4 + e) / cpg) * cpg);
        float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
        long long r = r0 + rl;
        for (; r + 3 * rstep < r1; r += 4 * rstep) {      // four independent 16-byte loads in flight
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ld4(x, (r + u * rstep) * C + 4 * c4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d[4] = {v[u].x - k[0], v[u].y - k[1], v[u].z - k[2], v[u].w - k[3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) { a1[e] += d[e]; a2[e] = fmaf(d[e], d[e], a2[e]); }
            }
        }
        for (; r < r1; r += rstep) {
            const float4 v = ld4(x, r * C + 4 * c4);
            const float d[4] = {v.x - k[0], v.y - k[1], v.z - k[2], v.w - k[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) { a1[e] += d[e]; a2[e] = fmaf(d[e], d[e], a2[e]); }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(&s_acc[2 * (4 * c4 + e)], a1[e]);
            atomicAdd(&s_acc[2 * (4 * c4 + e) + 1], a2[e]);
        }
    } else {
        const long long n = (r1 - r0) * C;
        for (long long i = tid; i < n; i += 256) {
            const int c = (int)(i % C);
            const float d = ldf(x, r0 * C + i) - ldf(x, row_first * C + (c / cpg) * cpg);
            atomicAdd(&s_acc[2 * c], d);
            atomicAdd(&s_acc[2 * c + 1], d * d);
        }
    }
    __syncthreads();
    // the channels of a group share the shift: fold them in LDS; then ONE value per (group, moment) leaves the workgroup —
    // PART: written to this workgroup's slot of the partials array (no atomics, no zeroed accumulator: the apply kernel adds the
    // <= 64 slots of a row group in its prologue); otherwise one global atomic into the zeroed statistics
    for (int i = tid; i < 2 * n_groups; i += 256) {
        const int g = i >> 1, which = i & 1;
        float t = 0.f;
        for (int j = 0; j < cpg; ++j) t += s_acc[2 * (g * cpg + j) + which];
        if (PART) final_stats[((sr * gridDim.x + blockIdx.x) * n_groups + g) * 2 + which] = t;
        else atomicAdd(&stats[2 * (sr * n_groups + g) + which], t);
    }
}

// (S1, S2) about the shift k  ->  (sum x, sum (x-mean)^2)
template <typename TX>
__global__ void norm_stats_finalize_kernel(const TX* __restrict__ x, long long rows_per_stat, int C, int cpg, int n_groups,
                                           long long n_stats, float* __restrict__ stats) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_stats) return;
    const long long sr = s / n_groups;
    const int g = (int)(s - sr * n_groups);
    const float k = ldf(x, sr * rows_per_stat * C + g * cpg);
    const float cnt = (float)rows_per_stat * (float)cpg;
    const float S1 = stats[2 * s], S2 = stats[2 * s + 1];
    const float m1 = S1 / cnt;
    float css = S2 - S1 * m1;           // sum (x-mean)^2 = S2 - S1^2/n
    if (css < 0.f) css = 0.f;
    stats[2 * s] = (k + m1) * cnt;      // sum x
    stats[2 * s + 1] = css;
}

// Depth-to-space addressing of UpsampleConv (DCResNet_models.py:13-15): cat([x]*4, dim=1) + pixel_shuffle(2) gives
// up[c][2h+i][2w+j] = x[(4c+2i+j) mod C][h][w], i.e. for C % 4 == 0 the C output channels are the C/4 channels of the
// plain depth-to-space tensor  ps[n][2h+i][2w+j][c'] = x[n][h][w][4c'+2i+j]  repeated four times (the conv that follows
// runs on ps with its filter summed over the four channel groups, cslgan_fold_channels4_f32).  Element (row r, channel
// 4c'+p) of x[N*H*W][C] lands at ps offset d2s_offset(r, p) + c'.
__device__ __forceinline__ long long d2s_offset(long long r, int p, int H, int W, int Cq) {
    const int w = (int)(r % W);
    const long long t = r / W;
    const int h = (int)(t % H);
    const long long n = t / H;
    return ((n * 2 * H + 2 * h + (p >> 1)) * 2 * W + 2 * w + (p & 1)) * Cq;
}

// y = act((x - mean_s) * rstd_s * gamma[c] + beta[c])
// D2S: y (and xs, the raw input, when given) are written in the depth-to-space layout [N][2H][2W][C/4].
template <bool VEC, bool D2S, typename TX, typename TY>
__global__ void norm_apply_kernel(const TX* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                  const float* __restrict__ stats, long long total, long long rows_per_stat, int C, int cpg,
                                  int n_groups, float eps, int relu, TY* __restrict__ y, int H, int W, TY* __restrict__ xs) {
    const float inv_cnt = 1.f / ((float)rows_per_stat * (float)cpg);
    const long long per_stat_elems = rows_per_stat * C;
    const long long stride = (long long)gridDim.x * blockDim.x;
    if (VEC) {
        const long long n4 = total >> 2;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            const long long e0 = i << 2;
            const int c = (int)(e0 % C);
            const long long sr = e0 / per_stat_elems;
            const float4 v = ld4(x, e0);
            float in[4] = {v.x, v.y, v.z, v.w}, o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long long s = sr * n_groups + (c + e) / cpg;
                const float mean = stats[2 * s] * inv_cnt;
                const float rstd = rsqrtf(stats[2 * s + 1] * inv_cnt + eps);
                float t = (in[e] - mean) * rstd * gamma[c + e] + beta[c + e];
                o[e] = (relu && t < 0.f) ? 0.f : t;
            }
            if (D2S) {
                const long long r = e0 / C;
                const int Cq = C >> 2, cq = c >> 2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {          // lanes hold consecutive c': each of the four stores is coalesced
                    const long long off = d2s_offset(r, e, H, W, Cq) + cq;
                    st1(y, off, o[e]);
                    if (xs) st1(xs, off, in[e]);
                }
            } else {
                st4(y, e0, o[0], o[1], o[2], o[3]);
            }
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const int c = (int)(i % C);
            const long long s = (i / per_stat_elems) * n_groups + c / cpg;
            const float mean = stats[2 * s] * inv_cnt;
            const float rstd = rsqrtf(stats[2 * s + 1] * inv_cnt + eps);
            const float xi = ldf(x, i);
            float t = (xi - mean) * rstd * gamma[c] + beta[c];
            t = (relu && t < 0.f) ? 0.f : t;
            if (D2S) {
                const long long off = d2s_offset(i / C, c & 3, H, W, C >> 2) + (c >> 2);
                st1(y, off, t);
                if (xs) st1(xs, off, xi);
            } else {
                st1(y, i, t);
            }
        }
    }
}

// The vector path of norm_apply_kernel with the index arithmetic taken out of the loop.  grid = (slices, statistic row groups): a
// workgroup walks a slice of ONE row group (an image for GroupNorm), and since 256 % (C/4) == 0 a thread keeps the same four
// channels for the whole walk — mean, rstd, gamma and beta are loop constants and an element costs a load, four fused operations
// and a store.  norm_apply_kernel<true, ...> spent two 64-bit divisions, four 32-bit ones, eight statistic loads and four rsqrt per
// four elements: 206 us for the generator's 128x128x64 bfloat16 tensors (537 MB, 2.6 TB/s) — arithmetic, not memory.
template <bool D2S, typename TX, typename TY>
__global__ __launch_bounds__(256) void norm_apply_rows_kernel(const TX* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ stats, long long rows_per_stat, int C, int cpg, int n_groups,
                                                              float eps, int relu, TY* __restrict__ y, int H, int W, TY* __restrict__ xs,
                                                              const float* __restrict__ part, int n_part, float* __restrict__ stats_out, int chan) {
    const int c4n = C >> 2;                           // a power of two (256 % c4n == 0)
    const int lg = __ffs(c4n) - 1;
    const int c = (threadIdx.x & (c4n - 1)) << 2;
    const long long sr = blockIdx.y;
    const float cnt = (float)rows_per_stat * (float)cpg;
    const float inv_cnt = 1.f / cnt;
    const long long base = sr * rows_per_stat * C;
    float mean[4], rstd[4], ga[4], be[4];
    if (part) {
        // Two-launch form (round 4): the statistics kernel left one (S1, S2) pair about the shift k per workgroup and group; add the
        // row group's n_part slots here (2 * n_groups threads, n_part loads each), finish them in LDS, and let the row group's first
        // workgroup publish the final (sum x, centred sum of squares) pairs for whoever reads the statistics later (the backward).
        extern __shared__ float s_st[];               // [2 * n_groups]: (sum x, centred sum of squares)
        for (int i = threadIdx.x; chan && i < n_groups; i += 256) {
            // chan: the partials come from a conv epilogue (cslgan_conv_t.gn_part): per slot the sum of its cnt / n_part values and
            // their sum of squares about the slot's own mean — combined exactly: M2 = sum_b M2_b + n_b (mean_b - mean)^2
            float sum = 0.f;
            for (int b = 0; b < n_part; ++b) sum += part[((sr * n_part + b) * n_groups + i) * 2];
            const float nb = cnt / (float)n_part, mean_all = sum / cnt;
            float css = 0.f;
            for (int b = 0; b < n_part; ++b) {
                const float* q = part + ((sr * n_part + b) * n_groups + i) * 2;
                const float dm = q[0] / nb - mean_all;
                css += q[1] + nb * dm * dm;
            }
            s_st[2 * i] = sum;
            s_st[2 * i + 1] = css;
            if (stats_out && blockIdx.x == 0) {
                stats_out[2 * (sr * n_groups + i)] = sum;
                stats_out[2 * (sr * n_groups + i) + 1] = css;
            }
        }
        for (int i = threadIdx.x; !chan && i < n_groups; i += 256) {
            float S1 = 0.f, S2 = 0.f;
            for (int b = 0; b < n_part; ++b) {
                const float* q = part + ((sr * n_part + b) * n_groups + i) * 2;
                S1 += q[0];
                S2 += q[1];
            }
            const float k = ldf(x, base + (long long)i * cpg);
            const float m1 = S1 / cnt;
            float css = S2 - S1 * m1;
            css = css < 0.f ? 0.f : css;
            s_st[2 * i] = (k + m1) * cnt;
            s_st[2 * i + 1] = css;
            if (stats_out && blockIdx.x == 0) {
                stats_out[2 * (sr * n_groups + i)] = (k + m1) * cnt;
                stats_out[2 * (sr * n_groups + i) + 1] = css;
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int g = (c + e) / cpg;
            mean[e] = s_st[2 * g] * inv_cnt;
            rstd[e] = rsqrtf(s_st[2 * g + 1] * inv_cnt + eps);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long s = sr * n_groups + (c + e) / cpg;
            mean[e] = stats[2 * s] * inv_cnt;
            rstd[e] = rsqrtf(stats[2 * s + 1] * inv_cnt + eps);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        ga[e] = gamma[c + e];
        be[e] = beta[c + e];
    }
    const unsigned n4 = (unsigned)((rows_per_stat * C) >> 2);          // launcher: rows_per_stat * C < 2^32
    const bool per_image = rows_per_stat == (long long)H * W;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
        const long long e0 = base + ((long long)i << 2);
        const float4 v = ld4(x, e0);
        const float in[4] = {v.x, v.y, v.z, v.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = (in[e] - mean[e]) * rstd[e] * ga[e] + be[e];
            o[e] = (relu && t < 0.f) ? 0.f : t;
        }
        if (D2S) {
            const unsigned rl = i >> lg;              // row within the group
            const int Cq = C >> 2, cq = c >> 2;
            long long off[4];
            if (per_image) {
                const unsigned h = rl / (unsigned)W, w = rl - h * (unsigned)W;
#pragma unroll
                for (int e = 0; e < 4; ++e) off[e] = (((sr * 2 * H + 2 * h + (e >> 1)) * 2 * W + 2 * w + (e & 1)) * Cq) + cq;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) off[e] = d2s_offset(sr * rows_per_stat + rl, e, H, W, Cq) + cq;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {             // lanes hold consecutive c': each of the four stores is coalesced
                st1(y, off[e], o[e]);
                if (xs) st1(xs, off[e], in[e]);
            }
        } else {
            st4(y, e0, o[0], o[1], o[2], o[3]);
        }
    }
}

// ps[n][2h+i][2w+j][c'] = x[n][h][w][4c'+2i+j]  (INVERSE: x from ps — the backward of the forward and vice versa)
template <bool INVERSE>
__global__ void depth_to_space_kernel(const float* __restrict__ in, long long rows, int H, int W, int C, float* __restrict__ out) {
    const int Cq = C >> 2;
    const long long n4 = rows * Cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / Cq;
        const int cq = (int)(i - r * Cq);
        if (INVERSE) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = in[d2s_offset(r, e, H, W, Cq) + cq];
            reinterpret_cast<float4*>(out)[i] = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            const float4 v = reinterpret_cast<const float4*>(in)[i];
            const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) out[d2s_offset(r, e, H, W, Cq) + cq] = a[e];
        }
    }
}

// The conv after the depth-to-space sees four identical channel groups, so it equals a conv over C/4 channels with
//   wf[row][c'] = sum_q w[row][c' + q*C/4]      (row = (k, r, s) of a KRSC filter)
// UNFOLD is its transpose (the gradient of w from the gradient of wf): gw[row][c' + q*C/4] = gwf[row][c'].
template <bool UNFOLD>
__global__ void fold_channels4_kernel(const float* __restrict__ in, long long rows, int C, float* __restrict__ out) {
    const int Cq = C >> 2;
    const long long n = rows * Cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / Cq;
        const int cq = (int)(i - r * Cq);
        if (UNFOLD) {
            const float v = in[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) out[r * C + q * Cq + cq] = v;
        } else {
            const float* b = in + r * C + cq;
            out[i] = (b[0] + b[Cq]) + (b[2 * Cq] + b[3 * Cq]);
        }
    }
}

// eval-mode BatchNorm: the (sum, centred sum of squares) pair norm_apply_kernel consumes, from the running statistics
__global__ void bn_eval_stats_kernel(const float* __restrict__ rm, const float* __restrict__ rv, int C, float cnt,
                                     float* __restrict__ stats) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    stats[2 * c] = rm[c] * cnt;
    stats[2 * c + 1] = rv[c] * cnt;
}

// ---- normalisation backward (GroupNorm / BatchNorm + optional ReLU) -------------------------------
// With xh = (x-mean_s)*rstd_s, dy' = dy * [y > 0] (ReLU fused; y is the forward output):
//   per (row-group sr, channel c):  A = sum_rows dy',  Bq = sum_rows dy' * xh        -> ab[(sr*C + c)*2 + {0,1}]
//   d gamma[c] = sum_sr Bq,  d beta[c] = sum_sr A
//   per statistic s: s1 = sum_{c in s} gamma_c A, s2 = sum_{c in s} gamma_c Bq
//   dx = rstd_s * (gamma_c dy' - s1/cnt - xh * s2/cnt)
template <bool VEC>
__global__ __launch_bounds__(256) void norm_bwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ y, const float* __restrict__ stats,
                                                             long long rows_per_stat, int C, int cpg, int n_groups, float eps,
                                                             int relu, int rows_per_block, float* __restrict__ ab) {
    extern __shared__ float s_acc[];   // [2*C]
    const int tid = threadIdx.x;
    for (int c = tid; c < 2 * C; c += 256) s_acc[c] = 0.f;
    __syncthreads();
    const long long sr = blockIdx.y;
    const long long r0 = sr * rows_per_stat + (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    const long long rend = (sr + 1) * rows_per_stat;
    if (r1 > rend) r1 = rend;
    const float inv_cnt = 1.f / ((float)rows_per_stat * (float)cpg);
    if (VEC) {
        const int c4n = C >> 2;
        const int c4 = tid % c4n, rl = tid / c4n, rstep = 256 / c4n;
        float mean[4], rstd[4], a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long s = sr * n_groups + (4 * c4 + e) / cpg;
            mean[e] = stats[2 * s] * inv_cnt;
            rstd[e] = rsqrtf(stats[2 * s + 1] * inv_cnt + eps);
        }
        for (long long r = r0 + rl; r < r1; r += rstep) {
            const float4 xv = *reinterpret_cast<const float4*>(x + r * C + 4 * c4);
            const float4 gv = *reinterpret_cast<const float4*>(dy + r * C + 4 * c4);
            float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
            if (relu) yv = *reinterpret_cast<const float4*>(y + r * C + 4 * c4);
            const float xi[4] = {xv.x, xv.y, xv.z, xv.w}, gi[4] = {gv.x, gv.y, gv.z, gv.w}, yi[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g = yi[e] > 0.f ? gi[e] : 0.f;
                a[e] += g;
                b[e] = fmaf(g, (xi[e] - mean[e]) * rstd[e], b[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(&s_acc[2 * (4 * c4 + e)], a[e]);
            atomicAdd(&s_acc[2 * (4 * c4 + e) + 1], b[e]);
        }
    } else {
        const long long n = (r1 - r0) * C;
        for (long long i = tid; i < n; i += 256) {
            const int c = (int)(i % C);
            const long long s = sr * n_groups + c / cpg;
            const float m = stats[2 * s] * inv_cnt, rs = rsqrtf(stats[2 * s + 1] * inv_cnt + eps);
            const long long off = r0 * C + i;
            const float g = (!relu || y[off] > 0.f) ? dy[off] : 0.f;
            atomicAdd(&s_acc[2 * c], g);
            atomicAdd(&s_acc[2 * c + 1], g * (x[off] - m) * rs);
        }
    }
    __syncthreads();
    for (int c = tid; c < 2 * C; c += 256) atomicAdd(&ab[sr * 2 * C + c], s_acc[c]);
}

// gs[s] = (s1, s2);  dgamma/dbeta accumulate over row groups
__global__ void norm_bwd_finalize_kernel(const float* __restrict__ ab, const float* __restrict__ gamma, long long n_row_groups,
                                         int C, int cpg, int n_groups, float* __restrict__ gs, float* __restrict__ dgamma,
                                         float* __restrict__ dbeta) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n_stats = n_row_groups * n_groups;
    if (i < n_stats) {
        const long long sr = i / n_groups;
        const int g = (int)(i - sr * n_groups);
        float s1 = 0.f, s2 = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            s1 = fmaf(gamma[c], ab[(sr * C + c) * 2], s1);
            s2 = fmaf(gamma[c], ab[(sr * C + c) * 2 + 1], s2);
        }
        gs[2 * i] = s1;
        gs[2 * i + 1] = s2;
    }
    if (i < C) {
        float da = 0.f, db = 0.f;
        for (long long sr = 0; sr < n_row_groups; ++sr) {
            da += ab[(sr * C + i) * 2];
            db += ab[(sr * C + i) * 2 + 1];
        }
        dbeta[i] = da;
        dgamma[i] = db;
    }
}

__global__ void norm_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
                                      const float* __restrict__ gamma, const float* __restrict__ stats, const float* __restrict__ gs,
                                      long long total, long long rows_per_stat, int C, int cpg, int n_groups, float eps, int relu,
                                      float* __restrict__ dx) {
    const float inv_cnt = 1.f / ((float)rows_per_stat * (float)cpg);
    const long long per_stat_elems = rows_per_stat * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long s = (i / per_stat_elems) * n_groups + c / cpg;
        const float mean = stats[2 * s] * inv_cnt, rstd = rsqrtf(stats[2 * s + 1] * inv_cnt + eps);
        const float xh = (x[i] - mean) * rstd;
        const float g = (!relu || y[i] > 0.f) ? dy[i] : 0.f;
        dx[i] = rstd * (gamma[c] * g - gs[2 * s] * inv_cnt - xh * gs[2 * s + 1] * inv_cnt);
    }
}

// BatchNorm running statistics (torch semantics: running_var uses the unbiased batch variance)
__global__ void bn_running_kernel(const float* __restrict__ stats, int C, float cnt, float momentum, float* __restrict__ rm,
                                  float* __restrict__ rv) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = stats[2 * c] / cnt;
    const float var_unb = cnt > 1.f ? stats[2 * c + 1] / (cnt - 1.f) : 0.f;
    rm[c] = (1.f - momentum) * rm[c] + momentum * mean;
    rv[c] = (1.f - momentum) * rv[c] + momentum * var_unb;
}

template <typename TX, typename TY>
static int launch_norm_apply(const TX* x, const float* gamma, const float* beta, long long R, long long rows_per_stat, int C,
                             int cpg, int n_groups, float eps, int relu, float* stats, TY* y, int d2s_H, int d2s_W,
                             TY* xs, bool vec, hipStream_t st, const float* part = nullptr, int n_part = 0, int chan = 0) {
    const long long total = R * C;
    static const int rows_env = [] { const char* e = getenv("CSLGAN_NORM_ROWS"); return e ? atoi(e) : 1; }();
    const long long n_rg = R / rows_per_stat;
    if (vec && rows_env && rows_per_stat * C < (1ll << 32) && n_rg <= 65535) {
        long long bx = (rows_per_stat * C / 4 + 255) / 256;
        const long long cap = (4096 + n_rg - 1) / n_rg;                 // about 4096 workgroups in all
        bx = bx > cap ? cap : (bx < 1 ? 1 : bx);
        const dim3 g2((unsigned)bx, (unsigned)n_rg), b2(256);
        const size_t lds = part ? sizeof(float) * 2 * n_groups : 0;
        if (d2s_W > 0) hipLaunchKernelGGL((norm_apply_rows_kernel<true, TX, TY>), g2, b2, lds, st, x, gamma, beta, stats, rows_per_stat, C, cpg, n_groups, eps, relu, y, d2s_H, d2s_W, xs, part, n_part, stats, chan);
        else hipLaunchKernelGGL((norm_apply_rows_kernel<false, TX, TY>), g2, b2, lds, st, x, gamma, beta, stats, rows_per_stat, C, cpg, n_groups, eps, relu, y, 0, 0, (TY*)nullptr, part, n_part, stats, chan);
        return check_launch("norm_apply_rows_kernel");
    }
    long long nb = (total / 4 + 255) / 256;
    nb = nb > 4096 ? 4096 : (nb < 1 ? 1 : nb);
    const dim3 g((unsigned)nb), b(256);
    if (d2s_W > 0) {
        if (vec) hipLaunchKernelGGL((norm_apply_kernel<true, true, TX, TY>), g, b, 0, st, x, gamma, beta, stats, total, rows_per_stat, C, cpg, n_groups, eps, relu, y, d2s_H, d2s_W, xs);
        else hipLaunchKernelGGL((norm_apply_kernel<false, true, TX, TY>), g, b, 0, st, x, gamma, beta, stats, total, rows_per_stat, C, cpg, n_groups, eps, relu, y, d2s_H, d2s_W, xs);
    } else {
        if (vec) hipLaunchKernelGGL((norm_apply_kernel<true, false, TX, TY>), g, b, 0, st, x, gamma, beta, stats, total, rows_per_stat, C, cpg, n_groups, eps, relu, y, 0, 0, (TY*)nullptr);
        else hipLaunchKernelGGL((norm_apply_kernel<false, false, TX, TY>), g, b, 0, st, x, gamma, beta, stats, total, rows_per_stat, C, cpg, n_groups, eps, relu, y, 0, 0, (TY*)nullptr);
    }
    return check_launch("norm_apply_kernel");
}

static bool norm_vec_ok(const void* x, const void* y, const void* xs, int C, int cpg) {
    return (C % 4 == 0) && ((C / 4) <= 256) && (256 % (C / 4) == 0) && (cpg % 4 == 0 || 4 % cpg == 0) && aligned16(x) && aligned16(y) &&
           (!xs || aligned16(xs));
}

template <typename TX, typename TY>
static int launch_norm(const TX* x, const float* gamma, const float* beta, long long R, long long rows_per_stat, int C, int cpg,
                       int n_groups, float eps, int relu, float* stats, TY* y, int d2s_H, int d2s_W, TY* xs, hipStream_t st,
                       float* scratch = nullptr) {
    const long long n_row_groups = R / rows_per_stat;
    const long long n_stats = n_row_groups * n_groups;
    const bool vec = norm_vec_ok(x, y, xs, C, cpg);
    // about 1024 workgroups in all, each with at least NS_ROWS_MIN rows: fewer, longer blocks mean fewer global atomics
    long long rpb = (rows_per_stat * n_row_groups / 1024 + NS_ROWS_MIN - 1) / NS_ROWS_MIN * NS_ROWS_MIN;
    rpb = rpb < NS_ROWS_MIN ? NS_ROWS_MIN : (rpb > 4096 ? 4096 : rpb);
    const dim3 grid((unsigned)((rows_per_stat + rpb - 1) / rpb), (unsigned)n_row_groups), block(256);
    const size_t lds = sizeof(float) * C;
    // Two-launch form: per-workgroup partial statistics into `scratch` (2 * n_stats * grid.x floats, grid.x <= CSLGAN_NORM_PARTIAL_BLOCKS),
    // summed by the apply kernel's prologue — no zeroed accumulator, no global atomics, no finalize launch (round 3: four launches per
    // normalisation, 36 per D-step; its ticket-fused attempt needed a device-scope fence per workgroup and lost).
    static const int part_env = [] { const char* e = getenv("CSLGAN_NORM_PARTIALS"); return e ? atoi(e) : 1; }();
    static const int rows_env2 = [] { const char* e = getenv("CSLGAN_NORM_ROWS"); return e ? atoi(e) : 1; }();
    if (part_env && scratch && vec && rows_env2 && grid.x <= CSLGAN_NORM_PARTIAL_BLOCKS && rows_per_stat * C < (1ll << 32) && n_row_groups <= 65535 &&
        2 * (size_t)n_groups * sizeof(float) <= 32768) {
        hipLaunchKernelGGL((norm_stats_kernel<true, true, TX>), grid, block, 2 * lds, st, x, rows_per_stat, C, cpg, n_groups, (int)rpb, stats, scratch, n_stats);
        const int rc1 = check_launch("norm_stats_kernel");
        if (rc1) return rc1;
        return launch_norm_apply(x, gamma, beta, R, rows_per_stat, C, cpg, n_groups, eps, relu, stats, y, d2s_H, d2s_W, xs, vec, st, scratch, (int)grid.x);
    }
    if (int rc = zero_floats(stats, (size_t)2 * n_stats, st)) return rc;
    if (vec) hipLaunchKernelGGL((norm_stats_kernel<true, false, TX>), grid, block, 2 * lds, st, x, rows_per_stat, C, cpg, n_groups, (int)rpb, stats, (float*)nullptr, n_stats);
    else hipLaunchKernelGGL((norm_stats_kernel<false, false, TX>), grid, block, 2 * lds, st, x, rows_per_stat, C, cpg, n_groups, (int)rpb, stats, (float*)nullptr, n_stats);
    int rc = check_launch("norm_stats_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(norm_stats_finalize_kernel<TX>, dim3((unsigned)((n_stats + 127) / 128)), dim3(128), 0, st, x, rows_per_stat, C, cpg,
                       n_groups, n_stats, stats);
    rc = check_launch("norm_stats_finalize_kernel");
    if (rc) return rc;
    return launch_norm_apply(x, gamma, beta, R, rows_per_stat, C, cpg, n_groups, eps, relu, stats, y, d2s_H, d2s_W, xs, vec, st);
}

// d2s_W > 0 asks for the depth-to-space output layout: rows are (n, h, w) with w fastest, H = rows_per_image / d2s_W
static int check_d2s(long long rows_per_image, int C, int d2s_W, const void* xs, int* H_out) {
    *H_out = 0;
    if (d2s_W <= 0) {
        CSLGAN_REQUIRE(d2s_W == 0 && !xs, "norm: x_shuffled needs d2s_W > 0");
        return CSLGAN_OK;
    }
    CSLGAN_REQUIRE(C % 4 == 0, "norm: depth-to-space output needs C %% 4 == 0 (C=%d)", C);
    CSLGAN_REQUIRE(rows_per_image % d2s_W == 0, "norm: d2s_W=%d does not divide the %lld pixels of an image", d2s_W, rows_per_image);
    *H_out = (int)(rows_per_image / d2s_W);
    return CSLGAN_OK;
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

__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                long long n, float lr, float b1, float b2, float eps, float wd, const int* __restrict__ step_dev) {
    const float st = (float)(*step_dev);
    const float bc1 = 1.f - powf(b1, st), bc2_sqrt = sqrtf(1.f - powf(b2, st));
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
                             float eps, int relu, float* stats_ws, float* y, int d2s_W, float* x_shuffled, float* scratch, void* stream) {
    CSLGAN_REQUIRE(x && gamma && beta && stats_ws && y, "groupnorm: null argument");
    CSLGAN_REQUIRE(N > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0, "groupnorm: C=%d not divisible by groups=%d", C, groups);
    CSLGAN_REQUIRE(N <= 65535 && C <= 8192, "groupnorm: N or C too large");
    int H = 0;
    int rc = check_d2s(HW, C, d2s_W, x_shuffled, &H);
    if (rc) return rc;
    return launch_norm(x, gamma, beta, (long long)N * HW, HW, C, C / groups, groups, eps, relu, stats_ws, y, H, d2s_W, x_shuffled,
                       (hipStream_t)stream, scratch);
}

// (scale, shift) of every (image, channel) from the conv-epilogue partials: one workgroup per image.  The n_part (<= 64) pairs of
// a group are read by 256 / n_groups threads in parallel (a lone thread per channel walking all 64 slots took 23-33 us on the
// generator's last block: latency, not bytes), combined as in the apply kernel's prologue (Chan), in a fixed order.
__global__ __launch_bounds__(256) void gn_affine_parts_kernel(const float* __restrict__ part, int n_part, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int C, int cpg, float cnt, float eps,
                                                              float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ float s_red[256];
    __shared__ float s_mean[256], s_rstd[256];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int n_groups = C / cpg;                       // <= 256 (host)
    const int ways = 256 / n_groups;                    // threads per group
    const int g = tid % n_groups, k = tid / n_groups;
    const float* base = part + (long long)n * n_part * n_groups * 2;
    const float nb = cnt / (float)n_part;
    float sum = 0.f;
    if (k < ways) for (int b = k; b < n_part; b += ways) sum += base[(b * n_groups + g) * 2];
    s_red[tid] = sum;
    __syncthreads();
    if (tid < n_groups) {
        float t = 0.f;
        for (int j = 0; j < ways; ++j) t += s_red[j * n_groups + tid];
        s_mean[tid] = t / cnt;
    }
    __syncthreads();
    float css = 0.f;
    if (k < ways) {
        const float mean = s_mean[g];
        for (int b = k; b < n_part; b += ways) {
            const float* q = base + (b * n_groups + g) * 2;
            const float dm = q[0] / nb - mean;
            css += q[1] + nb * dm * dm;
        }
    }
    s_red[tid] = css;
    __syncthreads();
    if (tid < n_groups) {
        float t = 0.f;
        for (int j = 0; j < ways; ++j) t += s_red[j * n_groups + tid];
        s_rstd[tid] = rsqrtf(t / cnt + eps);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const float a = gamma[c] * s_rstd[c / cpg];
        scale[(long long)n * C + c] = a;
        shift[(long long)n * C + c] = beta[c] - s_mean[c / cpg] * a;
    }
}

int cslgan_groupnorm_affine_parts_f32(const float* part, int n_part, const float* gamma, const float* beta, int N, int HW, int C, int groups,
                                      float eps, float* scale, float* shift, void* stream) {
    CSLGAN_REQUIRE(part && gamma && beta && scale && shift, "groupnorm_affine_parts: null argument");
    CSLGAN_REQUIRE(N > 0 && HW > 0 && C > 0 && groups > 0 && groups <= 256 && C % groups == 0 && n_part >= 1 && HW == 64 * n_part,
                   "groupnorm_affine_parts: needs C %% groups == 0, groups <= 256 and HW == 64 * n_part");
    hipLaunchKernelGGL(gn_affine_parts_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, part, n_part,
                       gamma, beta, C, C / groups, (float)HW * (float)(C / groups), eps, scale, shift);
    return check_launch("gn_affine_parts_kernel");
}

// The apply half on statistics left by a conv epilogue (include/cslgan.h: cslgan_conv_t.gn_part).
int cslgan_groupnorm_apply_parts_f32(const float* x, const float* gamma, const float* beta, int N, int HW, int C, int groups, float eps, int relu,
                                     const float* part, int n_part, float* stats_ws, float* y, int d2s_W, float* x_shuffled, void* stream) {
    CSLGAN_REQUIRE(x && gamma && beta && stats_ws && y && part, "groupnorm_apply_parts: null argument");
    CSLGAN_REQUIRE(N > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0, "groupnorm_apply_parts: C=%d not divisible by groups=%d", C, groups);
    CSLGAN_REQUIRE(N <= 65535 && C <= 8192 && n_part >= 1 && n_part <= CSLGAN_NORM_PARTIAL_BLOCKS && HW == 64 * n_part,
                   "groupnorm_apply_parts: needs HW == 64 * n_part <= %d * 64", CSLGAN_NORM_PARTIAL_BLOCKS);
    int H = 0;
    int rc = check_d2s(HW, C, d2s_W, x_shuffled, &H);
    if (rc) return rc;
    static const int rows_env = [] { const char* e = getenv("CSLGAN_NORM_ROWS"); return e ? atoi(e) : 1; }();
    CSLGAN_REQUIRE(norm_vec_ok(x, y, x_shuffled, C, C / groups) && rows_env && (long long)HW * C < (1ll << 32) && 2 * (size_t)groups * sizeof(float) <= 32768,
                   "groupnorm_apply_parts: shape not taken by the row-walking apply kernel");
    return launch_norm_apply(x, gamma, beta, (long long)N * HW, HW, C, C / groups, groups, eps, relu, stats_ws, y, H, d2s_W, x_shuffled, true,
                             (hipStream_t)stream, part, n_part, 1);
}

// GroupNorm (+ReLU) at the head of the bf16-stored chain (csrc/igemm_bf16s.hip): x fp32 or bfloat16, y (and x_shuffled) bfloat16;
// statistics and arithmetic in fp32 exactly as cslgan_groupnorm_act_f32.
int cslgan_groupnorm_act_bf16s(const void* x, int x_bf16, const float* gamma, const float* beta, int N, int HW, int C, int groups,
                               float eps, int relu, float* stats_ws, void* y_bf16, int d2s_W, void* x_shuffled_bf16, void* stream) {
    CSLGAN_REQUIRE(x && gamma && beta && stats_ws && y_bf16, "groupnorm_bf16s: null argument");
    CSLGAN_REQUIRE(N > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0, "groupnorm_bf16s: C=%d not divisible by groups=%d", C, groups);
    CSLGAN_REQUIRE(N <= 65535 && C <= 8192, "groupnorm_bf16s: N or C too large");
    int H = 0;
    int rc = check_d2s(HW, C, d2s_W, x_shuffled_bf16, &H);
    if (rc) return rc;
    bf16_t* y = reinterpret_cast<bf16_t*>(y_bf16);
    bf16_t* xs = reinterpret_cast<bf16_t*>(x_shuffled_bf16);
    if (x_bf16)
        return launch_norm(reinterpret_cast<const bf16_t*>(x), gamma, beta, (long long)N * HW, HW, C, C / groups, groups, eps, relu, stats_ws, y, H,
                           d2s_W, xs, (hipStream_t)stream);
    return launch_norm(reinterpret_cast<const float*>(x), gamma, beta, (long long)N * HW, HW, C, C / groups, groups, eps, relu, stats_ws, y, H, d2s_W,
                       xs, (hipStream_t)stream);
}

int cslgan_batchnorm_act_f32(const float* x, const float* gamma, const float* beta, int64_t rows, int C, float eps, int relu,
                             float momentum, float* running_mean, float* running_var, float* stats_ws, float* y,
                             int64_t rows_per_image, int d2s_W, float* x_shuffled, float* scratch, void* stream) {
    CSLGAN_REQUIRE(x && gamma && beta && stats_ws && y, "batchnorm: null argument");
    CSLGAN_REQUIRE(rows > 0 && C > 0 && C <= 8192, "batchnorm: bad sizes");
    CSLGAN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "batchnorm: running_mean/var must both be given or both null");
    int H = 0;
    CSLGAN_REQUIRE(d2s_W == 0 || (rows_per_image > 0 && rows % rows_per_image == 0), "batchnorm: rows_per_image must divide rows");
    int rc = check_d2s(rows_per_image, C, d2s_W, x_shuffled, &H);
    if (rc) return rc;
    rc = launch_norm(x, gamma, beta, rows, rows, C, 1, C, eps, relu, stats_ws, y, H, d2s_W, x_shuffled, (hipStream_t)stream, scratch);
    if (rc) return rc;
    if (running_mean) {
        hipLaunchKernelGGL(bn_running_kernel, dim3((unsigned)((C + 127) / 128)), dim3(128), 0, (hipStream_t)stream, stats_ws, C,
                           (float)rows, momentum, running_mean, running_var);
        return check_launch("bn_running_kernel");
    }
    return CSLGAN_OK;
}

int cslgan_batchnorm_eval_act_f32(const float* x, const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, int64_t rows, int C, float eps, int relu, float* stats_ws, float* y,
                                  int64_t rows_per_image, int d2s_W, float* x_shuffled, void* stream) {
    CSLGAN_REQUIRE(x && gamma && beta && running_mean && running_var && stats_ws && y, "batchnorm_eval: null argument");
    CSLGAN_REQUIRE(rows > 0 && C > 0 && C <= 8192, "batchnorm_eval: bad sizes");
    CSLGAN_REQUIRE(d2s_W == 0 || (rows_per_image > 0 && rows % rows_per_image == 0), "batchnorm_eval: rows_per_image must divide rows");
    int H = 0;
    int rc = check_d2s(rows_per_image, C, d2s_W, x_shuffled, &H);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((unsigned)((C + 127) / 128)), dim3(128), 0, st, running_mean, running_var, C,
                       (float)rows, stats_ws);
    rc = check_launch("bn_eval_stats_kernel");
    if (rc) return rc;
    return launch_norm_apply(x, gamma, beta, rows, rows, C, 1, C, eps, relu, stats_ws, y, H, d2s_W, x_shuffled,
                             norm_vec_ok(x, y, x_shuffled, C, 1), st);
}

// ---- input pipeline (round 4): uint8 NHWC image rows -> normalised fp32 NHWC, with the per-image horizontal flip -------------------
// out[n][h][w][c] = src[n][h][flip[n] ? W-1-w : w][c] * scale + bias     (datasets.py:41-47: ToTensor, RandomHorizontalFlip, Normalize)
__global__ void u8_to_f32_nhwc_kernel(const unsigned char* __restrict__ src, const unsigned char* __restrict__ flip, int N, int H, int W, int C,
                                      float scale, float bias, float* __restrict__ out) {
    const long long total = (long long)N * H * W * C;
    const long long per_img = (long long)H * W * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / per_img;
        long long si = i;
        if (flip && flip[n]) {
            const long long r = i - n * per_img;
            const long long row = r / ((long long)W * C);
            const int wc = (int)(r - row * W * C);
            const int w = wc / C, c = wc - w * C;
            si = n * per_img + row * W * C + (long long)(W - 1 - w) * C + c;
        }
        out[i] = (float)src[si] * scale + bias;
    }
}

int cslgan_u8_to_f32_nhwc(const void* src_u8, const void* flip_u8, int N, int H, int W, int C, float scale, float bias, float* out, void* stream) {
    CSLGAN_REQUIRE(src_u8 && out, "u8_to_f32_nhwc: null argument");
    CSLGAN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "u8_to_f32_nhwc: non-positive dimension");
    const long long total = (long long)N * H * W * C;
    unsigned nb = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(u8_to_f32_nhwc_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const unsigned char*>(src_u8),
                       reinterpret_cast<const unsigned char*>(flip_u8), N, H, W, C, scale, bias, out);
    return check_launch("u8_to_f32_nhwc_kernel");
}

int cslgan_depth_to_space_f32(const float* x, int N, int H, int W, int C, int inverse, float* y, void* stream) {
    CSLGAN_REQUIRE(x && y, "depth_to_space: null argument");
    CSLGAN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "depth_to_space: needs C %% 4 == 0 (C=%d)", C);
    CSLGAN_REQUIRE(aligned16(inverse ? y : x), "depth_to_space: the [N,H,W,C] tensor must be 16-byte aligned");
    const long long rows = (long long)N * H * W;
    const dim3 g(grid_for(rows * (C / 4))), b(256);
    if (inverse) hipLaunchKernelGGL((depth_to_space_kernel<true>), g, b, 0, (hipStream_t)stream, x, rows, H, W, C, y);
    else hipLaunchKernelGGL((depth_to_space_kernel<false>), g, b, 0, (hipStream_t)stream, x, rows, H, W, C, y);
    return check_launch("depth_to_space_kernel");
}

int cslgan_fold_channels4_f32(const float* in, int64_t rows, int C, int unfold, float* out, void* stream) {
    CSLGAN_REQUIRE(in && out, "fold_channels4: null argument");
    CSLGAN_REQUIRE(rows > 0 && C > 0 && C % 4 == 0, "fold_channels4: needs C %% 4 == 0 (C=%d)", C);
    const dim3 g(grid_for((long long)rows * (C / 4))), b(256);
    if (unfold) hipLaunchKernelGGL((fold_channels4_kernel<true>), g, b, 0, (hipStream_t)stream, in, (long long)rows, C, out);
    else hipLaunchKernelGGL((fold_channels4_kernel<false>), g, b, 0, (hipStream_t)stream, in, (long long)rows, C, out);
    return check_launch("fold_channels4_kernel");
}

int cslgan_norm_act_bwd_f32(const float* x, const float* dy, const float* y, const float* gamma, const float* stats,
                            int64_t rows, int64_t rows_per_stat, int C, int groups, float eps, int relu, float* ws, float* dx,
                            float* dgamma, float* dbeta, void* stream) {
    CSLGAN_REQUIRE(x && dy && gamma && stats && ws && dx && dgamma && dbeta, "norm_bwd: null argument");
    CSLGAN_REQUIRE(!relu || y, "norm_bwd: relu backward needs the forward output");
    CSLGAN_REQUIRE(rows > 0 && rows_per_stat > 0 && rows % rows_per_stat == 0 && C > 0 && groups > 0 && C % groups == 0 && C <= 4096,
                   "norm_bwd: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const long long nrg = rows / rows_per_stat;
    CSLGAN_REQUIRE(nrg <= 65535, "norm_bwd: too many row groups");
    const int cpg = C / groups;
    float* ab = ws;                         // [nrg][C][2]
    float* gs = ws + nrg * C * 2;           // [nrg*groups][2]
    if (int rc = zero_floats(ab, (size_t)nrg * C * 2, st)) return rc;
    const bool vec = (C % 4 == 0) && ((C / 4) <= 256) && (256 % (C / 4) == 0) && aligned16(x) && aligned16(dy) && (!relu || aligned16(y));
    long long rpb = (rows / 1024 + NS_ROWS_MIN - 1) / NS_ROWS_MIN * NS_ROWS_MIN;
    rpb = rpb < NS_ROWS_MIN ? NS_ROWS_MIN : (rpb > 4096 ? 4096 : rpb);
    const dim3 grid((unsigned)((rows_per_stat + rpb - 1) / rpb), (unsigned)nrg), block(256);
    if (vec) hipLaunchKernelGGL((norm_bwd_stats_kernel<true>), grid, block, sizeof(float) * 2 * C, st, x, dy, y, stats, (long long)rows_per_stat, C, cpg, groups, eps, relu, (int)rpb, ab);
    else hipLaunchKernelGGL((norm_bwd_stats_kernel<false>), grid, block, sizeof(float) * 2 * C, st, x, dy, y, stats, (long long)rows_per_stat, C, cpg, groups, eps, relu, (int)rpb, ab);
    int rc = check_launch("norm_bwd_stats_kernel");
    if (rc) return rc;
    const long long nfin = nrg * groups > C ? nrg * groups : C;
    hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((unsigned)((nfin + 127) / 128)), dim3(128), 0, st, ab, gamma, nrg, C, cpg, groups, gs, dgamma, dbeta);
    rc = check_launch("norm_bwd_finalize_kernel");
    if (rc) return rc;
    const long long total = rows * C;
    hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, st, x, dy, y, gamma, stats, gs, total,
                       (long long)rows_per_stat, C, cpg, groups, eps, relu, dx);
    return check_launch("norm_bwd_apply_kernel");
}

int64_t cslgan_norm_bwd_ws_floats(int64_t rows, int64_t rows_per_stat, int C, int groups) {
    const int64_t nrg = rows_per_stat > 0 ? rows / rows_per_stat : 0;
    return nrg * C * 2 + nrg * groups * 2;
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

int cslgan_adam_step_dev_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                             float weight_decay, const int32_t* step_dev, void* stream) {
    CSLGAN_REQUIRE(p && g && m && v && step_dev, "adam_dev: null argument");
    CSLGAN_REQUIRE(n >= 0, "adam_dev: bad n");
    if (n == 0) return CSLGAN_OK;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, lr, b1, b2,
                       eps, weight_decay, step_dev);
    return check_launch("adam_dev_kernel");
}

}  // extern "C"
