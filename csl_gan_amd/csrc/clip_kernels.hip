// Per-sample gradient-norm / clip / accumulate / Gaussian-noise kernels for gfx950.
//
// HBM-bound streaming work: 16-byte coalesced loads, wavefront (DPP shuffle) + LDS partial-norm
// reductions, one atomic per 32 KB chunk.  Roofline: bytes = 2 * P * 4 per clipped sample
// (one read for the norms, one for the weighted sum) — DESIGN.md "clip kernels".
//
// Reference semantics being replaced (file:line under /root/reference):
//   train.py:311-314  calc_sample_norms           -> sample_sqnorm_kernel
//   train.py:324      calc_clipping_factors       -> clip_factors_kernel
//   train.py:399-402  clip(); accum_grads_across_passes()  -> clip_accum_noise_kernel
//   train.py:484      engine-wrapped optimizer.step() noise + 1/B -> clip_accum_noise_kernel
//   backprop_clip.py:18-22 l2_clip                -> l2_clip_rows
//   gradient_penalty.py:52-53 gradients.norm(2, dim=1) -> row_l2norm (+ backward)
#include "common.h"

namespace cslgan {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local char g_kernel[128] = "";
void note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}

// 4 consecutive elements as floats: fp32 = one 16-byte load, bf16 = one 8-byte load widened by a 16-bit shift
struct bf16x4 { unsigned short v[4]; };
template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ float load1(const float* p) { return *p; }
    static constexpr int align = 16;
};
template <> struct Elem<unsigned short> {
    static __device__ __forceinline__ float4 load4(const unsigned short* p) {
        const uint2 r = *reinterpret_cast<const uint2*>(p);
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                           __uint_as_float(r.y & 0xffff0000u));
    }
    static __device__ __forceinline__ float load1(const unsigned short* p) { return __uint_as_float((unsigned)(*p) << 16); }
    static constexpr int align = 8;
};

constexpr int SQ_THREADS = 256;
constexpr int SQ_CHUNK = SQ_THREADS * 4 * 8;  // floats per block: 32 KB

struct SqArgs {
    int n_seg;
    const float* in[CSLGAN_MAX_SEGS];
    long long len[CSLGAN_MAX_SEGS];
    long long row_stride[CSLGAN_MAX_SEGS];
    int chunk_prefix[CSLGAN_MAX_SEGS + 1];  // blocks (chunks) per row before segment s
    int vec_ok[CSLGAN_MAX_SEGS];            // 16-byte loads allowed for this segment
};

template <typename T>
__global__ __launch_bounds__(SQ_THREADS) void sample_sqnorm_kernel(SqArgs a, long long n_rows, float* __restrict__ out_sq) {
    __shared__ float red[4];
    const int bx = blockIdx.x;
    const long long row = blockIdx.y;
    int s = 0;
#pragma unroll 1
    while (s + 1 < a.n_seg && bx >= a.chunk_prefix[s + 1]) ++s;
    const long long off = (long long)(bx - a.chunk_prefix[s]) * SQ_CHUNK;
    const long long len = a.len[s];
    long long n = len - off;
    if (n > SQ_CHUNK) n = SQ_CHUNK;
    const T* __restrict__ p = reinterpret_cast<const T*>(a.in[s]) + row * a.row_stride[s] + off;
    float acc = 0.f;
    if (a.vec_ok[s]) {
        const long long n4 = n >> 2;
#pragma unroll 8
        for (long long i = threadIdx.x; i < n4; i += SQ_THREADS) {
            const float4 v = Elem<T>::load4(p + 4 * i);
            acc = fmaf(v.x, v.x, acc);
            acc = fmaf(v.y, v.y, acc);
            acc = fmaf(v.z, v.z, acc);
            acc = fmaf(v.w, v.w, acc);
        }
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += SQ_THREADS) { const float v = Elem<T>::load1(p + i); acc = fmaf(v, v, acc); }
    } else {
#pragma unroll 4
        for (long long i = threadIdx.x; i < n; i += SQ_THREADS) { const float v = Elem<T>::load1(p + i); acc = fmaf(v, v, acc); }
    }
    const float tot = block_sum_256(acc, red);
    if (threadIdx.x == 0) atomicAdd(out_sq + (long long)s * n_rows + row, tot);
}

__global__ void clip_factors_kernel(const float* __restrict__ sq, int n_seg, long long n_rows,
                                    const float* __restrict__ max_norm, int flat, float eps,
                                    long long first_private_row, float* __restrict__ out_f,
                                    float* __restrict__ out_norm) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    if (flat) {
        float tot = 0.f;
        for (int s = 0; s < n_seg; ++s) tot += sq[(long long)s * n_rows + r];
        const float nrm = sqrtf(tot);
        float f = max_norm[0] / (nrm + eps);
        f = f > 1.f ? 1.f : f;
        out_f[r] = r < first_private_row ? 1.f : f;
        if (out_norm) out_norm[r] = nrm;
    } else {
        for (int s = 0; s < n_seg; ++s) {
            const float nrm = sqrtf(sq[(long long)s * n_rows + r]);
            float f = max_norm[s] / (nrm + eps);
            f = f > 1.f ? 1.f : f;
            out_f[(long long)s * n_rows + r] = r < first_private_row ? 1.f : f;
            if (out_norm) out_norm[(long long)s * n_rows + r] = nrm;
        }
    }
}

// ---- Philox4x32-10 ------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    // u1 in (0,1], u2 in [0,1)
    const float u1 = (float)(a >> 8) * (1.0f / 16777216.0f) + (0.5f / 16777216.0f);
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * __logf(u1));
    float sn, cs;
    __sincosf(6.283185307179586f * u2, &sn, &cs);
    z0 = r * cs;
    z1 = r * sn;
}

// ---- MeanSampler.sample (mean_sampler.py:75-84): gather + per-image jitter + per-pixel noise in one pass -------------------
//   out[i][e] = ms[label[i]][perm[i]][e] + noise_mean_std * z_i + noise_std * z_{i,e}     (Philox4x32-10 + Box-Muller)
// The reference draws on the host and copies the batch to the device every step (train.py:200-202, 214-216); as torch ops on
// the device it is an index kernel, two normal_ fills and two adds (five passes over the batch).
// perms == NULL: the kernel draws the permutations itself — image i takes entry (i mod num_samples) of the (i / num_samples)-th
// uniform random permutation of the num_samples mean samples (torch.cat of randperms, mean_sampler.py:76), found by ranking
// num_samples Philox keys in LDS (num_samples <= MS_MAX_PERM); labels == NULL with n_classes > 1: labels drawn uniformly
// (mean_sampler.py:77) and written to labels_out.  What were a rand, an argsort (arange + copies + bitonic sort), a randint and
// the index arithmetic around them — eight launches per draw, two draws per D-step — happens inside the gather.
constexpr int MS_MAX_PERM = 1024;
__global__ __launch_bounds__(256) void mean_sample_kernel(const float* __restrict__ ms, const long long* __restrict__ labels,
                                                          const long long* __restrict__ perms, int num_samples, int n_classes, long long len,
                                                          float noise_mean_std, float noise_std, unsigned long long seed,
                                                          unsigned long long offset, float* __restrict__ out,
                                                          long long* __restrict__ labels_out) {
    const long long i = blockIdx.y;
    const long long e4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // float4 index inside the image
    uint32_t rnd[4];
    long long perm_i;
    if (perms) {
        perm_i = perms[i];
    } else {
        __shared__ uint32_t s_key[MS_MAX_PERM];
        __shared__ int s_sel;
        const uint32_t rep = (uint32_t)(i / num_samples);
        const int pos = (int)(i % num_samples);
        for (int j = threadIdx.x; j < num_samples; j += blockDim.x) {
            philox4x32_10((uint32_t)j, rep, (uint32_t)offset, 0x7065726Du ^ (uint32_t)(offset >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
            s_key[j] = rnd[0];
        }
        __syncthreads();
        for (int j = threadIdx.x; j < num_samples; j += blockDim.x) {
            const uint32_t kj = s_key[j];
            int rank = 0;
            for (int m = 0; m < num_samples; ++m) {
                const uint32_t km = s_key[m];
                rank += (km < kj || (km == kj && m < j)) ? 1 : 0;
            }
            if (rank == pos) s_sel = j;
        }
        __syncthreads();
        perm_i = s_sel;
    }
    long long lab = 0;
    if (labels) lab = labels[i];
    else if (n_classes > 1) {
        philox4x32_10((uint32_t)i, 0x6C61626Cu, (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
        lab = (long long)(((unsigned long long)rnd[0] * (unsigned long long)n_classes) >> 32);
    }
    if (labels_out && blockIdx.x == 0 && threadIdx.x == 0) labels_out[i] = lab;
    if (e4 * 4 >= len) return;
    const float* src = ms + (lab * num_samples + perm_i) * len;
    float zi = 0.f, unused;
    if (noise_mean_std > 0.f) {                         // the image's jitter: the same counter in every thread of the image
        philox4x32_10((uint32_t)i, 0xFFFFFFFFu, (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
        box_muller(rnd[0], rnd[1], zi, unused);
    }
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (noise_std > 0.f) {
        philox4x32_10((uint32_t)e4, (uint32_t)i, (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
        box_muller(rnd[0], rnd[1], z[0], z[1]);
        box_muller(rnd[2], rnd[3], z[2], z[3]);
    }
    const float j = noise_mean_std * zi;
    if (e4 * 4 + 3 < len && (len & 3) == 0) {
        const float4 v = *reinterpret_cast<const float4*>(src + e4 * 4);
        *reinterpret_cast<float4*>(out + i * len + e4 * 4) =
            make_float4(v.x + j + noise_std * z[0], v.y + j + noise_std * z[1], v.z + j + noise_std * z[2], v.w + j + noise_std * z[3]);
    } else {
        for (int k = 0; k < 4 && e4 * 4 + k < len; ++k) out[i * len + e4 * 4 + k] = src[e4 * 4 + k] + j + noise_std * z[k];
    }
}

constexpr int CA_THREADS = 256;
constexpr int CA_COLS = CA_THREADS * 4;  // columns per block

struct CaArgs {
    int n_seg;
    const float* in[CSLGAN_MAX_SEGS];
    float* out[CSLGAN_MAX_SEGS];
    const float* noise[CSLGAN_MAX_SEGS];
    long long len[CSLGAN_MAX_SEGS];
    long long row_stride[CSLGAN_MAX_SEGS];
    int rows[CSLGAN_MAX_SEGS];                // > 0: the segment's own row count (ragged column sums), 0: n_rows
    int tile_prefix[CSLGAN_MAX_SEGS + 1];
    int vec_ok[CSLGAN_MAX_SEGS];
    const unsigned long long* call_counter;   // nullable: the Philox offset advances by 64 x *call_counter (graph replays)
};

template <typename T>
__global__ __launch_bounds__(CA_THREADS) void clip_accum_noise_kernel(CaArgs a, long long n_rows,
                                                                      const float* __restrict__ factors,
                                                                      int factors_per_seg,
                                                                      const float* __restrict__ noise_std,
                                                                      unsigned long long seed,
                                                                      unsigned long long offset, float scale,
                                                                      float beta) {
    const int bx = blockIdx.x;
    if (a.call_counter) offset += 64ull * (*a.call_counter);
    int s = 0;
#pragma unroll 1
    while (s + 1 < a.n_seg && bx >= a.tile_prefix[s + 1]) ++s;
    if (a.rows[s] > 0) n_rows = a.rows[s];
    const long long j0 = (long long)(bx - a.tile_prefix[s]) * CA_COLS + (long long)threadIdx.x * 4;
    const long long len = a.len[s];
    if (j0 >= len) return;
    const long long stride = a.row_stride[s];
    const T* __restrict__ base = reinterpret_cast<const T*>(a.in[s]) + j0;
    const float* __restrict__ f = factors ? factors + (factors_per_seg ? (long long)s * n_rows : 0) : nullptr;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool full = (j0 + 4 <= len);
    if (full && a.vec_ok[s]) {
#pragma unroll 8
        for (long long r = 0; r < n_rows; ++r) {
            const float4 v = Elem<T>::load4(base + r * stride);
            const float fr = f ? f[r] : 1.f;
            acc.x = fmaf(fr, v.x, acc.x);
            acc.y = fmaf(fr, v.y, acc.y);
            acc.z = fmaf(fr, v.z, acc.z);
            acc.w = fmaf(fr, v.w, acc.w);
        }
    } else {
        const int nv = full ? 4 : (int)(len - j0);
#pragma unroll 4
        for (long long r = 0; r < n_rows; ++r) {
            const T* q = base + r * stride;
            const float fr = f ? f[r] : 1.f;
            acc.x = fmaf(fr, Elem<T>::load1(q), acc.x);
            if (nv > 1) acc.y = fmaf(fr, Elem<T>::load1(q + 1), acc.y);
            if (nv > 2) acc.z = fmaf(fr, Elem<T>::load1(q + 2), acc.z);
            if (nv > 3) acc.w = fmaf(fr, Elem<T>::load1(q + 3), acc.w);
        }
    }
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    float sd = 0.f;
    if (noise_std) {
        sd = noise_std[s];
        if (a.noise[s]) {
            for (int i = 0; i < 4; ++i)
                if (j0 + i < len) z[i] = a.noise[s][j0 + i];
        } else {
            uint32_t rnd[4];
            const unsigned long long ctr = (unsigned long long)(j0 >> 2);
            philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)s + (uint32_t)(offset << 8),
                          (uint32_t)(offset >> 24), (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
            box_muller(rnd[0], rnd[1], z[0], z[1]);
            box_muller(rnd[2], rnd[3], z[2], z[3]);
        }
    }
    float* __restrict__ o = a.out[s] + j0;
    const float v[4] = {acc.x, acc.y, acc.z, acc.w};
    for (int i = 0; i < 4; ++i) {
        if (j0 + i < len) {
            const float val = scale * fmaf(sd, z[i], v[i]);
            o[i] = (beta != 0.f) ? fmaf(beta, o[i], val) : val;
        }
    }
}

// ---- row helpers ----------------------------------------------------------------------------
__global__ void zero_floats_kernel(float* __restrict__ p, long long n, int vec) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        if (i < n) reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else if (i < n) {
        p[i] = 0.f;
    }
}

__global__ void sqrt_kernel(float* __restrict__ v, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = sqrtf(v[i]);
}

// out[r][j] = in[r][j] * (sq[r] > C^2 ? C / sqrt(sq[r]) : 1)
__global__ void l2_clip_scale_kernel(const float* __restrict__ in, float* __restrict__ out, long long n_rows,
                                     long long len, float C, const float* __restrict__ sq) {
    const long long row = blockIdx.y;
    const float nrm = sqrtf(sq[row]);
    const float f = nrm > C ? C / nrm : 1.f;
    const float* p = in + row * len;
    float* o = out + row * len;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < len; j += (long long)gridDim.x * blockDim.x)
        o[j] = nrm > C ? p[j] * f : p[j];
}

// gin[r][j] = gnorm[r] * in[r][j] / norm[r]
__global__ void row_l2norm_bwd_kernel(const float* __restrict__ in, const float* __restrict__ norm,
                                      const float* __restrict__ gnorm, long long len, float* __restrict__ gin) {
    const long long row = blockIdx.y;
    const float f = gnorm[row] / norm[row];
    const float* p = in + row * len;
    float* o = gin + row * len;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < len; j += (long long)gridDim.x * blockDim.x)
        o[j] = p[j] * f;
}

int zero_floats(float* p, size_t n_floats, hipStream_t st) {
    if (n_floats == 0) return CSLGAN_OK;
    const int vec = (aligned16(p) && n_floats % 4 == 0) ? 1 : 0;
    const long long n = (long long)(vec ? n_floats / 4 : n_floats);
    hipLaunchKernelGGL(zero_floats_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, vec);
    return check_launch("zero_floats_kernel");
}

static bool aligned_to(const void* p, int a) { return (reinterpret_cast<uintptr_t>(p) & (uintptr_t)(a - 1)) == 0; }

static int launch_sqnorm(const float* const* in, const long long* len, const long long* stride, int n_seg,
                         long long n_rows, float* out_sq, hipStream_t st, bool bf16 = false) {
    SqArgs a;
    a.n_seg = n_seg;
    int tot = 0;
    for (int s = 0; s < n_seg; ++s) {
        a.in[s] = in[s];
        a.len[s] = len[s];
        a.row_stride[s] = stride[s];
        a.chunk_prefix[s] = tot;
        a.vec_ok[s] = aligned_to(in[s], bf16 ? 8 : 16) && (stride[s] % 4 == 0);
        tot += (int)((len[s] + SQ_CHUNK - 1) / SQ_CHUNK);
    }
    a.chunk_prefix[n_seg] = tot;
    for (int s = n_seg; s < CSLGAN_MAX_SEGS; ++s) { a.in[s] = nullptr; a.len[s] = 0; a.row_stride[s] = 0; a.vec_ok[s] = 0; a.chunk_prefix[s + 1] = tot; }
    if (int rc = zero_floats(out_sq, (size_t)n_seg * (size_t)n_rows, st)) return rc;
    if (tot == 0 || n_rows == 0) return CSLGAN_OK;
    note_kernel(bf16 ? "sample_sqnorm_kernel<bf16>" : "sample_sqnorm_kernel<float>");
    if (bf16) hipLaunchKernelGGL(sample_sqnorm_kernel<unsigned short>, dim3(tot, (unsigned)n_rows), dim3(SQ_THREADS), 0, st, a, n_rows, out_sq);
    else hipLaunchKernelGGL(sample_sqnorm_kernel<float>, dim3(tot, (unsigned)n_rows), dim3(SQ_THREADS), 0, st, a, n_rows, out_sq);
    return check_launch("sample_sqnorm_kernel");
}

// sq_accum[r] += ||in[r, :]||^2  (no zeroing: the caller's accumulator semantics, as the wgrad epilogue has)
int sqnorm_rows_accumulate(const float* in, long long n_rows, long long len, float* sq_accum, hipStream_t st) {
    if (n_rows <= 0 || len <= 0) return CSLGAN_OK;
    if (n_rows > 65535) { set_error("sqnorm_rows_accumulate: too many rows"); return CSLGAN_ERR_INVALID_ARG; }
    SqArgs a;
    for (int s = 0; s < CSLGAN_MAX_SEGS; ++s) { a.in[s] = nullptr; a.len[s] = 0; a.row_stride[s] = 0; a.vec_ok[s] = 0; a.chunk_prefix[s] = 0; }
    a.n_seg = 1;
    a.in[0] = in; a.len[0] = len; a.row_stride[0] = len;
    a.vec_ok[0] = aligned_to(in, 16) && (len % 4 == 0);
    const int tot = (int)((len + SQ_CHUNK - 1) / SQ_CHUNK);
    for (int s = 1; s <= CSLGAN_MAX_SEGS; ++s) a.chunk_prefix[s] = tot;
    hipLaunchKernelGGL(sample_sqnorm_kernel<float>, dim3(tot, (unsigned)n_rows), dim3(SQ_THREADS), 0, st, a, n_rows, sq_accum);
    return check_launch("sample_sqnorm_kernel");
}

}  // namespace cslgan

using namespace cslgan;

extern "C" {

int cslgan_version(void) { return CSLGAN_ABI_VERSION; }
const char* cslgan_last_error(void) { return cslgan::g_err; }
const char* cslgan_last_kernel(void) { return cslgan::g_kernel; }

int cslgan_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

static int sample_sqnorm_impl(const cslgan_segs_t* segs, int64_t n_rows, float* out_sq, void* stream, bool bf16) {
    CSLGAN_REQUIRE(segs && out_sq, "sample_sqnorm: null argument");
    CSLGAN_REQUIRE(segs->n_seg >= 1 && segs->n_seg <= CSLGAN_MAX_SEGS, "sample_sqnorm: n_seg=%d out of range", segs->n_seg);
    CSLGAN_REQUIRE(n_rows >= 0 && n_rows <= 65535, "sample_sqnorm: n_rows=%lld out of range", (long long)n_rows);
    const float* in[CSLGAN_MAX_SEGS];
    long long len[CSLGAN_MAX_SEGS], stride[CSLGAN_MAX_SEGS];
    for (int s = 0; s < segs->n_seg; ++s) {
        CSLGAN_REQUIRE(segs->in[s] || segs->len[s] == 0, "sample_sqnorm: segment %d is null", s);
        CSLGAN_REQUIRE(segs->len[s] >= 0 && segs->row_stride[s] >= segs->len[s], "sample_sqnorm: bad len/stride in segment %d", s);
        in[s] = segs->in[s]; len[s] = segs->len[s]; stride[s] = segs->row_stride[s];
    }
    return launch_sqnorm(in, len, stride, segs->n_seg, n_rows, out_sq, (hipStream_t)stream, bf16);
}

int cslgan_sample_sqnorm_f32(const cslgan_segs_t* segs, int64_t n_rows, float* out_sq, void* stream) {
    return sample_sqnorm_impl(segs, n_rows, out_sq, stream, false);
}

int cslgan_sample_sqnorm_bf16(const cslgan_segs_t* segs, int64_t n_rows, float* out_sq, void* stream) {
    return sample_sqnorm_impl(segs, n_rows, out_sq, stream, true);
}

int cslgan_clip_factors_f32(const float* sq, int n_seg, int64_t n_rows, const float* max_norm, int flat, float eps,
                            int64_t first_private_row, float* out_f, float* out_norm, void* stream) {
    CSLGAN_REQUIRE(sq && max_norm && out_f, "clip_factors: null argument");
    CSLGAN_REQUIRE(n_seg >= 1 && n_rows >= 0, "clip_factors: bad sizes");
    if (n_rows == 0) return CSLGAN_OK;
    const int th = 128;
    hipLaunchKernelGGL(clip_factors_kernel, dim3((unsigned)((n_rows + th - 1) / th)), dim3(th), 0, (hipStream_t)stream, sq,
                       n_seg, (long long)n_rows, max_norm, flat, eps, (long long)first_private_row, out_f, out_norm);
    return check_launch("clip_factors_kernel");
}

static int clip_accum_noise_impl(const cslgan_segs_t* segs, int64_t n_rows, const float* factors, int factors_per_seg,
                                 const float* noise_std, uint64_t seed, uint64_t offset, float scale, float beta,
                                 void* stream, bool bf16) {
    CSLGAN_REQUIRE(segs, "clip_accum_noise: null segs");
    CSLGAN_REQUIRE(segs->n_seg >= 1 && segs->n_seg <= CSLGAN_MAX_SEGS, "clip_accum_noise: n_seg=%d out of range", segs->n_seg);
    CSLGAN_REQUIRE(n_rows >= 0, "clip_accum_noise: n_rows < 0");
    CaArgs a;
    a.n_seg = segs->n_seg;
    a.call_counter = reinterpret_cast<const unsigned long long*>(segs->call_counter);
    int tot = 0;
    for (int s = 0; s < CSLGAN_MAX_SEGS; ++s) {
        if (s < segs->n_seg) {
            CSLGAN_REQUIRE(segs->out[s] || segs->len[s] == 0, "clip_accum_noise: out[%d] is null", s);
            CSLGAN_REQUIRE((segs->in[s] || n_rows == 0 || segs->len[s] == 0), "clip_accum_noise: in[%d] is null", s);
            CSLGAN_REQUIRE(segs->len[s] >= 0 && (n_rows == 0 || segs->row_stride[s] >= segs->len[s]), "clip_accum_noise: bad len/stride in segment %d", s);
            CSLGAN_REQUIRE(segs->rows[s] >= 0 && segs->rows[s] < (1ll << 31) && (segs->rows[s] == 0 || !factors), "clip_accum_noise: per-segment row counts need factors == NULL (segment %d)", s);
            a.in[s] = segs->in[s]; a.out[s] = segs->out[s]; a.noise[s] = segs->noise[s];
            a.len[s] = segs->len[s]; a.row_stride[s] = segs->row_stride[s]; a.rows[s] = (int)segs->rows[s];
            a.vec_ok[s] = aligned_to(segs->in[s], bf16 ? 8 : 16) && (segs->row_stride[s] % 4 == 0);
            a.tile_prefix[s] = tot;
            tot += (int)((segs->len[s] + CA_COLS - 1) / CA_COLS);
        } else {
            a.in[s] = nullptr; a.out[s] = nullptr; a.noise[s] = nullptr; a.len[s] = 0; a.row_stride[s] = 0; a.vec_ok[s] = 0; a.rows[s] = 0;
            a.tile_prefix[s] = tot;
        }
    }
    a.tile_prefix[CSLGAN_MAX_SEGS] = tot;
    for (int s = segs->n_seg; s <= CSLGAN_MAX_SEGS; ++s) a.tile_prefix[s] = tot;
    if (tot == 0) return CSLGAN_OK;
    note_kernel(bf16 ? "clip_accum_noise_kernel<bf16>" : "clip_accum_noise_kernel<float>");
    if (bf16) hipLaunchKernelGGL(clip_accum_noise_kernel<unsigned short>, dim3(tot), dim3(CA_THREADS), 0, (hipStream_t)stream, a, (long long)n_rows,
                                 factors, factors_per_seg, noise_std, (unsigned long long)seed, (unsigned long long)offset, scale, beta);
    else hipLaunchKernelGGL(clip_accum_noise_kernel<float>, dim3(tot), dim3(CA_THREADS), 0, (hipStream_t)stream, a, (long long)n_rows,
                            factors, factors_per_seg, noise_std, (unsigned long long)seed, (unsigned long long)offset, scale, beta);
    return check_launch("clip_accum_noise_kernel");
}

int cslgan_clip_accum_noise_f32(const cslgan_segs_t* segs, int64_t n_rows, const float* factors, int factors_per_seg,
                                const float* noise_std, uint64_t seed, uint64_t offset, float scale, float beta, void* stream) {
    return clip_accum_noise_impl(segs, n_rows, factors, factors_per_seg, noise_std, seed, offset, scale, beta, stream, false);
}

int cslgan_clip_accum_noise_bf16(const cslgan_segs_t* segs, int64_t n_rows, const float* factors, int factors_per_seg,
                                 const float* noise_std, uint64_t seed, uint64_t offset, float scale, float beta, void* stream) {
    return clip_accum_noise_impl(segs, n_rows, factors, factors_per_seg, noise_std, seed, offset, scale, beta, stream, true);
}

int cslgan_l2_clip_rows_f32(const float* in, float* out, int64_t n_rows, int64_t len, float C, float* norms_ws, void* stream) {
    CSLGAN_REQUIRE(in && out && norms_ws, "l2_clip_rows: null argument");
    CSLGAN_REQUIRE(n_rows >= 0 && n_rows <= 65535 && len >= 0, "l2_clip_rows: bad sizes");
    if (n_rows == 0 || len == 0) return CSLGAN_OK;
    const float* ins[1] = {in};
    long long lens[1] = {len}, strides[1] = {len};
    int rc = launch_sqnorm(ins, lens, strides, 1, n_rows, norms_ws, (hipStream_t)stream);
    if (rc) return rc;
    unsigned gx = (unsigned)((len + 1023) / 1024);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(l2_clip_scale_kernel, dim3(gx, (unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, in, out,
                       (long long)n_rows, (long long)len, C, norms_ws);
    return check_launch("l2_clip_scale_kernel");
}

int cslgan_mean_sample_f32(const float* mean_samples, int n_classes, int num_samples, int64_t len, const int64_t* labels,
                           const int64_t* perms, int64_t n, float noise_mean_std, float noise_std, uint64_t seed, uint64_t offset,
                           float* out, int64_t* labels_out, void* stream) {
    CSLGAN_REQUIRE(mean_samples && out, "mean_sample: null argument");
    CSLGAN_REQUIRE(n_classes >= 1 && num_samples >= 1 && len >= 1 && n >= 0 && n <= 65535, "mean_sample: bad sizes");
    CSLGAN_REQUIRE(perms || num_samples <= MS_MAX_PERM, "mean_sample: in-kernel permutations need num_samples <= %d", MS_MAX_PERM);
    CSLGAN_REQUIRE(labels || labels_out || n_classes == 1, "mean_sample: with more than one class give labels, or labels_out to receive drawn ones");
    CSLGAN_REQUIRE(aligned16(mean_samples) && aligned16(out), "mean_sample: tensors must be 16-byte aligned");
    if (n == 0) return CSLGAN_OK;
    const long long f4 = (len + 3) / 4;
    note_kernel("mean_sample_kernel");
    hipLaunchKernelGGL(mean_sample_kernel, dim3((unsigned)((f4 + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, mean_samples,
                       reinterpret_cast<const long long*>(labels), reinterpret_cast<const long long*>(perms), num_samples, n_classes,
                       (long long)len, noise_mean_std, noise_std, (unsigned long long)seed, (unsigned long long)offset, out,
                       reinterpret_cast<long long*>(labels_out));
    return check_launch("mean_sample_kernel");
}

int cslgan_row_l2norm_f32(const float* in, int64_t n_rows, int64_t len, float* out_norm, void* stream) {
    CSLGAN_REQUIRE(in && out_norm, "row_l2norm: null argument");
    CSLGAN_REQUIRE(n_rows >= 0 && n_rows <= 65535 && len >= 0, "row_l2norm: bad sizes");
    if (n_rows == 0) return CSLGAN_OK;
    const float* ins[1] = {in};
    long long lens[1] = {len}, strides[1] = {len};
    int rc = launch_sqnorm(ins, lens, strides, 1, n_rows, out_norm, (hipStream_t)stream);
    if (rc) return rc;
    hipLaunchKernelGGL(sqrt_kernel, dim3((unsigned)((n_rows + 127) / 128)), dim3(128), 0, (hipStream_t)stream, out_norm, (long long)n_rows);
    return check_launch("sqrt_kernel");
}

int cslgan_row_l2norm_bwd_f32(const float* in, const float* norm, const float* gnorm, int64_t n_rows, int64_t len,
                              float* gin, void* stream) {
    CSLGAN_REQUIRE(in && norm && gnorm && gin, "row_l2norm_bwd: null argument");
    CSLGAN_REQUIRE(n_rows >= 0 && n_rows <= 65535 && len >= 0, "row_l2norm_bwd: bad sizes");
    if (n_rows == 0 || len == 0) return CSLGAN_OK;
    unsigned gx = (unsigned)((len + 1023) / 1024);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(row_l2norm_bwd_kernel, dim3(gx, (unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, in, norm, gnorm,
                       (long long)len, gin);
    return check_launch("row_l2norm_bwd_kernel");
}

}  // extern "C"
