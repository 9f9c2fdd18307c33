// fp32 MFMA grouped weight-gradient kernel ("M-contiguous" implicit GEMM).  gfx950 only.
//
//   gw[g][m][n] = alpha * sum_{k in group g} GY[k][m] * X(k, n)     k -> (img, oy, ox),  n -> (tap, c)
//
// group == 1 gives the per-sample gradients (p.grad_sample) that the reference obtains from the
// Opacus-fork backward hooks (train.py:387; SURVEY.md §8 a7: unfold + einsum('noq,npq->nop')); the
// per-group sum of squares is reduced in the epilogue (wavefront shuffle -> LDS -> one atomic), so
// the separate norm pass over HBM disappears.  With gw == NULL only the norms are produced.
//
// The reduction index (pixels) is the strided one for both operands; a 16-byte global load is 4
// consecutive m (output channels of GY) or 4 consecutive n (input channels of X) of one pixel, and
// goes to a [k][m] LDS image; MFMA operands are ds_read_b32 (32 consecutive floats per half-wave:
// conflict-free).  v_mfma_f32_32x32x2_f32, exact fp32.
#include "common.h"
#include "igemm.h"

namespace cslgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MC_BK = 16;

template <int BM, int BN, int WAVES_M, int WAVES_N, bool VEC_A, bool VEC_B>
__global__ __launch_bounds__(256) void igemm_mc_kernel(const McParams p) {
    constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "bad tile");
    constexpr int A_PASS = (MC_BK * BM / 4) / 256, B_PASS = (MC_BK * BN / 4) / 256;
    constexpr int A_ROWS = 256 / (BM / 4), B_ROWS = 256 / (BN / 4);  // k rows covered per pass
    static_assert(A_PASS >= 1 && B_PASS >= 1, "tile too small");
    __shared__ __attribute__((aligned(16))) float As[2][MC_BK * BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][MC_BK * BN];
    __shared__ float s_red[4];

    const int tid = threadIdx.x;
    const int per_g = p.tiles_m * p.tiles_n;
    const int split = p.ksplit > 1 ? blockIdx.x % p.ksplit : 0;
    const int bid = p.ksplit > 1 ? blockIdx.x / p.ksplit : blockIdx.x;
    const int g = bid / per_g;
    const int tl = bid - g * per_g;
    const int tile_m = tl / p.tiles_n, tile_n = tl - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int PQ = p.P * p.Q;
    const int Ktot = p.group * PQ;
    const long long pix_base = (long long)g * p.group * PQ;

    // A loader: thread -> (k row a_kr + A_ROWS*i, 4 channels at a_mc)
    const int a_kr = tid / (BM / 4), a_mc = (tid % (BM / 4)) * 4;
    const int b_kr = tid / (BN / 4), b_nc = (tid % (BN / 4)) * 4;
    // B columns are fixed per thread: decode (tap, c) once
    int b_ty[4], b_tx[4], b_c[4];
    bool b_nok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int n = n0 + b_nc + e;
        b_nok[e] = n < p.Ndim;
        const int t = b_nok[e] ? n / p.C : 0;
        b_c[e] = n - t * p.C;
        b_ty[e] = p.ty[t];
        b_tx[e] = p.tx[t];
    }

    // two K tiles are in flight in registers: one tile of MFMAs (~1 us) does not cover the global latency of the gathers
    float4 ra[2][A_PASS], rb[2][B_PASS];

    auto load_tile = [&](int kt, int slot) {
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            const int kk = kt * MC_BK + a_kr + A_ROWS * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kk < Ktot) {
                const float* src = p.gy + (pix_base + kk) * p.Kc + m0 + a_mc;
                if (VEC_A) {
                    if (m0 + a_mc < p.Kc) v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (m0 + a_mc + 0 < p.Kc) v.x = src[0];
                    if (m0 + a_mc + 1 < p.Kc) v.y = src[1];
                    if (m0 + a_mc + 2 < p.Kc) v.z = src[2];
                    if (m0 + a_mc + 3 < p.Kc) v.w = src[3];
                }
                if (p.row_scale) {
                    const float sc = p.row_scale[g * p.group + kk / PQ];
                    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
                }
            }
            ra[slot][i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
            const int kk = kt * MC_BK + b_kr + B_ROWS * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kk < Ktot) {
                const int il = kk / PQ;
                const int pix = kk - il * PQ;
                const int oy = pix / p.Q, ox = pix - oy * p.Q;
                const long long img = (long long)g * p.group + il;
                const int by = oy * p.stride, bx = ox * p.stride;
                const int VH = p.H, VW = p.W;
                if (VEC_B) {
                    const int iy = by + b_ty[0], ix = bx + b_tx[0];
                    if (b_nok[0] && iy >= 0 && iy < VH && ix >= 0 && ix < VW)
                        v = *reinterpret_cast<const float4*>(p.x + ((img * p.H + iy) * p.W + ix) * p.C + b_c[0]);
                } else {
                    float t4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int iy = by + b_ty[e], ix = bx + b_tx[e];
                        t4[e] = (b_nok[e] && iy >= 0 && iy < VH && ix >= 0 && ix < VW)
                                    ? p.x[((img * p.H + iy) * p.W + ix) * p.C + b_c[e]] : 0.f;
                    }
                    v = make_float4(t4[0], t4[1], t4[2], t4[3]);
                }
            }
            rb[slot][i] = v;
        }
    };
    auto store_tile = [&](int buf, int slot) {
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) *reinterpret_cast<float4*>(&As[buf][(a_kr + A_ROWS * i) * BM + a_mc]) = ra[slot][i];
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) *reinterpret_cast<float4*>(&Bs[buf][(b_kr + B_ROWS * i) * BN + b_nc]) = rb[slot][i];
    };

    const int lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid / WAVES_N, wn = wid - wm * WAVES_N;
    const int arow0 = wm * TM * 32 + r, brow0 = wn * TN * 32 + r;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    const int nk_all = (Ktot + MC_BK - 1) / MC_BK;
    int kt0 = 0, nk = nk_all;
    if (p.ksplit > 1) {
        const int per = (nk_all + p.ksplit - 1) / p.ksplit;
        kt0 = split * per;
        nk = kt0 + per < nk_all ? kt0 + per : nk_all;
        if (kt0 >= nk) return;      // uniform
    }
    load_tile(kt0, 0);
    store_tile(0, 0);
    if (kt0 + 1 < nk) load_tile(kt0 + 1, 1);
    __syncthreads();
    auto compute_tile = [&](int kt, int buf) {
        int krem = Ktot - kt * MC_BK;
        if (krem > MC_BK) krem = MC_BK;
        const int nsteps = (krem + 1) >> 1;
#pragma unroll
        for (int s = 0; s < MC_BK / 2; ++s) {
            if (s < nsteps) {
                const int kr = 2 * s + h;
                float af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = As[buf][kr * BM + arow0 + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = Bs[buf][kr * BN + brow0 + j * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    // tile kt is computed from LDS buffer (kt-kt0)&1 while tile kt+1 waits in register slot (kt+1-kt0)&1 and the
    // loads of tile kt+2 are issued into the slot tile kt came through
    for (int kt = kt0; kt < nk; kt += 2) {
        if (kt + 2 < nk) load_tile(kt + 2, 0);
        compute_tile(kt, 0);
        if (kt + 1 < nk) store_tile(1, 1);
        __syncthreads();
        if (kt + 1 < nk) {
            if (kt + 3 < nk) load_tile(kt + 3, 1);
            compute_tile(kt + 1, 1);
            if (kt + 2 < nk) store_tile(0, 0);
            __syncthreads();
        }
    }

    // ---- epilogue: scale, store, per-group sum of squares ----------------------------------
    float ss = 0.f;
    float* __restrict__ outg = (p.gw && !p.out_bf16) ? p.gw + (long long)g * p.Kc * p.Ndim : nullptr;
    unsigned short* __restrict__ outh = (p.gw && p.out_bf16) ? reinterpret_cast<unsigned short*>(p.gw) + (long long)g * p.Kc * p.Ndim : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + r;
        if (n >= p.Ndim) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + wm * TM * 32 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                if (m >= p.Kc) continue;
                float val = p.alpha * acc[i][j][v];
                if (p.out_bf16) {      // what is stored is what gets clipped: norm of the rounded value
                    unsigned u = __float_as_uint(val);
                    u += 0x7FFFu + ((u >> 16) & 1u);
                    if (outh) outh[(long long)m * p.Ndim + n] = (unsigned short)(u >> 16);
                    val = __uint_as_float(u & 0xffff0000u);
                }
                ss = fmaf(val, val, ss);
                if (outg) {
                    if (p.ksplit > 1) atomicAdd(&outg[(long long)m * p.Ndim + n], val);
                    else outg[(long long)m * p.Ndim + n] = val;
                }
            }
        }
    }
    if (p.sq && p.ksplit <= 1) {
        const float tot = block_sum_256(ss, s_red);
        if (tid == 0) atomicAdd(p.sq + g, tot);
    }
}

// gb[g][k] = alpha * sum over the group's pixels of gy[.,k]; one block per (group, 64-channel tile)
__global__ __launch_bounds__(256) void bias_grad_grouped_kernel(const float* __restrict__ gy, int PQ, int K, int group,
                                                                float alpha, float* __restrict__ gb, float* __restrict__ sq) {
    __shared__ float part[4][64];
    __shared__ float red[4];
    const int g = blockIdx.y;
    const int k = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sl = threadIdx.x >> 6;  // 4 pixel slices
    const long long npix = (long long)group * PQ;
    const float* base = gy + (long long)g * npix * K;
    float acc = 0.f;
    if (k < K)
        for (long long px = sl; px < npix; px += 4) acc += base[px * K + k];
    part[sl][threadIdx.x & 63] = acc;
    __syncthreads();
    float ss = 0.f;
    if (threadIdx.x < 64 && k < K) {
        const float v = alpha * (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (gb) gb[(long long)g * K + k] = v;
        ss = v * v;
    }
    if (sq) {
        const float tot = block_sum_256(ss, red);
        if (threadIdx.x == 0) atomicAdd(sq + g, tot);
    }
}

// Same for K % 4 == 0, K <= 1024, 256 % (K/4) == 0: one workgroup per group, 16-byte loads (K/4 lanes cover a pixel's
// channels, 256/(K/4) pixel slices), four loads in flight per lane.  The scalar kernel above read one float per lane per
// iteration with a dependent add: 0.69 TB/s by rocprofv3 (0.28 ms per step for 190 MB).
__global__ __launch_bounds__(256) void bias_grad_grouped_vec_kernel(const float* __restrict__ gy, int PQ, int K, int group,
                                                                    float alpha, float* __restrict__ gb, float* __restrict__ sq) {
    extern __shared__ float s_part[];            // [K]
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int c4n = K >> 2, c4 = tid % c4n, sl = tid / c4n, nsl = 256 / c4n;
    const int g = blockIdx.x;
    const long long npix = (long long)group * PQ;
    const float* base = gy + (long long)g * npix * K + 4 * c4;
    for (int i = tid; i < K; i += 256) s_part[i] = 0.f;
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    long long px = sl;
    for (; px + 3 * nsl < npix; px += 4 * nsl) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(base + (px + u * nsl) * K);
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; px < npix; px += nsl) {
        const float4 v = *reinterpret_cast<const float4*>(base + px * K);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    atomicAdd(&s_part[4 * c4 + 0], acc.x);
    atomicAdd(&s_part[4 * c4 + 1], acc.y);
    atomicAdd(&s_part[4 * c4 + 2], acc.z);
    atomicAdd(&s_part[4 * c4 + 3], acc.w);
    __syncthreads();
    float ss = 0.f;
    for (int k = tid; k < K; k += 256) {
        const float v = alpha * s_part[k];
        if (gb) gb[(long long)g * K + k] = v;
        ss = fmaf(v, v, ss);
    }
    if (sq) {
        const float tot = block_sum_256(ss, red);
        if (tid == 0) atomicAdd(sq + g, tot);
    }
}

int sqnorm_rows_accumulate(const float* in, long long n_rows, long long len, float* sq_accum, hipStream_t st);   // clip_kernels.hip

bool wgh_eligible(const cslgan_conv_t* c, int out_bf16, const void* gy, const void* x);      // igemm_wgh.hip
bool x3w_eligible(const cslgan_conv_t* c, int out_bf16, const void* gy, const void* x);      // igemm_wgh.hip: the three-piece (bf16x3) form
bool x3w_quad_eligible(const cslgan_conv_t* c, int group, int out_bf16, const void* gy, const void* x);     // ... on 4x4 output grids
int launch_wgh(const cslgan_conv_t* c, const float* gy, const float* x, int group, float alpha, float* gw, float* sq, hipStream_t st,
               const float* row_scale, int n_seg = 0, const int* seg_first = nullptr, float* const* seg_gw = nullptr,
               float* const* seg_sq = nullptr);      // igemm_wgh.hip

int launch_mc_bf16(McParams& p, bool vecA, bool vecB, hipStream_t st, int nsplit);        // igemm_bf16.hip

template <int BM, int BN, int WM, int WN>
static int launch_mc_tile(McParams& p, bool vecA, bool vecB, hipStream_t st) {
    p.tiles_m = (p.Kc + BM - 1) / BM;
    p.tiles_n = (p.Ndim + BN - 1) / BN;
    // few workgroups with a long pixel loop (the 3-channel first layer: one 64x75 tile per sample, 1024+ pixels):
    // split the pixels over workgroups that atomically add into the zeroed fp32 output
    p.ksplit = 1;
    {
        const long long base = (long long)p.n_groups * p.tiles_m * p.tiles_n;
        const int nk_all = (p.group * p.P * p.Q + MC_BK - 1) / MC_BK;
        if (p.gw && !p.out_bf16 && base < 192 && nk_all >= 32) {
            long long want = (512 + base - 1) / base;
            const long long cap = nk_all / 8;
            p.ksplit = (int)(want < cap ? want : cap);
            if (p.ksplit < 1) p.ksplit = 1;
        }
    }
    if (p.ksplit > 1) {
        if (int rc = zero_floats(p.gw, (size_t)p.n_groups * p.Kc * p.Ndim, st)) return rc;
    }
    const long long nb = (long long)p.n_groups * p.tiles_m * p.tiles_n * p.ksplit;
    if (nb > 0x7fffffffll) { set_error("wgrad: grid too large"); return CSLGAN_ERR_INVALID_ARG; }
    const dim3 grid((unsigned)nb), block(256);
    note_kernel("igemm_mc_kernel<%d,%d>", BM, BN);
    if (vecA && vecB) hipLaunchKernelGGL((igemm_mc_kernel<BM, BN, WM, WN, true, true>), grid, block, 0, st, p);
    else if (vecA) hipLaunchKernelGGL((igemm_mc_kernel<BM, BN, WM, WN, true, false>), grid, block, 0, st, p);
    else if (vecB) hipLaunchKernelGGL((igemm_mc_kernel<BM, BN, WM, WN, false, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((igemm_mc_kernel<BM, BN, WM, WN, false, false>), grid, block, 0, st, p);
    int rc = check_launch("igemm_mc_kernel");
    if (rc) return rc;
    if (p.ksplit > 1 && p.sq) {
        // the per-group norm needs the completed sums: one pass of the contract norm kernel over the (small) result
        rc = sqnorm_rows_accumulate(p.gw, p.n_groups, (long long)p.Kc * p.Ndim, p.sq, st);
    }
    return rc;
}

}  // namespace cslgan

using namespace cslgan;

extern "C" {

static int wgrad_grouped_impl(const cslgan_conv_t* c, const float* gy, const float* x, int group, float alpha,
                              float* gw, float* sq, void* stream, int out_bf16, const float* row_scale = nullptr) {
    CSLGAN_REQUIRE(c && gy && x, "conv2d_wgrad: null argument");
    CSLGAN_REQUIRE(gw || sq, "conv2d_wgrad: neither gw nor sq requested");
    CSLGAN_REQUIRE(group >= 1 && c->N % group == 0, "conv2d_wgrad: N=%d not divisible by group=%d", c->N, group);
    CSLGAN_REQUIRE(c->R * c->S <= IG_MAX_TAPS, "conv2d_wgrad: too many taps");
    CSLGAN_REQUIRE(c->compute >= CSLGAN_COMPUTE_F32 && c->compute <= CSLGAN_COMPUTE_BF16X3, "conv2d_wgrad: unknown cslgan_conv_t.compute %d", c->compute);
    const int P = (c->H + 2 * c->pad - c->R) / c->stride + 1, Q = (c->W + 2 * c->pad - c->S) / c->stride + 1;
    CSLGAN_REQUIRE(P == c->P && Q == c->Q, "conv2d_wgrad: output %dx%d does not match P,Q=%d,%d", P, Q, c->P, c->Q);
    static const int c3_env = [] { const char* e = getenv("CSLGAN_C3"); return e ? atoi(e) : 1; }();
    if (c3_env && !row_scale && c3_wgrad_eligible(c, group, out_bf16, gy))     // the critic's RGB first layer (conv_c3.hip)
        return launch_c3_wgrad(c, gy, x, alpha, gw, sq, (hipStream_t)stream);
    static const int wgh_env = [] { const char* e = getenv("CSLGAN_WGH"); return e ? atoi(e) : 1; }();
    if (wgh_env && ((c->compute == CSLGAN_COMPUTE_F32 && wgh_eligible(c, out_bf16, gy, x)) ||
                    (c->compute == CSLGAN_COMPUTE_BF16X3 && (x3w_eligible(c, out_bf16, gy, x) || x3w_quad_eligible(c, group, out_bf16, gy, x)))))
        return launch_wgh(c, gy, x, group, alpha, gw, sq, (hipStream_t)stream, row_scale);
    McParams p{};
    p.gy = gy; p.x = x; p.N = c->N; p.H = c->H; p.W = c->W; p.C = c->C; p.P = c->P; p.Q = c->Q; p.Kc = c->K;
    p.T = c->R * c->S; p.Ndim = p.T * c->C; p.stride = c->stride; p.group = group; p.n_groups = c->N / group;
    p.alpha = alpha; p.gw = gw; p.sq = sq; p.out_bf16 = out_bf16; p.row_scale = row_scale;
    for (int t = 0; t < IG_MAX_TAPS; ++t) { p.ty[t] = 0; p.tx[t] = 0; }
    for (int kh = 0; kh < c->R; ++kh)
        for (int kw = 0; kw < c->S; ++kw) { p.ty[kh * c->S + kw] = (signed char)(kh - c->pad); p.tx[kh * c->S + kw] = (signed char)(kw - c->pad); }
    const bool vecA = (c->K % 4 == 0) && aligned16(gy);
    const bool vecB = (c->C % 4 == 0) && aligned16(x);
    if (c->compute != CSLGAN_COMPUTE_F32) return launch_mc_bf16(p, vecA, vecB, (hipStream_t)stream, c->compute == CSLGAN_COMPUTE_BF16X3 ? 3 : 1);
    if (c->K > 64) return launch_mc_tile<128, 128, 2, 2>(p, vecA, vecB, (hipStream_t)stream);
    // 64 output channels: a 64x256 tile reads 20 KB per K tile for the MACs a 64x128 tile does with 12 KB twice
    static const int wide64 = [] { const char* e = getenv("CSLGAN_MC_WIDE64"); return e ? atoi(e) : 1; }();
    if (wide64 && c->K > 32 && p.Ndim >= 1024 && (long long)p.n_groups * ((p.Ndim + 255) / 256) >= 256)
        return launch_mc_tile<64, 256, 1, 4>(p, vecA, vecB, (hipStream_t)stream);
    return launch_mc_tile<64, 128, 1, 4>(p, vecA, vecB, (hipStream_t)stream);
}

int cslgan_conv2d_wgrad_grouped_f32(const cslgan_conv_t* c, const float* gy, const float* x, int group, float alpha,
                                    float* gw, float* sq, void* stream) {
    return wgrad_grouped_impl(c, gy, x, group, alpha, gw, sq, stream, 0);
}

int cslgan_conv2d_wgrad_grouped_bf16out_f32(const cslgan_conv_t* c, const float* gy, const float* x, int group, float alpha,
                                            void* gw_bf16, float* sq, void* stream) {
    return wgrad_grouped_impl(c, gy, x, group, alpha, reinterpret_cast<float*>(gw_bf16), sq, stream, 1);
}

int cslgan_conv2d_wgrad_scaled_f32(const cslgan_conv_t* c, const float* gy, const float* x, const float* row_scale, int group,
                                   float alpha, float* gw, void* stream) {
    CSLGAN_REQUIRE(row_scale && gw, "conv2d_wgrad_scaled: null argument");
    return wgrad_grouped_impl(c, gy, x, group, alpha, gw, nullptr, stream, 0, row_scale);
}

int cslgan_conv2d_wgrad_blocks_f32(const cslgan_conv_t* c, const float* gy, const float* x, float alpha, int n_blocks,
                                   const int32_t* block_first, float* const* gw, float* const* sq, void* stream) {
    CSLGAN_REQUIRE(c && gy && x && block_first && gw && sq, "conv2d_wgrad_blocks: null argument");
    CSLGAN_REQUIRE(n_blocks >= 1 && n_blocks <= CSLGAN_MAX_WGRAD_BLOCKS, "conv2d_wgrad_blocks: 1..%d blocks", CSLGAN_MAX_WGRAD_BLOCKS);
    CSLGAN_REQUIRE(block_first[0] == 0, "conv2d_wgrad_blocks: the first block must start at sample 0");
    for (int i = 1; i < n_blocks; ++i)
        CSLGAN_REQUIRE(block_first[i] > block_first[i - 1] && block_first[i] < c->N, "conv2d_wgrad_blocks: block starts must increase inside [0, N)");
    const int P = (c->H + 2 * c->pad - c->R) / c->stride + 1, Q = (c->W + 2 * c->pad - c->S) / c->stride + 1;
    CSLGAN_REQUIRE(P == c->P && Q == c->Q, "conv2d_wgrad_blocks: output %dx%d does not match P,Q=%d,%d", P, Q, c->P, c->Q);
    CSLGAN_REQUIRE((c->compute == CSLGAN_COMPUTE_F32 && wgh_eligible(c, 0, gy, x)) || (c->compute == CSLGAN_COMPUTE_BF16X3 && x3w_eligible(c, 0, gy, x)),
                   "conv2d_wgrad_blocks: shape not taken by the LDS-resident kernels (fp32 or bf16x3, stride 1-2, 2..5 columns, K %% 64, C %% 64, 8x8-patchable output)");
    int first[CSLGAN_MAX_WGRAD_BLOCKS];
    for (int i = 0; i < n_blocks; ++i) first[i] = block_first[i];
    return launch_wgh(c, gy, x, 1, alpha, nullptr, nullptr, (hipStream_t)stream, nullptr, n_blocks, first, gw, sq);
}

int cslgan_bias_grad_grouped_f32(const float* gy, int N, int PQ, int K, int group, float alpha, float* gb, float* sq,
                                 void* stream) {
    CSLGAN_REQUIRE(gy && (gb || sq), "bias_grad: null argument");
    CSLGAN_REQUIRE(N > 0 && PQ > 0 && K > 0 && group >= 1 && N % group == 0, "bias_grad: bad sizes");
    CSLGAN_REQUIRE(N / group <= 65535, "bias_grad: too many groups");
    if (K % 4 == 0 && K <= 1024 && 256 % (K / 4) == 0 && aligned16(gy)) {
        hipLaunchKernelGGL(bias_grad_grouped_vec_kernel, dim3((unsigned)(N / group)), dim3(256), sizeof(float) * K, (hipStream_t)stream,
                           gy, PQ, K, group, alpha, gb, sq);
        return check_launch("bias_grad_grouped_vec_kernel");
    }
    hipLaunchKernelGGL(bias_grad_grouped_kernel, dim3((unsigned)((K + 63) / 64), (unsigned)(N / group)), dim3(256), 0,
                       (hipStream_t)stream, gy, PQ, K, group, alpha, gb, sq);
    return check_launch("bias_grad_grouped_kernel");
}

}  // extern "C"
