// fp32 MFMA implicit-GEMM convolution, "K-contiguous" form: forward conv / linear and the data
// gradient (transposed conv).  gfx950 only.
//
//   Out[m][n] = epilogue( sum_k A(m,k) * Wm[n][k] )      (index maps: igemm.h)
//
// Tiling: 256 threads = 4 wavefronts; block tile BM x BN, K tile 32.  Both operands have k
// contiguous in HBM (NHWC activations, KRSC filters), so a 16-byte global load is 4 consecutive
// k of one row; it lands in LDS as one ds_write_b128 into a [k/4][row][4] image (chunk stride
// padded by 16 B: the 8 lanes of a write group hit 8 distinct 16-B slots).  Each lane feeds
// v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = 157 TF chip peak) from one ds_read_b128
// per operand per 8 k: half-wave h owns k = 8g+4h+e at step e — the same permutation for A and B,
// so the sum is unchanged.  Global loads for tile t+1 are issued before the MFMAs of tile t and
// written to the other LDS buffer after them (one barrier per K tile).
//
// Replaces (reference file:line): torch.nn.Conv2d / nn.Linear forward DCResNet_models.py:131-132,
// 145, 13-17, 60-70, 95-104; MNIST_models.py:17-23, 41-46; and the autograd data-gradient of those.
#include "common.h"
#include "igemm.h"

namespace cslgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BM, int BN, int WAVES_M, int WAVES_N, bool VEC_A, bool VEC_B>
__global__ __launch_bounds__(256) void igemm_kc_kernel(const KcParams p) {
    constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "bad tile");
    constexpr int A_CH = BM * 4 + 4, B_CH = BN * 4 + 4;  // floats per k-chunk (padded)
    constexpr int A_PASS = BM / 32, B_PASS = BN / 32;
    __shared__ __attribute__((aligned(16))) float As[2][8 * A_CH];
    __shared__ __attribute__((aligned(16))) float Bs[2][8 * B_CH];
    __shared__ int s_tap[IG_MAX_TAPS];
    __shared__ int s_off[BM];
    __shared__ int s_roff[BM];

    const int tid = threadIdx.x;
    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tile_m = wg / p.tiles_n, tile_n = wg - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    if (tid < IG_MAX_TAPS) s_tap[tid] = ((int)p.ty[tid] << 16) | ((int)p.tx[tid] & 0xffff);

    // ---- per-thread loader coordinates -----------------------------------------------------
    const int lrow = tid >> 3;   // 0..31
    const int q = tid & 7;       // k-chunk within the tile
    int a_img[A_PASS], a_iy[A_PASS], a_ix[A_PASS];
    bool a_ok[A_PASS];
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
        const int m = m0 + lrow + 32 * i;
        a_ok[i] = m < p.M;
        const RowCoord rc = kc_decode_row(a_ok[i] ? m : 0, p.OHc, p.OWc);
        a_img[i] = rc.img;
        a_iy[i] = rc.oy * p.sy;
        a_ix[i] = rc.ox * p.sx;
    }
    const float* b_ptr[B_PASS];
    bool b_ok[B_PASS];
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
        const int n = n0 + lrow + 32 * i;
        b_ok[i] = n < p.Nn;
        b_ptr[i] = p.w + (long long)(b_ok[i] ? n : 0) * p.ldw;
    }
    __syncthreads();  // s_tap visible

    float4 ra[A_PASS], rb[B_PASS];

    auto load_tile = [&](int kt) {
        const int kb = kt * IG_BK + q * 4;
        if (VEC_A) {
            const bool kin = kb < p.Kdim;
            const int t = kin ? kb / p.AC : 0;
            const int c = kb - t * p.AC;
            const int tap = s_tap[t];
            const int ty = tap >> 16, tx = (int)(short)(tap & 0xffff);
#pragma unroll
            for (int i = 0; i < A_PASS; ++i) {
                const int iy = a_iy[i] + ty, ix = a_ix[i] + tx;
                const bool ok = kin && a_ok[i] && iy >= 0 && iy < p.VH && ix >= 0 && ix < p.VW;
                if (ok) {
                    const long long off = (((long long)a_img[i] * p.AH + (iy >> p.ups)) * p.AW + (ix >> p.ups)) * p.AC + c;
                    ra[i] = *reinterpret_cast<const float4*>(p.a + off);
                } else {
                    ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        } else {
            float tmp[A_PASS][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = kb + e;
                const bool kin = k < p.Kdim;
                const int t = kin ? k / p.AC : 0;
                const int c = k - t * p.AC;
                const int tap = s_tap[t];
                const int ty = tap >> 16, tx = (int)(short)(tap & 0xffff);
#pragma unroll
                for (int i = 0; i < A_PASS; ++i) {
                    const int iy = a_iy[i] + ty, ix = a_ix[i] + tx;
                    const bool ok = kin && a_ok[i] && iy >= 0 && iy < p.VH && ix >= 0 && ix < p.VW;
                    float v = 0.f;
                    if (ok) v = p.a[(((long long)a_img[i] * p.AH + (iy >> p.ups)) * p.AW + (ix >> p.ups)) * p.AC + c];
                    tmp[i][e] = v;
                }
            }
#pragma unroll
            for (int i = 0; i < A_PASS; ++i) ra[i] = make_float4(tmp[i][0], tmp[i][1], tmp[i][2], tmp[i][3]);
        }
        if (VEC_B) {
            const bool kin = kb < p.Kdim;
#pragma unroll
            for (int i = 0; i < B_PASS; ++i) {
                if (kin && b_ok[i]) rb[i] = *reinterpret_cast<const float4*>(b_ptr[i] + kb);
                else rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_PASS; ++i) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (b_ok[i] && (kb + e) < p.Kdim) ? b_ptr[i][kb + e] : 0.f;
                rb[i] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    };

    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PASS; ++i)
            *reinterpret_cast<float4*>(&As[buf][q * A_CH + (lrow + 32 * i) * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_PASS; ++i)
            *reinterpret_cast<float4*>(&Bs[buf][q * B_CH + (lrow + 32 * i) * 4]) = rb[i];
    };

    // ---- MFMA coordinates ------------------------------------------------------------------
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

    const int nk = (p.Kdim + IG_BK - 1) / IG_BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = 2 * g + h;
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(&As[buf][ch * A_CH + (arow0 + i * 32) * 4]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(&Bs[buf][ch * B_CH + (brow0 + j * 32) * 4]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------------------
    if (tid < BM) {
        const int m = m0 + tid;
        int off = -1, roff = 0;
        if (m < p.M) {
            if (p.dense_out && !p.res) {
                off = m * p.ldo;
            } else {
                const RowCoord rc = kc_decode_row(m, p.OHc, p.OWc);
                off = kc_out_offset(p, rc);
                if (p.res) roff = kc_res_offset(p, rc);
            }
        }
        s_off[tid] = off;
        s_roff[tid] = roff;
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + r;
        if (n >= p.Nn) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = wm * TM * 32 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                const int off = s_off[row];
                if (off < 0) continue;
                float val = acc[i][j][v] + bv;
                if (p.res) val += p.res[s_roff[row] + n];
                if (p.act == CSLGAN_ACT_LRELU02) val = val > 0.f ? val : 0.2f * val;
                else if (p.act == CSLGAN_ACT_RELU) val = val > 0.f ? val : 0.f;
                else if (p.act == CSLGAN_ACT_TANH) val = tanhf(val);
                if (p.mask) val *= (p.mask[off + n] > 0.f ? 1.f : 0.2f);
                p.out[off + n] = val;
            }
        }
    }
}

// Repack KRSC filters into the per-parity-class [C][taps][K] matrices the data gradient consumes.
//   wt[class_off + (c*Tc + t)*K + k] = w[((k*R + kh_t)*S + kw_t)*C + c]
struct RepackArgs {
    int K, R, S, C;
    int n_class;
    int cls_off[4];     // float offset of each class matrix in wt
    int cls_T[4];
    signed char kh[4][IG_MAX_TAPS], kw[4][IG_MAX_TAPS];
};

__global__ void repack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wt, RepackArgs a) {
    const int cls = blockIdx.y;
    const int Tc = a.cls_T[cls];
    const long long total = (long long)a.C * Tc * a.K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % a.K);
        const long long rest = i / a.K;
        const int t = (int)(rest % Tc);
        const int c = (int)(rest / Tc);
        wt[a.cls_off[cls] + i] = w[(((long long)k * a.R + a.kh[cls][t]) * a.S + a.kw[cls][t]) * a.C + c];
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_kc_tile(const KcParams& p, bool vecA, bool vecB, hipStream_t st) {
    KcParams q = p;
    q.tiles_m = (p.M + BM - 1) / BM;
    q.tiles_n = (p.Nn + BN - 1) / BN;
    const dim3 grid((unsigned)(q.tiles_m * q.tiles_n)), block(256);
    if (vecA && vecB) hipLaunchKernelGGL((igemm_kc_kernel<BM, BN, WM, WN, true, true>), grid, block, 0, st, q);
    else if (vecA) hipLaunchKernelGGL((igemm_kc_kernel<BM, BN, WM, WN, true, false>), grid, block, 0, st, q);
    else if (vecB) hipLaunchKernelGGL((igemm_kc_kernel<BM, BN, WM, WN, false, true>), grid, block, 0, st, q);
    else hipLaunchKernelGGL((igemm_kc_kernel<BM, BN, WM, WN, false, false>), grid, block, 0, st, q);
    return check_launch("igemm_kc_kernel");
}

int launch_kc(const KcParams& p, hipStream_t st) {
    if (p.M <= 0 || p.Nn <= 0) return CSLGAN_OK;
    const bool vecA = (p.AC % 4 == 0) && aligned16(p.a);
    const bool vecB = (p.ldw % 4 == 0) && (p.Kdim % 4 == 0) && aligned16(p.w);
    if (p.Nn > 64) {
        // small-M problems fill the chip better with 64-row tiles
        const long long t128 = (long long)((p.M + 127) / 128) * ((p.Nn + 127) / 128);
        if (t128 < 192) return launch_kc_tile<64, 128, 1, 4>(p, vecA, vecB, st);
        return launch_kc_tile<128, 128, 2, 2>(p, vecA, vecB, st);
    }
    if (p.Nn > 32) return launch_kc_tile<128, 64, 2, 2>(p, vecA, vecB, st);
    return launch_kc_tile<128, 32, 4, 1>(p, vecA, vecB, st);
}

static int fill_conv_fwd(const cslgan_conv_t* c, KcParams& p) {
    CSLGAN_REQUIRE(c->R * c->S <= IG_MAX_TAPS, "conv: %dx%d filter has more than %d taps", c->R, c->S, IG_MAX_TAPS);
    const int VH = c->upsample ? 2 * c->H : c->H, VW = c->upsample ? 2 * c->W : c->W;
    const int P = (VH + 2 * c->pad - c->R) / c->stride + 1, Q = (VW + 2 * c->pad - c->S) / c->stride + 1;
    CSLGAN_REQUIRE(P == c->P && Q == c->Q, "conv: output %dx%d does not match P,Q=%d,%d", P, Q, c->P, c->Q);
    CSLGAN_REQUIRE((long long)c->N * c->P * c->Q * c->K < (1ll << 31) && (long long)c->N * c->H * c->W * c->C < (1ll << 40),
                   "conv: tensor too large for 32-bit output offsets");
    p.AH = c->H; p.AW = c->W; p.AC = c->C; p.VH = VH; p.VW = VW; p.ups = c->upsample ? 1 : 0;
    p.M = c->N * c->P * c->Q; p.OHc = c->P; p.OWc = c->Q; p.sy = p.sx = c->stride;
    p.T = c->R * c->S; p.Kdim = p.T * c->C;
    for (int t = 0; t < IG_MAX_TAPS; ++t) { p.ty[t] = 0; p.tx[t] = 0; }
    for (int kh = 0; kh < c->R; ++kh)
        for (int kw = 0; kw < c->S; ++kw) { p.ty[kh * c->S + kw] = (signed char)(kh - c->pad); p.tx[kh * c->S + kw] = (signed char)(kw - c->pad); }
    p.Nn = c->K; p.ldw = p.Kdim;
    p.OHf = c->P; p.OWf = c->Q; p.osy = p.osx = 1; p.oy0 = p.ox0 = 0; p.ldo = c->K; p.dense_out = 1;
    return CSLGAN_OK;
}

}  // namespace cslgan

using namespace cslgan;

extern "C" {

int cslgan_conv2d_fwd_f32(const cslgan_conv_t* c, const float* x, const float* w, const float* bias,
                          const float* residual, int res_shift, int act, float* y, void* stream) {
    CSLGAN_REQUIRE(c && x && w && y, "conv2d_fwd: null argument");
    CSLGAN_REQUIRE(c->N > 0 && c->H > 0 && c->W > 0 && c->C > 0 && c->K > 0 && c->R > 0 && c->S > 0 && c->stride > 0 && c->pad >= 0,
                   "conv2d_fwd: non-positive dimension");
    CSLGAN_REQUIRE(res_shift == 0 || res_shift == 1, "conv2d_fwd: res_shift must be 0 or 1");
    CSLGAN_REQUIRE(act >= 0 && act <= 3, "conv2d_fwd: unknown activation %d", act);
    KcParams p{};
    int rc = fill_conv_fwd(c, p);
    if (rc) return rc;
    CSLGAN_REQUIRE(!residual || res_shift == 0 || (c->P % 2 == 0 && c->Q % 2 == 0), "conv2d_fwd: shifted residual needs even output dims");
    p.a = x; p.w = w; p.out = y; p.bias = bias; p.res = residual; p.res_shift = res_shift; p.mask = nullptr; p.act = act;
    return launch_kc(p, (hipStream_t)stream);
}

int cslgan_conv2d_dgrad_f32(const cslgan_conv_t* c, const float* gy, const float* w, float* wt_ws, const float* mask,
                            float* gx, void* stream) {
    CSLGAN_REQUIRE(c && gy && w && wt_ws && gx, "conv2d_dgrad: null argument");
    CSLGAN_REQUIRE(!c->upsample, "conv2d_dgrad: upsample-on-read convs have no data-gradient path yet");
    CSLGAN_REQUIRE(c->stride >= 1 && c->stride <= 2, "conv2d_dgrad: stride %d unsupported", c->stride);
    CSLGAN_REQUIRE(c->R * c->S <= IG_MAX_TAPS, "conv2d_dgrad: too many taps");
    CSLGAN_REQUIRE((long long)c->N * c->H * c->W * c->C < (1ll << 31), "conv2d_dgrad: tensor too large");
    const int s = c->stride;
    hipStream_t st = (hipStream_t)stream;
    // ---- class tables + filter repack ----
    RepackArgs ra{};
    ra.K = c->K; ra.R = c->R; ra.S = c->S; ra.C = c->C; ra.n_class = s * s;
    int off = 0;
    for (int py = 0; py < s; ++py)
        for (int px = 0; px < s; ++px) {
            const int cls = py * s + px;
            int T = 0;
            for (int kh = 0; kh < c->R; ++kh) {
                if (((py + c->pad - kh) % s + s) % s != 0) continue;
                for (int kw = 0; kw < c->S; ++kw) {
                    if (((px + c->pad - kw) % s + s) % s != 0) continue;
                    ra.kh[cls][T] = (signed char)kh; ra.kw[cls][T] = (signed char)kw; ++T;
                }
            }
            ra.cls_T[cls] = T; ra.cls_off[cls] = off; off += T * c->K * c->C;
        }
    {
        unsigned gxn = (unsigned)(((long long)c->K * c->C * c->R * c->S / (s * s) + 255) / 256);
        if (gxn > 1024) gxn = 1024;
        if (gxn < 1) gxn = 1;
        hipLaunchKernelGGL(repack_dgrad_kernel, dim3(gxn, (unsigned)(s * s)), dim3(256), 0, st, w, wt_ws, ra);
        int rc = check_launch("repack_dgrad_kernel");
        if (rc) return rc;
    }
    for (int py = 0; py < s; ++py)
        for (int px = 0; px < s; ++px) {
            const int cls = py * s + px;
            const int OHc = (c->H - py + s - 1) / s, OWc = (c->W - px + s - 1) / s;
            if (OHc <= 0 || OWc <= 0) continue;
            KcParams p{};
            p.a = gy; p.AH = c->P; p.AW = c->Q; p.AC = c->K; p.VH = c->P; p.VW = c->Q; p.ups = 0;
            p.M = c->N * OHc * OWc; p.OHc = OHc; p.OWc = OWc; p.sy = p.sx = 1;
            p.T = ra.cls_T[cls]; p.Kdim = p.T * c->K;
            for (int t = 0; t < IG_MAX_TAPS; ++t) { p.ty[t] = 0; p.tx[t] = 0; }
            for (int t = 0; t < p.T; ++t) {
                p.ty[t] = (signed char)((py + c->pad - ra.kh[cls][t]) / s);
                p.tx[t] = (signed char)((px + c->pad - ra.kw[cls][t]) / s);
            }
            p.w = wt_ws + ra.cls_off[cls]; p.Nn = c->C; p.ldw = p.Kdim;
            p.out = gx; p.OHf = c->H; p.OWf = c->W; p.osy = p.osx = s; p.oy0 = py; p.ox0 = px; p.ldo = c->C;
            p.dense_out = (s == 1) ? 1 : 0;
            p.bias = nullptr; p.res = nullptr; p.res_shift = 0; p.mask = mask; p.act = CSLGAN_ACT_NONE;
            if (p.T == 0) {
                // no tap reaches this class: gradient is zero there (cannot happen for R,S >= stride)
                set_error("conv2d_dgrad: empty tap class");
                return CSLGAN_ERR_INVALID_ARG;
            }
            int rc = launch_kc(p, st);
            if (rc) return rc;
        }
    return CSLGAN_OK;
}

}  // extern "C"
