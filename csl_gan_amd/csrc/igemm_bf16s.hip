// bf16 STORAGE path (BASELINE.json configs[4]: "CelebA 128x128 DCResNet bf16 ... HBM-bound per-sample grads"; SURVEY §8(d)
// "bf16 storage / fp32 accumulate").  gfx950 only.
//
// igemm_bf16.hip computes on the bf16 matrix cores but reads fp32 tensors and rounds them on the way into LDS: a 128x128
// tile then moves 32 KB per 1.05 MFLOP and the kernels sit on L2 bandwidth at 0.10-0.13 of the bf16 MFMA peak.  Here the
// ACTIVATIONS, the ACTIVATION GRADIENTS and a pre-rounded copy of the FILTERS are bfloat16 in HBM (fp32 master weights, fp32
// accumulation, fp32 weight gradients / norms / clip / noise / Adam):
//
//   igemm_kcs_kernel   forward conv / linear and data gradient: a 16-byte global load is 8 consecutive k of one row = exactly
//                      one operand of v_mfma_f32_32x32x16_bf16, stored to LDS as it is (no conversion, half the bytes of the
//                      fp32 loader, K tile 64); the epilogue (bias, residual, activation, LeakyReLU mask) writes bf16 or fp32.
//   igemm_mcs_kernel   grouped / per-sample weight gradient: the reduction index (pixel) is the slow one of both operands, so a
//                      thread loads the same 8 channels of 8 consecutive pixels (8 x 16 bytes, a pixel row's 256 bytes
//                      coalesced across 16 lanes) and transposes the 8x8 block in registers into 8 LDS operands.
//   round / repack     the fp32 master filter rounded once per parameter version (plain KRSC order for the forward conv,
//                      per-parity-class [C][taps][K] matrices for the data gradient), cached by the caller.
//   casts, activation backward, bias gradient on bf16 tensors.
//
// Numerics: every product is bf16(a)*bf16(b) exactly, summed in fp32; what is new against CSLGAN_COMPUTE_BF16 on fp32 tensors is
// ONE extra rounding per stored activation / activation gradient (relative 2^-9), i.e. the same order as the operand rounding
// the matrix core needs anyway.
//
// Replaces (reference file:line): nn.Conv2d / nn.Linear forward and autograd data gradients (DCResNet_models.py:131-132,145),
// the Opacus-fork per-sample weight gradients (train.py:373,387), F.leaky_relu's backward, bias gradients.
#include <stdlib.h>
#include "common.h"
#include "igemm.h"

namespace cslgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned S_OOB16 = 0xFFFFFFF0u;

__device__ __forceinline__ unsigned short f2bf(float v) {      // round to nearest even (v_cvt_pk_bf16_f32 semantics)
    const f32x2 t = {v, 0.f};
    return (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2)) & 0xffffu);
}
__device__ __forceinline__ unsigned f2bf_pk(float lo, float hi) {
    const f32x2 t = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
}
__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }

// ---- K-contiguous form on bf16 tensors -------------------------------------------------------------------------------------
struct KsParams {
    const void* a;               // bf16 [img][AH][AW][AC]
    unsigned a_bytes, w_bytes;
    unsigned ac_recip;           // ceil(2^32 / AC)
    int AH, AW, AC, VH, VW, sy, sx;
    const void* w;               // bf16 class matrices [Nn][Kdim] at element offset cls[].w_off
    int Nn;
    void* out;                   // bf16 or fp32 (template)
    int OHf, OWf, osy, osx, ldo, dense_out;
    const float* bias;
    const void* res;             // same indexing as out; bf16 when res_bf16
    int res_bf16;
    const void* mask;            // same indexing as out; bf16 when mask_bf16
    int mask_bf16;
    int act;
    int n_cls, tiles_m, tiles_n;
    const void* w3;              // igemm_halos: the filter as bf16 in step-major order [chunk * T + tap][n][16 k] (split_filter_x3_kernel<1>)
    int w3_off[IG_MAX_CLS];      // igemm_halos: element offset of each class's step-major matrix in w3 (data-gradient parity classes)
    KcClass cls[IG_MAX_CLS];
};

constexpr int KS_BK = 64;        // k per LDS tile = 8 entries of 8 bf16 (16 bytes)

// Epilogue of the gather and halo kernels.  The MFMAs are issued with the FILTER fragment as the first operand and the pixel fragment
// as the second, so the accumulator's lane index r is the PIXEL and its register index v the output channel
// (v & 3) + 8 (v >> 2) + 4 h: a lane holds 4 x 4 consecutive channels of one pixel.  Bias, residual and mask are read as 4-element
// vectors and the result leaves as one 8-byte (bf16) or 16-byte (fp32) store per group; the lanes (r, h = 0) and (r, h = 1) write
// adjacent pieces.  With the pixel on v (the usual operand order) every element was its own 2-byte store, 64 B contiguous per 32
// lanes: 64 store instructions per lane and tile, 0.05 of the 0.30 ms of the critic's first 64-channel conv.
template <int TM, int TN, bool OUT_BF16>
__device__ __forceinline__ void ks_epilogue(const f32x16 (&acc)[TM][TN], const KsParams& p, const int* s_off, int row0, int col0, int r, int h) {
    // vector path: 4-channel groups never straddle the channel count or a row, and every base pointer is 16-byte aligned
    const bool vec = !(p.Nn & 3) && !(p.ldo & 3) &&
        !((reinterpret_cast<unsigned long long>(p.out) | reinterpret_cast<unsigned long long>(p.res) | reinterpret_cast<unsigned long long>(p.mask) |
           reinterpret_cast<unsigned long long>(p.bias)) & 15ull);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int off = s_off[row0 + i * 32 + r];
        if (off < 0) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = col0 + j * 32 + 8 * g + 4 * h;
                if (n >= p.Nn) continue;
                float val[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = acc[i][j][4 * g + e];
                if (vec) {
                    if (p.bias) {
                        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
                        val[0] += b.x; val[1] += b.y; val[2] += b.z; val[3] += b.w;
                    }
                    if (p.res) {
                        if (p.res_bf16) {
                            const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.res) + off + n);
                            val[0] += __uint_as_float(u.x << 16); val[1] += __uint_as_float(u.x & 0xffff0000u);
                            val[2] += __uint_as_float(u.y << 16); val[3] += __uint_as_float(u.y & 0xffff0000u);
                        } else {
                            const float4 u = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + off + n);
                            val[0] += u.x; val[1] += u.y; val[2] += u.z; val[3] += u.w;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (p.act == CSLGAN_ACT_LRELU02) val[e] = val[e] > 0.f ? val[e] : 0.2f * val[e];
                        else if (p.act == CSLGAN_ACT_RELU) val[e] = val[e] > 0.f ? val[e] : 0.f;
                        else if (p.act == CSLGAN_ACT_TANH) val[e] = tanhf(val[e]);
                    }
                    if (p.mask) {
                        float mv[4];
                        if (p.mask_bf16) {
                            const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.mask) + off + n);
                            mv[0] = __uint_as_float(u.x << 16); mv[1] = __uint_as_float(u.x & 0xffff0000u);
                            mv[2] = __uint_as_float(u.y << 16); mv[3] = __uint_as_float(u.y & 0xffff0000u);
                        } else {
                            const float4 u = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.mask) + off + n);
                            mv[0] = u.x; mv[1] = u.y; mv[2] = u.z; mv[3] = u.w;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) val[e] *= (mv[e] > 0.f ? 1.f : 0.2f);
                    }
                    if (OUT_BF16) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.out) + off + n) = make_uint2(f2bf_pk(val[0], val[1]), f2bf_pk(val[2], val[3]));
                    else *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + off + n) = make_float4(val[0], val[1], val[2], val[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e >= p.Nn) continue;
                        float x = val[e] + (p.bias ? p.bias[n + e] : 0.f);
                        if (p.res) x += p.res_bf16 ? bf2f(reinterpret_cast<const unsigned short*>(p.res)[off + n + e]) : reinterpret_cast<const float*>(p.res)[off + n + e];
                        if (p.act == CSLGAN_ACT_LRELU02) x = x > 0.f ? x : 0.2f * x;
                        else if (p.act == CSLGAN_ACT_RELU) x = x > 0.f ? x : 0.f;
                        else if (p.act == CSLGAN_ACT_TANH) x = tanhf(x);
                        if (p.mask) {
                            const float mv = p.mask_bf16 ? bf2f(reinterpret_cast<const unsigned short*>(p.mask)[off + n + e]) : reinterpret_cast<const float*>(p.mask)[off + n + e];
                            x *= (mv > 0.f ? 1.f : 0.2f);
                        }
                        if (OUT_BF16) reinterpret_cast<unsigned short*>(p.out)[off + n + e] = f2bf(x);
                        else reinterpret_cast<float*>(p.out)[off + n + e] = x;
                    }
                }
            }
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool OUT_BF16>
__global__ __launch_bounds__(256, 2) void igemm_kcs_kernel(const KsParams p) {
    constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "bad tile");
    constexpr int A_ES = BM + 1, B_ES = BN + 1;             // uint4 (16-byte) units per LDS entry, padded by 16 B
    constexpr int A_PASS = BM / 32, B_PASS = BN / 32;
    __shared__ __attribute__((aligned(16))) uint4 As[2][8 * A_ES];
    __shared__ __attribute__((aligned(16))) uint4 Bs[2][8 * B_ES];
    __shared__ int s_tap[IG_MAX_TAPS];
    __shared__ int s_off[BM];

    const int tid = threadIdx.x;
    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tile_mg = wg / p.tiles_n, tile_n = wg - tile_mg * p.tiles_n;
    int ci = 0;
#pragma unroll 1
    while (ci + 1 < p.n_cls && tile_mg >= p.cls[ci + 1].tile0) ++ci;
    const KcClass& kc = p.cls[ci];
    const int M = kc.M, OHc = kc.OHc, OWc = kc.OWc, Kdim = kc.Kdim;
    const int m0 = (tile_mg - kc.tile0) * BM, n0 = tile_n * BN;

    if (tid < IG_MAX_TAPS) s_tap[tid] = ((int)kc.ty[tid] << 16) | ((int)kc.tx[tid] & 0xffff);

    const int lrow = tid >> 3;   // 0..31
    const int e = tid & 7;       // LDS entry (8 consecutive k) within the 64-k tile: 8 lanes read 128 contiguous bytes of a row
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(reinterpret_cast<const unsigned short*>(p.w) + kc.w_off), 0, p.w_bytes - 2u * (unsigned)kc.w_off, 0x00020000);
    int a_img[A_PASS], a_iy[A_PASS], a_ix[A_PASS];
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
        const int m = m0 + lrow + 32 * i;
        const bool ok = m < M;
        const RowCoord rc = kc_decode_row(ok ? m : 0, OHc, OWc, kc.patch);
        a_img[i] = rc.img * p.AH * p.AW * p.AC;
        a_iy[i] = ok ? rc.oy * p.sy : -(1 << 20);
        a_ix[i] = rc.ox * p.sx;
    }
    unsigned b_off[B_PASS];
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
        const int n = n0 + lrow + 32 * i;
        b_off[i] = n < p.Nn ? 2u * (unsigned)n * (unsigned)Kdim : S_OOB16;
    }
    __syncthreads();

    // Two register sets: the loads of K tile t+2 are issued before the MFMAs of tile t and written to LDS after the MFMAs of tile
    // t+1 — two tiles of matrix work (>= 2 x 512 MFMA cycles per wave) to cover the L2 / HBM latency of a gathered load.  With one
    // set (loads one tile ahead) the kernel waited on its loads every iteration: 390-500 TF.
    u32x4 ra0[A_PASS], rb0[B_PASS], ra1[A_PASS], rb1[B_PASS];
    auto load_tile = [&](int kt, u32x4 (&ra)[A_PASS], u32x4 (&rb)[B_PASS]) {
        // Branch-free offsets: an invalid element ORs 0xFFFFFFF0 into its (always computed) offset, which the buffer descriptor's range
        // check turns into a zero load.  Written as `ok ? offset : OOB` the compiler put every load into its own exec-masked block
        // (nine s_and_saveexec per half iteration), which also kept it from issuing the eight loads back to back.
        const int kb = kt * KS_BK + e * 8;
        const bool kin = kb < Kdim;
        const int t = (int)__umulhi((unsigned)(kin ? kb : 0), p.ac_recip);
        const int c = (kin ? kb : 0) - t * p.AC;
        const int tap = s_tap[t];
        const int ty = tap >> 16, tx = (int)(short)(tap & 0xffff);
        const unsigned kbad = kin ? 0u : S_OOB16;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            const int iy = a_iy[i] + ty, ix = a_ix[i] + tx;
            const unsigned bad = ((unsigned)iy < (unsigned)p.VH && (unsigned)ix < (unsigned)p.VW) ? kbad : S_OOB16;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)((2u * (unsigned)(a_img[i] + (iy * p.AW + ix) * p.AC + c)) | bad), 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i)
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)((b_off[i] + 2u * (unsigned)kb) | kbad | (b_off[i] == S_OOB16 ? S_OOB16 : 0u)), 0, 0);
    };
    auto store_tile = [&](int buf, const u32x4 (&ra)[A_PASS], const u32x4 (&rb)[B_PASS]) {
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) As[buf][e * A_ES + lrow + 32 * i] = __builtin_bit_cast(uint4, ra[i]);
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) Bs[buf][e * B_ES + lrow + 32 * i] = __builtin_bit_cast(uint4, rb[i]);
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

    auto mma_tile = [&](int buf) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = __builtin_bit_cast(bf16x8, As[buf][(2 * s + h) * A_ES + arow0 + i * 32]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = __builtin_bit_cast(bf16x8, Bs[buf][(2 * s + h) * B_ES + brow0 + j * 32]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);   // filter first: see ks_epilogue
        }
    };

    const int nk = (Kdim + KS_BK - 1) / KS_BK;
    load_tile(0, ra0, rb0);
    load_tile(1, ra1, rb1);          // past the last tile every offset is out of range -> zeros, never multiplied
    store_tile(0, ra0, rb0);
    __syncthreads();
    // ONE loop exit: with a break between the halves the epilogue had two live-in states for the accumulators and the register
    // allocator copied all 64 of them (32 v_mov_b64) plus the staging registers between the halves of every iteration — 107 moves per
    // 16 MFMAs, most of the "8 vector instructions per MFMA" the PMC counters showed.  An odd tile count multiplies one all-zero tile.
    for (int kt = 0; kt < nk; kt += 2) {
        load_tile(kt + 2, ra0, rb0);
        mma_tile(0);
        store_tile(1, ra1, rb1);     // tile kt + 1 (zeros past the end)
        __syncthreads();
        load_tile(kt + 3, ra1, rb1);
        mma_tile(1);
        store_tile(0, ra0, rb0);     // tile kt + 2
        __syncthreads();
    }

    // ---- epilogue: output offsets per row (residual and mask share them), then bias / residual / activation / mask ----------
    if (tid < BM) {
        const int m = m0 + tid;
        int off = -1;
        if (m < M) {
            if (p.dense_out && !kc.patch) {
                off = m * p.ldo;
            } else {
                const RowCoord rc = kc_decode_row(m, OHc, OWc, kc.patch);
                off = ((rc.img * p.OHf + rc.oy * p.osy + kc.oy0) * p.OWf + rc.ox * p.osx + kc.ox0) * p.ldo;
            }
        }
        s_off[tid] = off;
    }
    __syncthreads();
    ks_epilogue<TM, TN, OUT_BF16>(acc, p, s_off, wm * TM * 32, n0 + wn * TN * 32, r, h);
}

// ---- stride-1 convs with an LDS-resident input halo (the generator's 5x5 / 3x3 convs on bf16 activations) ---------------------
// igemm_halo_x3_kernel<BN, 1> (igemm_bf16.hip) with bfloat16 activations in HBM: a workgroup owns two 8x8 output patches x BN
// channels; per 16-channel chunk the (8+R-1) x (8+S-1) halo of each patch is staged in LDS ONCE (a pixel's 16 channels are two
// 16-byte loads, stored as loaded — no conversion) and every tap reads its A fragments from that image at a per-tap offset; the
// filter never touches LDS: it is streamed pre-rounded in step-major order straight into a 4-deep register ring.  Against the
// gather form above the input is read from L2 / HBM once per chunk instead of once per tap.
constexpr int HS_MAX = 12 * 12;        // pixels per patch halo (8+4 squared: up to 5x5 taps)

template <int BN, bool OUT_BF16>
__global__ __launch_bounds__(256, 2) void igemm_halos_kernel(const KsParams p) {
    constexpr int RING = 4;
    constexpr int PATCHES = 2, WN = 2;                       // waves 2 (M: one 8x8 patch each) x 2 (N halves)
    constexpr int BM = 64 * PATCHES, TM = 2, TN = BN / (32 * WN);
    __shared__ __attribute__((aligned(16))) uint2 Hs[PATCHES * HS_MAX * 4];     // [patch][pixel][4 x (4 ch bf16)]
    __shared__ int s_tapoff[IG_MAX_TAPS];
    __shared__ int s_off[BM];

    const int tid = threadIdx.x;
    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tile_mg = wg / p.tiles_n, tile_n = wg - tile_mg * p.tiles_n;
    int ci = 0;
#pragma unroll 1
    while (ci + 1 < p.n_cls && tile_mg >= p.cls[ci + 1].tile0) ++ci;      // data gradient: one class per output parity, each its own tiles
    const KcClass& kc = p.cls[ci];
    const int M = kc.M, OHc = kc.OHc, OWc = kc.OWc, T = kc.T;
    const int m0 = (tile_mg - kc.tile0) * BM, n0 = tile_n * BN;
    const int HW_ = kc.halo_w, hpix = kc.halo_h * kc.halo_w;
    const int img_stride = p.AH * p.AW * p.AC;

    // tap offset in uint2 units; bit 0 = parity of the tap's halo-row offset (selects the swizzled base)
    if (tid < IG_MAX_TAPS) s_tapoff[tid] = ((((int)kc.ty[tid] - kc.ty_min) * HW_ + ((int)kc.tx[tid] - kc.tx_min)) * 4) | (((int)kc.ty[tid] - kc.ty_min) & 1);

    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, p.a_bytes, 0x00020000);
    int p_img[PATCHES], p_y0[PATCHES], p_x0[PATCHES];
    bool p_ok[PATCHES];
#pragma unroll
    for (int pp = 0; pp < PATCHES; ++pp) {
        const int m = m0 + 64 * pp;
        p_ok[pp] = m < M;
        const RowCoord rc = kc_decode_row(p_ok[pp] ? m : 0, OHc, OWc, 1);     // first row of the patch = its top-left pixel
        p_img[pp] = rc.img * img_stride;
        p_y0[pp] = rc.oy + kc.ty_min;
        p_x0[pp] = rc.ox + kc.tx_min;
    }
    const int lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid / WN, wn = wid - wm * WN;             // wm = patch index
    // filter: lane (r, h), tile j reads 8 consecutive k of filter row n0 + wn*TN*32 + j*32 + r of one step = one 16-byte load
    const __amdgpu_buffer_rsrc_t w3_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(reinterpret_cast<const unsigned short*>(p.w3) + p.w3_off[ci]), 0, 2u * (unsigned)p.Nn * (unsigned)kc.Kdim, 0x00020000);
    const unsigned step_bytes = 32u * (unsigned)p.Nn;
    unsigned b_off[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + r;
        b_off[j] = n < p.Nn ? 32u * (unsigned)n + 16u * (unsigned)h : S_OOB16;
    }
    u32x4 rb[RING][TN];
    const int n_steps = (p.AC >> 4) * T;
    auto load_b = [&](int step, int slot) {                 // filter slice of `step` (= chunk * T + tap); zeros past the end
        const unsigned kb = (unsigned)step * step_bytes;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            rb[slot][j] = __builtin_amdgcn_raw_buffer_load_b128(w3_rsrc, (int)((b_off[j] + kb) | ((step >= n_steps || b_off[j] == S_OOB16) ? S_OOB16 : 0u)), 0, 0);
        }
    };
    // halo staging: PATCHES x hpix pixels x 2 halves of 8 channels; <= PATCHES*144*2/256 = 2.25 / 4.5 16-byte loads per thread
    constexpr int HREG = (PATCHES * HS_MAX * 2 + 255) / 256;
    u32x4 rh[HREG];
    const int h_total = PATCHES * hpix * 2;
    auto fetch_halo = [&](int cc) {
#pragma unroll
        for (int j = 0; j < HREG; ++j) {
            const int idx = tid + 256 * j;
            const int half = idx & 1, pixg = idx >> 1;
            const int pp = idx < h_total ? pixg / hpix : 0;
            const int pix = pixg - pp * hpix;
            const int hy = pix / HW_, hx = pix - hy * HW_;
            // (select chains, not p_y0[pp]: a dynamically indexed register array goes to scratch)
            int sy0 = p_y0[0], sx0 = p_x0[0], simg = p_img[0];
            bool sok = p_ok[0];
#pragma unroll
            for (int q = 1; q < PATCHES; ++q) {
                sy0 = pp == q ? p_y0[q] : sy0; sx0 = pp == q ? p_x0[q] : sx0; simg = pp == q ? p_img[q] : simg; sok = pp == q ? p_ok[q] : sok;
            }
            const int iy = sy0 + hy, ix = sx0 + hx;
            const bool ok = idx < h_total && sok && (unsigned)iy < (unsigned)p.VH && (unsigned)ix < (unsigned)p.VW;
            rh[j] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)((2u * (unsigned)(simg + (iy * p.AW + ix) * p.AC + cc * 16 + half * 8)) | (ok ? 0u : S_OOB16)), 0, 0);
        }
    };
    auto commit_halo = [&]() {
#pragma unroll
        for (int j = 0; j < HREG; ++j) {
            const int idx = tid + 256 * j;
            if (idx < h_total) {
                const int half = idx & 1, pixg = idx >> 1;
                const int pp = pixg / hpix;
                const int pix = pixg - pp * hpix;
                // the two 16-byte halves of a pixel are swapped on odd halo rows (the two patch rows a 16-lane ds_read_b128 group
                // covers then hit disjoint banks)
                const int at = (pp * HS_MAX + pix) * 4 + 2 * (half ^ ((pix / HW_) & 1));
                *reinterpret_cast<uint4*>(&Hs[at]) = __builtin_bit_cast(uint4, rh[j]);
            }
        }
    };

    // uint2 offset of this lane's pixel (tap 0,0 corner) per MFMA tile; a tap landing on an odd halo row reads the other 16-byte half
    // of the pixel (half-swap swizzle): a_pix + (a_half ^ (odd << 1)) — arithmetic, not a [2][TM] table (that went to scratch)
    int a_pix[TM], a_half[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int qq = i * 32 + r;                           // row within the patch
        a_pix[i] = (wm * HS_MAX + (qq >> 3) * HW_ + (qq & 7)) * 4;
        a_half[i] = (h ^ ((qq >> 3) & 1)) << 1;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    fetch_halo(0);
#pragma unroll
    for (int q = 0; q < RING; ++q) load_b(q, q);
    commit_halo();
    __syncthreads();

    bf16x8 af_n[TM];
    auto read_a = [&](int t) {
        const int tw = s_tapoff[t], toff = tw & ~1, odd = tw & 1;
#pragma unroll
        for (int i = 0; i < TM; ++i) af_n[i] = *reinterpret_cast<const bf16x8*>(&Hs[a_pix[i] + (a_half[i] ^ (odd << 1)) + toff]);
    };
    read_a(0);

    auto k_step = [&](int s, int SL) {
        const int cc = s / T, t = s - cc * T;
        const int t_fetch = T > 3 ? T - 3 : 0;
        const bool last_chunk = (cc + 1) * 16 >= p.AC;
        if (t == t_fetch && !last_chunk) fetch_halo(cc + 1);
        bf16x8 bf[TN], af[TM];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = __builtin_bit_cast(bf16x8, rb[SL][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = af_n[i];
        load_b(s + RING, SL);
        const bool boundary = t == T - 1;
        if (!boundary) read_a(t + 1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);   // filter first: see ks_epilogue
        if (boundary) {
            if (!last_chunk) {                 // the halo image is the only shared state: one barrier pair per CHUNK, none per tap
                __syncthreads();
                commit_halo();
                __syncthreads();
            }
            if (s + 1 < n_steps) read_a(0);
        }
    };
    for (int s = 0; s < n_steps; s += RING) {
        k_step(s, 0);
        if (s + 1 < n_steps) k_step(s + 1, 1);
        if (s + 2 < n_steps) k_step(s + 2, 2);
        if (s + 3 < n_steps) k_step(s + 3, 3);
    }
    __syncthreads();

    if (tid < BM) {
        const int m = m0 + tid;
        int off = -1;
        if (m < M) {
            const RowCoord rc = kc_decode_row(m, OHc, OWc, 1);
            off = ((rc.img * p.OHf + rc.oy * p.osy + kc.oy0) * p.OWf + rc.ox * p.osx + kc.ox0) * p.ldo;
        }
        s_off[tid] = off;
    }
    __syncthreads();
    ks_epilogue<TM, TN, OUT_BF16>(acc, p, s_off, wm * 64, n0 + wn * TN * 32, r, h);
}

// ---- 1x1 convs with 16 / 32 / 64 input channels (the generator's shortcut convs on the shuffled C/4 channels, DCResNet_models.py:22) ----
// A stream over the pixels: the whole reduction is 1-4 MFMA k-steps, so the gather kernel's 64-k tile was mostly padding (C = 16: a
// quarter of it) and its LDS round trip pure overhead — 0.26 ms for a layer whose traffic (67 MB in, 268 MB out at 128x128) is 0.08 ms.
// Here a wavefront owns 32 pixels: its A operand is one 16-byte load per k-step straight from the bf16 rows (a pixel's channels are
// contiguous, 32 pixels = 512-2048 contiguous bytes), the filter lives in registers for the whole kernel, no LDS, no barriers.
template <int CK, int TN, bool OUT_BF16>
__global__ __launch_bounds__(256) void conv1x1s_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ wb, const float* __restrict__ bias,
                                                       long long M, int K, int act, void* __restrict__ y) {
    constexpr int C = CK * 16;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    bf16x8 bw[TN][CK];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int s = 0; s < CK; ++s) {
            const int n = 32 * j + r;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (n < K) v = *reinterpret_cast<const uint4*>(wb + (long long)n * C + 16 * s + 8 * h);
            bw[j][s] = __builtin_bit_cast(bf16x8, v);
        }
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = (bias && 32 * j + r < K) ? bias[32 * j + r] : 0.f;
    const long long n_tiles = (M + 31) / 32;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * 4;
    for (long long t = wave0; t < n_tiles; t += n_waves) {
        const long long m = 32 * t + r;
        f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
#pragma unroll
        for (int s = 0; s < CK; ++s) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (m < M) v = *reinterpret_cast<const uint4*>(x + m * C + 16 * s + 8 * h);
            const bf16x8 a = __builtin_bit_cast(bf16x8, v);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[j][s], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = 32 * j + r;
            if (n >= K) continue;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const long long row = 32 * t + (v & 3) + 8 * (v >> 2) + 4 * h;
                if (row >= M) continue;
                float val = acc[j][v] + bv[j];
                if (act == CSLGAN_ACT_LRELU02) val = val > 0.f ? val : 0.2f * val;
                else if (act == CSLGAN_ACT_RELU) val = val > 0.f ? val : 0.f;
                else if (act == CSLGAN_ACT_TANH) val = tanhf(val);
                if (OUT_BF16) reinterpret_cast<unsigned short*>(y)[row * K + n] = f2bf(val);
                else reinterpret_cast<float*>(y)[row * K + n] = val;
            }
        }
    }
}

static bool conv1x1s_eligible(const cslgan_conv_t* c, const void* residual) {
    static const int env = [] { const char* e = getenv("CSLGAN_CONV1X1S"); return e ? atoi(e) : 1; }();
    return env && !residual && c->R == 1 && c->S == 1 && c->stride == 1 && c->pad == 0 && (c->C == 16 || c->C == 32 || c->C == 64) &&
           c->K % 32 == 0 && c->K >= 32 && c->K <= 128 && (long long)c->N * c->H * c->W >= 65536;
}

template <int CK>
static int launch_conv1x1s_ck(const cslgan_conv_t* c, const void* x, const void* wb, const float* bias, int act, void* y, int y_bf16, hipStream_t st) {
    const long long M = (long long)c->N * c->H * c->W;
    long long nb = (M / 32 + 3) / 4;
    nb = nb > 2048 ? 2048 : (nb < 1 ? 1 : nb);
    const dim3 grid((unsigned)nb), block(256);
    const unsigned short* xh = reinterpret_cast<const unsigned short*>(x);
    const unsigned short* wh = reinterpret_cast<const unsigned short*>(wb);
    note_kernel("conv1x1s_kernel<%d>", CK * 16);
#define CSL_C1S(TN)                                                                                                          \
    do {                                                                                                                     \
        if (y_bf16) hipLaunchKernelGGL((conv1x1s_kernel<CK, TN, true>), grid, block, 0, st, xh, wh, bias, M, c->K, act, y);   \
        else hipLaunchKernelGGL((conv1x1s_kernel<CK, TN, false>), grid, block, 0, st, xh, wh, bias, M, c->K, act, y);         \
    } while (0)
    const int tn = c->K / 32;
    if (tn == 1) CSL_C1S(1);
    else if (tn == 2) CSL_C1S(2);
    else if (tn == 3) CSL_C1S(3);
    else CSL_C1S(4);
#undef CSL_C1S
    return check_launch("conv1x1s_kernel");
}

int split_filter_x3(const float* w, int Nn, int T, int C, void* w3, hipStream_t st, int pieces);      // igemm_bf16.hip

// stride-1 classes (one for a forward conv; the output-parity classes of a strided data gradient), each on an 8x8-patchable grid of at
// least 16x16, channels a multiple of 16, 2..25 taps within a 12x12 halo, >= 64 filters
static bool halos_eligible(const KsParams& p) {
    static const int env = [] { const char* e = getenv("CSLGAN_HALOS"); return e ? atoi(e) : 1; }();
    if (!env || p.n_cls < 1 || p.sy != 1 || p.sx != 1 || (p.AC & 15) || p.Nn < 64 || !aligned16(p.a)) return false;
    for (int c = 0; c < p.n_cls; ++c) {
        const KcClass& k = p.cls[c];
        if (k.T < 2 || (k.M & 63) || (k.OHc & 7) || (k.OWc & 7) || k.OHc < 16 || k.OWc < 16) return false;
        int ymin = 127, ymax = -128, xmin = 127, xmax = -128;
        for (int t = 0; t < k.T; ++t) {
            ymin = k.ty[t] < ymin ? k.ty[t] : ymin; ymax = k.ty[t] > ymax ? k.ty[t] : ymax;
            xmin = k.tx[t] < xmin ? k.tx[t] : xmin; xmax = k.tx[t] > xmax ? k.tx[t] : xmax;
        }
        if (ymax - ymin > 4 || xmax - xmin > 4) return false;
    }
    return true;
}

static int launch_halos(KsParams& p, bool out_bf16, hipStream_t st) {
    int tm = 0;
    long long w_el = 0;
    for (int c = 0; c < p.n_cls; ++c) {
        KcClass& k = p.cls[c];
        int ymin = 127, ymax = -128, xmin = 127, xmax = -128;
        for (int t = 0; t < k.T; ++t) {
            ymin = k.ty[t] < ymin ? k.ty[t] : ymin; ymax = k.ty[t] > ymax ? k.ty[t] : ymax;
            xmin = k.tx[t] < xmin ? k.tx[t] : xmin; xmax = k.tx[t] > xmax ? k.tx[t] : xmax;
        }
        k.ty_min = ymin; k.tx_min = xmin; k.halo_h = 8 + ymax - ymin; k.halo_w = 8 + xmax - xmin;
        k.patch = 1; k.tile0 = tm;
        tm += (k.M + 127) / 128;
        p.w3_off[c] = k.w_off;              // a class's step-major matrix has the size of its plain one: same offsets
        w_el += (long long)p.Nn * k.Kdim;
    }
    const KcClass& k0 = p.cls[0];
    const long long n_img = k0.M / ((long long)k0.OHc * k0.OWc);
    const long long a_b = 2ll * n_img * p.AH * p.AW * p.AC;
    CSLGAN_REQUIRE(a_b < 0xFFFFFFF0ll && 2ll * w_el < 0xFFFFFFF0ll, "igemm_halos: operand larger than 4 GB");
    p.a_bytes = (unsigned)a_b;
    const bool wide = p.Nn > 64;
    // (A four-patch form for the 64-filter layers — one patch per wave, each wave all 64 filters, four MFMAs per step instead of two —
    // was built and measured same-box: 312 vs 396 TF on the 128x128x64 layer, 257 vs 324 and 309 vs 363 on the others.  The steps are
    // paced by the filter slices every wave streams from L1 / L2, not by MFMAs per operand load; it was removed again.)
    p.tiles_m = tm;
    p.tiles_n = wide ? (p.Nn + 127) / 128 : 1;
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n)), block(256);
    note_kernel("igemm_halos_kernel<%d>", wide ? 128 : 64);
    if (wide) {
        if (out_bf16) hipLaunchKernelGGL((igemm_halos_kernel<128, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((igemm_halos_kernel<128, false>), grid, block, 0, st, p);
    } else {
        if (out_bf16) hipLaunchKernelGGL((igemm_halos_kernel<64, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((igemm_halos_kernel<64, false>), grid, block, 0, st, p);
    }
    return check_launch("igemm_halos_kernel");
}

template <int BM, int BN, int WM, int WN>
static int launch_kcs_tile(KsParams& p, bool out_bf16, hipStream_t st) {
    int tm = 0;
    for (int c = 0; c < p.n_cls; ++c) {
        p.cls[c].tile0 = tm;
        tm += (p.cls[c].M + BM - 1) / BM;
    }
    p.tiles_m = tm;
    p.tiles_n = (p.Nn + BN - 1) / BN;
    const long long tiles = (long long)p.tiles_m * p.tiles_n;
    if (tiles > 0x7fffffffll) { set_error("igemm_kcs: grid too large"); return CSLGAN_ERR_INVALID_ARG; }
    const dim3 grid((unsigned)tiles), block(256);
    note_kernel("igemm_kcs_kernel<%d,%d>", BM, BN);
    if (out_bf16) hipLaunchKernelGGL((igemm_kcs_kernel<BM, BN, WM, WN, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((igemm_kcs_kernel<BM, BN, WM, WN, false>), grid, block, 0, st, p);
    return check_launch("igemm_kcs_kernel");
}

static int launch_kcs(KsParams& p, bool out_bf16, hipStream_t st) {
    long long rows = 0, rows128 = 0;
    for (int c = 0; c < p.n_cls; ++c) { rows += p.cls[c].M; rows128 += (p.cls[c].M + 127) / 128; }
    if (rows <= 0 || p.Nn <= 0) return CSLGAN_OK;
    const long long n_img = p.cls[0].M / ((long long)p.cls[0].OHc * p.cls[0].OWc);
    const long long a_b = 2ll * n_img * p.AH * p.AW * p.AC;
    long long w_end = 0, kmax = 0;
    for (int c = 0; c < p.n_cls; ++c) {
        const long long en = (long long)p.cls[c].w_off + (long long)p.Nn * p.cls[c].Kdim;
        w_end = en > w_end ? en : w_end;
        kmax = p.cls[c].Kdim > kmax ? p.cls[c].Kdim : kmax;
        CSLGAN_REQUIRE(p.cls[c].Kdim % 8 == 0 && p.cls[c].w_off % 8 == 0, "igemm_kcs: reduction length must be a multiple of 8");
    }
    CSLGAN_REQUIRE(p.AC % 8 == 0 && aligned16(p.a) && aligned16(p.w), "igemm_kcs: channels must be a multiple of 8 and operands 16-byte aligned");
    CSLGAN_REQUIRE(a_b < 0xFFFFFFF0ll && 2 * w_end < 0xFFFFFFF0ll, "igemm_kcs: operand larger than 4 GB");
    CSLGAN_REQUIRE((kmax + KS_BK) * (long long)p.AC < (1ll << 32), "igemm_kcs: K too large for reciprocal division");
    p.a_bytes = (unsigned)a_b;
    p.w_bytes = (unsigned)(2 * w_end);
    p.ac_recip = (unsigned)(((1ull << 32) + (unsigned long long)p.AC - 1) / (unsigned long long)p.AC);
    for (int c = 0; c < p.n_cls; ++c) {
        KcClass& k = p.cls[c];
        k.patch = (k.T > 1 && k.OHc % 8 == 0 && k.OWc % 8 == 0) ? 1 : 0;
    }
    if (p.Nn <= 64) return launch_kcs_tile<128, 64, 2, 2>(p, out_bf16, st);
    if (rows128 * ((p.Nn + 127) / 128) >= 256) return launch_kcs_tile<128, 128, 2, 2>(p, out_bf16, st);
    return launch_kcs_tile<64, 128, 1, 4>(p, out_bf16, st);
}

// ---- filters: fp32 master -> bf16 operand copies ----------------------------------------------------------------------------
__global__ void round_bf16_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, long long n) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        reinterpret_cast<uint2*>(out)[i] = make_uint2(f2bf_pk(v.x, v.y), f2bf_pk(v.z, v.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[(n4 << 2) + threadIdx.x] = f2bf(in[(n4 << 2) + threadIdx.x]);
}

__global__ void widen_bf16_kernel(const unsigned short* __restrict__ in, float* __restrict__ out, long long n) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const uint2 v = reinterpret_cast<const uint2*>(in)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                                                        __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[(n4 << 2) + threadIdx.x] = bf2f(in[(n4 << 2) + threadIdx.x]);
}

static unsigned stream_blocks(long long n_items) {
    long long nb = (n_items + 255) / 256;
    return (unsigned)(nb > 4096 ? 4096 : (nb < 1 ? 1 : nb));
}

// data-gradient classes: wt[off_cls + (c*Tc + t)*K + k] = bf16(w[((k*R + kh)*S + kw)*C + c]) for the (kh, kw) of the class's tap t
struct DgradRepack {
    int K, R, S, C, n_cls;
    int step_major;      // 1: igemm_halos' layout [(k/16) * Tc + t][c][k % 16] per class instead of the plain [c][t][k]
    int cls_off[IG_MAX_CLS], cls_T[IG_MAX_CLS];
    signed char kh[IG_MAX_CLS][IG_MAX_TAPS], kw[IG_MAX_CLS][IG_MAX_TAPS];
};

__global__ void repack_dgrad_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wt, DgradRepack a) {
    const int cls = blockIdx.y;
    const int Tc = a.cls_T[cls];
    const long long total = (long long)a.C * Tc * a.K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % a.K);
        const long long rest = i / a.K;
        const int t = (int)(rest % Tc), c = (int)(rest / Tc);
        const long long dst = a.step_major ? ((((long long)(k >> 4) * Tc + t) * a.C + c) << 4) + (k & 15) : i;
        wt[a.cls_off[cls] + dst] = f2bf(w[(((long long)k * a.R + a.kh[cls][t]) * a.S + a.kw[cls][t]) * a.C + c]);
    }
}

// ---- M-contiguous form (weight gradient) on bf16 tensors ----------------------------------------------------------------------
//   gw[g][m][n] = alpha * sum_{k in group g} GY[k][m] * X(k, n),   k -> (img, oy, ox),  n -> (tap, c)
struct MsParams {
    const void* gy;      // bf16 [N][P][Q][Kc]
    const void* x;       // bf16 [N][H][W][C]
    int N, H, W, C, P, Q, Kc, T, Ndim, stride, group, n_groups;
    float alpha;
    void* gw;            // [n_groups][Kc][Ndim] fp32 (bf16 when out_bf16) or null
    int out_bf16;
    float* sq;           // [n_groups] or null
    int tiles_m, tiles_n, ksplit;
    const float* row_scale;   // igemm_mcs_tr_kernel<., true>: [N] fp32 weight of each sample's WHOLE gradient, applied in fp32 to the sample's
                              // accumulated product (clip-weighted sums: sum_b f_b g_b with every g_b exactly the bf16-MFMA gradient)
    signed char ty[IG_MAX_TAPS], tx[IG_MAX_TAPS];
};

constexpr int MS_BK = 64;        // pixels per LDS tile

// Q8: Q % 8 == 0, so the 8 consecutive pixels a thread gathers lie in one output row of one image (one decode per tile)
template <bool Q8>
__global__ __launch_bounds__(256, 2) void igemm_mcs_kernel(const MsParams p) {
    constexpr int ES = 128 + 1;                              // uint4 units per LDS entry (128 rows x 16 B + 16 B pad)
    __shared__ __attribute__((aligned(16))) uint4 As[2][8 * ES];
    __shared__ __attribute__((aligned(16))) uint4 Bs[2][8 * ES];
    __shared__ float s_red[4];

    const int tid = threadIdx.x;
    const int per_g = p.tiles_m * p.tiles_n;
    const int split = p.ksplit > 1 ? blockIdx.x % p.ksplit : 0;
    const int bid = p.ksplit > 1 ? blockIdx.x / p.ksplit : blockIdx.x;
    const int g = bid / per_g;
    const int tl = bid - g * per_g;
    const int tile_m = tl / p.tiles_n, tile_n = tl - tile_m * p.tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const int PQ = p.P * p.Q;
    const int Ktot = p.group * PQ;
    const long long pix_base = (long long)g * p.group * PQ;

    const bool is_a = tid < 128;
    const int lt = tid & 127;
    const int c8 = (lt & 15) * 8;        // first of this thread's 8 rows (m or n) within the tile: 16 lanes cover 256 contiguous bytes of a pixel
    const int kg = lt >> 4;              // its group of 8 pixels within the 64-pixel tile = its LDS entry
    // X columns are fixed per thread: 8 consecutive n share a tap (C % 8 == 0)
    const int nb = n0 + c8;
    const bool b_ok = nb < p.Ndim;
    const int b_t = b_ok ? nb / p.C : 0;
    const int b_c = nb - b_t * p.C;
    const int b_ty = p.ty[b_t], b_tx = p.tx[b_t];
    const bool a_ok = m0 + c8 < p.Kc;
    const unsigned short* __restrict__ gyh = reinterpret_cast<const unsigned short*>(p.gy);
    const unsigned short* __restrict__ xh = reinterpret_cast<const unsigned short*>(p.x);

    uint4 rv[8];
    auto load_tile = [&](int kt) {
        const int kk0 = kt * MS_BK + kg * 8;
        if (is_a) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (a_ok && kk0 + j < Ktot) v = *reinterpret_cast<const uint4*>(gyh + (pix_base + kk0 + j) * p.Kc + m0 + c8);
                rv[j] = v;
            }
        } else if (Q8) {
            const int il = kk0 / PQ;
            const int pix = kk0 - il * PQ;
            const int oy = pix / p.Q, ox0 = pix - oy * p.Q;
            const long long img = (long long)g * p.group + il;
            const int iy = oy * p.stride + b_ty;
            const bool row_ok = b_ok && kk0 < Ktot && iy >= 0 && iy < p.H;
            const unsigned short* src = xh + ((img * p.H + iy) * p.W) * p.C + b_c;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ix = (ox0 + j) * p.stride + b_tx;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (row_ok && ix >= 0 && ix < p.W) v = *reinterpret_cast<const uint4*>(src + (long long)ix * p.C);
                rv[j] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = kk0 + j;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (b_ok && kk < Ktot) {
                    const int il = kk / PQ;
                    const int pix = kk - il * PQ;
                    const int oy = pix / p.Q, ox = pix - oy * p.Q;
                    const long long img = (long long)g * p.group + il;
                    const int iy = oy * p.stride + b_ty, ix = ox * p.stride + b_tx;
                    if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) v = *reinterpret_cast<const uint4*>(xh + ((img * p.H + iy) * p.W + ix) * p.C + b_c);
                }
                rv[j] = v;
            }
        }
    };
    // rv[j] = channels c8..c8+7 of pixel j  ->  entry kg, row c8+ch: the 8 pixels of channel ch (an 8x8 transpose of 16-bit values).
    // Row R of an entry is stored at slot swz(R) = (R & ~7) | ((R + (R >> 3)) & 7): the 8 lanes that write the same channel of 8
    // consecutive row blocks (128 bytes apart) land on 8 different 16-byte bank groups, and a 32-row fragment read still covers
    // the same 512 contiguous bytes.  (Rotating the DATA per lane instead made rv[] dynamically indexed -> scratch.)
    auto store_tile = [&](int buf) {
        uint4* dst = (is_a ? As[buf] : Bs[buf]) + kg * ES + c8;
        const int rot = (c8 >> 3) & 7;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            unsigned w4[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const uint4 a = rv[2 * jj], b = rv[2 * jj + 1];
                const unsigned lo = (ch >> 1) == 0 ? a.x : ((ch >> 1) == 1 ? a.y : ((ch >> 1) == 2 ? a.z : a.w));
                const unsigned hi = (ch >> 1) == 0 ? b.x : ((ch >> 1) == 1 ? b.y : ((ch >> 1) == 2 ? b.z : b.w));
                w4[jj] = (ch & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
            }
            dst[(ch + rot) & 7] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
    };
    auto swz = [](int R) { return (R & ~7) | ((R + (R >> 3)) & 7); };

    const int lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    const int arow[2] = {swz(wm * 64 + r), swz(wm * 64 + r + 32)}, brow[2] = {swz(wn * 64 + r), swz(wn * 64 + r + 32)};

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    const int nk_all = (Ktot + MS_BK - 1) / MS_BK;
    int kt0 = 0, nk = nk_all;
    if (p.ksplit > 1) {
        const int per = (nk_all + p.ksplit - 1) / p.ksplit;
        kt0 = split * per;
        nk = kt0 + per < nk_all ? kt0 + per : nk_all;
        if (kt0 >= nk) return;      // uniform across the workgroup
    }
    load_tile(kt0);
    store_tile(0);
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = __builtin_bit_cast(bf16x8, As[buf][(2 * s + h) * ES + arow[i]]);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = __builtin_bit_cast(bf16x8, Bs[buf][(2 * s + h) * ES + brow[j]]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: scale, store, per-group sum of squares (as igemm_mc) ------------------------
    float ss = 0.f;
    float* __restrict__ outg = (p.gw && !p.out_bf16) ? reinterpret_cast<float*>(p.gw) + (long long)g * p.Kc * p.Ndim : nullptr;
    unsigned short* __restrict__ outh = (p.gw && p.out_bf16) ? reinterpret_cast<unsigned short*>(p.gw) + (long long)g * p.Kc * p.Ndim : nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + r;
        if (n >= p.Ndim) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + wm * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                if (m >= p.Kc) continue;
                float val = p.alpha * acc[i][j][v];
                if (p.out_bf16) {      // what is stored is what gets clipped: norm of the rounded value
                    const unsigned short u = f2bf(val);
                    if (outh) outh[(long long)m * p.Ndim + n] = u;
                    val = bf2f(u);
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

// Same contraction with NO register transpose: the [pixel][channel] tiles go to LDS as they are loaded (row-major, 256-byte rows,
// 16-byte chunks XOR-swizzled by the row so that both the row writes and the transposed reads are conflict-free) and the MFMA
// operands come back through ds_read_b64_tr_b16, gfx950's transposing LDS read: per 16-lane group a block of 4 rows (pixels) x 16
// columns (channels) is delivered column-major, i.e. each lane receives 4 consecutive k of ITS channel; two such reads are one
// operand of v_mfma_f32_32x32x16_bf16.  The register transpose above costs ~100 vector instructions per thread and K tile beside
// 16 MFMAs per wave (measured 176-208 TF); here a K tile is 8 global loads, 8 ds_write_b128 and 32 transposed reads per thread.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// SCALED (clip-weighted sums for ghost clipping): P*Q is a multiple of the 64-pixel K tile, so every K tile lies
// in ONE sample; when a sample's last tile has been multiplied its accumulated product is folded into a second accumulator with the
// sample's fp32 weight (acc_sum += f_b * acc; acc = 0).  The weight never touches a bfloat16 operand: the summed contribution of
// sample b is f_b times exactly the gradient the unscaled kernel materialises, so the clipped sum keeps the sensitivity bound to fp32
// rounding (scaling gy by f_b before the matrix core would round f_b * gy to bfloat16: a 2^-8 relative slack on that bound).
template <bool Q8, bool SCALED = false>
__global__ __launch_bounds__(256, 2) void igemm_mcs_tr_kernel(const MsParams p) {
    constexpr int OPB = 64 * 256;                           // bytes of one operand tile: 64 pixels x 128 channels x 2 B
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * OPB];     // [buffer][operand]
    __shared__ float s_red[4];

    const int tid = threadIdx.x;
    const int per_g = p.tiles_m * p.tiles_n;
    const int split = p.ksplit > 1 ? blockIdx.x % p.ksplit : 0;
    const int bid = p.ksplit > 1 ? blockIdx.x / p.ksplit : blockIdx.x;
    const int g = bid / per_g;
    const int tl = bid - g * per_g;
    const int tile_m = tl / p.tiles_n, tile_n = tl - tile_m * p.tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const int PQ = p.P * p.Q;
    const int Ktot = p.group * PQ;
    const long long pix_base = (long long)g * p.group * PQ;

    const bool is_a = tid < 128;
    const int lt = tid & 127;
    const int c16 = lt & 15;             // this thread's 16-byte chunk (8 channels) of a pixel row: 16 lanes cover its 256 bytes
    const int c8 = c16 * 8;
    const int kg = lt >> 4;              // its group of 8 pixels within the 64-pixel tile
    const int nb = n0 + c8;
    const bool b_ok = nb < p.Ndim;
    const int b_t = b_ok ? nb / p.C : 0;
    const int b_c = nb - b_t * p.C;
    const int b_ty = p.ty[b_t], b_tx = p.tx[b_t];
    const bool a_ok = m0 + c8 < p.Kc;
    // buffer loads with branch-free offsets (an invalid element ORs 0xFFFFFFF0 into its offset -> the range check returns zeros): see
    // igemm_kcs_kernel — as `if (valid) v = *ptr` every load sat in its own exec-masked block
    const __amdgpu_buffer_rsrc_t gy_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.gy), 0, (unsigned)(2ll * p.N * PQ * p.Kc), 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (unsigned)(2ll * p.N * p.H * p.W * p.C), 0x00020000);
    auto bld = [](__amdgpu_buffer_rsrc_t r, unsigned off) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
        return make_uint4(v[0], v[1], v[2], v[3]);
    };

    // two register sets: tile t+2 is loaded under the MFMAs of tile t and written to LDS after those of tile t+1 (see igemm_kcs_kernel)
    uint4 rv0[8], rv1[8];
    int k_lim = Ktot;            // pixels this workgroup may read (its K range when the reduction is split): past it every load is zero
    auto load_tile = [&](int kt, uint4 (&rv)[8]) {
        const int kk0 = kt * MS_BK + kg * 8;
        if (is_a) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                rv[j] = bld(gy_rsrc, (2u * (((unsigned)pix_base + (unsigned)(kk0 + j)) * (unsigned)p.Kc + (unsigned)(m0 + c8))) | ((a_ok && kk0 + j < k_lim) ? 0u : S_OOB16));
        } else if (Q8) {
            const int il = kk0 / PQ;
            const int pix = kk0 - il * PQ;
            const int oy = pix / p.Q, ox0 = pix - oy * p.Q;
            const int img = g * p.group + il;           // (32-bit: the entry checks that both operands are below 4 GB)
            const int iy = oy * p.stride + b_ty;
            const unsigned row_bad = (b_ok && kk0 < k_lim && iy >= 0 && iy < p.H) ? 0u : S_OOB16;
            const unsigned src = 2u * (unsigned)(((img * p.H + iy) * p.W) * p.C + b_c);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ix = (ox0 + j) * p.stride + b_tx;
                rv[j] = bld(x_rsrc, (src + 2u * (unsigned)(ix * p.C)) | row_bad | ((unsigned)ix < (unsigned)p.W ? 0u : S_OOB16));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = kk0 + j;
                const int kc2 = kk < k_lim ? kk : 0;
                const int il = kc2 / PQ;
                const int pix = kc2 - il * PQ;
                const int oy = pix / p.Q, ox = pix - oy * p.Q;
                const int img = g * p.group + il;
                const int iy = oy * p.stride + b_ty, ix = ox * p.stride + b_tx;
                const bool ok = b_ok && kk < k_lim && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                rv[j] = bld(x_rsrc, (2u * (unsigned)(((img * p.H + iy) * p.W + ix) * p.C + b_c)) | (ok ? 0u : S_OOB16));
            }
        }
    };
    // row (pixel) kr of a tile, 16-byte chunk ch: byte offset 256*kr + 16*(ch ^ (((kr & 3) << 2) | ((kr >> 2) & 3)))
    auto store_tile = [&](int buf, const uint4 (&rv)[8]) {
        unsigned char* dst = smem + (buf * 2 + (is_a ? 0 : 1)) * OPB;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kr = kg * 8 + j;
            const int sw = ((j & 3) << 2) | ((kr >> 2) & 3);
            *reinterpret_cast<uint4*>(dst + 256 * kr + 16 * (c16 ^ sw)) = rv[j];
        }
    };

    const int lane = tid & 63, wid = tid >> 6;
    const int h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    // transposed-read addresses (bytes within an operand tile) of this lane for k-step 0: [4-row half u][32-row tile i]
    const int gq = (lane >> 2) & 3, gp = lane & 3, ghalf = (lane >> 4) & 1;       // row q and column quad p within the 16-lane group; which 16 of the 32 rows
    int a_tr[2][2], b_tr[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kr = 8 * h + 4 * u + gq;
            const int sw = (gq << 2) | ((2 * h + u) & 3);
            const int cha = (wm * 64 + i * 32) / 8 + 2 * ghalf + (gp >> 1), chb = (wn * 64 + i * 32) / 8 + 2 * ghalf + (gp >> 1);
            a_tr[u][i] = 256 * kr + 16 * (cha ^ sw) + 8 * (gp & 1);
            b_tr[u][i] = 256 * kr + 16 * (chb ^ sw) + 8 * (gp & 1);
        }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    const int nk_all = (Ktot + MS_BK - 1) / MS_BK;
    int kt0 = 0, nk = nk_all;
    if (p.ksplit > 1) {
        const int per = (nk_all + p.ksplit - 1) / p.ksplit;
        kt0 = split * per;
        nk = kt0 + per < nk_all ? kt0 + per : nk_all;
        if (kt0 >= nk) return;      // uniform across the workgroup
    }
    k_lim = nk * MS_BK < Ktot ? nk * MS_BK : Ktot;
    auto mma_tile = [&](int buf) {
        const unsigned char* As = smem + (buf * 2) * OPB;
        const unsigned char* Bs = As + OPB;
#pragma unroll
        for (int s = 0; s < 4; ++s) {       // 16 pixels per step: rows 16 s .. 16 s + 15 = +4096 bytes per step
            bf16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(As + a_tr[0][i] + 4096 * s));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(As + a_tr[1][i] + 4096 * s));
                const uint4 w = make_uint4(__builtin_bit_cast(uint2, lo).x, __builtin_bit_cast(uint2, lo).y, __builtin_bit_cast(uint2, hi).x, __builtin_bit_cast(uint2, hi).y);
                af[i] = __builtin_bit_cast(bf16x8, w);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Bs + b_tr[0][j] + 4096 * s));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Bs + b_tr[1][j] + 4096 * s));
                const uint4 w = make_uint4(__builtin_bit_cast(uint2, lo).x, __builtin_bit_cast(uint2, lo).y, __builtin_bit_cast(uint2, hi).x, __builtin_bit_cast(uint2, hi).y);
                bf[j] = __builtin_bit_cast(bf16x8, w);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };
    if (SCALED) {
        // one register set (the second accumulator needs the registers); a sample = tiles_per_sample consecutive K tiles
        f32x16 acc_sum[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc_sum[i][j][v] = 0.f;
        const int tps = PQ / MS_BK;
        load_tile(kt0, rv0);
        store_tile(0, rv0);
        __syncthreads();
        int in_sample = 0;
        for (int kt = kt0; kt < nk; ++kt) {
            const int buf = (kt - kt0) & 1;
            if (kt + 1 < nk) load_tile(kt + 1, rv0);
            mma_tile(buf);
            if (++in_sample == tps) {            // uniform: the sample's product is complete
                in_sample = 0;
                const float f = p.row_scale[(long long)g * p.group + kt / tps];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int v = 0; v < 16; ++v) { acc_sum[i][j][v] = fmaf(f, acc[i][j][v], acc_sum[i][j][v]); acc[i][j][v] = 0.f; }
            }
            if (kt + 1 < nk) store_tile(buf ^ 1, rv0);
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = acc_sum[i][j];
    } else {
    load_tile(kt0, rv0);
    load_tile(kt0 + 1, rv1);         // past the group's pixels every lane loads zeros (kk >= Ktot)
    store_tile(0, rv0);
    __syncthreads();
    for (int kt = kt0; kt < nk; kt += 2) {      // one loop exit (see igemm_kcs_kernel): an odd tile count multiplies one all-zero tile
        load_tile(kt + 2, rv0);
        mma_tile(0);
        store_tile(1, rv1);          // tile kt + 1 (zeros past k_lim)
        __syncthreads();
        load_tile(kt + 3, rv1);
        mma_tile(1);
        store_tile(0, rv0);          // tile kt + 2
        __syncthreads();
    }
    }

    // ---- epilogue: scale, store, per-group sum of squares (as igemm_mc) ------------------------
    const int r = lane & 31;
    float ss = 0.f;
    float* __restrict__ outg = (p.gw && !p.out_bf16) ? reinterpret_cast<float*>(p.gw) + (long long)g * p.Kc * p.Ndim : nullptr;
    unsigned short* __restrict__ outh = (p.gw && p.out_bf16) ? reinterpret_cast<unsigned short*>(p.gw) + (long long)g * p.Kc * p.Ndim : nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + r;
        if (n >= p.Ndim) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + wm * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                if (m >= p.Kc) continue;
                float val = p.alpha * acc[i][j][v];
                if (p.out_bf16) {
                    const unsigned short u = f2bf(val);
                    if (outh) outh[(long long)m * p.Ndim + n] = u;
                    val = bf2f(u);
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

int sqnorm_rows_accumulate(const float* in, long long n_rows, long long len, float* sq_accum, hipStream_t st);   // clip_kernels.hip

// ---- pointwise kernels on bf16 tensors -----------------------------------------------------------------------------------------
// out = g * (y > 0 ? 1 : slope): LeakyReLU / ReLU backward from the OUTPUT's sign, 8 values per lane
__global__ __launch_bounds__(256) void act_bwd_bf16_kernel(const uint4* __restrict__ g, const uint4* __restrict__ y, long long n8, float slope,
                                                           uint4* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const uint4 gv = g[i], yv = y[i];
        const unsigned gs[4] = {gv.x, gv.y, gv.z, gv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
        unsigned o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float g0 = __uint_as_float(gs[q] << 16), g1 = __uint_as_float(gs[q] & 0xffff0000u);
            const float y0 = __uint_as_float(ys[q] << 16), y1 = __uint_as_float(ys[q] & 0xffff0000u);
            o[q] = f2bf_pk(y0 > 0.f ? g0 : slope * g0, y1 > 0.f ? g1 : slope * g1);
        }
        out[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// gb[g][k] = alpha * sum over the group's pixels of gy[., k] (bf16 gy, fp32 sums); K % 8 == 0, K <= 2048, 256 % (K/8) == 0
__global__ __launch_bounds__(256) void bias_grad_bf16_kernel(const unsigned short* __restrict__ gy, int PQ, int K, int group, float alpha,
                                                             float* __restrict__ gb, float* __restrict__ sq) {
    extern __shared__ float s_part[];            // [K]
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int c8n = K >> 3, c8 = tid % c8n, sl = tid / c8n, nsl = 256 / c8n;
    const int g = blockIdx.x;
    const long long npix = (long long)group * PQ;
    const unsigned short* base = gy + (long long)g * npix * K + 8 * c8;
    for (int i = tid; i < K; i += 256) s_part[i] = 0.f;
    __syncthreads();
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    long long px = sl;
    for (; px + 3 * nsl < npix; px += 4 * nsl) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(base + (px + u * nsl) * K);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned d[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc[2 * q] += __uint_as_float(d[q] << 16); acc[2 * q + 1] += __uint_as_float(d[q] & 0xffff0000u); }
        }
    }
    for (; px < npix; px += nsl) {
        const uint4 v = *reinterpret_cast<const uint4*>(base + px * K);
        const unsigned d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc[2 * q] += __uint_as_float(d[q] << 16); acc[2 * q + 1] += __uint_as_float(d[q] & 0xffff0000u); }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) atomicAdd(&s_part[8 * c8 + q], acc[q]);
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

// ---- the critic's head on bf16 features (nn.Linear(C, 1), DCResNet_models.py:145) as streams -------------------------------------
// y[n] = act(<x[n,:], bf16(w)> + b): one workgroup per row (as linear_k1.hip, x bfloat16)
__global__ __launch_bounds__(256) void linear_k1s_fwd_kernel(const unsigned short* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                             long long C, int act, float* __restrict__ y) {
    __shared__ float s_red[4];
    const long long n = blockIdx.x;
    const uint4* xr = reinterpret_cast<const uint4*>(x + n * C);
    float acc = 0.f;
    for (long long i = threadIdx.x; i < (C >> 3); i += 256) {
        const uint4 a = xr[i];
        const float4 b0 = reinterpret_cast<const float4*>(w)[2 * i], b1 = reinterpret_cast<const float4*>(w)[2 * i + 1];
        acc = fmaf(__uint_as_float(a.x << 16), bf2f(f2bf(b0.x)), acc); acc = fmaf(__uint_as_float(a.x & 0xffff0000u), bf2f(f2bf(b0.y)), acc);
        acc = fmaf(__uint_as_float(a.y << 16), bf2f(f2bf(b0.z)), acc); acc = fmaf(__uint_as_float(a.y & 0xffff0000u), bf2f(f2bf(b0.w)), acc);
        acc = fmaf(__uint_as_float(a.z << 16), bf2f(f2bf(b1.x)), acc); acc = fmaf(__uint_as_float(a.z & 0xffff0000u), bf2f(f2bf(b1.y)), acc);
        acc = fmaf(__uint_as_float(a.w << 16), bf2f(f2bf(b1.z)), acc); acc = fmaf(__uint_as_float(a.w & 0xffff0000u), bf2f(f2bf(b1.w)), acc);
    }
    const float tot = block_sum_256(acc, s_red);
    if (threadIdx.x == 0) {
        float val = tot + (bias ? bias[0] : 0.f);
        if (act == CSLGAN_ACT_LRELU02) val = val > 0.f ? val : 0.2f * val;
        else if (act == CSLGAN_ACT_RELU) val = val > 0.f ? val : 0.f;
        else if (act == CSLGAN_ACT_TANH) val = tanhf(val);
        y[n] = val;
    }
}

// gx[n,:] = bf16( gy[n] * bf16(w) (* lrelu'(mask[n,:])) ), 8 channels per lane
__global__ __launch_bounds__(256) void linear_k1s_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w, const unsigned short* __restrict__ mask,
                                                               long long C8, unsigned short* __restrict__ gx) {
    const long long n = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= C8) return;
    const float g = gy[n];
    const float4 b0 = reinterpret_cast<const float4*>(w)[2 * i], b1 = reinterpret_cast<const float4*>(w)[2 * i + 1];
    float o[8] = {g * bf2f(f2bf(b0.x)), g * bf2f(f2bf(b0.y)), g * bf2f(f2bf(b0.z)), g * bf2f(f2bf(b0.w)),
                  g * bf2f(f2bf(b1.x)), g * bf2f(f2bf(b1.y)), g * bf2f(f2bf(b1.z)), g * bf2f(f2bf(b1.w))};
    if (mask) {
        const uint4 m = reinterpret_cast<const uint4*>(mask + n * C8 * 8)[i];
        const unsigned md[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            o[2 * q] *= __uint_as_float(md[q] << 16) > 0.f ? 1.f : 0.2f;
            o[2 * q + 1] *= __uint_as_float(md[q] & 0xffff0000u) > 0.f ? 1.f : 0.2f;
        }
    }
    reinterpret_cast<uint4*>(gx + n * C8 * 8)[i] = make_uint4(f2bf_pk(o[0], o[1]), f2bf_pk(o[2], o[3]), f2bf_pk(o[4], o[5]), f2bf_pk(o[6], o[7]));
}

// gw[g,:] = alpha * sum_{n in group g} gy[n] * x[n,:] (fp32), sq[g] += ||gw[g,:]||^2: the head's per-sample / grouped weight gradient
// is a scaled copy (a short weighted sum) of bf16 feature rows
__global__ __launch_bounds__(256) void linear_k1s_wgrad_kernel(const float* __restrict__ gy, const unsigned short* __restrict__ x, long long C8, int group,
                                                               float alpha, float* __restrict__ gw, float* __restrict__ sq, const float* __restrict__ row_scale) {
    __shared__ float s_red[4];
    const long long g = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    if (i < C8) {
        for (int r = 0; r < group; ++r) {
            const long long n = g * group + r;
            const float s = row_scale ? gy[n] * row_scale[n] : gy[n];        // clip-weighted sums: the weight meets the fp32 cotangent
            const uint4 a = reinterpret_cast<const uint4*>(x + n * C8 * 8)[i];
            const unsigned d[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[2 * q] = fmaf(s, __uint_as_float(d[q] << 16), acc[2 * q]);
                acc[2 * q + 1] = fmaf(s, __uint_as_float(d[q] & 0xffff0000u), acc[2 * q + 1]);
            }
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { acc[q] *= alpha; ss = fmaf(acc[q], acc[q], ss); }
    if (gw && i < C8) {
        float4* o = reinterpret_cast<float4*>(gw + g * C8 * 8) + 2 * i;
        o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
    if (sq) {
        const float tot = block_sum_256(ss, s_red);
        if (threadIdx.x == 0) atomicAdd(sq + g, tot);
    }
}

}  // namespace cslgan

using namespace cslgan;

static int check_conv_s(const cslgan_conv_t* c, const char* who) {
    CSLGAN_REQUIRE(c->N > 0 && c->H > 0 && c->W > 0 && c->C > 0 && c->K > 0 && c->R > 0 && c->S > 0 && c->stride > 0 && c->pad >= 0,
                   "%s: non-positive dimension", who);
    CSLGAN_REQUIRE(c->R * c->S <= IG_MAX_TAPS, "%s: %dx%d filter has more than %d taps", who, c->R, c->S, IG_MAX_TAPS);
    const int P = (c->H + 2 * c->pad - c->R) / c->stride + 1, Q = (c->W + 2 * c->pad - c->S) / c->stride + 1;
    CSLGAN_REQUIRE(P == c->P && Q == c->Q, "%s: output %dx%d does not match P,Q=%d,%d", who, P, Q, c->P, c->Q);
    CSLGAN_REQUIRE((long long)c->N * c->P * c->Q * c->K < (1ll << 31) && (long long)c->N * c->H * c->W * c->C < (1ll << 31) &&
                   (long long)c->K * c->R * c->S * c->C < (1ll << 31), "%s: tensor too large for 32-bit offsets", who);
    return CSLGAN_OK;
}

extern "C" {

int cslgan_cast_f32_bf16(const float* in, void* out, int64_t n, void* stream) {
    CSLGAN_REQUIRE(in && out && n >= 0, "cast_f32_bf16: bad argument");
    CSLGAN_REQUIRE(aligned16(in) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0, "cast_f32_bf16: misaligned");
    if (n == 0) return CSLGAN_OK;
    hipLaunchKernelGGL(round_bf16_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, in, reinterpret_cast<unsigned short*>(out), (long long)n);
    return check_launch("round_bf16_kernel");
}

int cslgan_cast_bf16_f32(const void* in, float* out, int64_t n, void* stream) {
    CSLGAN_REQUIRE(in && out && n >= 0, "cast_bf16_f32: bad argument");
    CSLGAN_REQUIRE(aligned16(out) && (reinterpret_cast<uintptr_t>(in) & 7u) == 0, "cast_bf16_f32: misaligned");
    if (n == 0) return CSLGAN_OK;
    hipLaunchKernelGGL(widen_bf16_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const unsigned short*>(in), out, (long long)n);
    return check_launch("widen_bf16_kernel");
}

// y = act(conv(x, w) + bias [+ residual]) with x bf16 [N,H,W,C] (C % 8 == 0), the filter read from wb_ws = bf16(w) in KRSC order
// (written here when repack != 0), y bf16 or fp32.
int cslgan_conv2d_fwd_bf16s(const cslgan_conv_t* c, const void* x, const float* w, void* wb_ws, int repack, const float* bias,
                            const void* residual, int res_bf16, int act, void* y, int y_bf16, void* stream) {
    CSLGAN_REQUIRE(c && x && w && wb_ws && y, "conv2d_fwd_bf16s: null argument");
    int rc = check_conv_s(c, "conv2d_fwd_bf16s");
    if (rc) return rc;
    CSLGAN_REQUIRE(act >= 0 && act <= 3, "conv2d_fwd_bf16s: unknown activation %d", act);
    CSLGAN_REQUIRE(c->C % 8 == 0, "conv2d_fwd_bf16s: C=%d is not a multiple of 8", c->C);
    hipStream_t st = (hipStream_t)stream;
    if (c->K == 1 && c->H == 1 && c->W == 1 && c->R == 1 && c->S == 1 && c->stride == 1 && c->pad == 0 && !residual && !y_bf16 && c->N <= 65535 &&
        aligned16(x) && aligned16(w)) {       // the critic's head: a dot product per row
        note_kernel("linear_k1s_fwd_kernel");
        hipLaunchKernelGGL(linear_k1s_fwd_kernel, dim3((unsigned)c->N), dim3(256), 0, st, reinterpret_cast<const unsigned short*>(x), w, bias,
                           (long long)c->C, act, reinterpret_cast<float*>(y));
        return check_launch("linear_k1s_fwd_kernel");
    }
    const long long wn = (long long)c->K * c->R * c->S * c->C;
    if (conv1x1s_eligible(c, residual) && aligned16(x) && aligned16(w) && aligned16(wb_ws)) {      // the generator's shortcut convs: a stream over pixels
        if (repack) {
            hipLaunchKernelGGL(round_bf16_kernel, dim3(stream_blocks(wn / 4)), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(wb_ws), wn);
            rc = check_launch("round_bf16_kernel");
            if (rc) return rc;
        }
        if (c->C == 16) return launch_conv1x1s_ck<1>(c, x, wb_ws, bias, act, y, y_bf16, st);
        if (c->C == 32) return launch_conv1x1s_ck<2>(c, x, wb_ws, bias, act, y, y_bf16, st);
        return launch_conv1x1s_ck<4>(c, x, wb_ws, bias, act, y, y_bf16, st);
    }
    KsParams p{};
    p.a = x; p.AH = c->H; p.AW = c->W; p.AC = c->C; p.VH = c->H; p.VW = c->W; p.sy = p.sx = c->stride;
    p.w = wb_ws; p.Nn = c->K; p.out = y; p.OHf = c->P; p.OWf = c->Q; p.osy = p.osx = 1; p.ldo = c->K; p.dense_out = 1;
    p.bias = bias; p.res = residual; p.res_bf16 = res_bf16; p.mask = nullptr; p.mask_bf16 = 0; p.act = act;
    p.n_cls = 1;
    KcClass& k = p.cls[0];
    k.M = c->N * c->P * c->Q; k.OHc = c->P; k.OWc = c->Q; k.T = c->R * c->S; k.Kdim = k.T * c->C; k.w_off = 0; k.oy0 = k.ox0 = 0;
    for (int t = 0; t < IG_MAX_TAPS; ++t) { k.ty[t] = 0; k.tx[t] = 0; }
    for (int kh = 0; kh < c->R; ++kh)
        for (int kw = 0; kw < c->S; ++kw) { k.ty[kh * c->S + kw] = (signed char)(kh - c->pad); k.tx[kh * c->S + kw] = (signed char)(kw - c->pad); }
    // the layout of the bf16 filter copy follows the kernel, and the kernel follows the SHAPE alone (a layer always takes the same
    // route, so a cached copy is always in the layout its reader expects): step-major for the LDS-halo form, plain KRSC otherwise
    const bool halo = halos_eligible(p) && aligned16(w) && aligned16(wb_ws) && wn % 4 == 0;
    if (repack) {
        CSLGAN_REQUIRE(aligned16(w) && aligned16(wb_ws), "conv2d_fwd_bf16s: filter must be 16-byte aligned");
        if (halo) {
            rc = split_filter_x3(w, c->K, c->R * c->S, c->C, wb_ws, st, 1);
        } else {
            hipLaunchKernelGGL(round_bf16_kernel, dim3(stream_blocks(wn / 4)), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(wb_ws), wn);
            rc = check_launch("round_bf16_kernel");
        }
        if (rc) return rc;
    }
    if (halo) {
        p.w3 = wb_ws;
        return launch_halos(p, y_bf16 != 0, st);
    }
    return launch_kcs(p, y_bf16 != 0, st);
}

// gx = conv_transpose(gy, w) (* lrelu'(mask)) with gy bf16 [N,P,Q,K] (K % 8 == 0), the per-parity-class filter matrices read from
// wt_ws (bf16, K*R*S*C elements, written here when repack != 0), gx bf16 or fp32, mask (nullable) of gx's shape and element type.
int cslgan_conv2d_dgrad_bf16s(const cslgan_conv_t* c, const void* gy, const float* w, void* wt_ws, int repack, const void* mask,
                              void* gx, int gx_bf16, void* stream) {
    CSLGAN_REQUIRE(c && gy && w && wt_ws && gx, "conv2d_dgrad_bf16s: null argument");
    int rc = check_conv_s(c, "conv2d_dgrad_bf16s");
    if (rc) return rc;
    CSLGAN_REQUIRE(c->stride >= 1 && c->stride <= 2, "conv2d_dgrad_bf16s: stride %d unsupported", c->stride);
    CSLGAN_REQUIRE(c->K % 8 == 0, "conv2d_dgrad_bf16s: K=%d is not a multiple of 8", c->K);
    const int s = c->stride;
    hipStream_t st = (hipStream_t)stream;
    DgradRepack ra{};
    ra.K = c->K; ra.R = c->R; ra.S = c->S; ra.C = c->C;
    KsParams p{};
    p.a = gy; p.AH = c->P; p.AW = c->Q; p.AC = c->K; p.VH = c->P; p.VW = c->Q; p.sy = p.sx = 1;
    p.w = wt_ws; p.Nn = c->C; p.out = gx; p.OHf = c->H; p.OWf = c->W; p.osy = p.osx = s; p.ldo = c->C;
    p.dense_out = (s == 1) ? 1 : 0;
    p.bias = nullptr; p.res = nullptr; p.mask = mask; p.mask_bf16 = gx_bf16; p.act = CSLGAN_ACT_NONE;
    int off = 0, ncls = 0;
    for (int py = 0; py < s; ++py)
        for (int px = 0; px < s; ++px) {
            const int OHc = (c->H - py + s - 1) / s, OWc = (c->W - px + s - 1) / s;
            if (OHc <= 0 || OWc <= 0) continue;
            const int cls = ncls++;
            KcClass& k = p.cls[cls];
            for (int t = 0; t < IG_MAX_TAPS; ++t) { k.ty[t] = 0; k.tx[t] = 0; }
            int T = 0;
            for (int kh = 0; kh < c->R; ++kh) {
                if (((py + c->pad - kh) % s + s) % s != 0) continue;
                for (int kw = 0; kw < c->S; ++kw) {
                    if (((px + c->pad - kw) % s + s) % s != 0) continue;
                    ra.kh[cls][T] = (signed char)kh; ra.kw[cls][T] = (signed char)kw;
                    k.ty[T] = (signed char)((py + c->pad - kh) / s);
                    k.tx[T] = (signed char)((px + c->pad - kw) / s);
                    ++T;
                }
            }
            CSLGAN_REQUIRE(T > 0, "conv2d_dgrad_bf16s: a parity class has no taps (filter smaller than stride)");
            k.M = c->N * OHc * OWc; k.OHc = OHc; k.OWc = OWc; k.T = T; k.Kdim = T * c->K; k.w_off = off; k.oy0 = py; k.ox0 = px;
            ra.cls_T[cls] = T; ra.cls_off[cls] = off; off += T * c->K * c->C;
        }
    p.n_cls = ncls; ra.n_cls = ncls;
    // the LDS-halo form reads step-major class matrices; like the forward entry, the route (and with it the layout of the cached bf16
    // copy) follows the SHAPE alone
    // Measured (128x128, same run, gather vs halo): conv2's data gradient into 64 channels 0.652 vs 0.682 ms at 384 rows (247 / 236 TF),
    // conv3's into 128 channels 0.386 vs 0.473 ms (417 / 340 TF): with 4-9 taps per class the halo's reuse does not pay for its
    // one-MFMA-tile-per-wave steps, so the data gradient stays on the gather form (CSLGAN_HALOS_DGRAD=1 routes it to the halo form).
    static const int halo_dgrad_env = [] { const char* e = getenv("CSLGAN_HALOS_DGRAD"); return e ? atoi(e) : 0; }();
    const bool halo = halo_dgrad_env && c->K % 16 == 0 && halos_eligible(p) && aligned16(wt_ws);
    ra.step_major = halo ? 1 : 0;
    if (repack) {
        unsigned gxn = (unsigned)(((long long)c->K * c->C * c->R * c->S / (s * s) + 255) / 256);
        gxn = gxn > 1024 ? 1024 : (gxn < 1 ? 1 : gxn);
        hipLaunchKernelGGL(repack_dgrad_bf16_kernel, dim3(gxn, (unsigned)ncls), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(wt_ws), ra);
        rc = check_launch("repack_dgrad_bf16_kernel");
        if (rc) return rc;
    }
    if (halo) {
        p.w3 = wt_ws;
        return launch_halos(p, gx_bf16 != 0, st);
    }
    return launch_kcs(p, gx_bf16 != 0, st);
}

// gw[N/group][K][R][S][C] (fp32, or bf16 when gw_bf16) and / or sq[N/group] += ||alpha * gw_g||^2 from bf16 gy [N,P,Q,K] and
// bf16 x [N,H,W,C] (K % 8 == 0, C % 8 == 0).
int cslgan_conv2d_wgrad_grouped_bf16s(const cslgan_conv_t* c, const void* gy, const void* x, int group, float alpha, void* gw, int gw_bf16,
                                      float* sq, void* stream) {
    CSLGAN_REQUIRE(c && gy && x, "conv2d_wgrad_bf16s: null argument");
    CSLGAN_REQUIRE(gw || sq, "conv2d_wgrad_bf16s: neither gw nor sq requested");
    int rc = check_conv_s(c, "conv2d_wgrad_bf16s");
    if (rc) return rc;
    CSLGAN_REQUIRE(group >= 1 && c->N % group == 0, "conv2d_wgrad_bf16s: N=%d not divisible by group=%d", c->N, group);
    CSLGAN_REQUIRE(c->K % 8 == 0 && c->C % 8 == 0 && aligned16(gy) && aligned16(x), "conv2d_wgrad_bf16s: K and C must be multiples of 8, operands 16-byte aligned");
    CSLGAN_REQUIRE(2ll * c->N * c->P * c->Q * c->K < 0xFFFFFFF0ll && 2ll * c->N * c->H * c->W * c->C < 0xFFFFFFF0ll, "conv2d_wgrad_bf16s: operand larger than 4 GB");
    hipStream_t st = (hipStream_t)stream;
    MsParams p{};
    p.gy = gy; p.x = x; p.N = c->N; p.H = c->H; p.W = c->W; p.C = c->C; p.P = c->P; p.Q = c->Q; p.Kc = c->K;
    p.T = c->R * c->S; p.Ndim = p.T * c->C; p.stride = c->stride; p.group = group; p.n_groups = c->N / group;
    p.alpha = alpha; p.gw = gw; p.sq = sq; p.out_bf16 = gw_bf16;
    for (int t = 0; t < IG_MAX_TAPS; ++t) { p.ty[t] = 0; p.tx[t] = 0; }
    for (int kh = 0; kh < c->R; ++kh)
        for (int kw = 0; kw < c->S; ++kw) { p.ty[kh * c->S + kw] = (signed char)(kh - c->pad); p.tx[kh * c->S + kw] = (signed char)(kw - c->pad); }
    p.tiles_m = (p.Kc + 127) / 128;
    p.tiles_n = (p.Ndim + 127) / 128;
    p.ksplit = 1;
    {
        const long long base = (long long)p.n_groups * p.tiles_m * p.tiles_n;
        const int nk_all = (p.group * p.P * p.Q + MS_BK - 1) / MS_BK;
        if (p.gw && !p.out_bf16 && base < 192 && nk_all >= 8) {
            const long long want = (512 + base - 1) / base, cap = nk_all / 2;
            p.ksplit = (int)(want < cap ? want : cap);
            if (p.ksplit < 1) p.ksplit = 1;
        }
    }
    if (p.ksplit > 1) {
        rc = zero_floats(reinterpret_cast<float*>(p.gw), (size_t)p.n_groups * p.Kc * p.Ndim, st);
        if (rc) return rc;
    }
    const long long nb = (long long)p.n_groups * p.tiles_m * p.tiles_n * p.ksplit;
    CSLGAN_REQUIRE(nb <= 0x7fffffffll, "conv2d_wgrad_bf16s: grid too large");
    static const int tr_env = [] { const char* e = getenv("CSLGAN_MCS_TR"); return e ? atoi(e) : 1; }();
    if (tr_env) {
        note_kernel("igemm_mcs_tr_kernel<128,128>");
        if (c->Q % 8 == 0) hipLaunchKernelGGL(igemm_mcs_tr_kernel<true>, dim3((unsigned)nb), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(igemm_mcs_tr_kernel<false>, dim3((unsigned)nb), dim3(256), 0, st, p);
    } else {
        note_kernel("igemm_mcs_kernel<128,128>");
        if (c->Q % 8 == 0) hipLaunchKernelGGL(igemm_mcs_kernel<true>, dim3((unsigned)nb), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(igemm_mcs_kernel<false>, dim3((unsigned)nb), dim3(256), 0, st, p);
    }
    rc = check_launch("igemm_mcs_kernel");
    if (rc) return rc;
    if (p.ksplit > 1 && p.sq) rc = sqnorm_rows_accumulate(reinterpret_cast<float*>(p.gw), p.n_groups, (long long)p.Kc * p.Ndim, p.sq, st);
    return rc;
}

// gw[N/group][K][R][S][C] (fp32) = alpha * sum_{n in group} row_scale[n] * (per-sample weight gradient of sample n), from bf16 gy and
// bf16 x: clip() + accumulate for ghost-clipped layers (train.py:399-402) in the bf16 storage mode.  row_scale [N] fp32 is applied in
// fp32 to each sample's accumulated product (igemm_mcs_tr_kernel<., true>), never to a bfloat16 operand.  Needs P*Q % 64 == 0.
int cslgan_conv2d_wgrad_scaled_bf16s(const cslgan_conv_t* c, const void* gy, const void* x, const float* row_scale, int group, float alpha,
                                     float* gw, void* stream) {
    CSLGAN_REQUIRE(c && gy && x && row_scale && gw, "conv2d_wgrad_scaled_bf16s: null argument");
    int rc = check_conv_s(c, "conv2d_wgrad_scaled_bf16s");
    if (rc) return rc;
    CSLGAN_REQUIRE(group >= 1 && c->N % group == 0, "conv2d_wgrad_scaled_bf16s: N=%d not divisible by group=%d", c->N, group);
    CSLGAN_REQUIRE(c->K % 8 == 0 && c->C % 8 == 0 && aligned16(gy) && aligned16(x), "conv2d_wgrad_scaled_bf16s: K and C must be multiples of 8, operands 16-byte aligned");
    CSLGAN_REQUIRE(2ll * c->N * c->P * c->Q * c->K < 0xFFFFFFF0ll && 2ll * c->N * c->H * c->W * c->C < 0xFFFFFFF0ll, "conv2d_wgrad_scaled_bf16s: operand larger than 4 GB");
    CSLGAN_REQUIRE((c->P * c->Q) % MS_BK == 0, "conv2d_wgrad_scaled_bf16s: P*Q=%d is not a multiple of %d (a K tile must lie in one sample)", c->P * c->Q, MS_BK);
    hipStream_t st = (hipStream_t)stream;
    MsParams p{};
    p.gy = gy; p.x = x; p.N = c->N; p.H = c->H; p.W = c->W; p.C = c->C; p.P = c->P; p.Q = c->Q; p.Kc = c->K;
    p.T = c->R * c->S; p.Ndim = p.T * c->C; p.stride = c->stride; p.group = group; p.n_groups = c->N / group;
    p.alpha = alpha; p.gw = gw; p.sq = nullptr; p.out_bf16 = 0; p.row_scale = row_scale; p.ksplit = 1;
    for (int t = 0; t < IG_MAX_TAPS; ++t) { p.ty[t] = 0; p.tx[t] = 0; }
    for (int kh = 0; kh < c->R; ++kh)
        for (int kw = 0; kw < c->S; ++kw) { p.ty[kh * c->S + kw] = (signed char)(kh - c->pad); p.tx[kh * c->S + kw] = (signed char)(kw - c->pad); }
    p.tiles_m = (p.Kc + 127) / 128;
    p.tiles_n = (p.Ndim + 127) / 128;
    const long long nb = (long long)p.n_groups * p.tiles_m * p.tiles_n;
    CSLGAN_REQUIRE(nb <= 0x7fffffffll, "conv2d_wgrad_scaled_bf16s: grid too large");
    note_kernel("igemm_mcs_tr_kernel<128,128,scaled>");
    if (c->Q % 8 == 0) hipLaunchKernelGGL((igemm_mcs_tr_kernel<true, true>), dim3((unsigned)nb), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((igemm_mcs_tr_kernel<false, true>), dim3((unsigned)nb), dim3(256), 0, st, p);
    return check_launch("igemm_mcs_tr_kernel");
}

// The head nn.Linear(C, 1) on bf16 features x [N, C] (C % 8 == 0): data gradient gx[n,:] = bf16(gy[n] * bf16(w) (* lrelu'(mask)))
// with fp32 gy [N] and bf16 mask / gx [N, C] ...
int cslgan_linear_k1_dgrad_bf16s(const float* gy, const float* w, const void* mask, int N, int64_t C, void* gx, void* stream) {
    CSLGAN_REQUIRE(gy && w && gx && N > 0 && N <= 65535 && C > 0 && C % 8 == 0, "linear_k1_dgrad_bf16s: bad argument");
    CSLGAN_REQUIRE(aligned16(w) && aligned16(gx) && (!mask || aligned16(mask)), "linear_k1_dgrad_bf16s: misaligned");
    const long long C8 = C / 8;
    note_kernel("linear_k1s_dgrad_kernel");
    hipLaunchKernelGGL(linear_k1s_dgrad_kernel, dim3((unsigned)((C8 + 255) / 256), (unsigned)N), dim3(256), 0, (hipStream_t)stream, gy, w,
                       reinterpret_cast<const unsigned short*>(mask), C8, reinterpret_cast<unsigned short*>(gx));
    return check_launch("linear_k1s_dgrad_kernel");
}

// ... and its grouped weight gradient gw[N/group, C] = alpha * sum_{n in g} gy[n] x[n,:] (fp32; nullable) and / or
// sq[N/group] += ||gw_g||^2.
int cslgan_linear_k1_wgrad_bf16s(const float* gy, const void* x, const float* row_scale, int N, int64_t C, int group, float alpha, float* gw, float* sq,
                                 void* stream) {
    CSLGAN_REQUIRE(gy && x && (gw || sq) && N > 0 && C > 0 && C % 8 == 0 && group >= 1 && N % group == 0 && N / group <= 65535,
                   "linear_k1_wgrad_bf16s: bad argument");
    CSLGAN_REQUIRE(aligned16(x) && (!gw || aligned16(gw)), "linear_k1_wgrad_bf16s: misaligned");
    const long long C8 = C / 8;
    note_kernel("linear_k1s_wgrad_kernel");
    hipLaunchKernelGGL(linear_k1s_wgrad_kernel, dim3((unsigned)((C8 + 255) / 256), (unsigned)(N / group)), dim3(256), 0, (hipStream_t)stream, gy,
                       reinterpret_cast<const unsigned short*>(x), C8, group, alpha, gw, sq, row_scale);
    return check_launch("linear_k1s_wgrad_kernel");
}

int cslgan_act_bwd_bf16(const void* g, const void* y, int64_t n, float slope, void* out, void* stream) {
    CSLGAN_REQUIRE(g && y && out && n >= 0, "act_bwd_bf16: bad argument");
    CSLGAN_REQUIRE(n % 8 == 0 && aligned16(g) && aligned16(y) && aligned16(out), "act_bwd_bf16: needs a multiple of 8 elements, 16-byte aligned");
    if (n == 0) return CSLGAN_OK;
    hipLaunchKernelGGL(act_bwd_bf16_kernel, dim3(stream_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint4*>(g),
                       reinterpret_cast<const uint4*>(y), (long long)(n / 8), slope, reinterpret_cast<uint4*>(out));
    return check_launch("act_bwd_bf16_kernel");
}

int cslgan_bias_grad_grouped_bf16(const void* gy, int N, int PQ, int K, int group, float alpha, float* gb, float* sq, void* stream) {
    CSLGAN_REQUIRE(gy && (gb || sq), "bias_grad_bf16: null argument");
    CSLGAN_REQUIRE(N > 0 && PQ > 0 && K > 0 && group >= 1 && N % group == 0, "bias_grad_bf16: bad sizes");
    CSLGAN_REQUIRE(K % 8 == 0 && K <= 2048 && 256 % (K / 8) == 0 && aligned16(gy), "bias_grad_bf16: K must be a multiple of 8 with 256 %% (K/8) == 0");
    hipLaunchKernelGGL(bias_grad_bf16_kernel, dim3((unsigned)(N / group)), dim3(256), sizeof(float) * K, (hipStream_t)stream,
                       reinterpret_cast<const unsigned short*>(gy), PQ, K, group, alpha, gb, sq);
    return check_launch("bias_grad_bf16_kernel");
}

}  // extern "C"
