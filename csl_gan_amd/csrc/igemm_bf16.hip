// bf16-MFMA implicit-GEMM convolution family (fp32 tensors in HBM, operands rounded to bfloat16 on their way into LDS,
// fp32 accumulate): forward conv / linear + data gradient (K-contiguous form) and the grouped weight gradient
// (M-contiguous form).  gfx950 only.  BASELINE.json configs[4] ("CelebA 128x128 DCResNet bf16 ..."), selected with
// cslgan_conv_t.compute == CSLGAN_COMPUTE_BF16 (`--compute_dtype bf16`).
//
// Same index maps, classes, epilogues and host-side setup as the fp32 kernels (igemm.h: KcParams / McParams); only the
// inner product changes: v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s dense, 16x the fp32 MFMA rate) instead of
// v_mfma_f32_32x32x2_f32.  A lane's operand of that instruction is 8 consecutive k of one row = 16 bytes, so the LDS image
// is [k/8][row][8 x bf16]: written as packed pairs (v_cvt_pk_bf16_f32, round-to-nearest-even) and read back with ONE
// ds_read_b128 per operand per 16 k, 16 consecutive rows per 16-lane group -> conflict-free.
//
//   K-contiguous (forward, data gradient): both operands have k contiguous in HBM; a 16-byte global load is 4 consecutive k
//     of one row, two such loads of a lane pair make one 8-k LDS entry.  Half-wave h of k-step s reads entry 2s+h — the same
//     permutation of k for A and B, so the sum is unchanged.
//   M-contiguous (weight gradient): the reduction index (pixel) is the slow one for both operands, so each thread gathers
//     the SAME 4 channels of 8 consecutive pixels (8 x 16-byte loads, coalesced across the lanes of a pixel row) and
//     transposes in registers: 4 LDS entries of 8 k each (ds_write_b128).
//
// Numerics: every product is bf16(a)*bf16(b) exactly (8-bit mantissas), summed in fp32 — relative operand error <= 2^-9.
// The fp32 path stays the default and the headline; this one is held to a bf16 tolerance stated in its tests.
//
// Replaces (reference file:line): the same calls as igemm_kc.hip / igemm_mc.hip — nn.Conv2d / nn.Linear forward and
// autograd data gradients (DCResNet_models.py:131-132,145, MNIST_models.py:41-46) and the Opacus-fork per-sample weight
// gradients (train.py:373,387).
#include <stdlib.h>
#include "common.h"
#include "igemm.h"

namespace cslgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {      // v_cvt_pk_bf16_f32: RNE, lo in bits 0..15
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ uint2 pack4_bf16(const float4& v) { return make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)); }

constexpr unsigned OOB16 = 0xFFFFFFF0u;
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}

// ---- fp32 from three bfloat16 pieces ---------------------------------------------------------------------------------------
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): three 8-bit mantissas cover fp32's 24 bits
// (|x - hi - mid - lo| <= 2^-24 |x|).  A product a*b is then the sum of 9 piece products, each EXACT in fp32 (8 x 8 bits);
// dropping the three smallest (mid*lo, lo*mid, lo*lo: <= 2^-23 |a||b| together) leaves SIX bf16 MFMAs per fp32 MFMA step:
//     a*b ~= hi*hi + (hi*mid + mid*hi) + (hi*lo + lo*hi + mid*mid)
// at 16x the fp32 MFMA rate each — 2.67x the fp32 matrix rate for a per-product error of about one fp32 ulp
// (CSLGAN_COMPUTE_BF16X3; the same construction vendor BLAS libraries ship as "fp32 emulation").  Small terms are added first.
struct bf16x3_t { uint2 hi, mid, lo; };
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ bf16x3_t split4_bf16(const float4& v) {
    bf16x3_t r;
    r.hi = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
    const float r0 = v.x - bf_lo(r.hi.x), r1 = v.y - bf_hi(r.hi.x), r2 = v.z - bf_lo(r.hi.y), r3 = v.w - bf_hi(r.hi.y);   // exact
    r.mid = make_uint2(pack_bf16(r0, r1), pack_bf16(r2, r3));
    r.lo = make_uint2(pack_bf16(r0 - bf_lo(r.mid.x), r1 - bf_hi(r.mid.x)), pack_bf16(r2 - bf_lo(r.mid.y), r3 - bf_hi(r.mid.y)));
    return r;
}

// ---- K-contiguous: Out[m][n] = epilogue( sum_k A(m,k) * Wm[n][k] ) ------------------------------------------------------
// K tile = 32 (two 16-k MFMA steps); LDS entry e = k/8 in the tile (4 entries), each [rows][8 bf16] + 16 B pad.
// NSPLIT = 1: plain bf16 operands, two LDS buffers.  NSPLIT = 3: three bf16 pieces per operand (fp32 emulation), one LDS
// buffer (3 x the image; 64 KB per workgroup is the limit) with the next tile held in registers across the MFMAs.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool VEC_A, bool VEC_B, int NSPLIT>
__global__ __launch_bounds__(256, 2) void igemm_kc_bf16_kernel(const KcParams p) {
    constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "bad tile");
    constexpr int A_ES = BM * 2 + 2, B_ES = BN * 2 + 2;      // uint2 (8-byte) units per LDS entry, padded by 16 B
    constexpr int A_PASS = BM / 32, B_PASS = BN / 32;
    constexpr int NBUF = NSPLIT == 1 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) uint2 As[NBUF][NSPLIT][4 * A_ES];
    __shared__ __attribute__((aligned(16))) uint2 Bs[NBUF][NSPLIT][4 * B_ES];
    __shared__ int s_tap[IG_MAX_TAPS];
    __shared__ int s_off[BM];
    __shared__ int s_roff[BM];

    const int tid = threadIdx.x;
    const int nwg = p.tiles_m * p.tiles_n;
    const int split = blockIdx.x / nwg;
    const int wg = xcd_remap(blockIdx.x - split * nwg, nwg);
    const int tile_mg = wg / p.tiles_n, tile_n = wg - tile_mg * p.tiles_n;
    int ci = 0;
#pragma unroll 1
    while (ci + 1 < p.n_cls && tile_mg >= p.cls[ci + 1].tile0) ++ci;
    const KcClass& kc = p.cls[ci];
    const int M = kc.M, OHc = kc.OHc, OWc = kc.OWc, Kdim = kc.Kdim;
    const int m0 = (tile_mg - kc.tile0) * BM, n0 = tile_n * BN;
    const float* __restrict__ wbase = p.w + kc.w_off;

    if (tid < IG_MAX_TAPS) s_tap[tid] = ((int)kc.ty[tid] << 16) | ((int)kc.tx[tid] & 0xffff);

    const int lrow = tid >> 3;   // 0..31
    const int q = tid & 7;       // 4-k chunk within the 32-k tile; chunks 2e, 2e+1 form LDS entry e
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wbase), 0, p.w_bytes - 4u * (unsigned)kc.w_off, 0x00020000);
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
        b_off[i] = n < p.Nn ? 4u * (unsigned)n * (unsigned)Kdim : OOB16;
    }
    __syncthreads();

    float4 ra[A_PASS], rb[B_PASS];
    int k_end = Kdim;
    auto a_offset = [&](int i, int ty, int tx, int c, bool kin) -> unsigned {
        const int iy = a_iy[i] + ty, ix = a_ix[i] + tx;
        const bool ok = kin && (unsigned)iy < (unsigned)p.VH && (unsigned)ix < (unsigned)p.VW;
        // branch-free: an invalid element ORs 0xFFFFFFF0 into its (always computed) offset -> the descriptor's range check returns zeros;
        // as `ok ? offset : OOB16` every load sat in its own exec-masked block (found in csrc/igemm_bf16s.hip: +7-10 % there)
        return (4u * (unsigned)(a_img[i] + (iy * p.AW + ix) * p.AC + c)) | (ok ? 0u : OOB16);
    };
    auto load_tile = [&](int kt) {
        const int kb = kt * IG_BK + q * 4;
        if (VEC_A) {
            const bool kin = kb < k_end;
            const int t = kin ? (p.AC == 1 ? kb : (int)__umulhi((unsigned)kb, p.ac_recip)) : 0;
            const int c = kb - t * p.AC;
            const int tap = s_tap[t];
            const int ty = tap >> 16, tx = (int)(short)(tap & 0xffff);
#pragma unroll
            for (int i = 0; i < A_PASS; ++i) ra[i] = bload4(a_rsrc, a_offset(i, ty, tx, c, kin));
        } else {
            float t4[A_PASS][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = kb + e;
                const bool kin = k < k_end;
                const int t = kin ? (p.AC == 1 ? k : (int)__umulhi((unsigned)k, p.ac_recip)) : 0;
                const int c = k - t * p.AC;
                const int tap = s_tap[t];
                const int ty = tap >> 16, tx = (int)(short)(tap & 0xffff);
#pragma unroll
                for (int i = 0; i < A_PASS; ++i) t4[i][e] = bload1(a_rsrc, a_offset(i, ty, tx, c, kin));
            }
#pragma unroll
            for (int i = 0; i < A_PASS; ++i) ra[i] = make_float4(t4[i][0], t4[i][1], t4[i][2], t4[i][3]);
        }
        if (VEC_B) {
            const unsigned kofs = kb < k_end ? 4u * (unsigned)kb : OOB16;
#pragma unroll
            for (int i = 0; i < B_PASS; ++i) rb[i] = bload4(w_rsrc, (b_off[i] + 4u * (unsigned)kb) | ((b_off[i] == OOB16 || kofs == OOB16) ? OOB16 : 0u));
        } else {
#pragma unroll
            for (int i = 0; i < B_PASS; ++i) {
                float t4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t4[e] = bload1(w_rsrc, (b_off[i] != OOB16 && (kb + e) < k_end) ? b_off[i] + 4u * (unsigned)(kb + e) : OOB16);
                rb[i] = make_float4(t4[0], t4[1], t4[2], t4[3]);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            const int at = (q >> 1) * A_ES + (lrow + 32 * i) * 2 + (q & 1);
            if (NSPLIT == 1) {
                As[buf][0][at] = pack4_bf16(ra[i]);
            } else {
                const bf16x3_t t = split4_bf16(ra[i]);
                As[buf][0][at] = t.hi; As[buf][NSPLIT > 1 ? 1 : 0][at] = t.mid; As[buf][NSPLIT > 2 ? 2 : 0][at] = t.lo;
            }
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
            const int at = (q >> 1) * B_ES + (lrow + 32 * i) * 2 + (q & 1);
            if (NSPLIT == 1) {
                Bs[buf][0][at] = pack4_bf16(rb[i]);
            } else {
                const bf16x3_t t = split4_bf16(rb[i]);
                Bs[buf][0][at] = t.hi; Bs[buf][NSPLIT > 1 ? 1 : 0][at] = t.mid; Bs[buf][NSPLIT > 2 ? 2 : 0][at] = t.lo;
            }
        }
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

    const int nk_all = (Kdim + IG_BK - 1) / IG_BK;
    int kt0 = 0, kt1 = nk_all;
    if (p.ksplit > 1) {
        const int per = (nk_all + p.ksplit - 1) / p.ksplit;
        kt0 = split * per;
        kt1 = kt0 + per < nk_all ? kt0 + per : nk_all;
        if (kt0 >= kt1) return;   // uniform across the workgroup
        k_end = kt1 * IG_BK < Kdim ? kt1 * IG_BK : Kdim;
    }
    load_tile(kt0);
    store_tile(0);
    __syncthreads();

    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = NBUF == 2 ? ((kt - kt0) & 1) : 0;
        load_tile(kt + 1);       // past the last tile every offset is out of range -> zeros, never read
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ea = (2 * s + h) * A_ES, eb = (2 * s + h) * B_ES;
            if (NSPLIT == 1) {
                bf16x8 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(&As[buf][0][ea + (arow0 + i * 32) * 2]);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][0][eb + (brow0 + j * 32) * 2]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            } else {
                bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[c][i] = *reinterpret_cast<const bf16x8*>(&As[buf][NSPLIT > c ? c : 0][ea + (arow0 + i * 32) * 2]);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[c][j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][NSPLIT > c ? c : 0][eb + (brow0 + j * 32) * 2]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {      // smallest terms first: hi*lo, lo*hi, mid*mid, then hi*mid, mid*hi, then hi*hi
                        f32x16 a = acc[i][j];
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], a, 0, 0, 0);
                        acc[i][j] = a;
                    }
            }
        }
        if (NBUF == 2) {
            store_tile(buf ^ 1);
            __syncthreads();
        } else {
            __syncthreads();             // every wavefront has read the tile
            store_tile(0);
            __syncthreads();
        }
    }

    // ---- epilogue (as igemm_kc) ---------------------------------------------------------------
    if (tid < BM) {
        const int m = m0 + tid;
        int off = -1, roff = 0;
        if (m < M) {
            if (p.dense_out && !p.res && !kc.patch) {
                off = m * p.ldo;
            } else {
                const RowCoord rc = kc_decode_row(m, OHc, OWc, kc.patch);
                off = kc_out_offset(p, kc, rc);
                if (p.res) roff = kc_res_offset(p, kc, rc);
            }
        }
        s_off[tid] = off;
        s_roff[tid] = roff;
    }
    __syncthreads();
    const bool atomic_out = p.ksplit > 1;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + r;
        if (n >= p.Nn) continue;
        const float bv = (p.bias && split == 0) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = wm * TM * 32 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                const int off = s_off[row];
                if (off < 0) continue;
                float val = acc[i][j][v] + bv;
                if (atomic_out) {
                    atomicAdd(p.out + off + n, val);
                    continue;
                }
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

template <int BM, int BN, int WM, int WN, int NSPLIT>
static int launch_kc_bf16_tile(KcParams& p, bool vecA, bool vecB, hipStream_t st, long long out_elems) {
    int tm = 0;
    for (int c = 0; c < p.n_cls; ++c) {
        p.cls[c].tile0 = tm;
        tm += (p.cls[c].M + BM - 1) / BM;
    }
    p.tiles_m = tm;
    p.tiles_n = (p.Nn + BN - 1) / BN;
    const int tiles = p.tiles_m * p.tiles_n;
    p.ksplit = 1;
    int nk_max = 0;
    for (int c = 0; c < p.n_cls; ++c) {
        const int nk = (p.cls[c].Kdim + IG_BK - 1) / IG_BK;
        nk_max = nk > nk_max ? nk : nk_max;
    }
    if (tiles < 96 && nk_max >= 16 && p.act == CSLGAN_ACT_NONE && !p.res && !p.mask && out_elems > 0) {
        const int want = (256 + tiles - 1) / tiles, cap = nk_max / 4;
        p.ksplit = want < cap ? want : cap;
        if (p.ksplit < 1) p.ksplit = 1;
    }
    if (p.ksplit > 1) {
        if (int rc = zero_floats(p.out, (size_t)out_elems, st)) return rc;
    }
    const dim3 grid((unsigned)(tiles * p.ksplit)), block(256);
    note_kernel(NSPLIT == 1 ? "igemm_kc_bf16_kernel<%d,%d>" : "igemm_kc_bf16x3_kernel<%d,%d>", BM, BN);
    if (vecA && vecB) hipLaunchKernelGGL((igemm_kc_bf16_kernel<BM, BN, WM, WN, true, true, NSPLIT>), grid, block, 0, st, p);
    else if (vecA) hipLaunchKernelGGL((igemm_kc_bf16_kernel<BM, BN, WM, WN, true, false, NSPLIT>), grid, block, 0, st, p);
    else if (vecB) hipLaunchKernelGGL((igemm_kc_bf16_kernel<BM, BN, WM, WN, false, true, NSPLIT>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((igemm_kc_bf16_kernel<BM, BN, WM, WN, false, false, NSPLIT>), grid, block, 0, st, p);
    return check_launch("igemm_kc_bf16_kernel");
}

// Called by launch_kc (igemm_kc.hip) after the operand-size checks, when KcParams::bf16 is set.
bool x3h_eligible(const KcParams& p);          // igemm_x3.hip: the LDS-halo form (stride-1 tap classes, pre-split filter)
int launch_x3h(KcParams& p, hipStream_t st);

int launch_kc_bf16(KcParams& p, hipStream_t st, long long out_elems) {
    if (x3h_eligible(p)) return launch_x3h(p, st);      // needs the pre-split / pre-rounded filter (p.w3)
    if (p.gn_part || p.in_scale) { set_error("conv2d_fwd: gn_part / in_scale given but the shape does not run on the LDS-halo kernel"); return CSLGAN_ERR_INVALID_ARG; }
    if (p.acc_classes) { set_error("igemm_kc_bf16: accumulated classes only run on the LDS-halo form"); return CSLGAN_ERR_INVALID_ARG; }
    for (int c = 0; c < p.n_cls; ++c) {
        KcClass& k = p.cls[c];
        k.patch = (k.T > 1 && k.OHc % 8 == 0 && k.OWc % 8 == 0) ? 1 : 0;
    }
    bool kd4 = true;
    for (int c = 0; c < p.n_cls; ++c) kd4 = kd4 && (p.cls[c].Kdim % 4 == 0) && (p.cls[c].w_off % 4 == 0);
    const bool vecA = (p.AC % 4 == 0) && aligned16(p.a);
    const bool vecB = kd4 && aligned16(p.w);
    long long rows = 0;
    for (int c = 0; c < p.n_cls; ++c) rows += (p.cls[c].M + 127) / 128;
    if (p.bf16 == 3) {
        if (p.Nn <= 64) return launch_kc_bf16_tile<128, 64, 2, 2, 3>(p, vecA, vecB, st, out_elems);
        if (rows * ((p.Nn + 127) / 128) >= 256) return launch_kc_bf16_tile<128, 128, 2, 2, 3>(p, vecA, vecB, st, out_elems);
        return launch_kc_bf16_tile<64, 128, 1, 4, 3>(p, vecA, vecB, st, out_elems);
    }
    if (p.Nn <= 64) return launch_kc_bf16_tile<128, 64, 2, 2, 1>(p, vecA, vecB, st, out_elems);
    if (rows * ((p.Nn + 127) / 128) >= 256) return launch_kc_bf16_tile<128, 128, 2, 2, 1>(p, vecA, vecB, st, out_elems);
    return launch_kc_bf16_tile<64, 128, 1, 4, 1>(p, vecA, vecB, st, out_elems);
}

// ---- M-contiguous: gw[g][m][n] = alpha * sum_{k in group g} GY[k][m] * X(k, n) ----------------------------------------
// K tile = 32 pixels.  Threads 0..127 gather GY, threads 128..255 gather X: thread -> (4 consecutive m or n, one group of
// 8 consecutive pixels) = 8 x 16-byte loads, transposed in registers into 4 LDS entries [k/8][m][8 bf16].
constexpr int MCB_BK = 32;

template <int BM, int BN, int WAVES_M, int WAVES_N, bool VEC_A, bool VEC_B, int NSPLIT>
__global__ __launch_bounds__(256, 2) void igemm_mc_bf16_kernel(const McParams p) {
    constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1 && BM == 128 && BN == 128, "128x128 tile only");
    constexpr int ES = 128 * 2 + 2;                          // uint2 units per LDS entry (128 rows x 16 B + 16 B pad)
    constexpr int NBUF = NSPLIT == 1 ? 2 : 1;                // three pieces per operand: one buffer (64 KB per workgroup)
    __shared__ __attribute__((aligned(16))) uint2 As[NBUF][NSPLIT][4 * ES];
    __shared__ __attribute__((aligned(16))) uint2 Bs[NBUF][NSPLIT][4 * ES];
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

    const bool is_a = tid < 128;
    const int lt = tid & 127;
    const int c4 = (lt & 31) * 4;        // first of this thread's 4 rows (m or n) within the tile
    const int kg = lt >> 5;              // its group of 8 pixels within the 32-pixel tile
    // X columns are fixed per thread: decode (tap, c) once
    int b_ty[4], b_tx[4], b_c[4];
    bool b_nok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int n = n0 + c4 + e;
        b_nok[e] = n < p.Ndim;
        const int t = b_nok[e] ? n / p.C : 0;
        b_c[e] = n - t * p.C;
        b_ty[e] = p.ty[t];
        b_tx[e] = p.tx[t];
    }

    float4 rv[8];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = kt * MCB_BK + kg * 8 + j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kk < Ktot) {
                if (is_a) {
                    const float* src = p.gy + (pix_base + kk) * p.Kc + m0 + c4;
                    if (VEC_A) {
                        if (m0 + c4 < p.Kc) v = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (m0 + c4 + 0 < p.Kc) v.x = src[0];
                        if (m0 + c4 + 1 < p.Kc) v.y = src[1];
                        if (m0 + c4 + 2 < p.Kc) v.z = src[2];
                        if (m0 + c4 + 3 < p.Kc) v.w = src[3];
                    }
                    if (p.row_scale) {
                        const float sc = p.row_scale[g * p.group + kk / PQ];
                        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
                    }
                } else {
                    const int il = kk / PQ;
                    const int pix = kk - il * PQ;
                    const int oy = pix / p.Q, ox = pix - oy * p.Q;
                    const long long img = (long long)g * p.group + il;
                    const int by = oy * p.stride, bx = ox * p.stride;
                    if (VEC_B) {
                        const int iy = by + b_ty[0], ix = bx + b_tx[0];
                        if (b_nok[0] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                            v = *reinterpret_cast<const float4*>(p.x + ((img * p.H + iy) * p.W + ix) * p.C + b_c[0]);
                    } else {
                        float t4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int iy = by + b_ty[e], ix = bx + b_tx[e];
                            t4[e] = (b_nok[e] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                                        ? p.x[((img * p.H + iy) * p.W + ix) * p.C + b_c[e]] : 0.f;
                        }
                        v = make_float4(t4[0], t4[1], t4[2], t4[3]);
                    }
                }
            }
            rv[j] = v;
        }
    };
    // rv[j] = (row c4+0..3) at pixel j of the group  ->  entry kg, row c4+e: the 8 pixels of row e, 16 bytes
    auto store_tile = [&](int buf) {
        const int rot = lt & 3;             // lanes start on different rows: fewer LDS bank conflicts on the 64-byte row stride
#pragma unroll
        for (int c = 0; c < NSPLIT; ++c) {
            uint2* dst = (is_a ? As[buf][c] : Bs[buf][c]) + kg * ES;
            const uint4 w0 = make_uint4(pack_bf16(rv[0].x, rv[1].x), pack_bf16(rv[2].x, rv[3].x), pack_bf16(rv[4].x, rv[5].x), pack_bf16(rv[6].x, rv[7].x));
            const uint4 w1 = make_uint4(pack_bf16(rv[0].y, rv[1].y), pack_bf16(rv[2].y, rv[3].y), pack_bf16(rv[4].y, rv[5].y), pack_bf16(rv[6].y, rv[7].y));
            const uint4 w2 = make_uint4(pack_bf16(rv[0].z, rv[1].z), pack_bf16(rv[2].z, rv[3].z), pack_bf16(rv[4].z, rv[5].z), pack_bf16(rv[6].z, rv[7].z));
            const uint4 w3 = make_uint4(pack_bf16(rv[0].w, rv[1].w), pack_bf16(rv[2].w, rv[3].w), pack_bf16(rv[4].w, rv[5].w), pack_bf16(rv[6].w, rv[7].w));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int er = (e + rot) & 3;
                const uint4 w = er == 0 ? w0 : (er == 1 ? w1 : (er == 2 ? w2 : w3));
                *reinterpret_cast<uint4*>(&dst[(c4 + er) * 2]) = w;
            }
            if (c + 1 < NSPLIT) {           // next piece: what the pieces so far leave of each value (exact subtractions)
                const unsigned* u0 = &w0.x; const unsigned* u1 = &w1.x; const unsigned* u2 = &w2.x; const unsigned* u3 = &w3.x;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int wd = j >> 1;
                    rv[j].x -= (j & 1) ? bf_hi(u0[wd]) : bf_lo(u0[wd]);
                    rv[j].y -= (j & 1) ? bf_hi(u1[wd]) : bf_lo(u1[wd]);
                    rv[j].z -= (j & 1) ? bf_hi(u2[wd]) : bf_lo(u2[wd]);
                    rv[j].w -= (j & 1) ? bf_hi(u3[wd]) : bf_lo(u3[wd]);
                }
            }
        }
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

    const int nk_all = (Ktot + MCB_BK - 1) / MCB_BK;
    int kt0 = 0, nk = nk_all;
    if (p.ksplit > 1) {
        const int per = (nk_all + p.ksplit - 1) / p.ksplit;
        kt0 = split * per;
        nk = kt0 + per < nk_all ? kt0 + per : nk_all;
        if (kt0 >= nk) return;      // uniform
    }
    load_tile(kt0);
    store_tile(0);
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        const int buf = NBUF == 2 ? ((kt - kt0) & 1) : 0;
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int eo = (2 * s + h) * ES;
            if (NSPLIT == 1) {
                bf16x8 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(&As[buf][0][eo + (arow0 + i * 32) * 2]);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][0][eo + (brow0 + j * 32) * 2]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            } else {
                bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[c][i] = *reinterpret_cast<const bf16x8*>(&As[buf][NSPLIT > c ? c : 0][eo + (arow0 + i * 32) * 2]);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[c][j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][NSPLIT > c ? c : 0][eo + (brow0 + j * 32) * 2]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        f32x16 a = acc[i][j];
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], a, 0, 0, 0);
                        acc[i][j] = a;
                    }
            }
        }
        if (NBUF == 2) {
            if (kt + 1 < nk) store_tile(buf ^ 1);
            __syncthreads();
        } else {
            __syncthreads();
            if (kt + 1 < nk) store_tile(0);
            __syncthreads();
        }
    }

    // ---- epilogue: scale, store, per-group sum of squares (as igemm_mc) ------------------------
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

int sqnorm_rows_accumulate(const float* in, long long n_rows, long long len, float* sq_accum, hipStream_t st);   // clip_kernels.hip

// Called by the wgrad entries (igemm_mc.hip) when cslgan_conv_t.compute == CSLGAN_COMPUTE_BF16.
int launch_mc_bf16(McParams& p, bool vecA, bool vecB, hipStream_t st, int nsplit) {
    p.tiles_m = (p.Kc + 127) / 128;
    p.tiles_n = (p.Ndim + 127) / 128;
    p.ksplit = 1;
    {
        const long long base = (long long)p.n_groups * p.tiles_m * p.tiles_n;
        const int nk_all = (p.group * p.P * p.Q + MCB_BK - 1) / MCB_BK;
        if (p.gw && !p.out_bf16 && base < 192 && nk_all >= 16) {
            const long long want = (512 + base - 1) / base, cap = nk_all / 4;
            p.ksplit = (int)(want < cap ? want : cap);
            if (p.ksplit < 1) p.ksplit = 1;
        }
    }
    if (p.ksplit > 1) {
        if (int rc = zero_floats(p.gw, (size_t)p.n_groups * p.Kc * p.Ndim, st)) return rc;
    }
    const long long nb = (long long)p.n_groups * p.tiles_m * p.tiles_n * p.ksplit;
    if (nb > 0x7fffffffll) { set_error("wgrad_bf16: grid too large"); return CSLGAN_ERR_INVALID_ARG; }
    const dim3 grid((unsigned)nb), block(256);
    if (nsplit == 3) {
        note_kernel("igemm_mc_bf16x3_kernel<128,128>");
        if (vecA && vecB) hipLaunchKernelGGL((igemm_mc_bf16_kernel<128, 128, 2, 2, true, true, 3>), grid, block, 0, st, p);
        else if (vecA) hipLaunchKernelGGL((igemm_mc_bf16_kernel<128, 128, 2, 2, true, false, 3>), grid, block, 0, st, p);
        else if (vecB) hipLaunchKernelGGL((igemm_mc_bf16_kernel<128, 128, 2, 2, false, true, 3>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((igemm_mc_bf16_kernel<128, 128, 2, 2, false, false, 3>), grid, block, 0, st, p);
    } else {
        note_kernel("igemm_mc_bf16_kernel<128,128>");
        if (vecA && vecB) hipLaunchKernelGGL((igemm_mc_bf16_kernel<128, 128, 2, 2, true, true, 1>), grid, block, 0, st, p);
        else if (vecA) hipLaunchKernelGGL((igemm_mc_bf16_kernel<128, 128, 2, 2, true, false, 1>), grid, block, 0, st, p);
        else if (vecB) hipLaunchKernelGGL((igemm_mc_bf16_kernel<128, 128, 2, 2, false, true, 1>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((igemm_mc_bf16_kernel<128, 128, 2, 2, false, false, 1>), grid, block, 0, st, p);
    }
    int rc = check_launch("igemm_mc_bf16_kernel");
    if (rc) return rc;
    if (p.ksplit > 1 && p.sq) rc = sqnorm_rows_accumulate(p.gw, p.n_groups, (long long)p.Kc * p.Ndim, p.sq, st);
    return rc;
}

}  // namespace cslgan
