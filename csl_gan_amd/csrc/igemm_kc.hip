// fp32 MFMA implicit-GEMM convolution, "K-contiguous" form: forward conv / linear and the data
// gradient (transposed conv).  gfx950 only.
//
//   Out[m][n] = epilogue( sum_k A(m,k) * Wm[n][k] )      (index maps and classes: igemm.h)
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
// One launch covers every output class of an op (the 4 parity classes of a stride-2 data gradient),
// so small layers still put >= 256 workgroups on the chip;
// layers with few tiles and a long K (the [B,8192]x[8192,1] critic head) split K over workgroups.
//
// Replaces (reference file:line): torch.nn.Conv2d / nn.Linear forward DCResNet_models.py:131-132,
// 145, 16 (the conv of UpsampleConv, on the depth-to-space tensor), 60-70, 95-104; MNIST_models.py:17-23, 41-46; and the autograd data-gradient of those.
#include <stdlib.h>
#include "common.h"
#include "igemm.h"

namespace cslgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Range-checked buffer loads: an offset at or beyond the descriptor's byte count returns 0, so padding taps,
// ragged rows and the K tail need neither a branch nor a select — the loader is straight-line code that the
// scheduler can interleave with MFMAs.  32-bit byte offsets (tensors are < 4 GB; checked on the host).
constexpr unsigned OOB = 0xFFFFFFF0u;
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}

// NBUF = 2: LDS double buffer, one barrier per K tile (2 workgroups/CU for the 128x128 tile).
// NBUF = 1: single LDS buffer, two barriers per K tile, half the LDS -> twice the resident wavefronts,
//           which hide the loader's address arithmetic and LDS latency behind other waves' MFMAs.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool VEC_A, bool VEC_B, int NBUF>
__global__ __launch_bounds__(256, (NBUF == 1 ? 4 : 2)) void igemm_kc_kernel(const KcParams p) {
    constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "bad tile");
    constexpr int A_CH = BM * 4 + 4, B_CH = BN * 4 + 4;  // floats per k-chunk (padded)
    constexpr int A_PASS = BM / 32, B_PASS = BN / 32;
    __shared__ __attribute__((aligned(16))) float As[NBUF][8 * A_CH];
    __shared__ __attribute__((aligned(16))) float Bs[NBUF][8 * B_CH];
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

    // ---- per-thread loader coordinates -----------------------------------------------------
    const int lrow = tid >> 3;   // 0..31
    const int q = tid & 7;       // k-chunk within the tile
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wbase), 0, p.w_bytes - 4u * (unsigned)kc.w_off, 0x00020000);
    int a_img[A_PASS], a_iy[A_PASS], a_ix[A_PASS];   // a_img: image base in elements; a_iy = -2^20 marks a row past M
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
        const int m = m0 + lrow + 32 * i;
        const bool ok = m < M;
        const RowCoord rc = kc_decode_row(ok ? m : 0, OHc, OWc, kc.patch);
        a_img[i] = rc.img * p.AH * p.AW * p.AC;
        a_iy[i] = ok ? rc.oy * p.sy : -(1 << 20);
        a_ix[i] = rc.ox * p.sx;
    }
    unsigned b_off[B_PASS];      // byte offset of row n in this class's filter matrix, OOB when n >= Nn
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
        const int n = n0 + lrow + 32 * i;
        b_off[i] = n < p.Nn ? 4u * (unsigned)n * (unsigned)Kdim : OOB;
    }
    __syncthreads();  // s_tap visible

    float4 ra[A_PASS], rb[B_PASS];

    auto a_offset = [&](int i, int ty, int tx, int c, bool kin) -> unsigned {
        const int iy = a_iy[i] + ty, ix = a_ix[i] + tx;
        const bool ok = kin && (unsigned)iy < (unsigned)p.VH && (unsigned)ix < (unsigned)p.VW;
        const unsigned e = (unsigned)(a_img[i] + (iy * p.AW + ix) * p.AC + c);
        return ok ? 4u * e : OOB;
    };

    int k_end = Kdim;            // exclusive K bound of this workgroup's slice (split-K), set below
    // The loader is split in two so that no address arithmetic sits between a barrier and the loads it
    // feeds: calc_offsets(kt) produces the byte offsets of tile kt (pure VALU, scheduled freely among the
    // MFMAs of the previous tile), issue_loads() turns the stored offsets into buffer loads.
    unsigned oa[VEC_A ? A_PASS : A_PASS * 4], ob[VEC_B ? B_PASS : B_PASS * 4];
    // part: 0 = everything, 1 = A rows [0, A_PASS/2), 2 = A rows [A_PASS/2, A_PASS), 3 = B rows
    auto calc_offsets = [&](int kt, int part = 0) {
        const int kb = kt * IG_BK + q * 4;
        const int Kdim = k_end;  // shadows the class's Kdim: beyond this workgroup's K slice everything is OOB
        const int a_lo = part == 2 ? A_PASS / 2 : 0, a_hi = part == 1 ? A_PASS / 2 : (part == 3 ? 0 : A_PASS);
        if (VEC_A) {
            const bool kin = kb < Kdim;
            const int t = kin ? (p.AC == 1 ? kb : (int)__umulhi((unsigned)kb, p.ac_recip)) : 0;   // kb / AC (range checked on the host)
            const int c = kb - t * p.AC;
            const int tap = s_tap[t];
            const int ty = tap >> 16, tx = (int)(short)(tap & 0xffff);
#pragma unroll
            for (int i = 0; i < A_PASS; ++i)
                if (i >= a_lo && i < a_hi) oa[i] = a_offset(i, ty, tx, c, kin);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = kb + e;
                const bool kin = k < Kdim;
                const int t = kin ? (p.AC == 1 ? k : (int)__umulhi((unsigned)k, p.ac_recip)) : 0;
                const int c = k - t * p.AC;
                const int tap = s_tap[t];
                const int ty = tap >> 16, tx = (int)(short)(tap & 0xffff);
#pragma unroll
                for (int i = 0; i < A_PASS; ++i)
                    if (i >= a_lo && i < a_hi) oa[i * 4 + e] = a_offset(i, ty, tx, c, kin);
            }
        }
        if (part == 1 || part == 2) return;
        if (VEC_B) {
            const unsigned kofs = kb < Kdim ? 4u * (unsigned)kb : OOB;
#pragma unroll
            for (int i = 0; i < B_PASS; ++i) ob[i] = (b_off[i] == OOB || kofs == OOB) ? OOB : b_off[i] + kofs;
        } else {
#pragma unroll
            for (int i = 0; i < B_PASS; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) ob[i * 4 + e] = (b_off[i] != OOB && (kb + e) < Kdim) ? b_off[i] + 4u * (unsigned)(kb + e) : OOB;
        }
    };
    auto issue_loads = [&]() {
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            if (VEC_A) ra[i] = buf_load4(a_rsrc, oa[i]);
            else ra[i] = make_float4(buf_load1(a_rsrc, oa[i * 4]), buf_load1(a_rsrc, oa[i * 4 + 1]), buf_load1(a_rsrc, oa[i * 4 + 2]), buf_load1(a_rsrc, oa[i * 4 + 3]));
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
            if (VEC_B) rb[i] = buf_load4(w_rsrc, ob[i]);
            else rb[i] = make_float4(buf_load1(w_rsrc, ob[i * 4]), buf_load1(w_rsrc, ob[i * 4 + 1]), buf_load1(w_rsrc, ob[i * 4 + 2]), buf_load1(w_rsrc, ob[i * 4 + 3]));
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

    const int nk_all = (Kdim + IG_BK - 1) / IG_BK;
    int kt0 = 0, kt1 = nk_all;
    if (p.ksplit > 1) {
        const int per = (nk_all + p.ksplit - 1) / p.ksplit;
        kt0 = split * per;
        kt1 = kt0 + per < nk_all ? kt0 + per : nk_all;
        if (kt0 >= kt1) return;   // uniform across the workgroup
        k_end = kt1 * IG_BK < Kdim ? kt1 * IG_BK : Kdim;
    }
    calc_offsets(kt0);
    issue_loads();
    store_tile(0);
    calc_offsets(kt0 + 1);
    __syncthreads();

    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = NBUF == 2 ? ((kt - kt0) & 1) : 0;
        // LDS fragments are software-pipelined one k-group ahead of the MFMAs that consume them, and the
        // next tile's global loads (address arithmetic + 8 x dwordx4) are issued between the first and the
        // second k-group so that they execute in the shadow of MFMAs instead of in front of them.
        float4 af[2][TM], bf[2][TN];
        auto load_frags = [&](int g, int slot) {
            const int ch = 2 * g + h;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[slot][i] = *reinterpret_cast<const float4*>(&As[buf][ch * A_CH + (arow0 + i * 32) * 4]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[slot][j] = *reinterpret_cast<const float4*>(&Bs[buf][ch * B_CH + (brow0 + j * 32) * 4]);
        };
        issue_loads();       // tile kt+1 (offsets ready since the previous iteration; past the last tile all are OOB -> 0)
        load_frags(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cur = g & 1;
            if (g < 3) load_frags(g + 1, cur ^ 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].x, bf[cur][j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].y, bf[cur][j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].z, bf[cur][j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].w, bf[cur][j].w, acc[i][j], 0, 0, 0);
                }
            // a third of tile kt+2's offset arithmetic rides behind each of the first three k-groups
            if (g < 3) {
                calc_offsets(kt + 2, g + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (NBUF == 2) {
            store_tile(buf ^ 1);             // unconditional (the buffer written after the last tile is never read)
            __syncthreads();
        } else {
            __syncthreads();                 // every wave has finished reading the tile
            if (kt + 1 < kt1) {
                store_tile(0);
                __syncthreads();
            }
        }
    }

    // ---- epilogue --------------------------------------------------------------------------
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

// Repack KRSC filters into per-class [Nout][taps][Cred] matrices.
//   transposed == 1 (data gradient): wt[off + (c*Tc + t)*K + k]      = w[((k*R + kh_t)*S + kw_t)*C + c]
//   transposed == 0 (stride-2 forward parity classes): wt[off + (k*Tc + t)*C + c] = the (kh,kw) of tap t
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float x) {      // the rounding v_cvt_pk_bf16_f32 performs (finite inputs)
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    const f2_t v = {x, 0.f};
    return (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(v, b2_t)) & 0xffffu);
}

struct RepackArgs {
    int K, R, S, C;
    int n_class;
    int transposed;
    int cls_off[IG_MAX_CLS];
    int cls_T[IG_MAX_CLS];
    // transposed: the single (kh,kw) of tap t.  phases: [kh_lo,kh_hi) x [kw_lo,kw_hi) folded onto tap t
    signed char kh_lo[IG_MAX_CLS][IG_MAX_TAPS], kh_hi[IG_MAX_CLS][IG_MAX_TAPS];
    signed char kw_lo[IG_MAX_CLS][IG_MAX_TAPS], kw_hi[IG_MAX_CLS][IG_MAX_TAPS];
    // pieces > 0 (the LDS-halo kernels of the bf16 matrix cores, csrc/igemm_x3.hip): the same launch also writes every class matrix
    // split into `pieces` bfloat16 pieces in step-major order, w3[pieces * cls_off + piece * rows*Tc*red + ((red_ch / 16) * Tc + t) * rows * 16
    // + row * 16 + red_ch % 16]  (rows = C, red = K when transposed; rows = K, red = C otherwise; red % 16 == 0) — round 3 ran one
    // repack launch + one split launch PER CLASS (five launches per conv and direction, thirty per D-step on weights that change
    // every step)
    int pieces;
    unsigned short* w3;
};

__global__ void repack_filters_kernel(const float* __restrict__ w, float* __restrict__ wt, RepackArgs a) {
    const int cls = blockIdx.y;
    const int Tc = a.cls_T[cls];
    const long long total = (long long)a.C * Tc * a.K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int k, t, c;
        if (a.transposed) {
            k = (int)(i % a.K);
            const long long rest = i / a.K;
            t = (int)(rest % Tc);
            c = (int)(rest / Tc);
        } else {
            c = (int)(i % a.C);
            const long long rest = i / a.C;
            t = (int)(rest % Tc);
            k = (int)(rest / Tc);
        }
        float sum = 0.f;
        for (int kh = a.kh_lo[cls][t]; kh < a.kh_hi[cls][t]; ++kh)
            for (int kw = a.kw_lo[cls][t]; kw < a.kw_hi[cls][t]; ++kw)
                sum += w[(((long long)k * a.R + kh) * a.S + kw) * a.C + c];
        wt[a.cls_off[cls] + i] = sum;
        if (a.pieces == 4) {        // exact fp32 on the LDS-halo kernel (igemm_x3h<., 0, .>): the class matrix again, step-major, fp32
            const int rows = a.transposed ? a.C : a.K, row = a.transposed ? c : k, red = a.transposed ? k : c;
            reinterpret_cast<float*>(a.w3)[(long long)a.cls_off[cls] + ((((long long)(red >> 4) * Tc + t) * rows + row) << 4) + (red & 15)] = sum;
        } else if (a.pieces) {
            const int rows = a.transposed ? a.C : a.K, row = a.transposed ? c : k, red = a.transposed ? k : c;
            unsigned short* dst = a.w3 + (long long)a.pieces * a.cls_off[cls] + ((((long long)(red >> 4) * Tc + t) * rows + row) << 4) + (red & 15);
            const unsigned short hi = f32_to_bf16_rne(sum);
            dst[0] = hi;
            if (a.pieces == 3) {
                const float r1 = sum - __uint_as_float((unsigned)hi << 16);
                const unsigned short mid = f32_to_bf16_rne(r1);
                const long long plane = total;
                dst[plane] = mid;
                dst[2 * plane] = f32_to_bf16_rne(r1 - __uint_as_float((unsigned)mid << 16));
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_kc_tile(KcParams& p, bool vecA, bool vecB, hipStream_t st, long long out_elems) {
    int tm = 0;
    for (int c = 0; c < p.n_cls; ++c) {
        p.cls[c].tile0 = tm;
        tm += (p.cls[c].M + BM - 1) / BM;
    }
    p.tiles_m = tm;
    p.tiles_n = (p.Nn + BN - 1) / BN;
    const int tiles = p.tiles_m * p.tiles_n;
    // split K only for purely linear epilogues and launches that would leave most CUs idle
    p.ksplit = 1;
    int nk_max = 0;
    for (int c = 0; c < p.n_cls; ++c) {
        const int nk = (p.cls[c].Kdim + IG_BK - 1) / IG_BK;
        nk_max = nk > nk_max ? nk : nk_max;
    }
    if (tiles < 96 && nk_max >= 16 && p.act == CSLGAN_ACT_NONE && !p.res && !p.mask && out_elems > 0) {
        int want = (256 + tiles - 1) / tiles;
        const int cap = nk_max / 4;
        p.ksplit = want < cap ? want : cap;
        if (p.ksplit < 1) p.ksplit = 1;
    }
    if (p.ksplit > 1) {
        if (int rc = zero_floats(p.out, (size_t)out_elems, st)) return rc;
    }
    const dim3 grid((unsigned)(tiles * p.ksplit)), block(256);
    // NBUF = 1 (single LDS buffer, 4 workgroups/CU) was measured slower on every shape of the D-step
    // (G b1: 59 vs 102 TF; G b4: 107 vs 105 TF), so only the double-buffered form is instantiated.
    note_kernel("igemm_kc_kernel<%d,%d>", BM, BN);
    if (vecA && vecB) hipLaunchKernelGGL((igemm_kc_kernel<BM, BN, WM, WN, true, true, 2>), grid, block, 0, st, p);
    else if (vecA) hipLaunchKernelGGL((igemm_kc_kernel<BM, BN, WM, WN, true, false, 2>), grid, block, 0, st, p);
    else if (vecB) hipLaunchKernelGGL((igemm_kc_kernel<BM, BN, WM, WN, false, true, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((igemm_kc_kernel<BM, BN, WM, WN, false, false, 2>), grid, block, 0, st, p);
    return check_launch("igemm_kc_kernel");
}

static long long tiles_for(const KcParams& p, int BM, int BN) {
    long long tm = 0;
    for (int c = 0; c < p.n_cls; ++c) tm += (p.cls[c].M + BM - 1) / BM;
    return tm * ((p.Nn + BN - 1) / BN);
}

int launch_kc_bf16(KcParams& p, hipStream_t st, long long out_elems);     // igemm_bf16.hip
int split_filter_x3(const float* w, int Nn, int T, int C, void* w3, hipStream_t st, int pieces);      // igemm_x3.hip
bool x3h_eligible(const KcParams& p);
int launch_x3h(KcParams& p, hipStream_t st);
bool halo_eligible(const KcParams& p);          // igemm_halo.hip
int launch_halo(KcParams& p, hipStream_t st);
bool skinny_eligible(const KcParams& p);        // igemm_skinny.hip
int launch_skinny(KcParams& p, hipStream_t st);

// out_elems: total floats of the output tensor (needed to zero it when K is split), or 0 to forbid splitting
int launch_kc(KcParams& p, hipStream_t st, long long out_elems) {
    long long rows = 0;
    for (int c = 0; c < p.n_cls; ++c) rows += p.cls[c].M;
    if (rows <= 0 || p.Nn <= 0) return CSLGAN_OK;
    {
        const long long n_img = p.cls[0].M / ((long long)p.cls[0].OHc * p.cls[0].OWc);
        const long long a_b = (p.a_bf16 ? 2ll : 4ll) * n_img * p.AH * p.AW * p.AC;
        long long w_end = 0;
        for (int c = 0; c < p.n_cls; ++c) {
            const long long e = (long long)p.cls[c].w_off + (long long)p.Nn * p.cls[c].Kdim;
            w_end = e > w_end ? e : w_end;
        }
        if (a_b >= 0xFFFFFFF0ll || 4 * w_end >= 0xFFFFFFF0ll) {
            set_error("igemm_kc: operand larger than 4 GB");
            return CSLGAN_ERR_INVALID_ARG;
        }
        p.a_bytes = (unsigned)a_b;
        p.w_bytes = (unsigned)(4 * w_end);
        // k / AC as umulhi(k, ceil(2^32 / AC)): exact while k < 2^32 / AC
        long long kmax = 0;
        for (int c = 0; c < p.n_cls; ++c) kmax = p.cls[c].Kdim > kmax ? p.cls[c].Kdim : kmax;
        if (p.AC < 2 || (kmax + IG_BK) * (long long)p.AC >= (1ll << 32)) {
            if (p.AC == 1 && kmax + IG_BK < (1ll << 31)) p.ac_recip = 0xFFFFFFFFu;   // handled exactly below
            else { set_error("igemm_kc: K too large for reciprocal division"); return CSLGAN_ERR_INVALID_ARG; }
        } else {
            p.ac_recip = (unsigned)(((1ull << 32) + (unsigned long long)p.AC - 1) / (unsigned long long)p.AC);
        }
    }
    // 1..4 output channels: vector-ALU kernel, fp32 in every compute mode (a 32-wide MFMA tile would be 29/32 padding)
    static const int skinny_env = [] { const char* e = getenv("CSLGAN_KC_SKINNY"); return e ? atoi(e) : 1; }();
    if (skinny_env && !p.acc_classes && skinny_eligible(p)) {
        if (p.gn_part) { set_error("conv2d_fwd: gn_part is not produced by the 1..4-output-channel kernel"); return CSLGAN_ERR_INVALID_ARG; }
        return launch_skinny(p, st);
    }
    if (p.a_bf16) { set_error("igemm_kc: a bfloat16-stored input is only taken by the 1..4-output-channel kernel (64 input channels, stride 1)"); return CSLGAN_ERR_INVALID_ARG; }
    if (p.bf16) {
        return launch_kc_bf16(p, st, out_elems);
    }
    if (p.w3 && x3h_eligible(p)) return launch_x3h(p, st);          // exact fp32 on the round-4 halo kernel (step-major fp32 filter copy in p.w3)
    if (p.gn_part || p.in_scale) { set_error("conv2d_fwd: gn_part / in_scale given but the shape does not run on the LDS-halo kernel"); return CSLGAN_ERR_INVALID_ARG; }
    static const int halo_env = [] { const char* e = getenv("CSLGAN_KC_HALO"); return e ? atoi(e) : 1; }();
    if (halo_env && halo_eligible(p)) return launch_halo(p, st);
    static const int patch_env = [] { const char* e = getenv("CSLGAN_KC_PATCH"); return e ? atoi(e) : 1; }();
    for (int c = 0; c < p.n_cls; ++c) {
        KcClass& k = p.cls[c];
        k.patch = (patch_env && k.T > 1 && k.OHc % 8 == 0 && k.OWc % 8 == 0) ? 1 : 0;
    }
    bool kd4 = true;
    for (int c = 0; c < p.n_cls; ++c) kd4 = kd4 && (p.cls[c].Kdim % 4 == 0) && (p.cls[c].w_off % 4 == 0);
    const bool vecA = (p.AC % 4 == 0) && aligned16(p.a);
    const bool vecB = kd4 && aligned16(p.w);
    if (p.Nn <= 32) return launch_kc_tile<128, 32, 4, 1>(p, vecA, vecB, st, out_elems);
    if (p.Nn <= 64) {
        if (tiles_for(p, 128, 64) >= 192) return launch_kc_tile<128, 64, 2, 2>(p, vecA, vecB, st, out_elems);
        return launch_kc_tile<64, 64, 2, 2>(p, vecA, vecB, st, out_elems);
    }
    static const int t128 = [] { const char* e = getenv("CSLGAN_KC_T128"); return e ? atoi(e) : 300; }();
    // classes of one launch have different K (9/6/6/4 taps for a 5x5 stride-2 data gradient): with about one
    // workgroup per CU the launch lasts as long as its heaviest class, so multi-class launches want more, smaller tiles
    static const int tmc = [] { const char* e = getenv("CSLGAN_KC_TMC"); return e ? atoi(e) : 520; }();
    const int t64 = p.n_cls > 1 ? tmc : 192;
    if (tiles_for(p, 128, 128) >= t128) return launch_kc_tile<128, 128, 2, 2>(p, vecA, vecB, st, out_elems);
    if (tiles_for(p, 64, 128) >= t64) return launch_kc_tile<64, 128, 1, 4>(p, vecA, vecB, st, out_elems);
    return launch_kc_tile<64, 64, 2, 2>(p, vecA, vecB, st, out_elems);
}

static int check_conv(const cslgan_conv_t* c, const char* who) {
    CSLGAN_REQUIRE(c->N > 0 && c->H > 0 && c->W > 0 && c->C > 0 && c->K > 0 && c->R > 0 && c->S > 0 && c->stride > 0 && c->pad >= 0,
                   "%s: non-positive dimension", who);
    CSLGAN_REQUIRE(c->R * c->S <= IG_MAX_TAPS, "%s: %dx%d filter has more than %d taps", who, c->R, c->S, IG_MAX_TAPS);
    CSLGAN_REQUIRE(c->compute >= CSLGAN_COMPUTE_F32 && c->compute <= CSLGAN_COMPUTE_BF16X3, "%s: unknown cslgan_conv_t.compute %d", who, c->compute);
    const int VH = c->H, VW = c->W;
    const int P = (VH + 2 * c->pad - c->R) / c->stride + 1, Q = (VW + 2 * c->pad - c->S) / c->stride + 1;
    CSLGAN_REQUIRE(P == c->P && Q == c->Q, "%s: output %dx%d does not match P,Q=%d,%d", who, P, Q, c->P, c->Q);
    CSLGAN_REQUIRE((long long)c->N * c->P * c->Q * c->K < (1ll << 31) && (long long)c->N * VH * VW * c->C < (1ll << 31) &&
                   (long long)c->K * c->R * c->S * c->C < (1ll << 31), "%s: tensor too large for 32-bit offsets", who);
    return CSLGAN_OK;
}

static void clear_taps(KcClass& k) {
    for (int t = 0; t < IG_MAX_TAPS; ++t) { k.ty[t] = 0; k.tx[t] = 0; }
}

}  // namespace cslgan

using namespace cslgan;

extern "C" {

static int conv2d_fwd_impl(const cslgan_conv_t* c, const float* x, const float* w, const void* w3, const float* bias,
                           const float* residual, int act, float* y, void* stream, int x_bf16 = 0);

// cslgan_conv2d_fwd_f32 for a conv with 1..4 output channels (the generator's output conv, DCResNet_models.py:85) whose INPUT is
// stored as bfloat16 (bf16 storage mode): fp32 filter, fp32 arithmetic, fp32 output; only shapes the vector-ALU kernel takes.
int cslgan_conv2d_fwd_skinny_bf16in(const cslgan_conv_t* c, const void* x_bf16, const float* w, const float* bias, int act, float* y, void* stream) {
    return conv2d_fwd_impl(c, reinterpret_cast<const float*>(x_bf16), w, nullptr, bias, nullptr, act, y, stream, 1);
}

int cslgan_conv2d_fwd_f32(const cslgan_conv_t* c, const float* x, const float* w, const float* bias,
                          const float* residual, int act, float* y, void* stream) {
    return conv2d_fwd_impl(c, x, w, nullptr, bias, residual, act, y, stream);
}

int cslgan_split_filter_x3_f32(const float* w, int rows, int taps, int red, void* w3_ws, int pieces, void* stream) {
    CSLGAN_REQUIRE(w && w3_ws && rows > 0 && taps > 0 && red > 0 && (pieces == 0 || pieces == 1 || pieces == 3), "split_filter_x3: bad argument");
    CSLGAN_REQUIRE(aligned16(w) && aligned16(w3_ws) && ((long long)rows * taps * red) % 4 == 0, "split_filter_x3: filter must be 16-byte aligned with a multiple of 4 elements");
    return split_filter_x3(w, rows, taps, red, w3_ws, (hipStream_t)stream, pieces);
}

int cslgan_conv2d_fwd_x3_f32(const cslgan_conv_t* c, const float* x, const float* w, void* w3_ws, int repack, const float* bias,
                             const float* residual, int act, float* y, void* stream) {
    CSLGAN_REQUIRE(c && w && w3_ws, "conv2d_fwd_x3: null argument");
    CSLGAN_REQUIRE(aligned16(w) && aligned16(w3_ws) && ((long long)c->K * c->R * c->S * c->C) % 4 == 0, "conv2d_fwd_x3: filter must be 16-byte aligned with a multiple of 4 elements");
    if (repack) {       // CSLGAN_COMPUTE_F32: a step-major fp32 copy (the exact-fp32 form of the LDS-halo kernel, K*R*S*C floats)
        int rc = split_filter_x3(w, c->K, c->R * c->S, c->C, w3_ws, (hipStream_t)stream,
                                 c->compute == CSLGAN_COMPUTE_BF16X3 ? 3 : (c->compute == CSLGAN_COMPUTE_BF16 ? 1 : 0));
        if (rc) return rc;
    }
    return conv2d_fwd_impl(c, x, w, w3_ws, bias, residual, act, y, stream);
}

static int conv2d_fwd_impl(const cslgan_conv_t* c, const float* x, const float* w, const void* w3, const float* bias,
                           const float* residual, int act, float* y, void* stream, int x_bf16) {
    CSLGAN_REQUIRE(c && x && w && y, "conv2d_fwd: null argument");
    int rc = check_conv(c, "conv2d_fwd");
    if (rc) return rc;
    CSLGAN_REQUIRE(act >= 0 && act <= 3, "conv2d_fwd: unknown activation %d", act);
    static const int c3_env = [] { const char* e = getenv("CSLGAN_C3"); return e ? atoi(e) : 1; }();
    if (c->in_scale) {      // the input affine map is applied by the halo kernel's and the 1..4-output kernel's staging only
        CSLGAN_REQUIRE(c->in_shift && !x_bf16 && c->stride == 1 && c->R * c->S > 1 && aligned16(c->in_scale) && aligned16(c->in_shift),
                       "conv2d_fwd: in_scale needs in_shift, stride 1, a filter larger than 1x1 and 16-byte aligned tables");
    }
    if (c->gn_part) {       // GroupNorm partials come from the halo kernel's epilogue only (cslgan_conv_t.gn_part)
        const int cpg = c->gn_groups > 0 && c->K % c->gn_groups == 0 ? c->K / c->gn_groups : 0;
        CSLGAN_REQUIRE(w3 && !x_bf16 && c->stride == 1 && act == CSLGAN_ACT_NONE && c->P % 8 == 0 && c->Q % 8 == 0 && cpg >= 1 && cpg <= 32 &&
                       (cpg & (cpg - 1)) == 0 && c->R * c->S > 1 && (long long)c->P * c->Q / 64 <= CSLGAN_NORM_PARTIAL_BLOCKS,
                       "conv2d_fwd: gn_part needs cslgan_conv2d_fwd_x3_f32, stride 1, an 8x8-patchable output of <= 4096 pixels, no activation and K / gn_groups a power of two <= 32");
    }
    if (x_bf16) {
        CSLGAN_REQUIRE(c->K <= 4 && c->C == 64 && c->stride == 1 && !residual, "conv2d_fwd_skinny_bf16in: needs 1..4 output channels, 64 input channels, stride 1");
    } else
    if (c3_env && c3_fwd_eligible(c, residual))          // the critic's RGB first layer (conv_c3.hip), exact fp32 in every compute mode
        return launch_c3_fwd(c, x, w, bias, act, y, (hipStream_t)stream);
    if (conv1x1_eligible(c, x, w, residual))                     // the generator's shortcut convs (conv1x1.hip)
        return launch_conv1x1(c, x, w, bias, act, y, (hipStream_t)stream);
    if (linear_k1_shape(c) && aligned16(x) && aligned16(w))      // one output unit: a dot product per row (linear_k1.hip), fp32 in every mode
        return launch_linear_k1_fwd(c, x, w, bias, residual, act, y, (hipStream_t)stream);
    KcParams p{};
    p.a = x; p.AH = c->H; p.AW = c->W; p.AC = c->C;
    p.VH = c->H; p.VW = c->W;
    p.sy = p.sx = c->stride;
    p.w = w; p.w3 = w3; p.Nn = c->K; p.out = y; p.OHf = c->P; p.OWf = c->Q; p.osy = p.osx = 1; p.ldo = c->K; p.dense_out = 1;
    p.bias = bias; p.res = residual; p.mask = nullptr; p.act = act; p.bf16 = c->compute == CSLGAN_COMPUTE_BF16 ? 1 : (c->compute == CSLGAN_COMPUTE_BF16X3 ? 3 : 0);
    p.a_bf16 = x_bf16;
    p.part = reinterpret_cast<float*>(c->split_ws); p.part_floats = c->split_ws_floats; p.out_floats = (long long)c->N * c->P * c->Q * c->K;
    if (c->gn_part) { p.gn_part = c->gn_part; p.gn_cpg = c->K / c->gn_groups; p.gn_slots = c->P * c->Q / 64; p.part = nullptr; }
    if (c->in_scale) { p.in_scale = c->in_scale; p.in_shift = c->in_shift; p.in_relu = c->in_relu; p.part = nullptr; }
    p.n_cls = 1;
    KcClass& k = p.cls[0];
    k.M = c->N * c->P * c->Q; k.OHc = c->P; k.OWc = c->Q; k.T = c->R * c->S; k.Kdim = k.T * c->C; k.w_off = 0; k.oy0 = k.ox0 = 0;
    clear_taps(k);
    for (int kh = 0; kh < c->R; ++kh)
        for (int kw = 0; kw < c->S; ++kw) { k.ty[kh * c->S + kw] = (signed char)(kh - c->pad); k.tx[kh * c->S + kw] = (signed char)(kw - c->pad); }
    return launch_kc(p, (hipStream_t)stream, (long long)c->N * c->P * c->Q * c->K);
}

// Stride-2 forward conv as four stride-1 convs over the parity sub-images of x (class (a,b) = input pixels (2i+a, 2j+b)),
// all accumulating into the same output tile inside one igemm_halo workgroup.  igemm_kc re-gathers the 2-strided input
// window once per tap (the critic's forward convs ran at 82-105 TF); per class the window is a <= 10x10 halo in LDS.
// Falls back to cslgan_conv2d_fwd_f32 for shapes the halo kernel does not take.
static int conv2d_s2_fwd_impl(const cslgan_conv_t* c, const float* x, const float* w, float* wcls_ws, void* w3_ws, int repack,
                              const float* bias, int act, float* y, void* stream);

int cslgan_conv2d_s2_fwd_f32(const cslgan_conv_t* c, const float* x, const float* w, float* wcls_ws, int repack,
                             const float* bias, int act, float* y, void* stream) {
    return conv2d_s2_fwd_impl(c, x, w, wcls_ws, nullptr, repack, bias, act, y, stream);
}

// The same stride-2 forward conv with fp32 emulated from three bfloat16 pieces (cslgan_conv_t.compute = CSLGAN_COMPUTE_BF16X3) or
// plain bfloat16 operands (CSLGAN_COMPUTE_BF16) on the LDS-halo kernel of csrc/igemm_x3.hip: w3_ws receives the class matrices
// split into their pieces in step-major order (3 * K*R*S*C bfloat16; rewritten when repack != 0).  Shapes that kernel does not take
// run cslgan_conv2d_fwd_f32 in the same arithmetic.
int cslgan_conv2d_s2_fwd_x3_f32(const cslgan_conv_t* c, const float* x, const float* w, float* wcls_ws, void* w3_ws, int repack,
                                const float* bias, int act, float* y, void* stream) {
    CSLGAN_REQUIRE(c && w3_ws, "conv2d_s2_fwd_x3: null argument");
    return conv2d_s2_fwd_impl(c, x, w, wcls_ws, w3_ws, repack, bias, act, y, stream);
}

static int conv2d_s2_fwd_impl(const cslgan_conv_t* c, const float* x, const float* w, float* wcls_ws, void* w3_ws, int repack,
                              const float* bias, int act, float* y, void* stream) {
    CSLGAN_REQUIRE(c && x && w && wcls_ws && y, "conv2d_s2_fwd: null argument");
    int rc = check_conv(c, "conv2d_s2_fwd");
    if (rc) return rc;
    CSLGAN_REQUIRE(c->stride == 2 && c->R == c->S, "conv2d_s2_fwd: needs stride 2 and a square filter");
    CSLGAN_REQUIRE(act >= 0 && act <= 3, "conv2d_s2_fwd: unknown activation %d", act);
    hipStream_t st = (hipStream_t)stream;
    const int R = c->R, pad = c->pad;
    auto fl = [](int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); };
    RepackArgs ra{};
    ra.K = c->K; ra.R = R; ra.S = R; ra.C = c->C; ra.transposed = 0;
    KcParams p{};
    p.a = x; p.AH = c->H; p.AW = c->W; p.AC = c->C; p.VH = c->H; p.VW = c->W; p.sy = p.sx = 1;
    p.w = wcls_ws; p.Nn = c->K; p.out = y; p.OHf = c->P; p.OWf = c->Q; p.osy = p.osx = 1; p.ldo = c->K; p.dense_out = 0;
    p.bias = bias; p.res = nullptr; p.mask = nullptr; p.act = act; p.acc_classes = 1;
    p.part = reinterpret_cast<float*>(c->split_ws); p.part_floats = c->split_ws_floats; p.out_floats = (long long)c->N * c->P * c->Q * c->K;
    int n = 0, off = 0;
    bool ok = R * R <= IG_MAX_TAPS;
    for (int a = 0; ok && a < 2; ++a)
        for (int b = 0; ok && b < 2; ++b) {
            KcClass& k = p.cls[n];
            clear_taps(k);
            int T = 0;
            for (int kh = 0; kh < R; ++kh)
                for (int kw = 0; kw < R; ++kw) {
                    const int du = fl(kh - pad), dv = fl(kw - pad);
                    if ((kh - pad) - 2 * du != a || (kw - pad) - 2 * dv != b) continue;
                    k.ty[T] = (signed char)du; k.tx[T] = (signed char)dv;
                    ra.kh_lo[n][T] = (signed char)kh; ra.kh_hi[n][T] = (signed char)(kh + 1);
                    ra.kw_lo[n][T] = (signed char)kw; ra.kw_hi[n][T] = (signed char)(kw + 1);
                    ++T;
                }
            if (T == 0) continue;
            k.M = c->N * c->P * c->Q; k.OHc = c->P; k.OWc = c->Q; k.T = T; k.Kdim = T * c->C; k.w_off = off; k.oy0 = k.ox0 = 0;
            k.ay_mul = 2; k.ay_off = a; k.ax_mul = 2; k.ax_off = b;
            ra.cls_T[n] = T; ra.cls_off[n] = off; off += T * c->K * c->C;
            ++n;
        }
    p.n_cls = n; ra.n_class = n;
    static const int halo_env = [] { const char* e = getenv("CSLGAN_KC_HALO"); return e ? atoi(e) : 1; }();
    static const int s2_env = [] { const char* e = getenv("CSLGAN_S2_HALO"); return e ? atoi(e) : 1; }();
    // Measured on the critic (B=128 and the fused 384 rows, scripts/shape_times.py): 92 -> 109 TF on the 768-tile conv2
    // launch, no gain or a loss below ~500 tiles (four halo stagings per chunk for 2-9 taps each) -> igemm_kc keeps those.
    static const int s2_min_tiles = [] { const char* e = getenv("CSLGAN_S2_MIN_TILES"); return e ? atoi(e) : 512; }();
    const long long wide_tiles = ((long long)c->N * c->P * c->Q + 127) / 128 * ((c->K + 127) / 128);
    if (w3_ws) {        // the round-4 LDS-halo kernel: three-piece, plain bf16 or exact fp32 operands, any tile count
        p.bf16 = c->compute == CSLGAN_COMPUTE_BF16X3 ? 3 : (c->compute == CSLGAN_COMPUTE_BF16 ? 1 : 0);
        p.w3 = w3_ws;
        p.a_bytes = 0; p.w_bytes = 0;      // (launch_kc sets the operand sizes)
    }
    // The repack comes BEFORE any fallback: the caller caches "this workspace is current for this weight version" after a call with
    // repack != 0, and whether THIS call takes the halo kernel depends on its batch size (tile-count threshold, the 4x4-grid row
    // minimum).  A small-batch call that fell back without packing used to leave the workspace unwritten behind a fresh cache entry,
    // and a larger-batch call of the same step (the fused 384-row pass after the 128-row penalty branch, at batch sizes where only
    // one of them clears the threshold) then read it with repack = 0.
    if (repack && ok && n > 0) {
        const long long per = (long long)c->C * 9 * c->K;
        unsigned gxn = (unsigned)((per + 255) / 256);
        gxn = gxn > 1024 ? 1024 : (gxn < 1 ? 1 : gxn);
        if (w3_ws && c->C % 16 == 0 && aligned16(w3_ws)) { ra.pieces = p.bf16 == 3 ? 3 : (p.bf16 ? 1 : 4); ra.w3 = reinterpret_cast<unsigned short*>(w3_ws); }
        hipLaunchKernelGGL(repack_filters_kernel, dim3(gxn, (unsigned)n), dim3(256), 0, st, w, wcls_ws, ra);
        rc = check_launch("repack_filters_kernel");
        if (rc) return rc;
    }
    if (w3_ws) {
        if (!ok || n == 0 || !x3h_eligible(p))
            return cslgan_conv2d_fwd_f32(c, x, w, bias, nullptr, act, y, stream);
    } else if (!ok || n == 0 || !halo_env || !s2_env || wide_tiles < s2_min_tiles || c->compute != CSLGAN_COMPUTE_F32 || !halo_eligible(p))
        return cslgan_conv2d_fwd_f32(c, x, w, bias, nullptr, act, y, stream);
    return launch_kc(p, st, 0);
}

static int conv2d_dgrad_impl(const cslgan_conv_t* c, const float* gy, const float* w, float* wt_ws, int repack,
                             const float* mask, float* gx, void* stream, int gy_bf16, void* w3_ws = nullptr);

int cslgan_conv2d_dgrad_f32(const cslgan_conv_t* c, const float* gy, const float* w, float* wt_ws, int repack,
                            const float* mask, float* gx, void* stream) {
    return conv2d_dgrad_impl(c, gy, w, wt_ws, repack, mask, gx, stream, 0);
}

// cslgan_conv2d_dgrad_f32 with fp32 emulated from three bfloat16 pieces (CSLGAN_COMPUTE_BF16X3) or plain bfloat16 operands
// (CSLGAN_COMPUTE_BF16) on the LDS-halo kernel of csrc/igemm_x3.hip: w3_ws receives the repacked class matrices split into their
// pieces in step-major order (3 * K*R*S*C bfloat16; rewritten when repack != 0).  Shapes that kernel does not take run the gather
// kernels in the same arithmetic.
int cslgan_conv2d_dgrad_x3_f32(const cslgan_conv_t* c, const float* gy, const float* w, float* wt_ws, void* w3_ws, int repack,
                               const float* mask, float* gx, void* stream) {
    CSLGAN_REQUIRE(c && w3_ws, "conv2d_dgrad_x3: null argument");
    return conv2d_dgrad_impl(c, gy, w, wt_ws, repack, mask, gx, stream, 0, w3_ws);
}

// cslgan_conv2d_dgrad_f32 for a conv with 1..4 INPUT channels (the critic's RGB first layer: the gradient of the image,
// gradient_penalty.py:48-50) whose output gradient gy is stored as bfloat16: fp32 filter classes, fp32 arithmetic, fp32 gx.
int cslgan_conv2d_dgrad_skinny_bf16in(const cslgan_conv_t* c, const void* gy_bf16, const float* w, float* wt_ws, int repack, float* gx, void* stream) {
    CSLGAN_REQUIRE(c && c->C <= 4 && c->K == 64, "conv2d_dgrad_skinny_bf16in: needs 1..4 input channels and 64 output channels");
    return conv2d_dgrad_impl(c, reinterpret_cast<const float*>(gy_bf16), w, wt_ws, repack, nullptr, gx, stream, 1);
}

static int conv2d_dgrad_impl(const cslgan_conv_t* c, const float* gy, const float* w, float* wt_ws, int repack,
                             const float* mask, float* gx, void* stream, int gy_bf16, void* w3_ws) {
    CSLGAN_REQUIRE(c && gy && w && wt_ws && gx, "conv2d_dgrad: null argument");
    int rc = check_conv(c, "conv2d_dgrad");
    if (rc) return rc;
    CSLGAN_REQUIRE(c->stride >= 1 && c->stride <= 2, "conv2d_dgrad: stride %d unsupported", c->stride);
    if (!gy_bf16 && linear_k1_shape(c) && aligned16(w) && aligned16(gx) && (!mask || aligned16(mask)))      // gx[n,:] = gy[n] * w: no repack needed
        return launch_linear_k1_dgrad(c, gy, w, mask, gx, (hipStream_t)stream);
    const int s = c->stride;
    hipStream_t st = (hipStream_t)stream;
    RepackArgs ra{};
    ra.K = c->K; ra.R = c->R; ra.S = c->S; ra.C = c->C; ra.n_class = s * s; ra.transposed = 1;
    KcParams p{};
    p.a = gy; p.AH = c->P; p.AW = c->Q; p.AC = c->K; p.VH = c->P; p.VW = c->Q; p.sy = p.sx = 1;
    p.w = wt_ws; p.Nn = c->C; p.out = gx; p.OHf = c->H; p.OWf = c->W; p.osy = p.osx = s; p.ldo = c->C;
    p.dense_out = (s == 1) ? 1 : 0;
    p.bias = nullptr; p.res = nullptr; p.mask = mask; p.act = CSLGAN_ACT_NONE; p.bf16 = c->compute == CSLGAN_COMPUTE_BF16 ? 1 : (c->compute == CSLGAN_COMPUTE_BF16X3 ? 3 : 0);
    p.a_bf16 = gy_bf16;
    p.w3 = (w3_ws && c->K % 16 == 0 && aligned16(w3_ws)) ? w3_ws : nullptr;
    p.part = reinterpret_cast<float*>(c->split_ws); p.part_floats = c->split_ws_floats; p.out_floats = (long long)c->N * c->H * c->W * c->C;
    int off = 0, ncls = 0;
    for (int py = 0; py < s; ++py)
        for (int px = 0; px < s; ++px) {
            const int OHc = (c->H - py + s - 1) / s, OWc = (c->W - px + s - 1) / s;
            if (OHc <= 0 || OWc <= 0) continue;
            const int cls = ncls++;
            KcClass& k = p.cls[cls];
            clear_taps(k);
            int T = 0;
            for (int kh = 0; kh < c->R; ++kh) {
                if (((py + c->pad - kh) % s + s) % s != 0) continue;
                for (int kw = 0; kw < c->S; ++kw) {
                    if (((px + c->pad - kw) % s + s) % s != 0) continue;
                    ra.kh_lo[cls][T] = (signed char)kh; ra.kh_hi[cls][T] = (signed char)(kh + 1);
                    ra.kw_lo[cls][T] = (signed char)kw; ra.kw_hi[cls][T] = (signed char)(kw + 1);
                    k.ty[T] = (signed char)((py + c->pad - kh) / s);
                    k.tx[T] = (signed char)((px + c->pad - kw) / s);
                    ++T;
                }
            }
            CSLGAN_REQUIRE(T > 0, "conv2d_dgrad: a parity class has no taps (filter smaller than stride)");
            k.M = c->N * OHc * OWc; k.OHc = OHc; k.OWc = OWc; k.T = T; k.Kdim = T * c->K; k.w_off = off; k.oy0 = py; k.ox0 = px;
            ra.cls_T[cls] = T; ra.cls_off[cls] = off; off += T * c->K * c->C;
        }
    p.n_cls = ncls; ra.n_class = ncls;
    if (repack) {
        unsigned gxn = (unsigned)(((long long)c->K * c->C * c->R * c->S / (s * s) + 255) / 256);
        gxn = gxn > 1024 ? 1024 : (gxn < 1 ? 1 : gxn);
        if (p.w3) { ra.pieces = p.bf16 == 3 ? 3 : (p.bf16 ? 1 : 4); ra.w3 = reinterpret_cast<unsigned short*>(const_cast<void*>(p.w3)); }
        hipLaunchKernelGGL(repack_filters_kernel, dim3(gxn, (unsigned)ncls), dim3(256), 0, st, w, wt_ws, ra);
        rc = check_launch("repack_filters_kernel");
        if (rc) return rc;
    }
    // K can be split only for the dense (stride-1, single-class) form whose output we may zero here
    return launch_kc(p, st, (s == 1 && !mask) ? (long long)c->N * c->H * c->W * c->C : 0);
}

}  // extern "C"
