// fp32-accurate convolution on the bf16 matrix cores with an LDS-resident input halo ("x3 halo"): forward conv, stride-2
// forward conv as accumulated parity classes, and the parity classes of the data gradient.  gfx950 only.
//
//   Out[m][n] = epilogue( sum_k A(m,k) * Wm[n][k] )      (index maps and classes: igemm.h, same as igemm_halo.hip)
//
// Arithmetic (csrc/igemm_bf16.hip states the construction): every fp32 operand is the sum of three bfloat16 pieces
// hi + mid + lo (3 x 8 mantissa bits); a product is six exact bf16 x bf16 piece products (the three smallest of the nine are
// below 2^-23 |a||b| and dropped), summed smallest first in the fp32 accumulator of v_mfma_f32_32x32x16_bf16: six MFMAs per
// 16-k step at 16x the fp32 MFMA rate = 2.67x the exact-fp32 matrix rate at fp32 accuracy (tests: error against fp64 is
// below the exact-fp32 kernels').  NP = 1 is the plain bf16 form of the same kernel (--compute_dtype bf16 on fp32 tensors).
//
// Round 4 rewrite of igemm_halo_x3_kernel (round 2-3, in igemm_bf16.hip).  What rocprofv3 and the ISA showed there: matrix
// pipe 57 % busy.  Cause found in the ISA: the per-lane halo base `a_base[odd][i]` was indexed by a run-time tap parity, which
// put the array in SCRATCH — every K step did ds_read (tap table) -> s_waitcnt -> scratch_load -> s_waitcnt vmcnt(0) -> LDS
// fragment reads, and that vmcnt(0) also waited for the filter loads issued a few instructions earlier (an L2 round trip in
// front of every 24-MFMA block); the fragment reads were ds_read2_b64 (half rate: the LDS array was typed 8-byte aligned).
// Here:
//   * tap offsets are SCALAR arithmetic on an affine tap grid (row-major R' x S' taps with constant steps — every forward
//     filter and every parity class of a strided data gradient is one); nothing about a tap is looked up in memory;
//   * the swizzled fragment address is (base + tap) ^ parity: one v_add + v_xor, no table;
//   * filter slices ride a two-slot register ring that is refilled (with the slice of step s+2) right after the MFMAs of step s
//     have read it, and the A fragments of the next tap are read into the SAME registers tile by tile as each tile's MFMAs
//     release them — no write-after-read copies, no second register set (189 VGPRs for the 128x128 three-piece tile);
//   * the steady-state K loop has no branch inside or between its steps, so the compiler's counted s_waitcnt survive
//     (any skipped path made SIInsertWaitcnts fall back to vmcnt(0) / lgkmcnt(0) in front of every MFMA block);
//   * the halo image is double buffered: the next chunk's pixels are fetched at the chunk's first tap, split and written to
//     the OTHER image at its last tap, one barrier per 16-channel chunk (was: two, with the split in between);
//   * LDS images are 16-byte typed: fragment reads are ds_read_b128.
// GEN = true adds what igemm_halo<.,true> has: several classes per launch (the 9/6/6/4-tap parity classes of a 5x5 stride-2
// data gradient), class pairs, classes accumulated into one output (a stride-2 forward conv as four stride-1 convs over
// the parity sub-images of x) and four-image patches for 4x4 grids.
//
// Replaces (reference file:line): nn.Conv2d forward and its autograd data gradient, DCResNet_models.py:16,60-70,95-104,
// 131-132; gradient_penalty.py:48-54 (the double backward runs the same two ops).
#include <stdlib.h>
#include "common.h"
#include "igemm.h"

namespace cslgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned XOOB = 0xFFFFFFF0u;
constexpr unsigned XFAR = 0xFFFF0000u;  // an out-of-range byte offset that STAYS out of range when a chunk / step offset (< 64 KB) is added to it
constexpr int XH_MAX = 160;            // pixels of a patch's LDS image: 12 rows x 12 (8+4 squared: up to 5x5 taps), or four 6x6 halos at a
                                       // pitch of 40 (quad patches)
// LDS pitches are FIXED, whatever the halo's real width: the 16-lane groups of ds_read_b128 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31})
// touch four patch rows at once, and with 32-byte pixels only row pitches of 8 / 12 / 16 pixels (quad patches: rows of 6, images 40
// apart) keep their 16-byte reads on distinct banks.  A 3x3-tap class (halo 10 wide) indexed with ITS width ran 2-way conflicted on
// half the reads (rocprofv3: SQ_LDS_BANK_CONFLICT = 45 % of SQ_LDS_IDX_ACTIVE on the critic's launches, 0 on the generator's 12-wide halos).
constexpr int XROW = 12, XQROW = 6, XQIMG = 40;

__device__ __forceinline__ unsigned xpack(float lo, float hi) {      // v_cvt_pk_bf16_f32: RNE, lo in bits 0..15
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float xlo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float xhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
struct x3_t { uint2 hi, mid, lo; };
__device__ __forceinline__ x3_t xsplit4(const float4& v) {
    x3_t r;
    r.hi = make_uint2(xpack(v.x, v.y), xpack(v.z, v.w));
    const float r0 = v.x - xlo(r.hi.x), r1 = v.y - xhi(r.hi.x), r2 = v.z - xlo(r.hi.y), r3 = v.w - xhi(r.hi.y);   // exact
    r.mid = make_uint2(xpack(r0, r1), xpack(r2, r3));
    r.lo = make_uint2(xpack(r0 - xlo(r.mid.x), r1 - xhi(r.mid.x)), xpack(r2 - xlo(r.mid.y), r3 - xhi(r.mid.y)));
    return r;
}
__device__ __forceinline__ float4 xload4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

}  // namespace

// One workgroup = two 64-row patches (128 rows) x BN output channels; waves 2 (M: one patch each) x 2 (N).
// LDS: halo images [buffer][piece][patch][pixel][2 x 16 B], 2 x NP x 9216 B (55 KB for NP = 3: two workgroups per CU).
// AFF (single-class launches): the staged operand is max(in_scale[img][c] * a + in_shift[img][c], in_relu ? 0 : -inf) — GroupNorm
// (+ ReLU) of the producing layer applied between the global load and the split into pieces (cslgan_conv_t.in_scale); padding
// pixels stay zero.  A template parameter, not a flag: a conditional table load inside the chunk's fetch region would make
// SIInsertWaitcnts merge the counted waits of the loop conservatively (the lesson of the loop below).
template <int BN, int NP, bool GEN, bool AFF = false>
__global__ __launch_bounds__(256, 2) void igemm_x3h_kernel(const KcParams p) {
    constexpr int BM = 128, TM = 2, TN = BN / 64;
    // NP = 0: EXACT fp32 (round 4, second half): the same loop on v_mfma_f32_32x32x2_f32 — fp32 halo pixels (16 channels = 64 B) in ONE
    // image per buffer, the filter in step-major fp32 order [step][n][16], eight fp32 MFMAs per tile and 16-channel step.
    constexpr bool F32 = NP == 0;
    constexpr int NI = F32 ? 1 : NP;                           // LDS images per buffer
    constexpr int NO = F32 ? 2 : NP;                           // 16-byte operand registers per tile and step (fp32: the two 8-channel halves)
    constexpr int PXU = F32 ? 4 : 2;                           // uint4 units per halo pixel
    constexpr int IMG = 2 * XH_MAX * PXU;                      // uint4 units per image (2 patches)
    __shared__ __attribute__((aligned(16))) uint4 Hs[2][NI][IMG];
    __shared__ int s_off[BM];
    __shared__ int s_roff[BM];

    const int tid = threadIdx.x;
    const int nwg = p.tiles_m * p.tiles_n;
    // channel split (launch_x3h: launches of < 512 workgroups): csplit workgroups per tile, each on its own range of 16-channel
    // chunks, partial sums to p.part[split]; the splits of one tile are neighbours (same XCD: they share the tile's halo in L2)
    const int cs = p.csplit > 1 ? p.csplit : 1;
    const int wgs = xcd_remap(blockIdx.x, nwg * cs);
    const int split = cs > 1 ? wgs % cs : 0;
    const int wg = cs > 1 ? wgs / cs : wgs;
    const int tile_mg = wg / p.tiles_n, tile_n = wg - tile_mg * p.tiles_n;
    const bool accumulate = GEN && p.acc_classes;
    const int n_sub = accumulate ? p.n_cls : ((GEN && p.pair_mode) ? 2 : 1);
    const int lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;                     // wm = patch index
    const int n0 = tile_n * BN;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);

    f32x16 acc[TM][TN];
#pragma unroll 1
    for (int sub = 0; sub < n_sub; ++sub) {
        int ci = 0, tile_in_cls;
        if (accumulate) {
            ci = sub;
            tile_in_cls = tile_mg;
        } else if (GEN && p.pair_mode) {
            const int pg = tile_mg / p.tiles_per_cls;
            ci = p.pair_cls[pg][sub];
            tile_in_cls = tile_mg - pg * p.tiles_per_cls;
        } else {
            if (GEN) {
#pragma unroll 1
                while (ci + 1 < p.n_cls && tile_mg >= p.cls[ci + 1].tile0) ++ci;
            }
            tile_in_cls = tile_mg - p.cls[ci].tile0;
        }
        const KcClass& kc = p.cls[ci];
        const int M = kc.M, OHc = kc.OHc, OWc = kc.OWc, T = kc.T;
        const int m0 = tile_in_cls * BM;
        const int HW_ = kc.halo_w, HH_ = kc.halo_h;
        const bool quad = GEN && kc.patch == 2;                // 4x4 grids: a 64-row patch = four consecutive images
        const int hpix_img = HH_ * HW_;
        const int hpix = quad ? 4 * hpix_img : hpix_img;
        const int img_stride = p.AH * p.AW * p.AC;
        const int ay_mul = (GEN && kc.ay_mul) ? kc.ay_mul : 1, ay_off = GEN ? kc.ay_off : 0;
        const int ax_mul = (GEN && kc.ax_mul) ? kc.ax_mul : 1, ax_off = GEN ? kc.ax_off : 0;
        // affine tap grid (checked on the host): tap t = (i, j) = (t / nkw, t % nkw), halo offset (dy0 + i*ystep, dx0 + j*xstep)
        const int nkw = kc.nkw;
        const int ystep = T > nkw ? (int)kc.ty[nkw] - (int)kc.ty[0] : 0;
        const int xstep = nkw > 1 ? (int)kc.tx[1] - (int)kc.tx[0] : 0;
        const int dy0 = (int)kc.ty[0] - kc.ty_min, dx0 = (int)kc.tx[0] - kc.tx_min;

        // ---- halo staging plan: 2 patches x hpix pixels x 4 groups of 4 channels; <= 2*144*4/256 = 4.5 float4 per thread.
        // Global byte offset (chunk 0) and LDS slot of each element are fixed for the whole class: computed once.
        constexpr int HREG = (2 * XH_MAX * 4 + 255) / 256;
        unsigned h_goff[HREG];
        int h_lds[HREG];                                       // uint2 index into a piece image, -1 = nothing to write
        unsigned pp_mask = 0;                                  // AFF: bit j = the patch of staging slot j
        int aff_off[2] = {0, 0};                               // AFF: float index of this thread's 4 channels in the patch's image row of the tables
        {
            int p_img[2], p_y0[2], p_x0[2];
            bool p_ok[2];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int m = m0 + 64 * pp;
                p_ok[pp] = m < M;
                const RowCoord rc = kc_decode_row(p_ok[pp] ? m : 0, OHc, OWc, quad ? 0 : 1);   // first row of the patch = its top-left pixel
                p_img[pp] = rc.img * img_stride;
                if (AFF) aff_off[pp] = rc.img * p.AC + (tid & 3) * 4;
                p_y0[pp] = rc.oy + kc.ty_min;
                p_x0[pp] = rc.ox + kc.tx_min;
            }
            const int h_total = 2 * hpix * 4;
#pragma unroll
            for (int j = 0; j < HREG; ++j) {
                const int idx = tid + 256 * j;
                const int ch = idx & 3, pixg = idx >> 2;
                const int pp = pixg >= hpix ? 1 : 0;
                pp_mask |= (unsigned)pp << j;
                const int pix = pixg - pp * hpix;
                const int si = quad ? pix / hpix_img : 0;      // sub-image of a quad patch
                const int rem = pix - si * hpix_img;
                const int hy = rem / HW_, hx = rem - hy * HW_;
                const int iy = (p_y0[pp] + hy) * ay_mul + ay_off, ix = (p_x0[pp] + hx) * ax_mul + ax_off;
                const bool in_img = idx < h_total && p_ok[pp] && (unsigned)iy < (unsigned)p.VH && (unsigned)ix < (unsigned)p.VW;
                h_goff[j] = in_img ? 4u * (unsigned)(p_img[pp] + si * img_stride + (iy * p.AW + ix) * p.AC + ch * 4) : XFAR;
                // the two 16-byte halves of a pixel are swapped on odd halo rows: the patch rows a 16-lane ds_read_b128 group covers then
                // hit disjoint banks (the plain 32-byte pixel stride is 2-way conflicted)
                const int lpx = quad ? si * XQIMG + hy * XQROW + hx : hy * XROW + hx;
                // bf16 pieces: uint2 slot, the two 16-byte halves of a pixel swapped on odd halo rows; fp32: uint4 slot, the four 16-byte
                // chunks of a pixel rotated by (hx + hy) & 3 — both keep the 16-lane groups of ds_read_b128 on distinct banks (bank model in
                // the file header / DESIGN §4.14)
                h_lds[j] = idx >= h_total ? -1 : (F32 ? (pp * XH_MAX + lpx) * 4 + (ch ^ ((hx + hy) & 3)) : (pp * XH_MAX + lpx) * 4 + (ch ^ ((hy & 1) << 1)));
            }
        }
        const int n_cc_all = p.AC >> 4;
        const int cc_lo = cs > 1 ? split * n_cc_all / cs : 0;
        const int n_cc = cs > 1 ? (split + 1) * n_cc_all / cs - cc_lo : n_cc_all;      // this workgroup's chunks: cc_lo .. cc_lo + n_cc - 1
        const float lo = (AFF && !p.in_relu) ? -__builtin_inff() : 0.f;
        float4 rh[HREG];
        // AFF: scale / shift of the thread's 4 channels, per patch, for the chunk in flight.  Four named registers and selects: as
        // arrays picked by the slot's patch bit they were "promoted" to LDS (dynamic index), with an lgkmcnt(0) in every commit
        float4 sc0, sc1, sh0, sh1;
        // buffer descriptors made once: with plain pointers the compiler re-read p.in_scale / p.in_shift from the kernel arguments
        // inside the loop (s_load: lgkmcnt) and every such read drained the LDS fragment reads in flight (+10 % kernel time)
        const __amdgpu_buffer_rsrc_t sc_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(AFF ? p.in_scale : p.a), 0, 0xFFFFFFF0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t sh_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(AFF ? p.in_shift : p.a), 0, 0xFFFFFFF0u, 0x00020000);
        auto fetch_halo = [&](int cc) {
            const unsigned co = 64u * (unsigned)(cc + cc_lo);  // 16 channels x 4 B per chunk
#pragma unroll
            for (int j = 0; j < HREG; ++j) rh[j] = xload4(a_rsrc, h_goff[j] + co);      // an invalid element sits at XFAR: + co (< 64 KB) stays out of range
            if (AFF) {
                sc0 = xload4(sc_rsrc, 4u * (unsigned)aff_off[0] + co); sh0 = xload4(sh_rsrc, 4u * (unsigned)aff_off[0] + co);
                sc1 = xload4(sc_rsrc, 4u * (unsigned)aff_off[1] + co); sh1 = xload4(sh_rsrc, 4u * (unsigned)aff_off[1] + co);
            }
        };
        auto commit_halo = [&](int buf) {
#pragma unroll
            for (int j = 0; j < HREG; ++j) {
                if (AFF) {
                    const bool second = (pp_mask >> j) & 1u;
                    // padding stays zero (it pads the NORMALISED activation): scale and shift are multiplied by 0 there — the loaded
                    // value of such a slot is the buffer's out-of-range 0, so max(0 * 0 + 0, lo) = 0; arithmetic, not a branch
                    const float m = h_goff[j] != XFAR ? 1.f : 0.f;
                    rh[j].x = fmaxf(fmaf((second ? sc1.x : sc0.x) * m, rh[j].x, (second ? sh1.x : sh0.x) * m), lo);
                    rh[j].y = fmaxf(fmaf((second ? sc1.y : sc0.y) * m, rh[j].y, (second ? sh1.y : sh0.y) * m), lo);
                    rh[j].z = fmaxf(fmaf((second ? sc1.z : sc0.z) * m, rh[j].z, (second ? sh1.z : sh0.z) * m), lo);
                    rh[j].w = fmaxf(fmaf((second ? sc1.w : sc0.w) * m, rh[j].w, (second ? sh1.w : sh0.w) * m), lo);
                }
                if (h_lds[j] >= 0) {
                    uint2* img0 = reinterpret_cast<uint2*>(&Hs[buf][0][0]);
                    if (F32) {
                        Hs[buf][0][h_lds[j]] = make_uint4(__float_as_uint(rh[j].x), __float_as_uint(rh[j].y), __float_as_uint(rh[j].z), __float_as_uint(rh[j].w));
                    } else if (NP == 1) {
                        img0[h_lds[j]] = make_uint2(xpack(rh[j].x, rh[j].y), xpack(rh[j].z, rh[j].w));
                    } else {
                        const x3_t t3 = xsplit4(rh[j]);
                        img0[h_lds[j]] = t3.hi;
                        reinterpret_cast<uint2*>(&Hs[buf][NI > 1 ? 1 : 0][0])[h_lds[j]] = t3.mid;
                        reinterpret_cast<uint2*>(&Hs[buf][NI > 2 ? 2 : 0][0])[h_lds[j]] = t3.lo;
                    }
                }
            }
        };

        // ---- filter operand: straight to registers from the filter PRE-SPLIT into bfloat16 pieces in step-major order
        // [piece][step = chunk*T + tap][n][16 k] (split_filter_x3, cached per weight version by the caller): lane (r, h), tile j,
        // piece c reads 8 consecutive k of filter row n0 + wn*TN*32 + j*32 + r = one 16-byte load; the 32 rows x 32 B a
        // wave-load touches are 1 KB contiguous.
        // fp32 (NP = 0): ONE step-major fp32 copy [step][n][16 floats]; operand e of lane (r, h) = channels 8e + 4h .. + 3 of filter row n.
        const unsigned piece_bytes = F32 ? 32u : 2u * (unsigned)p.Nn * (unsigned)kc.Kdim;     // distance between a tile's operand registers
        const __amdgpu_buffer_rsrc_t w3_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.w3) + (F32 ? 4ll : 2ll * NP) * kc.w_off), 0,
            F32 ? 4u * (unsigned)p.Nn * (unsigned)kc.Kdim : NP * piece_bytes, 0x00020000);
        const unsigned n_bytes = F32 ? 64u : 32u;                 // bytes of one filter row in a step's slice
        const unsigned step_bytes = n_bytes * (unsigned)p.Nn;
        unsigned b_off[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * TN * 32 + j * 32 + r;
            b_off[j] = n < p.Nn ? n_bytes * (unsigned)n + 16u * (unsigned)h : XOOB;       // tested again in load_b
        }
        const int n_steps = n_cc * T;
        const unsigned step0_bytes = (unsigned)(cc_lo * T) * step_bytes;
        u32x4 B0[NO][TN], B1[NO][TN];                          // two-slot ring of filter slices: slot s % 2 is refilled with step s + 2
        auto load_b = [&](int step, u32x4 (&dst)[NO][TN]) {    // right after the MFMAs of step s have read it (no third slot, no copies)
            const unsigned kb = (unsigned)step * step_bytes + step0_bytes;
            const unsigned past = step >= n_steps ? XOOB : 0u;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const unsigned bad = past | (b_off[j] == XOOB ? XOOB : 0u);
#pragma unroll
                for (int c = 0; c < NO; ++c)
                    dst[c][j] = __builtin_amdgcn_raw_buffer_load_b128(w3_rsrc, (int)((b_off[j] + kb + c * piece_bytes) | bad), 0, 0);
            }
        };

        // ---- A fragments: lane (r, h) of tile i holds channels 8h..8h+7 of halo pixel (own pixel + tap) = one ds_read_b128 per piece.
        int a_idx[TM];                                         // uint4 index of the lane's pixel at tap offset (0,0) (bf16: swizzle bit included)
        int a_rot[TM];                                         // fp32: (lx + ly) & 3 of the lane's pixel
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int qq = i * 32 + r;                         // row within the patch
            const int ly = quad ? (qq >> 2) & 3 : qq >> 3;
            const int lx = quad ? qq & 3 : qq & 7;
            const int lpix = quad ? (qq >> 4) * XQIMG + ly * XQROW + lx : ly * XROW + lx;
            a_idx[i] = F32 ? (wm * XH_MAX + lpix) * 4 : (wm * XH_MAX + lpix) * 2 + (h ^ (ly & 1));
            a_rot[i] = (lx + ly) & 3;
        }
        const int HWL = quad ? XQROW : XROW;                  // LDS row pitch in pixels
        u32x4 A0[NO][TM], A1[NO][TM];                          // two sets: the next tap's fragments are read while this tap's MFMAs run
        auto read_a = [&](int buf, int dy, int dx, u32x4 (&dst)[NO][TM]) {      // halo offset of the tap: scalar
            const int poff = dy * HWL + dx;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (F32) {
                    const int rot = (a_rot[i] + dy + dx) & 3, base = a_idx[i] + 4 * poff;
#pragma unroll
                    for (int c = 0; c < NO; ++c) {
                        const uint4 v = Hs[buf][0][base + ((2 * c + h) ^ rot)];
                        dst[c][i] = u32x4{v.x, v.y, v.z, v.w};
                    }
                } else {
                    const int at = (a_idx[i] + 2 * poff) ^ (dy & 1);
#pragma unroll
                    for (int c = 0; c < NO; ++c) {
                        const uint4 v = Hs[buf][c < NI ? c : 0][at];
                        dst[c][i] = u32x4{v.x, v.y, v.z, v.w};
                    }
                }
            }
        };

        if (!accumulate || sub == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
        }

        // ---- prologue: halo of chunk 0 into image 0, filter slices of steps 0 and 1, fragments of tap 0 -------------------------
        fetch_halo(0);
        load_b(0, B0);
        load_b(1, B1);
        commit_halo(0);
        __syncthreads();
        read_a(0, dy0, dx0, A0);

        // One K step = one tap of one 16-channel chunk = one 16-k MFMA step on this step's filter slice b_cur.
        // Control flow is kept to what SIInsertWaitcnts can see through: it merges pending memory operations conservatively at every
        // join, and a path that skips a step (an `if (s + 1 < n_steps)` guard inside the loop) or a conditional fragment read turned the
        // counted waits of this loop into vmcnt(0) / lgkmcnt(0) — the L2 round trip of the filter loads issued one step earlier exposed
        // in front of every MFMA block.  So: the loop runs PAIRS of steps with no guard between them (an odd last step is peeled) and
        // the next tap's fragments are ALWAYS read inside the MFMA block: the next chunk's image is committed one step before the
        // chunk ends (after the MFMAs of tap T-2, followed by the chunk's one barrier), so the last tap's step reads tap 0 of the
        // NEW image like any other step reads its next tap.  The only branches are the two per-chunk events (fetch, commit + barrier).
        int ti = 0, tj = 0, cc = 0;                            // current tap (row, column of the tap grid) and chunk: scalar
        auto k_step = [&](int s, u32x4 (&b_cur)[NO][TN], u32x4 (&A)[NO][TM], u32x4 (&An)[NO][TM]) {
            int tjn = tj + 1, tin = ti;
            if (tjn == nkw) { tjn = 0; tin = ti + 1; }
            const int tn = tin * nkw + tjn;                    // index of the next tap
            const bool last_tap = tn == T;                     // this step is the chunk's last tap: the next one is tap 0 of chunk cc + 1
            const bool commit_now = tn == T - 1 || T == 1;     // ... the tap before it: the next image is written after this step's MFMAs
            if (last_tap) { tin = 0; tjn = 0; }
            const bool more_chunks = cc + 1 < n_cc;
            const int buf = cc & 1;
            const int rbuf = last_tap ? buf ^ 1 : buf;
            const int dyn = dy0 + tin * ystep, dxn = dx0 + tjn * xstep;
            if (ti == 0 && tj == 0 && more_chunks) fetch_halo(cc + 1);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {                  // smallest terms first
                    f32x16 a = acc[i][j];
                    // the FILTER fragment is the first operand: a lane's 16 results are then 4 x 4 consecutive channels of ITS pixel
                    // (row index of D = filter (v & 3) + 8 (v >> 2) + 4 h, column = pixel lane & 31) and leave as 16-byte stores
                    if constexpr (F32) {
#pragma unroll
                        for (int c = 0; c < 2; ++c) {          // k = 8c + 4h + e at MFMA e: the same permutation of k for both operands
                            a = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(b_cur[c][j].x), __uint_as_float(A[c][i].x), a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(b_cur[c][j].y), __uint_as_float(A[c][i].y), a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(b_cur[c][j].z), __uint_as_float(A[c][i].z), a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(b_cur[c][j].w), __uint_as_float(A[c][i].w), a, 0, 0, 0);
                        }
                    } else {
                        auto bf = [](const u32x4& v) { return __builtin_bit_cast(bf16x8, v); };
                        if (NP == 3) {
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(b_cur[NO - 1][j]), bf(A[0][i]), a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(b_cur[0][j]), bf(A[NO - 1][i]), a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(b_cur[NO > 1 ? 1 : 0][j]), bf(A[NO > 1 ? 1 : 0][i]), a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(b_cur[NO > 1 ? 1 : 0][j]), bf(A[0][i]), a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(b_cur[0][j]), bf(A[NO > 1 ? 1 : 0][i]), a, 0, 0, 0);
                        }
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(b_cur[0][j]), bf(A[0][i]), a, 0, 0, 0);
                    }
                    acc[i][j] = a;
                    if (i == 0 && j == 0) {
                        // The next tap's fragments are read AFTER the first tile's MFMAs are issued and pinned there: at the loop header
                        // SIInsertWaitcnts waits for every outstanding LDS read (lgkmcnt(0)) in front of the first MFMA — with the reads
                        // in front of it that wait exposed their latency every other step; here it only covers reads a step old.
                        __builtin_amdgcn_sched_barrier(0);
                        read_a(rbuf, dyn, dxn, An);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            load_b(s + 2, b_cur);                              // this slot is free again: one full step of lead for the L2 round trip
            if (commit_now && more_chunks) {
                commit_halo(buf ^ 1);                          // the other image: last read one chunk ago, before that chunk's barrier
                __syncthreads();                               // the new image is complete before the last tap's step reads it
            }
            if (last_tap) cc = cc + 1;
            ti = tin; tj = tjn;
        };
        {
            int s = 0;
#pragma unroll 1
            for (; s + 1 < n_steps; s += 2) {
                k_step(s, B0, A0, A1);
                k_step(s + 1, B1, A1, A0);
            }
            if (s < n_steps) k_step(s, B0, A0, A1);
        }
        __syncthreads();          // every wave is done with the halo images (the next class / the epilogue tables reuse LDS state)

        // ---- epilogue (as igemm_halo) -------------------------------------------------------------------------------------
        if (accumulate && sub + 1 < n_sub) continue;
        if (tid < BM) {
            const int m = m0 + tid;
            int off = -1, roff = 0;
            if (m < M) {
                const RowCoord rc = kc_decode_row(m, OHc, OWc, quad ? 0 : 1);
                off = kc_out_offset(p, kc, rc);
                if (p.res) roff = kc_res_offset(p, kc, rc);
            }
            s_off[tid] = off;
            s_roff[tid] = roff;
        }
        __syncthreads();
        // GroupNorm partials (cslgan_conv_t.gn_part; single-class launches): per channel the sum and the sum of squares of the STORED
        // values about the patch's first pixel (a sample of the same distribution: no cancellation), reduced over the patch's 64 rows
        const bool gn = !GEN && p.gn_part != nullptr;
        float4 gk[TN][4], g1[TN][4], g2[TN][4];
        // lane (r, h) holds, of tile (i, j), pixel row r and channels j*32 + 8 q + 4 h + (0..3) for q = 0..3: bias / residual / mask are
        // read and the result is stored as 4-element vectors (p.Nn % 4 == 0 and 16-byte aligned operands: checked by x3h_eligible)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * 64 + i * 32 + r;
            const int off = s_off[row], roff = s_roff[row];
            if (off < 0) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * TN * 32 + j * 32 + 8 * q + 4 * h;
                    if (n >= p.Nn) continue;
                    float4 val = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                    if (cs > 1) {           // partial sums: bias / activation / mask are x3h_split_reduce_kernel's
                        *reinterpret_cast<float4*>(p.part + (long long)split * p.out_floats + off + n) = val;
                        continue;
                    }
                    if (p.bias) {
                        const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                        val.x += bv.x; val.y += bv.y; val.z += bv.z; val.w += bv.w;
                    }
                    if (p.res) {
                        const float4 rv = *reinterpret_cast<const float4*>(p.res + roff + n);
                        val.x += rv.x; val.y += rv.y; val.z += rv.z; val.w += rv.w;
                    }
                    if (gn) {
                        if (i == 0) {
                            const int src = lane & 32;              // the lane of pixel row 0 in this half
                            gk[j][q] = make_float4(__shfl(val.x, src), __shfl(val.y, src), __shfl(val.z, src), __shfl(val.w, src));
                            g1[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
                            g2[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                        const float4 d = make_float4(val.x - gk[j][q].x, val.y - gk[j][q].y, val.z - gk[j][q].z, val.w - gk[j][q].w);
                        g1[j][q].x += d.x; g1[j][q].y += d.y; g1[j][q].z += d.z; g1[j][q].w += d.w;
                        g2[j][q].x = fmaf(d.x, d.x, g2[j][q].x); g2[j][q].y = fmaf(d.y, d.y, g2[j][q].y);
                        g2[j][q].z = fmaf(d.z, d.z, g2[j][q].z); g2[j][q].w = fmaf(d.w, d.w, g2[j][q].w);
                    }
                    if (p.act == CSLGAN_ACT_LRELU02) {
                        val.x = val.x > 0.f ? val.x : 0.2f * val.x; val.y = val.y > 0.f ? val.y : 0.2f * val.y;
                        val.z = val.z > 0.f ? val.z : 0.2f * val.z; val.w = val.w > 0.f ? val.w : 0.2f * val.w;
                    } else if (p.act == CSLGAN_ACT_RELU) {
                        val.x = val.x > 0.f ? val.x : 0.f; val.y = val.y > 0.f ? val.y : 0.f;
                        val.z = val.z > 0.f ? val.z : 0.f; val.w = val.w > 0.f ? val.w : 0.f;
                    } else if (p.act == CSLGAN_ACT_TANH) {
                        val.x = tanhf(val.x); val.y = tanhf(val.y); val.z = tanhf(val.z); val.w = tanhf(val.w);
                    }
                    if (p.mask) {
                        const float4 mv = *reinterpret_cast<const float4*>(p.mask + off + n);
                        val.x *= mv.x > 0.f ? 1.f : 0.2f; val.y *= mv.y > 0.f ? 1.f : 0.2f;
                        val.z *= mv.z > 0.f ? 1.f : 0.2f; val.w *= mv.w > 0.f ? 1.f : 0.2f;
                    }
                    *reinterpret_cast<float4*>(p.out + off + n) = val;
                }
            }
        }
        if (gn) {
            // per channel (mean, centred sum of squares) of the wave's patch -> LDS (the halo images are dead) -> one thread per
            // (patch, group) combines its channels exactly (Chan) and writes the pair the apply kernel's prologue expects
            float* s_ch = reinterpret_cast<float*>(&Hs[0][0][0]);          // [2 patches][BN channels][2]
            float* s_k = s_ch + 2 * BN * 2;                                // [2 patches][BN channels]: the shifts
            const bool valid = s_off[wm * 64] >= 0;
            // Sum over the 32 pixel-row lanes of a half as a halving butterfly: at every step a lane hands HALF of its values to its
            // partner and adds the partner's other half to the ones it keeps — NV/2 + NV/4 + ... = NV - NV/32 exchanges instead of
            // 5 NV (a plain all-lanes reduction of the NV = 64 values cost ~5 k cycles per wave: as much as the statistics launch it
            // replaced).  Lane r ends with values 2r, 2r+1 (NV = 64) or value r (NV = 32): value index = 2 * channel + moment with
            // channel = (j, q, e) flattened, so a lane finishes BOTH moments of one channel (or one moment of channel r / 2).
            constexpr int NV = TN * 32;
            float v[NV];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float a1[4] = {g1[j][q].x, g1[j][q].y, g1[j][q].z, g1[j][q].w}, a2[4] = {g2[j][q].x, g2[j][q].y, g2[j][q].z, g2[j][q].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[((j * 4 + q) * 4 + e) * 2] = a1[e]; v[((j * 4 + q) * 4 + e) * 2 + 1] = a2[e]; }
                }
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const int n = NV >> (s + 1), mask = 16 >> s;
                const bool up = (r & mask) != 0;
#pragma unroll
                for (int i = 0; i < n; ++i) {
                    const float keep = up ? v[n + i] : v[i], send = up ? v[i] : v[n + i];
                    v[i] = keep + __shfl_xor(send, mask);
                }
            }
            if (valid) {
                // the shifts: every lane of a half holds all of them; pixel-row lane 0 stores its half's 16 TN channels
                if (r == 0) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<float4*>(s_k + wm * BN + wn * TN * 32 + j * 32 + 8 * q + 4 * h) = gk[j][q];
                }
                if (NV == 64) {         // lane r: channel r = (j, q, e), both moments
                    const int cl = wn * TN * 32 + (r >> 4) * 32 + 8 * ((r >> 2) & 3) + 4 * h + (r & 3);
                    s_ch[(wm * BN + cl) * 2] = v[0];
                    s_ch[(wm * BN + cl) * 2 + 1] = v[1];
                } else {                // lane r: moment r & 1 of channel r >> 1 = (q, e)
                    const int cc = r >> 1, cl = wn * TN * 32 + 8 * (cc >> 2) + 4 * h + (cc & 3);
                    s_ch[(wm * BN + cl) * 2 + (r & 1)] = v[0];
                }
            }
            __syncthreads();
            const int cpg = p.gn_cpg, gpt = BN / cpg;                      // groups per filter tile (cpg divides 32)
            if (tid < 2 * gpt) {
                const int pp = tid / gpt, gl = tid - pp * gpt;
                const int m = m0 + 64 * pp, c0 = n0 + gl * cpg;
                if (m < M && c0 < p.Nn) {
                    const float* q = s_ch + (pp * BN + gl * cpg) * 2;       // (S1, S2) about the shift k of each channel, 64 rows
                    const float* kq = s_k + pp * BN + gl * cpg;
                    float sum = 0.f;
                    for (int c = 0; c < cpg; ++c) sum += kq[c] + q[2 * c] * (1.f / 64.f);
                    const float mg = sum / (float)cpg;
                    float m2 = 0.f;
                    for (int c = 0; c < cpg; ++c) {
                        const float m1 = q[2 * c] * (1.f / 64.f), dm = kq[c] + m1 - mg;
                        float c2 = q[2 * c + 1] - q[2 * c] * m1;
                        m2 += (c2 < 0.f ? 0.f : c2) + 64.f * dm * dm;
                    }
                    const int per = OHc * OWc, img = m / per, slot = (m - img * per) >> 6;
                    float* o = p.gn_part + (((long long)img * p.gn_slots + slot) * (p.Nn / cpg) + c0 / cpg) * 2;
                    o[0] = mg * 64.f * (float)cpg;
                    o[1] = m2;
                }
            }
        }
        __syncthreads();          // the next class of a pair reuses s_off / s_roff
    }   // sub
}

// Second launch of a channel-split igemm_x3h: out = epilogue(part[0] + part[1] + ... + part[cs-1]) in that fixed order (a float
// atomic would make the sum order — and the last bit of every output — depend on workgroup scheduling).  Output rows are Nn floats
// (ldo == Nn, checked by the host); bias / activation / LeakyReLU-mask as the kernel's own epilogue.
__global__ __launch_bounds__(256) void x3h_split_reduce_kernel(const float* __restrict__ part, int cs, long long stride, float* __restrict__ out,
                                                               long long n4, int Nn, const float* __restrict__ bias,
                                                               const float* __restrict__ mask, int act) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(part)[i];
        for (int s = 1; s < cs; ++s) {
            const float4 u = reinterpret_cast<const float4*>(part + s * stride)[i];
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        if (bias) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + (int)((i << 2) % Nn));
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        }
        if (act == CSLGAN_ACT_LRELU02) {
            v.x = v.x > 0.f ? v.x : 0.2f * v.x; v.y = v.y > 0.f ? v.y : 0.2f * v.y; v.z = v.z > 0.f ? v.z : 0.2f * v.z; v.w = v.w > 0.f ? v.w : 0.2f * v.w;
        } else if (act == CSLGAN_ACT_RELU) {
            v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
        } else if (act == CSLGAN_ACT_TANH) {
            v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w);
        }
        if (mask) {
            const float4 mv = reinterpret_cast<const float4*>(mask)[i];
            v.x *= mv.x > 0.f ? 1.f : 0.2f; v.y *= mv.y > 0.f ? 1.f : 0.2f; v.z *= mv.z > 0.f ? 1.f : 0.2f; v.w *= mv.w > 0.f ? 1.f : 0.2f;
        }
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

// The filter matrix of one class, w[n][t][c] fp32 (KRSC for a forward conv; a repacked class matrix of a data gradient or of a
// stride-2 forward conv), split into its bfloat16 pieces and re-laid STEP-major for igemm_x3h:
//   w3[piece][step = (c/16) * T + t][n][c % 16]      (needs C % 16 == 0; every step's [Nn][16] slice is contiguous)
template <int NP>
__global__ void split_filter_x3_kernel(const float* __restrict__ w, int Nn, int T, int C, unsigned short* __restrict__ w3) {
    const long long n_el = (long long)Nn * T * C, n4 = n_el >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const long long e = i << 2;                       // element (n, t, c..c+3)
        const int c = (int)(e % C);
        const long long nt = e / C;
        const int t = (int)(nt % T), n = (int)(nt / T);
        const long long dst = ((((long long)(c >> 4) * T + t) * Nn + n) << 4) + (c & 15);
        if (NP == 0) {          // exact fp32: the same step-major order, fp32 elements
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(w3) + dst) = reinterpret_cast<const float4*>(w)[i];
            continue;
        }
        const x3_t s3 = xsplit4(reinterpret_cast<const float4*>(w)[i]);
        *reinterpret_cast<uint2*>(w3 + dst) = s3.hi;           // = the round-to-nearest-even bfloat16 of w: all the NP = 1 form needs
        if (NP == 3) {
            *reinterpret_cast<uint2*>(w3 + n_el + dst) = s3.mid;
            *reinterpret_cast<uint2*>(w3 + 2 * n_el + dst) = s3.lo;
        }
    }
}

// w: the class matrix [Nn][T][C]; w3: its piece buffer (pieces * Nn*T*C bfloat16).  C % 16 != 0: nothing is written (the halo form
// does not take the shape and the workspace stays unused).
int split_filter_x3(const float* w, int Nn, int T, int C, void* w3, hipStream_t st, int pieces) {
    if (C % 16) return CSLGAN_OK;
    long long nb = ((long long)Nn * T * C / 4 + 255) / 256;
    nb = nb > 2048 ? 2048 : (nb < 1 ? 1 : nb);
    if (pieces == 0) hipLaunchKernelGGL(split_filter_x3_kernel<0>, dim3((unsigned)nb), dim3(256), 0, st, w, Nn, T, C, reinterpret_cast<unsigned short*>(w3));
    else if (pieces == 1) hipLaunchKernelGGL(split_filter_x3_kernel<1>, dim3((unsigned)nb), dim3(256), 0, st, w, Nn, T, C, reinterpret_cast<unsigned short*>(w3));
    else hipLaunchKernelGGL(split_filter_x3_kernel<3>, dim3((unsigned)nb), dim3(256), 0, st, w, Nn, T, C, reinterpret_cast<unsigned short*>(w3));
    return check_launch("split_filter_x3_kernel");
}

static void tap_range(const KcClass& k, int& ymin, int& ymax, int& xmin, int& xmax) {
    ymin = 127; ymax = -128; xmin = 127; xmax = -128;
    for (int t = 0; t < k.T; ++t) {
        ymin = k.ty[t] < ymin ? k.ty[t] : ymin; ymax = k.ty[t] > ymax ? k.ty[t] : ymax;
        xmin = k.tx[t] < xmin ? k.tx[t] : xmin; xmax = k.tx[t] > xmax ? k.tx[t] : xmax;
    }
}

// Row-major affine tap grid: returns taps per row (nkw), 0 if the table is not one.
static int affine_taps(const KcClass& k) {
    int nkw = 1;
    while (nkw < k.T && k.ty[nkw] == k.ty[0]) ++nkw;
    if (k.T % nkw) return 0;
    const int nkh = k.T / nkw;
    const int xs = nkw > 1 ? k.tx[1] - k.tx[0] : 0, ys = nkh > 1 ? k.ty[nkw] - k.ty[0] : 0;
    if ((nkw > 1 && xs != 1 && xs != -1) || (nkh > 1 && ys != 1 && ys != -1)) return 0;
    for (int t = 0; t < k.T; ++t)
        if (k.ty[t] != k.ty[0] + (t / nkw) * ys || k.tx[t] != k.tx[0] + (t % nkw) * xs) return 0;
    return nkw;
}

// Shapes the x3 halo form takes: stride-1 classes on 8x8-patchable (or 4x4) grids, channels a multiple of 16, 2..25 affine taps
// within a 12x12 (6x6) halo, >= 64 output channels, and the pre-split filter (p.w3).
bool x3h_eligible(const KcParams& p) {
    static const int env = [] { const char* e = getenv("CSLGAN_X3_HALO"); return e ? atoi(e) : 1; }();
    static const int env32 = [] { const char* e = getenv("CSLGAN_F32_HALO"); return e ? atoi(e) : 1; }();
    if (!p.bf16 && !env32) return false;          // exact fp32 on this kernel (NP = 0) is an A/B switch of its own
    if (!env || !p.w3 || p.sy != 1 || p.sx != 1 || (p.AC & 15) || p.Nn < 64 || p.ksplit > 1 || !aligned16(p.a) || !aligned16(p.w3)) return false;
    // 4-element epilogue vectors: channel counts and the output row pitch multiples of 4, every epilogue operand 16-byte aligned
    if ((p.Nn & 3) || (p.ldo & 3) || !aligned16(p.out) || (p.bias && !aligned16(p.bias)) || (p.res && !aligned16(p.res)) || (p.mask && !aligned16(p.mask))) return false;
    static const int quad_min = [] { const char* e = getenv("CSLGAN_X3_QUAD_MIN"); return e ? atoi(e) : 2048; }();
    for (int c = 0; c < p.n_cls; ++c) {
        const KcClass& k = p.cls[c];
        const bool quad = k.OHc == 4 && k.OWc == 4;
        if (k.T < 2 || (k.M & 63) || (k.Kdim & 3) || (k.w_off & 7)) return false;
        if (quad && (k.M < quad_min || !p.bf16)) return false;   // exact fp32 keeps its round-3 kernels on 4x4 grids (measured: 52-73 vs 74-84 TF)
        if (!quad && ((k.OHc & 7) || (k.OWc & 7))) return false;
        if (!affine_taps(k)) return false;
        int ymin, ymax, xmin, xmax;
        tap_range(k, ymin, ymax, xmin, xmax);
        const int lim = quad ? 2 : 4;       // four 6x6 halos fill the 144-pixel LDS image
        if (ymax - ymin > lim || xmax - xmin > lim) return false;
    }
    return true;
}

int launch_x3h(KcParams& p, hipStream_t st) {
    if (p.a_bytes >= XFAR) { set_error("igemm_x3h: input tensor of 4 GB"); return CSLGAN_ERR_INVALID_ARG; }
    int tm = 0;
    bool gen = p.acc_classes != 0 || p.n_cls > 1;
    for (int c = 0; c < p.n_cls; ++c) {
        KcClass& k = p.cls[c];
        int ymin, ymax, xmin, xmax;
        tap_range(k, ymin, ymax, xmin, xmax);
        const bool quad = k.OHc == 4 && k.OWc == 4;
        const int side = quad ? 4 : 8;
        k.ty_min = ymin; k.tx_min = xmin; k.halo_h = side + ymax - ymin; k.halo_w = side + xmax - xmin;
        k.patch = quad ? 2 : 1;
        k.nkw = affine_taps(k);
        k.tile0 = tm;
        tm += (k.M + 127) / 128;
        gen = gen || quad || k.ay_mul > 1 || k.ax_mul > 1;
    }
    if (p.gn_part && gen) { set_error("igemm_x3h: GroupNorm partials need a single-class stride-1 launch"); return CSLGAN_ERR_INVALID_ARG; }
    if (p.in_scale && (gen || p.bf16 == 1 || !p.in_shift || !aligned16(p.in_scale) || !aligned16(p.in_shift))) {
        set_error("igemm_x3h: the input affine map needs a single-class stride-1 launch in fp32 or three-piece arithmetic and 16-byte aligned tables");
        return CSLGAN_ERR_INVALID_ARG;
    }
    p.tiles_m = tm;
    p.ksplit = 1;
    p.pair_mode = 0;
    bool same_m = true, same_t = true;
    for (int c = 1; c < p.n_cls; ++c) { same_m = same_m && p.cls[c].M == p.cls[0].M; same_t = same_t && p.cls[c].T == p.cls[0].T; }
    bool wide = p.Nn > 64;
    if (p.acc_classes) p.tiles_m = (p.cls[0].M + 127) / 128;       // all classes in every workgroup
    // unequal classes (9/6/6/4 taps): the heaviest runs with the lightest in ONE workgroup (9+4, 6+6 K steps) while the halved grid
    // still fills the chip — the rule of igemm_halo.hip
    static const int pair_min = [] { const char* e = getenv("CSLGAN_X3_PAIR_MIN"); return e ? atoi(e) : 256; }();
    if (!p.acc_classes && p.n_cls == 4 && same_m && !same_t) {
        const int tpc = (p.cls[0].M + 127) / 128;
        const long long paired = 2ll * tpc * (wide ? (p.Nn + 127) / 128 : (p.Nn + 63) / 64);
        if (paired >= pair_min) {
            int o[4] = {0, 1, 2, 3};
            for (int i = 0; i < 4; ++i)
                for (int j = i + 1; j < 4; ++j)
                    if (p.cls[o[j]].T > p.cls[o[i]].T) { const int t = o[i]; o[i] = o[j]; o[j] = t; }
            p.pair_mode = 1;
            p.pair_cls[0][0] = o[0]; p.pair_cls[0][1] = o[3];
            p.pair_cls[1][0] = o[1]; p.pair_cls[1][1] = o[2];
            p.tiles_per_cls = tpc;
            p.tiles_m = 2 * tpc;
        }
    }
    // 64-wide N tiles when the 128-wide grid would leave CUs idle (128-row launches of the critic's last layers)
    static const int wide_min = [] { const char* e = getenv("CSLGAN_X3_WIDE_MIN"); return e ? atoi(e) : 192; }();
    if (wide && (long long)p.tiles_m * ((p.Nn + 127) / 128) < wide_min) wide = false;
    p.tiles_n = wide ? (p.Nn + 127) / 128 : (p.Nn + 63) / 64;
    // Channel split: a launch of fewer than 512 workgroups (two per CU) leaves CUs idle or with one wave per SIMD — the critic's
    // third and fourth convs run 64..384 tiles.  With scratch from the caller the 16-channel chunks are divided over `csplit`
    // workgroups per tile; cost model = rounds of 256 CUs x work per workgroup, ceil(tiles * s / 256) / s, smallest s on ties,
    // at least two chunks per workgroup (prologue + epilogue stay a small share).
    p.csplit = 1;
    static const int split_env = [] { const char* e = getenv("CSLGAN_X3_SPLIT"); return e ? atoi(e) : 8; }();       // max split, 0/1 = off
    const long long tiles = (long long)p.tiles_m * p.tiles_n;
    if (split_env > 1 && p.part && !p.res && p.ldo == p.Nn && p.out_floats > 0 && (p.out_floats & 3) == 0 && aligned16(p.part) && tiles < 512) {
        const int n_cc = p.AC >> 4;
        double best = (double)((tiles + 255) / 256);
        for (int s = 2; s <= split_env && n_cc / s >= 2 && (long long)s * p.out_floats <= p.part_floats; ++s) {
            const double cost = (double)((tiles * s + 255) / 256) / s + 0.02 * s;      // + the partial stores / the reduce pass
            if (cost < best - 1e-9) { best = cost; p.csplit = s; }
        }
    }
    const dim3 grid((unsigned)(tiles * p.csplit)), block(256);
    const bool x3 = p.bf16 == 3;
    if (p.csplit > 1)       // "/sN": N workgroups per tile + the reduce launch (the device kernel name is the part before the slash)
        note_kernel(x3 ? "igemm_x3h_kernel<%d,3,%s,false>/s%d" : (p.bf16 ? "igemm_x3h_kernel<%d,1,%s,false>/s%d" : "igemm_x3h_kernel<%d,0,%s,false>/s%d"), wide ? 128 : 64,
                    gen ? "true" : "false", p.csplit);
    else if (p.in_scale)
        note_kernel(x3 ? "igemm_x3h_kernel<%d,3,false,true>" : "igemm_x3h_kernel<%d,0,false,true>", wide ? 128 : 64);
    else
        note_kernel(x3 ? "igemm_x3h_kernel<%d,3,%s,false>" : (p.bf16 ? "igemm_x3h_kernel<%d,1,%s,false>" : "igemm_x3h_kernel<%d,0,%s,false>"), wide ? 128 : 64, gen ? "true" : "false");
    if (p.in_scale) {
        if (x3 && wide) hipLaunchKernelGGL((igemm_x3h_kernel<128, 3, false, true>), grid, block, 0, st, p);
        else if (x3) hipLaunchKernelGGL((igemm_x3h_kernel<64, 3, false, true>), grid, block, 0, st, p);
        else if (wide) hipLaunchKernelGGL((igemm_x3h_kernel<128, 0, false, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((igemm_x3h_kernel<64, 0, false, true>), grid, block, 0, st, p);
    } else if (!p.bf16) {
        if (wide && gen) hipLaunchKernelGGL((igemm_x3h_kernel<128, 0, true>), grid, block, 0, st, p);
        else if (wide) hipLaunchKernelGGL((igemm_x3h_kernel<128, 0, false>), grid, block, 0, st, p);
        else if (gen) hipLaunchKernelGGL((igemm_x3h_kernel<64, 0, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((igemm_x3h_kernel<64, 0, false>), grid, block, 0, st, p);
    } else if (x3) {
        if (wide && gen) hipLaunchKernelGGL((igemm_x3h_kernel<128, 3, true>), grid, block, 0, st, p);
        else if (wide) hipLaunchKernelGGL((igemm_x3h_kernel<128, 3, false>), grid, block, 0, st, p);
        else if (gen) hipLaunchKernelGGL((igemm_x3h_kernel<64, 3, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((igemm_x3h_kernel<64, 3, false>), grid, block, 0, st, p);
    } else {
        if (wide && gen) hipLaunchKernelGGL((igemm_x3h_kernel<128, 1, true>), grid, block, 0, st, p);
        else if (wide) hipLaunchKernelGGL((igemm_x3h_kernel<128, 1, false>), grid, block, 0, st, p);
        else if (gen) hipLaunchKernelGGL((igemm_x3h_kernel<64, 1, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((igemm_x3h_kernel<64, 1, false>), grid, block, 0, st, p);
    }
    if (int rc = check_launch("igemm_x3h_kernel")) return rc;
    if (p.csplit > 1) {
        const long long n4 = p.out_floats >> 2;
        long long nb = (n4 + 255) / 256;
        nb = nb > 4096 ? 4096 : nb;
        hipLaunchKernelGGL(x3h_split_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, p.part, p.csplit, p.out_floats, p.out, n4, p.Nn, p.bias, p.mask, p.act);
        return check_launch("x3h_split_reduce_kernel");
    }
    return CSLGAN_OK;
}

}  // namespace cslgan
