// Index math shared by the implicit-GEMM kernels and the host-side index tests.
// Everything here is plain integer arithmetic, compiled for host and device.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CSL_HD __host__ __device__ __forceinline__
#else
#define CSL_HD inline
#endif

namespace cslgan {

constexpr int IG_BK = 32;          // K depth of one LDS tile (KC kernel)
constexpr int IG_MAX_TAPS = 40;

// "K-contiguous" implicit GEMM:  Out[m][n] = sum_k A(m,k) * Wm[n][k]
//   m -> (img, oy, ox) over a per-image grid OHc x OWc
//   k -> (tap t, channel c), k = t*AC + c
//   A(m,k) = a[img][oy*sy+ty[t]][ox*sx+tx[t]][c]   (zero outside the VH x VW image)
// One launch covers up to 4 "classes" that share a, out and the channel counts but have their own
// row grid, tap table, filter matrix and output phase:
//   forward conv            : 1 class, sy=stride, ty[t]=kh-pad
//   data gradient, stride s : s*s output-parity classes, sy=1, ty[t]=(py+pad-kh)/s over kh == (py+pad) mod s
//   stride-2 forward (halo) : 4 input-parity classes accumulated into one output (acc_classes)
constexpr int IG_MAX_CLS = 4;
struct KcClass {
    int M, OHc, OWc;     // rows and per-image grid of this class
    int patch;           // 1: rows enumerate 8x8 patches (needs OHc % 8 == 0 && OWc % 8 == 0)
                         // 2 (igemm_halo only): a 4x4 grid; 64 consecutive rows = the 4x4 grids of four consecutive images
    int T, Kdim;         // taps, T*AC
    int w_off;           // float offset of this class's [Nn][Kdim] filter matrix from KcParams::w
    int oy0, ox0;        // output phase: out[img][oy*osy+oy0][ox*osx+ox0][n]
    int tile0;           // first m-tile (in the launch's concatenated m-tile space)
    int nkw;             // igemm_x3h: taps per row of the (row-major, affine) tap grid
    int ty_min, tx_min, halo_h, halo_w;   // igemm_halo: tap offset range and the (8+range) halo of an 8x8 patch
    int ay_mul, ay_off, ax_mul, ax_off;   // igemm_halo, sub-image view of a: class pixel (y,x) is a[y*ay_mul+ay_off][x*ax_mul+ax_off]
                                          // (0 = unset = identity); the parity sub-images of a stride-2 forward conv
    signed char ty[IG_MAX_TAPS], tx[IG_MAX_TAPS];
};
struct KcParams {
    const float* a;
    unsigned a_bytes, w_bytes;   // byte sizes of a and w (buffer-descriptor range checks)
    unsigned ac_recip;           // ceil(2^32 / AC)
    int AH, AW, AC;      // stored dims of a
    int VH, VW;          // dims used for the bounds test (== AH, AW)
    int sy, sx;
    const float* w;
    int Nn;
    float* out;
    int OHf, OWf, osy, osx, ldo;
    int dense_out;       // 1: out offset == m*ldo (single-class forward conv)
    const float* bias;
    const float* res;    // same indexing as out (the ResBlockUp shortcut added before the activation)
    const float* mask;   // same indexing as out
    int act;
    int n_cls;
    int tiles_m, tiles_n;   // total m-tiles over all classes, n-tiles
    int ksplit;             // >1: K tiles are divided over ksplit workgroups that atomically add into zeroed out
    int pair_mode;          // igemm_halo: 1 = a workgroup runs TWO classes back to back on the same m-tile index (heaviest with
    int pair_cls[2][2];     // lightest: the 9+4 / 6+6 tap classes of a 5x5 stride-2 data gradient), tiles_m = 2 * tiles_per_cls
    int tiles_per_cls;
    int bf16;               // 0: exact fp32 MFMA; 1: operands rounded to bfloat16, v_mfma_f32_32x32x16_bf16 (igemm_bf16.hip);
                            // 3: fp32 emulated from three bfloat16 pieces per operand, six bf16 MFMAs per step
    const void* w3;         // igemm_halo_x3: the filter pre-split into three bfloat16 pieces, [3][Nn][Kdim] (same k order as w), or null
    int a_bf16;             // igemm_skinny only: the input tensor a is stored as bfloat16 (bf16 storage mode, csrc/igemm_bf16s.hip)
    int acc_classes;        // igemm_halo: 1 = every workgroup runs ALL classes on its m-tile into ONE accumulator (the classes are
                            // partial sums of the same output: a stride-2 conv as four stride-1 convs over parity sub-images)
    float* part;            // igemm_x3h: scratch for channel-split partial sums (cslgan_conv_t.split_ws) or null
    long long part_floats;  // its capacity
    long long out_floats;   // floats of the output tensor (0: unknown, never split)
    const float* in_scale;  // igemm_x3h (single class) / igemm_skinny: per-(image, channel) affine map applied while staging a
    const float* in_shift;  // (cslgan_conv_t.in_scale / in_shift), or null
    int in_relu;
    float* gn_part;         // igemm_x3h (single class): per-patch GroupNorm partial statistics of the stored values (cslgan_conv_t.gn_part)
    int gn_cpg;             // channels per group
    int gn_slots;           // patches per image (OH * OW / 64)
    int csplit;             // igemm_x3h: > 1 = the 16-channel chunks are divided over csplit workgroups per tile, partials in `part`
    KcClass cls[IG_MAX_CLS];
};

struct RowCoord {
    int img, oy, ox;
};

// Row index -> output coordinate.  patch == 0: row-major over the image (a 128-row tile is a 2-row strip of a 64-wide
// image: 5x5 taps then read (2+4)/2 = 3x the tile's own pixels).  patch == 1 (grid dims multiples of 8): rows are
// enumerated 8x8 patch by patch, so a 64-row tile is an 8x8 square and a 128-row tile an 8x16 rectangle — halo
// factor (12x20)/128 = 1.9 for 5x5 — which is what cuts the L2 / Infinity-Cache re-reads of the input.
CSL_HD RowCoord kc_decode_row(int m, int OHc, int OWc, int patch = 0) {
    RowCoord r;
    const int per = OHc * OWc;
    r.img = m / per;
    const int rem = m - r.img * per;
    if (patch) {
        const int pid = rem >> 6, q = rem & 63;
        const int gw = OWc >> 3;
        const int gy = pid / gw, gx = pid - gy * gw;
        r.oy = (gy << 3) + (q >> 3);
        r.ox = (gx << 3) + (q & 7);
    } else {
        r.oy = rem / OWc;
        r.ox = rem - r.oy * OWc;
    }
    return r;
}

// Element offsets of an output row (and of its residual operand) in the NHWC output tensor.
CSL_HD int kc_out_offset(const KcParams& p, const KcClass& k, const RowCoord& rc) {
    return ((rc.img * p.OHf + rc.oy * p.osy + k.oy0) * p.OWf + rc.ox * p.osx + k.ox0) * p.ldo;
}

CSL_HD int kc_res_offset(const KcParams& p, const KcClass& k, const RowCoord& rc) {
    return kc_out_offset(p, k, rc);
}

// Bijective XCD-aware remap of a linear workgroup id: blocks b and b+8 share an XCD, so give each
// XCD a contiguous run of tiles (neighbouring tiles share an A panel -> same L2).
CSL_HD int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, local = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// "M-contiguous" implicit GEMM for weight gradients:
//   Out[g][m][n] = alpha * sum_{k in group g} GY[k][m] * X(k, n),   k -> (img, oy, ox),  n -> (tap, c)
struct McParams {
    const float* gy;     // [N][P][Q][Kc]  (m = output channel, contiguous)
    const float* x;      // [N][H][W][C]
    int N, H, W, C, P, Q, Kc;
    int T;               // taps R*S
    int Ndim;            // T*C
    int stride;
    int group;           // samples per group
    int n_groups;
    float alpha;
    float* gw;           // [n_groups][Kc][Ndim] or null (bf16 elements when out_bf16)
    int out_bf16;        // 1: gw is stored as bfloat16 (round-to-nearest-even); sq is taken over the ROUNDED values
    float* sq;           // [n_groups] or null
    int tiles_m, tiles_n;
    int ksplit;          // >1: the group's pixels are divided over ksplit workgroups that atomically add into zeroed gw
    const float* row_scale;   // nullable [N]: gy of sample n is multiplied by row_scale[n] on load (clip-weighted sums)
    signed char ty[IG_MAX_TAPS], tx[IG_MAX_TAPS];   // kh-pad, kw-pad
};

// conv_c3.hip: the 3 -> 64 channel 5x5 stride-2 first layer on unpadded RGB input
bool c3_fwd_eligible(const cslgan_conv_t* c, const float* residual);
int launch_c3_fwd(const cslgan_conv_t* c, const float* x, const float* w, const float* bias, int act, float* y, hipStream_t st, int y_bf16 = 0);
bool c3_wgrad_eligible(const cslgan_conv_t* c, int group, int out_bf16, const void* gy);
int launch_c3_wgrad(const cslgan_conv_t* c, const float* gy, const float* x, float alpha, float* gw, float* sq, hipStream_t st, int gy_bf16 = 0);

// linear_k1.hip: linear layers with one output unit as streams
bool linear_k1_shape(const cslgan_conv_t* c);
int launch_linear_k1_fwd(const cslgan_conv_t* c, const float* x, const float* w, const float* bias, const float* residual, int act,
                         float* y, hipStream_t st);
int launch_linear_k1_dgrad(const cslgan_conv_t* c, const float* gy, const float* w, const float* mask, float* gx, hipStream_t st);

// conv1x1.hip: 1x1 convs with 32 / 64 / 128 input channels as a stream over 128-row tiles
bool conv1x1_eligible(const cslgan_conv_t* c, const float* x, const float* w, const float* residual);
int launch_conv1x1(const cslgan_conv_t* c, const float* x, const float* w, const float* bias, int act, float* y, hipStream_t st);

}  // namespace cslgan
