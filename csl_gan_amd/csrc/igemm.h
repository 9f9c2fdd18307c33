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
constexpr int IG_MAX_TAPS = 32;

// "K-contiguous" implicit GEMM:  Out[m][n] = sum_k A(m,k) * Wm[n][k]
//   m -> (img, oy, ox) over a per-image grid OHc x OWc
//   k -> (tap t, channel c), k = t*AC + c
//   A(m,k) = a[img][(oy*sy+ty[t]) >> ups][(ox*sx+tx[t]) >> ups][c]   (zero outside the virtual VH x VW image)
// Forward conv: sy=stride, ty[t]=kh-pad.  Data gradient of a stride-s conv: one launch per output
// parity class with sy=1 and ty[t]=(py+pad-kh)/s over the taps kh == (py+pad) mod s.
struct KcParams {
    const float* a;
    int AH, AW, AC;      // stored dims of a
    int VH, VW;          // virtual dims used for the bounds test (2*AH,2*AW when ups==1)
    int ups;
    int M, OHc, OWc;     // rows and per-image grid
    int sy, sx;
    int T;
    int Kdim;            // T*AC
    const float* w;      // [Nn][ldw]
    int Nn, ldw;
    float* out;          // out[img][oy*osy+oy0][ox*osx+ox0][n], full dims OHf x OWf, ldo channels
    int OHf, OWf, osy, oy0, osx, ox0, ldo;
    int dense_out;       // 1: out offset == m*ldo (forward conv)
    const float* bias;
    const float* res;    // res[img][oyf>>rs][oxf>>rs][n]
    int res_shift;
    const float* mask;   // same indexing as out
    int act;
    int tiles_m, tiles_n;
    signed char ty[IG_MAX_TAPS], tx[IG_MAX_TAPS];
};

struct RowCoord {
    int img, oy, ox;
};

CSL_HD RowCoord kc_decode_row(int m, int OHc, int OWc) {
    RowCoord r;
    const int per = OHc * OWc;
    r.img = m / per;
    const int rem = m - r.img * per;
    r.oy = rem / OWc;
    r.ox = rem - r.oy * OWc;
    return r;
}

// Element offset into `a` of A(row, k) or -1 when the tap falls outside the image / k >= Kdim.
CSL_HD long long kc_a_offset(const KcParams& p, const RowCoord& rc, int k, int ty, int tx, int c) {
    (void)k;
    const int iy = rc.oy * p.sy + ty, ix = rc.ox * p.sx + tx;
    if (iy < 0 || iy >= p.VH || ix < 0 || ix >= p.VW) return -1;
    return (((long long)rc.img * p.AH + (iy >> p.ups)) * p.AW + (ix >> p.ups)) * p.AC + c;
}

CSL_HD int kc_out_offset(const KcParams& p, const RowCoord& rc) {
    return ((rc.img * p.OHf + rc.oy * p.osy + p.oy0) * p.OWf + rc.ox * p.osx + p.ox0) * p.ldo;
}

CSL_HD int kc_res_offset(const KcParams& p, const RowCoord& rc) {
    const int oyf = rc.oy * p.osy + p.oy0, oxf = rc.ox * p.osx + p.ox0;
    const int RH = p.OHf >> p.res_shift, RW = p.OWf >> p.res_shift;
    return ((rc.img * RH + (oyf >> p.res_shift)) * RW + (oxf >> p.res_shift)) * p.ldo;
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
    float* gw;           // [n_groups][Kc][Ndim] or null
    float* sq;           // [n_groups] or null
    int tiles_m, tiles_n;
    signed char ty[IG_MAX_TAPS], tx[IG_MAX_TAPS];   // kh-pad, kw-pad
};

}  // namespace cslgan
