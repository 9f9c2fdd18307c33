// The critic's first conv (DCResNet_models.py:118,131: 3 -> 64 channels, 5x5, stride 2, pad 2) — forward and per-sample weight
// gradient on RGB input as it lies in HBM (NHWC, 12-byte pixels).  gfx950 only.
//
// The generic implicit-GEMM kernels need 16-byte channel vectors, so this layer used to run on a zero-padded 4th channel: a
// fill + a copy per call, then 128x64 tiles whose reduction is 100 long, a quarter of it zeros (31-52 TF on a layer whose
// bound is its 33.5 MB output write).  Here the reduction index is k = (tap, channel) = 75 (+1 zero row), both kernels stage
// the input window in LDS as contiguous rows of W*3 floats and address it with per-lane / per-step offsets, fp32 MFMA
// (v_mfma_f32_32x32x2_f32) throughout — this layer is 0.3 % of the step's FLOP and computes in exact fp32 in every
// --compute_dtype.
//
//   c3_fwd_kernel   : workgroup = 8x16 output pixels x 64 channels of one image; the 19 x 35-pixel input window and the
//                     transposed filter [76][64] are LDS-resident; each wavefront owns two output rows (32 pixels) and runs
//                     38 k steps x 2 MFMAs; bias + activation in the epilogue, 128-byte contiguous stores.
//   c3_wgrad_kernel : workgroup = (image, 32 of the 64 output channels); it walks the image in strips of 128 output pixels
//                     (gy strip [128][32] and the input rows it needs in LDS, the next strip prefetched into registers under
//                     the MFMAs), eight wavefronts split the strip's 64 k steps, partial sums meet in LDS; the epilogue scales,
//                     stores gw[img][k][5][5][3] and adds the squared norm into sq[img].
#include "common.h"
#include "igemm.h"

namespace cslgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C3_K = 64, C3_R = 5, C3_KD = 75, C3_KSTEPS = 38;
constexpr int C3F_PH = 19, C3F_ROW = 35 * 3, C3F_PITCH = C3F_ROW + 1;       // forward window: 19 rows x 35 pixels

struct C3Params {
    const float* x;      // [N][H][W][3]
    const float* w;      // [64][5][5][3]
    const float* bias;   // [64] or null
    const float* gy;     // [N][P][Q][64]   (wgrad)
    float* y;            // [N][P][Q][64]   (fwd)
    float* gw;           // [N][64][75] or null (wgrad)
    float* sq;           // [N] or null (wgrad): += ||alpha * gw_n||^2
    int N, H, W, P, Q, act;
    int y_bf16;          // fwd: y is stored as bfloat16 (round to nearest even) — csrc/igemm_bf16s.hip's storage mode
    int gy_bf16;         // wgrad: gy is bfloat16
    float alpha;
    int tiles_x, tiles_per_img, n_tiles;     // fwd
    int rows_per_strip, strips, pitch;        // wgrad
};

// offset of reduction index k = (ty*5 + tx)*3 + c inside a window whose rows are `pitch` floats apart
__device__ __forceinline__ constexpr int c3_koff(int k, int pitch) {
    return k >= C3_KD ? 0 : ((k / 3) / 5) * pitch + ((k / 3) % 5) * 3 + (k % 3);
}

constexpr int C3F_WLD = C3_K + 1;          // filter rows in LDS are 65 floats apart: the transposing store is conflict-free
constexpr int C3F_XREG = (C3F_PH * C3F_ROW + 255) / 256;

__global__ __launch_bounds__(256) void c3_fwd_kernel(const C3Params p) {
    __shared__ float Xs[C3F_PH * C3F_PITCH];
    __shared__ float Ws[(C3_KD + 1) * C3F_WLD];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // filter, transposed to [k][n] (read contiguously, k fastest); row 75 is the zero row that evens the reduction out to 38 steps of two
    for (int idx = tid; idx < C3_K * C3_KD; idx += 256) {
        const int n = idx / C3_KD, k = idx - n * C3_KD;
        Ws[k * C3F_WLD + n] = p.w[idx];
    }
    if (tid < C3_K) Ws[C3_KD * C3F_WLD + tid] = 0.f;
    const int py = 2 * wid + (r >> 4), px = r & 15;           // this lane's output pixel inside the 8x16 tile
    const int a_base = (2 * py) * C3F_PITCH + (2 * px) * 3;
    const int b_base = h * C3F_WLD + r;
    const long long row_f = (long long)p.W * 3;

    // the window of the NEXT tile travels in registers while this tile's MFMAs run
    float rx[C3F_XREG];
    auto fetch = [&](int tile) {
        const int img = tile / p.tiles_per_img, tt = tile - img * p.tiles_per_img;
        const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
        const int iy0 = 16 * ty - 2, jx0 = (32 * tx - 2) * 3;        // window origin (row, float column)
#pragma unroll
        for (int j = 0; j < C3F_XREG; ++j) {
            const int idx = tid + 256 * j;
            const int row = idx / C3F_ROW, col = idx - row * C3F_ROW;
            const int iy = iy0 + row, jx = jx0 + col;
            float v = 0.f;
            if (idx < C3F_PH * C3F_ROW && (unsigned)iy < (unsigned)p.H && (unsigned)jx < (unsigned)(p.W * 3))
                v = p.x[((long long)img * p.H + iy) * row_f + jx];
            rx[j] = v;
        }
    };
    if ((int)blockIdx.x < p.n_tiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const int img = tile / p.tiles_per_img, tt = tile - img * p.tiles_per_img;
        const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
        const int oy0 = ty * 8, ox0 = tx * 16;
        __syncthreads();                                           // the previous tile's reads are done (and Ws is written)
#pragma unroll
        for (int j = 0; j < C3F_XREG; ++j) {
            const int idx = tid + 256 * j;
            if (idx < C3F_PH * C3F_ROW) Xs[(idx / C3F_ROW) * C3F_PITCH + idx % C3F_ROW] = rx[j];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < p.n_tiles) fetch(tile + gridDim.x);
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
#pragma unroll
        for (int ks = 0; ks < C3_KSTEPS; ++ks) {
            const int o0 = c3_koff(2 * ks, C3F_PITCH), o1 = c3_koff(2 * ks + 1, C3F_PITCH);
            const float a = Xs[a_base + (h ? o1 : o0)];
            const float b0 = Ws[b_base + 2 * ks * C3F_WLD], b1 = Ws[b_base + 2 * ks * C3F_WLD + 32];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
        }
        // rows of the MFMA tile = the wavefront's 32 pixels, columns = output channels
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = j * 32 + r;
            const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = (v & 3) + 8 * (v >> 2) + 4 * h;
                const int oy = oy0 + 2 * wid + (m >> 4), ox = ox0 + (m & 15);
                float val = acc[j][v] + bv;
                if (p.act == CSLGAN_ACT_LRELU02) val = val > 0.f ? val : 0.2f * val;
                else if (p.act == CSLGAN_ACT_RELU) val = val > 0.f ? val : 0.f;
                else if (p.act == CSLGAN_ACT_TANH) val = tanhf(val);
                const long long o = (((long long)img * p.P + oy) * p.Q + ox) * C3_K + n;
                if (p.y_bf16) {
                    unsigned u = __float_as_uint(val);
                    u += 0x7FFFu + ((u >> 16) & 1u);
                    reinterpret_cast<unsigned short*>(p.y)[o] = (unsigned short)(u >> 16);
                } else {
                    p.y[o] = val;
                }
            }
        }
    }
}

constexpr int C3W_STRIP = 128;            // output pixels per strip
constexpr int C3W_MAXX = 2816;            // floats of the input rows of a strip, (2*rows+3) x pitch with rows*Q = 128: 7 x 399 (W = 128), 11 x 207, 19 x 111
constexpr int C3W_NT = 3;                 // 32-wide column tiles over the 75 (tap, channel) columns

constexpr int C3W_XREG = (C3W_MAXX + 511) / 512;
struct C3wStage { float4 g0, g1; float x[C3W_XREG]; };

__device__ __forceinline__ void c3w_fetch(const C3Params& p, C3wStage& st, int s, int img, int m0, int tid, int pitch, int x_total) {
    const long long g0 = ((long long)img * p.P * p.Q + (long long)s * C3W_STRIP) * C3_K + m0;
    if (p.gy_bf16) {
        const unsigned short* gh = reinterpret_cast<const unsigned short*>(p.gy);
        const uint2 a = *reinterpret_cast<const uint2*>(gh + g0 + (long long)(tid >> 3) * C3_K + (tid & 7) * 4);
        const uint2 b = *reinterpret_cast<const uint2*>(gh + g0 + (long long)((tid + 512) >> 3) * C3_K + (tid & 7) * 4);
        st.g0 = make_float4(__uint_as_float(a.x << 16), __uint_as_float(a.x & 0xffff0000u), __uint_as_float(a.y << 16), __uint_as_float(a.y & 0xffff0000u));
        st.g1 = make_float4(__uint_as_float(b.x << 16), __uint_as_float(b.x & 0xffff0000u), __uint_as_float(b.y << 16), __uint_as_float(b.y & 0xffff0000u));
    } else {
        st.g0 = *reinterpret_cast<const float4*>(p.gy + g0 + (long long)(tid >> 3) * C3_K + (tid & 7) * 4);
        st.g1 = *reinterpret_cast<const float4*>(p.gy + g0 + (long long)((tid + 512) >> 3) * C3_K + (tid & 7) * 4);
    }
    const int iy0 = 2 * s * p.rows_per_strip - 2;
    const long long row_f = (long long)p.W * 3;
#pragma unroll
    for (int j = 0; j < C3W_XREG; ++j) {
        const int idx = tid + 512 * j;
        const int row = idx / pitch, col = idx - row * pitch;
        const int iy = iy0 + row, jx = col - 6;              // two zero pixels on the left
        float v = 0.f;
        if (idx < x_total && (unsigned)iy < (unsigned)p.H && (unsigned)jx < (unsigned)(p.W * 3))
            v = p.x[((long long)img * p.H + iy) * row_f + jx];
        st.x[j] = v;
    }
}

__device__ __forceinline__ void c3w_commit(const C3wStage& st, float* Gs, float* Xs, int tid, int x_total) {
    *reinterpret_cast<float4*>(&Gs[tid * 4]) = st.g0;
    *reinterpret_cast<float4*>(&Gs[(tid + 512) * 4]) = st.g1;
#pragma unroll
    for (int j = 0; j < C3W_XREG; ++j) {
        const int idx = tid + 512 * j;
        if (idx < x_total) Xs[idx] = st.x[j];
    }
}

__global__ __launch_bounds__(512, 1) void c3_wgrad_kernel(const C3Params p) {
    __shared__ __attribute__((aligned(16))) float Gs[C3W_STRIP * 32];      // gy strip, this workgroup's 32 channels; later the 32 x 96 sums
    __shared__ float Xs[C3W_MAXX];
    __shared__ float Red[4 * 48 * 64];                                     // cross-wavefront reduction of the accumulators
    __shared__ float s_red[8];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int img = blockIdx.x >> 1, m0 = (blockIdx.x & 1) * 32;
    const int pitch = p.pitch, xrows = 2 * p.rows_per_strip + 3;
    const int qs = 31 - __builtin_clz((unsigned)p.Q);            // Q is a power of two (it divides 128)
    const int x_total = xrows * pitch;

    const int noff0 = c3_koff(r, pitch), noff1 = c3_koff(32 + r, pitch), noff2 = c3_koff(64 + r, pitch);

    f32x16 acc[C3W_NT];
#pragma unroll
    for (int j = 0; j < C3W_NT; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;

    // staging registers, two strips deep (HBM latency is several times a strip's 0.7 us of MFMAs): gy strip = 128 pixels x 32
    // channels = 1024 float4, two per thread; input rows <= 6 floats per thread
    C3wStage st0, st1;
    c3w_fetch(p, st0, 0, img, m0, tid, pitch, x_total);
    if (p.strips > 1) c3w_fetch(p, st1, 1, img, m0, tid, pitch, x_total);
#define C3W_STRIP_STEP(ST, SIDX)                                                                                         \
    {                                                                                                                    \
        __syncthreads(); /* every wavefront is done with the previous strip */                                           \
        c3w_commit(ST, Gs, Xs, tid, x_total);                                                                            \
        __syncthreads();                                                                                                 \
        /* two strips ahead: in flight under this strip's and the next strip's MFMAs */                                  \
        if ((SIDX) + 2 < p.strips) c3w_fetch(p, ST, (SIDX) + 2, img, m0, tid, pitch, x_total);                           \
        /* this wavefront's 16 pixels of the strip: 8 steps of two */                                                    \
        _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) {                                                               \
            const int pix = wid * 16 + 2 * ks + h;                                                                       \
            const int pr = pix >> qs, pc = pix & (p.Q - 1);                                                              \
            const float a = Gs[pix * 32 + r];                                                                            \
            const int xb = (2 * pr) * pitch + (2 * pc) * 3;                                                              \
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Xs[xb + noff0], acc[0], 0, 0, 0);                           \
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Xs[xb + noff1], acc[1], 0, 0, 0);                           \
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Xs[xb + noff2], acc[2], 0, 0, 0);                           \
        }                                                                                                                \
    }
    for (int s = 0; s < p.strips; s += 2) {
        C3W_STRIP_STEP(st0, s)
        if (s + 1 < p.strips) C3W_STRIP_STEP(st1, s + 1)
    }
#undef C3W_STRIP_STEP
    // ---- the eight partial [32 x 96] sums meet in LDS: three halving rounds, registers -> LDS -> registers (LDS float atomics
    // compile to compare-and-swap loops: 30 us for this step when tried) --------------------------------------------------
#pragma unroll
    for (int half = 4; half >= 1; half >>= 1) {
        __syncthreads();
        if (wid >= half && wid < 2 * half) {
#pragma unroll
            for (int j = 0; j < C3W_NT; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) Red[((wid - half) * 48 + j * 16 + v) * 64 + lane] = acc[j][v];
        }
        __syncthreads();
        if (wid < half) {
#pragma unroll
            for (int j = 0; j < C3W_NT; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[j][v] += Red[(wid * 48 + j * 16 + v) * 64 + lane];
        }
    }
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int j = 0; j < C3W_NT; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) Gs[((v & 3) + 8 * (v >> 2) + 4 * h) * 96 + j * 32 + r] = acc[j][v];
    }
    __syncthreads();
    float ss = 0.f;
    float* out = p.gw ? p.gw + ((long long)img * C3_K + m0) * C3_KD : nullptr;
    for (int idx = tid; idx < 32 * C3_KD; idx += 512) {
        const int m = idx / C3_KD, n = idx - m * C3_KD;
        const float val = p.alpha * Gs[m * 96 + n];
        ss = fmaf(val, val, ss);
        if (out) out[idx] = val;
    }
    if (p.sq) {
        ss = wave_sum(ss);
        if (lane == 0) s_red[wid] = ss;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += s_red[w];
            atomicAdd(p.sq + img, t);
        }
    }
}

static bool c3_shape(const cslgan_conv_t* c) {
    return c->C == 3 && c->K == C3_K && c->R == C3_R && c->S == C3_R && c->stride == 2 && c->pad == 2 &&
           c->P * 2 == c->H && c->Q * 2 == c->W;
}

bool c3_fwd_eligible(const cslgan_conv_t* c, const float* residual) {
    return c3_shape(c) && residual == nullptr && (c->P & 7) == 0 && (c->Q & 15) == 0;
}

int launch_c3_fwd(const cslgan_conv_t* c, const float* x, const float* w, const float* bias, int act, float* y, hipStream_t st, int y_bf16) {
    C3Params p{};
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.act = act; p.y_bf16 = y_bf16;
    p.N = c->N; p.H = c->H; p.W = c->W; p.P = c->P; p.Q = c->Q;
    p.tiles_x = c->Q / 16; p.tiles_per_img = (c->P / 8) * p.tiles_x;
    const long long nt = (long long)c->N * p.tiles_per_img;
    CSLGAN_REQUIRE(nt < (1ll << 31), "conv2d_fwd (3-channel first layer): too many tiles");
    p.n_tiles = (int)nt;
    const unsigned grid = (unsigned)(nt < 512 ? nt : 512);      // persistent: 2 workgroups per CU, each prefetching its next tile
    note_kernel("c3_fwd_kernel");
    hipLaunchKernelGGL(c3_fwd_kernel, dim3(grid), dim3(256), 0, st, p);
    return check_launch("c3_fwd_kernel");
}

// strips of 128 output pixels that are whole rows: Q divides 128 and the rows of a strip divide P; W <= 128
bool c3_wgrad_eligible(const cslgan_conv_t* c, int group, int out_bf16, const void* gy) {
    if (!c3_shape(c) || group != 1 || out_bf16 || !aligned16(gy)) return false;
    if (c->Q < 16 || c->Q > 64 || (C3W_STRIP % c->Q) != 0) return false;
    const int rows = C3W_STRIP / c->Q;
    return c->P % rows == 0 && (2 * rows + 3) * ((c->W + 4) * 3 + 3) <= C3W_MAXX;
}

int launch_c3_wgrad(const cslgan_conv_t* c, const float* gy, const float* x, float alpha, float* gw, float* sq, hipStream_t st, int gy_bf16) {
    C3Params p{};
    p.x = x; p.gy = gy; p.gw = gw; p.sq = sq; p.alpha = alpha; p.gy_bf16 = gy_bf16;
    p.N = c->N; p.H = c->H; p.W = c->W; p.P = c->P; p.Q = c->Q;
    p.rows_per_strip = C3W_STRIP / c->Q; p.strips = c->P / p.rows_per_strip;
    p.pitch = (c->W + 4) * 3 + 3;             // = 15 (mod 32) for W = 32, 64, 128: the 75 columns of a step spread over the banks
    note_kernel("c3_wgrad_kernel");
    hipLaunchKernelGGL(c3_wgrad_kernel, dim3(2u * (unsigned)c->N), dim3(512), 0, st, p);
    return check_launch("c3_wgrad_kernel");
}

}  // namespace cslgan

using namespace cslgan;

extern "C" {

// The RGB first layer with a bfloat16-STORED output / output gradient (the bf16 storage mode of csrc/igemm_bf16s.hip); fp32 image,
// fp32 arithmetic, fp32 weight gradients as in the fp32 entries.  Shapes: exactly those of c3_fwd_eligible / c3_wgrad_eligible.
int cslgan_conv2d_c3_fwd_bf16out(const cslgan_conv_t* c, const float* x, const float* w, const float* bias, int act, void* y_bf16, void* stream) {
    CSLGAN_REQUIRE(c && x && w && y_bf16, "conv2d_c3_fwd_bf16out: null argument");
    CSLGAN_REQUIRE(act >= 0 && act <= 3, "conv2d_c3_fwd_bf16out: unknown activation %d", act);
    CSLGAN_REQUIRE(c3_fwd_eligible(c, nullptr), "conv2d_c3_fwd_bf16out: not the 3 -> 64 channel 5x5 stride-2 first layer on a 16x32-tileable image");
    return launch_c3_fwd(c, x, w, bias, act, reinterpret_cast<float*>(y_bf16), (hipStream_t)stream, 1);
}

int cslgan_conv2d_c3_wgrad_bf16gy(const cslgan_conv_t* c, const void* gy_bf16, const float* x, float alpha, float* gw, float* sq, void* stream) {
    CSLGAN_REQUIRE(c && gy_bf16 && x && (gw || sq), "conv2d_c3_wgrad_bf16gy: null argument");
    CSLGAN_REQUIRE(c3_wgrad_eligible(c, 1, 0, gy_bf16), "conv2d_c3_wgrad_bf16gy: shape not taken by the first-layer kernel");
    return launch_c3_wgrad(c, reinterpret_cast<const float*>(gy_bf16), x, alpha, gw, sq, (hipStream_t)stream, 1);
}

}  // extern "C"
