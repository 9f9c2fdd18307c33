// fp32 convolution with 1..4 output channels (the generator's 64 -> 3 output conv, the data gradient of the
// critic's 3 -> 64 first conv), on the vector ALU with an LDS-resident input halo.  gfx950 only.
//
// A 32-wide MFMA tile spends 29/32 of its work on padding when the GEMM's N is 3 (rocprofv3: 0.22 ms for the
// 1.8 GFLOP output conv, 5.9 TFLOP/s for the first layer's data gradient).  With N this small the op is a
// stream over the input: every input value meets only NJ filter values.  One workgroup owns one 8x8 output
// patch: it stages the patch's (8+range)^2 x 64-channel halo in LDS once, and 16 lanes share each output pixel
// (4 channels per lane, one ds_read_b128 per tap) with that lane's T x NJ filter float4s held in registers; the
// 16 partial sums meet in a 4-step butterfly.  Same KcParams classes / epilogue contract as igemm_kc.
#include "common.h"
#include "igemm.h"

namespace cslgan {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned SOOB = 0xFFFFFFF0u;
constexpr int SK_MAXT = 9;             // taps per class held in registers
constexpr int SK_C = 64;               // input channels (16 lanes x float4)
constexpr int SK_HALO = 12 * 12;

__device__ __forceinline__ float4 sbuf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// ALL: the classes of the launch (the four parity classes of a stride-2 data gradient) have the same grid and read the same
// input patch: one workgroup stages the union halo once and runs the classes one after the other (a quarter of the halo traffic;
// launch_skinny stores the union extents in every class).
template <int NJ, bool ALL>
__global__ __launch_bounds__(256) void igemm_skinny_kernel(const KcParams p) {
    __shared__ float4 Hs[SK_HALO * 16];

    const int tid = threadIdx.x;
    const int b = xcd_remap(blockIdx.x, p.tiles_m);
    int ci = 0;
    if (!ALL) {
#pragma unroll 1
        while (ci + 1 < p.n_cls && b >= p.cls[ci + 1].tile0) ++ci;
    }
    const KcClass& kc = p.cls[ci];
    const int HW_ = kc.halo_w, hpix = kc.halo_h * kc.halo_w;
    const RowCoord rc0 = kc_decode_row((ALL ? b : b - kc.tile0) * 64, kc.OHc, kc.OWc, 1);     // the patch's top-left pixel
    const int y0 = rc0.oy + kc.ty_min, x0 = rc0.ox + kc.tx_min;
    const int img_base = rc0.img * p.AH * p.AW * SK_C;

    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
    // ---- halo: hpix pixels x 16 float4 ------------------------------------------------------------------
    const int cl = tid & 15;
    if (p.a_bf16) {
        // bf16-stored input (csrc/igemm_bf16s.hip's storage mode): a 16-byte load is 8 channels, widened to two float4 of the fp32 LDS
        // image (the arithmetic stays fp32).  Two 4-byte loads per lane ran the 64 -> 3 output conv at 0.45 ms against 0.34 ms for the
        // fp32 input: the kernel is bound by its load instructions, so the bf16 form must not issue more of them than the fp32 one.
        constexpr int HREG8 = (SK_HALO * 8 + 255) / 256;
        uint4 r8[HREG8];
#pragma unroll
        for (int j = 0; j < HREG8; ++j) {
            const int idx = tid + 256 * j;
            const int c8 = idx & 7, pix = idx >> 3;
            const int hy = pix / HW_, hx = pix - hy * HW_;
            const int iy = y0 + hy, ix = x0 + hx;
            const bool ok = pix < hpix && (unsigned)iy < (unsigned)p.VH && (unsigned)ix < (unsigned)p.VW;
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)(ok ? 2u * (unsigned)(img_base + (iy * p.AW + ix) * SK_C + c8 * 8) : SOOB), 0, 0);
            r8[j] = make_uint4(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int j = 0; j < HREG8; ++j) {
            const int idx = tid + 256 * j;
            if ((idx >> 3) < hpix) {
                const uint4 v = r8[j];
                Hs[2 * idx] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
                Hs[2 * idx + 1] = make_float4(__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u), __uint_as_float(v.w << 16), __uint_as_float(v.w & 0xffff0000u));
            }
        }
    } else {
        constexpr int HREG = (SK_HALO * 16 + 255) / 256;
        float4 rh[HREG];
        unsigned okm = 0;
        // cslgan_conv_t.in_scale: GroupNorm (+ ReLU) of the producing layer as a per-(image, channel) affine map applied on the way
        // into LDS (one patch = one image, and a thread keeps the same four channels for all its slots); padding stays zero
        const bool aff = p.in_scale != nullptr;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (aff) {
            sc = *reinterpret_cast<const float4*>(p.in_scale + rc0.img * SK_C + cl * 4);
            sh = *reinterpret_cast<const float4*>(p.in_shift + rc0.img * SK_C + cl * 4);
        }
#pragma unroll
        for (int j = 0; j < HREG; ++j) {
            const int idx = tid + 256 * j;
            const int c4 = idx & 15, pix = idx >> 4;
            const int hy = pix / HW_, hx = pix - hy * HW_;
            const int iy = y0 + hy, ix = x0 + hx;
            const bool ok = pix < hpix && (unsigned)iy < (unsigned)p.VH && (unsigned)ix < (unsigned)p.VW;
            okm |= (ok ? 1u : 0u) << j;
            rh[j] = sbuf_load4(a_rsrc, ok ? 4u * (unsigned)(img_base + (iy * p.AW + ix) * SK_C + c4 * 4) : SOOB);
        }
#pragma unroll
        for (int j = 0; j < HREG; ++j) {
            const int idx = tid + 256 * j;
            if (aff) {
                const float lo = p.in_relu ? 0.f : -__builtin_inff();
                const bool in = (okm >> j) & 1u;
                rh[j].x = in ? fmaxf(fmaf(sc.x, rh[j].x, sh.x), lo) : 0.f;
                rh[j].y = in ? fmaxf(fmaf(sc.y, rh[j].y, sh.y), lo) : 0.f;
                rh[j].z = in ? fmaxf(fmaf(sc.z, rh[j].z, sh.z), lo) : 0.f;
                rh[j].w = in ? fmaxf(fmaf(sc.w, rh[j].w, sh.w), lo) : 0.f;
            }
            if ((idx >> 4) < hpix) Hs[idx] = rh[j];
        }
    }
    __syncthreads();

    const int wid = tid >> 6, pg = (tid & 63) >> 4;
    float bv = 0.f;
    if (p.bias && cl < p.Nn) bv = p.bias[cl];
#pragma unroll 1
    for (int cc = ALL ? 0 : ci; cc < (ALL ? p.n_cls : ci + 1); ++cc) {
    const KcClass& kk = p.cls[cc];
    const int Tc = kk.T;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w + kk.w_off), 0,
                                                                             p.w_bytes - 4u * (unsigned)kk.w_off, 0x00020000);
    // ---- this lane's filter values: 4 channels x T taps x NJ outputs ------------------------------------
    float4 wr[SK_MAXT][NJ];
    int toff[SK_MAXT];
#pragma unroll
    for (int t = 0; t < SK_MAXT; ++t) {
        toff[t] = t < Tc ? (((int)kk.ty[t] - kk.ty_min) * HW_ + ((int)kk.tx[t] - kk.tx_min)) * 16 : 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            wr[t][j] = sbuf_load4(w_rsrc, (t < Tc && j < p.Nn) ? 4u * (unsigned)(j * kk.Kdim + t * SK_C + cl * 4) : SOOB);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int q = wid * 16 + it * 4 + pg;            // pixel of the patch: 16 lanes each
        const int qy = q >> 3, qx = q & 7;
        const int base = (qy * HW_ + qx) * 16 + cl;
        float acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
#pragma unroll
        for (int t = 0; t < SK_MAXT; ++t) {
            if (t < Tc) {                                 // uniform
                const float4 a = Hs[base + toff[t]];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[j] = fmaf(a.x, wr[t][j].x, acc[j]);
                    acc[j] = fmaf(a.y, wr[t][j].y, acc[j]);
                    acc[j] = fmaf(a.z, wr[t][j].z, acc[j]);
                    acc[j] = fmaf(a.w, wr[t][j].w, acc[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            acc[j] += __shfl_xor(acc[j], 1);
            acc[j] += __shfl_xor(acc[j], 2);
            acc[j] += __shfl_xor(acc[j], 4);
            acc[j] += __shfl_xor(acc[j], 8);
        }
        if (cl < p.Nn) {
            float val = acc[0];
#pragma unroll
            for (int j = 1; j < NJ; ++j) val = cl == j ? acc[j] : val;
            val += bv;
            const RowCoord rc = {rc0.img, rc0.oy + qy, rc0.ox + qx};
            const int off = kc_out_offset(p, kk, rc);
            if (p.res) val += p.res[kc_res_offset(p, kk, rc) + cl];
            if (p.act == CSLGAN_ACT_LRELU02) val = val > 0.f ? val : 0.2f * val;
            else if (p.act == CSLGAN_ACT_RELU) val = val > 0.f ? val : 0.f;
            else if (p.act == CSLGAN_ACT_TANH) val = tanhf(val);
            if (p.mask) val *= (p.mask[off + cl] > 0.f ? 1.f : 0.2f);
            p.out[off + cl] = val;
        }
    }
    }
}

// Eligibility: at most 4 output channels, 64 input channels, stride 1, 8x8-patchable class grids, at most 9 taps per
// class within a 12x12 halo.
bool skinny_eligible(const KcParams& p) {
    if (p.Nn > 4 || p.Nn < 1 || p.sy != 1 || p.sx != 1 || p.AC != SK_C) return false;
    if (!aligned16(p.a) || !aligned16(p.w)) return false;
    for (int c = 0; c < p.n_cls; ++c) {
        const KcClass& k = p.cls[c];
        if (k.T < 1 || k.T > SK_MAXT || (k.OHc & 7) || (k.OWc & 7) || (k.M & 63) || (k.w_off & 3)) return false;
        int ymin = 127, ymax = -128, xmin = 127, xmax = -128;
        for (int t = 0; t < k.T; ++t) {
            ymin = k.ty[t] < ymin ? k.ty[t] : ymin; ymax = k.ty[t] > ymax ? k.ty[t] : ymax;
            xmin = k.tx[t] < xmin ? k.tx[t] : xmin; xmax = k.tx[t] > xmax ? k.tx[t] : xmax;
        }
        if (ymax - ymin > 4 || xmax - xmin > 4) return false;
    }
    return true;
}

int launch_skinny(KcParams& p, hipStream_t st) {
    int tm = 0;
    for (int c = 0; c < p.n_cls; ++c) {
        KcClass& k = p.cls[c];
        int ymin = 127, ymax = -128, xmin = 127, xmax = -128;
        for (int t = 0; t < k.T; ++t) {
            ymin = k.ty[t] < ymin ? k.ty[t] : ymin; ymax = k.ty[t] > ymax ? k.ty[t] : ymax;
            xmin = k.tx[t] < xmin ? k.tx[t] : xmin; xmax = k.tx[t] > xmax ? k.tx[t] : xmax;
        }
        k.ty_min = ymin; k.tx_min = xmin; k.halo_h = 8 + ymax - ymin; k.halo_w = 8 + xmax - xmin;
        k.patch = 1;
        k.tile0 = tm;                  // in 8x8 patches
        tm += k.M / 64;
    }
    // classes with one grid (the parity classes of a stride-2 data gradient on an even image) share their input patch: one
    // workgroup per patch cell runs them all from the union halo
    bool all = p.n_cls > 1;
    int uy0 = 127, uy1 = -128, ux0 = 127, ux1 = -128;
    for (int c = 0; c < p.n_cls; ++c) {
        const KcClass& k = p.cls[c];
        all = all && k.OHc == p.cls[0].OHc && k.OWc == p.cls[0].OWc && k.M == p.cls[0].M;
        uy0 = k.ty_min < uy0 ? k.ty_min : uy0; ux0 = k.tx_min < ux0 ? k.tx_min : ux0;
        uy1 = k.ty_min + k.halo_h > uy1 ? k.ty_min + k.halo_h : uy1; ux1 = k.tx_min + k.halo_w > ux1 ? k.tx_min + k.halo_w : ux1;
    }
    static const int all_env = [] { const char* e = getenv("CSLGAN_SKINNY_ALL"); return e ? atoi(e) : 1; }();
    all = all && all_env && (uy1 - uy0) * (ux1 - ux0) <= SK_HALO;
    if (all) {
        for (int c = 0; c < p.n_cls; ++c) {
            KcClass& k = p.cls[c];
            k.ty_min = uy0; k.tx_min = ux0; k.halo_h = uy1 - uy0; k.halo_w = ux1 - ux0;
        }
        tm = p.cls[0].M / 64;
    }
    p.tiles_m = tm;
    p.tiles_n = 1;
    p.ksplit = 1;
    const dim3 grid((unsigned)tm), block(256);
    note_kernel("igemm_skinny_kernel<%d%s>", p.Nn < 4 ? p.Nn : 4, all ? ",all" : "");
#define CSL_SKINNY(NJ_)                                                                              \
    if (all) hipLaunchKernelGGL((igemm_skinny_kernel<NJ_, true>), grid, block, 0, st, p);            \
    else hipLaunchKernelGGL((igemm_skinny_kernel<NJ_, false>), grid, block, 0, st, p);
    switch (p.Nn) {
        case 1: CSL_SKINNY(1) break;
        case 2: CSL_SKINNY(2) break;
        case 3: CSL_SKINNY(3) break;
        default: CSL_SKINNY(4) break;
    }
#undef CSL_SKINNY
    return check_launch("igemm_skinny_kernel");
}


// ---- dense weight gradient of a conv with 1..4 output channels (the generator's 64 -> 3 output conv, G step) ----------
//   gw[j][t][c] = alpha * sum_{img, pixel} gy[img][pixel][j] * x[img][pixel + t][c]
// As an MFMA GEMM this has M = 3 (1.8 TFLOP/s, 1.0 ms).  Same stream as the forward: a workgroup walks a strided list of
// 8x8 patches, stages each patch's x halo (64 channels) and its 64 x NJ gy values in LDS, and every lane keeps
// T x NJ float4 accumulators (its 4 channels) over ALL its patches; the workgroup's 16 pixel slots are added at the
// end (shuffle within a wavefront, then LDS) and each workgroup writes ONE partial [NJ][T][64] row, which the caller
// column-sums (deterministic: no float atomics).
struct SkinnyWgradParams {
    const float* gy;     // [N][P][Q][NJ]
    const float* x;      // [N][H][W][64]
    int N, H, W, P, Q, T, n_patches;
    unsigned x_bytes;
    float alpha;
    float* partial;      // [gridDim.x][NJ*T*64]
    signed char ty[SK_MAXT], tx[SK_MAXT];
    int ty_min, tx_min, halo_h, halo_w;
};

template <int NJ>
__global__ __launch_bounds__(256) void skinny_wgrad_kernel(const SkinnyWgradParams p) {
    __shared__ float4 Hs[SK_HALO * 16];
    __shared__ float Gs[64 * 4];
    const int tid = threadIdx.x;
    const int T = p.T, HW_ = p.halo_w, hpix = p.halo_h * p.halo_w;
    const int cl = tid & 15, wid = tid >> 6, pg = (tid & 63) >> 4;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    int toff[SK_MAXT];
#pragma unroll
    for (int t = 0; t < SK_MAXT; ++t) toff[t] = t < T ? (((int)p.ty[t] - p.ty_min) * HW_ + ((int)p.tx[t] - p.tx_min)) * 16 : 0;
    float4 acc[SK_MAXT][NJ];
#pragma unroll
    for (int t = 0; t < SK_MAXT; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[t][j] = make_float4(0.f, 0.f, 0.f, 0.f);

    constexpr int HREG = (SK_HALO * 16 + 255) / 256;
    const int per_img = (p.P >> 3) * (p.Q >> 3);
    for (int pt = blockIdx.x; pt < p.n_patches; pt += gridDim.x) {
        const int img = pt / per_img, rem = pt - img * per_img;
        const int gyy = rem / (p.Q >> 3), gxx = rem - gyy * (p.Q >> 3);
        const int oy0 = gyy << 3, ox0 = gxx << 3;
        float4 rh[HREG];
#pragma unroll
        for (int j = 0; j < HREG; ++j) {
            const int idx = tid + 256 * j;
            const int c4 = idx & 15, pix = idx >> 4;
            const int hy = pix / HW_, hx = pix - hy * HW_;
            const int iy = oy0 + p.ty_min + hy, ix = ox0 + p.tx_min + hx;
            const bool ok = pix < hpix && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            rh[j] = sbuf_load4(x_rsrc, ok ? 4u * (unsigned)(((img * p.H + iy) * p.W + ix) * SK_C + c4 * 4) : SOOB);
        }
        float gv = 0.f;
        if (tid < 64 * NJ) {
            const int q = tid / NJ, j = tid - q * NJ;
            gv = p.gy[((long long)(img * p.P + oy0 + (q >> 3)) * p.Q + ox0 + (q & 7)) * NJ + j];
        }
        __syncthreads();                 // every lane has finished the previous patch
#pragma unroll
        for (int j = 0; j < HREG; ++j) {
            const int idx = tid + 256 * j;
            if ((idx >> 4) < hpix) Hs[idx] = rh[j];
        }
        if (tid < 64 * NJ) Gs[(tid / NJ) * 4 + (tid % NJ)] = gv;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int q = wid * 16 + it * 4 + pg;
            const int base = ((q >> 3) * HW_ + (q & 7)) * 16 + cl;
            float g[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) g[j] = Gs[q * 4 + j];
#pragma unroll
            for (int t = 0; t < SK_MAXT; ++t) {
                if (t < T) {
                    const float4 a = Hs[base + toff[t]];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        acc[t][j].x = fmaf(g[j], a.x, acc[t][j].x);
                        acc[t][j].y = fmaf(g[j], a.y, acc[t][j].y);
                        acc[t][j].z = fmaf(g[j], a.z, acc[t][j].z);
                        acc[t][j].w = fmaf(g[j], a.w, acc[t][j].w);
                    }
                }
            }
        }
    }
    // ---- add the 16 pixel slots (4 per wavefront x 4 wavefronts) that share a channel group ---------------------
    __syncthreads();
    float* red = reinterpret_cast<float*>(Hs);          // [4 waves][NJ*T][64]  (<= 4*36*64*4 B = 36.9 KB)
#pragma unroll
    for (int t = 0; t < SK_MAXT; ++t) {
        if (t < T) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float4 v = acc[t][j];
                v.x += __shfl_xor(v.x, 16); v.y += __shfl_xor(v.y, 16); v.z += __shfl_xor(v.z, 16); v.w += __shfl_xor(v.w, 16);
                v.x += __shfl_xor(v.x, 32); v.y += __shfl_xor(v.y, 32); v.z += __shfl_xor(v.z, 32); v.w += __shfl_xor(v.w, 32);
                if (pg == 0) *reinterpret_cast<float4*>(&red[((wid * NJ + j) * T + t) * 64 + cl * 4]) = v;
            }
        }
    }
    __syncthreads();
    const int n_out = NJ * T * 64;
    float* dst = p.partial + (long long)blockIdx.x * n_out;
    for (int i = tid; i < n_out; i += 256)
        dst[i] = p.alpha * (red[i] + red[n_out + i] + red[2 * n_out + i] + red[3 * n_out + i]);
}

}  // namespace cslgan

using namespace cslgan;

extern "C" {

int cslgan_conv2d_wgrad_skinny_f32(const cslgan_conv_t* c, const float* gy, const float* x, float alpha, float* partial,
                                   int n_blocks, void* stream) {
    CSLGAN_REQUIRE(c && gy && x && partial, "conv2d_wgrad_skinny: null argument");
    CSLGAN_REQUIRE(c->K >= 1 && c->K <= 4 && c->C == SK_C, "conv2d_wgrad_skinny: needs 1..4 output and 64 input channels");
    CSLGAN_REQUIRE(c->stride == 1 && c->R * c->S <= SK_MAXT && c->R <= 5 && c->S <= 5, "conv2d_wgrad_skinny: needs stride 1 and at most 9 taps");
    CSLGAN_REQUIRE(c->N > 0 && c->P == c->H + 2 * c->pad - c->R + 1 && c->Q == c->W + 2 * c->pad - c->S + 1, "conv2d_wgrad_skinny: inconsistent output size");
    CSLGAN_REQUIRE((c->P & 7) == 0 && (c->Q & 7) == 0, "conv2d_wgrad_skinny: output grid must be a multiple of 8x8");
    CSLGAN_REQUIRE(n_blocks >= 1 && aligned16(x), "conv2d_wgrad_skinny: bad workspace / alignment");
    CSLGAN_REQUIRE(4ll * c->N * c->H * c->W * SK_C < 0xFFFFFFF0ll, "conv2d_wgrad_skinny: input larger than 4 GB");
    SkinnyWgradParams p{};
    p.gy = gy; p.x = x; p.N = c->N; p.H = c->H; p.W = c->W; p.P = c->P; p.Q = c->Q; p.T = c->R * c->S;
    p.n_patches = c->N * (c->P >> 3) * (c->Q >> 3);
    p.x_bytes = (unsigned)(4ll * c->N * c->H * c->W * SK_C);
    p.alpha = alpha; p.partial = partial;
    for (int kh = 0; kh < c->R; ++kh)
        for (int kw = 0; kw < c->S; ++kw) { p.ty[kh * c->S + kw] = (signed char)(kh - c->pad); p.tx[kh * c->S + kw] = (signed char)(kw - c->pad); }
    p.ty_min = -c->pad; p.tx_min = -c->pad; p.halo_h = 8 + c->R - 1; p.halo_w = 8 + c->S - 1;
    const dim3 grid((unsigned)n_blocks), block(256);
    note_kernel("skinny_wgrad_kernel<%d>", c->K);
    switch (c->K) {
        case 1: hipLaunchKernelGGL((skinny_wgrad_kernel<1>), grid, block, 0, (hipStream_t)stream, p); break;
        case 2: hipLaunchKernelGGL((skinny_wgrad_kernel<2>), grid, block, 0, (hipStream_t)stream, p); break;
        case 3: hipLaunchKernelGGL((skinny_wgrad_kernel<3>), grid, block, 0, (hipStream_t)stream, p); break;
        default: hipLaunchKernelGGL((skinny_wgrad_kernel<4>), grid, block, 0, (hipStream_t)stream, p); break;
    }
    return check_launch("skinny_wgrad_kernel");
}

}  // extern "C"
