// fp32 MFMA convolution with an LDS-resident input halo, for the stride-1 tap classes (5x5 "same" convs,
// the parity classes of the strided data gradient).  gfx950 only.
//
// igemm_kc_kernel re-gathers the A operand from global memory once per filter tap: 25 (or 9) passes over the
// same input window, each with its own address arithmetic, 16-byte loads and ds_write_b128.  rocprofv3 showed
// those re-reads overflowing the XCD's L2 (FETCH_SIZE = 3x the algorithmic bytes, served by the Infinity
// Cache) and the loader occupying a fifth of every K tile.  Here a workgroup owns two 8x8 output patches
// (128 rows) x BN channels; for every 32-channel chunk it stages the (8+R-1) x (8+S-1) input halo of each patch
// in LDS ONCE, and all taps read their A fragments straight from that image with a per-tap LDS offset — the
// global->LDS traffic of A drops by the tap count, and a K step (one tap of one chunk) only streams its 32 x BN
// filter slice.  Pixels are padded to 36 floats in LDS so the 16 lanes of a ds_read_b128 group hit distinct banks.
//
// Same MFMA schedule as igemm_kc (v_mfma_f32_32x32x2_f32, half-wave h owns k = 8g+4h+e), same epilogue.
#include "common.h"
#include "igemm.h"

namespace cslgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned HOOB = 0xFFFFFFF0u;
constexpr int PIX = 36;                // floats per halo pixel in LDS (32 channels + 4 pad)
constexpr int HALO_MAX = 12 * 12;      // pixels per patch halo (8+4 squared: up to 5x5 taps)

__device__ __forceinline__ float4 hbuf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// GEN = false: the lean instantiation for single-class / equal-class launches on 8x8 patches (the generator's convs);
// GEN = true adds class pairs and four-image patches (their index arithmetic costs the big convs 1-4 % when compiled in).
template <int BN, bool GEN>
__global__ __launch_bounds__(256, 2) void igemm_halo_kernel(const KcParams p) {
    constexpr int BM = 128, TM = 2, TN = BN / 64;            // waves 2 (M) x 2 (N); each wave = one patch x BN/2 channels
    constexpr int B_CH = BN * 4 + 4, B_PASS = BN / 32;
    __shared__ __attribute__((aligned(16))) float Hs[2 * HALO_MAX * PIX];
    __shared__ __attribute__((aligned(16))) float Bs[2][8 * B_CH];
    __shared__ int s_tapoff[IG_MAX_TAPS];
    __shared__ int s_off[BM];
    __shared__ int s_roff[BM];

    const int tid = threadIdx.x;
    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tile_mg = wg / p.tiles_n, tile_n = wg - tile_mg * p.tiles_n;
    const bool accumulate = GEN && p.acc_classes;
    const int n_sub = accumulate ? p.n_cls : ((GEN && p.pair_mode) ? 2 : 1);
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
#pragma unroll 1
        while (ci + 1 < p.n_cls && tile_mg >= p.cls[ci + 1].tile0) ++ci;
        tile_in_cls = tile_mg - p.cls[ci].tile0;
    }
    const KcClass& kc = p.cls[ci];
    const int M = kc.M, OHc = kc.OHc, OWc = kc.OWc, T = kc.T;
    const int m0 = tile_in_cls * BM, n0 = tile_n * BN;
    const int HW_ = kc.halo_w, HH_ = kc.halo_h;
    const bool quad = GEN && kc.patch == 2;          // 4x4 grids: a 64-row patch = four consecutive images
    const int hpix_img = HH_ * HW_;
    const int hpix = quad ? 4 * hpix_img : hpix_img;
    const int img_stride = p.AH * p.AW * p.AC;
    const int ay_mul = (GEN && kc.ay_mul) ? kc.ay_mul : 1, ay_off = GEN ? kc.ay_off : 0;
    const int ax_mul = (GEN && kc.ax_mul) ? kc.ax_mul : 1, ax_off = GEN ? kc.ax_off : 0;

    if (tid < IG_MAX_TAPS) s_tapoff[tid] = (((int)kc.ty[tid] - kc.ty_min) * HW_ + ((int)kc.tx[tid] - kc.tx_min)) * PIX;

    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w + kc.w_off), 0,
                                                                             p.w_bytes - 4u * (unsigned)kc.w_off, 0x00020000);
    // ---- the two patches of this tile: image and origin of their halos ----------------------------------
    int p_img[2], p_y0[2], p_x0[2];
    bool p_ok[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int m = m0 + 64 * pp;
        p_ok[pp] = m < M;
        const RowCoord rc = kc_decode_row(p_ok[pp] ? m : 0, OHc, OWc, quad ? 0 : 1);   // first row of the patch = its top-left pixel
        p_img[pp] = rc.img * img_stride;
        p_y0[pp] = rc.oy + kc.ty_min;
        p_x0[pp] = rc.ox + kc.tx_min;
    }
    // ---- B loader coordinates (filter slice of one tap, one 32-channel chunk) ----------------------------
    const int lrow = tid >> 3, q = tid & 7;
    unsigned b_off[B_PASS];
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
        const int n = n0 + lrow + 32 * i;
        b_off[i] = n < p.Nn ? 4u * ((unsigned)n * (unsigned)kc.Kdim + (unsigned)(q * 4)) : HOOB;
    }
    float4 rb[B_PASS];
    auto load_b = [&](int kbase) {       // kbase = t*AC + cc*32, or -1 past the end
#pragma unroll
        for (int i = 0; i < B_PASS; ++i)
            rb[i] = hbuf_load4(w_rsrc, (kbase < 0 || b_off[i] == HOOB) ? HOOB : b_off[i] + 4u * (unsigned)kbase);
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) *reinterpret_cast<float4*>(&Bs[buf][q * B_CH + (lrow + 32 * i) * 4]) = rb[i];
    };
    // ---- halo staging: 2 patches x hpix pixels x 8 chunks of 4 channels; <= 2*144*8/256 = 9 float4 per thread.
    // fetch_halo() issues the loads into registers (called at the top of a chunk's LAST tap step, so the global
    // latency hides under that step's MFMAs); commit_halo() writes them to LDS once every wave has left the chunk.
    constexpr int HREG = (2 * HALO_MAX * 8 + 255) / 256;
    float4 rh[HREG];
    const int h_total = 2 * hpix * 8;
    auto fetch_halo = [&](int cc) {
#pragma unroll
        for (int j = 0; j < HREG; ++j) {
            const int idx = tid + 256 * j;
            const int ch = idx & 7, pixg = idx >> 3;
            const int pp = pixg >= hpix ? 1 : 0;
            const int pix = pixg - pp * hpix;
            const int si = quad ? pix / hpix_img : 0;         // sub-image of a quad patch
            const int rem = pix - si * hpix_img;
            const int hy = rem / HW_, hx = rem - hy * HW_;
            const int iy = (p_y0[pp] + hy) * ay_mul + ay_off, ix = (p_x0[pp] + hx) * ax_mul + ax_off;
            const bool ok = idx < h_total && p_ok[pp] && (unsigned)iy < (unsigned)p.VH && (unsigned)ix < (unsigned)p.VW;
            const unsigned off = ok ? 4u * (unsigned)(p_img[pp] + si * img_stride + (iy * p.AW + ix) * p.AC + cc * 32 + ch * 4) : HOOB;
            rh[j] = hbuf_load4(a_rsrc, off);
        }
    };
    auto commit_halo = [&]() {
#pragma unroll
        for (int j = 0; j < HREG; ++j) {
            const int idx = tid + 256 * j;
            if (idx < h_total) {
                const int ch = idx & 7, pixg = idx >> 3;
                const int pp = pixg >= hpix ? 1 : 0;
                const int pix = pixg - pp * hpix;
                *reinterpret_cast<float4*>(&Hs[(pp * HALO_MAX + pix) * PIX + ch * 4]) = rh[j];
            }
        }
    };

    // ---- MFMA coordinates ---------------------------------------------------------------------------------
    const int lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;                   // wm = patch index
    int a_base[TM];                                          // LDS float offset of this lane's pixel (tap 0,0 corner) per MFMA tile
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int qq = i * 32 + r;                           // row within the patch
        const int lpix = quad ? (qq >> 4) * hpix_img + ((qq >> 2) & 3) * HW_ + (qq & 3) : (qq >> 3) * HW_ + (qq & 7);
        a_base[i] = (wm * HALO_MAX + lpix) * PIX + h * 4;
    }
    const int brow0 = wn * TN * 32 + r;

    if (!accumulate || sub == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    }

    const int n_cc = p.AC >> 5;
    const int n_steps = n_cc * T;
    // prologue: halo of chunk 0, filter slice of step 0
    fetch_halo(0);
    commit_halo();
    load_b(0);
    store_b(0);
    __syncthreads();

    int t = 0, cc = 0;
    for (int s = 0; s < n_steps; ++s) {
        const int buf = s & 1;
        int tn = t + 1, ccn = cc;
        if (tn == T) { tn = 0; ccn = cc + 1; }
        load_b(s + 1 < n_steps ? tn * p.AC + ccn * 32 : -1);
        const bool restage = ccn != cc && s + 1 < n_steps;    // uniform: this is the chunk's last tap
        if (restage) fetch_halo(ccn);
        const int toff = s_tapoff[t];
        float4 af[2][TM], bf[2][TN];
        auto load_frags = [&](int g, int slot) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[slot][i] = *reinterpret_cast<const float4*>(&Hs[a_base[i] + toff + g * 8]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[slot][j] = *reinterpret_cast<const float4*>(&Bs[buf][(2 * g + h) * B_CH + (brow0 + j * 32) * 4]);
        };
        load_frags(0, 0);
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
        }
        store_b(buf ^ 1);
        __syncthreads();
        if (restage) {       // every wave has finished the chunk's taps: replace the halo image
            commit_halo();
            __syncthreads();
        }
        t = tn;
        cc = ccn;
    }

    // ---- epilogue (as igemm_kc) ---------------------------------------------------------------------------
    if (accumulate && sub + 1 < n_sub) { __syncthreads(); continue; }
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
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + r;
        if (n >= p.Nn) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = wm * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
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
    __syncthreads();          // the next class of a pair reuses every LDS array
    }   // sub
}

// Eligibility: every class stride 1, 8x8-patchable grid, channel count a multiple of 32,
// 2..25 taps within a 12x12 halo, enough output channels to fill the 64-wide N tile.
bool halo_eligible(const KcParams& p) {
    if (p.sy != 1 || p.sx != 1 || (p.AC & 31) || p.Nn < 64 || p.ksplit > 1) return false;
    for (int c = 0; c < p.n_cls; ++c) {
        const KcClass& k = p.cls[c];
        const bool quad = k.OHc == 4 && k.OWc == 4;
        if (k.T < 2 || (k.M & 63)) return false;
        if (quad && k.M < 4096) return false;       // too few four-image patches to fill the chip: igemm_kc's small tiles win
        if (!quad && ((k.OHc & 7) || (k.OWc & 7))) return false;
        int ymin = 127, ymax = -128, xmin = 127, xmax = -128;
        for (int t = 0; t < k.T; ++t) {
            ymin = k.ty[t] < ymin ? k.ty[t] : ymin; ymax = k.ty[t] > ymax ? k.ty[t] : ymax;
            xmin = k.tx[t] < xmin ? k.tx[t] : xmin; xmax = k.tx[t] > xmax ? k.tx[t] : xmax;
        }
        const int lim = quad ? 2 : 4;       // four 6x6 halos fill the 144-pixel LDS image
        if (ymax - ymin > lim || xmax - xmin > lim) return false;
    }
    return true;
}

int launch_halo(KcParams& p, hipStream_t st) {
    int tm = 0;
    for (int c = 0; c < p.n_cls; ++c) {
        KcClass& k = p.cls[c];
        int ymin = 127, ymax = -128, xmin = 127, xmax = -128;
        for (int t = 0; t < k.T; ++t) {
            ymin = k.ty[t] < ymin ? k.ty[t] : ymin; ymax = k.ty[t] > ymax ? k.ty[t] : ymax;
            xmin = k.tx[t] < xmin ? k.tx[t] : xmin; xmax = k.tx[t] > xmax ? k.tx[t] : xmax;
        }
        const bool quad = k.OHc == 4 && k.OWc == 4;
        const int side = quad ? 4 : 8;
        k.ty_min = ymin; k.tx_min = xmin; k.halo_h = side + ymax - ymin; k.halo_w = side + xmax - xmin;
        k.patch = quad ? 2 : 1;
        k.tile0 = tm;
        tm += (k.M + 127) / 128;
    }
    p.tiles_m = tm;
    p.ksplit = 1;
    // Equal classes (forward convs): 128-wide N tiles.  Unequal classes (9/6/6/4 taps of a 5x5 stride-2 data
    // gradient): 64-wide tiles, and the heaviest class is paired with the lightest in ONE workgroup so every workgroup
    // carries the same number of K steps — when the halved grid still fills the chip.  Measured on the critic's data
    // gradients at 128 / 384 rows (scripts/dgrad_sweep.py): 63 -> 81, 68 -> 75, 76 -> 96, 83 -> 98, 65 -> 79 TF.
    bool same_m = true, same_t = true;
    for (int c = 1; c < p.n_cls; ++c) { same_m = same_m && p.cls[c].M == p.cls[0].M; same_t = same_t && p.cls[c].T == p.cls[0].T; }
    static const int wide_min = [] { const char* e = getenv("CSLGAN_HALO_WIDE_MIN"); return e ? atoi(e) : 0; }();
    bool wide = p.Nn > 64 && same_t && (long long)tm * ((p.Nn + 127) / 128) >= wide_min;
    p.pair_mode = 0;
    if (p.acc_classes) {        // all classes in every workgroup: one class's tiles, balanced by construction
        p.tiles_m = (p.cls[0].M + 127) / 128;
        wide = p.Nn > 64 && (long long)p.tiles_m * ((p.Nn + 127) / 128) >= 256;
    }
    static const int pair_min = [] { const char* e = getenv("CSLGAN_HALO_PAIR_MIN"); return e ? atoi(e) : 256; }();
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
    p.tiles_n = wide ? (p.Nn + 127) / 128 : (p.Nn + 63) / 64;
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n)), block(256);
    bool gen = p.pair_mode != 0 || p.acc_classes != 0;
    for (int c = 0; c < p.n_cls; ++c) gen = gen || p.cls[c].patch == 2 || p.cls[c].ay_mul > 1 || p.cls[c].ax_mul > 1;
    note_kernel("igemm_halo_kernel<%d,%s>", wide ? 128 : 64, gen ? "true" : "false");
    if (wide && gen) hipLaunchKernelGGL((igemm_halo_kernel<128, true>), grid, block, 0, st, p);
    else if (wide) hipLaunchKernelGGL((igemm_halo_kernel<128, false>), grid, block, 0, st, p);
    else if (gen) hipLaunchKernelGGL((igemm_halo_kernel<64, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((igemm_halo_kernel<64, false>), grid, block, 0, st, p);
    return check_launch("igemm_halo_kernel");
}

}  // namespace cslgan
