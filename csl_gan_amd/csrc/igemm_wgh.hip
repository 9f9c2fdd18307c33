// fp32 MFMA grouped weight gradient with LDS-resident operands, for 8x8-patchable outputs.  gfx950 only.
//
//   gw[g][m][r][s][c] = alpha * sum_{img in group g} sum_{pixel} GY[img][pixel][m] * X[img][pixel*stride + (r,s) - pad][c]
//
// igemm_mc streams a [16 pixels x 128] slice of each operand per K tile and re-reads the input window once per filter
// tap (32 FLOP per byte moved into LDS; 60-77 TF, the weakest MFMA kernel of the step).  Here a workgroup owns
// 128 output channels x 64 input channels x the S taps of ONE filter row: per 8x8 output patch it stages the 64 x 128
// output-gradient patch and the 8-row input slab that row needs ((8-1)*stride + S columns x 64 channels) in LDS ONCE and
// runs all S taps from it — 2 + S fragment reads for 2*S MFMAs per k step, S x fewer input bytes per MAC.
// Both operands have the pixel (the reduction index) as the slow dimension, so fragments are ds_read_b32 of 32
// consecutive channels: conflict-free, no transposes.  v_mfma_f32_32x32x2_f32, exact fp32; 4 wavefronts = 2 (m) x 2 (c).
#include "common.h"
#include "igemm.h"

namespace cslgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_BM = 128, WG_BC = 64, WG_MAXS = 5;
constexpr int WG_LDX = WG_BC + 4;      // floats per pixel of the input slab in LDS
constexpr int WG_MAXW = 7 * 2 + WG_MAXS;   // staged columns at stride 2

struct WghParams {
    const float* gy;     // [N][P][Q][K]
    const float* x;      // [N][H][W][C]
    float* gw;           // [n_groups][K][R][S][C] or null
    float* sq;           // [n_groups] or null
    int N, H, W, C, P, Q, K, R, S, stride, pad, group, n_groups;
    float alpha;
    int tiles_m, tiles_c, ppi;   // K/128, C/64, patches per image
    int ksplit;                  // workgroups per (group, tile, filter row): they take every ksplit-th patch and add atomically
    int xw;                      // staged columns: 7*stride + S
    int xcd;                     // 1: XCD-aware workgroup order (xcd_remap)
    const float* row_scale;      // nullable [N]: gy of sample n is multiplied by row_scale[n] while it is staged (clip-weighted sums)
    // row blocks with their own outputs (cslgan_conv2d_wgrad_blocks_f32): groups [seg_first[s], seg_first[s+1]) write
    // seg_gw[s] + (g - seg_first[s]) * K*R*S*C (nothing when null) and add their squared norm into seg_sq[s][g - seg_first[s]]
    int n_seg;
    int seg_first[CSLGAN_MAX_WGRAD_BLOCKS + 1];
    float* seg_gw[CSLGAN_MAX_WGRAD_BLOCKS];
    float* seg_sq[CSLGAN_MAX_WGRAD_BLOCKS];
};

// TM = 32-row MFMA tiles per wavefront along m: 2 -> 128 output channels per workgroup, 1 -> 64 (K = 64 layers)
template <int S, int TM>
__global__ __launch_bounds__(256, 2) void igemm_wgh_kernel(const WghParams p) {
    constexpr int BM = 64 * TM;
    constexpr int LDG = BM + 4;               // floats per pixel row of the gy patch in LDS
    __shared__ __attribute__((aligned(16))) float Gs[64 * LDG];
    __shared__ __attribute__((aligned(16))) float Xs[8 * WG_MAXW * WG_LDX];
    __shared__ float s_red[4];

    const int tid = threadIdx.x;
    // XCD-aware order (env CSLGAN_WGH_XCD=0: plain): the workgroups of one group — filter rows x channel tiles — re-read the same gy rows
    // and input slabs; hardware deals consecutive workgroup ids round-robin over the eight XCDs, i.e. eight L2s each fetching that
    // group's tensors from the fabric.  With the remap an XCD owns a contiguous run of logical ids.
    int bid = p.xcd ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int split = bid % p.ksplit; bid /= p.ksplit;
    const int r = bid % p.R; bid /= p.R;
    const int tc = bid % p.tiles_c; bid /= p.tiles_c;
    const int tm = bid % p.tiles_m;
    const int g = bid / p.tiles_m;
    const int m0 = tm * BM, c0 = tc * WG_BC;

    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;

    f32x16 acc[TM][S];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][s][v] = 0.f;

    const int n_patch = p.group * p.ppi;
    const int pq8 = p.Q >> 3;
    const int x_slots = 8 * p.xw * 16;          // float4 slots of the input slab
    for (int pi = split; pi < n_patch; pi += p.ksplit) {
        const int il = pi / p.ppi, pr = pi - il * p.ppi;
        const long long img = (long long)g * p.group + il;
        const int py0 = (pr / pq8) << 3, px0 = (pr - (pr / pq8) * pq8) << 3;
        __syncthreads();                        // every wavefront is done with the previous patch
        // ---- gy patch: 64 pixels x BM channels, 4*TM float4 per thread, staged four at a time ------------------
        constexpr int M4 = BM / 4;             // float4 per pixel
        const float rs = p.row_scale ? p.row_scale[img] : 1.f;
#pragma unroll
        for (int half = 0; half < TM; ++half) {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = tid + 256 * (half * 4 + j);
                const int pix = idx / M4, m4 = idx - pix * M4;
                const long long gpix = (img * p.P + py0 + (pix >> 3)) * p.Q + px0 + (pix & 7);
                v[j] = *reinterpret_cast<const float4*>(p.gy + gpix * p.K + m0 + m4 * 4);
                v[j].x *= rs; v[j].y *= rs; v[j].z *= rs; v[j].w *= rs;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = tid + 256 * (half * 4 + j);
                const int pix = idx / M4, m4 = idx - pix * M4;
                *reinterpret_cast<float4*>(&Gs[pix * LDG + m4 * 4]) = v[j];
            }
        }
        // ---- input slab of filter row r: 8 rows x xw columns x 64 channels ------------------------------------
        for (int base = 0; base < x_slots; base += 256 * 4) {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = base + tid + 256 * j;
                const int c4 = idx & 15, pc = idx >> 4;
                const int row = pc / p.xw, col = pc - row * p.xw;
                const int iy = (py0 + row) * p.stride + r - p.pad, ix = px0 * p.stride - p.pad + col;
                v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < x_slots && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                    v[j] = *reinterpret_cast<const float4*>(p.x + ((img * p.H + iy) * p.W + ix) * p.C + c0 + c4 * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = base + tid + 256 * j;
                if (idx < x_slots) *reinterpret_cast<float4*>(&Xs[(idx >> 4) * WG_LDX + (idx & 15) * 4]) = v[j];
            }
        }
        __syncthreads();
        // ---- 32 k steps of two pixels: 2 + S fragment reads, 2*S MFMAs ------------------------------------------
#pragma unroll 4
        for (int ks = 0; ks < 32; ++ks) {
            const int q = 2 * ks + h;
            const int qy = q >> 3, qx = q & 7;
            float a[TM], b[S];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Gs[q * LDG + wm * 32 * TM + i * 32 + l31];
            const float* xrow = &Xs[(qy * p.xw + qx * p.stride) * WG_LDX + wn * 32 + l31];
#pragma unroll
            for (int s = 0; s < S; ++s) b[s] = xrow[s * WG_LDX];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int s = 0; s < S; ++s)
                    acc[i][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[s], acc[i][s], 0, 0, 0);
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------
    float ss = 0.f;
    const long long row_len = (long long)p.R * S * p.C;      // floats per output channel m
    float* __restrict__ outg = p.gw ? p.gw + (long long)g * p.K * row_len : nullptr;
    float* sqg = p.sq ? p.sq + g : nullptr;
    if (p.n_seg > 0) {
        int sg = 0;
#pragma unroll 1
        while (sg + 1 < p.n_seg && g >= p.seg_first[sg + 1]) ++sg;
        const int gl = g - p.seg_first[sg];
        outg = p.seg_gw[sg] ? p.seg_gw[sg] + (long long)gl * p.K * row_len : nullptr;
        sqg = p.seg_sq[sg] ? p.seg_sq[sg] + gl : nullptr;
    }
    const long long col0 = (long long)r * S * p.C;
    const int c = c0 + wn * 32 + l31;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + wm * 32 * TM + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                const float val = p.alpha * acc[i][s][v];
                ss = fmaf(val, val, ss);
                if (outg) {
                    float* dst = outg + (long long)m * row_len + col0 + (long long)s * p.C + c;
                    if (p.ksplit > 1) atomicAdd(dst, val);
                    else *dst = val;
                }
            }
    if (sqg && p.ksplit <= 1) {
        const float tot = block_sum_256(ss, s_red);
        if (tid == 0) atomicAdd(sqg, tot);
    }
}

bool wgh_eligible(const cslgan_conv_t* c, int out_bf16, const void* gy, const void* x);

// ---- the same contraction with fp32 emulated from three bfloat16 pieces on the bf16 matrix cores (round 4) ----------------------------
// cslgan_conv_t.compute == CSLGAN_COMPUTE_BF16X3 (csrc/igemm_bf16.hip states the arithmetic: x = hi + mid + lo in bfloat16, six exact
// piece products per multiply-add, smallest first, fp32 accumulate — fp32-accurate at 2.67x the fp32 matrix rate).
//
// Same ownership as igemm_wgh_kernel — a workgroup = (group, 64 output channels, 64 input channels, filter row r), all S = 5 taps of the
// row, accumulators 5 x (32 m x 32 c) per wavefront — but the reduction index (pixel) is the SLOW index of both operands while
// v_mfma_f32_32x32x16_bf16 wants 8 consecutive k per lane, and each operand value must be split into its pieces exactly once:
//   * a stage = HALF an 8x8 output patch (4 rows x 8 = 32 pixels = two 16-k MFMA steps): its gy rows [32 px][64 m] and the input slab the
//     filter row needs (4 rows x (7*stride + 5) columns x 64 c) are loaded as fp32 (16-byte loads, range-checked), scaled by the sample's
//     clip weight where one is given (in fp32, BEFORE the split: the same product the fp32 kernel forms), split into three bfloat16
//     pieces and written to LDS as [piece][32-channel plane][pixel][32 ch] — 64-byte rows;
//   * MFMA operands come back through ds_read_b64_tr_b16 (gfx950's transposing LDS read: a 16-lane group addresses 4 pixel rows x 16
//     channels and every lane receives 4 consecutive pixels of ITS channel; two reads = one operand).  The four rows a group reads are
//     consecutive output pixels: consecutive slab pixels at stride 1 (64-byte pitch: bank bases 0/16/32/48) and every other slab pixel
//     at stride 2 (96-byte pitch: 0/48/32/16) — conflict-free in both cases;
//   * the next stage's global loads are issued before the MFMA phase and held in registers across it (7 float4), so a stage is
//     barrier - split/store - barrier - MFMAs, and with two workgroups per CU one multiplies while the other splits.
// Per wavefront and 16-k step: 6 + 30 transposing reads (8 B per lane) for 30 MFMAs.
typedef float x3w_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 x3w_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 x3w_bf16x8 __attribute__((ext_vector_type(8)));
typedef short x3w_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) x3w_s16x4* x3w_lds_ptr;
typedef unsigned int x3w_u32x4 __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ unsigned w_pack(float lo, float hi) {
    const x3w_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, x3w_bf16x2));
}
__device__ __forceinline__ float w_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float w_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ void w_split4(const float4& v, uint2& hi, uint2& mid, uint2& lo) {
    hi = make_uint2(w_pack(v.x, v.y), w_pack(v.z, v.w));
    const float r0 = v.x - w_lo(hi.x), r1 = v.y - w_hi(hi.x), r2 = v.z - w_lo(hi.y), r3 = v.w - w_hi(hi.y);   // exact
    mid = make_uint2(w_pack(r0, r1), w_pack(r2, r3));
    lo = make_uint2(w_pack(r0 - w_lo(mid.x), r1 - w_hi(mid.x)), w_pack(r2 - w_lo(mid.y), r3 - w_hi(mid.y)));
}
__device__ __forceinline__ float4 w_bld(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const x3w_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ x3w_bf16x8 w_tr_pair(const unsigned char* lo_addr, const unsigned char* hi_addr) {
    const x3w_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((x3w_lds_ptr)lo_addr);
    const x3w_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((x3w_lds_ptr)hi_addr);
    const uint4 w = make_uint4(__builtin_bit_cast(uint2, lo).x, __builtin_bit_cast(uint2, lo).y, __builtin_bit_cast(uint2, hi).x, __builtin_bit_cast(uint2, hi).y);
    return __builtin_bit_cast(x3w_bf16x8, w);
}
}  // namespace

// QUAD (stride 2 only): 4x4 output grids (the critic's last conv) — a stage is the 2 x 16 output pixels of TWO consecutive samples, each
// with its own 4-row x 11-column slab; needs an even group (dense and clip-weighted sums; per-sample gradients of that layer are
// never formed: ghost clipping takes its norms from Gram matrices).
template <int STRIDE, bool QUAD = false>
__global__ __launch_bounds__(256, 2) void igemm_x3w_kernel(const WghParams p) {
    constexpr int S = 5;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    constexpr int XW = QUAD ? 3 * STRIDE + S : 7 * STRIDE + S;      // staged columns of the slab (of one sample for QUAD)
    constexpr int XPITCH = STRIDE == 1 ? 64 : 96;      // bytes per slab pixel in a 32-channel plane (see the bank note above)
    constexpr int XPIX = QUAD ? 2 * 4 * XW : 4 * XW;   // slab pixels per stage (4 input rows; two samples for QUAD)
    // +64 B per plane: a 16-lane ds_write_b64 group stores one pixel's 64 B into EACH 32-channel plane, and stores bank on (a/4) mod 32
    // (not mod 64 like the reads): plane strides of 0 mod 128 B put both halves on the same 16 banks — every stage store 2-way
    // conflicted, 26 % of the kernel's LDS-array cycles (profiles/r04_pmc_x3_kernels.txt, first collection).  64 mod 128 B: disjoint.
    constexpr int A_PLANE = 32 * 64 + 64, X_PLANE = XPIX * XPITCH + 64;
    static_assert((A_PLANE / 4) % 32 == 16 && (X_PLANE / 4) % 32 == 16, "plane stride must be 64 mod 128 bytes");
    constexpr int NX = (XPIX * 16 + 255) / 256;        // float4 of the slab per thread
    __shared__ __attribute__((aligned(16))) unsigned char As[3][2][A_PLANE];     // [piece][m half][pixel][32 m]
    __shared__ __attribute__((aligned(16))) unsigned char Xs[3][2][X_PLANE];     // [piece][c half][slab pixel][32 c]
    __shared__ float s_red[4];

    const int tid = threadIdx.x;
    // XCD-aware order (env CSLGAN_WGH_XCD=0: plain): the workgroups of one group — filter rows x channel tiles — re-read the same gy rows
    // and input slabs; hardware deals consecutive workgroup ids round-robin over the eight XCDs, i.e. eight L2s each fetching that
    // group's tensors from the fabric.  With the remap an XCD owns a contiguous run of logical ids.
    int bid = p.xcd ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int split = bid % p.ksplit; bid /= p.ksplit;
    const int r = bid % p.R; bid /= p.R;
    const int tc = bid % p.tiles_c; bid /= p.tiles_c;
    const int tm = bid % p.tiles_m;
    const int g = bid / p.tiles_m;
    const int m0 = tm * 64, c0 = tc * WG_BC;

    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    const int gq = (lane >> 2) & 3, gp = lane & 3, ghalf = (lane >> 4) & 1;

    const __amdgpu_buffer_rsrc_t gy_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gy), 0, (unsigned)(4ll * p.N * p.P * p.Q * p.K), 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (unsigned)(4ll * p.N * p.H * p.W * p.C), 0x00020000);

    f32x16 acc[S];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[s][v] = 0.f;

    // ---- staging coordinates (fixed per thread) --------------------------------------------------------------------------------------
    // gy: 32 px x 16 float4 = 512 -> 2 per thread; x: XPIX px x 16 float4 -> NX per thread
    int a_pix[2], a_m4[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int idx = tid + 256 * j; a_pix[j] = idx >> 4; a_m4[j] = idx & 15; }
    float4 rg[2], rx[NX];
    float rs_next = 1.f;
    const int n_patch = p.group * p.ppi;
    const int pq8 = p.Q >> 3;
    // stage st = 2 * patch + half, this workgroup's patches: split, split + ksplit, ...
    float rs_next2 = 1.f;                              // QUAD: the second sample's clip weight
    auto load_stage = [&](int pi, int half) {
        if (QUAD) {                                    // pi = pair of samples (2 pi, 2 pi + 1) of the group; 32 consecutive gy pixels
            const int img0 = g * p.group + 2 * pi;
            rs_next = p.row_scale ? p.row_scale[img0] : 1.f;
            rs_next2 = p.row_scale ? p.row_scale[img0 + 1] : 1.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                rg[j] = w_bld(gy_rsrc, 4u * ((unsigned)(img0 * 16 + a_pix[j]) * (unsigned)p.K + (unsigned)(m0 + a_m4[j] * 4)));
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const int idx = tid + 256 * j;
                const int c4 = idx & 15, pc = idx >> 4;
                const int smp = pc / (4 * XW), pr2 = pc - smp * 4 * XW;
                const int row = pr2 / XW, col = pr2 - row * XW;
                const int iy = row * STRIDE + r - p.pad, ix = col - p.pad;
                const bool ok = pc < XPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                rx[j] = w_bld(x_rsrc, (4u * (unsigned)((((img0 + smp) * p.H + iy) * p.W + ix) * p.C + c0 + c4 * 4)) | (ok ? 0u : OOB));
            }
            return;
        }
        const int il = pi / p.ppi, pr = pi - il * p.ppi;
        const int img = g * p.group + il;
        const int py0 = ((pr / pq8) << 3) + 4 * half, px0 = (pr - (pr / pq8) * pq8) << 3;
        rs_next = p.row_scale ? p.row_scale[img] : 1.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned gpix = (unsigned)((img * p.P + py0 + (a_pix[j] >> 3)) * p.Q + px0 + (a_pix[j] & 7));
            rg[j] = w_bld(gy_rsrc, 4u * (gpix * (unsigned)p.K + (unsigned)(m0 + a_m4[j] * 4)));
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int idx = tid + 256 * j;
            const int c4 = idx & 15, pc = idx >> 4;
            const int row = pc / XW, col = pc - row * XW;
            const int iy = (py0 + row) * STRIDE + r - p.pad, ix = px0 * STRIDE - p.pad + col;
            const bool ok = pc < XPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            rx[j] = w_bld(x_rsrc, (4u * (unsigned)(((img * p.H + iy) * p.W + ix) * p.C + c0 + c4 * 4)) | (ok ? 0u : OOB));
        }
    };
    auto store_stage = [&](float rs, float rs2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float4 v = rg[j];
            if (QUAD && a_pix[j] >= 16) rs = rs2;      // (j = 1 for every thread: pixels 16..31 are the second sample's)
            v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs;
            uint2 hi, mid, lo;
            w_split4(v, hi, mid, lo);
            const int at = a_pix[j] * 64 + (a_m4[j] & 7) * 8, pl = a_m4[j] >> 3;
            *reinterpret_cast<uint2*>(&As[0][pl][at]) = hi;
            *reinterpret_cast<uint2*>(&As[1][pl][at]) = mid;
            *reinterpret_cast<uint2*>(&As[2][pl][at]) = lo;
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int idx = tid + 256 * j;
            const int c4 = idx & 15, pc = idx >> 4;
            if (pc < XPIX) {
                uint2 hi, mid, lo;
                w_split4(rx[j], hi, mid, lo);
                const int at = pc * XPITCH + (c4 & 7) * 8, pl = c4 >> 3;
                *reinterpret_cast<uint2*>(&Xs[0][pl][at]) = hi;
                *reinterpret_cast<uint2*>(&Xs[1][pl][at]) = mid;
                *reinterpret_cast<uint2*>(&Xs[2][pl][at]) = lo;
            }
        }
    };
    // transposed-read byte offsets of this lane inside a plane, k-step 0, tap 0: [u = which 4-pixel half of the lane's 8 k]
    int a_tr[2], b_tr[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        a_tr[u] = (8 * h + 4 * u + gq) * 64 + 32 * ghalf + 8 * gp;                          // pixel k = 16 ks + 8 h + 4 u + gq
        if (QUAD) b_tr[u] = ((2 * h + u) * XW + gq * STRIDE) * XPITCH + 32 * ghalf + 8 * gp;   // k -> (sample ks, qy = 2 h + u, qx = gq)
        else b_tr[u] = (h * XW + (4 * u + gq) * STRIDE) * XPITCH + 32 * ghalf + 8 * gp;     // k -> (qy = 2 ks + h, qx = 4 u + gq)
    }
    constexpr int KS_STEP = (QUAD ? 4 * XW : 2 * XW) * XPITCH;      // slab bytes between the two k-steps of a stage
    // Ten (k-step, tap) units per stage, six MFMAs each.  The compiler's own schedule read a tap's fragments right in front of its
    // MFMAs and waited for them (lgkmcnt(2..3) a dozen times per stage: ~1.5 k cycles of exposed LDS latency beside 1.9 k of MFMAs,
    // 41-44 % matrix-pipe busy).  Software-pipelined by hand as in igemm_x3h: the NEXT unit's fragments are read right after this
    // unit's first MFMA is issued and pinned there (sched_barrier), so five MFMAs (160 cycles) cover their latency; the second
    // k-step's gy fragments ride along with the first unit.
    auto read_a = [&](int ks, x3w_bf16x8 (&a)[3]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a[c] = w_tr_pair(&As[c][wm][a_tr[0] + ks * 16 * 64], &As[c][wm][a_tr[1] + ks * 16 * 64]);
    };
    auto read_b = [&](int ks, int s, x3w_bf16x8 (&b)[3]) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            b[c] = w_tr_pair(&Xs[c][wn][b_tr[0] + ks * KS_STEP + s * XPITCH], &Xs[c][wn][b_tr[1] + ks * KS_STEP + s * XPITCH]);
    };
    auto mma_stage = [&]() {
        x3w_bf16x8 a0[3], a1[3], b0[3], b1[3];
        read_a(0, a0);
        read_b(0, 0, b0);
#pragma unroll
        for (int i = 0; i < 2 * S; ++i) {
            const int ks = i / S, s = i - ks * S;
            x3w_bf16x8 (&a)[3] = ks ? a1 : a0;
            x3w_bf16x8 (&b)[3] = (i & 1) ? b1 : b0;
            x3w_bf16x8 (&bn)[3] = (i & 1) ? b0 : b1;
            f32x16 t = acc[s];                       // smallest terms first
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], t, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < 2 * S) read_b((i + 1) / S, (i + 1) % S, bn);
            if (i == 0) read_a(1, a1);
            __builtin_amdgcn_sched_barrier(0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], t, 0, 0, 0);
            acc[s] = t;
        }
    };

    // this workgroup's stages (0 when split is past the end): patches split, split + ksplit, ... (two halves each), or sample pairs
    const int n_units = QUAD ? p.group >> 1 : n_patch;
    const int n_stage = (QUAD ? 1 : 2) * ((n_units - split + p.ksplit - 1) / p.ksplit);
    if (n_stage > 0) load_stage(split, 0);
#pragma unroll 1
    for (int st = 0; st < n_stage; ++st) {
        const float rs = rs_next, rs2 = rs_next2;
        __syncthreads();                        // every wavefront is done with the previous stage's images
        store_stage(rs, rs2);
        __syncthreads();
        if (st + 1 < n_stage) {
            if (QUAD) load_stage(split + (st + 1) * p.ksplit, 0);
            else load_stage(split + ((st + 1) >> 1) * p.ksplit, (st + 1) & 1);
        }
        mma_stage();
    }

    // ---- epilogue (as igemm_wgh_kernel, one 32-row tile per wavefront along m) -------------------------------------------------------------
    float ss = 0.f;
    const long long row_len = (long long)p.R * S * p.C;      // floats per output channel m
    float* __restrict__ outg = p.gw ? p.gw + (long long)g * p.K * row_len : nullptr;
    float* sqg = p.sq ? p.sq + g : nullptr;
    if (p.n_seg > 0) {
        int sg = 0;
#pragma unroll 1
        while (sg + 1 < p.n_seg && g >= p.seg_first[sg + 1]) ++sg;
        const int gl = g - p.seg_first[sg];
        outg = p.seg_gw[sg] ? p.seg_gw[sg] + (long long)gl * p.K * row_len : nullptr;
        sqg = p.seg_sq[sg] ? p.seg_sq[sg] + gl : nullptr;
    }
    const long long col0 = (long long)r * S * p.C;
    const int c = c0 + wn * 32 + l31;
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int m = m0 + wm * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
            const float val = p.alpha * acc[s][v];
            ss = fmaf(val, val, ss);
            if (outg) {
                float* dst = outg + (long long)m * row_len + col0 + (long long)s * p.C + c;
                if (p.ksplit > 1) atomicAdd(dst, val);
                else *dst = val;
            }
        }
    if (sqg && p.ksplit <= 1) {
        const float tot = block_sum_256(ss, s_red);
        if (tid == 0) atomicAdd(sqg, tot);
    }
}

// Shapes the three-piece form takes: what igemm_wgh takes with 5 filter columns, both tensors below 4 GB (32-bit buffer offsets).
bool x3w_eligible(const cslgan_conv_t* c, int out_bf16, const void* gy, const void* x) {
    static const int env = [] { const char* e = getenv("CSLGAN_X3W"); return e ? atoi(e) : 1; }();
    return env && wgh_eligible(c, out_bf16, gy, x) && c->S == 5 && 4ll * c->N * c->P * c->Q * c->K < 0xFFFFFFF0ll &&
           4ll * c->N * c->H * c->W * c->C < 0xFFFFFFF0ll;
}

// The 4x4-output form (the critic's last conv): stride 2, 5 filter columns, pad 2, 8x8 input, channel counts multiples of 64, an even
// number of samples per group.
bool x3w_quad_eligible(const cslgan_conv_t* c, int group, int out_bf16, const void* gy, const void* x) {
    static const int env = [] { const char* e = getenv("CSLGAN_X3W_QUAD"); return e ? atoi(e) : 1; }();
    return env && !out_bf16 && c->stride == 2 && c->S == 5 && c->R <= 5 && c->P == 4 && c->Q == 4 && c->H == 8 && c->W == 8 && c->pad == 2 &&
           c->K % 64 == 0 && c->C % WG_BC == 0 && group % 2 == 0 && aligned16(gy) && aligned16(x) &&
           4ll * c->N * 16 * c->K < 0xFFFFFFF0ll && 4ll * c->N * 64 * c->C < 0xFFFFFFF0ll;
}

int sqnorm_rows_accumulate(const float* in, long long n_rows, long long len, float* sq_accum, hipStream_t st);   // clip_kernels.hip

// Shapes this kernel takes (the rest stays on igemm_mc).
bool wgh_eligible(const cslgan_conv_t* c, int out_bf16, const void* gy, const void* x) {
    return !out_bf16 && (c->stride == 1 || c->stride == 2) && c->S >= 2 && c->S <= WG_MAXS && c->K % 64 == 0 &&
           c->C % WG_BC == 0 && (c->P & 7) == 0 && (c->Q & 7) == 0 && aligned16(gy) && aligned16(x);
}

int launch_wgh(const cslgan_conv_t* c, const float* gy, const float* x, int group, float alpha, float* gw, float* sq, hipStream_t st,
               const float* row_scale, int n_seg, const int* seg_first, float* const* seg_gw, float* const* seg_sq) {
    const bool x3 = c->compute == CSLGAN_COMPUTE_BF16X3;      // the caller has checked x3w_eligible / x3w_quad_eligible
    const bool quad = x3 && c->P == 4 && c->Q == 4;
    WghParams p{};
    p.gy = gy; p.x = x; p.gw = gw; p.sq = sq; p.row_scale = row_scale;
    p.n_seg = n_seg;
    for (int i = 0; i < n_seg; ++i) { p.seg_first[i] = seg_first[i]; p.seg_gw[i] = seg_gw[i]; p.seg_sq[i] = seg_sq[i]; }
    if (n_seg > 0) { p.seg_first[n_seg] = c->N / group; p.gw = nullptr; p.sq = nullptr; }
    p.N = c->N; p.H = c->H; p.W = c->W; p.C = c->C; p.P = c->P; p.Q = c->Q; p.K = c->K; p.R = c->R; p.S = c->S;
    p.stride = c->stride; p.pad = c->pad; p.group = group; p.n_groups = c->N / group; p.alpha = alpha;
    // K = 64, 192, ...: 64-channel m tiles (forcing them on K = 128 to even out 640 workgroups over 512 slots was measured:
    // 176 -> 171 us on conv2, 210 -> 235 us on conv3 — not kept)
    const bool half_m = x3 || c->K % WG_BM != 0;          // the three-piece kernel owns 64 output channels per workgroup
    p.tiles_m = half_m ? c->K / 64 : c->K / WG_BM; p.tiles_c = c->C / WG_BC; p.ppi = (c->P >> 3) * (c->Q >> 3);
    p.xw = 7 * c->stride + c->S;
    const long long base = (long long)p.n_groups * p.tiles_m * p.tiles_c * c->R;
    const int n_patch = quad ? group / 2 : group * p.ppi;       // units the patch loop walks (sample pairs for the 4x4 form)
    p.ksplit = 1;
    if (gw && n_seg == 0 && base < 768 && n_patch >= 8) {      // few tiles, long patch loops: split the patches, add atomically
        long long want = (1280 + base - 1) / base;
        const long long cap = n_patch / 4;
        p.ksplit = (int)(want < cap ? want : cap);
        if (p.ksplit < 1) p.ksplit = 1;
    }
    const size_t out_floats = (size_t)p.n_groups * c->K * c->R * c->S * c->C;
    if (p.ksplit > 1) {
        if (int rc = zero_floats(gw, out_floats, st)) return rc;
    }
    const long long nb = base * p.ksplit;
    if (nb > 0x7fffffffll) { set_error("wgrad: grid too large"); return CSLGAN_ERR_INVALID_ARG; }
    static const int xcd_env = [] { const char* e = getenv("CSLGAN_WGH_XCD"); return e ? atoi(e) : 1; }();
    p.xcd = xcd_env;
    const dim3 grid((unsigned)nb), block(256);
    if (x3) {
        note_kernel(quad ? "igemm_x3w_kernel<%d,quad>" : "igemm_x3w_kernel<%d>", c->stride);
        if (quad) hipLaunchKernelGGL((igemm_x3w_kernel<2, true>), grid, block, 0, st, p);
        else if (c->stride == 1) hipLaunchKernelGGL(igemm_x3w_kernel<1>, grid, block, 0, st, p);
        else hipLaunchKernelGGL(igemm_x3w_kernel<2>, grid, block, 0, st, p);
        int rc = check_launch("igemm_x3w_kernel");
        if (rc) return rc;
        if (p.ksplit > 1 && sq) rc = sqnorm_rows_accumulate(gw, p.n_groups, (long long)c->K * c->R * c->S * c->C, sq, st);
        return rc;
    }
    note_kernel("igemm_wgh_kernel<%d,%d>", c->S < 2 ? 2 : c->S, half_m ? 1 : 2);
    if (half_m) {
        switch (c->S) {
            case 2: hipLaunchKernelGGL((igemm_wgh_kernel<2, 1>), grid, block, 0, st, p); break;
            case 3: hipLaunchKernelGGL((igemm_wgh_kernel<3, 1>), grid, block, 0, st, p); break;
            case 4: hipLaunchKernelGGL((igemm_wgh_kernel<4, 1>), grid, block, 0, st, p); break;
            default: hipLaunchKernelGGL((igemm_wgh_kernel<5, 1>), grid, block, 0, st, p); break;
        }
    } else {
        switch (c->S) {
            case 2: hipLaunchKernelGGL((igemm_wgh_kernel<2, 2>), grid, block, 0, st, p); break;
            case 3: hipLaunchKernelGGL((igemm_wgh_kernel<3, 2>), grid, block, 0, st, p); break;
            case 4: hipLaunchKernelGGL((igemm_wgh_kernel<4, 2>), grid, block, 0, st, p); break;
            default: hipLaunchKernelGGL((igemm_wgh_kernel<5, 2>), grid, block, 0, st, p); break;
        }
    }
    int rc = check_launch("igemm_wgh_kernel");
    if (rc) return rc;
    if (p.ksplit > 1 && sq) rc = sqnorm_rows_accumulate(gw, p.n_groups, (long long)c->K * c->R * c->S * c->C, sq, st);
    return rc;
}

}  // namespace cslgan
