// Per-sample weight-gradient norms of a conv layer WITHOUT forming the gradient ("ghost" norms), for layers
// with at most 64 output pixels per sample.  gfx950 only.
//
//   gW_b = alpha * GY_b^T XU_b          GY_b: [PQ x K] output gradient, XU_b: [PQ x T*C] unfolded input
//   ||gW_b||^2 = alpha^2 * sum_{p,p'} (GY_b GY_b^T)[p,p'] * (XU_b XU_b^T)[p,p']
//
// The critic's last conv has PQ = 16 output pixels and K x T*C = 512 x 6400: the direct product costs
// 2*16*512*6400 = 105 MFLOP per sample, the two PQ x PQ Gram matrices 2*16*16*(512+6400) = 3.5 MFLOP (30x less;
// 3.7x less for the 64-pixel layer before it).  Rows of both operands are contiguous in k, so the loader and the
// MFMA fragment reads are igemm_kc's: 16-byte loads into a [row][32+4] LDS image, ds_read_b128 fragments with
// k = 8g+4h+e (both factors of A A^T use the same k permutation, so the sum over k is unchanged).
//
// One workgroup per sample.  PQ <= 32: one 32x32 tile, the four wavefronts take every fourth 32-deep K chunk and
// their partial Gram matrices are added in LDS.  32 < PQ <= 64: four 32x32 tiles, one per wavefront.
#include "common.h"
#include "igemm.h"

namespace cslgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GramParams {
    const float* gy;     // [N][P][Q][K]
    const float* x;      // [N][H][W][C]
    int N, H, W, C, K, PQ, Q, T, stride;
    int n1, n2;          // 32-deep chunks of GY rows (K/32) and of XU rows (T*C/32)
    float alpha2;
    float* sq;           // [N], accumulated
    signed char ty[IG_MAX_TAPS], tx[IG_MAX_TAPS];
};

constexpr int GR_LD = 36;

template <int TILES>
__global__ __launch_bounds__(256) void gram_sqnorm_kernel(const GramParams p) {
    constexpr int ROWS = TILES == 1 ? 128 : 64;      // LDS rows per buffer: 4 K-chunks x 32 pixels, or 1 chunk x 64 pixels
    constexpr int LPT = ROWS * 8 / 256;              // float4 loads per thread per iteration
    __shared__ __attribute__((aligned(16))) float As[2][ROWS * GR_LD];
    __shared__ float s_red[TILES == 1 ? 2 * 1024 : 4];

    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int cpc = p.C >> 5;                        // 32-channel chunks per tap
    const int n_chunks = p.n1 + p.n2;
    const int n_iter = TILES == 1 ? (n_chunks + 3) / 4 : n_chunks;

    // loader coordinates: row -> (chunk slot, pixel), fixed per thread
    int l_pix[LPT], l_slot[LPT], l_py[LPT], l_px[LPT];
    const int j4 = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int row = (tid >> 3) + 32 * i;
        l_slot[i] = TILES == 1 ? row >> 5 : 0;
        l_pix[i] = TILES == 1 ? row & 31 : row;
        l_py[i] = l_pix[i] / p.Q;
        l_px[i] = l_pix[i] - l_py[i] * p.Q;
    }
    // DEPTH chunk-iterations are in flight in registers: with one workgroup per sample and ~1 us of global latency per
    // gather, a single prefetched iteration left the MFMAs idle nine tenths of the time (rocprofv3: 0.13 ms per launch).
    constexpr int DEPTH = TILES == 1 ? 4 : 6;
    float4 rg[DEPTH][LPT];
    auto load = [&](int it, float4* dst) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int c = TILES == 1 ? 4 * it + l_slot[i] : it;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (l_pix[i] < p.PQ && c < n_chunks) {
                if (c < p.n1) {
                    v = *reinterpret_cast<const float4*>(p.gy + ((long long)b * p.PQ + l_pix[i]) * p.K + c * 32 + j4);
                } else {
                    const int c2 = c - p.n1;
                    const int t = c2 / cpc, cc = c2 - t * cpc;
                    const int iy = l_py[i] * p.stride + p.ty[t], ix = l_px[i] * p.stride + p.tx[t];
                    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                        v = *reinterpret_cast<const float4*>(p.x + (((long long)b * p.H + iy) * p.W + ix) * p.C + cc * 32 + j4);
                }
            }
            dst[i] = v;
        }
    };
    auto store = [&](int buf, const float4* src) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int row = (tid >> 3) + 32 * i;
            *reinterpret_cast<float4*>(&As[buf][row * GR_LD + j4]) = src[i];
        }
    };

    const int lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int rowA = TILES == 1 ? wid * 32 + r : (wid >> 1) * 32 + r;
    const int rowB = TILES == 1 ? wid * 32 + r : (wid & 1) * 32 + r;

    f32x16 acc1, acc2;
#pragma unroll
    for (int v = 0; v < 16; ++v) { acc1[v] = 0.f; acc2[v] = 0.f; }

#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(d, rg[d]);          // past-the-end iterations load zeros
    int buf = 0;
    for (int it0 = 0; it0 < n_iter; it0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int it = it0 + d;
            if (it < n_iter) {                                // uniform
                store(buf, rg[d]);
                load(it + DEPTH, rg[d]);
                __syncthreads();                              // also orders this store after every wave's reads two iterations back
                const int c = TILES == 1 ? 4 * it + wid : it; // this wavefront's chunk
                if (c < n_chunks) {
                    float4 af[4], bf[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        af[g] = *reinterpret_cast<const float4*>(&As[buf][rowA * GR_LD + 8 * g + 4 * h]);
                        bf[g] = *reinterpret_cast<const float4*>(&As[buf][rowB * GR_LD + 8 * g + 4 * h]);
                    }
                    if (c < p.n1) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g].x, bf[g].x, acc1, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g].y, bf[g].y, acc1, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g].z, bf[g].z, acc1, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g].w, bf[g].w, acc1, 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g].x, bf[g].x, acc2, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g].y, bf[g].y, acc2, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g].z, bf[g].z, acc2, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g].w, bf[g].w, acc2, 0, 0, 0);
                        }
                    }
                }
                buf ^= 1;
            }
        }
    }
    __syncthreads();

    float prod = 0.f;
    if (TILES == 1) {
        // add the four wavefronts' partial Gram matrices, then multiply them entry by entry
        for (int i = tid; i < 2 * 1024; i += 256) s_red[i] = 0.f;
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = (v & 3) + 8 * (v >> 2) + 4 * h;
            atomicAdd(&s_red[row * 32 + r], acc1[v]);
            atomicAdd(&s_red[1024 + row * 32 + r], acc2[v]);
        }
        __syncthreads();
        for (int i = tid; i < 1024; i += 256) prod = fmaf(s_red[i], s_red[1024 + i], prod);
        __syncthreads();
        const float tot = block_sum_256(prod, s_red);
        if (tid == 0) atomicAdd(p.sq + b, p.alpha2 * tot);
    } else {
#pragma unroll
        for (int v = 0; v < 16; ++v) prod = fmaf(acc1[v], acc2[v], prod);
        const float tot = block_sum_256(prod, s_red);
        if (tid == 0) atomicAdd(p.sq + b, p.alpha2 * tot);
    }
}


// ---- layers with at most 16 output pixels AND at most 16 input pixels per stride-parity class ------------------
// (the critic's last conv: 8x8x256 -> 4x4x512, 5x5 stride 2).  The unfolded Gram matrix is assembled from the
// pixel-pair Gram matrices of the input instead of being multiplied out tap by tap:
//   (XU XU^T)[p,p'] = sum_t < x[s*p + t], x[s*p' + t] >  and  s*p + t, s*p' + t always share their parity class,
// so with XX_c = X_c X_c^T (X_c: the <= 16 pixels of class c, K = C) the sum over the T taps is T lookups.
// Work per sample: s^2 * 2*16*16*C + 2*16*16*K FLOP (0.8 MFLOP for the last critic conv; the tap-by-tap Gram form
// needs 3.5, the product 105).  One wavefront per parity class on v_mfma_f32_16x16x4_f32 with the operands loaded
// straight into fragment registers (A and B of X X^T are the same registers); K of GY GY^T is split over the four
// wavefronts.  ~100 KB read per sample: the kernel is a stream over gy and x.
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GramSmallParams {
    const float* gy;     // [N][PQ][K]
    const float* x;      // [N][H][W][C]
    int N, H, W, C, K, PQ, Q, T, s;
    float alpha2;
    float* sq;
    signed char tcls[IG_MAX_TAPS], tdy[IG_MAX_TAPS], tdx[IG_MAX_TAPS];   // tap -> parity class, offsets on the class grid
    int Hc[4], Wc[4];    // class grid dims
};

__global__ __launch_bounds__(256) void gram_sqnorm_small_kernel(const GramSmallParams p) {
    __shared__ float XXs[4][16 * 17];
    __shared__ float G1s[16 * 17];
    __shared__ float s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.x;
    const int row = lane & 15, q = lane >> 4;
    for (int i = tid; i < 16 * 17; i += 256) G1s[i] = 0.f;
    __syncthreads();

    // ---- XX_c for class c = wid ---------------------------------------------------------------------------
    const int nc = p.s * p.s;
    f32x4 accx = {0.f, 0.f, 0.f, 0.f};
    if (wid < nc) {
        const int cy = wid / p.s, cx = wid - cy * p.s;
        const int Hc = p.Hc[wid], Wc = p.Wc[wid];
        const int ly = row / Wc, lx = row - ly * Wc;
        const bool ok = row < Hc * Wc;
        const float* src = p.x + (((long long)b * p.H + (ly * p.s + cy)) * p.W + (lx * p.s + cx)) * p.C + 4 * q;
        for (int j0 = 0; j0 < p.C; j0 += 128) {          // 8 float4 per lane in flight
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = (ok && j0 + 16 * j < p.C) ? *reinterpret_cast<const float4*>(src + j0 + 16 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                accx = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j].x, v[j].x, accx, 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j].y, v[j].y, accx, 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j].z, v[j].z, accx, 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j].w, v[j].w, accx, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) XXs[wid][(4 * q + i) * 17 + row] = accx[i];
    }
    // ---- this wavefront's quarter of GY GY^T ------------------------------------------------------------
    {
        f32x4 accg = {0.f, 0.f, 0.f, 0.f};
        const int kq = p.K >> 2;
        const bool ok = row < p.PQ;
        const float* src = p.gy + ((long long)b * p.PQ + row) * p.K + wid * kq + 4 * q;
        for (int j0 = 0; j0 < kq; j0 += 128) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = (ok && j0 + 16 * j < kq) ? *reinterpret_cast<const float4*>(src + j0 + 16 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                accg = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j].x, v[j].x, accg, 0, 0, 0);
                accg = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j].y, v[j].y, accg, 0, 0, 0);
                accg = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j].z, v[j].z, accg, 0, 0, 0);
                accg = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j].w, v[j].w, accg, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(&G1s[(4 * q + i) * 17 + row], accg[i]);
    }
    __syncthreads();
    // ---- sum_{p,p'} G1[p,p'] * sum_t XX_{c(t)}[loc(p,t), loc(p',t)] --------------------------------------
    float prod = 0.f;
    {
        const int pa = tid >> 4, pb = tid & 15;
        if (pa < p.PQ && pb < p.PQ) {
            const int ay = pa / p.Q, ax = pa - ay * p.Q, by = pb / p.Q, bx = pb - by * p.Q;
            float g2 = 0.f;
            for (int t = 0; t < p.T; ++t) {
                const int c = p.tcls[t], Hc = p.Hc[c], Wc = p.Wc[c];
                const int ya = ay + p.tdy[t], xa = ax + p.tdx[t], yb = by + p.tdy[t], xb = bx + p.tdx[t];
                if ((unsigned)ya < (unsigned)Hc && (unsigned)xa < (unsigned)Wc && (unsigned)yb < (unsigned)Hc && (unsigned)xb < (unsigned)Wc)
                    g2 += XXs[c][(ya * Wc + xa) * 17 + yb * Wc + xb];
            }
            prod = G1s[pa * 17 + pb] * g2;
        }
    }
    const float tot = block_sum_256(prod, s_red);
    if (tid == 0) atomicAdd(p.sq + b, p.alpha2 * tot);
}

// ---- layers with at most 64 output pixels AND at most 64 input pixels per stride-parity class ------------------------------
// (the critic's third conv: 16x16x128 -> 8x8x256, 5x5 stride 2).  The same pixel-pair identity as above, on 32x32 MFMA tiles:
// per sample s^2 Gram matrices XX_c = X_c X_c^T (64 x 64 over the C channels) + GY GY^T (64 x 64 over K) = 6.3 MFLOP where the
// tap-by-tap Gram form needs 28 and the product 105 — and no [N][K][R][S][C] gradient is written (419 MB per launch for that
// layer at 128 samples, which made the per-sample product HBM-write bound).  The taps partition by class, so does the sum:
//   ||gW_b||^2 = sum_c sum_{p,p'} (GY GY^T)[p,p'] * sum_{t in c} XX_c[loc(p,t), loc(p',t)]
// One workgroup per (sample, class) — GY GY^T is recomputed per class (2.1 of the 6.3 MFLOP), which buys s^2 times the
// workgroups of a per-sample launch (128 samples alone leave half the chip idle).  Wavefront w owns the tile (w >> 1, w & 1) of
// both matrices; operands stream through one [64 rows][64 k] LDS block (16-byte loads, the next block in registers while the
// MFMAs run); XX_c then goes to LDS and every lane assembles the tap sum for the 16 (p,p') entries it holds of GY GY^T.
constexpr int G64_LD = 68;             // floats per row of the operand block
constexpr int G64_XLD = 65;            // floats per row of the stored XX_c

__global__ __launch_bounds__(256) void gram_sqnorm_cls64_kernel(const GramSmallParams p) {
    __shared__ __attribute__((aligned(16))) float As[64 * G64_LD];
    __shared__ float XXs[64 * G64_XLD];
    __shared__ float s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nc = p.s * p.s;
    const int b = blockIdx.x / nc, c = blockIdx.x - b * nc;
    const int xch = (p.C + 63) >> 6, gch = (p.K + 63) >> 6;
    const int n_chunks = xch + gch;
    const int l_row = tid >> 4, l_k = (tid & 15) * 4;          // loader: rows l_row + 16 i, 4 floats at l_k
    const int Hc = p.Hc[c], Wc = p.Wc[c];
    const int cy = c / p.s, cx = c - cy * p.s;

    const float* src[4];                                       // this thread's four rows of the class image / of gy
    bool xok[4], gok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = l_row + 16 * i;
        const int ly = row / Wc, lx = row - ly * Wc;
        xok[i] = row < Hc * Wc;
        gok[i] = row < p.PQ;
        src[i] = p.x + (((long long)b * p.H + (ly * p.s + cy)) * p.W + (lx * p.s + cx)) * p.C;
    }
    const float* gsrc = p.gy + (long long)b * p.PQ * p.K;
    float4 rg[4];
    auto load = [&](int q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < xch) {
                const int k0 = q * 64 + l_k;
                if (xok[i] && k0 < p.C) v = *reinterpret_cast<const float4*>(src[i] + k0);
            } else if (q < n_chunks) {
                const int k0 = (q - xch) * 64 + l_k;
                if (gok[i] && k0 < p.K) v = *reinterpret_cast<const float4*>(gsrc + (long long)(l_row + 16 * i) * p.K + k0);
            }
            rg[i] = v;
        }
    };
    f32x16 accx, accg;
#pragma unroll
    for (int v = 0; v < 16; ++v) { accx[v] = 0.f; accg[v] = 0.f; }
    const int rowA = (wid >> 1) * 32 + r, rowB = (wid & 1) * 32 + r;

    load(0);
    for (int q = 0; q < n_chunks; ++q) {
        __syncthreads();                                  // every wavefront is done with the previous block
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&As[(l_row + 16 * i) * G64_LD + l_k]) = rg[i];
        __syncthreads();
        load(q + 1);                                      // zeros past the end
        f32x16 a = q < xch ? accx : accg;                 // uniform
#pragma unroll
        for (int g = 0; g < 8; ++g) {                     // k = 8g + 4h + e on both sides
            const float4 af = *reinterpret_cast<const float4*>(&As[rowA * G64_LD + 8 * g + 4 * h]);
            const float4 bf = *reinterpret_cast<const float4*>(&As[rowB * G64_LD + 8 * g + 4 * h]);
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, a, 0, 0, 0);
        }
        if (q < xch) accx = a; else accg = a;
    }
    // ---- XX_c -> LDS -----------------------------------------------------------------------------------------------------
#pragma unroll
    for (int v = 0; v < 16; ++v)
        XXs[((wid >> 1) * 32 + (v & 3) + 8 * (v >> 2) + 4 * h) * G64_XLD + (wid & 1) * 32 + r] = accx[v];
    __syncthreads();
    // ---- sum over this lane's 16 entries (p, p') of GY GY^T[p,p'] * sum_{t in c} XX_c[loc(p,t), loc(p',t)] ------------------
    float prod = 0.f;
    const int pb = (wid & 1) * 32 + r;                    // p' (column), fixed per lane
    const int by = pb / p.Q, bx = pb - by * p.Q;
    int ay[16], ax[16];                                   // p (row) of accumulator entry v
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int pa = (wid >> 1) * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
        ay[v] = pa < p.PQ ? pa / p.Q : -100;
        ax[v] = pa - (pa / p.Q) * p.Q;
    }
    float gx[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) gx[v] = 0.f;
    if (pb < p.PQ) {
        for (int t = 0; t < p.T; ++t) {
            if (p.tcls[t] != c) continue;                 // uniform
            const int dy = p.tdy[t], dx = p.tdx[t];
            const int yb = by + dy, xb = bx + dx;
            if ((unsigned)yb >= (unsigned)Hc || (unsigned)xb >= (unsigned)Wc) continue;
            const float* col = &XXs[yb * Wc + xb];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int ya = ay[v] + dy, xa = ax[v] + dx;
                if ((unsigned)ya < (unsigned)Hc && (unsigned)xa < (unsigned)Wc) gx[v] += col[(ya * Wc + xa) * G64_XLD];
            }
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) prod = fmaf(accg[v], gx[v], prod);
    }
    const float tot = block_sum_256(prod, s_red);
    if (tid == 0) atomicAdd(p.sq + b, p.alpha2 * tot);
}

}  // namespace cslgan

using namespace cslgan;

extern "C" {

int cslgan_conv2d_wgrad_sqnorm_gram_f32(const cslgan_conv_t* c, const float* gy, const float* x, float alpha, float* sq,
                                        void* stream) {
    CSLGAN_REQUIRE(c && gy && x && sq, "wgrad_sqnorm_gram: null argument");
    CSLGAN_REQUIRE(c->N > 0 && c->H > 0 && c->W > 0 && c->R > 0 && c->S > 0 && c->stride > 0 && c->pad >= 0, "wgrad_sqnorm_gram: non-positive dimension");
    
    CSLGAN_REQUIRE(c->R * c->S <= IG_MAX_TAPS, "wgrad_sqnorm_gram: too many taps");
    const int P = (c->H + 2 * c->pad - c->R) / c->stride + 1, Q = (c->W + 2 * c->pad - c->S) / c->stride + 1;
    CSLGAN_REQUIRE(P == c->P && Q == c->Q, "wgrad_sqnorm_gram: output %dx%d does not match P,Q=%d,%d", P, Q, c->P, c->Q);
    CSLGAN_REQUIRE(P * Q >= 1 && P * Q <= 64, "wgrad_sqnorm_gram: needs at most 64 output pixels per sample, got %d", P * Q);
    CSLGAN_REQUIRE(c->K % 32 == 0 && c->C % 32 == 0, "wgrad_sqnorm_gram: K=%d and C=%d must be multiples of 32", c->K, c->C);
    CSLGAN_REQUIRE(aligned16(gy) && aligned16(x), "wgrad_sqnorm_gram: operands must be 16-byte aligned");
    {   // few pixels per stride-parity class: assemble the unfolded Gram matrix from pixel-pair Gram matrices
        const int st = c->stride;
        bool cls_ok = P * Q <= 64 && st >= 1 && st <= 2;
        int max_pix = 0;
        GramSmallParams sp{};
        for (int cls = 0; cls_ok && cls < st * st; ++cls) {
            const int cy = cls / st, cx = cls % st;
            sp.Hc[cls] = c->H > cy ? (c->H - cy + st - 1) / st : 0;
            sp.Wc[cls] = c->W > cx ? (c->W - cx + st - 1) / st : 0;
            const int pix = sp.Hc[cls] * sp.Wc[cls];
            cls_ok = cls_ok && pix >= 1 && pix <= 64;
            max_pix = pix > max_pix ? pix : max_pix;
        }
        const bool small = cls_ok && P * Q <= 16 && max_pix <= 16 && c->K % 64 == 0 && c->C % 16 == 0;
        static const int cls64_env = [] { const char* e = getenv("CSLGAN_GRAM_CLS64"); return e ? atoi(e) : 1; }();
        const bool cls64 = cls_ok && !small && cls64_env && c->K % 4 == 0 && c->C % 4 == 0;
        if (small || cls64) {
            auto fl = [](int v, int d) { return v >= 0 ? v / d : -((-v + d - 1) / d); };
            sp.gy = gy; sp.x = x; sp.N = c->N; sp.H = c->H; sp.W = c->W; sp.C = c->C; sp.K = c->K; sp.PQ = P * Q; sp.Q = Q;
            sp.T = c->R * c->S; sp.s = st; sp.alpha2 = alpha * alpha; sp.sq = sq;
            for (int kh = 0; kh < c->R; ++kh)
                for (int kw = 0; kw < c->S; ++kw) {
                    const int t = kh * c->S + kw, oy = kh - c->pad, ox = kw - c->pad;
                    const int dy = fl(oy, st), dx = fl(ox, st);
                    sp.tdy[t] = (signed char)dy; sp.tdx[t] = (signed char)dx;
                    sp.tcls[t] = (signed char)((oy - dy * st) * st + (ox - dx * st));
                }
            if (small) {
                note_kernel("gram_sqnorm_small_kernel");
                hipLaunchKernelGGL(gram_sqnorm_small_kernel, dim3((unsigned)c->N), dim3(256), 0, (hipStream_t)stream, sp);
                return check_launch("gram_sqnorm_small_kernel");
            }
            note_kernel("gram_sqnorm_cls64_kernel");
            hipLaunchKernelGGL(gram_sqnorm_cls64_kernel, dim3((unsigned)(c->N * st * st)), dim3(256), 0, (hipStream_t)stream, sp);
            return check_launch("gram_sqnorm_cls64_kernel");
        }
    }
    GramParams p{};
    p.gy = gy; p.x = x; p.N = c->N; p.H = c->H; p.W = c->W; p.C = c->C; p.K = c->K; p.PQ = P * Q; p.Q = Q; p.T = c->R * c->S;
    p.stride = c->stride; p.n1 = c->K / 32; p.n2 = p.T * (c->C / 32); p.alpha2 = alpha * alpha; p.sq = sq;
    for (int t = 0; t < IG_MAX_TAPS; ++t) { p.ty[t] = 0; p.tx[t] = 0; }
    for (int kh = 0; kh < c->R; ++kh)
        for (int kw = 0; kw < c->S; ++kw) { p.ty[kh * c->S + kw] = (signed char)(kh - c->pad); p.tx[kh * c->S + kw] = (signed char)(kw - c->pad); }
    const dim3 grid((unsigned)c->N), block(256);
    note_kernel("gram_sqnorm_kernel<%d>", p.PQ <= 32 ? 1 : 4);
    if (p.PQ <= 32) hipLaunchKernelGGL((gram_sqnorm_kernel<1>), grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gram_sqnorm_kernel<4>), grid, block, 0, (hipStream_t)stream, p);
    return check_launch("gram_sqnorm_kernel");
}

}  // extern "C"
