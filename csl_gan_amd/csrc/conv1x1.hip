// 1x1 convolutions with few input channels (the generator's shortcut convs, DCResNet_models.py:22: C/4 = 32..128 shuffled channels ->
// 64..512 filters) as a stream: y[m][:] = act(x[m][:] W^T + b).  On the generic 128x64 implicit-GEMM tile the whole reduction is one
// K tile, so a launch was prologue + epilogue (24-45 TF, 2.3 TB/s on a layer that only moves bytes).  Here a workgroup walks 128-row
// tiles: the tile's rows sit in LDS (the next tile in registers while this one is multiplied), the filter comes through LDS 64 output
// channels at a time, each wavefront owns 32 rows x 64 outputs (two 32x32 fp32 MFMA tiles, k = 8g + 4h + e on both operands so a
// fragment is one ds_read_b128), bias / activation in the epilogue, 128-byte contiguous stores.  gfx950 only.
#include "common.h"
#include "igemm.h"

namespace cslgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct C1Params {
    const float* x;      // [M][C]
    const float* w;      // [K][C]
    const float* bias;   // [K] or null
    float* y;            // [M][K]
    long long M;
    int K, act, n_tiles;
};

template <int C>
__global__ __launch_bounds__(256, C == 32 ? 3 : (C == 64 ? 2 : 1)) void conv1x1_kernel(const C1Params p) {
    constexpr int LD = C + 4, C4 = C / 4;
    constexpr int AREG = 128 * C4 / 256;                 // float4 of a row tile per thread (4, 8 or 16)
    constexpr int WREG = 64 * C4 / 256;                  // float4 of a 64-filter slice per thread
    __shared__ __attribute__((aligned(16))) float As[128 * LD];
    __shared__ __attribute__((aligned(16))) float Ws[64 * LD];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    float4 ra[AREG];
    auto fetch = [&](int tile) {
#pragma unroll
        for (int j = 0; j < AREG; ++j) {
            const int idx = tid + 256 * j;
            const long long row = (long long)tile * 128 + idx / C4;
            ra[j] = row < p.M ? *reinterpret_cast<const float4*>(p.x + row * C + (idx % C4) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if ((int)blockIdx.x < p.n_tiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        __syncthreads();                                  // the previous tile's reads are done
#pragma unroll
        for (int j = 0; j < AREG; ++j) {
            const int idx = tid + 256 * j;
            *reinterpret_cast<float4*>(&As[(idx / C4) * LD + (idx % C4) * 4]) = ra[j];
        }
        if (tile + (int)gridDim.x < p.n_tiles) fetch(tile + gridDim.x);
        for (int n0 = 0; n0 < p.K; n0 += 64) {
            __syncthreads();                              // As is written / the previous filter slice is consumed
#pragma unroll
            for (int j = 0; j < WREG; ++j) {
                const int idx = tid + 256 * j;
                const int n = idx / C4;
                *reinterpret_cast<float4*>(&Ws[n * LD + (idx % C4) * 4]) =
                    *reinterpret_cast<const float4*>(p.w + (long long)(n0 + n) * C + (idx % C4) * 4);
            }
            __syncthreads();
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
#pragma unroll 4
            for (int g = 0; g < C / 8; ++g) {
                const float4 a = *reinterpret_cast<const float4*>(&As[(wid * 32 + r) * LD + 8 * g + 4 * h]);
                const float4 b0 = *reinterpret_cast<const float4*>(&Ws[r * LD + 8 * g + 4 * h]);
                const float4 b1 = *reinterpret_cast<const float4*>(&Ws[(32 + r) * LD + 8 * g + 4 * h]);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc[1], 0, 0, 0);
            }
            const long long row0 = (long long)tile * 128 + wid * 32 + 4 * h;      // this lane's rows: row0 + (v & 3) + 8 * (v >> 2)
            const int rows_left = (int)(p.M - row0 < 32 ? p.M - row0 : 32);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + j * 32 + r;
                const float bv = p.bias ? p.bias[n] : 0.f;
                float* yj = p.y + row0 * p.K + n;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int ro = (v & 3) + 8 * (v >> 2);
                    if (ro >= rows_left) continue;
                    float val = acc[j][v] + bv;
                    if (p.act == CSLGAN_ACT_LRELU02) val = val > 0.f ? val : 0.2f * val;
                    else if (p.act == CSLGAN_ACT_RELU) val = val > 0.f ? val : 0.f;
                    else if (p.act == CSLGAN_ACT_TANH) val = tanhf(val);
                    yj[ro * p.K] = val;
                }
            }
        }
    }
}

bool conv1x1_eligible(const cslgan_conv_t* c, const float* x, const float* w, const float* residual) {
    static const int env = [] { const char* e = getenv("CSLGAN_CONV1X1"); return e ? atoi(e) : 1; }();
    return env && !residual && c->compute == CSLGAN_COMPUTE_F32 && c->R == 1 && c->S == 1 && c->stride == 1 && c->pad == 0 &&
           // measured at 128 images: 82 -> 60 us (C = 32, 64x64), 60 -> 49 us (C = 64, 32x32); 128 channels on <= 16x16 grids have too
           // few row tiles to stream (42 -> 48 us, 26 -> 82 us) and stay on the generic kernel
           (c->C == 32 || c->C == 64) && (c->K & 63) == 0 && c->K >= 64 && (long long)c->N * c->H * c->W >= 65536 &&
           aligned16(x) && aligned16(w);
}

int launch_conv1x1(const cslgan_conv_t* c, const float* x, const float* w, const float* bias, int act, float* y, hipStream_t st) {
    C1Params p{};
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.act = act; p.K = c->K;
    p.M = (long long)c->N * c->H * c->W;
    const long long nt = (p.M + 127) / 128;
    CSLGAN_REQUIRE(nt < (1ll << 31), "conv2d_fwd (1x1): too many rows");
    p.n_tiles = (int)nt;
    const int per_cu = c->C == 128 ? 1 : (c->C == 64 ? 2 : 3);       // LDS 101 / 52 / 28 KB per workgroup; registers allow 1 / 2 / 3
    const long long cap = 256ll * per_cu;
    const unsigned grid = (unsigned)(nt < cap ? nt : cap);
    note_kernel("conv1x1_kernel<%d>", c->C);
    if (c->C == 32) hipLaunchKernelGGL(conv1x1_kernel<32>, dim3(grid), dim3(256), 0, st, p);
    else if (c->C == 64) hipLaunchKernelGGL(conv1x1_kernel<64>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv1x1_kernel<128>, dim3(grid), dim3(256), 0, st, p);
    return check_launch("conv1x1_kernel");
}

}  // namespace cslgan
