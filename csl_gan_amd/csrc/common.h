// Shared host/device helpers for libcslgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/cslgan.h"

namespace cslgan {

void set_error(const char* fmt, ...);
// Name of the device kernel most recently launched by the calling thread (printf-style; read back through
// cslgan_last_kernel()): bench.py tags its HIP-event timings with it, so roofline figures name a KERNEL as
// rocprofv3 --kernel-trace lists it, not a C-ABI entry that may dispatch to several.
void note_kernel(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return CSLGAN_ERR_LAUNCH;
    }
    return CSLGAN_OK;
}

#define CSLGAN_REQUIRE(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            cslgan::set_error(__VA_ARGS__);       \
            return CSLGAN_ERR_INVALID_ARG;        \
        }                                         \
    } while (0)

constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for 256-thread blocks; result valid in thread 0.  red: 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = red[0] + red[1] + red[2] + red[3];
    return r;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// Zeroes n_floats floats on the stream with a kernel of this library (clip_kernels.hip).  hipMemsetAsync is not used anywhere:
// as a memset node of a captured HIP graph (ROCm 7.2) a 128-byte fill replayed with one stale dword per 16 bytes
// (scripts/dbg_is6.py: row norms of a finite input came back 7.6e8 / NaN on every fourth row), and a fill kernel is an
// ordinary kernel node.
int zero_floats(float* p, size_t n_floats, hipStream_t st);
__device__ __forceinline__ bool aligned16_dev(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace cslgan
