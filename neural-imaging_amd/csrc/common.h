// Shared device/host helpers for the nimg HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nimg.h"

#define NIMG_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return NIMG_ERR_LAUNCH;        \
    } while (0)

namespace nimg {

constexpr int WAVE = 64;

__device__ __forceinline__ float lrelu(float v, float alpha) { return v > 0.0f ? v : alpha * v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// map a possibly out-of-range coordinate according to the padding mode; returns false if it is a zero-pad sample
// mode 0 = zeros (CONSTANT), 1 = SYMMETRIC (edge sample repeated), 2 = REFLECT (edge sample not repeated)
__device__ __forceinline__ bool map_coord(int& g, int size, int pad_mode) {
    if (g >= 0 && g < size) return true;
    if (pad_mode == 1) g = g < 0 ? -1 - g : 2 * size - 1 - g;
    else if (pad_mode == 2) g = g < 0 ? -g : 2 * size - 2 - g;
    else return false;
    return g >= 0 && g < size;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace nimg
