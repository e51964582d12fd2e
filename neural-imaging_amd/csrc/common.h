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
    // branch-free (selects only): this sits in the tile-staging address arithmetic of every convolution kernel, where
    // a chain of data-dependent branches per coordinate costs more issue slots than the loads it guards
    const bool inside = (unsigned)g < (unsigned)size;
    const int sym = g < 0 ? -1 - g : 2 * size - 1 - g;
    const int ref = g < 0 ? -g : 2 * size - 2 - g;
    const int m = pad_mode == 1 ? sym : ref;
    const bool mapped = pad_mode != 0 && (unsigned)m < (unsigned)size;
    g = inside ? g : (mapped ? m : g);
    return inside || mapped;
}

// XCD-aware workgroup order.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2; the
// convolution kernels number their work items so that neighbours share operands (the output-channel tiles of one spatial
// tile, halo-sharing spatial tiles, the channel blocks of one split-K slab).  Giving every XCD one contiguous range of
// items lets those neighbours meet in one L2 instead of fetching the operand from HBM once per XCD (+2.8 % step
// throughput in bf16 mode, measured A/B on one box; neutral in float32 mode).
__device__ __forceinline__ int xcd_order(int bid) {
#ifdef NIMG_NO_XCD_ORDER
    return bid;
#else
    const int g = (int)gridDim.x, x = bid & 7, r = g & 7;        // XCD x runs ids x, x + 8, ...: g / 8 of them (+1 if x < g % 8)
    return x * (g >> 3) + (x < r ? x : r) + (bid >> 3);
#endif
}

typedef float nimg_f32x16 __attribute__((ext_vector_type(16)));

// Epilogue of the MFMA convolution kernels.  A 32x32 accumulator tile leaves every lane with 16 values of ONE output
// channel (4 B per lane per store instruction = store-issue bound, ~1.3 TB/s measured).  The tile is therefore turned
// around through LDS - written channel-contiguous per pixel (conflict-free ds_write_b32), read back as float4 along the
// NHWC channel axis - so every lane stores 16 B and 16 lanes cover one pixel's 64 channels contiguously.
//   acc:  NI accumulators of M-fragment `mi` (32 pixels x NI*32 channels);  lds: this wave's scratch, 32*(NI*32+4) floats
//   emit(row, c, float4 v): row = pixel index inside the fragment (0..31), c = first of 4 channels inside the wave's
//                           NI*32-channel strip; applies bias/activation/mask and stores.
// Every wave of the workgroup must call this the same number of times (it contains workgroup barriers).
constexpr int EPI_PAD = 4;
// BLOCK_SYNC = true: the scratch aliases tiles other waves may still be reading -> workgroup barriers.
// BLOCK_SYNC = false: private per-wave scratch; a wave's LDS operations complete in order, so only the compiler has to
// be kept from reordering (wave_barrier) - no workgroup barrier in a latency-bound tile loop.
template <int NI, bool BLOCK_SYNC = true, typename Emit>
__device__ __forceinline__ void epilogue_via_lds(const nimg_f32x16 (&acc)[NI], float* lds, int lane, Emit emit) {
    constexpr int RS = NI * 32 + EPI_PAD;
    const int half = lane >> 5, n = lane & 31;
    if (BLOCK_SYNC) __syncthreads();                   // LDS region free (previous user done)
    else __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int j = 0; j < 16; ++j) lds[((j & 3) + 8 * (j >> 2) + 4 * half) * RS + ni * 32 + n] = acc[ni][j];
    if (BLOCK_SYNC) __syncthreads();
    else __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < (32 * NI * 8) / 64; ++it) {
        const int idx = it * 64 + lane, row = idx / (NI * 8), c4 = idx % (NI * 8);
        const float4 v = *reinterpret_cast<const float4*>(lds + row * RS + c4 * 4);
        emit(row, c4 * 4, v);
    }
}

// Fused bias + LeakyReLU + 2x2/2 max-pool epilogue (FAN: conv -> lrelu -> MaxPool2D, models/forensics.py:69-77) for a
// fragment whose 32 rows are two adjacent 16-pixel tile rows (TW == 16): the window of pooled column pc is rows
// {2pc, 2pc+1, 16+2pc, 17+2pc}.  Same LDS turn-around as above; one lane finishes 4 channels of one pooled pixel.
//   emit(pc, c, float4 pooled, uchar4 argmax): argmax = position in the window, row-major, first maximum wins
//   (the tie rule of nimg_maxpool2_fwd); bias4(c) returns the 4 biases of strip-local channel c.
template <int NI, bool BLOCK_SYNC = true, typename Bias, typename Emit>
__device__ __forceinline__ void pool_via_lds(const nimg_f32x16 (&acc)[NI], float* lds, int lane, float alpha, Bias bias4,
                                             Emit emit) {
    constexpr int RS = NI * 32 + EPI_PAD;
    const int half = lane >> 5, n = lane & 31;
    if (BLOCK_SYNC) __syncthreads();
    else __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int j = 0; j < 16; ++j) lds[((j & 3) + 8 * (j >> 2) + 4 * half) * RS + ni * 32 + n] = acc[ni][j];
    if (BLOCK_SYNC) __syncthreads();
    else __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int idx = it * 64 + lane, pc = idx / (NI * 8), c = (idx % (NI * 8)) * 4;
        const float4 b = bias4(c);
        float m[4];
        unsigned char k[4];
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            const float4 r = *reinterpret_cast<const float4*>(lds + (2 * pc + (pos & 1) + 16 * (pos >> 1)) * RS + c);
            const float v[4] = {lrelu(r.x + b.x, alpha), lrelu(r.y + b.y, alpha), lrelu(r.z + b.z, alpha),
                                lrelu(r.w + b.w, alpha)};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (pos == 0 || v[e] > m[e]) { m[e] = v[e]; k[e] = (unsigned char)pos; }
        }
        emit(pc, c, make_float4(m[0], m[1], m[2], m[3]), make_uchar4(k[0], k[1], k[2], k[3]));
    }
}

// Lane-local form of pool_via_lds (same arithmetic, same tie rule, bit-identical results).  A lane's 16 accumulator registers
// are rows {0-3, 8-11, 16-19, 24-27} (+4 for lanes 32-63) of ONE output channel: with the fragment's 32 rows = two 16-pixel
// tile rows these are the COMPLETE 2x2 windows of pooled columns {0, 1, 4, 5} (+2) - registers (2q, 2q+1, 2q+8, 2q+9) for the
// lane's window q.  Bias, activation, maximum and arg-max therefore need no other lane; only the 4 pooled values and 4 arg-max
// bytes per fragment (instead of 16 values) go through LDS to come back as channel-contiguous vectors.
//   lds: this wave's scratch, >= 8 * (NI*32 + EPI_PAD) floats + 8 * NI*32 bytes;  bias1(c): bias of strip-local channel c
template <int NI, typename Bias, typename Emit>
__device__ __forceinline__ void pool_in_regs(const nimg_f32x16 (&acc)[NI], float* lds, int lane, float alpha, Bias bias1,
                                             Emit emit) {
    constexpr int RS = NI * 32 + EPI_PAD;
    unsigned char* lidx = reinterpret_cast<unsigned char*>(lds + 8 * RS);
    const int half = lane >> 5, n = lane & 31;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const float b = bias1(ni * 32 + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pc = (q & 1) + 4 * (q >> 1) + 2 * half;
            const float v0 = lrelu(acc[ni][2 * q] + b, alpha), v1 = lrelu(acc[ni][2 * q + 1] + b, alpha);
            const float v2 = lrelu(acc[ni][2 * q + 8] + b, alpha), v3 = lrelu(acc[ni][2 * q + 9] + b, alpha);
            float m = v0;
            unsigned char k = 0;
            if (v1 > m) { m = v1; k = 1; }
            if (v2 > m) { m = v2; k = 2; }
            if (v3 > m) { m = v3; k = 3; }
            lds[pc * RS + ni * 32 + n] = m;
            lidx[pc * (NI * 32) + ni * 32 + n] = k;
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int idx = it * 64 + lane, pc = idx / (NI * 8), c = (idx % (NI * 8)) * 4;
        const float4 v = *reinterpret_cast<const float4*>(lds + pc * RS + c);
        const uchar4 k = *reinterpret_cast<const uchar4*>(lidx + pc * (NI * 32) + c);
        emit(pc, c, v, k);
    }
}

// The two epilogues above for bf16-STORED outputs, eight channels per lane: one 16-byte store (and one 16-byte mask load)
// instead of two 8-byte ones - the store-issue rate, not the bytes, bounds a row-per-lane epilogue.  Same arithmetic per value.
//   emit(row, c, float4 lo, float4 hi): channels c .. c+3 and c+4 .. c+7
template <int NI, typename Emit>
__device__ __forceinline__ void epilogue_via_lds8(const nimg_f32x16 (&acc)[NI], float* lds, int lane, Emit emit) {
    constexpr int RS = NI * 32 + EPI_PAD;
    const int half = lane >> 5, n = lane & 31;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int j = 0; j < 16; ++j) lds[((j & 3) + 8 * (j >> 2) + 4 * half) * RS + ni * 32 + n] = acc[ni][j];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < (32 * NI * 4) / 64; ++it) {
        const int idx = it * 64 + lane, row = idx / (NI * 4), c8 = idx % (NI * 4);
        const float4 lo = *reinterpret_cast<const float4*>(lds + row * RS + c8 * 8);
        const float4 hi = *reinterpret_cast<const float4*>(lds + row * RS + c8 * 8 + 4);
        emit(row, c8 * 8, lo, hi);
    }
}

//   emit(pc, c, float4 lo, float4 hi, uint2 argmax bytes of the 8 channels)
template <int NI, typename Bias, typename Emit>
__device__ __forceinline__ void pool_in_regs8(const nimg_f32x16 (&acc)[NI], float* lds, int lane, float alpha, Bias bias1,
                                              Emit emit) {
    constexpr int RS = NI * 32 + EPI_PAD;
    unsigned char* lidx = reinterpret_cast<unsigned char*>(lds + 8 * RS);
    const int half = lane >> 5, n = lane & 31;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const float b = bias1(ni * 32 + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pc = (q & 1) + 4 * (q >> 1) + 2 * half;
            // the window is pooled on the raw sums, bias + activation go to the winner: x -> lrelu(x + b) is non-decreasing, so the
            // pooled VALUE is bit-identical to pooling the activated values (12 instead of 21 VALU per window); the arg-max can
            // differ only where two unequal sums round to the same activated value (the gradient then takes another, equal, entry)
#ifdef NIMG_POOL_ACT_FIRST                 // A/B: the activated values compared (the form of pool_in_regs)
            const float v0 = lrelu(acc[ni][2 * q] + b, alpha), v1 = lrelu(acc[ni][2 * q + 1] + b, alpha);
            const float v2 = lrelu(acc[ni][2 * q + 8] + b, alpha), v3 = lrelu(acc[ni][2 * q + 9] + b, alpha);
#else
            const float v0 = acc[ni][2 * q], v1 = acc[ni][2 * q + 1], v2 = acc[ni][2 * q + 8], v3 = acc[ni][2 * q + 9];
#endif
            float m = v0;
            unsigned char k = 0;
            if (v1 > m) { m = v1; k = 1; }
            if (v2 > m) { m = v2; k = 2; }
            if (v3 > m) { m = v3; k = 3; }
#ifdef NIMG_POOL_ACT_FIRST
            lds[pc * RS + ni * 32 + n] = m;
#else
            lds[pc * RS + ni * 32 + n] = lrelu(m + b, alpha);
#endif
            lidx[pc * (NI * 32) + ni * 32 + n] = k;
        }
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int ITEMS = 8 * NI * 4;                    // 8 pooled columns x NI*4 groups of 8 channels
#pragma unroll
    for (int it = 0; it * 64 < ITEMS; ++it) {
        const int idx = it * 64 + lane, pc = idx / (NI * 4), c = (idx % (NI * 4)) * 8;
        if (ITEMS % 64 != 0 && idx >= ITEMS) break;
        const float4 lo = *reinterpret_cast<const float4*>(lds + pc * RS + c);
        const float4 hi = *reinterpret_cast<const float4*>(lds + pc * RS + c + 4);
        const uint2 k = *reinterpret_cast<const uint2*>(lidx + pc * (NI * 32) + c);
        emit(pc, c, lo, hi, k);
    }
}

// dst[i] = sum_k partial[k][i] over `splits` slabs, in a fixed order => deterministic (split-K weight gradients, fused
// bias sums).  A workgroup of 256 covers 16 float4 columns x 16 slab segments: thread (col, seg) adds slabs seg, seg+16,
// ... (two independent accumulators), then the 16 segment sums are added in order through LDS - so thousands of slabs
// of a small filter are reduced with 16-way parallelism per column instead of one serial chain per thread.
__device__ __forceinline__ void reduce_slabs(const float* __restrict__ partial, float* __restrict__ dst, long count,
                                             int splits, int accumulate, long block, long nblocks) {
    __shared__ float4 red4[256];
    const int tid = threadIdx.x, col = tid & 15, seg = tid >> 4;
    if ((count & 3) == 0) {
        const long c4 = count >> 2, ncb = (c4 + 15) >> 4;
        for (long cb = block; cb < ncb; cb += nblocks) {
            const long i = cb * 16 + col;
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
            if (i < c4) {
                const float4* src = reinterpret_cast<const float4*>(partial) + i;
                int k = seg;
                // eight slabs requested before the first is added (two per round trip left the reduction waiting on memory
                // latency: 16 dependent round trips for 512 slabs); the additions keep their order - a0 takes slabs k, k + 32,
                // ..., a1 the ones in between - so the sums are bit-identical to the two-at-a-time loop
#ifndef NIMG_SLABS_PIPE_OFF
                for (; k + 112 < splits; k += 128) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = src[(long)(k + 16 * u) * c4];
#pragma unroll
                    for (int u = 0; u < 8; u += 2) {
                        a0.x += v[u].x; a0.y += v[u].y; a0.z += v[u].z; a0.w += v[u].w;
                        a1.x += v[u + 1].x; a1.y += v[u + 1].y; a1.z += v[u + 1].z; a1.w += v[u + 1].w;
                    }
                }
#endif
                for (; k + 16 < splits; k += 32) {
                    const float4 v0 = src[(long)k * c4], v1 = src[(long)(k + 16) * c4];
                    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
                    a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
                }
                if (k < splits) {
                    const float4 v = src[(long)k * c4];
                    a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
                }
            }
            __syncthreads();
            red4[tid] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
            __syncthreads();
            if (seg == 0 && i < c4) {
                float4 r = red4[col];
#pragma unroll
                for (int sgm = 1; sgm < 16; ++sgm) {
                    const float4 v = red4[sgm * 16 + col];
                    r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
                }
                float4* d = reinterpret_cast<float4*>(dst) + i;
                if (accumulate) { const float4 o = *d; r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w; }
                *d = r;
            }
        }
        return;
    }
    for (long i = block * 256 + tid; i < count; i += nblocks * 256) {
        float sacc = 0.f;
        for (int k = 0; k < splits; ++k) sacc += partial[(long)k * count + i];
        dst[i] = accumulate ? dst[i] + sacc : sacc;
    }
}

// one launch, two reductions: workgroups [0, blocks1) reduce the weight slabs, the rest the bias partials
// (internal linkage: every translation unit that includes this header gets its own copy)
static __global__ __launch_bounds__(256) void reduce_slabs2_kernel(const float* __restrict__ p1, float* __restrict__ d1, long n1,
                                                            int splits1, int blocks1, const float* __restrict__ p2,
                                                            float* __restrict__ d2, long n2, int splits2,
                                                            int accumulate) {
    if ((int)blockIdx.x < blocks1) reduce_slabs(p1, d1, n1, splits1, accumulate, blockIdx.x, blocks1);
    else reduce_slabs(p2, d2, n2, splits2, accumulate, blockIdx.x - blocks1, gridDim.x - blocks1);
}

static inline int reduce_grid(long count) {
    const long g = (count & 3) == 0 ? (count / 4 + 15) / 16 : (count + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// launch helper: dw slabs (+ optional db partials) in one kernel
static inline void launch_reduce2(const float* p1, float* d1, long n1, int splits1, const float* p2, float* d2, long n2,
                                  int splits2, int accumulate, hipStream_t s) {
    const int b1 = reduce_grid(n1), b2 = (p2 && d2) ? reduce_grid(n2) : 0;
    hipLaunchKernelGGL(reduce_slabs2_kernel, dim3(b1 + b2), dim3(256), 0, s, p1, d1, n1, splits1, b1, p2, d2, n2, splits2,
                       accumulate);
}

// ---- batched form: the slab reductions of MANY weight gradients in one launch (~30 launches of 4 - 19 us per
// training step, most of them latency, not bytes).  A weight-gradient entry point called in DEFERRED mode launches its partial-sum
// kernel only and describes the reduction it still owes in a ReduceEntry (host memory); nimg_reduce_slabs_batch() later runs up to
// REDUCE_BATCH_MAX of them as one grid - the entries travel BY VALUE as kernel arguments (no device table, nothing to copy,
// capturable in a HIP graph), every workgroup finds its entry in a prefix table and runs reduce_slabs() exactly as
// reduce_slabs2_kernel would: the sums are bit-identical to the per-layer launches.
struct ReduceEntry {
    const float* p1; float* d1; long n1;          // dw slabs -> dw
    const float* p2; float* d2; long n2;          // db partials -> db (or null)
    int splits1, splits2, accumulate, blocks1;    // blocks1 = reduce_grid(n1) (filled by the producer)
};
static_assert(sizeof(ReduceEntry) == 64, "ReduceEntry is part of the C ABI (nimg_reduce_entry_bytes)");
constexpr int REDUCE_BATCH_MAX = 40;
struct ReduceBatch {
    ReduceEntry e[REDUCE_BATCH_MAX];
    int first_block[REDUCE_BATCH_MAX + 1];
    int n;
};
static __global__ __launch_bounds__(256) void reduce_slabs_batch_kernel(const ReduceBatch b) {
    int k = 0;
    while (k + 1 < b.n && (int)blockIdx.x >= b.first_block[k + 1]) ++k;       // <= 40 scalar compares
    const ReduceEntry& e = b.e[k];
    const int blk = (int)blockIdx.x - b.first_block[k], total = b.first_block[k + 1] - b.first_block[k];
    if (blk < e.blocks1) reduce_slabs(e.p1, e.d1, e.n1, e.splits1, e.accumulate, blk, e.blocks1);
    else reduce_slabs(e.p2, e.d2, e.n2, e.splits2, e.accumulate, blk - e.blocks1, total - e.blocks1);
}
static inline void fill_reduce_entry(ReduceEntry* e, const float* p1, float* d1, long n1, int splits1, const float* p2, float* d2,
                                     long n2, int splits2, int accumulate) {
    e->p1 = p1; e->d1 = d1; e->n1 = n1; e->splits1 = splits1;
    const bool two = p2 && d2;
    e->p2 = two ? p2 : nullptr; e->d2 = two ? d2 : nullptr; e->n2 = two ? n2 : 0; e->splits2 = two ? splits2 : 0;
    e->accumulate = accumulate; e->blocks1 = reduce_grid(n1);
}

// ---- chained form: the reduction a weight gradient owes, run in the PROLOGUE of the next weight-gradient kernel of its stream
// (nimg_conv2d_wgrad_bf16_chained): every thread of every workgroup takes float4 columns gtid, gtid + nthreads, ... and performs
// reduce_slabs()'s additions for that column in reduce_slabs()'s order - segment sums r_seg = a0 + a1 with a0 += slabs seg, seg + 32,
// ..., a1 += slabs seg + 16, seg + 48, ..., then r_0 + r_1 + ... + r_15, then the accumulate term - so the result is bit-identical
// to the separate launch, without its launch, its wait for free CUs beside the chip-filling kernels, or a second pass of barriers.
// The slabs were written by the previous kernel of the same stream: visible at the kernel boundary.
__device__ __forceinline__ void reduce_seq(const float* __restrict__ partial, float* __restrict__ dst, long count, int splits,
                                           int accumulate, long gtid, long nthreads) {
    if ((count & 3) == 0) {
        const long c4 = count >> 2;
        for (long i = gtid; i < c4; i += nthreads) {
            const float4* src = reinterpret_cast<const float4*>(partial) + i;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int seg = 0; seg < 16; ++seg) {
                float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
                for (int k = seg; k < splits; k += 32) {
                    const float4 v0 = src[(long)k * c4];
                    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
                    if (k + 16 < splits) {
                        const float4 v1 = src[(long)(k + 16) * c4];
                        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
                    }
                }
                const float4 sg = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
                if (seg == 0) r = sg;
                else { r.x += sg.x; r.y += sg.y; r.z += sg.z; r.w += sg.w; }
            }
            float4* d = reinterpret_cast<float4*>(dst) + i;
            if (accumulate) { const float4 o = *d; r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w; }
            *d = r;
        }
        return;
    }
    for (long i = gtid; i < count; i += nthreads) {
        float sacc = 0.f;
        for (int k = 0; k < splits; ++k) sacc += partial[(long)k * count + i];
        dst[i] = accumulate ? dst[i] + sacc : sacc;
    }
}
// one-dimensional grids and workgroups only (what the weight-gradient kernels launch)
__device__ __forceinline__ void reduce_entry_inline(const ReduceEntry& e) {
    if (e.n1 <= 0 || !e.p1 || !e.d1) return;
    const long gtid = (long)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (long)gridDim.x * blockDim.x;
    reduce_seq(e.p1, e.d1, e.n1, e.splits1, e.accumulate, gtid, nthreads);
    if (e.p2 && e.d2 && e.n2 > 0) reduce_seq(e.p2, e.d2, e.n2, e.splits2, e.accumulate, gtid, nthreads);
}
static inline ReduceEntry empty_reduce_entry() {
    ReduceEntry e;
    fill_reduce_entry(&e, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, 0, 0);
    e.blocks1 = 0;
    return e;
}

// ---- in-kernel finish of a split-K sum by the LAST-ARRIVING workgroup.  The separate
// reduce_slabs2 launch behind every weight gradient (~32 per training step) read slabs that were still in the Infinity Cache,
// but each cost a launch on the side stream and a pass over every slab by a handful of workgroups.  Here the workgroups
// that contribute to one dw TILE share arrival counters: a workgroup writes its slab tile, publishes it (agent-scope release),
// draws a ticket; the one that draws the last ticket of its GROUP of `group` consecutive splits sums the group's slabs in split
// order (into the final tensor when there is one group, else in place into the group's first slab), and the last group to finish
// sums the group sums in group order.  Every sum has ONE fixed order => deterministic, whoever arrives last.  The hand-off is
// the recipe of cdna_hip_programming.md section 6 G16 (counter form): every storing wave drains vmcnt, workgroup barrier, ONE
// lane: release fence + asm wait + relaxed agent fetch_add; the reducer: ONE lane acquire fence, barrier, plain loads - correct
// for any placement of a tile's splits on XCDs / CUs.  Counters: caller-provided, zero before the first launch, and the last
// arriver stores 0 again (nimg_bind_tickets); a tile owns 1 + NG words: [0] = arrivals of groups, [1 + g] = arrivals in group g.
struct TicketJob {
    unsigned* cnt;                 // this TILE's counters (null: no in-kernel finish)
    float* slab[2];                // 0: dw slabs, 1: db partials (may be null); slab k at slab[i] + k * stride[i]
    long stride[2];
    float* dst[2];                 // dw, db (item offsets inside a slab = offsets inside the destination)
    int splits, group, accumulate;
};
struct TicketItem { int which; long off; };          // `which` 0 / 1 as above, `off` in floats, a multiple of 4 (one float4 per item)

// every thread of the workgroup calls this after its last slab store; flag_word: 4 bytes of the kernel's LDS array nobody reads
// any more.  True in every thread of the workgroup that drew ticket `expected - 1`.
__device__ __forceinline__ bool ticket_arrive(unsigned* cnt, unsigned expected, unsigned* flag_word) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = t + 1 == expected;
        if (last) {
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // zero again for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        *reinterpret_cast<volatile unsigned*>(flag_word) = last ? 1u : 0u;
    }
    __syncthreads();
    return *reinterpret_cast<volatile unsigned*>(flag_word) != 0u;
}

// items [0, items) of this tile: sum of slabs k0, k0 + kstep, ... (n of them), ascending, 8 loads in flight per thread
template <int NT, typename Map>
__device__ __forceinline__ void ticket_sum(const TicketJob& j, int k0, int kstep, int n, bool final, int items, Map map) {
    for (int it = threadIdx.x; it < items; it += NT) {
        const TicketItem m = map(it);
        // (selects, not j.slab[m.which]: a dynamically indexed member array puts the whole struct on the stack - 80 bytes of scratch
        //  per lane and the scratch set-up of every launch, used or not)
        const long stride = m.which ? j.stride[1] : j.stride[0];
        const long st = stride * kstep;
        float* src = (m.which ? j.slab[1] : j.slab[0]) + (long)k0 * stride + m.off;
        float4 a = *reinterpret_cast<const float4*>(src);
        int k = 1;
        for (; k + 8 <= n; k += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (long)(k + u) * st);
#pragma unroll
            for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
        for (; k < n; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(src + (long)k * st);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        if (final) {
            float4* d = reinterpret_cast<float4*>((m.which ? j.dst[1] : j.dst[0]) + m.off);
            if (j.accumulate) { const float4 o = *d; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
            *d = a;
        } else {
            *reinterpret_cast<float4*>(src) = a;
        }
    }
}

// the whole finish: call with every thread of the workgroup (NT threads) after the workgroup's slab stores
template <int NT, typename Map>
__device__ __forceinline__ void ticket_finish(const TicketJob& j, int split, int items, Map map, unsigned* flag_word) {
    const int G = j.group, NG = (j.splits + G - 1) / G, g = split / G;
    const int members = j.splits - g * G < G ? j.splits - g * G : G;
    if (!ticket_arrive(j.cnt + 1 + g, (unsigned)members, flag_word)) return;
    ticket_sum<NT>(j, g * G, 1, members, NG == 1, items, map);
    if (NG == 1) return;
    if (!ticket_arrive(j.cnt, (unsigned)NG, flag_word)) return;
    ticket_sum<NT>(j, 0, G, NG, true, items, map);
}

// group size for `splits` slabs: one level up to 24, else ~sqrt (two levels of <= 16 .. 23 reads per item)
static inline int ticket_group(int splits) {
    if (splits <= 24) return splits < 1 ? 1 : splits;
    int g = 1;
    while (g * g < splits) ++g;
    return g;
}
__device__ __forceinline__ int ticket_words_per_tile_dev(int splits, int group) { return 1 + (splits + group - 1) / group; }
static inline int ticket_words_per_tile(int splits) { const int g = ticket_group(splits); return 1 + (splits + g - 1) / g; }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace nimg

// zeroed counter words bound to `stream` by nimg_bind_tickets (pointwise.hip), or null when fewer than `words` are bound
extern "C" unsigned* nimg_internal_tickets(hipStream_t stream, size_t words);
