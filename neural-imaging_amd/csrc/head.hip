// The FAN's head in throughput mode (models/forensics.py:76-94: Conv2D(nf, 1x1, leaky_relu) -> GlobalAveragePooling2D -> Dense
// softmax) as two skinny bf16 GEMMs with everything around them fused in:
//
//   head_fwd:    act = LeakyReLU(X W + b), gap[n] = mean over the image's pixels of act - the activation tensor itself NEVER
//                reaches HBM: the backward pass only needs the SIGN of each value (LeakyReLU'), which leaves as one bit per
//                value (mask[px][co / 32], 1/32 of the float32 tensor the generic path wrote, re-read for the pooling and re-read
//                twice by the backward pass);
//   head_dgrad:  dX = (dAct W^T) * LeakyReLU'(X), where dAct[px][co] = g[n][co] * (bit ? 1 : alpha), g[n] = Wd dlogits[n] / HW, is
//                BUILT in registers as the matrix instruction's A operand from the mask bits and the 256 values of g - the
//                (N, HW, C) float32 gradient tensor the pooling's backward used to write (84 MB at 320 images) does not exist;
//   head_dact:   that tensor as bf16, for the weight-gradient kernel only (side stream, off the critical path).
//
// Was (C4, 320 images of 16 x 16 x 256): 1x1 forward 57 us + pooling 20 us, pooling backward 32 us + 1x1 input gradient 105 us on
// the launch stream.  Geometry: one workgroup per image, one wave per 32 pixels (HW = 32 NW), N dimension = all C channels in
// NF = C / 32 fragments of v_mfma_f32_32x32x16_bf16 (128 accumulator registers at C = 256), K in chunks of 64 through LDS
// (weights: [row][64 + 8] bf16, register-staged one chunk ahead).  C in {64, 128, 256}, HW in {64, 128, 256}; anything else keeps
// the generic kernels (ops.head_fused_ok).
#include "common.h"

namespace {

using namespace nimg;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct HeadParams {
    const void* x;            // fwd: (N * HW, C) bf16 input of the 1x1 layer
    const void* wimg;         // bf16 weight image (nimg_conv_weights_bf16 layout, taps = 1): fwd mode 0, dgrad mode 1
    const float* bias;        // fwd: (C)
    unsigned* mask;           // (N * HW, C / 32) words, bit j of word f = act[px][32 f + j] > 0.  fwd: out (or null); dgrad: in
    float* gap;               // fwd: (N, C) out
    const float* dlogits;     // dgrad: (N, K)
    const float* wdense;      // dgrad: (C, K)
    const void* in_mask;      // dgrad: (N * HW, C) bf16 - LeakyReLU' of the layer below is taken from its sign - or null
    void* dx;                 // dgrad: (N * HW, C) bf16 out
    int N, HW, C, K;
    float alpha;
};

constexpr int ROWB = 144;     // LDS bytes per staged row: 64 bf16 + 16 B pad
constexpr int ROWF = 272;     // dgrad epilogue scratch: 64 float32 + 16 B pad per pixel row

// forward epilogue, accumulator element J of a fragment: bias, LeakyReLU into the pooling sum; the sign bits of the wave's 64 values
// (one ballot = channels l32 of pixel rows r0 and r0 + 4) go to lanes r0 / r0 + 4 of `w` - every lane ends up with the word of ITS row
template <int J>
__device__ __forceinline__ void head_rows(const f32x16& a, float bias, float alpha, bool want_mask, float& s, unsigned& w) {
    if constexpr (J < 16) {
        const float v = a[J] + bias;
        const bool pos = v > 0.f;
        s += pos ? v : alpha * v;
        if (want_mask) {                                         // (wave-uniform)
            const unsigned long long bal = __ballot(pos);
            constexpr int r0 = 8 * (J >> 2) + (J & 3);
            // (s_nop: the ballot is a VALU write of an SGPR pair; v_writelane reading it as DATA right behind it saw the old value
            //  on gfx950 - the compiler inserts no wait states for operands of inline asm)
            asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %2\n\tv_writelane_b32 %0, %3, %4"
                         : "+v"(w) : "s"((unsigned)bal), "n"(r0), "s"((unsigned)(bal >> 32)), "n"(r0 + 4));
        }
        head_rows<J + 1>(a, bias, alpha, want_mask, s, w);
    }
}

template <int NW, int NF, int MODE>
__global__ __launch_bounds__(NW * 64, 2) void head_kernel(const HeadParams p) {
    constexpr int C = NF * 32, KC = C / 64, NT = NW * 64, HW = NW * 32;
    constexpr int BI = C * 8 / NT;                               // 16-byte items of a weight chunk per thread
    constexpr int SB = C * ROWB;                                 // (behind it - sA - fwd: the input chunk; dgrad: g and alpha g)
    constexpr int EPI = MODE == 1 ? NW * 32 * ROWF : 0;          // dgrad epilogue: [32][64 + 4] float32 per wave
    static_assert(C * 8 % NT == 0, "weight chunk divides over the threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* sB = smem_raw;
    unsigned char* sA = smem_raw + (SB > EPI ? SB : EPI);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x;
    const int half = lane >> 5, l32 = lane & 31;
    const long px0 = (long)n * HW;
    const unsigned char* wimg = reinterpret_cast<const unsigned char*>(p.wimg);

    f32x16 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[f][j] = 0.f;

    // ---- staging: weight chunk kc = image sub-chunks 4 kc .. 4 kc + 3, each [C rows][16] bf16
    u32x4 preB[BI], preA[4];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int item = tid + i * NT, r = item >> 3, s = (item >> 1) & 3, h = item & 1;
            preB[i] = *reinterpret_cast<const u32x4*>(wimg + ((long)((kc * 4 + s) * C + r)) * 32 + h * 16);
        }
        if constexpr (MODE == 0) {
            const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x) + px0 * C * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {                        // HW rows x 8 items = 4 per thread
                const int item = tid + i * NT, r = item >> 3, s = item & 7;
                preA[i] = *reinterpret_cast<const u32x4*>(xb + ((long)r * C + kc * 64 + s * 8) * 2);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int item = tid + i * NT, r = item >> 3, s = (item >> 1) & 3, h = item & 1;
            *reinterpret_cast<u32x4*>(sB + r * ROWB + s * 32 + h * 16) = preB[i];
        }
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int item = tid + i * NT, r = item >> 3, s = item & 7;
                *reinterpret_cast<u32x4*>(sA + r * ROWB + s * 16) = preA[i];
            }
        }
    };

    float biasr[NF];                                             // fwd: this lane's NF output channels (a load inside the epilogue is
    if constexpr (MODE == 0) {                                   // a memory round trip on the critical path)
#pragma unroll
        for (int f = 0; f < NF; ++f) biasr[f] = p.bias[f * 32 + l32];
    }
    unsigned mrow[NF];                                           // dgrad: the C mask bits of this lane's pixel
    if constexpr (MODE == 1) {
        // g[co] = (sum_j dlogits[n][j] wdense[co][j]) / HW, the gradient of the pooled feature spread over the image's pixels
        float* sG = reinterpret_cast<float*>(sA);
        for (int co = tid; co < C; co += NT) {
            float a = 0.f;
            for (int j = 0; j < p.K; ++j) a = fmaf(p.dlogits[(long)n * p.K + j], p.wdense[(long)co * p.K + j], a);
            a /= (float)HW;
            sG[co] = a;
            sG[C + co] = a * p.alpha;
        }
        const unsigned* mr = p.mask + (px0 + wave * 32 + l32) * NF;
#pragma unroll
        for (int f = 0; f < NF; ++f) mrow[f] = mr[f];
    }
    fetch(0);
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        if (kc > 0) __syncthreads();                             // every wave is done with the previous chunk
        commit();
        __syncthreads();
        if (kc + 1 < KC) fetch(kc + 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a;
            if constexpr (MODE == 0) {
                a = *reinterpret_cast<const bf16x8*>(sA + (wave * 32 + l32) * ROWB + ks * 32 + half * 16);
            } else {
                const float* sG = reinterpret_cast<const float*>(sA);
                const int co = kc * 64 + ks * 16;                // this lane: channels co + 8 half .. + 7
                const unsigned bits = (mrow[co >> 5] >> ((co & 31) + 8 * half)) & 0xffu;
                const float4 g0 = *reinterpret_cast<const float4*>(sG + co + 8 * half);
                const float4 g1 = *reinterpret_cast<const float4*>(sG + co + 8 * half + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(sG + C + co + 8 * half);
                const float4 h1 = *reinterpret_cast<const float4*>(sG + C + co + 8 * half + 4);
                const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = (__bf16)(((bits >> e) & 1u) ? gv[e] : hv[e]);
            }
            bf16x8 bfr[NF];                                      // every fragment requested before the first product waits
#pragma unroll
            for (int f = 0; f < NF; ++f)
                bfr[f] = *reinterpret_cast<const bf16x8*>(sB + (f * 32 + l32) * ROWB + ks * 32 + half * 16);
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfr[f], acc[f], 0, 0, 0);
        }
    }
    __syncthreads();                                             // the tiles are free: the epilogues reuse them

    if constexpr (MODE == 0) {
        // accumulator element j of fragment f: channel 32 f + l32, pixel row 8 (j / 4) + 4 half + j % 4 of this wave's 32
        float* sGap = reinterpret_cast<float*>(smem_raw);        // [NW][C]
        unsigned mword[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const float bias = biasr[f];
            float s = 0.f;
            mword[f] = 0u;
            head_rows<0>(acc[f], bias, p.alpha, p.mask != nullptr, s, mword[f]);
            s += __shfl_xor(s, 32, 64);                          // the two row halves of the fragment
            if (half == 0) sGap[wave * C + f * 32 + l32] = s;
        }
        if (p.mask && half == 0) {
            unsigned* mr = p.mask + (px0 + wave * 32 + l32) * NF;
#pragma unroll
            for (int f = 0; f < NF; ++f) mr[f] = mword[f];
        }
        __syncthreads();
        for (int co = tid; co < C; co += NT) {
            float s = sGap[co];
#pragma unroll
            for (int w = 1; w < NW; ++w) s += sGap[w * C + co];
            p.gap[(long)n * C + co] = s / (float)HW;
        }
    } else {
        // dX tile of this wave, 64 channels at a time through its own LDS scratch: written from the accumulator layout, read back
        // as 16-byte runs along the channels, times LeakyReLU' of the layer below, stored as 16-byte runs
        unsigned char* scr = smem_raw + wave * (32 * ROWF);
        const unsigned char* im = reinterpret_cast<const unsigned char*>(p.in_mask);
        unsigned char* dx = reinterpret_cast<unsigned char*>(p.dx);
#pragma unroll
        for (int fp = 0; fp < NF / 2; ++fp) {
#pragma unroll
            for (int ff = 0; ff < 2; ++ff)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int row = 8 * (j >> 2) + 4 * half + (j & 3);
                    *reinterpret_cast<float*>(scr + row * ROWF + (ff * 32 + l32) * 4) = acc[2 * fp + ff][j];
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = i * 64 + lane, row = idx >> 3, seg = idx & 7;
                const float4 v0 = *reinterpret_cast<const float4*>(scr + row * ROWF + seg * 32);
                const float4 v1 = *reinterpret_cast<const float4*>(scr + row * ROWF + seg * 32 + 16);
                float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                const long o = ((px0 + wave * 32 + row) * C + fp * 64 + seg * 8) * 2;
                if (im) {
                    const bf16x8 m = *reinterpret_cast<const bf16x8*>(im + o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] *= (float)m[e] > 0.f ? 1.0f : p.alpha;
                }
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)f[e];
                *reinterpret_cast<bf16x8*>(dx + o) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// dAct[px][co] = g[n][co] * (bit ? 1 : alpha) as bf16, for the weight-gradient kernel
__global__ __launch_bounds__(256) void head_dact_kernel(const unsigned* __restrict__ mask, const float* __restrict__ dlogits,
                                                        const float* __restrict__ wdense, void* __restrict__ dact, int hw, int c,
                                                        int k, float alpha, int parts) {
    __shared__ float sG[512];
    const int n = blockIdx.x / parts, part = blockIdx.x % parts, tid = threadIdx.x;
    for (int co = tid; co < c; co += 256) {
        float a = 0.f;
        for (int j = 0; j < k; ++j) a = fmaf(dlogits[(long)n * k + j], wdense[(long)co * k + j], a);
        a /= (float)hw;
        sG[co] = a;
        sG[256 + co] = a * alpha;
    }
    __syncthreads();
    const int c8 = c >> 3, nf = c >> 5;
    for (int i = part * 256 + tid; i < hw * c8; i += parts * 256) {           // one 16-byte run of 8 channels per item
        const int px = i / c8, co = (i % c8) * 8;
        const unsigned bits = (mask[((long)n * hw + px) * nf + (co >> 5)] >> (co & 31)) & 0xffu;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (__bf16)(((bits >> e) & 1u) ? sG[co + e] : sG[256 + co + e]);
        *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(dact) + ((long)n * hw + px) * c + co) = v;
    }
}

template <int NW, int NF, int MODE>
int launch(const HeadParams& p, hipStream_t s) {
    constexpr int C = NF * 32, HW = NW * 32;
    constexpr int SB = C * ROWB, EPI = MODE == 1 ? NW * 32 * ROWF : 0, SA = MODE == 0 ? HW * ROWB : 2 * C * 4;
    constexpr size_t lds = (size_t)(SB > EPI ? SB : EPI) + SA;
    static_assert(lds <= 160 * 1024 && (size_t)NW * C * 4 <= lds, "LDS");
    auto k = head_kernel<NW, NF, MODE>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)p.N), dim3(NW * 64), lds, s, p);
    return hipGetLastError() == hipSuccess ? NIMG_OK : NIMG_ERR_LAUNCH;
}

template <int MODE>
int dispatch(const HeadParams& p, hipStream_t s) {
#define NIMG_HEAD(NW_, NF_) if (p.HW == NW_ * 32 && p.C == NF_ * 32) return launch<NW_, NF_, MODE>(p, s)
    NIMG_HEAD(8, 8); NIMG_HEAD(4, 8); NIMG_HEAD(2, 8);
    NIMG_HEAD(8, 4); NIMG_HEAD(4, 4); NIMG_HEAD(2, 4);
    NIMG_HEAD(8, 2); NIMG_HEAD(4, 2); NIMG_HEAD(2, 2);
#undef NIMG_HEAD
    return NIMG_ERR_ARG;
}

}  // namespace

extern "C" {

int nimg_head_fused_ok(int hw, int c) {
    return (hw == 64 || hw == 128 || hw == 256) && (c == 64 || c == 128 || c == 256) ? 1 : 0;
}

int nimg_head_fwd(const void* x, const void* wimg, const float* bias, unsigned* mask, float* gap, int n, int hw, int c,
                  float alpha, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!x || !wimg || !bias || !gap || n < 0 || !nimg_head_fused_ok(hw, c)) return NIMG_ERR_ARG;
    HeadParams p = {};
    p.x = x; p.wimg = wimg; p.bias = bias; p.mask = mask; p.gap = gap; p.N = n; p.HW = hw; p.C = c; p.alpha = alpha;
    return dispatch<0>(p, (hipStream_t)stream);
}

int nimg_head_dgrad(const unsigned* mask, const float* dlogits, const float* wdense, int k, const void* wimg_t,
                    const void* in_mask, void* dx, int n, int hw, int c, float alpha, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!mask || !dlogits || !wdense || !wimg_t || !dx || n < 0 || k <= 0 || !nimg_head_fused_ok(hw, c)) return NIMG_ERR_ARG;
    HeadParams p = {};
    p.wimg = wimg_t; p.mask = const_cast<unsigned*>(mask); p.dlogits = dlogits; p.wdense = wdense; p.in_mask = in_mask; p.dx = dx;
    p.N = n; p.HW = hw; p.C = c; p.K = k; p.alpha = alpha;
    return dispatch<1>(p, (hipStream_t)stream);
}

int nimg_head_dact(const unsigned* mask, const float* dlogits, const float* wdense, int k, void* dact, int n, int hw, int c,
                   float alpha, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!mask || !dlogits || !wdense || !dact || n < 0 || k <= 0 || hw <= 0 || c <= 0 || c > 256 || (c & 31)) return NIMG_ERR_ARG;
    const int parts = n >= 1024 ? 1 : (n >= 256 ? 4 : 16);
    hipLaunchKernelGGL(head_dact_kernel, dim3((unsigned)(n * parts)), dim3(256), 0, (hipStream_t)stream, mask, dlogits, wdense,
                       dact, hw, c, k, alpha, parts);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

}  // extern "C"
