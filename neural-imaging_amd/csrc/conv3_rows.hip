// Row-band STREAMING form of the 3x3 / stride 1 / SAME convolution over bf16-stored activations with 32 output channels - the UNet's
// level-1 layers (models/pipelines.py:190-216: ec12, dc41, dc42 forward; dc42 / dc5 input gradients), throughput mode.
//
// Why another form.  At 128 x 128 x 32 channels these layers are byte-bound (9 x 32 x 32 MACs per 128 B moved) and the tile
// kernels (conv_fwd_bf16_kernel / conv3_dma_kernel: one 16 x 16 tile per workgroup, prologue - two 16-channel chunks - epilogue)
// reach ~3 TB/s: a workgroup has ONE tile's bytes in flight, its neighbours on the CU are in other phases, and every tile pays a
// halo (18 x 18 for 16 x 16: 1.27 x the reads), an address prologue and a drained pipeline.  Here a workgroup is PERSISTENT over a
// band of image rows of full width:
//   * the input arrives one image row at a time by LDS-DMA (buffer_load_dwordx4 ... lds, no staging registers, no write pass)
//     into a ring of 2 RB + 2 row slots; the RB rows of step s + 1 are requested at the top of step s and land under its
//     matrix work and its stores - a steady stream instead of load / compute / store phases;
//   * no horizontal halo (the slot carries one zero pixel left and right), one row of vertical halo per band side;
//   * the whole weight set (9 x Cin x 32 bf16 = 18 / 36 KB) is staged ONCE per workgroup;
//   * a wave owns one 32-pixel column block of the RB rows of a step: accumulators stay in registers across the 9 taps and all
//     input channels, summed in the tile kernels' order (16-channel chunk, then ky, then kx) - the results are BIT-IDENTICAL;
//   * output rows leave as whole 2 KB runs (32 pixels x 32 channels x bf16), 16 bytes per lane, optionally with the 2x2 max-pool of
//     the row pair next to them (NIMG_POOL_ALSO semantics: pooled on the raw sums, bias + LeakyReLU on the winner).
// LDS image of a row slot: per 32-channel plane (W + 2) pixels x 64 B; the four 16-byte pieces of a pixel are XOR-swizzled by
// (pixel >> 2) & 3 - applied on the SOURCE address of the DMA (which writes lane-linear) - so a ds_read_b128 of 16 consecutive
// pixels covers all 64 banks.  Inputs with 64 channels (a decoder layer's [up, skip] pair, or one 64-channel tensor) are two planes.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"

namespace {

using namespace nimg;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int r_u32x4 __attribute__((ext_vector_type(4)));

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) - the operand ring below must be indexed
// with constants whatever the unroller decides (a 36-tap pipeline left rolled puts the ring into scratch memory)
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

struct RowsParams {
    const void* in1;              // (N, H, W, C1) bf16
    const void* in2;              // (N, H, W, C2) bf16 or null
    const __bf16* wb;             // [Cin / 16][9][32][16] (nimg_conv_weights_bf16, mode 0 forward / mode 1 input gradient)
    const float* bias;            // 32 or null
    const void* mask;             // optional (N, H, W, 32) bf16: out *= mask > 0 ? 1 : alpha (the previous layer's LeakyReLU')
    void* out;                    // (N, H, W, 32) bf16 (float32 with out_f32)
    void* out2;                   // second 32-channel output (64 output channels split over two tensors) or null
    int out_f32;                  // out holds float32
    int ostride;                  // channels per pixel of out (and mask): 32, or 64 when both output blocks go to ONE tensor
    void* pool_out;               // optional (N, H / 2, W / 2, 32) bf16
    float* d2s;                   // optional (N, 2 H, 2 W, 3) float32: the layer has 12 output channels and leaves as clip(depth_to_space(.., 2), 0, 1)
    int cout;                     // 32, or 12 with d2s
    int C1, C2, N, H;
    int units, bands, BH;         // work units = N x bands, a band = BH rows (BH % RB == 0)
    int act;                      // 1: LeakyReLU(alpha) on the output
    int ablate;                   // diagnostics (NIMG_ROWS_ABLATE, results are wrong): 1 no stores, 2 no matrix loop, 4 no row requests after a unit's first
    float alpha;
};

__device__ __forceinline__ void glds16(r_u32x4 rsrc, unsigned lds_addr, unsigned voff, int soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

template <int NP, int W, int RB, int PFD, int NCW, int NO>
struct RowsGeom {
    static constexpr int NR = (PFD + 1) * RB + 2;               // ring slots: the step's window + PFD steps of rows in flight
    static constexpr int PLANE = (W + 2) * 64;                  // bytes of one 32-channel plane of a row slot
    static constexpr int SLOT = NP * PLANE;
    static constexpr int WBYTES = NP * 2 * 9 * (32 * NO) * 32;  // weights: [k-step][tap][co][2 x 16 B]
    static constexpr int EPI = 32 * (32 + EPI_PAD) * 4;         // per-wave epilogue scratch
    static constexpr size_t LDS = (size_t)NR * SLOT + WBYTES + NCW * EPI;
};

// NCW = computing waves: 4 (one per SIMD) or 8 (two per SIMD: one wave's epilogue - LDS turn-around, conversions, stores - runs
// under its partner's matrix instructions); wave NCW is the loader
// NO = 32-channel output blocks: 1, or 2 for an input gradient that leaves as two 32-channel tensors (a decoder layer's [up, skip])
template <int NP, int W, int RB, int PFD, int NCW, int NO>
__global__ __launch_bounds__(64 * (NCW + 1)) __attribute__((amdgpu_waves_per_eu(1, NCW == 8 ? 3 : 2))) void conv3_rows_kernel(const RowsParams p) {
    using G = RowsGeom<NP, W, RB, PFD, NCW, NO>;
    constexpr int NR = G::NR, PLANE = G::PLANE, SLOT = G::SLOT;
    constexpr int KSTEPS = 2 * NP;                              // 16-channel MFMA k-steps
    constexpr int MF = W / 32, RG = NCW / MF, RW = RB / RG;     // column blocks, row groups of the computing waves, rows per wave
    constexpr int PIECES = W / 16;                              // 1 KB DMA pieces per plane row
    constexpr int NTHR = 64 * (NCW + 1);                        // the computing waves + the loader wave
    static_assert(W % 64 == 0 && MF <= 4 && NCW % MF == 0 && RB % RG == 0 && RW >= 1, "geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                                   // ring first: DMA addresses stay small
    uint4* sW = reinterpret_cast<uint4*>(smem + NR * SLOT);
    const unsigned sA_addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* elds = reinterpret_cast<float*>(smem + NR * SLOT + G::WBYTES + (wave % NCW) * G::EPI);
    const int half = lane >> 5, co = lane & 31;
    const int pf = wave % MF, rg = wave / MF;

    // ---- DMA source offsets of this wave's pieces inside a plane row: LDS position q (16 B units from pixel 1) holds pixel
    //      x = q / 4, piece s' = q % 4 of the swizzled image = piece s = s' ^ (((x + 1) >> 2) & 3) of the source pixel
    unsigned voff[NP][PIECES];                                  // (only the loader wave uses them)
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        const bool second = p.in2 && pl * 32 >= p.C1;
        const int ct = second ? p.C2 : p.C1, c0 = second ? pl * 32 - p.C1 : pl * 32;
#pragma unroll
        for (int k = 0; k < PIECES; ++k) {
            const int q = k * 64 + lane, x = q >> 2, sp = (q & 3) ^ (((x + 1) >> 2) & 3);
            voff[pl][k] = (unsigned)(x * ct * 2 + c0 * 2 + sp * 16);
        }
    }
    const long row_bytes1 = (long)W * p.C1 * 2, row_bytes2 = (long)W * p.C2 * 2;
    const unsigned long a1 = (unsigned long)p.in1, a2 = (unsigned long)(p.in2 ? p.in2 : p.in1);
    const r_u32x4 rs1 = {(unsigned)a1, (unsigned)(a1 >> 32) & 0xffffu, (unsigned)((long)p.N * p.H * row_bytes1), 0x00020000u};
    const r_u32x4 rs2 = {(unsigned)a2, (unsigned)(a2 >> 32) & 0xffffu,
                         (unsigned)((long)p.N * p.H * (p.in2 ? row_bytes2 : row_bytes1)), 0x00020000u};

    // loader wave: request image row y of image n into ring position pos.  A row outside the image is requested too, at
    // out-of-range offsets: the buffer descriptor answers with zeros, which land in the slot like any other row - the number
    // of transfers per batch is then a constant and the loader can wait with a COUNTED s_waitcnt
    auto request = [&](int n, int y, int pos) {
        const bool inside = (unsigned)y < (unsigned)p.H;
        static_for<NP>([&](auto PL) {
            constexpr int pl = decltype(PL)::value;
            const bool second = p.in2 && pl * 32 >= p.C1;
            const long rb = second ? row_bytes2 : row_bytes1;
            // (wave-uniform by construction; readfirstlane makes it provable for the "s" operands of the asm statement)
            const int soff = __builtin_amdgcn_readfirstlane(inside ? (int)(((long)n * p.H + y) * rb) : 0);
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(sA_addr + (unsigned)(pos * SLOT + pl * PLANE + 64)));
            r_u32x4 rs = second ? rs2 : rs1;
            rs[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)rs[0]);
            rs[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)rs[1]);
            rs[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)rs[2]);
            rs[3] = (unsigned)__builtin_amdgcn_readfirstlane((int)rs[3]);
            static_for<PIECES>([&](auto K) {
                constexpr int k = decltype(K)::value;
                glds16(rs, dst + k * 1024, inside ? voff[pl][k] : 0x80000000u, soff);
            });
        });
    };

    // the first unit's rows are requested BEFORE the weights are staged: their flight covers the staging
    constexpr int DPB = RB * NP * PIECES;                        // transfers of one step's batch of new rows
    const int steps = p.BH / RB;
    auto request_unit = [&](int unit) {
        const int n = unit / p.bands, y0 = (unit % p.bands) * p.BH;
        for (int i = 0; i < RB + 2; ++i) request(n, y0 - 1 + i, i);
        if (PFD == 2 && steps > 1)
            for (int i = 0; i < RB; ++i) request(n, y0 + RB + 1 + i, (RB + 2 + i) % NR);
    };
    if (wave == NCW && (int)blockIdx.x < p.units) request_unit(blockIdx.x);
    // ---- once per workgroup: weights -> LDS ([k-step][tap][co] rows of two 16-byte halves, XOR-swizzled by (co >> 3) & 1),
    //      the zero pixels left and right of every slot
    for (int item = tid; item < KSTEPS * 9 * (32 * NO) * 2; item += NTHR) {
        const int h8 = item & 1, row = item >> 1;               // row = (ks * 9 + tap) * 32 NO + co
        const int cow = row % (32 * NO), kt = row / (32 * NO);   // the image in HBM has p.cout rows per (k-step, tap)
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (cow < p.cout) v = *reinterpret_cast<const uint4*>(p.wb + ((long)kt * p.cout + cow) * 16 + h8 * 8);
        sW[row * 2 + (h8 ^ ((row >> 3) & 1))] = v;
    }
    for (int item = tid; item < NR * NP * 2 * 4; item += NTHR) {
        const int q = item & 3, side = (item >> 2) & 1, pl = item >> 3;      // pl = slot * NP + plane
        *reinterpret_cast<uint4*>(sA + pl * PLANE + (side ? (W + 1) * 64 : 0) + q * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    const __bf16* maskp = reinterpret_cast<const __bf16*>(p.mask);
    __bf16* outp = reinterpret_cast<__bf16*>(p.out);
    __bf16* poolp = reinterpret_cast<__bf16*>(p.pool_out);
    float bias_l = 0.f, bias8[NO][8];
#pragma unroll
    for (int ni = 0; ni < NO; ++ni)
#pragma unroll
        for (int e = 0; e < 8; ++e) bias8[ni][e] = 0.f;
    float bias_d[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};         // d2s: lane = (pixel, output row parity dy) takes channels 6 dy .. + 5
    if (p.bias && p.d2s) {
#pragma unroll
        for (int e = 0; e < 6; ++e) bias_d[e] = p.bias[half * 6 + e];
    } else if (p.bias) {
        bias_l = p.bias[co];
        // the epilogue hands lane l the channels 8 (l & 3) .. + 7 of a pixel on every call: their biases live in registers (a global
        // load inside the epilogue is a full memory round trip per row for a wave that has its SIMD to itself)
#pragma unroll
        for (int ni = 0; ni < NO; ++ni)
#pragma unroll
            for (int e = 0; e < 8; ++e) bias8[ni][e] = p.bias[ni * 32 + (lane & 3) * 8 + e];
    }
    const int b_lane = co * 2 + (half ^ ((co >> 3) & 1));       // + ((ks * 9 + tap) * NO + ni) * 64
    const bool loader = wave == NCW;
    for (int unit = blockIdx.x; unit < p.units; unit += gridDim.x) {
        const int n = unit / p.bands, y0 = (unit % p.bands) * p.BH;
        __syncthreads();                                         // the previous unit's reads are done
        if (loader) {
            if (unit != (int)blockIdx.x) request_unit(unit);
            if (PFD == 2 && steps > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPB) : "memory");          // transfers complete in order
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                                         // rows of step 0 landed
        for (int s = 0; s < steps; ++s) {
            if (loader) {
                // the rows of steps s + 1 .. s + PFD travel while the four computing waves work on step s; only THIS wave waits for
                // them - the computing waves never wait on the vector-memory counter (their stores would be in it)
                if (s + PFD < steps && !(p.ablate & 4))
                    for (int i = 0; i < RB; ++i) request(n, y0 + (s + PFD) * RB + 1 + i, ((s + PFD) * RB + 2 + i) % NR);
                if (PFD == 2 && s + 2 < steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPB) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                continue;
            }
            const int base = (s * RB + rg * RW) % NR;            // ring position of input row (first output row of the wave) - 1
            f32x16 acc[RW][NO];
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int ni = 0; ni < NO; ++ni)
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[r][ni][j] = 0.f;
            int slot_off[RW + 2];
#pragma unroll
            for (int i = 0; i < RW + 2; ++i) slot_off[i] = ((base + i) % NR) * SLOT;
            // One software pipeline over the step's taps (k-step, ky, kx - the tile kernels' summation order): with one computing wave
            // per SIMD nobody else covers an LDS round trip, so the operands of tap t + DIST are requested in front of the matrix
            // instructions of tap t (a ring of DIST + 1 operand sets, everything unrolled, the order pinned per tap).
            constexpr int NT = KSTEPS * 9, DIST = 2;
            uint4 ra[DIST + 1][RW], rbw[DIST + 1][NO];
            auto fetch_tap = [&](auto T, auto S) {
                constexpr int t = decltype(T)::value, slot = decltype(S)::value;
                constexpr int ks = t / 9, ky = (t % 9) / 3, kx = t % 3;
                constexpr int pl = ks >> 1;
                const int sub = (ks & 1) * 2 + half;
#pragma unroll
                for (int ni = 0; ni < NO; ++ni) rbw[slot][ni] = sW[(t * NO + ni) * 64 + b_lane];
                const int px = pf * 32 + co + kx;                // slot pixel index of output pixel (pf * 32 + lane) under tap kx
                const int a_off = pl * PLANE + px * 64 + ((sub ^ ((px >> 2) & 3)) << 4);
#pragma unroll
                for (int r = 0; r < RW; ++r) ra[slot][r] = *reinterpret_cast<const uint4*>(sA + slot_off[r + ky] + a_off);
            };
            if (!(p.ablate & 2)) {
                static_for<DIST>([&](auto D) { fetch_tap(D, D); });
                __builtin_amdgcn_sched_group_barrier(0x100, DIST * (RW + NO), 0);
                static_for<NT>([&](auto T) {
                    constexpr int t = decltype(T)::value;
                    if constexpr (t + DIST < NT)
                        fetch_tap(std::integral_constant<int, t + DIST>{}, std::integral_constant<int, (t + DIST) % (DIST + 1)>{});
#pragma unroll
                    for (int r = 0; r < RW; ++r)
#pragma unroll
                        for (int ni = 0; ni < NO; ++ni)
                            acc[r][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                *reinterpret_cast<const bf16x8*>(&ra[t % (DIST + 1)][r]),
                                *reinterpret_cast<const bf16x8*>(&rbw[t % (DIST + 1)][ni]), acc[r][ni], 0, 0, 0);
                    if constexpr (t + DIST < NT) __builtin_amdgcn_sched_group_barrier(0x100, RW + NO, 0);    // DS reads of tap t + DIST
                    __builtin_amdgcn_sched_group_barrier(0x008, RW * NO, 0);                                 // the matrix instructions of tap t
                });
            }
            // ---- epilogue: bias, activation, mask, 16-byte stores; then the pooled row pairs
            const int yw = y0 + s * RB + rg * RW;
            if (NO == 1 && p.d2s) {
                // last layer of the UNet (pipelines.py:216-223): 12 channels = the 2 x 2 x 3 values of depth_to_space(2); lane (pixel,
                // dy) writes the six floats of output row 2 y + dy, pixels 2 x and 2 x + 1, clipped to [0, 1] (the straight-through
                // clip's forward value) - 768 contiguous bytes per wave and row; the 12-channel tensor never exists in HBM
                constexpr int RS = 32 + EPI_PAD;
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int j = 0; j < 16; ++j) elds[((j & 3) + 8 * (j >> 2) + 4 * half) * RS + co] = acc[r][0][j];
                    __builtin_amdgcn_wave_barrier();
                    float2* dst = reinterpret_cast<float2*>(
                        p.d2s + ((((long)n * 2 * p.H + 2 * (yw + r) + half) * (2L * W)) + 2 * (pf * 32 + co)) * 3);
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        const float2 v = *reinterpret_cast<const float2*>(elds + co * RS + half * 6 + 2 * e);
                        const float a = fminf(fmaxf(v.x + bias_d[2 * e], 0.f), 1.f), b = fminf(fmaxf(v.y + bias_d[2 * e + 1], 0.f), 1.f);
                        if (!(p.ablate & 1)) dst[e] = make_float2(a, b);
                    }
                }
                __syncthreads();
                continue;
            }
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const long rowbase = (((long)n * p.H + yw + r) * W + pf * 32) * p.ostride;
#pragma unroll
                for (int ni = 0; ni < NO; ++ni) {
                    void* dstp = (ni == 0 || p.ostride == 64) ? p.out : p.out2;
                    const int cofs = p.ostride == 64 ? ni * 32 : 0;
                    epilogue_via_lds8<1>(reinterpret_cast<const f32x16(&)[1]>(acc[r][ni]), elds, lane, [&](int row, int c, float4 lo, float4 hi) {
                        float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += bias8[ni][e];      // c == 8 (lane & 3) on every call
                        if (p.act == 1) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] = lrelu(f[e], p.alpha);
                        }
                        const long o = rowbase + (long)row * p.ostride + cofs + c;
                        if (maskp) {
                            const bf16x8 m = *reinterpret_cast<const bf16x8*>(maskp + o);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] *= (float)m[e] > 0.f ? 1.0f : p.alpha;
                        }
                        if (p.ablate & 1) return;
                        if (p.out_f32) {
                            float4* d4 = reinterpret_cast<float4*>(reinterpret_cast<float*>(dstp) + o);
                            d4[0] = make_float4(f[0], f[1], f[2], f[3]);
                            d4[1] = make_float4(f[4], f[5], f[6], f[7]);
                        } else {
                            bf16x8 ov;
#pragma unroll
                            for (int e = 0; e < 8; ++e) ov[e] = (__bf16)f[e];
                            *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(dstp) + o) = ov;
                        }
                    });
                }
            }
            if (NO == 1 && poolp) {
                // rows (r, r + 1), pixels (2 i, 2 i + 1): registers j, j + 1 of one lane (j even) - the window of pooled pixel
                // {0, 1, 4, 5, 8, 9, 12, 13}[j / 2] + 2 half; pooled on the raw sums, bias + LeakyReLU on the winner
                constexpr int RS = 32 + EPI_PAD;
#pragma unroll
                for (int r = 0; r + 1 < RW; r += 2) {
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int j = 2 * q;
                        const float m = fmaxf(fmaxf(acc[r][0][j], acc[r][0][j + 1]), fmaxf(acc[r + 1][0][j], acc[r + 1][0][j + 1]));
                        const int pp = ((j & 3) >> 1) + 4 * (j >> 2) + 2 * half;       // pooled pixel inside the block (0..15)
                        elds[pp * RS + co] = p.act == 1 ? lrelu(m + bias_l, p.alpha) : m + bias_l;
                    }
                    __builtin_amdgcn_wave_barrier();
                    const int pp = lane >> 2, c = (lane & 3) * 8;
                    const float4 lo = *reinterpret_cast<const float4*>(elds + pp * RS + c);
                    const float4 hi = *reinterpret_cast<const float4*>(elds + pp * RS + c + 4);
                    bf16x8 ov;
                    ov[0] = (__bf16)lo.x; ov[1] = (__bf16)lo.y; ov[2] = (__bf16)lo.z; ov[3] = (__bf16)lo.w;
                    ov[4] = (__bf16)hi.x; ov[5] = (__bf16)hi.y; ov[6] = (__bf16)hi.z; ov[7] = (__bf16)hi.w;
                    const long o = ((((long)n * (p.H >> 1) + ((yw + r) >> 1)) * (W >> 1)) + pf * 16 + pp) * 32 + c;
                    *reinterpret_cast<bf16x8*>(poolp + o) = ov;
                }
            }
            __syncthreads();                                     // step s is read out; the loader's rows of step s + 1 have landed
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The UNet's FIRST layer in the same streaming form (pipelines.py:190: Conv2D(32, 3x3) on the 4-plane RAW stack, float32 input):
// K = (tap, ci) packed - 36 of 48 slots in three 16-wide k-steps, the order of conv_fwd_packed_bf16_kernel (k = 4 tap + ci, padded
// slots meet zero weights), so the results are bit-identical to it.  A row slot is (W + 2) pixels x 16 B of float32; a lane builds
// its operand from two 16-byte reads (the four channels of two taps) converted to bf16; the three weight fragments live in
// registers.  The layer is 3 matrix instructions per 32 pixels - the kernel is its output stream (2 KB runs, 16 bytes per lane).
struct RowsC4Params {
    const float* in;              // (N, H, W, 4) float32
    const float* w;               // (3, 3, 4, 32) float32 HWIO
    const float* bias;            // 32 or null
    void* out;                    // (N, H, W, 32) bf16
    int N, H, units, bands, BH, act;
    float alpha;
};

template <int W, int RB, int NCW>
__global__ __launch_bounds__(64 * (NCW + 1)) __attribute__((amdgpu_waves_per_eu(1, 3))) void conv3_rows_c4_kernel(const RowsC4Params p) {
    constexpr int NR = 2 * RB + 2, SLOT = (W + 2) * 16;
    constexpr int MF = W / 32, RG = NCW / MF, RW = RB / RG, PIECES = W / 64, NTHR = 64 * (NCW + 1);
    static_assert(W % 64 == 0 && NCW % MF == 0 && RB % RG == 0 && RW >= 1, "geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;
    const unsigned sA_addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* elds = reinterpret_cast<float*>(smem + NR * SLOT + (wave % NCW) * (32 * (32 + EPI_PAD) * 4));
    const int half = lane >> 5, co = lane & 31;
    const int pf = wave % MF, rg = wave / MF;
    const unsigned long a1 = (unsigned long)p.in;
    const long row_bytes = (long)W * 16;
    const r_u32x4 rs1 = {(unsigned)a1, (unsigned)(a1 >> 32) & 0xffffu, (unsigned)((long)p.N * p.H * row_bytes), 0x00020000u};
    auto request = [&](int n, int y, int pos) {
        const bool inside = (unsigned)y < (unsigned)p.H;
        const int soff = __builtin_amdgcn_readfirstlane(inside ? (int)(((long)n * p.H + y) * row_bytes) : 0);
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(sA_addr + (unsigned)(pos * SLOT + 16)));
        r_u32x4 rs = rs1;
        rs[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)rs[0]);
        rs[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)rs[1]);
        rs[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)rs[2]);
        rs[3] = (unsigned)__builtin_amdgcn_readfirstlane((int)rs[3]);
        static_for<PIECES>([&](auto K) {
            constexpr int k = decltype(K)::value;
            glds16(rs, dst + k * 1024, inside ? (unsigned)((k * 64 + lane) * 16) : 0x80000000u, soff);
        });
    };
    const int steps = p.BH / RB;
    auto request_unit = [&](int unit) {
        const int n = unit / p.bands, y0 = (unit % p.bands) * p.BH;
        for (int i = 0; i < RB + 2; ++i) request(n, y0 - 1 + i, i);
    };
    if (wave == NCW && (int)blockIdx.x < p.units) request_unit(blockIdx.x);
    for (int item = tid; item < NR * 2; item += NTHR)          // the zero pixel left and right of every slot
        *reinterpret_cast<uint4*>(sA + (item >> 1) * SLOT + ((item & 1) ? (W + 1) * 16 : 0)) = make_uint4(0u, 0u, 0u, 0u);
    // weight fragments: lane (co, half) holds k = 16 s + 8 half .. + 7 of output channel co
    bf16x8 wf[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 16 + half * 8 + j;
            wf[ks][j] = (__bf16)(k < 36 ? p.w[k * 32 + co] : 0.f);
        }
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = p.bias ? p.bias[(lane & 3) * 8 + e] : 0.f;
    __bf16* outp = reinterpret_cast<__bf16*>(p.out);
    const bool loader = wave == NCW;
    for (int unit = blockIdx.x; unit < p.units; unit += gridDim.x) {
        const int n = unit / p.bands, y0 = (unit % p.bands) * p.BH;
        __syncthreads();
        if (loader) {
            if (unit != (int)blockIdx.x) request_unit(unit);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        for (int s = 0; s < steps; ++s) {
            if (loader) {
                if (s + 1 < steps)
                    for (int i = 0; i < RB; ++i) request(n, y0 + (s + 1) * RB + 1 + i, ((s + 1) * RB + 2 + i) % NR);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                continue;
            }
            const int base = (s * RB + rg * RW) % NR;
            int slot_off[RW + 2];
#pragma unroll
            for (int i = 0; i < RW + 2; ++i) slot_off[i] = ((base + i) % NR) * SLOT;
            const int yw = y0 + s * RB + rg * RW;
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                f32x16 acc;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    float f[8];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        // the lane's tap of this k-step: 4 ks + u (lanes 0 - 31) or 4 ks + 2 + u (lanes 32 - 63); a padded slot
                        // (tap >= 9) meets zero weights, any finite data will do: tap 0.  Both candidates are compile-time taps -
                        // a select between two addresses, no register array indexed at run time
                        constexpr int NTAP = 9;
                        const int t0 = ks * 4 + u < NTAP ? ks * 4 + u : 0, t1 = ks * 4 + 2 + u < NTAP ? ks * 4 + 2 + u : 0;
                        const int o0 = slot_off[r + t0 / 3] + (t0 % 3) * 16, o1 = slot_off[r + t1 / 3] + (t1 % 3) * 16;
                        const float4 v = *reinterpret_cast<const float4*>(sA + (half ? o1 : o0) + (pf * 32 + co) * 16);
                        f[4 * u] = v.x; f[4 * u + 1] = v.y; f[4 * u + 2] = v.z; f[4 * u + 3] = v.w;
                    }
                    bf16x8 a;
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] = (__bf16)f[e];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wf[ks], acc, 0, 0, 0);
                }
                const long rowbase = (((long)n * p.H + yw + r) * W + pf * 32) * 32;
                epilogue_via_lds8<1>(reinterpret_cast<const f32x16(&)[1]>(acc), elds, lane, [&](int row, int c, float4 lo, float4 hi) {
                    float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += bias8[e];
                    if (p.act == 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = lrelu(f[e], p.alpha);
                    }
                    bf16x8 ov;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ov[e] = (__bf16)f[e];
                    *reinterpret_cast<bf16x8*>(outp + rowbase + row * 32 + c) = ov;
                });
            }
            __syncthreads();
        }
    }
}

template <int NP, int W, int RB, int PFD, int NCW, int NO = 1>
int launch_rows(const RowsParams& p, hipStream_t s) {
    using G = RowsGeom<NP, W, RB, PFD, NCW, NO>;
    static_assert(G::LDS <= 160 * 1024, "LDS");
    auto k = conv3_rows_kernel<NP, W, RB, PFD, NCW, NO>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
        attr = true;
    }
    static const int max_wg = getenv("NIMG_ROWS_WGS") ? atoi(getenv("NIMG_ROWS_WGS")) : 256;
    const int grid = p.units < max_wg ? p.units : max_wg;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * (NCW + 1)), G::LDS, s, p);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

}  // namespace

extern "C" {

/* see include/nimg.h */
int nimg_conv3_rows_bf16(const void* in1, int c1, const void* in2, int c2, const void* wb, const float* bias, const void* mask,
                         void* out, void* out2, void* pool_out, int n, int h, int wd, int cout, int act, float alpha, int flags,
                         void* stream) {
    if (n == 0) return NIMG_OK;
    if (!in1 || !wb || !out || n < 0 || (wd != 128 && wd != 64) || h < 4 || (h & 3) || (flags & ~NIMG_ROWS_F32_OUT)) return NIMG_ERR_ARG;
    const bool level2 = wd == 64;              // 64-pixel rows: 32 | 64 -> 64 channels in one tensor (the UNet's second level)
    if (level2) {
        if (cout != 64 || out2 || pool_out || flags || c2 != 0 || in2 || (c1 != 32 && c1 != 64)) return NIMG_ERR_ARG;
    } else {
        if (!((cout == 32 && !out2) || (cout == 64 && out2 && c1 == 32 && c2 == 0 && !pool_out && !mask && !flags))) return NIMG_ERR_ARG;
        if (!((c1 == 32 && c2 == 0 && !in2) || (c1 == 64 && c2 == 0 && !in2) || (c1 == 32 && c2 == 32 && in2))) return NIMG_ERR_ARG;
    }
    if (pool_out && (h & 1)) return NIMG_ERR_ARG;
    if ((long)n * h * wd * (c1 > c2 ? c1 : c2) * 2 >= (1l << 31) - 65536) return NIMG_ERR_ARG;
    RowsParams p;
    p.in1 = in1; p.in2 = in2; p.wb = (const __bf16*)wb; p.bias = bias; p.mask = mask; p.out = out; p.pool_out = pool_out;
    p.out2 = out2; p.out_f32 = (flags & NIMG_ROWS_F32_OUT) ? 1 : 0; p.ostride = level2 ? 64 : 32;
    p.C1 = c1; p.C2 = c2; p.N = n; p.H = h; p.act = act ? 1 : 0; p.alpha = alpha; p.d2s = nullptr; p.cout = cout;
    static const int ablate = getenv("NIMG_ROWS_ABLATE") ? atoi(getenv("NIMG_ROWS_ABLATE")) : 0;
    p.ablate = ablate;
    // bands: the largest power-of-two band height that still gives every CU a unit (256 units at 64 images x 128 rows: 32 rows)
    static const int bh_env = getenv("NIMG_ROWS_BH") ? atoi(getenv("NIMG_ROWS_BH")) : 0;
    int bh = bh_env > 0 ? bh_env : 32;
    while (bh > 4 && (h % bh != 0 || (long)n * (h / bh) < 256)) bh >>= 1;
    if (h % bh != 0) return NIMG_ERR_ARG;
    p.BH = bh; p.bands = h / bh; p.units = n * p.bands;
    hipStream_t s = (hipStream_t)stream;
    // rows per step: 4 where the ring (2 RB + 2 slots) fits, 2 for the 64-channel inputs (a slot is 16.6 KB there)
    static const int rb_env = getenv("NIMG_ROWS_RB") ? atoi(getenv("NIMG_ROWS_RB")) : 0;
    static const int pfd_env = getenv("NIMG_ROWS_PFD") ? atoi(getenv("NIMG_ROWS_PFD")) : 2;
    static const int ncw_env = getenv("NIMG_ROWS_NCW") ? atoi(getenv("NIMG_ROWS_NCW")) : 8;
    (void)rb_env;
    if (level2) return c1 == 32 ? launch_rows<1, 64, 4, 1, 8, 2>(p, s) : launch_rows<2, 64, 2, 1, 4, 2>(p, s);
    if (cout == 64) return launch_rows<1, 128, 4, 1, 8, 2>(p, s);
    if (c1 + c2 == 32) {
        if (ncw_env == 4) return pfd_env == 1 ? launch_rows<1, 128, 4, 1, 4>(p, s) : launch_rows<1, 128, 4, 2, 4>(p, s);
        return launch_rows<1, 128, 4, 1, 8>(p, s);
    }
    return launch_rows<2, 128, 2, 1, 4>(p, s);            // (a 64-channel slot is 16.6 KB: eight scratch areas do not fit beside the ring)
}

/* see include/nimg.h */
int nimg_conv3_rows_c4_bf16(const float* in, const float* w, const float* bias, void* out, int n, int h, int wd, int act, float alpha,
                            void* stream) {
    if (n == 0) return NIMG_OK;
    if (!in || !w || !out || n < 0 || wd != 128 || h < 4 || (h & 3)) return NIMG_ERR_ARG;
    if ((long)n * h * wd * 16 >= (1l << 31) - 65536) return NIMG_ERR_ARG;
    RowsC4Params p;
    p.in = in; p.w = w; p.bias = bias; p.out = out; p.N = n; p.H = h; p.act = act ? 1 : 0; p.alpha = alpha;
    int bh = 32;
    while (bh > 4 && (h % bh != 0 || (long)n * (h / bh) < 256)) bh >>= 1;
    if (h % bh != 0) return NIMG_ERR_ARG;
    p.BH = bh; p.bands = h / bh; p.units = n * p.bands;
    constexpr int RB = 4, NCW = 8;
    constexpr size_t lds = (size_t)(2 * RB + 2) * (128 + 2) * 16 + (size_t)NCW * 32 * (32 + nimg::EPI_PAD) * 4;
    auto k = conv3_rows_c4_kernel<128, RB, NCW>;
    const int grid = p.units < 512 ? p.units : 512;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * (NCW + 1)), lds, (hipStream_t)stream, p);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

/* see include/nimg.h */
int nimg_conv3_rows_d2s_bf16(const void* in, int c1, const void* wb, const float* bias, float* y, int n, int h, int wd, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!in || !wb || !y || n < 0 || c1 != 32 || wd != 128 || h < 4 || (h & 3)) return NIMG_ERR_ARG;
    if ((long)n * h * wd * c1 * 2 >= (1l << 31) - 65536) return NIMG_ERR_ARG;
    RowsParams p;
    p.in1 = in; p.in2 = nullptr; p.wb = (const __bf16*)wb; p.bias = bias; p.mask = nullptr; p.out = nullptr; p.pool_out = nullptr;
    p.out2 = nullptr; p.out_f32 = 0; p.ostride = 32; p.d2s = y; p.cout = 12;
    p.C1 = c1; p.C2 = 0; p.N = n; p.H = h; p.act = 0; p.alpha = 0.f; p.ablate = 0;
    int bh = 32;
    while (bh > 4 && (h % bh != 0 || (long)n * (h / bh) < 256)) bh >>= 1;
    if (h % bh != 0) return NIMG_ERR_ARG;
    p.BH = bh; p.bands = h / bh; p.units = n * p.bands;
    return launch_rows<1, 128, 4, 1, 8>(p, (hipStream_t)stream);
}

}  // extern "C"
