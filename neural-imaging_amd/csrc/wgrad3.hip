// 3x3 / stride 1 / SAME weight gradient of the UNet's convolutions (models/pipelines.py:191-216 under tape.gradient, :84-88) in
// throughput mode, bf16-stored operands: the "all taps in one wave" scheme of wgrad5.hip at kernel size 3.
//
// conv_wgrad_bf16_kernel<3, ...> (conv_bf16.hip) stages one 8 x 16 tile at a time through registers into a single LDS buffer
// behind two barriers; a tile is ~1 us of MFMA work for the two workgroups of a CU, shorter than the global latency of the
// next tile's loads, so every tile exposes part of a memory round trip (the 19 GFLOP layers take ~40 us at 0.5 PFLOP/s).
// Here:
//   * a wave owns all 9 taps of a 16 ci x 32 co block (18 accumulators of v_mfma_f32_16x16x32_bf16 = 72 AGPRs, two waves per
//     SIMD); K step = 8 columns x 4 rows; per kernel row the input operand is read once as 12 pixels per K group and the three
//     kx taps are its first 8 pixels, the window shifted by one pixel (v_alignbit_b32) and a second, 2-pixel-shifted read;
//   * tiles are double-buffered in LDS and the global loads of the tile AFTER the next one are already in flight (two register
//     sets): one barrier per tile, two tiles of latency cover;
//   * workgroup = 8 waves = 2 ci fragments x NB co pairs x KG row groups (NB * KG = 4): 32 ci x 32 NB co.  Cout = 32 layers
//     (UNet level 1) run NB = 1 / KG = 4 on 16-row tiles, Cout = 64 NB = 2 / KG = 2, Cout % 128 == 0 NB = 4 / KG = 1; the row
//     groups' partial sums are folded through LDS behind the loop; one slab per workgroup, fixed-order slab reduction;
//   * concatenated inputs (decoder: [up, skip]) are two tensors: the ci block picks its tensor;
//   * the bias gradient is summed from the dz fragments the ci-block-0 workgroups hold anyway.
// LDS layouts as in wgrad5.hip: pixel-major tiles, row strides padded by 32 B so that the two rows a 32-lane transpose read
// serves hit complementary bank halves; dz pixel stride 64 B (NB = 1), 192 B (NB = 2) or 320 B (NB = 4): four consecutive pixels
// on four bank quarters.
#include <stdlib.h>

#include "common.h"

namespace {

using namespace nimg;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Wg3Params {
    const void* in1;               // (N, H, W, C1) bf16
    const void* in2;               // (N, H, W, C2) bf16 or null
    const void* dz;                // (N, H, W, Cout) bf16
    float* partial;                // [slabs][9][C1 + C2][Cout]
    float* db_partial;             // [slabs][Cout] or null
    int C1, C2, Cout, N, H, W;
    int tiles_y, tiles_x, work_per_split;
    // in-kernel finish by the last-arriving workgroup of a dw tile (common.h ticket_finish); tickets == null: slabs only
    unsigned* tickets;
    float* dw;
    float* db;
    int splits, group, accumulate;
    ReduceEntry pre;               // the reduction the PREVIOUS weight gradient of this stream owes (common.h reduce_entry_inline), or empty
};

template <int NB, int TH>
struct Wg3Geom {
    static constexpr int KG = 4 / NB, CB = 32 * NB;
    static constexpr int THH = TH + 2, TWH = 18;
    static constexpr int ZS = NB == 1 ? 64 : (NB == 2 ? 192 : 320);
    static constexpr int IRS = TWH * 64 + 32, ZRS = 16 * ZS + 32;
    static constexpr int IBYTES = THH * IRS, ZBYTES = TH * ZRS, BUF = IBYTES + ZBYTES;
    static constexpr int IROWS = 512 / (TWH * 4);                // halo rows per staging pass (72 threads per row)
    static constexpr int IP = (THH + IROWS - 1) / IROWS;
    static constexpr int ZITEMS = TH * 16 * 4 * NB, ZP = ZITEMS / 512;
    static constexpr int ST = (TH / 4) * 2;                      // K steps per tile: 4 rows x 8 columns each
    static constexpr size_t FOLD = KG > 1 ? (size_t)(8 / KG) * (18 * 4 * 64 + 2 * 64) * 4 : 0;
    static constexpr size_t LDS = 2 * (size_t)BUF > FOLD ? 2 * (size_t)BUF : FOLD;
    static_assert(ZITEMS % 512 == 0 && TH % 4 == 0 && ST % KG == 0, "tile divides over threads / steps over row groups");
    static_assert(LDS <= 160 * 1024, "LDS");
};

template <int NB, int TH>
__global__ __launch_bounds__(512, 1) void conv3_wgrad_alltaps_kernel(const Wg3Params p) {
    using G = Wg3Geom<NB, TH>;
    constexpr int KG = G::KG, CB = G::CB, ZS = G::ZS, IRS = G::IRS, ZRS = G::ZRS, IP = G::IP, ZP = G::ZP, ST = G::ST;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    reduce_entry_inline(p.pre);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ai = wave & 1, bj = (wave >> 1) % NB, kg = (wave >> 1) / NB;
    const int Cin = p.C1 + p.C2;
    const int cib = Cin / 32, cob = p.Cout / CB;
    int bid = xcd_order(blockIdx.x);
    const int tile_id = bid % (cib * cob);
    const int ci0 = (bid % cib) * 32;
    bid /= cib;
    const int co0 = (bid % cob) * CB;
    const int split = bid / cob;
    const int q = lane >> 4, g = lane & 15;
    // this ci block's tensor (decoder layers read [up, skip] as two tensors; C1 % 32 == 0)
    const bool second = ci0 >= p.C1;
    const void* inp = second ? p.in2 : p.in1;
    const int Cx = second ? p.C2 : p.C1, cx0 = second ? ci0 - p.C1 : ci0;

    f32x4 acc[2][9];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[f][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};
    const bool do_bias = p.db_partial && ci0 == 0 && ai == 0;

    const int tiles = p.tiles_y * p.tiles_x;
    const int work_total = p.N * tiles;
    const int w_begin = split * p.work_per_split;
    const int w_end = min(work_total, w_begin + p.work_per_split);

    // ---- staging maps.  Halo tile: 72 threads per row (18 pixels x 4 eight-channel slots), IROWS rows per pass
    const int ihy0 = tid < G::IROWS * 72 ? tid / 72 : 100000, ihx = (tid % 72) >> 2;
    const int ic8 = (tid & 3) * 8;
    const int icommit = ihx * 64 + (tid & 3) * 16;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(inp), 0, (int)((long)p.N * p.H * p.W * Cx * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.dz), 0, (int)((long)p.N * p.H * p.W * p.Cout * 2), 0x00020000);
    u32x4 preI[2][IP], preZ[2][ZP];          // two tiles in flight
    auto fetch = [&](int wk, u32x4 (&pi)[IP], u32x4 (&pz)[ZP]) {
        const int n = wk / tiles, tile = wk - n * tiles;
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        const int iy0 = ty * TH - 1, ix0 = tx * 16 - 1;
        const int gx = ix0 + ihx;
        const bool okx = (unsigned)gx < (unsigned)p.W;
#pragma unroll
        for (int i = 0; i < IP; ++i) {
            const int gy = iy0 + ihy0 + G::IROWS * i;
            const bool ok = okx & ((unsigned)gy < (unsigned)p.H) & (ihy0 + G::IROWS * i < G::THH);
            const unsigned off = ok ? (unsigned)((((n * p.H + gy) * p.W + gx) * Cx + cx0 + ic8) * 2) : 0x80000000u;
            pi[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < ZP; ++i) {
            const int item = tid + i * 512, px = item / (4 * NB), slot = item % (4 * NB);
            const unsigned e = (unsigned)((((n * p.H + ty * TH + (px >> 4)) * p.W + tx * 16 + (px & 15)) * p.Cout + co0 + slot * 8) * 2);
            pz[i] = __builtin_amdgcn_raw_buffer_load_b128(rz, e, 0, 0);
        }
    };
    auto commit = [&](unsigned char* buf, const u32x4 (&pi)[IP], const u32x4 (&pz)[ZP]) {
#pragma unroll
        for (int i = 0; i < IP; ++i)
            if (ihy0 + G::IROWS * i < G::THH) *reinterpret_cast<u32x4*>(buf + (ihy0 + G::IROWS * i) * IRS + icommit) = pi[i];
        unsigned char* zb = buf + G::IBYTES;
#pragma unroll
        for (int i = 0; i < ZP; ++i) {
            const int item = tid + i * 512, px = item / (4 * NB), slot = item % (4 * NB);
            *reinterpret_cast<u32x4*>(zb + (px >> 4) * ZRS + (px & 15) * ZS + slot * 16) = pz[i];
        }
    };
    const int a_lane = q * IRS + (g >> 2) * 64 + ai * 32 + (g & 3) * 8;
    const int z_lane = q * ZRS + (g >> 2) * ZS + bj * 64 + (g & 3) * 8;
    auto tr4 = [](const unsigned char* a) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a);
    };

    // prologue: tile 0 committed, tile 1 in flight
    if (w_begin < w_end) {
        fetch(w_begin, preI[0], preZ[0]);
        if (w_begin + 1 < w_end) fetch(w_begin + 1, preI[1], preZ[1]);
        commit(smem_raw, preI[0], preZ[0]);
    }
    __syncthreads();
    auto tile_body = [&](int wk, int par, u32x4 (&pi_next)[IP], u32x4 (&pz_next)[ZP], u32x4 (&pi_far)[IP], u32x4 (&pz_far)[ZP]) {
        // entering: LDS buffer `par` holds tile wk, (pi_next, pz_next) hold tile wk + 1 (in flight or landed), the other register
        // set is free: it takes tile wk + 2 now
        if (wk + 2 < w_end) fetch(wk + 2, pi_far, pz_far);
        const unsigned char* sI = smem_raw + par * G::BUF;
        const unsigned char* sZ = sI + G::IBYTES;
        auto readA = [&](const unsigned char* ir, unsigned (&d)[10]) {
            const s16x4 e0 = tr4(ir), e1 = tr4(ir + 256), e2 = tr4(ir + 512);    // pixels 0-3, 4-7, 8-11 (8, 9 used)
            const s16x4 m0 = tr4(ir + 128), m1 = tr4(ir + 384);                 // pixels 2-5, 6-9
            d[0] = ((const unsigned*)&e0)[0]; d[1] = ((const unsigned*)&e0)[1];
            d[2] = ((const unsigned*)&e1)[0]; d[3] = ((const unsigned*)&e1)[1];
            d[4] = ((const unsigned*)&e2)[0]; d[5] = ((const unsigned*)&e2)[1];
            d[6] = ((const unsigned*)&m0)[0]; d[7] = ((const unsigned*)&m0)[1];
            d[8] = ((const unsigned*)&m1)[0]; d[9] = ((const unsigned*)&m1)[1];
        };
        auto readB = [&](const unsigned char* zr, unsigned (&b)[2][4]) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const s16x4 b0 = tr4(zr + f * 32), b1 = tr4(zr + f * 32 + 4 * ZS);
                b[f][0] = ((const unsigned*)&b0)[0]; b[f][1] = ((const unsigned*)&b0)[1];
                b[f][2] = ((const unsigned*)&b1)[0]; b[f][3] = ((const unsigned*)&b1)[1];
            }
        };
        unsigned dcur[10], bcur[2][4];
        readA(sI + a_lane + (kg >> 1) * (4 * IRS) + (kg & 1) * 512, dcur);
        readB(sZ + z_lane + (kg >> 1) * (4 * ZRS) + (kg & 1) * (8 * ZS), bcur);
#pragma unroll 1
        for (int st = kg; st < ST; st += KG) {
            const int sn = st + KG < ST ? st + KG : st;              // last step: a harmless re-read
            const unsigned char* ia = sI + a_lane + (st >> 1) * (4 * IRS) + (st & 1) * 512;
            const unsigned char* ian = sI + a_lane + (sn >> 1) * (4 * IRS) + (sn & 1) * 512;
            const unsigned char* zan = sZ + z_lane + (sn >> 1) * (4 * ZRS) + (sn & 1) * (8 * ZS);
            unsigned bnxt[2][4];
            const bf16x8 bfr[2] = {*reinterpret_cast<const bf16x8*>(bcur[0]), *reinterpret_cast<const bf16x8*>(bcur[1])};
            if (do_bias) {
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        bsum[f] += __uint_as_float(bcur[f][e] << 16) + __uint_as_float(bcur[f][e] & 0xffff0000u);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                unsigned dnxt[10];
                if (ky == 1) readB(zan, bnxt);
                readA(ky < 2 ? ia + (ky + 1) * IRS : ian, dnxt);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    unsigned a4[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (kx == 0) a4[k] = dcur[k];
                        else if (kx == 1) a4[k] = __builtin_amdgcn_alignbit(dcur[k + 1], dcur[k], 16);
                        else a4[k] = dcur[6 + k];
                    }
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(a4);
#pragma unroll
                    for (int f = 0; f < 2; ++f)
                        acc[f][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[f], acc[f][ky * 3 + kx], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) {        // one MFMA, one operand shift, one or two LDS requests at a time
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (ky == 1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int k = 0; k < 10; ++k) dcur[k] = dnxt[k];
            }
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int k = 0; k < 4; ++k) bcur[f][k] = bnxt[f][k];
        }
        if (wk + 1 < w_end) commit(smem_raw + (par ^ 1) * G::BUF, pi_next, pz_next);
        __syncthreads();
    };
    for (int wk = w_begin; wk < w_end; wk += 2) {       // two tiles per trip: the register sets alternate without copies
        tile_body(wk, 0, preI[1], preZ[1], preI[0], preZ[0]);
        if (wk + 1 < w_end) tile_body(wk + 1, 1, preI[0], preZ[0], preI[1], preZ[1]);
    }
    // ---- fold the row groups (same block, disjoint pixel rows) through LDS: group k parks, group 0 adds, k = 1 .. KG - 1
    if constexpr (KG > 1) {
        float* red = reinterpret_cast<float*>(smem_raw);
        const int slot = ai + 2 * bj;                             // this wave's place inside its row group (8 / KG waves each)
        constexpr int WSZ = 18 * 4 * 64 + 2 * 64;                 // floats per parked wave: 18 accumulators + 2 bias sums
#pragma unroll 1
        for (int k = 1; k < KG; ++k) {
            if (kg == k) {
#pragma unroll
                for (int f = 0; f < 2; ++f) {
#pragma unroll
                    for (int t = 0; t < 9; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j) red[slot * WSZ + ((f * 9 + t) * 4 + j) * 64 + lane] = acc[f][t][j];
                    red[slot * WSZ + 18 * 4 * 64 + f * 64 + lane] = bsum[f];
                }
            }
            __syncthreads();
            if (kg == 0) {
#pragma unroll
                for (int f = 0; f < 2; ++f) {
#pragma unroll
                    for (int t = 0; t < 9; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[f][t][j] += red[slot * WSZ + ((f * 9 + t) * 4 + j) * 64 + lane];
                    bsum[f] += red[slot * WSZ + 18 * 4 * 64 + f * 64 + lane];
                }
            }
            __syncthreads();
        }
    }
    const bool writer = KG == 1 || kg == 0;
    if (writer && do_bias) {                    // lanes g, g + 16, g + 32, g + 48 hold the four pixel groups of one channel
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            float v = bsum[f];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16) p.db_partial[(long)split * p.Cout + co0 + bj * 32 + f * 16 + lane] = v;
        }
    }
    float* slab = p.partial + (long)split * 9 * Cin * p.Cout;
    if (writer) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    slab[((long)t * Cin + ci0 + 16 * ai + 4 * q + j) * p.Cout + co0 + bj * 32 + f * 16 + g] = acc[f][t][j];
    }
    if (p.tickets == nullptr) return;
    // ---- the last workgroup of this (ci block, co block) tile to get here sums the tile over the splits, in a fixed order
    TicketJob job;
    job.cnt = p.tickets + (long)tile_id * ticket_words_per_tile_dev(p.splits, p.group);
    job.slab[0] = p.partial; job.stride[0] = (long)9 * Cin * p.Cout; job.dst[0] = p.dw;
    job.slab[1] = p.db_partial; job.stride[1] = p.Cout; job.dst[1] = p.db;
    job.splits = p.splits; job.group = p.group; job.accumulate = p.accumulate;
    constexpr int C4 = CB / 4;
    const int items = 9 * 32 * C4 + ((p.db_partial && ci0 == 0) ? C4 : 0);
    const int Cout = p.Cout;
    ticket_finish<512>(job, split, items, [=](int it) {
        TicketItem m;
        if (it >= 9 * 32 * C4) { m.which = 1; m.off = co0 + (it - 9 * 32 * C4) * 4; return m; }
        const int c4 = it % C4, row = it / C4;                   // row = t * 32 + r
        m.which = 0;
        m.off = ((long)(row >> 5) * Cin + ci0 + (row & 31)) * Cout + co0 + c4 * 4;
        return m;
    }, reinterpret_cast<unsigned*>(smem_raw));
}

template <int NB, int TH>
int launch(Wg3Params p, int max_slabs, hipStream_t stream) {
    using G = Wg3Geom<NB, TH>;
    p.tiles_y = p.H / TH;
    p.tiles_x = p.W / 16;
    const long work = (long)p.N * p.tiles_y * p.tiles_x;
    const long blocks_io = (long)((p.C1 + p.C2) / 32) * (p.Cout / G::CB);
    static const int target_env = getenv("NIMG_WGRAD3_BLOCKS") ? atoi(getenv("NIMG_WGRAD3_BLOCKS")) : 0;
    const int target = target_env > 0 ? target_env : 256;            // 72 AGPRs + ~100 VGPRs: one 8-wave workgroup per CU
    long splits = (target + blocks_io - 1) / blocks_io;
    if (splits > max_slabs) splits = max_slabs;
    if (splits > work) splits = work;
    if (splits < 1) return 0;
    const long wps = (work + splits - 1) / splits;
    splits = (work + wps - 1) / wps;
    p.work_per_split = (int)wps;
    // in-kernel finish: counters bound to this stream (nimg_bind_tickets), 16-byte aligned destinations
    p.splits = (int)splits;
    p.group = ticket_group((int)splits);
    const size_t words = (size_t)blocks_io * ticket_words_per_tile((int)splits);
    if (p.dw && !(((uintptr_t)p.dw | (uintptr_t)p.db) & 15)) p.tickets = nimg_internal_tickets(stream, words);
    auto kern = conv3_wgrad_alltaps_kernel<NB, TH>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    hipLaunchKernelGGL(kern, dim3((unsigned)(blocks_io * splits)), dim3(512), G::LDS, stream, p);
    if (hipGetLastError() != hipSuccess) return -1;
    return p.tickets ? -2 - (int)splits : (int)splits;            // <= -3: finished in-kernel, nothing left to reduce
}

}  // namespace

// Weight-gradient slabs (+ bias partials) of a 3x3 / stride 1 / SAME (zero padding) layer from its bf16 input(s) and bf16 output
// gradient.  Returns the number of slabs written to partial[slab][9][c1 + c2][cout] (db_partial[slab][cout]), 0 when the shape is
// not this kernel's (the caller falls back to conv_wgrad_bf16_kernel), -1 on a launch error.  With dw != null and arrival
// counters bound to the stream the sums are FINISHED in the kernel (dw / db written or accumulated into): returns -2 - slabs.
int nimg_internal_wgrad3_alltaps(const void* in1, int c1, const void* in2, int c2, const void* dz, int cout, float* partial,
                                 float* db_partial, int n, int h, int wd, int max_slabs, hipStream_t stream, float* dw, float* db,
                                 int accumulate, const void* pre) {
    if (getenv("NIMG_NO_WGRAD3_ALLTAPS") != nullptr) return 0;
    if ((c1 % 32) || (c2 % 32) || (cout % 32) || (wd % 16) || (h % 8) || max_slabs < 1 || (c2 > 0 && !in2)) return 0;
    const long px = (long)n * h * wd;
    if (px * (c1 > c2 ? c1 : c2) * 2 >= (1l << 31) - 65536 || px * cout * 2 >= (1l << 31) - 65536) return 0;
    Wg3Params p;
    p.in1 = in1; p.in2 = in2; p.dz = dz; p.partial = partial; p.db_partial = db_partial;
    p.C1 = c1; p.C2 = c2; p.Cout = cout; p.N = n; p.H = h; p.W = wd;
    p.tiles_y = p.tiles_x = p.work_per_split = 0;
    p.tickets = nullptr; p.dw = dw; p.db = db_partial ? db : nullptr; p.splits = p.group = 0; p.accumulate = accumulate;
    p.pre = pre ? *reinterpret_cast<const ReduceEntry*>(pre) : empty_reduce_entry();
    // output channels per workgroup: 32 NB.  A/B: NIMG_WGRAD3_NB = 1 | 2 | 4 forces the block width (read per call)
    const char* nb_env = getenv("NIMG_WGRAD3_NB");
    int nb = nb_env ? atoi(nb_env) : 0;
    if (nb != 1 && nb != 2 && nb != 4) nb = cout % 128 == 0 ? 4 : (cout % 64 == 0 ? 2 : 1);
    while (nb > 1 && cout % (32 * nb)) nb >>= 1;
    if (nb == 4) return launch<4, 8>(p, max_slabs, stream);
    if (nb == 2) return launch<2, 8>(p, max_slabs, stream);
    return (h % 16 == 0) ? launch<1, 16>(p, max_slabs, stream) : launch<1, 8>(p, max_slabs, stream);
}
