// 5x5 weight gradient of the FAN's fused conv + pool layers (models/forensics.py:69-77 under tape.gradient, :118-124),
// throughput mode, "all taps in one wave" form (round 3).  dW[ky][kx][ci][co] = sum over pixels of in[y+ky-2][x+kx-2][ci] *
// dz[y][x][co], dz = the 2x2 un-pooling of (pooled gradient, arg-max bytes); GEMM view: M = ci, N = co, K = pixels.
//
// conv_wgrad_bf16_kernel<5, ...> (conv_bf16.hip) deals the 25 taps to 8 waves: every wave re-reads the dz fragment and one
// shifted input fragment per tap (12 ds_read_b64_tr_b16 per 8 MFMAs); all 8 waves of the ONE workgroup a CU holds stage the
// next tile at the same time behind two barriers (nothing multiplies meanwhile), and a tile costs ~500 instructions of address
// arithmetic per wave.  Here a wave owns ALL 25 taps of one 16 ci x 32 co block as 25 x 2 accumulators of
// v_mfma_f32_16x16x32_bf16 (200 AGPRs.  hipcc keeps MFMA accumulators in the AGPR half only: the 32 ci x 32 co block of
// v_mfma_f32_32x32x16_bf16 - 400 accumulator registers, half the instructions per FLOP - compiles to v_accvgpr_read / _write
// shuffles between the halves, ~30 per MFMA, or to scratch spills.  Splitting the accumulators by hand - 16 taps through the
// builtin (AGPRs), 9 through an inline-asm MFMA with a VGPR-constrained accumulator and its own s_nop wait states - compiles
// clean and is correct, but measured 8 % SLOWER (profiles/r03_f_wgrad5_m32_ab.txt: 1.58 vs 1.47 ms for the three layers): its
// 8-row tiles double the staging instructions, which one wave per SIMD cannot hide.  One wave per SIMD on the 512-entry file) and walks a tile 8 columns x 4 rows (K = 32 pixels) at a time:
//   * the two dz fragments of the step (4 transpose reads) feed 50 MFMAs;
//   * per kernel row ky the input operand is read as 12 consecutive pixels per K group (3 transpose reads: 8 own + 4 halo):
//     kx = 0 / 4 are its first / last 8 pixels, kx = 1 / 3 that register window shifted by one pixel (K = pixels, two bf16 per
//     register: v_alignbit_b32), kx = 2 one more 8-pixel read, and every shifted operand serves both dz fragments - 29 reads +
//     40 VALU per 50 MFMAs instead of 150 reads.  (With ONE wave per SIMD the instruction count is the budget: a wave issues
//     about one instruction per 4 cycles, a 16x16x32 MFMA occupies the pipe for 16);
//   * tiles are double-buffered in LDS: the global loads of tile t+1 are issued before the MFMA loop of tile t and committed
//     behind it, ONE barrier per tile; halo pixels outside the image are out-of-range buffer offsets (the hardware returns 0);
//   * pooled gradient + arg-max bytes are fetched once per POOLED pixel and routed to the four window positions on the way
//     into LDS (the 8-wave kernel fetches them once per full-resolution pixel);
//   * the bias gradient (= the plain sum of the pooled gradient) is summed from the dz fragments the ci-block-0 workgroups
//     hold anyway.
// Workgroup = 4 waves = 32 ci x 64 co (wave w: input channels 16 (w & 1) .. +15, output channels 32 (w >> 1) .. +31) x a split
// of the pixels; one slab per workgroup, the usual fixed-order slab reduction finishes.
//
// LDS layouts (both pixel-major, what ds_read_b64_tr_b16 turns into K-major fragments): input [halo row][20 pixels][32 ci],
// 64 B per pixel; dz [row][16 pixels][64 co], 192 B per pixel (4 consecutive pixels -> 4 bank quarters).  The four K groups of
// an MFMA are the same 8 columns of four consecutive tile rows.  A 16-channel fragment uses 32 of a pixel's bytes, so a
// 32-lane read group (two K groups = two rows) is conflict-free when the rows sit 32 B apart mod 256: row strides 1312 B
// (20 x 64 + 32) and 3104 B (16 x 192 + 32).  That holds at ANY pixel offset, so the kx = 2 tap is a second, 2-pixel-shifted
// read (a re-numbered register window would have to be copied: MFMA operand tuples are 64-bit aligned).
#include <stdlib.h>

#include "common.h"

namespace {

using namespace nimg;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct Wg5Params {
    const void* in;                // (N, H, W, Cin) bf16
    const void* g;                 // (N, H/2, W/2, Cout) bf16 pooled gradient (already x LeakyReLU')
    const unsigned char* idx;      // arg-max bytes of the forward max-pool, same shape
    float* partial;                // [slabs][25][Cin][Cout]
    float* db_partial;             // [slabs][Cout] or null
    int Cin, Cout, N, H, W;
    int tiles_y, tiles_x, work_per_split;
};

// 0xFF in every byte of k equal to pos; route = keep a channel's gradient iff its arg-max byte names this window position
__device__ __forceinline__ unsigned eq_bytes(unsigned k, unsigned pos) {
    const unsigned x = k ^ (pos * 0x01010101u);
    return (((x | (x >> 1)) & 0x01010101u) ^ 0x01010101u) * 0xFFu;
}
__device__ __forceinline__ u32x4 route(u32x4 g, unsigned k0, unsigned k1, unsigned pos) {
    const unsigned m0 = eq_bytes(k0, pos), m1 = eq_bytes(k1, pos);
    u32x4 r;
    r[0] = g[0] & __builtin_amdgcn_perm(m0, m0, 0x01010000u);
    r[1] = g[1] & __builtin_amdgcn_perm(m0, m0, 0x03030202u);
    r[2] = g[2] & __builtin_amdgcn_perm(m1, m1, 0x01010000u);
    r[3] = g[3] & __builtin_amdgcn_perm(m1, m1, 0x03030202u);
    return r;
}

template <int TH>
struct Wg5Geom {
    static constexpr int CB = 64;
    static constexpr int THH = TH + 4, TWH = 20, NPIXH = THH * TWH, NPIX = TH * 16;
    static constexpr int ZS = 192, IRS = TWH * 64 + 32, ZRS = 16 * ZS + 32;       // pixel / row strides (see the header)
    static constexpr int IBYTES = THH * IRS, ZBYTES = TH * ZRS, BUF = IBYTES + ZBYTES;
    static constexpr int IP = (THH + 2) / 3;                     // input passes: 3 halo rows (80 threads each) per pass
    static constexpr int ZITEMS = (TH / 2) * 8 * 8, ZP = ZITEMS / 256;            // pooled items (8 channels) per thread
    static constexpr size_t LDS = (size_t)2 * BUF;
    static_assert(ZITEMS % 256 == 0 && TH % 4 == 0, "pooled tile divides over the threads; K step = 4 rows");
    static_assert(LDS <= 160 * 1024, "two tile buffers fit the LDS");
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// KX3L: the kx = 3 operand comes from two more LDS reads (3-pixel offset) instead of four more v_alignbit (A/B switch)
// SCHED: how the LDS requests of the next kernel row are placed among the MFMAs of the current one.  Left to itself hipcc sinks
// them next to their first use and waits lgkmcnt(0) there (one exposed LDS latency per kernel row with one wave per SIMD).
// 1: sched_barrier behind the request block (requests first, then the row's MFMAs); 2: sched_group_barrier pipeline, one
// request + two operand shifts per MFMA.
template <int TH, bool KX3L, int SCHED>
__global__ __launch_bounds__(256, 1) void conv5_wgrad_alltaps_kernel(const Wg5Params p) {
    using G = Wg5Geom<TH>;
    constexpr int ZS = G::ZS, IRS = G::IRS, ZRS = G::ZRS, IP = G::IP, ZP = G::ZP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ai = wave & 1, bj = wave >> 1;                       // this wave's 16 input channels, its 32 output channels
    const int cib = p.Cin / 32, cob = p.Cout / 64;
    int bid = xcd_order(blockIdx.x);
    const int ci0 = (bid % cib) * 32;
    bid /= cib;
    const int co0 = (bid % cob) * 64;
    const int split = bid / cob;
    const int q = lane >> 4, g = lane & 15;                        // K group (= tile row of the step), row / column of the fragment
    const int Hp = p.H >> 1, Wp = p.W >> 1;

    f32x4 acc[2][25];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int t = 0; t < 25; ++t) acc[f][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};
    const bool do_bias = p.db_partial && ci0 == 0 && ai == 0;

    const int tiles = p.tiles_y * p.tiles_x;
    const int work_total = p.N * tiles;
    const int w_begin = split * p.work_per_split;
    const int w_end = min(work_total, w_begin + p.work_per_split);

    // ---- staging maps, constant per thread.  Halo tile: 80 threads per row (20 pixels x 4 eight-channel slots), 3 rows per
    // pass (threads 240..255 idle): the row of pass q is 3 q + tid / 80 - nothing per pass to keep in registers
    const int ihy0 = tid < 240 ? tid / 80 : 100000, ihx = (tid % 80) >> 2;
    const int ic8 = (tid & 3) * 8;
    const int icommit = ihx * 64 + (tid & 3) * 16;
    const int zc8 = (tid & 7) * 8;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.in), 0, (int)((long)p.N * p.H * p.W * p.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.g), 0, (int)((long)p.N * Hp * Wp * p.Cout * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(p.idx), 0, (int)((long)p.N * Hp * Wp * p.Cout), 0x00020000);
    u32x4 preI[IP], preZ[ZP];
    u32x2 preK[ZP];
    auto fetch = [&](int wk) {
        const int n = wk / tiles, tile = wk - n * tiles;
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        const int iy0 = ty * TH - 2, ix0 = tx * 16 - 2;
        const int gx = ix0 + ihx;
        const bool okx = (unsigned)gx < (unsigned)p.W;
#pragma unroll
        for (int i = 0; i < IP; ++i) {
            const int gy = iy0 + ihy0 + 3 * i;
            const bool ok = okx & ((unsigned)gy < (unsigned)p.H) & (ihy0 + 3 * i < G::THH);
            const unsigned off = ok ? (unsigned)((((n * p.H + gy) * p.W + gx) * p.Cin + ci0 + ic8) * 2) : 0x80000000u;
            preI[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < ZP; ++i) {
            const int ppix = (tid + i * 256) >> 3;                 // pooled pixel of the (TH/2) x 8 pooled tile
            const unsigned e = (unsigned)(((n * Hp + ty * (TH / 2) + (ppix >> 3)) * Wp + tx * 8 + (ppix & 7)) * p.Cout + co0 + zc8);
            preZ[i] = __builtin_amdgcn_raw_buffer_load_b128(rz, e * 2, 0, 0);
            preK[i] = __builtin_amdgcn_raw_buffer_load_b64(rk, e, 0, 0);
        }
    };
    auto commit = [&](unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < IP; ++i)
            if (ihy0 + 3 * i < G::THH) *reinterpret_cast<u32x4*>(buf + (ihy0 + 3 * i) * IRS + icommit) = preI[i];
        unsigned char* zb = buf + G::IBYTES;
#pragma unroll
        for (int i = 0; i < ZP; ++i) {
            const int ppix = (tid + i * 256) >> 3;
            unsigned char* zp = zb + ((ppix >> 3) * 2) * ZRS + ((ppix & 7) * 2) * ZS + zc8 * 2;       // window position 0
#pragma unroll
            for (int pos = 0; pos < 4; ++pos)
                *reinterpret_cast<u32x4*>(zp + (pos >> 1) * ZRS + (pos & 1) * ZS) = route(preZ[i], preK[i][0], preK[i][1], (unsigned)pos);
        }
    };
    // per-lane parts of the transpose-read addresses (ds_read_b64_tr_b16, semantics in conv_bf16.hip "tr_read8"): inside a
    // 16-lane group lane g addresses pixel (g >> 2) of 4, channels (g & 3)*4 .. +3 of a 16-channel slot, and receives channel
    // g of the 4 pixels - the K(= pixel)-major fragment both MFMA operands need, straight from the pixel-major tiles.
    const int a_lane = q * IRS + (g >> 2) * 64 + ai * 32 + (g & 3) * 8;    // + step row / column, + ky * IRS, + pixel block
    const int z_lane = q * ZRS + (g >> 2) * ZS + bj * 64 + (g & 3) * 8;    // + f * 32
    auto tr4 = [](const unsigned char* a) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a);
    };

    if (w_begin < w_end) {
        fetch(w_begin);
        commit(smem_raw);
    }
    __syncthreads();
    for (int wk = w_begin; wk < w_end; ++wk) {
        const int par = (wk - w_begin) & 1;
        const bool more = wk + 1 < w_end;
        if (more) fetch(wk + 1);
        const unsigned char* sI = smem_raw + par * G::BUF;
        const unsigned char* sZ = sI + G::IBYTES;
        // software pipeline: the operands of the next kernel row (or of the next step) are requested before the ten MFMAs of
        // the current one - with ONE wave per SIMD nobody else covers the LDS latency
        constexpr int ND = KX3L ? 14 : 10;
        auto readA = [&](const unsigned char* ir, unsigned (&d)[ND]) {          // ir: this lane's pixel 0 of the halo row
            const s16x4 e0 = tr4(ir), e1 = tr4(ir + 256), e2 = tr4(ir + 512);    // pixels 0-3, 4-7, 8-11
            const s16x4 m0 = tr4(ir + 128), m1 = tr4(ir + 384);                 // pixels 2-5, 6-9
            d[0] = ((const unsigned*)&e0)[0]; d[1] = ((const unsigned*)&e0)[1];
            d[2] = ((const unsigned*)&e1)[0]; d[3] = ((const unsigned*)&e1)[1];
            d[4] = ((const unsigned*)&e2)[0]; d[5] = ((const unsigned*)&e2)[1];
            d[6] = ((const unsigned*)&m0)[0]; d[7] = ((const unsigned*)&m0)[1];
            d[8] = ((const unsigned*)&m1)[0]; d[9] = ((const unsigned*)&m1)[1];
            if constexpr (KX3L) {
                const s16x4 o0 = tr4(ir + 192), o1 = tr4(ir + 448);             // pixels 3-6, 7-10
                d[10] = ((const unsigned*)&o0)[0]; d[11] = ((const unsigned*)&o0)[1];
                d[12] = ((const unsigned*)&o1)[0]; d[13] = ((const unsigned*)&o1)[1];
            }
        };
        auto readB = [&](const unsigned char* zr, unsigned (&b)[2][4]) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const s16x4 b0 = tr4(zr + f * 32), b1 = tr4(zr + f * 32 + 4 * ZS);
                b[f][0] = ((const unsigned*)&b0)[0]; b[f][1] = ((const unsigned*)&b0)[1];
                b[f][2] = ((const unsigned*)&b1)[0]; b[f][3] = ((const unsigned*)&b1)[1];
            }
        };
        constexpr int STEPS = TH / 2;                                // (row group of 4, column half): 4 rows x 8 columns each
        unsigned dcur[ND], bcur[2][4];
        readA(sI + a_lane, dcur);
        readB(sZ + z_lane, bcur);
#pragma unroll 1
        for (int st = 0; st < STEPS; ++st) {
            const int sn = st + 1 < STEPS ? st + 1 : st;            // last step: a harmless re-read
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
            for (int ky = 0; ky < 5; ++ky) {
                unsigned dnxt[ND];
                if (ky == 2) readB(zan, bnxt);                       // the next step's dz fragments: early, its first MFMA needs them
                readA(ky < 4 ? ia + (ky + 1) * IRS : ian, dnxt);
                if constexpr (SCHED == 1) __builtin_amdgcn_sched_barrier(0);          // the requests stay in front of this row's MFMAs
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    unsigned a4[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // pixel window [kx, kx + 8): registers E = d[0..5] hold pixels 0..11, M = d[6..9] pixels 2..9
                        if (kx == 0) a4[k] = dcur[k];
                        else if (kx == 1) a4[k] = __builtin_amdgcn_alignbit(dcur[k + 1], dcur[k], 16);
                        else if (kx == 2) a4[k] = dcur[6 + k];
                        else if (kx == 3) a4[k] = KX3L ? dcur[10 + k]
                                                       : __builtin_amdgcn_alignbit(k < 3 ? dcur[7 + k] : dcur[5], dcur[6 + k], 16);
                        else a4[k] = dcur[2 + k];
                    }
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(a4);
#pragma unroll
                    for (int f = 0; f < 2; ++f)
                        acc[f][ky * 5 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[f], acc[f][ky * 5 + kx], 0, 0, 0);
                }
                if constexpr (SCHED == 2) {
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                  // VALU (operand shifts)
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                  // DS read
                    }
                }
                if constexpr (SCHED == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < ND; ++k) dcur[k] = dnxt[k];
            }
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int k = 0; k < 4; ++k) bcur[f][k] = bnxt[f][k];
        }
        if (more) commit(smem_raw + (par ^ 1) * G::BUF);
        __syncthreads();
    }
    if (do_bias) {                              // lanes g, g + 16, g + 32, g + 48 hold the four pixel groups of one channel
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            float v = bsum[f];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16) p.db_partial[(long)split * p.Cout + co0 + bj * 32 + f * 16 + lane] = v;
        }
    }
    float* slab = p.partial + (long)split * 25 * p.Cin * p.Cout;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int t = 0; t < 25; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                slab[((long)t * p.Cin + ci0 + 16 * ai + 4 * q + j) * p.Cout + co0 + bj * 32 + f * 16 + g] = acc[f][t][j];
}

template <int TH, bool KX3L, int SCHED>
int launch(Wg5Params p, int max_slabs, hipStream_t stream) {
    using G = Wg5Geom<TH>;
    p.tiles_y = p.H / TH;
    p.tiles_x = p.W / 16;
    const long work = (long)p.N * p.tiles_y * p.tiles_x;
    const long blocks_io = (long)(p.Cin / 32) * (p.Cout / 64);
    static const int target = getenv("NIMG_WGRAD5_ALLTAPS_BLOCKS") ? atoi(getenv("NIMG_WGRAD5_ALLTAPS_BLOCKS")) : 256;
    long splits = (target + blocks_io - 1) / blocks_io;
    if (splits > max_slabs) splits = max_slabs;
    if (splits > work) splits = work;
    if (splits < 1) return 0;
    const long wps = (work + splits - 1) / splits;
    splits = (work + wps - 1) / wps;
    p.work_per_split = (int)wps;
    auto kern = conv5_wgrad_alltaps_kernel<TH, KX3L, SCHED>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    hipLaunchKernelGGL(kern, dim3((unsigned)(blocks_io * splits)), dim3(256), G::LDS, stream, p);
    if (hipGetLastError() != hipSuccess) return -1;
    return (int)splits;
}




// ------------------------------------------------------------------------------------------------------------------
// Structured-sparsity form.  dz is the 2x2 UN-POOLING of the pooled gradient: of the four pixels of a pooling window exactly one
// carries a channel's gradient.  v_smfmac_f32_16x16x64_bf16 multiplies a 2:4-sparse A (two kept elements + two 2-bit positions
// per group of four K values) by a dense B over K = 64 in the cycles the dense 16x16x32 takes.  With K = pixels,
//     D[co][ci] += A[co][pixel] B[pixel][ci],     A = dz^T compressed, B = the (tap-shifted) input,
// and the K groups chosen as COLUMN STRIPS - the four pixels (r0..r3, c) of a step of four tile rows: the upper window (rows r0,
// r1) and the lower one (r2, r3) put at most one value each into the strip - the pooled tensor IS the compressed operand: kept
// element 0 = the upper window's gradient if its arg-max sits in this column (else 0) at position (arg-max row), kept element 1 =
// the lower window's at position 2 + (arg-max row).  No un-pooling pass, no routed tile in LDS (10 KB pooled instead of 49 KB),
// and a step covers 64 pixels with the 50 matrix instructions that covered 32.  (tools/probe/smfmac_probe.hip measured the
// operand layout used below - A lane group g holds logical K 16 g .. 16 g + 15, B lane group b element j holds K 8 b + j for
// j < 8 and 32 + 8 b + j - 8 above; the two kept elements of a group may carry ANY positions - and the issue rate: 1.92 x the
// dense-equivalent products of v_mfma_f32_32x32x16_bf16.)
// B operand of tap (ky, kx) for lane group b: the strips of columns 4 b + kx + {0, 1, 2, 3}, rows ky .. ky + 3 of the step - ONE
// ds_read_b64_tr_b16 per strip (its four "pixels" are the four rows), four reads = the eight operand registers, nothing to
// shift or copy.  A wave owns 16 input x ALL 64 output channels of the workgroup for HALF the taps (13 x 4 accumulators of
// four registers = 208): one operand (4 reads) feeds four matrix instructions - with two it was the LDS (4 waves x 4 reads x 2
// cycles per 32 cycles of matrix work) that bound the kernel.  The waves of a workgroup are (input-channel half) x (taps 0..12 |
// 12..24; tap 12 is computed twice and stored once).  (Sharing registers between taps costs hipcc a v_mov per shared
// register - its operand tuples cannot overlap - or, with two horizontal pixels per register, a v_alignbit per odd tap: the
// first version of this kernel, 60 + 58 VALU per 50 instructions, profiles/r03_z_wgrad5_sparse_v1_ab.txt.)
// Logical K groups of lane group b: 2 b, 2 b + 1 (j < 8) and 8 + 2 b, 9 + 2 b = columns 4 b .. 4 b + 3, so A lane group g lists
// the columns c, c + 1, c + 4, c + 5 with c = {0, 8, 2, 10}[g] = both column parities of the pooled pixels (upper | lower, c / 2)
// and (upper | lower, c / 2 + 2): one transpose read of the pooled tile gives (upper, lower) pairs, a 16-bit column-parity mask
// tile splits them over the even / odd column strip, and the index word comes the same way from a 16-bit tile of pre-shifted
// position pieces (both written by the staging pass from the arg-max bytes).
// LDS: input pixels 96 B apart, rows 2080 B; pooled pixels 160 B apart, rows 1312 B - the eight 32-byte pieces of every
// 32-lane read fall into the eight bank octets.
template <int TH>
struct Wg5SGeom {
    static constexpr int THH = TH + 4, TWH = 20;
    static constexpr int XP = 96, XR = TWH * XP + 160;            // input pixel / row stride
    static constexpr int GP = 160, GR = 8 * GP + 32;              // pooled pixel / row stride (gradient, mask and index tiles alike)
    static constexpr int IBYTES = THH * XR, GBYTES = (TH / 2) * GR, BUF = IBYTES + 3 * GBYTES;
    static constexpr int IP = (THH + 2) / 3;
    static constexpr int ZITEMS = (TH / 2) * 8 * 8, ZP = ZITEMS / 256;
    static constexpr size_t LDS = (size_t)2 * BUF;
    static_assert(ZITEMS % 256 == 0 && TH % 4 == 0, "pooled tile divides over the threads; a step is 4 rows");
    static_assert(LDS <= 160 * 1024, "two tile buffers fit the LDS");
};
typedef __bf16 bf16x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));

template <int TH>
__global__ __launch_bounds__(256, 1) void conv5_wgrad_sparse_kernel(const Wg5Params p) {
    using G = Wg5SGeom<TH>;
    constexpr int XP = G::XP, XR = G::XR, GP = G::GP, GR = G::GR, IP = G::IP, ZP = G::ZP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ai = wave & 1, th = wave >> 1;                       // this wave's 16 input channels, its half of the taps
    const int tap0 = 12 * th;                                      // taps tap0 .. tap0 + 12
    const int cib = p.Cin / 32, cob = p.Cout / 64;
    int bid = xcd_order(blockIdx.x);
    const int ci0 = (bid % cib) * 32;
    bid /= cib;
    const int co0 = (bid % cob) * 64;
    const int split = bid / cob;
    const int q = lane >> 4, g = lane & 15;                        // lane group (K range), row / column of the fragment
    const int Hp = p.H >> 1, Wp = p.W >> 1;

    constexpr int NF = 4, NTAP = 13;
    f32x4 acc[NF][NTAP];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NTAP; ++t) acc[f][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[NF] = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.db_partial && ci0 == 0 && ai == 0 && th == 0;

    const int tiles = p.tiles_y * p.tiles_x;
    const int work_total = p.N * tiles;
    const int w_begin = split * p.work_per_split;
    const int w_end = min(work_total, w_begin + p.work_per_split);

    // ---- staging maps (as conv5_wgrad_alltaps_kernel; the pooled gradient goes to LDS AS IT IS, next to the mask and index tiles)
    const int ihy0 = tid < 240 ? tid / 80 : 100000, ihx = (tid % 80) >> 2;
    const int ic8 = (tid & 3) * 8;
    const int icommit = ihx * XP + (tid & 3) * 16;
    const int zc8 = (tid & 7) * 8;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.in), 0, (int)((long)p.N * p.H * p.W * p.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.g), 0, (int)((long)p.N * Hp * Wp * p.Cout * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(p.idx), 0, (int)((long)p.N * Hp * Wp * p.Cout), 0x00020000);
    u32x4 preI[IP], preZ[ZP];
    u32x2 preK[ZP];
    auto fetch = [&](int wk) {
        const int n = wk / tiles, tile = wk - n * tiles;
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        const int iy0 = ty * TH - 2, ix0 = tx * 16 - 2;
        const int gx = ix0 + ihx;
        const bool okx = (unsigned)gx < (unsigned)p.W;
#pragma unroll
        for (int i = 0; i < IP; ++i) {
            const int gy = iy0 + ihy0 + 3 * i;
            const bool ok = okx & ((unsigned)gy < (unsigned)p.H) & (ihy0 + 3 * i < G::THH);
            const unsigned off = ok ? (unsigned)((((n * p.H + gy) * p.W + gx) * p.Cin + ci0 + ic8) * 2) : 0x80000000u;
            preI[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < ZP; ++i) {
            const int ppix = (tid + i * 256) >> 3;                 // pooled pixel of the (TH/2) x 8 pooled tile
            const unsigned e = (unsigned)(((n * Hp + ty * (TH / 2) + (ppix >> 3)) * Wp + tx * 8 + (ppix & 7)) * p.Cout + co0 + zc8);
            preZ[i] = __builtin_amdgcn_raw_buffer_load_b128(rz, e * 2, 0, 0);
            preK[i] = __builtin_amdgcn_raw_buffer_load_b64(rk, e, 0, 0);
        }
    };
    auto commit = [&](unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < IP; ++i)
            if (ihy0 + 3 * i < G::THH) *reinterpret_cast<u32x4*>(buf + (ihy0 + 3 * i) * XR + icommit) = preI[i];
        unsigned char* gb = buf + G::IBYTES;
#pragma unroll
        for (int i = 0; i < ZP; ++i) {
            const int ppix = (tid + i * 256) >> 3, pr = ppix >> 3, pc = ppix & 7;
            const int o = pr * GR + pc * GP + zc8 * 2;
            *reinterpret_cast<u32x4*>(gb + o) = preZ[i];
            // arg-max byte k = 2 (row) + (column) of the window, per channel -> two 16-bit tiles: the column-parity mask (0xffff:
            // the value belongs to the odd column strip) and this window's piece of the index word: position (row, or 2 + row
            // for the lower window of a step = odd pooled row) at the two kept slots it can occupy, (u, u + 2) + 4 x bit 1 of the
            // pooled column - the order the A fragment lists its windows in
            const unsigned u = (unsigned)(pr & 1), s8 = 8u * (unsigned)((pc >> 1) & 1);
            u32x4 m16, x16;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const unsigned e2 = __builtin_amdgcn_perm(0u, preK[i][d >> 1], (d & 1) ? 0x0c030c02u : 0x0c010c00u);     // two bytes -> two halves
                m16[d] = (e2 & 0x00010001u) * 0xffffu;
                const unsigned val = ((e2 >> 1) & 0x00010001u) + u * 0x00020002u;
                x16[d] = ((val << (2 * u)) | (val << (2 * u + 4))) << s8;
            }
            *reinterpret_cast<u32x4*>(gb + G::GBYTES + o) = m16;
            *reinterpret_cast<u32x4*>(gb + 2 * G::GBYTES + o) = x16;
        }
    };
    // transpose-read addresses.  Input strip: lane g of a 16-lane group supplies "pixel" j = g >> 2 = row j of the strip,
    // channels (g & 3) * 4 .. + 3 of its wave's 16; lane group q = image columns 4 q ..
    const int jx = g >> 2;
    const int x_lane = jx * XR + 4 * q * XP + ai * 32 + (g & 3) * 8;      // + (step row + ky) * XR + (kx + i) * XP
    // pooled tiles: lane group q lists the windows (upper, c), (lower, c), (upper, c + 2), (lower, c + 2), c = {0, 4, 1, 5}[q]
    const int pcA = (q & 1) * 4 + (q >> 1);
    const int z_lane = (jx & 1) * GR + (pcA + 2 * (jx >> 1)) * GP + (g & 3) * 8;               // + f * 32, + step * 2 * GR
    auto tr2 = [](const unsigned char* a) {
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a);
        return *reinterpret_cast<const u32x2*>(&v);
    };

    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem_raw;
    if (w_begin < w_end) {
        fetch(w_begin);
        commit(smem_raw);
    }
    __syncthreads();
    for (int wk = w_begin; wk < w_end; ++wk) {
        const int par = (wk - w_begin) & 1;
        const bool more = wk + 1 < w_end;
        if (more) fetch(wk + 1);
        const unsigned char* sI = smem_raw + par * G::BUF;
        const unsigned char* sG = sI + G::IBYTES;
        auto trl = [](unsigned lds_addr) {                 // transpose read at a 32-bit LDS address
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(unsigned long)lds_addr);
            return *reinterpret_cast<const u32x2*>(&v);
        };
        auto readB = [&](unsigned xr) {                    // xr: LDS address of this lane's first strip of the tap
            // neighbouring taps read the same strips: left to itself hipcc merges those loads and then copies every shared
            // register into each tap's operand tuple (120 v_accvgpr_mov per 50 matrix instructions) - an opaque address per tap
            // keeps the loads apart
            asm volatile("" : "+v"(xr));
            const u32x2 r0 = trl(xr), r1 = trl(xr + XP), r2 = trl(xr + 2 * XP), r3 = trl(xr + 3 * XP);
            const u32x4 q0 = __builtin_shufflevector(r0, r1, 0, 1, 2, 3), q1 = __builtin_shufflevector(r2, r3, 0, 1, 2, 3);
            const u32x8 b8 = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7);
            return *reinterpret_cast<const bf16x16*>(&b8);
        };
        auto readA = [&](const unsigned char* zr, unsigned (&a)[NF][4], unsigned (&ix)[NF]) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const u32x2 v = tr2(zr + f * 32), m = tr2(zr + G::GBYTES + f * 32), k = tr2(zr + 2 * G::GBYTES + f * 32);
                a[f][0] = v[0] & ~m[0]; a[f][1] = v[0] & m[0];           // (upper, lower) of pooled column c: even strip, odd strip
                a[f][2] = v[1] & ~m[1]; a[f][3] = v[1] & m[1];           // ... of pooled column c + 2
                const unsigned t = k[0] | k[1];
                ix[f] = t | (t >> 16);                                   // bits 15..0 are read
                // bias gradient = the plain sum of the pooled gradient: two v_dot2c_f32_bf16 against (1, 1), unconditionally - a
                // branch on do_bias here cuts the fragment reads of a tile into 16 blocks, each waiting for its own LDS round trip
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                const unsigned ones = 0x3f803f80u, v0 = v[0], v1 = v[1];
                bsum[f] = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2*>(&v0), *reinterpret_cast<const bf16x2*>(&ones), bsum[f], false);
                bsum[f] = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2*>(&v1), *reinterpret_cast<const bf16x2*>(&ones), bsum[f], false);
            }
        };
        constexpr int STEPS = TH / 4, NT = STEPS * NTAP, DIST = 3;    // this wave's taps of a tile; operand requests run DIST taps ahead
        const unsigned x0 = lds0 + (unsigned)(par * G::BUF + x_lane);
        auto tap_addr = [&](int T) {
            const int tap = tap0 + T % NTAP;                        // wave-uniform: scalar arithmetic
            return x0 + (unsigned)((T / NTAP) * (4 * XR) + (tap / 5) * XR + (tap % 5) * XP);
        };
        // the gradient fragments of all steps of the tile first (48 reads, 80 registers), then ONE software pipeline over the
        // wave's 52 taps of the tile: with one wave per SIMD nobody else covers the LDS latency (~100 cycles), so the four strips
        // of tap T + 3 are requested in front of the instructions of tap T - a ring of four operand buffers, everything
        // unrolled (static register numbers), the order pinned per tap
        unsigned a4[STEPS][NF][4], ix[STEPS][NF];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) readA(sG + z_lane + st * (2 * GR), a4[st], ix[st]);
        bf16x16 ring[DIST + 1];
#pragma unroll
        for (int d = 0; d < DIST; ++d) ring[d] = readB(tap_addr(d));
        // the pipeline description starts with the requests issued so far: 3 per fragment and step, 4 per tap
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * NF * STEPS + 4 * DIST, 0);
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            const int st = T / NTAP, t = T % NTAP;
            if (T + DIST < NT) ring[(T + DIST) % (DIST + 1)] = readB(tap_addr(T + DIST));
#pragma unroll
            for (int f = 0; f < NF; ++f)
                acc[f][t] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(*reinterpret_cast<const bf16x8*>(a4[st][f]), ring[T % (DIST + 1)],
                                                                      acc[f][t], (int)ix[st][f], 0, 0);
            if (T + DIST < NT) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);      // DS read x 4
            __builtin_amdgcn_sched_group_barrier(0x008, NF, 0);                         // MFMA x 4
        }
        if (more) commit(smem_raw + (par ^ 1) * G::BUF);
        __syncthreads();
    }
    if (do_bias) {                              // lanes g, g + 16, g + 32, g + 48 hold the four window groups of one channel
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            float v = bsum[f];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16) p.db_partial[(long)split * p.Cout + co0 + f * 16 + lane] = v;
        }
    }
    // D row m = 4 q + j = output channel, column = lane & 15 = input channel: 16-byte stores along the output channels; the
    // second tap half starts at tap 12, which the first one stores
    float* slab = p.partial + (long)split * 25 * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        if (th == 1 && t == 0) continue;
#pragma unroll
        for (int f = 0; f < NF; ++f)
            *reinterpret_cast<f32x4*>(slab + ((long)(tap0 + t) * p.Cin + ci0 + 16 * ai + g) * p.Cout + co0 + f * 16 + 4 * q) = acc[f][t];
    }
}

// Eight-wave form of conv5_wgrad_sparse_kernel: TWO waves per SIMD.  The four-wave form above leaves one wave per SIMD to issue a
// tile's ~1000 instructions - fetch, arg-max pieces, fragment reads, 208 matrix instructions - strictly one behind the other:
// 37 % of its cycles on the matrix pipe (profiles/r03_pmc_wgrad5_sparse_v3.json), 40 % of its time in the staging in front of the
// tap pipeline (profiles/r03_ab_wgrad5_sparse_ablation.txt).  Here a wave owns 16 input x 64 output channels x a QUARTER of the
// taps (7 | 6 | 6 | 6: 112 accumulator registers), keeps the gradient fragments of ONE step of four rows at a time (20 registers,
// the next step's requested under the current one's matrix instructions) and stages half as much per thread - everything fits
// the 256 registers two waves per SIMD leave each, and one wave's staging / LDS round trips run under the partner's matrix work.
// Waves w and w + 4 share a SIMD: tap quarters (0, 2) = 13 taps and (1, 3) = 12.  Same LDS layout, same tile loop, same slabs.
template <int TH>
__global__ __launch_bounds__(512, 1) void conv5_wgrad_sparse8_kernel(const Wg5Params p) {
    using G = Wg5SGeom<TH>;
    constexpr int XP = G::XP, XR = G::XR, GP = G::GP, GR = G::GR;
    constexpr int IP = (G::THH + 5) / 6, ZP = G::ZITEMS / 512;
    static_assert(G::ZITEMS % 512 == 0, "pooled tile divides over the threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ai = wave & 1, tg = wave >> 1;                       // this wave's 16 input channels, its quarter of the taps
    const int tap0 = tg == 0 ? 0 : 1 + 6 * tg;                     // taps 0..6 | 7..12 | 13..18 | 19..24
    const int cib = p.Cin / 32, cob = p.Cout / 64;
    int bid = xcd_order(blockIdx.x);
    const int ci0 = (bid % cib) * 32;
    bid /= cib;
    const int co0 = (bid % cob) * 64;
    const int split = bid / cob;
    const int q = lane >> 4, g = lane & 15;
    const int Hp = p.H >> 1, Wp = p.W >> 1;

    constexpr int NF = 4, NTAP = 7;
    f32x4 acc[NF][NTAP];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NTAP; ++t) acc[f][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[NF] = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.db_partial && ci0 == 0 && wave == 0;

    const int tiles = p.tiles_y * p.tiles_x;
    const int work_total = p.N * tiles;
    const int w_begin = split * p.work_per_split;
    const int w_end = min(work_total, w_begin + p.work_per_split);

    // ---- staging maps: 480 threads cover six halo rows of 20 pixels x four 16-byte pieces per pass; one pooled pixel piece each
    const int ihy0 = tid < 480 ? tid / 80 : 100000, ihx = (tid % 80) >> 2;
    const int ic8 = (tid & 3) * 8;
    const int icommit = ihx * XP + (tid & 3) * 16;
    const int zc8 = (tid & 7) * 8;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.in), 0, (int)((long)p.N * p.H * p.W * p.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.g), 0, (int)((long)p.N * Hp * Wp * p.Cout * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(p.idx), 0, (int)((long)p.N * Hp * Wp * p.Cout), 0x00020000);
    u32x4 preI[IP], preZ[ZP];
    u32x2 preK[ZP];
    auto fetch = [&](int wk) {
        const int n = wk / tiles, tile = wk - n * tiles;
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        const int iy0 = ty * TH - 2, ix0 = tx * 16 - 2;
        const int gx = ix0 + ihx;
        const bool okx = (unsigned)gx < (unsigned)p.W;
#pragma unroll
        for (int i = 0; i < IP; ++i) {
            const int gy = iy0 + ihy0 + 6 * i;
            const bool ok = okx & ((unsigned)gy < (unsigned)p.H) & (ihy0 + 6 * i < G::THH);
            const unsigned off = ok ? (unsigned)((((n * p.H + gy) * p.W + gx) * p.Cin + ci0 + ic8) * 2) : 0x80000000u;
            preI[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < ZP; ++i) {
            const int ppix = (tid + i * 512) >> 3;
            const unsigned e = (unsigned)(((n * Hp + ty * (TH / 2) + (ppix >> 3)) * Wp + tx * 8 + (ppix & 7)) * p.Cout + co0 + zc8);
            preZ[i] = __builtin_amdgcn_raw_buffer_load_b128(rz, e * 2, 0, 0);
            preK[i] = __builtin_amdgcn_raw_buffer_load_b64(rk, e, 0, 0);
        }
    };
    auto commit = [&](unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < IP; ++i)
            if (ihy0 + 6 * i < G::THH) *reinterpret_cast<u32x4*>(buf + (ihy0 + 6 * i) * XR + icommit) = preI[i];
        unsigned char* gb = buf + G::IBYTES;
#pragma unroll
        for (int i = 0; i < ZP; ++i) {
            const int ppix = (tid + i * 512) >> 3, pr = ppix >> 3, pc = ppix & 7;
            const int o = pr * GR + pc * GP + zc8 * 2;
            *reinterpret_cast<u32x4*>(gb + o) = preZ[i];
            const unsigned u = (unsigned)(pr & 1), s8 = 8u * (unsigned)((pc >> 1) & 1);     // the pieces of conv5_wgrad_sparse_kernel
            u32x4 m16, x16;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const unsigned e2 = __builtin_amdgcn_perm(0u, preK[i][d >> 1], (d & 1) ? 0x0c030c02u : 0x0c010c00u);
                m16[d] = (e2 & 0x00010001u) * 0xffffu;
                const unsigned val = ((e2 >> 1) & 0x00010001u) + u * 0x00020002u;
                x16[d] = ((val << (2 * u)) | (val << (2 * u + 4))) << s8;
            }
            *reinterpret_cast<u32x4*>(gb + G::GBYTES + o) = m16;
            *reinterpret_cast<u32x4*>(gb + 2 * G::GBYTES + o) = x16;
        }
    };
    const int jx = g >> 2;
    const int x_lane = jx * XR + 4 * q * XP + ai * 32 + (g & 3) * 8;
    const int pcA = (q & 1) * 4 + (q >> 1);
    const int z_lane = (jx & 1) * GR + (pcA + 2 * (jx >> 1)) * GP + (g & 3) * 8;
    auto tr2 = [](const unsigned char* a) {
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a);
        return *reinterpret_cast<const u32x2*>(&v);
    };
    auto trl = [](unsigned lds_addr) {
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(unsigned long)lds_addr);
        return *reinterpret_cast<const u32x2*>(&v);
    };

    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem_raw;
    if (w_begin < w_end) {
        fetch(w_begin);
        commit(smem_raw);
    }
    __syncthreads();
    for (int wk = w_begin; wk < w_end; ++wk) {
#ifndef NIMG_W8_ABL
#define NIMG_W8_ABL 0
#endif
        constexpr int ABL = NIMG_W8_ABL;        // diagnostic builds: 1 = stage the first tile only, 2 = no tap pipeline, 4 = one B operand
        const int par = (ABL & 1) ? 0 : (wk - w_begin) & 1;
        const bool more = wk + 1 < w_end && !(ABL & 1);
        if (more) fetch(wk + 1);
        const unsigned char* sG = smem_raw + par * G::BUF + G::IBYTES;
        auto readB = [&](unsigned xr) {
            asm volatile("" : "+v"(xr));                   // one opaque address per tap: see conv5_wgrad_sparse_kernel
            const u32x2 r0 = trl(xr), r1 = trl(xr + XP), r2 = trl(xr + 2 * XP), r3 = trl(xr + 3 * XP);
            const u32x4 q0 = __builtin_shufflevector(r0, r1, 0, 1, 2, 3), q1 = __builtin_shufflevector(r2, r3, 0, 1, 2, 3);
            const u32x8 b8 = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7);
            return *reinterpret_cast<const bf16x16*>(&b8);
        };
        auto readA = [&](const unsigned char* zr, unsigned (&a)[NF][4], unsigned (&ix)[NF]) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const u32x2 v = tr2(zr + f * 32), m = tr2(zr + G::GBYTES + f * 32), k = tr2(zr + 2 * G::GBYTES + f * 32);
                a[f][0] = v[0] & ~m[0]; a[f][1] = v[0] & m[0];
                a[f][2] = v[1] & ~m[1]; a[f][3] = v[1] & m[1];
                const unsigned t = k[0] | k[1];
                ix[f] = t | (t >> 16);
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                const unsigned ones = 0x3f803f80u, v0 = v[0], v1 = v[1];
                bsum[f] = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2*>(&v0), *reinterpret_cast<const bf16x2*>(&ones), bsum[f], false);
                bsum[f] = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2*>(&v1), *reinterpret_cast<const bf16x2*>(&ones), bsum[f], false);
            }
        };
#ifndef NIMG_W8_DIST
#define NIMG_W8_DIST 2
#endif
        constexpr int STEPS = TH / 4, NT = STEPS * NTAP, DIST = NIMG_W8_DIST;
        const unsigned x0 = lds0 + (unsigned)(par * G::BUF + x_lane);
        auto tap_addr = [&](int T) {
            const int tap = tap0 + T % NTAP;                        // wave-uniform: scalar arithmetic
            return x0 + (unsigned)((T / NTAP) * (4 * XR) + (tap / 5) * XR + (tap % 5) * XP);
        };
        const bool seven = tg == 0;                                 // the first quarter owns a seventh tap
        if constexpr ((ABL & 2) == 0) {
        unsigned a4[2][NF][4], ix[2][NF];
        readA(sG + z_lane, a4[0], ix[0]);
        bf16x16 ring[DIST + 1];
#pragma unroll
        for (int d = 0; d < DIST; ++d) ring[d] = readB(tap_addr(d));
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            const int st = T / NTAP, t = T % NTAP;
            if (t == 1 && st + 1 < STEPS) readA(sG + z_lane + (st + 1) * (2 * GR), a4[(st + 1) & 1], ix[(st + 1) & 1]);
            if (T + DIST < NT && !(ABL & 4)) {
                if ((T + DIST) % NTAP < 6 || seven) ring[(T + DIST) % (DIST + 1)] = readB(tap_addr(T + DIST));
            }
            if (t < 6 || seven) {
#pragma unroll
                for (int f = 0; f < NF; ++f)
                    acc[f][t] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(*reinterpret_cast<const bf16x8*>(a4[st & 1][f]),
                                                                          ring[(ABL & 4) ? 0 : T % (DIST + 1)], acc[f][t], (int)ix[st & 1][f], 0, 0);
            }
        }
        }
        if (more) commit(smem_raw + (par ^ 1) * G::BUF);
        __syncthreads();
    }
    if (do_bias) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            float v = bsum[f];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16) p.db_partial[(long)split * p.Cout + co0 + f * 16 + lane] = v;
        }
    }
    float* slab = p.partial + (long)split * 25 * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        if (t == 6 && tg != 0) continue;
#pragma unroll
        for (int f = 0; f < NF; ++f)
            *reinterpret_cast<f32x4*>(slab + ((long)(tap0 + t) * p.Cin + ci0 + 16 * ai + g) * p.Cout + co0 + f * 16 + 4 * q) = acc[f][t];
    }
}

template <int TH>
int launch_sparse(Wg5Params p, int max_slabs, hipStream_t stream) {
    using G = Wg5SGeom<TH>;
    p.tiles_y = p.H / TH;
    p.tiles_x = p.W / 16;
    const long work = (long)p.N * p.tiles_y * p.tiles_x;
    const long blocks_io = (long)(p.Cin / 32) * (p.Cout / 64);
    static const int target = getenv("NIMG_WGRAD5_ALLTAPS_BLOCKS") ? atoi(getenv("NIMG_WGRAD5_ALLTAPS_BLOCKS")) : 256;
    long splits = (target + blocks_io - 1) / blocks_io;
    if (splits > max_slabs) splits = max_slabs;
    if (splits > work) splits = work;
    if (splits < 1) return 0;
    const long wps = (work + splits - 1) / splits;
    splits = (work + wps - 1) / wps;
    p.work_per_split = (int)wps;
    // NIMG_WGRAD5_W4: the four-wave form (one wave per SIMD), A/B runs; read per call
    const bool w8 = getenv("NIMG_WGRAD5_W4") == nullptr;
    auto kern = w8 ? conv5_wgrad_sparse8_kernel<TH> : conv5_wgrad_sparse_kernel<TH>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    hipLaunchKernelGGL(kern, dim3((unsigned)(blocks_io * splits)), dim3(w8 ? 512 : 256), G::LDS, stream, p);
    if (hipGetLastError() != hipSuccess) return -1;
    return (int)splits;
}

}  // namespace

// Weight-gradient slabs (+ bias partials) of a 5x5 / stride 1 / SAME layer from its bf16 input and the POOLED bf16 gradient +
// arg-max bytes.  Returns the number of slabs written to partial[slab][25][cin][cout] (db_partial[slab][cout]), 0 when the
// shape is not this kernel's (the caller falls back to conv_wgrad_bf16_kernel), -1 on a launch error.
int nimg_internal_wgrad5_alltaps(const void* in, int cin, const void* g, const unsigned char* idx, int cout, float* partial,
                                 float* db_partial, int n, int h, int wd, int max_slabs, hipStream_t stream) {
    static const bool off = getenv("NIMG_NO_WGRAD5_ALLTAPS") != nullptr;
    if (off || (cin % 32) || (cout % 64) || (wd % 16) || (h % 8) || max_slabs < 1) return 0;
    const long in_bytes = (long)n * h * wd * cin * 2, g_bytes = (long)n * (h / 2) * (wd / 2) * cout * 2;
    if (in_bytes >= (1l << 31) - 65536 || g_bytes >= (1l << 31) - 65536) return 0;
    Wg5Params p;
    p.in = in; p.g = g; p.idx = idx; p.partial = partial; p.db_partial = db_partial;
    p.Cin = cin; p.Cout = cout; p.N = n; p.H = h; p.W = wd;
    p.tiles_y = p.tiles_x = p.work_per_split = 0;
    // NIMG_NO_WGRAD5_SPARSE is read per call (A/B in one process); h % 16: the sparse form is built for 16-row tiles
    if (!getenv("NIMG_NO_WGRAD5_SPARSE") && h % 16 == 0) return launch_sparse<16>(p, max_slabs, stream);
    static const bool th8 = getenv("NIMG_WGRAD5_TH8") != nullptr, kx3l = getenv("NIMG_WGRAD5_KX3L") != nullptr;
    static const int sched = getenv("NIMG_WGRAD5_SCHED") ? atoi(getenv("NIMG_WGRAD5_SCHED")) : 2;
    const int variant = ((th8 || (h % 16)) ? 4 : 0) | (kx3l ? 2 : 0) | (sched == 1 ? 1 : 0);
    switch (variant) {
        case 0: return launch<16, false, 2>(p, max_slabs, stream);
        case 1: return launch<16, false, 1>(p, max_slabs, stream);
        case 2: return launch<16, true, 2>(p, max_slabs, stream);
        case 3: return launch<16, true, 1>(p, max_slabs, stream);
        case 4: return launch<8, false, 2>(p, max_slabs, stream);
        case 5: return launch<8, false, 1>(p, max_slabs, stream);
        case 6: return launch<8, true, 2>(p, max_slabs, stream);
        default: return launch<8, true, 1>(p, max_slabs, stream);
    }
}
