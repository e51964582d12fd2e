/*
 * oracle/djpeg_ref.c - CPU restatement (plain C, float32) of the reference's differentiable JPEG
 * forward pass.  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load the library built from this file; the product path never does.
 *
 * Follows models/jpeg.py:91-159 (DifferentiableJPEG.call) with rounding mode 'soft'/'round'
 * (models/layers.py:122-128: forward = tf.round = round-half-to-even) and quantisation tables from
 * compression/jpeg_helpers.py:264-305 (passed in by the caller).
 *
 * This file pins the CANONICAL float32 EVALUATION ORDER that the HIP kernel
 * (neural-imaging_amd/csrc/djpeg.hip) reproduces bit-for-bit, so that the integer index tensor
 * rint(X/Q) - the bit-exact sub-contract of BASELINE.json - can be compared with == :
 *   colour  : acc = bias; acc = fmaf(255*r, c1, acc); fmaf(255*g, c2, acc); fmaf(255*b, c3, acc); minus 127
 *   DCT     : row pass T = b*F^T (fmaf chain over j ascending, acc0 = 0), column pass X = F*T (chain over i)
 *   quant   : u = X / Q (IEEE division), r = rintf(u), Xd = r * Q
 *   IDCT    : S = F^T*Xd (chain over u ascending), xi = S*F (chain over v ascending)
 *   colour^-1: q = xi + 127 ; acc = bias; fmaf(q0,c1); fmaf(q1,c2); fmaf(q2,c3); / 255 ; clamp [0,1]
 * TensorFlow's own summation order inside conv2d/matmul is unspecified (Eigen), so bit-exactness against
 * TF itself is not claimable; against float64 the indices differ only at rounding ties (tests measure it).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; fmaf() is the only fused operation)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static const float DCT_F[8][8] = {
    {0.3536f, 0.3536f, 0.3536f, 0.3536f, 0.3536f, 0.3536f, 0.3536f, 0.3536f},
    {0.4904f, 0.4157f, 0.2778f, 0.0975f, -0.0975f, -0.2778f, -0.4157f, -0.4904f},
    {0.4619f, 0.1913f, -0.1913f, -0.4619f, -0.4619f, -0.1913f, 0.1913f, 0.4619f},
    {0.4157f, -0.0975f, -0.4904f, -0.2778f, 0.2778f, 0.4904f, 0.0975f, -0.4157f},
    {0.3536f, -0.3536f, -0.3536f, 0.3536f, 0.3536f, -0.3536f, -0.3536f, 0.3536f},
    {0.2778f, -0.4904f, 0.0975f, 0.4157f, -0.4157f, -0.0975f, 0.4904f, -0.2778f},
    {0.1913f, -0.4619f, 0.4619f, -0.1913f, -0.1913f, 0.4619f, -0.4619f, 0.1913f},
    {0.0975f, -0.2778f, 0.4157f, -0.4904f, 0.4904f, -0.4157f, 0.2778f, -0.0975f}};

/* models/jpeg.py:74-75, bias in column 0; products like -1.402*128 are evaluated in double then cast (numpy) */
static const float COLOR_F[3][4] = {
    {0.0f, 0.299f, 0.587f, 0.114f},
    {128.0f, -0.168736f, -0.331264f, 0.5f},
    {128.0f, 0.5f, -0.418688f, -0.081312f}};
static const float COLOR_I[3][4] = {
    {(float)(-1.402 * 128), 1.0f, 0.0f, 1.402f},
    {(float)(1.058272 * 128), 1.0f, -0.344136f, -0.714136f},
    {(float)(-1.772 * 128), 1.0f, 1.772f, 0.0f}};

/* x,y: (n,h,w,3) float32 NHWC.  q: (3,8,8) float32 tables for Y,Cb,Cr.
 * idx (optional): (n,3,h/8,w/8,8,8) int16 quantisation indices rint(X/Q).
 * xd  (optional): same shape float32 dequantised coefficients (second output of the reference model).
 * returns 0, or -1 on bad shape. */
int djpeg_ref_forward(const float *x, float *y, const float *q, int16_t *idx, float *xd, int n, int h, int w)
{
    if (h % 8 || w % 8 || n < 0) return -1;
    const int hb = h / 8, wb = w / 8;
    for (int im = 0; im < n; ++im)
        for (int by = 0; by < hb; ++by)
            for (int bx = 0; bx < wb; ++bx) {
                float blk[3][8][8], out[3][8][8];
                for (int i = 0; i < 8; ++i)
                    for (int j = 0; j < 8; ++j) {
                        const float *px = x + (((size_t)im * h + by * 8 + i) * w + bx * 8 + j) * 3;
                        const float r = 255.0f * px[0], g = 255.0f * px[1], b = 255.0f * px[2];
                        for (int c = 0; c < 3; ++c) {
                            float acc = COLOR_F[c][0];
                            acc = fmaf(r, COLOR_F[c][1], acc);
                            acc = fmaf(g, COLOR_F[c][2], acc);
                            acc = fmaf(b, COLOR_F[c][3], acc);
                            blk[c][i][j] = acc - 127.0f;
                        }
                    }
                for (int c = 0; c < 3; ++c) {
                    float t[8][8], X[8][8], s[8][8];
                    for (int i = 0; i < 8; ++i)            /* row pass: T = b * F^T */
                        for (int v = 0; v < 8; ++v) {
                            float acc = 0.0f;
                            for (int j = 0; j < 8; ++j) acc = fmaf(blk[c][i][j], DCT_F[v][j], acc);
                            t[i][v] = acc;
                        }
                    for (int u = 0; u < 8; ++u)            /* column pass: X = F * T */
                        for (int v = 0; v < 8; ++v) {
                            float acc = 0.0f;
                            for (int i = 0; i < 8; ++i) acc = fmaf(t[i][v], DCT_F[u][i], acc);
                            const float qq = q[c * 64 + u * 8 + v];
                            const float r = rintf(acc / qq);       /* half-to-even in the default rounding mode */
                            X[u][v] = r * qq;
                            const size_t o = ((((size_t)im * 3 + c) * hb + by) * wb + bx) * 64 + u * 8 + v;
                            if (idx) idx[o] = (int16_t)r;
                            if (xd) xd[o] = X[u][v];
                        }
                    for (int i = 0; i < 8; ++i)
                        for (int v = 0; v < 8; ++v) {
                            float acc = 0.0f;
                            for (int u = 0; u < 8; ++u) acc = fmaf(DCT_F[u][i], X[u][v], acc);
                            s[i][v] = acc;
                        }
                    for (int i = 0; i < 8; ++i)
                        for (int j = 0; j < 8; ++j) {
                            float acc = 0.0f;
                            for (int v = 0; v < 8; ++v) acc = fmaf(s[i][v], DCT_F[v][j], acc);
                            out[c][i][j] = acc + 127.0f;
                        }
                }
                for (int i = 0; i < 8; ++i)
                    for (int j = 0; j < 8; ++j) {
                        float *py = y + (((size_t)im * h + by * 8 + i) * w + bx * 8 + j) * 3;
                        for (int c = 0; c < 3; ++c) {
                            float acc = COLOR_I[c][0];
                            acc = fmaf(out[0][i][j], COLOR_I[c][1], acc);
                            acc = fmaf(out[1][i][j], COLOR_I[c][2], acc);
                            acc = fmaf(out[2][i][j], COLOR_I[c][3], acc);
                            acc = acc / 255.0f;
                            py[c] = acc < 0.0f ? 0.0f : (acc > 1.0f ? 1.0f : acc);
                        }
                    }
            }
    return 0;
}

/* IJG quality scaling, compression/jpeg_helpers.py:264-305 (float32 arithmetic like the numpy reference). */
int djpeg_ref_qtable(int quality, int channel, float *out64)
{
    static const float luma[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55,
                                   14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                   18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                                   49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
    static const float chroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                                     24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
    if (quality > 100) quality = 100;
    if (quality < 1) quality = 1;
    /* python: 5000 / q (true division, float64) or 200 - 2q (int); the product with float32 t promotes to ... */
    const double s = quality < 50 ? 5000.0 / quality : 200.0 - 2.0 * quality;
    const float *t = channel == 0 ? luma : chroma;
    for (int k = 0; k < 64; ++k) {
        /* numpy: float32 array * python scalar -> float32 (scalar is weakly typed) */
        float v = floorf((t[k] * (float)s + 50.0f) / 100.0f);
        if (v < 1.0f) v = 1.0f;
        if (v > 255.0f) v = 255.0f;
        out64[k] = v;
    }
    return 0;
}
