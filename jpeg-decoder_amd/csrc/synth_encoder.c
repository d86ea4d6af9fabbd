/*
 * synth_encoder.c -- deterministic baseline-DCT JPEG *encoder* used to manufacture the
 * synthetic workloads BASELINE.json names (1080p/4K 4:2:0, 4:4:4, optional DRI).  It is a
 * workload generator for tests and bench.py, not part of the decode path: the GPU box has no
 * network and no guaranteed Pillow, and the reference ships no encoder.
 *
 * Baseline sequential, 8-bit, 3 components (ids 1,2,3), Annex-K quantisation tables scaled the
 * libjpeg way, Annex-K (non-optimised) Huffman tables, chroma 1x1 (4:4:4) or 2x2 (4:2:0) with
 * box-filter down-sampling, optional restart interval (DRI + RST0..7 cycling).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- Annex K tables (ITU T.81 K.1, K.3) ------------------------------------------------ */

static const uint8_t K_LUMA_Q[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t K_CHROMA_Q[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

static const uint8_t DC_L_BITS[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t DC_C_BITS[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t DC_VALS[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t AC_L_BITS[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t AC_L_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07,
    0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0,
    0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49,
    0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
    0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5,
    0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa};
static const uint8_t AC_C_BITS[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t AC_C_VALS[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71,
    0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0,
    0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
    0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa};

static uint8_t ZZ[64]; /* ZZ[k] = natural index (row*8+col) of zigzag position k */

static void init_zz(void) {
    if (ZZ[63]) return;
    int r = 0, c = 0;
    for (int k = 0; k < 64; ++k) {
        ZZ[k] = (uint8_t)(r * 8 + c);
        if ((r + c) % 2 == 0) { /* moving up-right */
            if (c == 7) ++r; else if (r == 0) ++c; else { --r; ++c; }
        } else {
            if (r == 7) ++c; else if (c == 0) ++r; else { ++r; --c; }
        }
    }
}

/* ---- output buffer / bit writer ---------------------------------------------------------- */

typedef struct {
    uint8_t *p;
    size_t n, cap;
    uint32_t acc;
    int nacc;
    int overflow;
} obuf;

static void put8(obuf *o, unsigned v) {
    if (o->n < o->cap) o->p[o->n] = (uint8_t)v; else o->overflow = 1;
    o->n++;
}
static void put16(obuf *o, unsigned v) { put8(o, v >> 8); put8(o, v & 255); }

static void putbits(obuf *o, unsigned code, int len) {
    for (int i = len - 1; i >= 0; --i) {
        o->acc = (o->acc << 1) | ((code >> i) & 1u);
        if (++o->nacc == 8) {
            put8(o, o->acc & 255);
            if ((o->acc & 255) == 0xFF) put8(o, 0); /* byte stuffing */
            o->acc = 0; o->nacc = 0;
        }
    }
}
static void flushbits(obuf *o) { /* pad with 1-bits to the byte boundary */
    while (o->nacc) putbits(o, 1, 1);
}

typedef struct { uint16_t code[256]; uint8_t len[256]; } enctab;

static void make_enctab(enctab *t, const uint8_t bits[16], const uint8_t *vals) {
    memset(t, 0, sizeof *t);
    unsigned code = 0; int k = 0;
    for (int L = 1; L <= 16; ++L) {
        for (int i = 0; i < bits[L - 1]; ++i) { t->code[vals[k]] = (uint16_t)code++; t->len[vals[k]] = (uint8_t)L; ++k; }
        code <<= 1;
    }
}

static void put_dht(obuf *o, int tc_th, const uint8_t bits[16], const uint8_t *vals) {
    int n = 0; for (int i = 0; i < 16; ++i) n += bits[i];
    put16(o, 0xFFC4); put16(o, 2 + 1 + 16 + n); put8(o, tc_th);
    for (int i = 0; i < 16; ++i) put8(o, bits[i]);
    for (int i = 0; i < n; ++i) put8(o, vals[i]);
}

static int bitsize(int v) { int a = v < 0 ? -v : v, n = 0; while (a) { ++n; a >>= 1; } return n; }

static void encode_block(obuf *o, const int q[64] /* zigzag order */, int *pred, const enctab *dc, const enctab *ac) {
    int diff = q[0] - *pred; *pred = q[0];
    int s = bitsize(diff);
    putbits(o, dc->code[s], dc->len[s]);
    if (s) putbits(o, (unsigned)(diff < 0 ? diff - 1 : diff) & ((1u << s) - 1), s);
    int run = 0;
    for (int k = 1; k < 64; ++k) {
        if (q[k] == 0) { ++run; continue; }
        while (run > 15) { putbits(o, ac->code[0xF0], ac->len[0xF0]); run -= 16; }
        s = bitsize(q[k]);
        int sym = (run << 4) | s;
        putbits(o, ac->code[sym], ac->len[sym]);
        putbits(o, (unsigned)(q[k] < 0 ? q[k] - 1 : q[k]) & ((1u << s) - 1), s);
        run = 0;
    }
    if (run) putbits(o, ac->code[0], ac->len[0]);
}

static double CT[8][8]; /* forward DCT basis */
static void init_ct(void) {
    if (CT[0][0] != 0) return;
    for (int u = 0; u < 8; ++u)
        for (int x = 0; x < 8; ++x)
            CT[u][x] = (u == 0 ? sqrt(0.125) : 0.5) * cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
}

static void fdct_quant(const double s[64], const uint8_t qnat[64], int qz[64]) {
    double t[64], f[64];
    for (int r = 0; r < 8; ++r)
        for (int u = 0; u < 8; ++u) { double a = 0; for (int x = 0; x < 8; ++x) a += s[r * 8 + x] * CT[u][x]; t[r * 8 + u] = a; }
    for (int v = 0; v < 8; ++v)
        for (int u = 0; u < 8; ++u) { double a = 0; for (int y = 0; y < 8; ++y) a += t[y * 8 + u] * CT[v][y]; f[v * 8 + u] = a; }
    for (int k = 0; k < 64; ++k) { int n = ZZ[k]; qz[k] = (int)lround(f[n] / qnat[n]); }
}

static void scale_q(const uint8_t base[64], int quality, uint8_t out[64]) {
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (int i = 0; i < 64; ++i) { int v = (base[i] * scale + 50) / 100; out[i] = (uint8_t)(v < 1 ? 1 : v > 255 ? 255 : v); }
}

/*
 * Encode interleaved RGB (w*h*3).  subsamp: 0 = 4:4:4, 1 = 4:2:0, 2 = 4:2:2 (luma 2x1),
 * 3 = 4:4:0 (luma 1x2), 4 = grayscale (one component, id 1, luma only), 5 = 4:1:1 (luma 4x1),
 * 6 = 4:1:0 (luma 4x2), 7 = luma 1x4, 8 = luma 2x4; 0x100 | H << 4 | V = luma H x V (1..4 each), chroma
 * 1x1.  restart_mcus: 0 = no DRI.
 * Returns the number of bytes the stream needs; if it exceeds cap the buffer content is
 * undefined (call again with a bigger buffer).
 */
size_t jsynth_encode_rgb(const uint8_t *rgb, int w, int h, int quality, int subsamp,
                         int restart_mcus, uint8_t *out, size_t cap) {
    init_zz(); init_ct();
    obuf o = {out, 0, cap, 0, 0, 0};
    uint8_t ql[64], qc[64];
    scale_q(K_LUMA_Q, quality, ql); scale_q(K_CHROMA_Q, quality, qc);
    int ssh = (subsamp == 1 || subsamp == 2 || subsamp == 8) ? 2 : (subsamp == 5 || subsamp == 6) ? 4 : 1;
    int ssv = (subsamp == 1 || subsamp == 3 || subsamp == 6) ? 2 : (subsamp == 7 || subsamp == 8) ? 4 : 1;
    if (subsamp & 0x100) { ssh = (subsamp >> 4) & 15; ssv = subsamp & 15; }
    if (ssh < 1 || ssh > 4 || ssv < 1 || ssv > 4) return 0;
    int mcuw = 8 * ssh, mcuh = 8 * ssv;
    int mx = (w + mcuw - 1) / mcuw, my = (h + mcuh - 1) / mcuh;
    int pw = mx * mcuw, ph = my * mcuh;

    /* planes, edge-replicated to the padded size */
    float *Y = malloc(sizeof(float) * (size_t)pw * ph), *Cb = malloc(sizeof(float) * (size_t)pw * ph),
          *Cr = malloc(sizeof(float) * (size_t)pw * ph);
    for (int y = 0; y < ph; ++y) {
        int sy = y < h ? y : h - 1;
        for (int x = 0; x < pw; ++x) {
            int sx = x < w ? x : w - 1;
            const uint8_t *p = rgb + ((size_t)sy * w + sx) * 3;
            double R = p[0], G = p[1], B = p[2];
            size_t i = (size_t)y * pw + x;
            Y[i] = (float)(0.299 * R + 0.587 * G + 0.114 * B);
            Cb[i] = (float)(-0.168736 * R - 0.331264 * G + 0.5 * B + 128.0);
            Cr[i] = (float)(0.5 * R - 0.418688 * G - 0.081312 * B + 128.0);
        }
    }

    put16(&o, 0xFFD8);
    /* DQT: two tables in one segment, zigzag order */
    put16(&o, 0xFFDB); put16(&o, 2 + 65 * 2);
    put8(&o, 0); for (int k = 0; k < 64; ++k) put8(&o, ql[ZZ[k]]);
    put8(&o, 1); for (int k = 0; k < 64; ++k) put8(&o, qc[ZZ[k]]);
    /* SOF0 */
    const int gray = subsamp == 4, nc = gray ? 1 : 3;
    put16(&o, 0xFFC0); put16(&o, 8 + 3 * nc); put8(&o, 8); put16(&o, (unsigned)h); put16(&o, (unsigned)w); put8(&o, nc);
    put8(&o, 1); put8(&o, (ssh << 4) | ssv); put8(&o, 0);
    if (!gray) {
        put8(&o, 2); put8(&o, 0x11); put8(&o, 1);
        put8(&o, 3); put8(&o, 0x11); put8(&o, 1);
    }
    put_dht(&o, 0x00, DC_L_BITS, DC_VALS); put_dht(&o, 0x10, AC_L_BITS, AC_L_VALS);
    put_dht(&o, 0x01, DC_C_BITS, DC_VALS); put_dht(&o, 0x11, AC_C_BITS, AC_C_VALS);
    if (restart_mcus > 0) { put16(&o, 0xFFDD); put16(&o, 4); put16(&o, (unsigned)restart_mcus); }
    /* SOS */
    put16(&o, 0xFFDA); put16(&o, 6 + 2 * nc); put8(&o, nc);
    put8(&o, 1); put8(&o, 0x00);
    if (!gray) { put8(&o, 2); put8(&o, 0x11); put8(&o, 3); put8(&o, 0x11); }
    put8(&o, 0); put8(&o, 63); put8(&o, 0);

    enctab dcl, acl, dcc, acc_;
    make_enctab(&dcl, DC_L_BITS, DC_VALS); make_enctab(&acl, AC_L_BITS, AC_L_VALS);
    make_enctab(&dcc, DC_C_BITS, DC_VALS); make_enctab(&acc_, AC_C_BITS, AC_C_VALS);

    int pred[3] = {0, 0, 0}, rst = 0;
    long nm = 0;
    double s[64]; int qz[64];
    for (int j = 0; j < my; ++j)
        for (int i = 0; i < mx; ++i, ++nm) {
            if (restart_mcus > 0 && nm && nm % restart_mcus == 0) {
                flushbits(&o);
                put16(&o, 0xFFD0 + (rst & 7)); ++rst;
                pred[0] = pred[1] = pred[2] = 0;
            }
            for (int v = 0; v < ssv; ++v)
                for (int hh = 0; hh < ssh; ++hh) {
                    int bx = i * mcuw + 8 * hh, by = j * mcuh + 8 * v;
                    for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) s[y * 8 + x] = Y[(size_t)(by + y) * pw + bx + x] - 128.0;
                    fdct_quant(s, ql, qz);
                    encode_block(&o, qz, &pred[0], &dcl, &acl);
                }
            for (int c = 1; c < nc; ++c) {
                const float *P = c == 1 ? Cb : Cr;
                for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) {
                    double a = 0;
                    for (int dy = 0; dy < ssv; ++dy) for (int dx = 0; dx < ssh; ++dx)
                        a += P[(size_t)(j * mcuh + y * ssv + dy) * pw + i * mcuw + x * ssh + dx];
                    s[y * 8 + x] = a / (ssh * ssv) - 128.0;
                }
                fdct_quant(s, qc, qz);
                encode_block(&o, qz, &pred[c], &dcc, &acc_);
            }
        }
    flushbits(&o);
    put16(&o, 0xFFD9);
    free(Y); free(Cb); free(Cr);
    return o.n;
}

/* ---- deterministic test pattern ------------------------------------
 * per channel c: 128 + 60 sin(x/(37+11c)+c) + 50 cos(y/(53-7c)) + 20 sin((x+y)/7) + N(0,sigma),
 * clipped to u8.  seed changes the noise and shifts the phases so images of a batch differ. */

static uint64_t xs(uint64_t *s) {
    uint64_t x = *s; x ^= x >> 12; x ^= x << 25; x ^= x >> 27; *s = x;
    return x * 0x2545F4914F6CDD1DULL;
}

void jsynth_pattern(uint8_t *rgb, int w, int h, uint64_t seed, double sigma) {
    uint64_t st = seed * 0x9E3779B97F4A7C15ULL + 0x1234567ULL;
    if (!st) st = 1;
    for (int i = 0; i < 8; ++i) xs(&st);
    double ph = (double)(seed % 1000) * 0.37;
    double *fx = malloc(sizeof(double) * 3 * (size_t)w), *fy = malloc(sizeof(double) * 3 * (size_t)h),
           *fd = malloc(sizeof(double) * (size_t)(w + h));
    for (int c = 0; c < 3; ++c) {
        for (int x = 0; x < w; ++x) fx[c * (size_t)w + x] = 60.0 * sin(x / (37.0 + 11 * c) + c + ph);
        for (int y = 0; y < h; ++y) fy[c * (size_t)h + y] = 50.0 * cos(y / (53.0 - 7 * c) + 0.5 * ph);
    }
    for (int d = 0; d < w + h; ++d) fd[d] = 20.0 * sin(d / 7.0);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; c += 1) {
                /* Box-Muller, one normal per sample (the second is discarded: simplicity) */
                double u1 = ((xs(&st) >> 11) + 1.0) * (1.0 / 9007199254740993.0);
                double u2 = (xs(&st) >> 11) * (1.0 / 9007199254740992.0);
                double n = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2) * sigma;
                double v = 128.0 + fx[c * (size_t)w + x] + fy[c * (size_t)h + y] + fd[x + y] + n;
                long r = lround(v);
                rgb[((size_t)y * w + x) * 3 + c] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
            }
    free(fx); free(fy); free(fd);
}

/* pattern + encode in one call; returns bytes needed */
size_t jsynth_make(int w, int h, uint64_t seed, int quality, int subsamp, int restart_mcus,
                   uint8_t *out, size_t cap) {
    uint8_t *rgb = malloc((size_t)w * h * 3);
    if (!rgb) return 0;
    jsynth_pattern(rgb, w, h, seed, 6.0);
    size_t n = jsynth_encode_rgb(rgb, w, h, quality, subsamp, restart_mcus, out, cap);
    free(rgb);
    return n;
}
