/*
 * oracle/jpeg_oracle.c -- CPU restatement of the reference decode path.  See jpeg_oracle.h:
 * TEST INFRASTRUCTURE ONLY, never linked or loaded by the product.
 *
 * Build with -O3 -std=c99 -ffp-contract=off -lm (the reference's own optimisation level,
 * Makefile:6; contraction off so x86 and any other host agree bit-for-bit).
 */
#include "jpeg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * constant tables
 * ---------------------------------------------------------------------------------------- */

const unsigned ojpeg_zigzag[8][8] = {
    /* zigzag index of natural (row, col): src/decoding.c:9-18.  Generated once by walking the
     * anti-diagonals and frozen here; oracle_selftest() re-derives it. */
    { 0,  1,  5,  6, 14, 15, 27, 28}, { 2,  4,  7, 13, 16, 26, 29, 42},
    { 3,  8, 12, 17, 25, 30, 41, 43}, { 9, 11, 18, 24, 31, 40, 44, 53},
    {10, 19, 23, 32, 39, 45, 52, 54}, {20, 22, 33, 38, 46, 51, 55, 60},
    {21, 34, 37, 47, 50, 56, 59, 61}, {35, 36, 48, 49, 57, 58, 62, 63}};

/* src/decoding.c:20-29: C(u) cos((2x+1) u pi / 16) rounded to SIX decimals.  Rebuilt from the
 * definition; k/1e6 is the same double a 6-decimal literal parses to (both are the correctly
 * rounded value of the same rational). */
static double T[8][8];
static int tables_ready;

static void build_tables(void) {
    if (tables_ready) return;
    const double pi = 3.14159265358979323846;
    for (int u = 0; u < 8; ++u)
        for (int x = 0; x < 8; ++x) {
            double c = (u == 0 ? sqrt(0.5) : 1.0) * cos((2 * x + 1) * u * pi / 16.0);
            T[u][x] = (double)lround(c * 1e6) / 1e6;
        }
    tables_ready = 1;
}

/* ------------------------------------------------------------------------------------------
 * header parsing: src/jpeg.c:23-58, src/parsing.c
 * ---------------------------------------------------------------------------------------- */

static int be16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

/* src/parsing.c:13-59 */
static int parse_sof0(ojpeg_info *info, unsigned len, const uint8_t *d) {
    if (len < 6) return -1;
    unsigned nf = d[5];
    if (len != 6 + 3 * nf) return -1;
    info->precision = d[0];
    if (info->precision != 8) return -1;
    info->height = (uint16_t)be16(d + 1);
    if (info->height < 1) return -1;
    info->width = (uint16_t)be16(d + 3);
    if (info->width < 1) return -1;
    for (unsigned i = 0; i < nf; ++i) {
        unsigned id = d[6 + 3 * i], H = d[7 + 3 * i] >> 4, V = d[7 + 3 * i] & 15, tq = d[8 + 3 * i];
        if (H == 0 || H > 4 || V == 0 || V > 4 || tq >= 4) return -1;
        info->comp[id].H = (uint8_t)H;
        info->comp[id].V = (uint8_t)V;
        info->comp[id].Tq = (uint8_t)tq;
    }
    return 0;
}

/* src/parsing.c:61-78 + src/hufftree.c:43-62 (bytes consumed = 16 + sum counts) */
static int parse_dht(ojpeg_info *info, unsigned len, const uint8_t *d) {
    if (len < 1) return -1;
    size_t off = 0;
    do {
        unsigned cls = d[off] >> 4, id = d[off] & 15;
        ++off;
        if (cls >= 2 || id >= 4) return -1;
        ojpeg_hufftable *t = &info->huff[cls][id];
        memset(t, 0, sizeof *t);
        int n = 0;
        for (int l = 0; l < 16; ++l) { t->counts[l] = d[off + l]; n += d[off + l]; }
        /* the reference reads symbols without checking the segment length (hufftree.c:53);
         * the 64 KiB stack buffer makes that benign.  n <= 16*255 there; clip to our array. */
        for (int s = 0; s < n && s < 256; ++s) t->symbols[s] = d[off + 16 + s];
        t->nsymbols = n;
        t->defined = 1;
        off += 16 + (size_t)n;
    } while (off < len);
    return 0;
}

/* src/parsing.c:80-103 */
static int parse_dqt(ojpeg_info *info, unsigned len, const uint8_t *d) {
    if (!len || len % 65 != 0) return -1;
    size_t off = 0;
    do {
        unsigned prec = d[off] >> 4, id = d[off] & 15;
        ++off;
        if (prec >= 1 || id >= 4) return -1;
        memcpy(info->quant[id], d + off, 64);
        off += 64;
    } while (len -= 65);
    return 0;
}

/* src/parsing.c:109-142 */
static int parse_sos(ojpeg_info *info, unsigned len, const uint8_t *d) {
    unsigned ns = d[0];
    if (len != 1 + 2 * ns + 3) return -1;
    info->ncomp = (uint8_t)ns;
    unsigned mh = 0, mv = 0;
    for (unsigned i = 0; i < ns; ++i) {
        unsigned id = d[1 + 2 * i], td = d[2 + 2 * i] >> 4, ta = d[2 + 2 * i] & 15;
        if (td >= 4 || ta >= 4) return -1;
        if (i < 4) { info->scomp[i].id = (uint8_t)id; info->scomp[i].Td = (uint8_t)td; info->scomp[i].Ta = (uint8_t)ta; }
        if (info->comp[id].H > mh) mh = info->comp[id].H;
        if (info->comp[id].V > mv) mv = info->comp[id].V;
    }
    info->maxH = (uint8_t)mh;
    info->maxV = (uint8_t)mv;
    return 0;
}

int ojpeg_parse(const uint8_t *f, size_t len, ojpeg_info *info) {
    memset(info, 0, sizeof *info);
    size_t p = 0;
    if (len < 2 || be16(f) != 0xFFD8) return OJPEG_ERR_HEADER; /* src/jpeg.c:25 */
    p = 2;
    unsigned marker;
    do { /* src/jpeg.c:32-55 */
        if (p + 2 > len) return OJPEG_ERR_HEADER;
        marker = (unsigned)be16(f + p);
        p += 2;
        /* EXTENSION (like DRI below; the reference has neither): 0xFF fill bytes in front of a
         * marker and the length-less markers TEM / RSTn (T.81 B.1.1.2, B.1.1.3).  The reference
         * takes the two bytes after ANY marker word as a length (src/jpeg.c:32-37); only the
         * marker words 0xFFFF, 0xFF01 and 0xFFD0..0xFFD7 are treated differently here. */
        while (marker == 0xFFFF) {
            if (p >= len) return OJPEG_ERR_HEADER;
            marker = 0xFF00u | f[p++];
        }
        if (marker == 0xFF01 || (marker >= 0xFFD0 && marker <= 0xFFD7)) continue;
        if (p + 2 > len) return OJPEG_ERR_HEADER;
        unsigned seglen = (unsigned)be16(f + p);
        p += 2;
        if (seglen < 2) return OJPEG_ERR_HEADER;
        unsigned dl = seglen - 2;
        if (p + dl > len) return OJPEG_ERR_HEADER; /* short fread, src/jpeg.c:41 */
        const uint8_t *d = f + p;
        int e = 0;
        switch (marker) {
            case 0xFFC0: e = parse_sof0(info, dl, d); break;
            case 0xFFC4: e = parse_dht(info, dl, d); break;
            case 0xFFDB: e = parse_dqt(info, dl, d); break;
            case 0xFFDA: e = parse_sos(info, dl, d); break;
            case 0xFFDD: if (dl >= 2) info->restart_interval = (uint32_t)be16(d); break; /* extension */
            default: break; /* src/jpeg.c:50 */
        }
        if (e) return OJPEG_ERR_HEADER;
        p += dl;
    } while (marker != 0xFFDA);
    info->scan_offset = p;
    return OJPEG_OK;
}

/* ------------------------------------------------------------------------------------------
 * geometry: src/jpeg.c:66-87
 * ---------------------------------------------------------------------------------------- */

static unsigned blocks_per_mcu(const ojpeg_info *info) {
    unsigned n = 0;
    for (unsigned i = 0; i < info->ncomp; ++i)
        n += info->comp[info->scomp[i].id].H * info->comp[info->scomp[i].id].V;
    return n;
}

static size_t mcu_count(const ojpeg_info *info) {
    unsigned mw = 8u * info->maxH, mh = 8u * info->maxV;
    if (!mw || !mh) return 0;
    return (size_t)((info->width + mw - 1) / mw) * ((info->height + mh - 1) / mh);
}

size_t ojpeg_block_count(const ojpeg_info *info) { return mcu_count(info) * blocks_per_mcu(info); }

/* ------------------------------------------------------------------------------------------
 * bit reader: src/bitstream.c
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    const uint8_t *d;
    size_t n, pos;
    unsigned left;
    uint8_t c;
    int err;
} bitrd;

/* src/bitstream.c:7-18: drop the byte after a 0xFF data byte, whatever it is */
static void load_byte(bitrd *s) {
    s->left = 8;
    if (s->c == 0xFF) s->pos++;
    if (s->pos >= s->n) { s->err = OJPEG_ERR_EOF; s->c = 0; s->pos = s->n; return; }
    s->c = s->d[s->pos++];
}

/* src/bitstream.c:28-33 */
static unsigned get_bit(bitrd *s) {
    if (s->left == 0) load_byte(s);
    return (s->c >> --s->left) & 1u;
}

/* src/bitstream.c:37-45 */
static uint32_t get_bits(bitrd *s, unsigned count) {
    if (count > 32) { s->err = OJPEG_ERR_BITS; return 0; }
    uint32_t v = 0;
    for (unsigned i = 0; i < count; ++i) v = (v << 1) | get_bit(s);
    return v;
}

/* ------------------------------------------------------------------------------------------
 * Huffman: canonical assignment of src/hufftree.c:46-59 into an index-linked binary tree,
 * bitwise walk of :64-78.  The walk returns at the first node that carries a symbol; a
 * missing child is the reference's FAIL.
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    int child[2]; /* 0 = absent (hasLeft/hasRight, src/hufftree.h:15-22); index 0 is the root */
    int has_sym;
    uint8_t sym;
} onode;

typedef struct {
    onode *nodes;
    int n, cap;
} otree;

static int tree_new_node(otree *t) {
    if (t->n == t->cap) {
        t->cap = t->cap ? 2 * t->cap : 64;
        t->nodes = realloc(t->nodes, (size_t)t->cap * sizeof *t->nodes);
    }
    memset(&t->nodes[t->n], 0, sizeof *t->nodes);
    return t->n++;
}

/* src/hufftree.c:43-62 (code assignment) + :18-41 (insertion, 1 = right, 0 = left) */
static void tree_build(otree *t, const ojpeg_hufftable *h) {
    t->nodes = NULL; t->n = t->cap = 0;
    tree_new_node(t);
    int code_value = 0, k = 0;
    for (int L = 1; L <= 16; ++L) {
        for (int i = 0; i < h->counts[L - 1]; ++i) {
            unsigned short code = (unsigned short)code_value++;
            uint8_t sym = h->symbols[k < 256 ? k : 255];
            ++k;
            int cur = 0;
            for (int bit = L; bit > 0; --bit) {
                int dir = (code >> (bit - 1)) & 1;
                if (!t->nodes[cur].child[dir]) {
                    int nn = tree_new_node(t);
                    t->nodes[cur].child[dir] = nn;
                }
                cur = t->nodes[cur].child[dir];
            }
            t->nodes[cur].sym = sym;
            t->nodes[cur].has_sym = 1;
        }
        code_value <<= 1;
    }
}

static void tree_free(otree *t) { free(t->nodes); t->nodes = NULL; }

/* src/hufftree.c:64-78 */
static int decode_symbol(const otree *t, bitrd *s) {
    int cur = 0;
    do {
        unsigned b = get_bit(s);
        if (s->err) return -1;
        int nx = t->nodes[cur].child[b];
        if (!nx) { s->err = OJPEG_ERR_NOSYMBOL; return -1; }
        cur = nx;
    } while (!t->nodes[cur].has_sym);
    return t->nodes[cur].sym;
}

/* src/decoding.c:31-33.  size==0 is undefined behaviour there (shift by UINT_MAX); gcc/x86-64
 * yields 0, which is also what the JPEG EXTEND procedure defines. */
static int extend(unsigned size, uint32_t bits) {
    if (size == 0) return 0;
    return (bits >> (size - 1)) ? (int)bits : (int)((long)bits - (1L << size) + 1);
}

/* coefficient positions past 63 met since the library was loaded (test diagnostics only) */
unsigned long ojpeg_overruns = 0;

/* src/decoding.c:35-66 on quantised values (the multiply by quantTable is applied later) */

static int decode_block(int32_t q[64], const otree *dc, const otree *ac, bitrd *s, int *pred) {
    memset(q, 0, 64 * sizeof *q);
    int size = decode_symbol(dc, s);
    if (size < 0) return s->err;
    uint32_t bits = get_bits(s, (unsigned)size);
    if (s->err) return s->err;
    *pred += extend((unsigned)size, bits);
    q[0] = *pred;
    for (unsigned i = 1; i < 64; ++i) {
        int v = decode_symbol(ac, s);
        if (v < 0) return s->err;
        unsigned run = (unsigned)v >> 4, sz = (unsigned)v & 15;
        i += run;
        if (sz == 0) {
            if (run == 15) continue;
            break;
        }
        bits = get_bits(s, sz);
        if (s->err) return s->err;
        if (i < 64) q[i] = extend(sz, bits);
        else ++ojpeg_overruns; /* the reference stores past coeffVector here (decoding.c:65, no bound
                                  check): undefined behaviour, in practice a stack-protector abort */
    }
    return 0;
}

/*
 * Restart extension.  With honour_dri and restart_interval != 0 the scan is cut at every
 * RSTn marker (an 0xFF that is itself a data byte position -- i.e. not the dropped follower
 * of another 0xFF -- followed by 0xD0..0xD7).  Each interval is decoded with a fresh bit
 * reader and zeroed predictors (ITU T.81 F.2.1.3.1).  The reference cannot do this
 * (src/jpeg.c:50, src/jpeg.c:73): pinned by DRI/non-DRI coefficient identity instead.
 */
static size_t next_rst(const uint8_t *d, size_t n, size_t from) {
    size_t i = from;
    while (i + 1 < n) {
        if (d[i] == 0xFF) {
            if (d[i + 1] >= 0xD0 && d[i + 1] <= 0xD7) return i;
            i += 2; /* FF xx: xx is dropped (stuffing) */
        } else
            ++i;
    }
    return n;
}

int ojpeg_decode_qcoef(const ojpeg_info *info, const uint8_t *scan, size_t scan_len,
                       int32_t *qcoef, int honour_dri) {
    otree dc[4], ac[4];
    unsigned nc = info->ncomp < 4 ? info->ncomp : 4;
    for (unsigned i = 0; i < nc; ++i) {
        tree_build(&dc[i], &info->huff[0][info->scomp[i].Td]);
        tree_build(&ac[i], &info->huff[1][info->scomp[i].Ta]);
    }
    int rc = OJPEG_OK;
    size_t nmcu = mcu_count(info);
    uint32_t ri = (honour_dri && info->restart_interval) ? info->restart_interval : 0;
    int pred[4] = {0, 0, 0, 0}; /* src/jpeg.c:73 */
    bitrd s = {scan, scan_len, 0, 0, 0, 0};
    size_t seg_end = ri ? next_rst(scan, scan_len, 0) : scan_len;
    s.n = seg_end;
    size_t blk = 0;
    for (size_t m = 0; m < nmcu; ++m) {
        if (ri && m && m % ri == 0) {
            /* next interval */
            size_t start = seg_end + 2;
            if (start > scan_len) { rc = OJPEG_ERR_EOF; goto done; }
            seg_end = next_rst(scan, scan_len, start);
            s.d = scan; s.pos = start; s.n = seg_end; s.left = 0; s.c = 0; s.err = 0;
            memset(pred, 0, sizeof pred);
        }
        for (unsigned i = 0; i < info->ncomp; ++i) { /* src/jpeg.c:76-84 */
            unsigned H = info->comp[info->scomp[i].id].H, V = info->comp[info->scomp[i].id].V;
            for (unsigned b = 0; b < H * V; ++b) {
                int e = decode_block(qcoef + 64 * blk, &dc[i], &ac[i], &s, &pred[i]);
                if (e) { rc = e; goto done; }
                ++blk;
            }
        }
    }
done:
    for (unsigned i = 0; i < nc; ++i) { tree_free(&dc[i]); tree_free(&ac[i]); }
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * pixel stage
 * ---------------------------------------------------------------------------------------- */

void ojpeg_dequant_block(const int32_t qz[64], const uint8_t quant[64], int out[8][8]) {
    for (unsigned r = 0; r < 8; ++r)
        for (unsigned c = 0; c < 8; ++c) {
            unsigned z = ojpeg_zigzag[r][c];
            out[r][c] = qz[z] * quant[z]; /* src/decoding.c:46,65 then :68-72 */
        }
}

double ojpeg_idct_sample(const int M[8][8], unsigned x, unsigned y) {
    build_tables();
    double sum = 0; /* src/decoding.c:75-84: u outer, v inner, (M*T)*T, then /4 */
    for (unsigned u = 0; u < 8; ++u)
        for (unsigned v = 0; v < 8; ++v) sum += M[v][u] * T[u][x] * T[v][y];
    return sum / 4;
}

unsigned char ojpeg_sample_value(const int M[8][8], unsigned x, unsigned y) {
    long r = lround(ojpeg_idct_sample(M, x, y)) + 128; /* src/decoding.c:99 */
    return (unsigned char)(r < 0 ? 0 : r > 255 ? 255 : r);
}

/* src/decoding.c:90-112 for one data unit whose coefficients are already decoded */
static void store_unit(const ojpeg_info *info, uint8_t *img, const int M[8][8], unsigned chan,
                       unsigned gx, unsigned gy, unsigned hs, unsigned vs) {
    unsigned W = info->width, Hh = info->height;
    for (unsigned x = 0; x < 8 && (unsigned long)gx + x * hs < W; ++x)
        for (unsigned y = 0; y < 8 && (unsigned long)gy + y * vs < Hh; ++y) {
            unsigned char val = ojpeg_sample_value(M, x, y);
            unsigned px = gx + x * hs, py = gy + y * vs;
            for (unsigned h = 0; h < hs; ++h)
                for (unsigned v = 0; v < vs; ++v) {
                    /* the reference tests gx+h / gy+v here (decoding.c:104-105) and can write
                     * past the buffer on odd sizes; clip on the actual pixel instead */
                    if (px + h < W && py + v < Hh)
                        img[((size_t)(py + v) * W + (px + h)) * 3 + chan] = val;
                }
        }
}

void ojpeg_pixels_from_qcoef(const ojpeg_info *info, const int32_t *qcoef, uint8_t *img) {
    unsigned mw = 8u * info->maxH, mh = 8u * info->maxV;
    size_t blk = 0;
    /* Channels no scan component writes (grayscale: Cb, Cr).  The reference leaves them as the
     * caller's malloc left them (src/test.c:28, src/decoding.c:107), i.e. undefined; this build
     * defines them as 128, the neutral chroma, so that image_ycbcr_to_rgb yields gray. */
    unsigned written = 0;
    for (unsigned i = 0; i < info->ncomp; ++i)
        if (info->scomp[i].id >= 1 && info->scomp[i].id <= 3) written |= 1u << (info->scomp[i].id - 1);
    for (unsigned c = 0; c < 3; ++c)
        if (!(written >> c & 1))
            for (size_t p = 0; p < (size_t)info->width * info->height; ++p) img[3 * p + c] = 128;
    for (unsigned y = 0; y < info->height; y += mh)     /* src/jpeg.c:74 */
        for (unsigned x = 0; x < info->width; x += mw)  /* src/jpeg.c:75 */
            for (unsigned i = 0; i < info->ncomp; ++i) {
                unsigned id = info->scomp[i].id;
                unsigned H = info->comp[id].H, V = info->comp[id].V;
                for (unsigned v = 0; v < V; ++v)
                    for (unsigned h = 0; h < H; ++h) {
                        int M[8][8];
                        ojpeg_dequant_block(qcoef + 64 * blk, info->quant[info->comp[id].Tq], M);
                        ++blk;
                        /* channel = componentID - 1 (decoding.c:107); ids outside 1..3 are out
                         * of contract */
                        if (id >= 1 && id <= 3)
                            store_unit(info, img, M, id - 1, x + 8 * h, y + 8 * v,
                                       info->maxH / H, info->maxV / V);
                    }
            }
}

int ojpeg_decode_ycbcr(const ojpeg_info *info, const uint8_t *scan, size_t scan_len,
                       uint8_t *img, int honour_dri) {
    size_t nb = ojpeg_block_count(info);
    int32_t *q = malloc(nb * 64 * sizeof *q);
    if (!q) return OJPEG_ERR_EOF;
    int e = ojpeg_decode_qcoef(info, scan, scan_len, q, honour_dri);
    if (!e) ojpeg_pixels_from_qcoef(info, q, img);
    free(q);
    return e;
}

static double clampd(double n) { return n < 0 ? 0 : 255 < n ? 255 : n; } /* src/image.c:34-36 */

void ojpeg_ycbcr_to_rgb(uint8_t *img, unsigned width, unsigned height) {
    for (size_t i = 0; i < (size_t)width * height; ++i) { /* src/image.c:39-51 */
        uint8_t Y = img[3 * i], Cb = img[3 * i + 1], Cr = img[3 * i + 2];
        img[3 * i + 0] = (uint8_t)clampd(Y + 1.402 * (Cr - 128.0));
        img[3 * i + 1] = (uint8_t)clampd(Y - 0.34414 * (Cb - 128.0) - 0.71414 * (Cr - 128.0));
        img[3 * i + 2] = (uint8_t)clampd(Y + 1.772 * (Cb - 128.0));
    }
}

void ojpeg_rgb_to_ycbcr(uint8_t *img, unsigned width, unsigned height) {
    for (size_t i = 0; i < (size_t)width * height; ++i) { /* src/image.c:54-66 */
        uint8_t R = img[3 * i], G = img[3 * i + 1], B = img[3 * i + 2];
        img[3 * i + 0] = (uint8_t)clampd(0.299 * R + 0.587 * G + 0.114 * B);
        img[3 * i + 1] = (uint8_t)clampd(-0.1687 * R - 0.3313 * G + 0.5 * B + 128.0);
        img[3 * i + 2] = (uint8_t)clampd(0.5 * R - 0.4187 * G - 0.0813 * B + 128.0);
    }
}

uint8_t *ojpeg_decode_file_rgb(const uint8_t *file, size_t len, unsigned *w, unsigned *h,
                               int honour_dri, int *err) {
    ojpeg_info info;
    int e = ojpeg_parse(file, len, &info);
    if (e) { if (err) *err = e; return NULL; }
    uint8_t *img = calloc((size_t)info.width * info.height, 3);
    if (!img) { if (err) *err = OJPEG_ERR_EOF; return NULL; }
    e = ojpeg_decode_ycbcr(&info, file + info.scan_offset, len - info.scan_offset, img, honour_dri);
    if (e) { free(img); if (err) *err = e; return NULL; }
    ojpeg_ycbcr_to_rgb(img, info.width, info.height);
    *w = info.width; *h = info.height;
    if (err) *err = 0;
    return img;
}
