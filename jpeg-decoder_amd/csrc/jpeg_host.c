/*
 * jpeg_host.c -- the host C side of the drop-in API (include/jpeg.h, include/image.h).
 *
 * Header/marker parsing stays on the host, as in the reference (src/jpeg.c:23-58,
 * src/parsing.c); the decode itself is handed to the MI355X backend through the C ABI of
 * include/jpeg_gpu.h.  There is NO CPU decode path in this library: without a HIP device
 * jpeg_read_image / image_ycbcr_to_rgb print the backend's error and exit(EXIT_FAILURE), the
 * reference's own convention for fatal conditions (src/macros.h:8).
 *
 * Written from the behaviour documented in SURVEY.md section 9, not from the reference's text.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jpeg.h"
#include "jpeg_gpu.h"

/* ---- byte source: FILE* or memory, so both front ends share the marker loop ------------------ */

typedef struct {
    FILE *fp;
    const uint8_t *mem;
    size_t len, pos;
    int failed;
} source;

static int src_byte(source *s) {
    if (s->fp) {
        int c = getc(s->fp);
        if (c == EOF) s->failed = 1;
        return c;
    }
    if (s->pos >= s->len) { s->failed = 1; return EOF; }
    return s->mem[s->pos++];
}

static unsigned src_word(source *s) { /* big-endian 16 bits, first byte first (src/jpeg.c:16) */
    int hi = src_byte(s);
    int lo = src_byte(s);
    return ((unsigned)hi << 8 | (unsigned)lo) & 0xFFFFu;
}

static size_t src_read(source *s, uint8_t *dst, size_t n) {
    if (s->fp) return fread(dst, 1, n, s->fp);
    size_t avail = s->len - s->pos;
    if (n > avail) n = avail;
    memcpy(dst, s->mem + s->pos, n);
    s->pos += n;
    return n;
}

/* ---- Huffman code tree (the layout callers of the reference can see, src/hufftree.h:15-22) ---- */

static huffnode *node_new(void) {
    huffnode *n = calloc(1, sizeof *n);
    if (!n) { fprintf(stderr, "Could not allocate memory for huffnode.\n"); exit(EXIT_FAILURE); }
    return n;
}

static void node_free_children(huffnode *n) {
    if (n->hasLeft && n->left) { node_free_children(n->left); free(n->left); }
    if (n->hasRight && n->right) { node_free_children(n->right); free(n->right); }
    n->left = n->right = NULL;
    n->hasLeft = n->hasRight = false;
}

/* Canonical assignment (T.81 annex C; src/hufftree.c:46-59): codes of one length are
 * consecutive, the first code of the next length is (last + 1) << 1.  Returns payload bytes. */
static size_t tree_from_dht(huffnode *root, const uint8_t *payload) {
    node_free_children(root); /* a redefinition must not leak (the reference's memset does) */
    memset(root, 0, sizeof *root);
    const uint8_t *sym = payload + 16;
    unsigned next = 0;
    for (unsigned len = 1; len <= 16; ++len, next <<= 1)
        for (unsigned i = payload[len - 1]; i; --i, ++next) {
            huffnode *n = root;
            unsigned code = next & 0xFFFFu;
            for (unsigned bit = len; bit--;) {
                if ((code >> bit) & 1u) {
                    if (!n->hasRight) { n->right = node_new(); n->hasRight = true; }
                    n = n->right;
                } else {
                    if (!n->hasLeft) { n->left = node_new(); n->hasLeft = true; }
                    n = n->left;
                }
            }
            n->symbol = *sym++;
            n->hasSymbol = true;
        }
    return (size_t)(sym - payload);
}

/* src/hufftree.h:24-25, reachable through the reference's jpeg.h:4: the tree builder and its
 * release.  Host-side header work, as in the reference; the decode never walks the tree. */
size_t hufftree_parse(huffnode *buf, const unsigned char *data) {
    memset(buf, 0, sizeof *buf); /* src/hufftree.c:44: *buf may be uninitialised, never read it */
    return tree_from_dht(buf, data);
}
void hufftree_destroy(huffnode *tree) { node_free_children(tree); }

/* in-order walk: leaves in increasing code order -> counts per depth + symbols (a DHT payload) */
static void tree_flatten(const huffnode *n, unsigned depth, jdgpu_huff_spec *spec, unsigned *nsym) {
    if (depth && n->hasSymbol) {
        if (depth <= 16 && *nsym < 256) { spec->counts[depth - 1]++; spec->symbols[(*nsym)++] = n->symbol; }
        return; /* the walk of src/hufftree.c:64-78 stops at the first symbol */
    }
    if (depth >= 16) return;
    if (n->hasLeft && n->left) tree_flatten(n->left, depth + 1, spec, nsym);
    if (n->hasRight && n->right) tree_flatten(n->right, depth + 1, spec, nsym);
}

/* ---- segment payload parsers: SURVEY.md section 9 items 2-5 ----------------------------------- */

enum { M_SOI = 0xFFD8, M_SOF0 = 0xFFC0, M_DHT = 0xFFC4, M_DQT = 0xFFDB, M_SOS = 0xFFDA, M_DRI = 0xFFDD };

static int seg_frame(jpeg_info *ji, const uint8_t *d, unsigned n) {
    if (n < 6 || n != 6u + 3u * d[5]) return -1;
    if ((ji->precision = d[0]) != 8) return -1;                 /* 12-bit samples unsupported */
    if ((ji->height = (unsigned short)(d[1] << 8 | d[2])) == 0) return -1;  /* DNL unsupported */
    if ((ji->width = (unsigned short)(d[3] << 8 | d[4])) == 0) return -1;
    for (const uint8_t *c = d + 6; c < d + n; c += 3) {
        unsigned h = c[1] >> 4, v = c[1] & 15u;
        if (h < 1 || h > 4 || v < 1 || v > 4 || c[2] > 3) return -1;
        ji->componentInfo[c[0]].HSamplingFactor = (unsigned char)h;
        ji->componentInfo[c[0]].VSamplingFactor = (unsigned char)v;
        ji->componentInfo[c[0]].quantizationTableID = c[2];
    }
    return 0;
}

static int seg_huffman(jpeg_info *ji, const uint8_t *d, unsigned n) {
    if (n < 1) return -1;
    for (size_t at = 0; at < n;) {
        unsigned cls = d[at] >> 4, id = d[at] & 15u;
        if (cls > 1 || id > 3) return -1;
        /* the 16 counts and the symbols they announce must lie inside the segment.  The
         * reference does not check (src/parsing.c:61-78, src/hufftree.c:53 read on into whatever
         * its 64 KiB buffer held before); a table cut off by the segment end is rejected here. */
        if (at + 17 > n) return -1;
        size_t total = 0;
        for (unsigned l = 0; l < 16; ++l) total += d[at + 1 + l];
        if (at + 17 + total > n) return -1;
        at += 1 + tree_from_dht(&ji->huffmanTables[cls][id], d + at + 1);
    }
    return 0;
}

/* The same for the batch parser, which needs no tree: the DHT payload IS the flat description
 * (counts per length + symbols in code order) the descriptor carries, whenever the table is an ordinary
 * prefix code -- the canonical assignment never runs out of codes of a length and there are at most 256
 * symbols.  Then building the tree (a calloc per code bit) and flattening it again gives back these very
 * bytes: leaves in code order are leaves in (length, index) order.  A table that is not (the reference's
 * builder wraps around and overwrites, src/hufftree.c:18-41; what such a table means is defined by that
 * tree) sets `fallback`, and the file goes through the tree path.  27 -> ~3 us per file. */
typedef struct { jdgpu_huff_spec spec[2][4]; uint8_t defined[2][4]; int fallback; } flat_tables;

static int seg_huffman_flat(flat_tables *ft, const uint8_t *d, unsigned n) {
    if (n < 1) return -1;
    for (size_t at = 0; at < n;) {
        unsigned cls = d[at] >> 4, id = d[at] & 15u;
        if (cls > 1 || id > 3) return -1;
        if (at + 17 > n) return -1;
        size_t total = 0;
        unsigned code = 0;
        for (unsigned l = 0; l < 16; ++l) {
            const unsigned cnt = d[at + 1 + l];
            total += cnt;
            code += cnt;
            if (code > (1u << (l + 1))) ft->fallback = 1;   /* more codes of this length than exist */
            code <<= 1;
        }
        if (at + 17 + total > n) return -1;
        if (total > 256) ft->fallback = 1;
        if (!ft->fallback) {
            jdgpu_huff_spec *sp = &ft->spec[cls][id];
            memset(sp, 0, sizeof *sp);
            memcpy(sp->counts, d + at + 1, 16);
            memcpy(sp->symbols, d + at + 17, total);
            ft->defined[cls][id] = total != 0;
        }
        at += 17 + total;
    }
    return 0;
}

static int seg_quant(jpeg_info *ji, const uint8_t *d, unsigned n) {
    if (n == 0 || n % 65u) return -1;
    for (; n; n -= 65, d += 65) {
        if (d[0] >> 4) return -1;                              /* 16-bit tables need 12-bit samples */
        if ((d[0] & 15u) > 3) return -1;
        memcpy(ji->quantizationTables[d[0] & 15u], d + 1, 64); /* file order = zigzag order */
    }
    return 0;
}

static int seg_scan(scan_info *si, const jpeg_info *ji, const uint8_t *d, unsigned n) {
    unsigned ns = d[0];
    if (n != 1u + 2u * ns + 3u) return -1;
    si->componentCount = (unsigned char)ns;
    unsigned mh = 0, mv = 0;
    for (unsigned i = 0; i < ns; ++i) {
        unsigned id = d[1 + 2 * i], td = d[2 + 2 * i] >> 4, ta = d[2 + 2 * i] & 15u;
        if (td > 3 || ta > 3) return -1;
        if (i < 4) {
            si->componentInfo[i].componentID = (unsigned char)id;
            si->componentInfo[i].DCHuffmanTableID = (unsigned char)td;
            si->componentInfo[i].ACHuffmanTableID = (unsigned char)ta;
        }
        if (ji->componentInfo[id].HSamplingFactor > mh) mh = ji->componentInfo[id].HSamplingFactor;
        if (ji->componentInfo[id].VSamplingFactor > mv) mv = ji->componentInfo[id].VSamplingFactor;
    }
    si->maxHSamplingFactor = (unsigned char)mh;
    si->maxVSamplingFactor = (unsigned char)mv;
    return 0;
}

/* the marker loop of src/jpeg.c:23-58: SOI, then {marker, length, payload} until SOS */
static int read_header(jpeg_info *ji, scan_info *si, source *s, flat_tables *ft /* NULL: build the trees */) {
    static _Thread_local uint8_t payload[65536];
    memset(ji, 0, sizeof *ji);
    memset(si, 0, sizeof *si);
    if (src_word(s) != M_SOI || s->failed) return -1;
    for (;;) {
        unsigned marker = src_word(s);
        /* extension: T.81 B.1.1.2 allows any number of 0xFF fill bytes in front
         * of a marker, and TEM / RSTn stand alone, without a length.  The reference reads
         * marker + length unconditionally (src/jpeg.c:32-37) and takes the bytes that follow as
         * a length.  Every other marker word goes the reference's way. */
        while (marker == 0xFFFFu && !s->failed) marker = 0xFF00u | ((unsigned)src_byte(s) & 0xFFu);
        if (s->failed) return -1;
        if (marker == 0xFF01u || (marker >= 0xFFD0u && marker <= 0xFFD7u)) continue;
        unsigned length = src_word(s);
        if (length < 2) return -1;
        unsigned n = length - 2;
        if (src_read(s, payload, n) != n) return -1;
        int rc = 0;
        if (marker == M_SOF0) rc = seg_frame(ji, payload, n);
        else if (marker == M_DHT) rc = ft ? seg_huffman_flat(ft, payload, n) : seg_huffman(ji, payload, n);
        else if (marker == M_DQT) rc = seg_quant(ji, payload, n);
        else if (marker == M_DRI) { if (n >= 2) ji->restartInterval = (unsigned short)(payload[0] << 8 | payload[1]); }
        else if (marker == M_SOS) rc = seg_scan(si, ji, payload, n);
        /* every other segment is skipped, as in the reference (src/jpeg.c:50) */
        if (rc) return rc;
        if (marker == M_SOS) return 0;
    }
}

int jpeg_read_info(jpeg_info *jpegInfo, scan_info *scanInfo, FILE *fp) {
    source s = {fp, NULL, 0, 0, 0};
    int rc = read_header(jpegInfo, scanInfo, &s, NULL);
    return rc ? -1 : 0;
}

void jpeg_free(jpeg_info *info) {
    for (int c = 0; c < 2; ++c)
        for (int t = 0; t < 4; ++t) node_free_children(&info->huffmanTables[c][t]);
}

/* ---- jpeg_info + scan_info + bytes -> flat descriptor ------------------------------------------ */

static int desc_from_info(jdgpu_image_desc *d, const jpeg_info *ji, const scan_info *si,
                          const uint8_t *scan, size_t scan_len, const flat_tables *ft /* NULL: flatten the trees */) {
    memset(d, 0, sizeof *d);
    d->scan = scan; d->scan_len = scan_len;
    d->width = ji->width; d->height = ji->height;
    d->ncomp = si->componentCount;
    d->restart_interval = ji->restartInterval;
    for (unsigned i = 0; i < si->componentCount && i < 4; ++i) {
        unsigned id = si->componentInfo[i].componentID;
        d->comp[i].id = (uint8_t)id;
        d->comp[i].H = ji->componentInfo[id].HSamplingFactor;
        d->comp[i].V = ji->componentInfo[id].VSamplingFactor;
        d->comp[i].Tq = ji->componentInfo[id].quantizationTableID;
        d->comp[i].Td = si->componentInfo[i].DCHuffmanTableID;
        d->comp[i].Ta = si->componentInfo[i].ACHuffmanTableID;
    }
    memcpy(d->quant, ji->quantizationTables, sizeof d->quant);
    if (ft) {
        for (int c = 0; c < 2; ++c)
            for (int t = 0; t < 4; ++t)
                if (ft->defined[c][t]) { d->huff[c][t] = ft->spec[c][t]; d->huff_defined[c][t] = 1; }
        return 0;
    }
    for (int c = 0; c < 2; ++c)
        for (int t = 0; t < 4; ++t) {
            const huffnode *root = &ji->huffmanTables[c][t];
            if (!root->hasLeft && !root->hasRight) continue;
            unsigned nsym = 0;
            tree_flatten(root, 0, &d->huff[c][t], &nsym);
            d->huff_defined[c][t] = 1;
        }
    return 0;
}

int jdgpu_parse_file(const uint8_t *file, size_t len, jdgpu_image_desc *desc) {
    jpeg_info ji;
    scan_info si;
    {
        /* without trees (see seg_huffman_flat); a table that is no ordinary prefix code sends the file
         * through the tree path below, whose result defines what such a table means */
        flat_tables ft;
        memset(ft.defined, 0, sizeof ft.defined);
        ft.fallback = 0;
        source s = {NULL, file, len, 0, 0};
        int rc = read_header(&ji, &si, &s, &ft);
        if (!ft.fallback) {
            if (!rc) desc_from_info(desc, &ji, &si, file + s.pos, len - s.pos, &ft);
            return rc ? JDGPU_ERR_HEADER : JDGPU_OK;
        }
    }
    source s = {NULL, file, len, 0, 0};
    int rc = read_header(&ji, &si, &s, NULL);
    if (!rc) desc_from_info(desc, &ji, &si, file + s.pos, len - s.pos, NULL);
    jpeg_free(&ji);
    return rc ? JDGPU_ERR_HEADER : JDGPU_OK;
}

/* Many files at once, split over host threads: one core parses ~27 us per file (tree build +
 * flatten), which for a 4096-image batch is three times its GPU decode time. */
#include <pthread.h>
#include <unistd.h>
typedef struct {
    const uint8_t *const *files; const size_t *lens; jdgpu_image_desc *descs; int *status;
    size_t i0, i1, failed;
} parse_job;
static void *parse_run(void *arg) {
    parse_job *j = arg;
    for (size_t i = j->i0; i < j->i1; ++i) {
        int rc = jdgpu_parse_file(j->files[i], j->lens[i], &j->descs[i]);
        if (rc) { memset(&j->descs[i], 0, sizeof j->descs[i]); ++j->failed; }
        if (j->status) j->status[i] = rc;
    }
    return NULL;
}
long jdgpu_parse_files(const uint8_t *const *files, const size_t *lens, size_t n,
                       jdgpu_image_desc *descs, int *status, unsigned threads) {
    if ((!files || !lens || !descs) && n) return JDGPU_ERR_ARG;
    enum { MAXT = 64 };
    if (threads == 0) {
        long cpus = sysconf(_SC_NPROCESSORS_ONLN);
        threads = cpus < 1 ? 1 : cpus > 16 ? 16 : (unsigned)cpus; /* the GPU boxes give a process 16 */
    }
    if (threads > MAXT) threads = MAXT;
    if (threads > n / 32 + 1) threads = (unsigned)(n / 32 + 1);  /* a thread start costs ~ a few parses */
    parse_job job[MAXT];
    pthread_t th[MAXT];
    int running[MAXT];
    for (unsigned t = 0; t < threads; ++t) {
        job[t] = (parse_job){files, lens, descs, status, n * t / threads, n * (t + 1) / threads, 0};
        running[t] = t + 1 < threads && pthread_create(&th[t], NULL, parse_run, &job[t]) == 0;
        if (!running[t]) parse_run(&job[t]); /* the last share, and any thread that would not start */
    }
    size_t failed = 0;
    for (unsigned t = 0; t < threads; ++t) {
        if (running[t]) pthread_join(th[t], NULL);
        failed += job[t].failed;
    }
    return (long)failed;
}

/* ---- the decode entry points ---------------------------------------------------------------------- */

static void die(const char *msg) { /* src/macros.h:8 */
    fprintf(stderr, "%s\n", msg);
    exit(EXIT_FAILURE);
}

void jpeg_read_image(pixel *img, const jpeg_info *info, const scan_info *scanInfo, FILE *fp) {
    /* the reference pulls bytes one at a time while decoding (src/bitstream.c:7-18); the GPU
     * needs the whole scan, so take everything up to EOF */
    size_t cap = 1 << 16, n = 0;
    uint8_t *buf = malloc(cap);
    if (!buf) die("Could not allocate memory for the scan.");
    for (;;) {
        size_t got = fread(buf + n, 1, cap - n, fp);
        n += got;
        if (got == 0) break;
        if (n == cap) { cap *= 2; buf = realloc(buf, cap); if (!buf) die("Could not allocate memory for the scan."); }
    }
    if (ferror(fp)) die("Could not read file."); /* src/bitstream.c:16 */

    jdgpu_image_desc d;
    desc_from_info(&d, info, scanInfo, buf, n, NULL);
    jdgpu_batch *b = NULL;
    int rc = jdgpu_batch_create(&d, 1, JDGPU_OUT_YCBCR, &b);
    if (!rc) rc = jdgpu_batch_decode(b);
    if (!rc) rc = jdgpu_batch_sync(b);
    if (rc) die(jdgpu_last_error());
    uint32_t st = jdgpu_batch_status(b, 0);
    if (st & JDGPU_IMG_UNSUPPORTED) die("Unsupported component layout (need 3 components or 1, ids 1..3, sampling factors 1..4 that divide the largest one).");
    if (st & JDGPU_IMG_NOSYMBOL) die("Could not find next symbol in huffnode."); /* src/hufftree.c:69,72 */
    if (st & JDGPU_IMG_EOF) die("Reached end of file.");                         /* src/bitstream.c:17 */
    if (st & JDGPU_IMG_RANGE) die("A DC coefficient leaves 16 bits (the GPU backend keeps quantised coefficients in int16).");
    if (jdgpu_batch_read_pixels(b, 0, img)) die(jdgpu_last_error());
    jdgpu_batch_destroy(b);
    free(buf);
}

void image_ycbcr_to_rgb(pixel *img, unsigned width, unsigned height) {
    /* images that fit the library's static device scratch (anything up to 4096 x 2730) are staged there:
     * no hipMalloc / hipFree per call (0.16 ms of birb.jpg's 1.36 ms); larger ones take the allocating path */
    const size_t bytes = (size_t)width * height * 3;
    const int rc = bytes <= jdgpu_noalloc_max_bytes() ? jdgpu_ycbcr_to_rgb_noalloc(img, width, height)
                                                      : jdgpu_ycbcr_to_rgb(img, width, height);
    if (rc) die(jdgpu_last_error());
}

/* ---- image.h functions that are off the decode path (host C) -------------------------------------- */

static double clamp_255(double v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

void image_rgb_to_ycbcr(pixel *img, unsigned width, unsigned height) {
    /* src/image.c:54-66: JFIF forward transform in doubles, truncating store */
    size_t n = (size_t)width * height;
    for (pixel *p = img; p < img + n; ++p) {
        double r = p->RGB.R, g = p->RGB.G, b = p->RGB.B;
        p->YCbCr.Y = (unsigned char)clamp_255(0.299 * r + 0.587 * g + 0.114 * b);
        p->YCbCr.Cb = (unsigned char)clamp_255(-0.1687 * r - 0.3313 * g + 0.5 * b + 128.0);
        p->YCbCr.Cr = (unsigned char)clamp_255(0.5 * r - 0.4187 * g - 0.0813 * b + 128.0);
    }
}

void image_print(const pixel *img, unsigned imageWidth, unsigned imageHeight, unsigned maxPrintWidth,
                 unsigned maxPrintHeight, unsigned pixelWidth) {
    /* src/image.c:13-31: one background-colour escape + pixelWidth spaces per pixel, rows end
     * with a colour reset; columns stop where x*pixelWidth reaches maxPrintWidth */
    unsigned rows = imageHeight < maxPrintHeight ? imageHeight : maxPrintHeight;
    for (unsigned y = 0; y < rows; ++y) {
        const pixel *row = img + (size_t)y * imageWidth;
        for (unsigned x = 0; x < imageWidth && x * pixelWidth < maxPrintWidth; ++x) {
            printf("\x1b[48;2;%u;%u;%um", row[x].RGB.R, row[x].RGB.G, row[x].RGB.B);
            for (unsigned i = 0; i < pixelWidth; ++i) putchar(' ');
        }
        fputs("\x1b[0m\n", stdout);
    }
}
