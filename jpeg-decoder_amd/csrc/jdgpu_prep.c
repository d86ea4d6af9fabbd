/*
 * jdgpu_prep.c -- host-side preparation of a decode batch (plain C, no HIP).
 *
 * Turns n jdgpu_image_desc into device-ready flat arrays: per-image geometry records,
 * flattened Huffman lookup tables (replacing the tree walk of the reference's
 * src/hufftree.c:64-78; canonical code assignment as in src/hufftree.c:46-59), dequantisation
 * tables re-ordered for the pixel kernel, the concatenated scan bytes and the work-group
 * descriptors of the Huffman and pixel kernels.
 *
 * The unstuff stage (src/bitstream.c:7-18 plus restart-marker splitting) runs on the GPU
 * (k_unstuff_* in jdgpu.hip).  Its plain-C statement, which the tests check the GPU stage
 * against, is test infrastructure and lives in host_hooks.c / libjdhooks.so, outside this library.
 */
#include "jdgpu_internal.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

const uint8_t jd_zigzag_to_cm[64] = {
    /* zigzag position k -> col*8+row of the natural position (walk of the anti-diagonals,
     * T.81 figure A.6; same order the reference tabulates in src/decoding.c:9-18) */
    0,  8,  1,  2,  9,  16, 24, 17, 10, 3,  4,  11, 18, 25, 32, 40, 33, 26, 19, 12, 5,  6,
    13, 20, 27, 34, 41, 48, 56, 49, 42, 35, 28, 21, 14, 7,  15, 22, 29, 36, 43, 50, 57, 58,
    51, 44, 37, 30, 23, 31, 38, 45, 52, 59, 60, 53, 46, 39, 47, 54, 61, 62, 55, 63};

static void seterr(char *err, size_t n, const char *msg) {
    if (err && n) snprintf(err, n, "%s", msg);
}

/* ---- Huffman flattening ---------------------------------------------------------------------- */

int jd_lut_build(const jdgpu_huff_spec *spec, int is_dc, jd_lut *lut) {
    memset(lut, 0, sizeof *lut);
    lut->is_dc = (uint32_t)is_dc;
    memcpy(lut->symbols, spec->symbols, 256);
    uint32_t code = 0; /* src/hufftree.c:47: codeValue */
    int k = 0;
    const unsigned PB = JD_LUT_BITS;
    for (unsigned L = 1; L <= 16; ++L) {
        unsigned cnt = spec->counts[L - 1];
        if (L == PB + 1) lut->short_limit = code << (16 - L) > 65536u ? 65536u : code << (16 - L);
        if (L > PB) {
            /* canonical ranges are contiguous and increasing with L in left-justified space */
            uint64_t lim = ((uint64_t)code + cnt) << (16 - L);
            lut->limit[L - PB - 1] = lim > 65536 ? 65536u : (uint32_t)lim;
            lut->offset[L - PB - 1] = k - (int32_t)code;
        }
        for (unsigned i = 0; i < cnt; ++i, ++code, ++k) {
            if (k >= 256) return -1;
            if (L > PB) continue;
            if (code >= (1u << L)) return -1; /* over-subscribed table */
            unsigned sym = spec->symbols[k];
            unsigned size = is_dc ? sym : (sym & 15u), run = is_dc ? 0u : (sym >> 4);
            uint16_t e = (uint16_t)(L | (size << 5) | (run << 9));
            if (is_dc && sym > 15) e = (uint16_t)(L | 0x2000u); /* no such magnitude category */
            uint32_t lo = code << (PB - L), n = 1u << (PB - L);
            for (uint32_t j = 0; j < n; ++j) lut->hot.primary[lo + j] = e;
        }
        code <<= 1;
    }
    /* second level: every prefix the short codes leave open is resolved from the canonical
     * description, over all its 2^(16-PB) continuations, into the smallest table that is exact */
    const unsigned RB = 16 - PB, RN = 1u << RB;
    int overflow = 0;
    for (uint32_t p = 0; p < (1u << PB); ++p) {
        if (lut->hot.primary[p]) continue;
        uint16_t e[64];
        for (uint32_t x = 0; x < RN; ++x) {
            jd_sym s = jd_lut_lookup_canon(lut, (p << RB) | x);
            e[x] = (uint16_t)(s.len | (s.size << 5) | (s.run << 9) | (s.bad ? 0x2000u : 0u));
        }
        unsigned sb = 0; /* fewest further bits that decide the entry */
        for (; sb < RB; ++sb) {
            int same = 1;
            for (uint32_t x = 0; x < RN && same; ++x) same = e[x] == e[x & ~((RN >> sb) - 1u)];
            if (same) break;
        }
        if (sb == 0) { lut->hot.primary[p] = e[0]; continue; }
        if (overflow || lut->n_sub + (1u << sb) > JD_LUT_SUB) {
            overflow = 1;
            lut->hot.primary[p] = (uint16_t)(JD_LUT_LINK | JD_LUT_NOSUB);
            continue;
        }
        lut->hot.primary[p] = (uint16_t)(JD_LUT_LINK | (sb << 12) | lut->n_sub);
        for (uint32_t j = 0; j < (1u << sb); ++j) lut->hot.sub[lut->n_sub++] = e[j << (RB - sb)];
    }
    return 0;
}

/* Batches mostly repeat a handful of Huffman tables (the Annex K ones above all): every distinct
 * (class, counts, symbols) of a plan is flattened once, into plan->luts, and the images hold indices.
 * (Until round 3 every image carried its own copy of its six tables: 77 MB to build, upload and read
 * for a 4096-image batch.)  Open addressing over a hash of the specification; no limit on the number of
 * distinct tables but the 16-bit index. */
typedef struct {
    uint32_t cap;                      /* power of two */
    int32_t *slot;                     /* index into lut[] or -1 */
    const jdgpu_huff_spec **spec;      /* per distinct table: where its specification lies (caller's descriptors) */
    int *rc;                           /* jd_lut_build's verdict */
    uint32_t lut_cap;
} lut_index;

static uint32_t spec_hash(const jdgpu_huff_spec *spec, int is_dc) {
    uint32_t h = 2166136261u ^ (uint32_t)is_dc;
    const uint8_t *p = (const uint8_t *)spec;
    unsigned used = 16;
    for (unsigned i = 0; i < 16; ++i) used += spec->counts[i];   /* symbols beyond the counted ones do not matter */
    if (used > sizeof *spec) used = sizeof *spec;
    for (unsigned i = 0; i < used; ++i) h = (h ^ p[i]) * 16777619u;
    return h;
}
static int spec_equal(const jdgpu_huff_spec *a, const jdgpu_huff_spec *b) {
    if (a == b) return 1;
    if (memcmp(a->counts, b->counts, 16)) return 0;
    unsigned used = 0;
    for (unsigned i = 0; i < 16; ++i) used += a->counts[i];
    return memcmp(a->symbols, b->symbols, used > 256 ? 256 : used) == 0;
}

/* the index of (spec, is_dc) among the plan's distinct tables, flattening it on first sight; -1: out of
 * memory or more than 65535 distinct tables; *rc: jd_lut_build's verdict on the table */
static int32_t lut_intern(lut_index *ix, jd_plan *P, const jdgpu_huff_spec *spec, int is_dc, int *rc) {
    if (ix->cap == 0 || 2u * (P->n_luts + 1u) > ix->cap) {   /* grow and rehash */
        uint32_t cap = ix->cap ? ix->cap * 2u : 64u;
        int32_t *slot = malloc(cap * sizeof *slot);
        if (!slot) return -1;
        for (uint32_t i = 0; i < cap; ++i) slot[i] = -1;
        for (uint32_t k = 0; k < P->n_luts; ++k) {
            uint32_t h = spec_hash(ix->spec[k], (int)P->luts[k].is_dc) & (cap - 1u);
            while (slot[h] >= 0) h = (h + 1u) & (cap - 1u);
            slot[h] = (int32_t)k;
        }
        free(ix->slot); ix->slot = slot; ix->cap = cap;
    }
    uint32_t h = spec_hash(spec, is_dc) & (ix->cap - 1u);
    for (; ix->slot[h] >= 0; h = (h + 1u) & (ix->cap - 1u)) {
        const int32_t k = ix->slot[h];
        if ((int)P->luts[k].is_dc == is_dc && spec_equal(ix->spec[k], spec)) { *rc = ix->rc[k]; return k; }
    }
    if (P->n_luts >= 65535u) return -1;
    if (P->n_luts == ix->lut_cap) {
        uint32_t cap = ix->lut_cap ? ix->lut_cap * 2u : 16u;
        jd_lut *l = realloc(P->luts, cap * sizeof *l);
        if (!l) return -1;
        P->luts = l;
        const jdgpu_huff_spec **sp = realloc(ix->spec, cap * sizeof *sp);
        if (!sp) return -1;
        ix->spec = sp;
        int *r = realloc(ix->rc, cap * sizeof *r);
        if (!r) return -1;
        ix->rc = r;
        ix->lut_cap = cap;
    }
    const int32_t k = (int32_t)P->n_luts++;
    ix->rc[k] = jd_lut_build(spec, is_dc, &P->luts[k]);
    ix->spec[k] = spec;
    ix->slot[h] = k;
    *rc = ix->rc[k];
    return k;
}
static void lut_index_free(lut_index *ix) { free(ix->slot); free(ix->spec); free(ix->rc); memset(ix, 0, sizeof *ix); }

/* What the Huffman kernels keep in LDS for a work-group: the lookup tables of the slots and the
 * per-unit descriptors.  Images with equal keys may share a work-group (one wave each). */
typedef struct { uint16_t lut_idx[JD_LUT_SLOTS]; uint8_t nslots, bpm, long_units, comp[JD_MAX_BPM], dc[JD_MAX_BPM], ac[JD_MAX_BPM]; } table_key;

/* ---- gathering the scans into one staging buffer ------------------------------------------------
 * 1.5 GB for a 4096 x 1080p batch: a single thread spends ~300 ms in first-touch page faults and
 * the copy, so the images are split over a few threads. */
#include <pthread.h>
typedef struct { jd_plan *P; const jdgpu_image_desc *descs; size_t i0, i1; } gather_job;
static void *gather_run(void *arg) {
    gather_job *j = arg;
    for (size_t i = j->i0; i < j->i1; ++i) {
        const jd_image *im = &j->P->img[i];
        if (!im->ncomp) continue;
        size_t len = j->descs[i].scan_len;
        memcpy(j->P->raw + im->raw_off, j->descs[i].scan, len);
        memset(j->P->raw + im->raw_off + len, 0, (len + 16 + 15) / 16 * 16 - len);
    }
    return NULL;
}
void jd_plan_gather(jd_plan *P, const jdgpu_image_desc *descs, size_t n) {
    memset(P->raw + P->raw_bytes - 64, 0, 64);
    enum { MAXT = 16 };
    size_t nt = P->raw_bytes >> 24; /* one thread per 16 MiB, at most MAXT */
    if (nt > MAXT) nt = MAXT;
    if (nt > n) nt = n;
    gather_job job[MAXT];
    pthread_t th[MAXT];
    if (nt < 2) { gather_job j = {P, descs, 0, n}; gather_run(&j); return; }
    int running[MAXT];
    for (size_t t = 0; t < nt; ++t) {
        job[t].P = P; job[t].descs = descs; job[t].i0 = n * t / nt; job[t].i1 = n * (t + 1) / nt;
        running[t] = pthread_create(&th[t], NULL, gather_run, &job[t]) == 0;
        if (!running[t]) gather_run(&job[t]);
    }
    for (size_t t = 0; t < nt; ++t) if (running[t]) pthread_join(th[t], NULL);
}

/* ---- plan -------------------------------------------------------------------------------------- */

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int jd_plan_build(const jdgpu_image_desc *descs, size_t n, unsigned px_bytes, unsigned pitched, jd_plan *P, char *err, size_t errlen) {
    memset(P, 0, sizeof *P);
    P->n = n;
    P->px_bytes = px_bytes;
    P->pitched = pitched;
    P->img = calloc(n ? n : 1, sizeof *P->img);
    P->luts = NULL; P->n_luts = 0;
    if (!P->img) { seterr(err, errlen, "out of host memory"); return JDGPU_ERR_NOMEM; }

    lut_index lix; memset(&lix, 0, sizeof lix);
    table_key *keys = calloc(n ? n : 1, sizeof *keys);
    if (!keys) { seterr(err, errlen, "out of host memory"); return JDGPU_ERR_NOMEM; }
    size_t raw = 0, clean = 0, pix = 0;
    uint64_t blk = 0;
    uint32_t segs = 0, ivls = 0, np = 0, nr = 0, nu = 0;
    for (size_t i = 0; i < n; ++i) {
        const jdgpu_image_desc *d = &descs[i];
        jd_image *im = &P->img[i];
        /* three components, or one (grayscale, SURVEY 8f2: the absent channels read 128) */
        const unsigned nc = d->ncomp == 1 ? 1u : 3u;
        int ok = (d->ncomp == 3 || d->ncomp == 1) && d->width && d->height && d->scan_len < 0xFFFFFF00u;
        if (d->ncomp == 1 && (d->comp[0].H != 1 || d->comp[0].V != 1)) ok = 0; /* src/jpeg.c:80-81 would walk
                                                                                  H x V units per "MCU" */
        unsigned seen = 0, maxH = 0, maxV = 0, units = 0;
        for (unsigned c = 0; ok && c < nc; ++c) {
            units += (unsigned)d->comp[c].H * d->comp[c].V;
            unsigned id = d->comp[c].id;
            if (id < 1 || id > 3 || (seen >> id & 1)) ok = 0; /* channel = id-1, SURVEY 8(c) */
            seen |= 1u << id;
            /* factors 1..4, as the reference takes them (src/parsing.c:42-46) */
            if (d->comp[c].H < 1 || d->comp[c].H > 4 || d->comp[c].V < 1 || d->comp[c].V > 4) ok = 0;
            if (d->comp[c].Tq > 3 || d->comp[c].Td > 3 || d->comp[c].Ta > 3) ok = 0;
            if (ok && (!d->huff_defined[0][d->comp[c].Td] || !d->huff_defined[1][d->comp[c].Ta])) ok = 0;
            if (d->comp[c].H > maxH) maxH = d->comp[c].H;
            if (d->comp[c].V > maxV) maxV = d->comp[c].V;
        }
        if (units > JD_MAX_BPM) ok = 0; /* cannot happen: 3 components of at most 4x4 */
        /* The replication ratios must be whole numbers.  The reference stretches by the integer quotient
         * maxH / H (src/jpeg.c:82): with, say, luma 3x1 beside a 2x1 component that one's units cover 16 of
         * the MCU's 24 columns and the rest of its channel is whatever malloc left (src/test.c:28) -- no
         * output to be identical to. */
        for (unsigned c = 0; ok && c < nc; ++c)
            if (maxH % d->comp[c].H || maxV % d->comp[c].V) ok = 0;
        im->width = d->width; im->height = d->height;
        im->raw_off = raw; im->clean_off = clean; im->pix_off = pix; im->blk_off = blk;
        im->seg_off = segs; im->ivl_off = ivls;
        /* row stride in pixels.  Pitched: 3-byte pixels and the 1-byte rows of a plane need a multiple of
         * 256 pixels for rows on 256-byte boundaries, 4-byte pixels a multiple of 64 */
        im->stride = pitched ? (uint32_t)align_up(d->width, px_bytes == 4 ? 64 : 256) : d->width;
        pix += align_up((size_t)im->stride * d->height * px_bytes, 256);
        P->total_pixels += (uint64_t)d->width * d->height;
        if (!ok) { /* marked unsupported: no segments, no work-groups, pixels stay zero */
            im->ncomp = 0;
            continue;
        }
        im->ncomp = (uint8_t)nc; im->maxH = (uint8_t)maxH; im->maxV = (uint8_t)maxV;
        for (unsigned c = nc; c < 3; ++c) { /* placeholders the pixel kernel never samples */
            im->comp_H[c] = 1; im->comp_V[c] = 1; im->comp_tq[c] = 0;
            im->comp_chan[c] = (uint8_t)((d->comp[0].id - 1 + c) % 3);
        }
        /* lookup tables: slot per distinct (class,id) in order of first use */
        int slot_of[2][4]; memset(slot_of, -1, sizeof slot_of);
        unsigned nslots = 0, b = 0;
        for (unsigned c = 0; c < nc; ++c) {
            unsigned td = d->comp[c].Td, ta = d->comp[c].Ta;
            for (int cls = 0; cls < 2; ++cls) {   /* DC table, then AC table of the component */
                const unsigned t = cls ? ta : td;
                if (slot_of[cls][t] >= 0) continue;
                int brc = 0;
                const int32_t id = lut_intern(&lix, P, &d->huff[cls][t], !cls, &brc);
                if (id < 0) { lut_index_free(&lix); free(keys); seterr(err, errlen, "out of host memory (Huffman tables)"); return JDGPU_ERR_NOMEM; }
                if (brc) ok = 0;
                slot_of[cls][t] = (int)nslots;
                im->lut_idx[nslots] = keys[i].lut_idx[nslots] = (uint16_t)id;
                ++nslots;
            }
            im->comp_H[c] = d->comp[c].H; im->comp_V[c] = d->comp[c].V;
            im->comp_chan[c] = (uint8_t)(d->comp[c].id - 1); im->comp_tq[c] = d->comp[c].Tq;
            for (unsigned v = 0; v < d->comp[c].V; ++v)       /* src/jpeg.c:80-81: v outer, h inner */
                for (unsigned h = 0; h < d->comp[c].H; ++h, ++b) {
                    im->blk_comp[b] = (uint8_t)c; im->blk_h[b] = (uint8_t)h; im->blk_v[b] = (uint8_t)v;
                    im->blk_dc[b] = (uint8_t)slot_of[0][td]; im->blk_ac[b] = (uint8_t)slot_of[1][ta];
                    keys[i].comp[b] = (uint8_t)c; keys[i].dc[b] = im->blk_dc[b]; keys[i].ac[b] = im->blk_ac[b];
                }
        }
        if (!ok) { im->ncomp = 0; continue; }
        for (unsigned k = nslots; k < JD_LUT_SLOTS; ++k) im->lut_idx[k] = im->lut_idx[0];   /* unused slots: any valid table */
        im->bpm = (uint8_t)b;
        keys[i].bpm = (uint8_t)b; keys[i].nslots = (uint8_t)nslots;
        {
            int std_ids = nc == 3 && d->comp[0].id == 1 && d->comp[1].id == 2 && d->comp[2].id == 3;
            int c11 = d->comp[1].H == 1 && d->comp[1].V == 1 && d->comp[2].H == 1 && d->comp[2].V == 1;
            im->pix_mode = JD_PIX_GENERIC;
            if (std_ids && c11 && d->comp[0].H == 2 && d->comp[0].V == 2) im->pix_mode = JD_PIX_420;
            if (std_ids && c11 && d->comp[0].H == 1 && d->comp[0].V == 1) im->pix_mode = JD_PIX_444;
            if (std_ids && c11 && d->comp[0].H == 2 && d->comp[0].V == 1) im->pix_mode = JD_PIX_422;
            if (std_ids && c11 && d->comp[0].H == 1 && d->comp[0].V == 2) im->pix_mode = JD_PIX_440;
            if (nc == 1 && d->comp[0].id == 1) im->pix_mode = JD_PIX_GRAY;
        }
        for (unsigned t = 0; t < 4; ++t)
            for (unsigned k = 0; k < 64; ++k) im->quant[t][jd_zigzag_to_cm[k]] = d->quant[t][k];
        /* Narrow coefficient units keep the last zigzag coefficient in 4 bits; an image that breaks
         * that costs its whole batch a second pass.  With a quantisation step of 8 or more there a
         * value beyond +-7 needs a coefficient beyond +-60 at the highest frequency, which only
         * synthetic checkerboards have; below (quality 97 and up, where AC values of 8 bits and more
         * are the rule anyway) the image goes the wide way from the start.  A limit of 16 was
         * measured on config 5: its q93..95 images, 6 % of the batch, went wide and the write stage
         * took 7.2 instead of 6.3 ms. */
        im->pre_wide = 0;
        for (unsigned c = 0; c < nc; ++c)
            if (d->quant[d->comp[c].Tq][63] < JD_NARROW_MIN_Q63) im->pre_wide = 1;
        im->mcus_x = (d->width + 8u * maxH - 1) / (8u * maxH);   /* src/jpeg.c:66-75 */
        im->mcus_y = (d->height + 8u * maxV - 1) / (8u * maxV);
        uint32_t nmcu = im->mcus_x * im->mcus_y;
        im->n_blocks = nmcu * b;
        uint32_t ri = d->restart_interval;
        im->ri_blocks = ri ? ri * b : 0;
        im->n_intervals = ri ? (nmcu + ri - 1) / ri : 1;
        im->raw_len = (uint32_t)d->scan_len;
        {
            /* JDGPU_LONG_UNIT_BITS (experiments): threshold in scan bits per data unit */
            const char *e = getenv("JDGPU_LONG_UNIT_BITS");
            const uint64_t thr = e ? strtoull(e, NULL, 10) : JD_LONG_UNIT_BITS;
            im->long_units = (uint64_t)d->scan_len * 8u >= thr * im->n_blocks;
            keys[i].long_units = im->long_units;   /* the staged tables differ (symbol groups or not): no shared work-group */
        }
        im->seg_cap = (uint32_t)((d->scan_len + JD_SEG_BYTES - 1) / JD_SEG_BYTES) + im->n_intervals;
        if ((uint64_t)((im->seg_cap + 63) / 64) * 64 * JD_SEG_BYTES >= 0xFFFFF000ull) {
            /* the bit readers address an image's clean stream with 32-bit byte offsets */
            im->ncomp = 0; im->n_blocks = 0; im->n_intervals = 0; im->seg_cap = 0; im->raw_len = 0;
            continue;
        }
        raw += align_up(d->scan_len + 16, 16);
        clean += (size_t)((im->seg_cap + 63) / 64) * 64 * JD_SEG_BYTES; /* whole 64-segment groups */
        im->u_off = nu; im->chunk_off = nr;
        im->n_chunks = (uint32_t)((d->scan_len + JD_RAW_CHUNK - 1) / JD_RAW_CHUNK);
        nu += im->n_intervals + 1; nr += im->n_chunks;
        segs += im->seg_cap; ivls += im->n_intervals; blk += im->n_blocks;
        uint32_t strip = JD_STRIP_MCUS(maxH);
        np += im->mcus_y * ((im->mcus_x + strip - 1) / strip); /* single-row strips: regrouped below */
    }
    lut_index_free(&lix);
    P->raw_bytes = raw + 64; P->clean_bytes = clean + JD_SEG_BYTES * 2;
    P->n_segs_cap = segs; P->n_intervals = ivls; P->total_blocks = blk; P->pix_bytes = pix;
    /* rows per pixel work-group: as many as leave ~8 work-groups per CU slot round */
    {
        uint32_t rows = np / 8192u;
        P->pix_rows = rows < 1 ? 1 : rows > JD_PIX_ROWS_MAX ? JD_PIX_ROWS_MAX : rows;
    }
    P->n_pwg = np; P->n_rwg = nr; P->n_u = nu;
    P->rwg = calloc(nr ? nr : 1, sizeof *P->rwg);
    P->raw = NULL; /* staging memory comes from the caller: jd_plan_gather() */
    P->pwg = calloc(np ? np : 1, sizeof *P->pwg);
    if (!P->pwg || !P->rwg) { free(keys); seterr(err, errlen, "out of host memory"); return JDGPU_ERR_NOMEM; }
    np = nr = 0;
    for (size_t i = 0; i < n; ++i) {
        jd_image *im = &P->img[i];
        if (!im->ncomp) continue;
        for (uint32_t ch = 0; ch < im->n_chunks; ++ch) { P->rwg[nr].img = (uint32_t)i; P->rwg[nr].chunk = ch; ++nr; }
    }
    /* Huffman work-groups: JD_WG_WAVES waves of 64 segments, one descriptor per wave.  The waves
     * of a work-group share the tables in LDS, so they may belong to different images only when
     * those images have the same table key; images are emitted group by group (groups in order
     * of first appearance, at most MAXG shared ones; an image whose tables fell out of the
     * cache, or beyond MAXG groups, stands alone) and a work-group is padded at a group's end.
     * A 1080p image fills its 3 work-groups alone; thumbnails of one encoder pack 4 per group. */
    {
        enum { MAXG = 64 };
        size_t rep[MAXG]; uint32_t ngrp = 0;
        int32_t *grp = malloc((n ? n : 1) * sizeof *grp);
        uint32_t *cnt = calloc(n + MAXG + 1, sizeof *cnt), nsolo = 0;
        if (!grp || !cnt) { free(grp); free(cnt); free(keys); seterr(err, errlen, "out of host memory"); return JDGPU_ERR_NOMEM; }
        for (size_t i = 0; i < n; ++i) {
            grp[i] = -1;
            if (!P->img[i].ncomp) continue;
            int32_t g = -1;
            for (uint32_t q = 0; q < ngrp; ++q)
                if (memcmp(&keys[rep[q]], &keys[i], sizeof keys[i]) == 0) { g = (int32_t)q; break; }
            if (g < 0 && ngrp < MAXG) { rep[ngrp] = i; g = (int32_t)ngrp++; }
            grp[i] = g < 0 ? (int32_t)(MAXG + nsolo++) : g;   /* solo images: a group each */
            cnt[grp[i]] += (P->img[i].seg_cap + 63u) / 64u;
        }
        /* the write kernel runs 8-wave work-groups when no image of the batch needs more than 4
         * table slots (2 DC + 2 AC, the usual case): 2 x (64 KB of unit buffers + 11 KB of tables)
         * fit a CU's LDS, 16 waves per CU instead of 12 */
        uint32_t max_slots = 0;
        for (size_t i = 0; i < n; ++i)
            if (P->img[i].ncomp && keys[i].nslots > max_slots) max_slots = keys[i].nslots;
        uint32_t all_waves = 0;
        for (uint32_t g = 0; g < MAXG + nsolo; ++g) all_waves += cnt[g];
        /* ... and only when there are enough waves to fill the chip twice over with them (a batch
         * of thumbnails spreads better in 4-wave groups: C2 0.68 against 0.79 ms) */
        const char *force = getenv("JDGPU_WWG_WAVES");   /* tests: "8" / "4" take the other path where it is legal */
        uint32_t want_big = force ? (uint32_t)(atoi(force) != 4) : (uint32_t)(all_waves >= 8192u);
        P->wwg_waves = (max_slots <= JD_WWG_SLOTS_BIG && want_big) ? JD_WWG_WAVES_BIG : JD_WG_WAVES;
        uint32_t *first = malloc((n + MAXG + 1) * sizeof *first);
        if (!first) { free(grp); free(cnt); free(keys); seterr(err, errlen, "out of host memory"); return JDGPU_ERR_NOMEM; }
        uint32_t max_waves = 0;
        for (size_t i = 0; i < n; ++i)
            if (grp[i] >= 0 && (P->img[i].seg_cap + 63u) / 64u > max_waves) max_waves = (P->img[i].seg_cap + 63u) / 64u;
        /* pass 0: light kernels, pass 1: write kernel, pass 2: repair kernel.  The repair kernel's
         * waves index an image's repair LIST (entries [seg0, seg0 + 64)); from the second round
         * on only the first 64 entries of an image exist, so its list is ordered entry block by
         * entry block across the images of a group: the live waves of a late round then sit
         * together in few work-groups instead of one per image (a work-group with a single live
         * wave still holds 17 KB of LDS).  Per 4096 x 1080p sync 7.35 -> 7.15 ms, C4 4.22 -> 3.83 ms. */
        /* pass 3: the first light kernel, whose lanes take first_span consecutive segments each
         * (a wave then covers 64 * first_span segments) */
        {
            /* The span, from what was measured (sync stage, ms; 1080p 4:2:0 q85, 713 segments an image;
             * spans 1 / 2 / 3): 1024 images 1.82 / 1.78 / 1.80, 2048: 2.87 / 2.65 / 2.57, 3072: 3.93 /
             * 3.56 / 3.50, 4096: 4.98 / 4.48 / 4.34 (4: 5.41, 6: 5.95 -- the lanes of a wave then read
             * segments `span` apart, and the stream's 64-byte sectors are shared by fewer and fewer of
             * them); config 4 2.83 / 2.70 / 2.66, config 5 3.29 / 3.22 / 3.23.  So: 3 once the batch is
             * twice the waves the chip holds at span 1, 2 once it is as many, else 1 (a batch that
             * does not fill the chip is a matter of latency, and a longer lane is a longer kernel). */
            uint32_t best = all_waves >= 2u * JD_FIRST_SLOTS ? 3u : all_waves >= JD_FIRST_SLOTS ? 2u : 1u;
            if (best > JD_FIRST_SPAN_MAX) best = JD_FIRST_SPAN_MAX;
            uint32_t min_segs = JD_FIRST_SPAN_MIN_SEGS;
            {
                const char *fs = getenv("JDGPU_FIRST_SPAN");   /* tests and experiments: "3", or "3all" for images of any size */
                if (fs && atoi(fs) >= 1 && atoi(fs) <= 16) { best = (uint32_t)atoi(fs); if (strstr(fs, "all")) min_segs = 0; }
            }
            /* Round 3, after the light decodes learnt to take symbols in groups (the guess costs less now;
             * sync stage, spans 1 / 2 / 3): C3 - / 3.69 / 3.59, config 4 (an interval of 22 segments per MCU
             * row) 2.48 / 2.21 / 2.31, config 5 (mixed sizes) 2.94 / 2.99 / 3.11 ms.  So the longest span only in
             * large images without restart intervals; forced spans (JDGPU_FIRST_SPAN) apply as given. */
            const int forced = getenv("JDGPU_FIRST_SPAN") != NULL;
            for (size_t i = 0; i < n; ++i) {
                const jd_image *im = &P->img[i];
                uint32_t sp = im->seg_cap >= min_segs ? best : 1u;
                if (!forced && sp > 2u && (im->ri_blocks != 0 || im->seg_cap < JD_FIRST_SPAN3_MIN_SEGS)) sp = 2u;
                P->img[i].first_span = sp;
            }
        }
        /* The write kernel of a batch that leaves most of the chip idle (thumbnails, a single image)
         * is the latency of one lane's decode of a whole segment: there a segment is written by
         * JD_CP + 1 lanes, lane q from the state the light decodes recorded at checkpoint q - 1
         * (k_huff_write<.., SUB>), and a wave covers 64 / (JD_CP + 1) segments. */
        {
            const char *ws = getenv("JDGPU_WRITE_SUB");   /* tests and experiments: "1" / "4" */
            P->write_sub = (P->wwg_waves == JD_WG_WAVES && all_waves <= JD_WRITE_SUB_MAX_WAVES) ? JD_CP + 1u : 1u;
            if (ws && P->wwg_waves == JD_WG_WAVES) P->write_sub = atoi(ws) == 1 ? 1u : JD_CP + 1u;
        }
        for (int pass = 0; pass < 4; ++pass) {
            const uint32_t W = pass == 1 ? P->wwg_waves : JD_WG_WAVES;
#define SPAN_OF(i) (pass == 3 ? 64u * P->img[i].first_span : pass == 1 ? 64u / P->write_sub : 64u)   /* segments per wave */
            memset(cnt, 0, (n + MAXG + 1) * sizeof *cnt);
            for (size_t i = 0; i < n; ++i)
                if (grp[i] >= 0) cnt[grp[i]] += (P->img[i].seg_cap + SPAN_OF(i) - 1u) / SPAN_OF(i);
            /* first descriptor of every group, work-group aligned */
            uint32_t total = 0;
            for (uint32_t g = 0; g < MAXG + nsolo; ++g) {
                first[g] = total;
                total += (cnt[g] + W - 1) / W * W;
            }
            jd_huff_wg *L = malloc((total ? total : 1) * sizeof *L);
            if (!L) { free(first); free(grp); free(cnt); free(keys); seterr(err, errlen, "out of host memory"); return JDGPU_ERR_NOMEM; }
            for (uint32_t k = 0; k < total; ++k) { L[k].img = 0; L[k].seg0 = JD_NO_WAVE; }
            if (pass != 2 || all_waves < 8192u) { /* small batches: measured slower entry-block-major (C2 2.19 against 1.87 ms) */
                /* pass 3: the images whose lanes take several segments first -- their waves run that
                 * many times as long, and the kernel should not end on them */
                for (int sweep = pass == 3 ? 0 : 1; sweep < 2; ++sweep)
                    for (size_t i = 0; i < n; ++i) {
                        if (grp[i] < 0) continue;
                        if (pass == 3 && (P->img[i].first_span > 1u) != (sweep == 0)) continue;
                        for (uint32_t sg = 0; sg < P->img[i].seg_cap; sg += SPAN_OF(i)) {
                            uint32_t k = first[grp[i]]++;
                            L[k].img = (uint32_t)i; L[k].seg0 = sg;
                        }
                    }
            } else {
                for (uint32_t e = 0; e < max_waves; ++e)
                    for (size_t i = 0; i < n; ++i) {
                        if (grp[i] < 0 || e * 64u >= P->img[i].seg_cap) continue;
                        uint32_t k = first[grp[i]]++;
                        L[k].img = (uint32_t)i; L[k].seg0 = e * 64u;
                    }
            }
            /* padding descriptors name an image of their own work-group (its tables are staged) */
            for (uint32_t k = 0; k < total; ++k)
                if (L[k].seg0 == JD_NO_WAVE) L[k].img = L[k - k % W].img;
            if (pass == 1) { P->wwg = L; P->n_wwg = total / W; }
            else if (pass == 0) { P->hwg = L; P->n_hwg = total / W; }
            else if (pass == 2) P->rpl = L;
            else { P->fwg = L; P->n_fwg = total / W; }
        }
        free(first);
        free(grp); free(cnt);
    }
    free(keys);
    /* pixel work-groups, grouped by layout (one kernel instantiation each) */
    for (uint32_t mode = 0; mode < JD_PIX_MODES; ++mode) {
        P->pwg_mode_first[mode] = np;
        for (size_t i = 0; i < n; ++i) {
            jd_image *im = &P->img[i];
            if (!im->ncomp || im->pix_mode != mode) continue;
            uint32_t strip = JD_STRIP_MCUS(im->maxH);
            for (uint32_t r = 0; r < im->mcus_y; r += P->pix_rows)
                for (uint32_t c = 0; c < im->mcus_x; c += strip) {
                    P->pwg[np].img = (uint32_t)i; P->pwg[np].mcu_row = (uint16_t)r; P->pwg[np].mcu_col0 = (uint16_t)c; ++np;
                }
        }
    }
    P->pwg_mode_first[JD_PIX_MODES] = np;
    P->n_pwg = np; /* after grouping rows */
    return JDGPU_OK;
}

unsigned jdgpu_segment_bytes(void) { return JD_SEG_BYTES; }

void jd_plan_free(jd_plan *P) {
    free(P->img); free(P->luts); if (P->raw_owned) free(P->raw); free(P->hwg); free(P->pwg); free(P->rwg); free(P->wwg); free(P->rpl); free(P->fwg);
    memset(P, 0, sizeof *P);
}
