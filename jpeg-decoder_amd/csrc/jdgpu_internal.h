/*
 * jdgpu_internal.h -- structures shared by the host preparation code (jdgpu_prep.c, C) and the
 * HIP translation unit (jdgpu.hip).  Plain C; everything here is laid out for the device.
 */
#ifndef JDGPU_INTERNAL_H
#define JDGPU_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

#include "jpeg_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- bit-segment geometry ------------------------------------------------------------------
 * The unstuffed ("clean") entropy-coded stream of an image is laid out restart interval after
 * restart interval, each padded with 0xFF to a whole number of segments, so that every
 * interval starts a segment with a known decoder state.  One lane decodes one segment. */
#ifndef JD_SEG_WORDS
#define JD_SEG_WORDS   128u                   /* 32-bit words per segment */
#endif
#define JD_SEG_BYTES   (JD_SEG_WORDS * 4u)    /* 512 B  */
#define JD_SEG_BITS    (JD_SEG_BYTES * 8u)    /* 4096 b */
#define JD_WG_SEGS     256u                   /* segments (= lanes) per work-group, sync kernels */
#define JD_MAX_BPM     48u                    /* data units per MCU: 3 components of at most 4x4 (src/parsing.c:42-46) */
#define JD_LUT_BITS    10u                    /* primary Huffman lookup width */
#define JD_LUT_SLOTS   6u                     /* distinct tables a 3-component scan can use */

/* flattened Huffman table: replaces the pointer-chased tree of src/hufftree.c */
#define JD_LUT_SUB     384u                   /* second-level entries (the Annex K tables need about 220) */
#define JD_LUT_LINK    0x8000u
#define JD_LUT_NOSUB   0xFFFu
typedef struct {  /* the part the kernels keep in LDS */
    /* primary[next 16 bits >> (16-JD_LUT_BITS)], resolved entry: bits 4:0 bits consumed by the
     * code (1..16), 8:5 magnitude size, 12:9 zero run, 13 = no such code (the consumed length is
     * then that of the reference's failing tree walk).  DC tables: size = symbol, run = 0.
     * Link entry (bit 15, codes longer than JD_LUT_BITS below this prefix): 11:0 first entry in
     * sub[], 14:12 number s of further stream bits that index it (1..6);  11:0 = JD_LUT_NOSUB:
     * sub[] was too small for this table, use the canonical search below. */
    uint16_t primary[1u << JD_LUT_BITS];
    uint16_t sub[JD_LUT_SUB];
} jd_lut_hot; /* 2816 B */
typedef struct {
    jd_lut_hot hot;
    /* canonical description of the codes of length 11..16: with v = next 16 bits, the length is
     * the first L with v < limit[L-11]; symbol = symbols[(v >> (16-L)) + offset[L-11]].  The
     * host builds sub[] from it; the kernels fall back to it for JD_LUT_NOSUB links only. */
    uint32_t limit[6];
    int32_t  offset[6];
    uint8_t  symbols[256];
    uint32_t is_dc;
    uint32_t short_limit; /* 16-bit left-justified end of the codes of <= JD_LUT_BITS bits */
    uint32_t n_sub;       /* entries of sub[] in use */
    uint32_t pad;
} jd_lut; /* 2816 + 24 + 24 + 256 + 16 = 3136 B, multiple of 16 */

/* one lookup on the next 16 bits of the stream; shared by the kernels and the host test hook */
#ifdef __HIPCC__
#define JD_HD static __host__ __device__ __forceinline__
#else
#define JD_HD static inline
#endif
typedef struct { uint32_t len, size, run, bad; } jd_sym;

/* codes longer than JD_LUT_BITS, and prefixes no code has, from the canonical ranges */
JD_HD jd_sym jd_lut_lookup_canon(const jd_lut *t, uint32_t v16) {
    jd_sym s;
    uint32_t L = JD_LUT_BITS + 1;
    for (int i = 0; i < 6; ++i) L += (v16 >= t->limit[i]) ? 1u : 0u;
    if (L > 16 || v16 < t->short_limit) {
        /* no code has this prefix: the reference's "Could not find next symbol in huffnode."
         * (src/hufftree.c:69,72).  Its bitwise walk stops at the first missing child, i.e.
         * at the first bit where v16 exceeds the last assigned code (limit[5] - 1, canonical
         * codes fill the code space from 0 upwards): that many bits are consumed, which is
         * what decides between this failure and end-of-data.  Speculative lanes just need
         * to make progress. */
        uint32_t lim = t->limit[5];
        uint32_t diff = (v16 ^ (lim - 1u)) & 0xFFFFu;
        s.len = (lim == 0 || L <= 16) ? (lim == 0 ? 1u : 16u) : (uint32_t)__builtin_clz(diff | 1u) - 15u;
        s.size = 0; s.run = 0; s.bad = 1;
    } else {
        uint32_t idx = (uint32_t)((int32_t)(v16 >> (16 - L)) + t->offset[L - JD_LUT_BITS - 1]) & 255u;
        uint32_t sym = t->symbols[idx];
        s.len = L; s.size = sym & 15u; s.bad = 0;
        if (t->is_dc) { s.run = 0; s.bad = sym > 15u; } else { s.run = sym >> 4; }
    }
    return s;
}

/* h: the table's hot part (the kernels pass their LDS copy), t: the whole table */
JD_HD jd_sym jd_lut_lookup(const jd_lut_hot *h, const jd_lut *t, uint32_t v16) {
    jd_sym s;
    uint32_t e = h->primary[v16 >> (16 - JD_LUT_BITS)];
    if (__builtin_expect((e & JD_LUT_LINK) != 0, 0)) {
        const uint32_t first = e & 0xFFFu;
        if (__builtin_expect(first == JD_LUT_NOSUB, 0)) return jd_lut_lookup_canon(t, v16);
        const uint32_t rest = v16 & ((1u << (16 - JD_LUT_BITS)) - 1u);
        e = h->sub[first + (rest >> ((16 - JD_LUT_BITS) - ((e >> 12) & 7u)))];
    }
    s.len = e & 31u; s.size = (e >> 5) & 15u; s.run = (e >> 9) & 15u; s.bad = (e >> 13) & 1u;
    return s;
}

/* per-segment word written by the unstuff stage */
#define JD_SEG_FIRST 0x80000000u              /* first segment of a restart interval */
/* low 16 bits: number of valid bits in the segment (0..JD_SEG_BITS) */

/* decoder state at a segment boundary, packed */
/* bits 5:0 overshoot into the next segment, 12:6 zigzag index k (0 = expecting DC),
 * 18:13 data unit within the MCU (0 .. JD_MAX_BPM - 1), 19 decode error seen */
#define JD_ST_PACK(ovs, k, b) ((uint32_t)(ovs) | ((uint32_t)(k) << 6) | ((uint32_t)(b) << 13))
#define JD_ST_UNIT(st) (((st) >> 13) & 63u)
#define JD_ST_BAD 0x80000u
#define JD_ST_NONE 0xFFFFFFFFu

#define JD_NARROW_MIN_Q63 8u
#ifndef JD_LONG_UNIT_BITS
#define JD_LONG_UNIT_BITS 128u
#endif
typedef struct {
    uint64_t raw_off;      /* byte offset of the stuffed scan in the raw buffer */
    uint64_t clean_off;    /* byte offset of the clean stream (multiple of JD_SEG_BYTES) */
    uint64_t pix_off;      /* byte offset of the output image (multiple of 256) */
    uint64_t blk_off;      /* index of the image's first 8x8 block in the coefficient buffer */
    uint32_t raw_len;
    uint32_t seg_off;      /* index of the image's first segment in the per-segment arrays */
    uint32_t seg_cap;      /* segments reserved (upper bound) */
    uint32_t n_segs;       /* segments in use (host prep: exact; GPU unstuff: written by K0) */
    uint32_t n_blocks;     /* data units in the scan */
    uint32_t ri_blocks;    /* data units per restart interval, 0 = no restart markers */
    uint32_t n_intervals;
    uint32_t ivl_off;      /* index of the image's first interval in the interval arrays */
    uint32_t u_off;        /* index of its first entry in the interval-offset array (n_intervals+1 each) */
    uint32_t chunk_off;    /* index of its first raw chunk in the per-chunk arrays */
    uint32_t n_chunks;     /* raw chunks of JD_RAW_CHUNK bytes */
    uint32_t pix_mode;     /* JD_PIX_* fast path of the pixel kernel */
    uint16_t lut_idx[JD_LUT_SLOTS]; /* the image's table slots: indices into the batch's array of DISTINCT flattened
                              tables (a batch mostly repeats a handful: the Annex K ones above all) */
    uint32_t mcus_x, mcus_y;
    uint16_t width, height;
    uint8_t  ncomp, bpm, maxH, maxV;
    uint8_t  blk_comp[JD_MAX_BPM];  /* scan-component index of data unit b of an MCU */
    uint8_t  blk_h[JD_MAX_BPM];     /* its h, v position inside the component's MCU area */
    uint8_t  blk_v[JD_MAX_BPM];
    uint8_t  blk_dc[JD_MAX_BPM];    /* lut slot of its DC / AC table */
    uint8_t  blk_ac[JD_MAX_BPM];
    uint8_t  comp_H[4], comp_V[4], comp_chan[4], comp_tq[4];
    uint8_t  pre_wide;              /* 1: take the wide (int16, 128-byte unit) path whatever the light decodes see:
                                       a table whose last entry is below JD_NARROW_MIN_Q63 makes a last AC
                                       coefficient beyond the narrow unit's 4 bits likely enough */
    uint8_t  long_units;            /* 1: data units of many symbols (scan bits per unit >= JD_LONG_UNIT_BITS): the light
                                       decodes take their longer iteration (JD_ASTEPS_LONG, jdgpu.hip) */
    uint32_t stride;                /* output pixels from one row to the next: width, or with JDGPU_OUT_PITCHED the
                                       next count that puts every row on a 256-byte boundary */
    uint32_t first_span;            /* consecutive segments a lane of k_huff_first takes in this image (1 .. JD_FIRST_SPAN_MAX, jd_plan_build) */
    uint16_t quant[4][64];          /* dequantisation tables re-ordered for the pixel kernel:
                                       index = col*8 + row (column-major natural order) */
} jd_image; /* keep a multiple of 8 bytes */

/* work descriptors: one per work-group */
typedef struct { uint32_t img, seg0; } jd_huff_wg;              /* one WAVE: 64 segments of image img from seg0;
                                                                   JD_WG_WAVES consecutive entries form a work-group */
#define JD_WG_WAVES    (JD_WG_SEGS / 64u)
#ifndef JD_FIRST_SPAN_MAX
#define JD_FIRST_SPAN_MAX 3u   /* k_huff_first: a lane decodes up to this many consecutive segments -- only the
                                  first needs a guessed start state (and the overlap decode that guesses it),
                                  the others continue; jd_plan_build picks the span per batch */
#endif
#ifndef JD_CP
#define JD_CP 3u    /* checkpoints inside a segment (jdgpu.hip: decode_light) */
#endif
#ifndef JD_WRITE_SUB_MAX_WAVES
#define JD_WRITE_SUB_MAX_WAVES 2560u   /* a batch of at most this many 64-segment waves has its segments written by JD_CP + 1 lanes each */
#endif
#define JD_FIRST_SLOTS 8192u   /* waves of k_huff_first the chip holds at a time (256 CUs x 4 SIMDs x 8) */
#ifndef JD_FIRST_SPAN_MIN_SEGS
#define JD_FIRST_SPAN_MIN_SEGS 384u   /* ... in images of at least this many segments: a small image's last,
                                         partly filled wave would cost it more than the overlap decodes saved
                                         (config 5, sync stage: 384: 3.19, 192: 3.33, 128: 3.50, 64: 3.53 ms) */
#endif
#ifndef JD_FIRST_SPAN3_MIN_SEGS
#define JD_FIRST_SPAN3_MIN_SEGS 640u  /* ... and spans above 2 in images of at least this many segments, without restart intervals */
#endif
#define JD_NO_WAVE     0xFFFFFFFFu                             /* seg0 of a padding entry */
#ifndef JD_WWG_WAVES_BIG
#define JD_WWG_WAVES_BIG 8u                                    /* write kernel, batches whose images use ... */
#endif
#define JD_WWG_SLOTS_BIG 4u                                    /* ... at most this many table slots */
typedef struct { uint32_t img; uint16_t mcu_row, mcu_col0; } jd_pix_wg; /* strip of jd_plan::pix_rows MCU rows from mcu_row */
typedef struct { uint32_t img, chunk; } jd_raw_wg;               /* JD_RAW_CHUNK stuffed bytes */
#define JD_RAW_CHUNK 4096u /* 256 lanes x 16 bytes */
#ifndef JD_RAW_PER_WG
#define JD_RAW_PER_WG 3u    /* chunks a work-group of k_unstuff_count / k_unstuff_scatter takes, loads issued together
                               (unstuff stage, C3: 1: 1.50, 2: 1.36, 3: 1.30, 4: 1.36 ms) */
#endif

#ifndef JD_PIX_WAVES
#define JD_PIX_WAVES 4u    /* waves per pixel-kernel work-group; each wave owns 64 pixels of the strip */
#endif
#define JD_STRIP_PX (64u * JD_PIX_WAVES)  /* pixel-kernel strip width */
/* MCU columns a pixel work-group covers: every wave takes the whole MCUs that fit its 64 pixels (2 of the
 * 24-pixel MCUs of a layout with horizontal factor 3) */
#define JD_STRIP_MCUS(maxH) (JD_PIX_WAVES * (64u / (8u * (maxH))))
#define JD_JMAP 96u   /* most data units a wave has per MCU row: (64 / (8 maxH)) MCUs x (<= 3 * 4 * maxH) units */
#define JD_PIX_ROWS_MAX 16u  /* MCU rows a pixel-kernel work-group walks down at most (loads run one row ahead);
                                the plan takes fewer when the batch is too small to fill the chip otherwise */
#define JD_PIX_GENERIC 0u
#define JD_PIX_420 1u     /* ids 1,2,3 in scan order, Y 2x2, Cb/Cr 1x1 */
#define JD_PIX_444 2u     /* ids 1,2,3 in scan order, all 1x1 */
#define JD_PIX_422 3u     /* ... Y 2x1 */
#define JD_PIX_440 4u     /* ... Y 1x2 */
#define JD_PIX_GRAY 5u    /* one component, id 1, 1x1 */
#define JD_PIX_MODES 6u

/* ---- host preparation (jdgpu_prep.c) -------------------------------------------------------- */

typedef struct {
    size_t n;
    jd_image *img;           /* [n] */
    jd_lut *luts;            /* [n_luts]: the distinct (class, counts, symbols) of the batch, flattened once each */
    uint32_t n_luts;
    uint8_t *raw;            /* concatenated stuffed scans, each padded to 16 B: staging memory set by the caller
                                before jd_plan_gather(), not owned unless raw_owned */
    int raw_owned;
    size_t raw_bytes;
    size_t clean_bytes;      /* capacity of the clean stream buffer */
    uint32_t n_segs_cap;     /* total segment capacity */
    uint32_t n_intervals;    /* total */
    jd_huff_wg *hwg; uint32_t n_hwg;
    jd_huff_wg *rpl;                   /* the hwg waves in the repair kernel's order (n_hwg groups) */
    jd_huff_wg *fwg; uint32_t n_fwg;   /* k_huff_first: waves of 64 * first_span segments */
    jd_huff_wg *wwg; uint32_t n_wwg;   /* the same waves in work-groups of wwg_waves (write kernel) */
    uint32_t wwg_waves;
    uint32_t write_sub;                /* lanes of k_huff_write per segment: 1, or JD_CP + 1 in a small batch */
    jd_pix_wg *pwg;  uint32_t n_pwg; uint32_t pix_rows;
    uint32_t pwg_mode_first[JD_PIX_MODES + 1]; /* pwg is grouped by jd_image::pix_mode: [first[m], first[m+1]) */
    jd_raw_wg *rwg;  uint32_t n_rwg;
    uint32_t n_u;            /* entries of the interval-offset array */
    uint64_t total_pixels, total_blocks, pix_bytes;
    uint32_t px_bytes;     /* output bytes per pixel: 3, or 4 with JDGPU_OUT_4BYTE */
    uint32_t pitched;      /* rows start on 256-byte boundaries (JDGPU_OUT_PITCHED) */
} jd_plan;

int  jd_plan_build(const jdgpu_image_desc *descs, size_t n, unsigned px_bytes, unsigned pitched, jd_plan *plan, char *err, size_t errlen);
void jd_plan_gather(jd_plan *plan, const jdgpu_image_desc *descs, size_t n); /* copies the scans into plan->raw (threads) */
void jd_plan_free(jd_plan *plan);
int  jd_lut_build(const jdgpu_huff_spec *spec, int is_dc, jd_lut *lut);
uint32_t jd_unstuff_image(const uint8_t *src, size_t raw_len, int use_rst, uint32_t n_intervals,
                          uint8_t *dst, uint32_t *seg_info); /* host_hooks.c (libjdhooks.so, tests only) */

extern const uint8_t jd_zigzag_to_cm[64];   /* zigzag k -> col*8+row */

#ifdef __cplusplus
}
#endif
#endif
