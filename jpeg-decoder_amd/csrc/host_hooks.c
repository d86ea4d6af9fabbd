/*
 * host_hooks.c -- TEST INFRASTRUCTURE, built into libjdhooks.so, never linked into libjpegamd.so.
 *
 * Plain-C statements of two things the product does elsewhere, so that the CPU test-suite can
 * check them without a GPU: the flattened-table lookup the kernels use (jd_lut_build +
 * jd_lut_lookup, shared code from jdgpu_prep.c / jdgpu_internal.h) against the reference's tree
 * walk (src/hufftree.c:64-78), and the unstuff stage (k_unstuff_* in jdgpu.hip) against the
 * reference's byte reader (src/bitstream.c:7-18).  The GPU suite compares the device stage's
 * output (jdgpu_batch_debug_segments) with jdgpu_host_unstuff.
 */
#include "host_hooks.h"
#include "jdgpu_internal.h"

#include <stdlib.h>
#include <string.h>

/* ---- host statement of the unstuff stage --------------------------------------------------------
 * Reference semantics (src/bitstream.c:11-14): a 0xFF that is read as data stays in the stream
 * and the byte after it is dropped whatever its value.  Restart extension: with a restart
 * interval, 0xFF 0xD0..0xD7 is a marker -- both bytes go, and the next interval starts. */
uint32_t jd_unstuff_image(const uint8_t *src, size_t raw_len, int use_rst, uint32_t n_intervals,
                          uint8_t *dst, uint32_t *si) {
    uint32_t seg = 0, ivl = 0;
    size_t o = 0, ivl_start = 0, p = 0; /* o: write offset in dst */
    for (;;) {
        int at_end = p >= raw_len;
        int marker = !at_end && use_rst && src[p] == 0xFF && p + 1 < raw_len &&
                     src[p + 1] >= 0xD0 && src[p + 1] <= 0xD7;
        /* a marker beyond the last interval the frame geometry allows ends the data: nothing
         * after it can belong to the scan (the decoder stops at the next marker, T.81 E.2.4) */
        if (marker && ivl + 1 >= n_intervals) at_end = 1;
        if (at_end || marker) {
            /* close the interval: pad to a segment boundary, emit its segment words */
            size_t len = o - ivl_start;
            uint32_t ns = (uint32_t)((len + JD_SEG_BYTES - 1) / JD_SEG_BYTES);
            if (ns == 0) ns = 1;
            for (uint32_t s = 0; s < ns; ++s) {
                size_t bits = len * 8 > (size_t)s * JD_SEG_BITS ? len * 8 - (size_t)s * JD_SEG_BITS : 0;
                si[seg + s] = (uint32_t)(bits > JD_SEG_BITS ? JD_SEG_BITS : bits) | (s == 0 ? JD_SEG_FIRST : 0);
            }
            seg += ns;
            for (size_t q = o; q < ivl_start + (size_t)ns * JD_SEG_BYTES; ++q) dst[q] = 0xFF;
            o = ivl_start + (size_t)ns * JD_SEG_BYTES;
            ivl_start = o;
            ++ivl;
            if (at_end) break;
            p += 2;
            continue;
        }
        uint8_t c = src[p++];
        dst[o++] = c;
        if (c == 0xFF) ++p; /* drop the follower */
    }
    return seg;
}

/* ---- the hooks (host_hooks.h) ------------------------------------------------------------------ */

int jdgpu_host_lut_lookup(const jdgpu_huff_spec *spec, int is_dc, unsigned v16, unsigned *len,
                          unsigned *run, unsigned *size) {
    jd_lut lut;
    if (jd_lut_build(spec, is_dc, &lut)) return -2;
    jd_sym s = jd_lut_lookup(&lut.hot, &lut, v16 & 0xFFFFu);
    *len = s.len; *run = s.run; *size = s.size;
    return s.bad ? -1 : 0;
}

long jdgpu_host_unstuff(const uint8_t *scan, size_t len, unsigned use_rst, unsigned n_intervals,
                        uint8_t *clean, size_t clean_cap, uint32_t *seg_info, size_t seg_cap) {
    size_t need = (len + JD_SEG_BYTES - 1) / JD_SEG_BYTES + (n_intervals ? n_intervals : 1);
    if (seg_cap < need || clean_cap < need * JD_SEG_BYTES) return -1;
    return (long)jd_unstuff_image(scan, len, (int)use_rst, n_intervals ? n_intervals : 1, clean, seg_info);
}


/* Builds the batch plan for the descriptors and checks the work list of k_huff_first: every image's
 * segments [0, seg_cap) are covered exactly once by waves of 64 * first_span segments, padding
 * entries carry JD_NO_WAVE.  span[i] = the image's first_span (0 for an image the plan rejects),
 * *n_waves = live entries of the list.  0 ok, -1 plan failed, -2 list broken. */
int jdgpu_host_plan_first_list(const jdgpu_image_desc *descs, size_t n, uint32_t *span, uint32_t *n_waves) {
    jd_plan P;
    char err[128];
    if (jd_plan_build(descs, n, 3u, 0u, &P, err, sizeof err) != JDGPU_OK) return -1;
    int rc = 0;
    uint32_t *next = calloc(n ? n : 1, sizeof *next), live = 0;
    if (!next) { jd_plan_free(&P); return -1; }
    for (size_t i = 0; i < n; ++i) span[i] = P.img[i].ncomp ? P.img[i].first_span : 0u;
    for (uint32_t k = 0; k < P.n_fwg * JD_WG_WAVES && rc == 0; ++k) {
        const jd_huff_wg w = P.fwg[k];
        if (w.img >= n) { rc = -2; break; }
        if (w.seg0 == JD_NO_WAVE) continue;
        ++live;
        /* the waves of an image come in order, each where the one before ended */
        if (!P.img[w.img].ncomp || w.seg0 != next[w.img] || w.seg0 >= P.img[w.img].seg_cap) rc = -2;
        next[w.img] = w.seg0 + 64u * P.img[w.img].first_span;
    }
    for (size_t i = 0; i < n && rc == 0; ++i)
        if (P.img[i].ncomp && next[i] < P.img[i].seg_cap) rc = -2;
    *n_waves = live;
    free(next);
    jd_plan_free(&P);
    return rc;
}
