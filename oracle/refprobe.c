/*
 * oracle/refprobe.c -- harness around the UNMODIFIED reference, compiled from the sources
 * where they lie (/root/reference/src, see oracle/Makefile).  TEST INFRASTRUCTURE ONLY.
 *
 * The reference exposes no coefficient-level output (decode_coefficient_matrix is static,
 * src/decoding.c:35), so this file pulls src/decoding.c in as a translation unit and replays
 * the scan loop nest of src/jpeg.c:74-87 around it.  Everything else is the reference's own
 * object code.  Output lands in oracle/_ref/ (git-ignored); it is used by pin_oracle.py and
 * gen_golden.py, never by the product.
 *
 *   refprobe <file.jpg> <out-prefix>
 *     <out-prefix>.coef   int32 dequantised natural-order [nblocks][8][8], decode order
 *     <out-prefix>.ycbcr  jpeg_read_image output into a calloc'd buffer (w*h*3)
 *     <out-prefix>.rgb    after image_ycbcr_to_rgb
 *     <out-prefix>.dim    "w h nblocks\n"
 */
#include REFERENCE_DECODING_C /* -DREFERENCE_DECODING_C='"/root/reference/src/decoding.c"' */

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int dump(const char *prefix, const char *ext, const void *p, size_t n) {
    char path[4096];
    snprintf(path, sizeof path, "%s.%s", prefix, ext);
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    size_t w = fwrite(p, 1, n, f);
    fclose(f);
    return w == n ? 0 : -1;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: refprobe file.jpg out-prefix\n"); return 2; }
    jpeg_info ji;
    scan_info si;
    memset(&ji, 0, sizeof ji);
    memset(&si, 0, sizeof si);

    /* pass 1: coefficients */
    FILE *fp = fopen(argv[1], "rb");
    if (!fp) { perror("open"); return 2; }
    if (jpeg_read_info(&ji, &si, fp)) { fprintf(stderr, "header rejected\n"); return 3; }
    unsigned mw = 8u * si.maxHSamplingFactor, mh = 8u * si.maxVSamplingFactor;
    size_t nmcu = (size_t)((ji.width + mw - 1) / mw) * ((ji.height + mh - 1) / mh);
    size_t bpm = 0;
    for (unsigned i = 0; i < si.componentCount; ++i) {
        unsigned id = si.componentInfo[i].componentID;
        bpm += (size_t)ji.componentInfo[id].HSamplingFactor * ji.componentInfo[id].VSamplingFactor;
    }
    size_t nblk = nmcu * bpm;
    int *coef = malloc(nblk * 64 * sizeof *coef);
    bitstream bs = bitstream_create(fp);
    int pred[4] = {0, 0, 0, 0};
    size_t b = 0;
    for (size_t m = 0; m < nmcu; ++m)
        for (unsigned i = 0; i < si.componentCount; ++i) {
            unsigned id = si.componentInfo[i].componentID;
            unsigned nb = ji.componentInfo[id].HSamplingFactor * ji.componentInfo[id].VSamplingFactor;
            for (unsigned k = 0; k < nb; ++k, ++b)
                decode_coefficient_matrix((int(*)[8])(coef + 64 * b), &ji, si.componentInfo[i], &bs, pred + i);
        }
    fclose(fp);
    if (dump(argv[2], "coef", coef, nblk * 64 * sizeof *coef)) return 4;
    free(coef);
    jpeg_free(&ji);

    /* pass 2: the public API end to end */
    fp = fopen(argv[1], "rb");
    memset(&ji, 0, sizeof ji);
    memset(&si, 0, sizeof si);
    if (jpeg_read_info(&ji, &si, fp)) return 3;
    size_t npx = (size_t)ji.width * ji.height;
    pixel *im = calloc(npx, sizeof *im);
    jpeg_read_image(im, &ji, &si, fp);
    fclose(fp);
    if (dump(argv[2], "ycbcr", im, npx * 3)) return 4;
    image_ycbcr_to_rgb(im, ji.width, ji.height);
    if (dump(argv[2], "rgb", im, npx * 3)) return 4;
    char dim[64];
    int n = snprintf(dim, sizeof dim, "%u %u %zu\n", ji.width, ji.height, nblk);
    if (dump(argv[2], "dim", dim, (size_t)n)) return 4;
    free(im);
    jpeg_free(&ji);
    return 0;
}
