/*
 * oracle/refbench.c -- in-memory entry to the UNMODIFIED reference (linked into
 * oracle/_ref/libjpegref.so next to the reference's own objects).  TEST INFRASTRUCTURE ONLY:
 * used by bench.py's cpu_baseline leg (kind "reference") and by tests as a checker.
 *
 * The reference API takes FILE* (src/jpeg.h:31-32); fmemopen gives it one over a buffer.
 * Calls exactly what src/test.c:23,34,36,40 calls.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <string.h>

#include "jpeg.h"
#include "image.h"

/* returns 0 on success; out must hold w*h*3 bytes (pass NULL to only query the size).
 * mode 0 = YCbCr (jpeg_read_image only), 1 = RGB (+ image_ycbcr_to_rgb). */
int refbench_decode(const unsigned char *file, size_t len, unsigned char *out, unsigned *w,
                    unsigned *h, int mode) {
    FILE *fp = fmemopen((void *)file, len, "rb");
    if (!fp) return -2;
    jpeg_info ji;
    scan_info si;
    memset(&ji, 0, sizeof ji);
    memset(&si, 0, sizeof si);
    if (jpeg_read_info(&ji, &si, fp)) { fclose(fp); return -1; }
    *w = ji.width;
    *h = ji.height;
    if (out) {
        jpeg_read_image((pixel *)out, &ji, &si, fp);
        if (mode) image_ycbcr_to_rgb((pixel *)out, ji.width, ji.height);
    }
    jpeg_free(&ji);
    fclose(fp);
    return 0;
}
