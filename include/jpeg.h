/*
 * include/jpeg.h -- drop-in for the reference's src/jpeg.h (cefqrn/jpeg-decoder).
 *
 * Keeps the reference's public types and the three entry points of src/jpeg.h:31-34 so that
 * src/test.c (or any caller of the reference) builds and links unchanged against
 * libjpegamd.so.  Field names, order and offsets of jpeg_info / scan_info / huffnode are the
 * reference's (src/jpeg.h:7-29, src/hufftree.h:15-22): sizeof(jpeg_info) == 1224,
 * offsetof quantizationTables == 768, huffmanTables == 1024, width == 1216;
 * sizeof(scan_info) == 15.
 *
 * Differences, all additive:
 *   - restartInterval sits in what is tail padding in the reference's jpeg_info (offset 1222),
 *     so size and every existing offset are unchanged.  The reference ignores DRI
 *     (src/jpeg.c:50); jpeg_read_info here records it and jpeg_read_image honours it.
 *   - jpeg_read_info zeroes *jpegInfo first, so jpeg_free on the failure path (src/test.c:44-45)
 *     never walks uninitialised pointers.
 *   - jpeg_read_image decodes on the MI355X (see include/jpeg_gpu.h); it reads the rest of
 *     `fp` to EOF instead of byte-by-byte.  Fatal conditions print the reference's messages
 *     (src/bitstream.c:16-17, src/hufftree.c:69,72) and exit(EXIT_FAILURE) like src/macros.h:8.
 */
#ifndef JPEG_H
#define JPEG_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "hufftree.h" /* src/jpeg.h:4: huffnode, and with it bitstream */
#include "image.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    struct jpeg_component_info {
        unsigned char HSamplingFactor;
        unsigned char VSamplingFactor;
        unsigned char quantizationTableID;
    } componentInfo[256];                      /* indexed by component id */
    unsigned char  quantizationTables[4][64];  /* file (zigzag) order */
    huffnode       huffmanTables[2][4];        /* [class][id] roots */
    unsigned short width;
    unsigned short height;
    unsigned char  precision;                  /* only 8 is accepted */
    unsigned short restartInterval;            /* MCUs per restart interval, 0 = none (extension) */
} jpeg_info;

typedef struct {
    struct scan_component_info {
        unsigned char componentID;
        unsigned char DCHuffmanTableID;
        unsigned char ACHuffmanTableID;
    } componentInfo[4];                        /* scan order */
    unsigned char maxHSamplingFactor;
    unsigned char maxVSamplingFactor;
    unsigned char componentCount;
} scan_info;

/* src/jpeg.c:23-58: 0 on success, -1 on a malformed or unsupported header; on return fp is at
 * the first entropy-coded byte. */
int  jpeg_read_info(jpeg_info *jpegInfo, scan_info *scanInfo, FILE *fp);

/* src/jpeg.c:63-88: fills img (width*height pixels, YCbCr). */
void jpeg_read_image(pixel *img, const jpeg_info *info, const scan_info *scanInfo, FILE *fp);

/* src/jpeg.c:93-99 (frees every table, not just ids 0-1) */
void jpeg_free(jpeg_info *info);

#ifdef __cplusplus
}
#endif
#endif
