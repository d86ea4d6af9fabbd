/*
 * include/image.h -- drop-in for the reference's src/image.h (cefqrn/jpeg-decoder).
 *
 * Same type, same macro, same three prototypes (src/image.h:4-19), so code written against
 * the reference compiles unchanged.  `pixel` stays a 3-byte union: sizeof(pixel) == 3, rows
 * are width*3 bytes with no padding (src/image.h:14, src/decoding.c:107).
 *
 * image_ycbcr_to_rgb runs on the MI355X (HIP kernel in csrc/jdgpu.hip) and reproduces the
 * double-precision, truncating arithmetic of src/image.c:39-51 exactly.  image_print and
 * image_rgb_to_ycbcr are host C (off the decode path).
 */
#ifndef IMAGE_H
#define IMAGE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef union {
    struct { unsigned char Y, Cb, Cr; } YCbCr;
    struct { unsigned char R, G, B; } RGB;
    unsigned char data[3];
} pixel;

/* bytes needed for a width x height image (src/image.h:14) */
#define image_size(width, height) ((width) * (height) * sizeof(pixel))

/* ANSI 24-bit colour dump, src/image.c:13-31 */
void image_print(const pixel *img, unsigned imageWidth, unsigned imageHeight,
                 unsigned maxPrintWidth, unsigned maxPrintHeight, unsigned pixelWidth);

/* in-place colour transforms, src/image.c:39-51 and :54-66 */
void image_ycbcr_to_rgb(pixel *img, unsigned width, unsigned height);
void image_rgb_to_ycbcr(pixel *img, unsigned width, unsigned height);

#ifdef __cplusplus
}
#endif
#endif
