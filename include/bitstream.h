/*
 * include/bitstream.h -- the reference's src/bitstream.h:7-17, visible to its callers through
 * jpeg.h:4 -> hufftree.h:4.  Kept so that code which names the type keeps compiling.
 *
 * The MI355X path does not use it: the byte reader with its drop-the-byte-after-0xFF rule
 * (src/bitstream.c:7-18) is the unstuff stage on the GPU, the bit extraction (src/bitstream.c:28-45)
 * lives in the Huffman kernels' register windows.  The three functions below are the reference's
 * serial decode primitives; libjpegamd.so deliberately does NOT define them (there is no CPU
 * decode path in the product), so a program that calls them fails at link time instead of
 * silently decoding on the host.
 */
#ifndef BITSTREAM_H
#define BITSTREAM_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/bitstream.h:7-11 */
typedef struct {
    FILE          *_fp;
    unsigned       _bitsLeft;
    unsigned char  _c;
} bitstream;

/* src/bitstream.h:13-16 -- declared for source compatibility only, not exported (see above) */
bitstream     bitstream_create(FILE *fp);
unsigned      bitstream_get_bit(bitstream *str);
uint_fast32_t bitstream_get_bits(bitstream *str, unsigned count);

#ifdef __cplusplus
}
#endif
#endif
