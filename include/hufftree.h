/*
 * include/hufftree.h -- the reference's src/hufftree.h:10-27, included by its jpeg.h:4.
 *
 * huffnode is part of jpeg_info's layout (src/jpeg.h:14) and stays byte-compatible.
 * hufftree_parse / hufftree_destroy are host-side header work and are exported by libjpegamd.so
 * (jpeg_read_info builds the same trees).  hufftree_decode_next_symbol is the reference's
 * bit-by-bit tree walk (src/hufftree.c:64-78): the GPU path replaces it with flattened lookup
 * tables, and the library does not define it -- see include/bitstream.h.
 */
#ifndef HUFFTREE_H
#define HUFFTREE_H

#include "bitstream.h"

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/hufftree.h:10-13 */
typedef enum HuffTableClass { CLASS_DC = 0, CLASS_AC = 1 } HuffTableClass;

/* src/hufftree.h:15-22: binary code tree, left = 0 bit, right = 1 bit.  The root lives inside
 * jpeg_info; children are heap nodes owned by jpeg_info until jpeg_free. */
typedef struct huffnode {
    struct huffnode *left;
    struct huffnode *right;
    uint8_t symbol;
    bool hasLeft;
    bool hasRight;
    bool hasSymbol;
} huffnode;

/* src/hufftree.c:43-62: canonical code assignment from a DHT payload (16 counts + symbols);
 * returns the bytes consumed.  *buf is overwritten without being read. */
size_t hufftree_parse(huffnode *buf, const unsigned char *data);
/* src/hufftree.c:87-90: frees the children (the root belongs to the caller) */
void hufftree_destroy(huffnode *tree);

/* src/hufftree.h:27 -- declared for source compatibility only, not exported */
uint8_t hufftree_decode_next_symbol(const huffnode *tree, bitstream *str);

#ifdef __cplusplus
}
#endif
#endif
