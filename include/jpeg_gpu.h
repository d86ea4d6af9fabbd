/*
 * include/jpeg_gpu.h -- C ABI of the MI355X decode backend (libjpegamd.so).
 *
 * This is the boundary SURVEY.md section 8(b) defines: plain pointers and sizes, int status
 * returns, no C++ types, no exit() inside.  jpeg_read_image() / image_ycbcr_to_rgb() of
 * include/jpeg.h / include/image.h are a batch of one routed through these entry points; the
 * batched calls are the throughput product.  Each declaration cites the reference code it
 * replaces (paths relative to /root/reference/).
 */
#ifndef JPEG_GPU_H
#define JPEG_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-image descriptor: everything jpeg_read_image gets from jpeg_info + scan_info + fp
 *      (src/jpeg.c:63-88), as flat data -------------------------------------------------- */

typedef struct {
    uint8_t counts[16];   /* codes of length 1..16: the DHT payload src/hufftree.c:43-62 parses */
    uint8_t symbols[256]; /* symbols in code order */
} jdgpu_huff_spec;

typedef struct {
    const uint8_t *scan;  /* entropy-coded bytes, byte-stuffed, from the byte after SOS to EOF:
                             what src/bitstream.c:7-18 pulls through getc */
    size_t   scan_len;
    uint16_t width, height;          /* src/jpeg.h:15-16 */
    uint8_t  ncomp;                  /* scan_info.componentCount, 1..3 */
    struct {
        uint8_t id;                  /* component id; output channel = id-1 (src/decoding.c:107) */
        uint8_t H, V, Tq;            /* src/jpeg.h:8-12 */
        uint8_t Td, Ta;              /* src/jpeg.h:21-25 */
    } comp[4];                       /* scan order (src/jpeg.c:76) */
    uint8_t  quant[4][64];           /* zigzag order, src/jpeg.h:13 */
    jdgpu_huff_spec huff[2][4];      /* [class][id], src/jpeg.h:14 */
    uint8_t  huff_defined[2][4];
    uint16_t restart_interval;       /* MCUs; 0 = none.  The reference ignores DRI (src/jpeg.c:50) */
} jdgpu_image_desc;

/* ---- status codes ---------------------------------------------------------------------- */

enum {
    JDGPU_OK = 0,
    JDGPU_ERR_HEADER = -1,       /* jpeg_read_info's -1 (src/jpeg.c:26-54, src/parsing.c) */
    JDGPU_ERR_NO_DEVICE = -10,   /* no usable HIP device: the product has no CPU fallback */
    JDGPU_ERR_HIP = -11,         /* a HIP runtime call failed (jdgpu_last_error() has the text) */
    JDGPU_ERR_ARG = -12,
    JDGPU_ERR_NOMEM = -13,
    JDGPU_ERR_UNSUPPORTED = -14, /* sampling factors / component ids outside SURVEY 8(c) contract */
    JDGPU_ERR_NOT_CONVERGED = -15
};

/* per-image status bits written by the kernels (jdgpu_batch_status) */
enum {
    JDGPU_IMG_EOF = 1,        /* "Reached end of file."                    src/bitstream.c:17 */
    JDGPU_IMG_NOSYMBOL = 2,   /* "Could not find next symbol in huffnode." src/hufftree.c:69,72 */
    JDGPU_IMG_UNSUPPORTED = 4,
    JDGPU_IMG_UNSYNCED = 8,   /* internal: segment chain not yet consistent (retried by decode) */
    /* a DC value left the 16 bits the coefficient buffer holds.  The reference accumulates the
     * predictor in an int and goes on (src/decoding.c:44-46); no stream of 8-bit samples gets there
     * (|DC| <= 1024 * 8 / Q), so this backend reports the image instead of wrapping the value:
     * pixels all zero, like the two failures above. */
    JDGPU_IMG_RANGE = 16
};

/* flags for jdgpu_batch_create */
enum {
    JDGPU_OUT_RGB = 0,        /* fused YCbCr->RGB store: jpeg_read_image + image_ycbcr_to_rgb */
    JDGPU_OUT_YCBCR = 1,      /* what jpeg_read_image alone leaves in the buffer */
    JDGPU_PROFILE = 2,        /* record HIP events around every stage of each decode */
    /* 4: unused (a debug flag of round 1 that unstuffed on the host; the stage has one
     * implementation now, on the GPU) */
    /* output layouts other than pixel[] of src/image.h:4-12 (row f3 of SURVEY 8: the 3-byte
     * interleaved layout is what the reference's callers get; loaders want one of these).  Same
     * sample values, rows without padding; combine with JDGPU_OUT_RGB or JDGPU_OUT_YCBCR. */
    JDGPU_OUT_4BYTE = 8,      /* 4 interleaved bytes per pixel, the fourth = 255 (RGBA / YCbCrA) */
    JDGPU_OUT_PLANAR = 16,    /* three planes of width*height bytes, in channel order ([3][H][W]) */
    /* padded pitch (row f3): every output row starts on a 256-byte boundary.  The row stride is the
     * width rounded up to 256 pixels (3-byte pixels, planes) or 64 pixels (4-byte pixels);
     * jdgpu_batch_pixel_pitch tells it in bytes.  Padding bytes are zero and are never written by a
     * decode.  Combines with every layout above; rows of any width are then stored with whole
     * dwords (the unpadded 3-byte layout goes byte by byte when width % 4 != 0). */
    JDGPU_OUT_PITCHED = 32,
    /* the descriptors' `scan` pointers are DEVICE addresses (what jdgpu_parse_files_device leaves in
     * them): the scans are copied on the device, nothing is staged or uploaded.  The memory they
     * point into may be released when jdgpu_batch_create returns. */
    JDGPU_SCAN_ON_DEVICE = 64
};

/* stages reported by jdgpu_batch_stage_ms */
enum {
    JDGPU_STAGE_UNSTUFF = 0,  /* replaces src/bitstream.c:7-18 */
    JDGPU_STAGE_SYNC = 1,     /* start state of every segment: speculative decode, chain check, repair rounds
                                 (no counterpart: src/jpeg.c:74-87 is serial) */
    JDGPU_STAGE_SCAN = 2,     /* block offsets + DC prediction: src/jpeg.c:73, src/decoding.c:44 */
    JDGPU_STAGE_WRITE = 3,    /* coefficient write: src/decoding.c:40-66 */
    JDGPU_STAGE_PIXEL = 4,    /* src/decoding.c:46,65,68-110 + src/image.c:39-51 */
    JDGPU_STAGE_TOTAL = 5,
    JDGPU_NSTAGES = 6
};

/* ---- device ---------------------------------------------------------------------------- */

int         jdgpu_device_count(void);
/* Bind the calling process to a device; idempotent for the same device.  ONE process drives ONE
 * GPU (images are sharded over processes, one per GPU): the buffer cache, the pinned
 * staging buffer and the device constants are process-wide, so once a device is bound a call with
 * another one returns JDGPU_ERR_ARG and changes nothing.  Entry points that need a device and find
 * none bound bind device 0.  No counterpart in the reference (CPU only). */
int         jdgpu_init(int device);
const char *jdgpu_last_error(void);          /* thread-local text of the last failure */
/* Device buffers of destroyed batches are kept for the next jdgpu_batch_create (allocating and
 * freeing tens of GB per batch would stall a caller that creates one batch while another
 * decodes); this releases what is cached.  Bound: JDGPU_POOL_MB (default 131072, 0 = no cache).
 * No counterpart in the reference (its only buffer is the caller's malloc, src/test.c:28). */
void        jdgpu_trim(void);

/* ---- host-side header parsing on a whole file in memory ---------------------------------
 * jpeg_read_info (src/jpeg.c:23-58) over a buffer, flattened into a descriptor whose `scan`
 * points into `file`.  0 or JDGPU_ERR_HEADER / JDGPU_ERR_UNSUPPORTED. */
int jdgpu_parse_file(const uint8_t *file, size_t len, jdgpu_image_desc *desc);

/* The same over n files on `threads` host threads (0 = one per available core, at most 16).
 * status (may be NULL) receives the per-file return code; a rejected file leaves a zeroed
 * descriptor, which jdgpu_batch_create carries as JDGPU_IMG_UNSUPPORTED without decoding it.
 * Returns the number of rejected files, or JDGPU_ERR_ARG.  (Row f1 of SURVEY 8: the header loop
 * of src/jpeg.c:23-58 is serial per file and independent between files.) */
long jdgpu_parse_files(const uint8_t *const *files, const size_t *lens, size_t n,
                       jdgpu_image_desc *descs, int *status, unsigned threads);

/* Device-side batched header parsing (row f1 of SURVEY 8): the same marker loop (src/jpeg.c:23-58,
 * src/parsing.c:13-142) run by a kernel, one lane per file, over n whole files that lie in DEVICE
 * memory: file i = d_files[offsets[i] .. offsets[i+1]) (offsets: host array of n + 1).  descs and
 * status are host arrays; the descriptors are byte for byte those of jdgpu_parse_file except that
 * `scan` is a device address into d_files: hand them to jdgpu_batch_create with
 * JDGPU_SCAN_ON_DEVICE.  Rejected files: as in jdgpu_parse_files.  One case is not decided by the
 * kernel: a DHT that assigns more codes than its lengths can hold (the reference's tree builder
 * wraps and overwrites, src/hufftree.c:46-59; no encoder writes such a table) -- that file is
 * copied back and goes through jdgpu_parse_file, so the result is the same.  Returns the number
 * of rejected files or a negative JDGPU_ERR_*.
 * Padding: the scan gather of jdgpu_batch_create(JDGPU_SCAN_ON_DEVICE) reads whole aligned 32-bit
 * words, i.e. up to 3 bytes beyond the end of a scan (never beyond the 4-byte boundary that follows
 * it): d_files must be readable up to the next multiple of 4 after offsets[n].  Any hipMalloc'd
 * buffer is (allocations are whole 4 KiB pages); a sub-allocator that places the files at the very
 * end of a mapping must round its size up to 4. */
long jdgpu_parse_files_device(const uint8_t *d_files, const uint64_t *offsets, size_t n,
                              jdgpu_image_desc *descs, int *status);

/* Pinned host memory for the files of coming batches (no counterpart in the reference, which reads one
 * file through getc, src/bitstream.c:12-14): a loader that reads its files into such a region spares every
 * batch creation the copy into the library's staging buffer.  jdgpu_batch_create and
 * jdgpu_batch_create_from_files notice by themselves that all scans / files of a batch lie in one such
 * region and upload the span that holds them in one DMA from where it lies (a batch that uses less than
 * half of the bytes between its first and last file goes through the staging buffer as before).  The
 * region must stay as it is until the creation call returns.  NULL when the allocation fails. */
void *jdgpu_pinned_alloc(size_t bytes);
void  jdgpu_pinned_free(void *p);

/* ---- batched decode: replaces jpeg_read_image (src/jpeg.c:63-88) and, with JDGPU_OUT_RGB,
 *      image_ycbcr_to_rgb (src/image.c:39-51), for n independent images ---------------------- */

typedef struct jdgpu_batch jdgpu_batch;

/* Flattens Huffman tables to lookup tables (replaces the tree walk of src/hufftree.c:64-78),
 * uploads scan bytes + tables into HBM and allocates coefficient and pixel buffers.  The
 * descriptors and their scan pointers may be released after this returns. */
int  jdgpu_batch_create(const jdgpu_image_desc *descs, size_t n, unsigned flags, jdgpu_batch **out);
/* From whole files in host memory to a batch without the host looking inside them: the files go
 * through the pinned staging buffer in one upload, the headers are parsed on the device
 * (jdgpu_parse_files_device), the scans are gathered on the device.  status (may be NULL):
 * per-file JDGPU_OK / JDGPU_ERR_HEADER; a rejected file is carried as JDGPU_IMG_UNSUPPORTED.
 * Replaces the caller-side sequence jpeg_read_info + jpeg_read_image of src/test.c:23-32 for n files. */
int  jdgpu_batch_create_from_files(const uint8_t *const *files, const size_t *lens, size_t n, unsigned flags,
                                   jdgpu_batch **out, int *status);
/* Enqueue the whole decode (unstuff, sync, scan, write, pixel kernels) on the batch's stream.
 * Inputs are already resident in HBM; may be called repeatedly. */
int  jdgpu_batch_decode(jdgpu_batch *b);
/* Wait for the stream; returns JDGPU_OK or the first failure (per-image detail: _status).
 * An image whose status has JDGPU_IMG_EOF or JDGPU_IMG_NOSYMBOL (the reference exits there,
 * src/macros.h:8) or JDGPU_IMG_RANGE has all-zero pixels; its coefficients are undefined. */
int  jdgpu_batch_sync(jdgpu_batch *b);
void jdgpu_batch_destroy(jdgpu_batch *b);

size_t   jdgpu_batch_size(const jdgpu_batch *b);
uint32_t jdgpu_batch_status(const jdgpu_batch *b, size_t i);        /* JDGPU_IMG_* bits */
uint64_t jdgpu_batch_total_pixels(const jdgpu_batch *b);            /* sum of width*height */
uint64_t jdgpu_batch_total_blocks(const jdgpu_batch *b);            /* 8x8 data units */
uint64_t jdgpu_batch_total_scan_bytes(const jdgpu_batch *b);
size_t   jdgpu_batch_image_blocks(const jdgpu_batch *b, size_t i);
/* width and height of image i as the header gave them (src/jpeg.h:15-16); 0 x 0 for a rejected file */
int      jdgpu_batch_image_size(const jdgpu_batch *b, size_t i, unsigned *width, unsigned *height);
/* width*height*3 bytes, layout of `pixel[]` (src/image.h:4-14); with JDGPU_OUT_4BYTE,
 * JDGPU_OUT_PLANAR or JDGPU_OUT_PITCHED the layout chosen at creation, padding included
 * (jdgpu_batch_pixel_bytes tells the size) */
int      jdgpu_batch_read_pixels(jdgpu_batch *b, size_t i, void *dst);
size_t   jdgpu_batch_pixel_bytes(const jdgpu_batch *b, size_t i);
/* bytes from one row to the next (of a plane, with JDGPU_OUT_PLANAR: the planes are then
 * pitch * height bytes apart): width * bytes per pixel, or the padded pitch with JDGPU_OUT_PITCHED.
 * pixel_bytes = pitch * height (* 3 for planes).  src/image.h:4-14 has no pitch: rows are dense. */
size_t   jdgpu_batch_pixel_pitch(const jdgpu_batch *b, size_t i);
/* device pointer, same layout; image i starts on a 256-byte boundary, and images of a batch lie
 * one after the other -- a batch of equal-sized images whose size is a multiple of 256 bytes
 * (1920x1080: yes) is one dense [n][...] array starting at image 0 */
void    *jdgpu_batch_device_pixels(jdgpu_batch *b, size_t i);
/* Quantised coefficients, [nblocks][64] int16 in ZIGZAG order, DC un-differenced, decode order
 * (the values src/decoding.c:44,65 computes before the multiply by the quantisation table).
 * Whatever form the device keeps them in (64-byte units of int8 for images without large AC
 * values, 128-byte units of int16 otherwise) is expanded here. */
int      jdgpu_batch_read_coefficients(jdgpu_batch *b, size_t i, int16_t *dst);
/* HIP-event milliseconds per stage of the most recent decode (needs JDGPU_PROFILE). */
int      jdgpu_batch_stage_ms(jdgpu_batch *b, float ms[JDGPU_NSTAGES]);
/* Where the creation of the batch spent its host time, in milliseconds: building the plan; copying scans or
 * files into the library's pinned staging buffer (0 when they came from jdgpu_pinned_alloc memory or from the
 * device); the uploads (scans / files, tables, work lists) and the device-side gather; everything else
 * (allocations, header parse on the device and its read-back). */
enum { JDGPU_CREATE_PLAN = 0, JDGPU_CREATE_STAGE = 1, JDGPU_CREATE_UPLOAD = 2, JDGPU_CREATE_OTHER = 3, JDGPU_NCREATE = 4 };
int      jdgpu_batch_create_ms(const jdgpu_batch *b, float ms[JDGPU_NCREATE]);
/* number of extra passes the last decode needed: 0 in the common case; each one costs a host round
 * trip, the repair rounds again and a scan/write/pixel pass over the images that were left
 * unfinished (a segment chain still broken, a coefficient beyond the 64-byte unit form) -- the
 * other images of the batch are not touched again */
int      jdgpu_batch_retries(const jdgpu_batch *b);
/* development aid: how many segments the start-state repair listed in each of its rounds during
 * the last decode pass (summed over the images); rounds beyond the built-in number read 0 */
int      jdgpu_batch_repair_counts(jdgpu_batch *b, uint32_t *out, size_t rounds);

/* bytes per bit-segment (one lane decodes one segment) */
unsigned jdgpu_segment_bytes(void);

/* ---- debug read-back for the tests ------------------------------------------------------------ */
/* the clean (unstuffed, interval-padded) stream and per-segment words the unstuff stage left on
 * the device for image i: each restart interval padded with 0xFF to a whole segment, segments one
 * after the other; seg_info[i]: bit 31 = first segment of an interval, low 16 bits = valid bits in
 * the segment (the format of the tests' host statement, csrc/host_hooks.h).  Returns the segment count. */
long jdgpu_batch_debug_states(jdgpu_batch *b, size_t i, uint32_t *st_in, uint32_t *st_exit, size_t cap);
long jdgpu_batch_debug_segments(jdgpu_batch *b, size_t i, uint8_t *clean, size_t clean_cap,
                                uint32_t *seg_info, size_t seg_cap);

/* ---- image.h on the device: src/image.c:39-51 over a host buffer (H2D, kernel, D2H) ------ */
int jdgpu_ycbcr_to_rgb(void *host_pixels, unsigned width, unsigned height);
/* The same without any allocation.  NO COUNTERPART IN THE REFERENCE: it is for callers that cannot
 * allocate device memory where they call -- tight serving loops, harnesses that forbid
 * hipMalloc, code that must not stall on the allocator.  jdgpu_ycbcr_to_rgb stages the pixels in
 * a device buffer it allocates and frees per call; this variant stages them in a static device
 * array of 32 MiB that the library holds on its GPU from load to process end, so it calls no
 * hipMalloc / hipFree (nor any other allocating routine).  Limit: width * height * 3 <= 32 MiB
 * (11 184 810 pixels; 3840 x 2160 fits), larger inputs return JDGPU_ERR_ARG and leave the buffer
 * untouched.  All calls of a process share the one array: they are serialised by a lock inside
 * the library (concurrent callers wait, they do not race); the array keeps the previous call's
 * bytes, which never matter -- a call reads only what it has just uploaded. */
int jdgpu_ycbcr_to_rgb_noalloc(void *host_pixels, unsigned width, unsigned height);
/* the limit above, in bytes (no reference counterpart either): image_ycbcr_to_rgb of include/image.h
 * (src/image.c:39-51) stages through the static array whenever width * height * 3 fits */
size_t jdgpu_noalloc_max_bytes(void);
/* src/image.c:54-66 the same way (row f4 of SURVEY 8; off the decode path: image_rgb_to_ycbcr of
 * include/image.h stays host C, this is for callers whose pixels are large) */
int jdgpu_rgb_to_ycbcr(void *host_pixels, unsigned width, unsigned height);

/* The text image_print writes to stdout (src/image.c:13-31), formatted on the device into `out`
 * (row f4 of SURVEY 8; off the decode path: image_print of include/image.h stays host C, this is
 * for prints of many cells): per printed pixel ESC[48;2;R;G;Bm + pixelWidth spaces, per row
 * ESC[0m + newline; rows = min(height, maxPrintHeight), columns while x * pixelWidth <
 * maxPrintWidth.  *len receives the text's size; out == NULL only queries it; a buffer smaller
 * than the text returns JDGPU_ERR_ARG and leaves it untouched.  No terminating NUL. */
int jdgpu_image_format(const void *host_pixels, unsigned width, unsigned height, unsigned maxPrintWidth,
                       unsigned maxPrintHeight, unsigned pixelWidth, char *out, size_t cap, size_t *len);

#ifdef __cplusplus
}
#endif
#endif
