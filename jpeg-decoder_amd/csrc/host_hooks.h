/*
 * host_hooks.h -- test-only hooks of libjdhooks.so (see host_hooks.c); not part of the C ABI of
 * include/jpeg_gpu.h and not exported by libjpegamd.so.
 */
#ifndef JD_HOST_HOOKS_H
#define JD_HOST_HOOKS_H

#include <stddef.h>
#include <stdint.h>

#include "jpeg_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* decode the code at the top of the 16-bit window v16 with the flattened tables the kernels use:
 * 0 ok, -1 no such code, -2 bad table */
int  jdgpu_host_lut_lookup(const jdgpu_huff_spec *spec, int is_dc, unsigned v16, unsigned *len,
                           unsigned *run, unsigned *size);
/* returns the number of segments written (each restart interval padded with 0xFF to a whole
 * segment, segments stored one after the other), -1 if the buffers are too small.
 * seg_info[i]: bit 31 = first segment of an interval, low 16 bits = valid bits in the segment. */
long jdgpu_host_unstuff(const uint8_t *scan, size_t len, unsigned use_rst, unsigned n_intervals,
                        uint8_t *clean, size_t clean_cap, uint32_t *seg_info, size_t seg_cap);

/* batch plan of the descriptors: checks the work list of k_huff_first (every segment of every image
 * covered once by waves of 64 * span segments); span[i] = the span the plan gave image i,
 * *n_waves = waves on the list.  0 ok, -1 no plan, -2 list broken */
int  jdgpu_host_plan_first_list(const jdgpu_image_desc *descs, size_t n, uint32_t *span, uint32_t *n_waves);

#ifdef __cplusplus
}
#endif
#endif
