/*
 * jdgpu.hip -- MI355X (gfx950, wave64) baseline-JPEG decode kernels and the C ABI of
 * include/jpeg_gpu.h.  Written for CDNA4 only: no CUDA paths, no portability layer.
 *
 * Pipeline of one jdgpu_batch_decode (all on one HIP stream, inputs resident in HBM):
 *
 *   unstuff   k_unstuff_*      stuffed scan bytes -> clean stream, one restart interval after
 *                              the other, each padded to whole 4096-bit segments
 *                              (replaces src/bitstream.c:7-18 of the reference)
 *   sync      k_huff_first     one lane per segment: guess the start state from the tail of
 *                              the predecessor segment (self-synchronisation), decode
 *             k_chain_check    list the segments whose start state is not their predecessor's
 *             k_huff_repair    exit state and decode those again, a fixed number of rounds
 *   scan      k_seg_scan       segmented prefix sums: first data-unit index and DC predictors
 *                              at every segment start (src/jpeg.c:73, src/decoding.c:44)
 *   write     k_huff_write     final decode from the now-known start states; writes quantised
 *                              int16 coefficients (src/decoding.c:40-66)
 *   pixel     k_pixel          dequantise, separable fp64 IDCT through LDS, level shift, clamp,
 *                              replicate chroma, YCbCr->RGB, coalesced 12-byte stores
 *                              (src/decoding.c:46,65,68-110, src/image.c:39-51); one
 *                              instantiation per chroma layout and output layout
 *
 * The Huffman kernels take one (image, first segment) descriptor per WAVE; the plan
 * (jdgpu_prep.c) lets waves of different images share a work-group when the images' flattened
 * tables are identical.  Host side, at the end of the file: the C ABI, a cache of device buffers
 * (batches are created while others decode), the pinned staging buffer for uploads.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <thread>
#include <vector>

#include "jdgpu_internal.h"

/* ============================================================================================
 * small helpers
 * ========================================================================================== */

static thread_local char g_err[512] = "";

#define HIPCHK(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            snprintf(g_err, sizeof g_err, "%s:%d %s -> %s", __FILE__, __LINE__, #call,            \
                     hipGetErrorString(e_));                                                      \
            return JDGPU_ERR_HIP;                                                                 \
        }                                                                                         \
    } while (0)

#define SEG_MASK 0xFFFFu /* valid-bit count in a per-segment word */
#define JD_REPAIR_ROUNDS 10u /* chain check + repair launches per pass */

/* The clean stream is stored transposed in groups of 64 segments, 16 bytes at a time: bytes
 * [16q, 16q+16) of segment s live at [(s / 64) * (JD_SEG_BYTES/16) + q][s % 64], so the 64 lanes
 * of a wave -- one segment each, all roughly equally far into their segments -- read from the
 * same few cache lines, and the unstuff stage can still store 16 bytes at a time. */
__host__ __device__ __forceinline__ size_t clean_word_index(uint32_t seg, uint32_t w) {
    return (((size_t)(seg >> 6) * (JD_SEG_WORDS / 4u) + (w >> 2)) * 64u + (seg & 63u)) * 4u + (w & 3u);
}
__host__ __device__ __forceinline__ size_t clean_byte_addr(uint32_t seg, uint32_t o) {
    return clean_word_index(seg, o >> 2) * 4u + (o & 3u);
}

/* ============================================================================================
 * unstuff: stuffed scan bytes -> clean, interval-padded stream (replaces src/bitstream.c:7-18)
 *
 * Reference semantics: a 0xFF read as data stays, the byte after it is dropped whatever it is.
 * So a byte is a dropped "follower" iff the run of 0xFF bytes right before it has odd length.
 * Restart extension: with a restart interval, an active 0xFF followed by 0xD0..0xD7 is a marker:
 * both bytes go and the next interval starts; the marker after the last interval the frame
 * geometry allows ends the data.
 *
 *   k_unstuff_count    per 4 KiB chunk: kept bytes, markers
 *   k_unstuff_chunks   per image: exclusive scan of the chunk counts
 *   k_unstuff_marks    per chunk: clean offset of every interval start (restart images only)
 *   k_unstuff_layout   per image: segments per interval, per-segment words, tail padding
 *   k_unstuff_scatter  per chunk: kept bytes to their padded position
 * ========================================================================================== */

struct RawLane {
    uint32_t w[4];       /* the lane's 16 bytes, little endian */
    uint32_t nx;         /* the byte after them (0 past the end of the scan) */
    uint32_t n;          /* how many of the 16 exist */
    uint32_t d0;         /* 1 = the first byte is a dropped follower */
};

/* the lane's 16 bytes (raw regions are 16-byte aligned and zero padded) */
__device__ __forceinline__ uint4 raw_lane_fetch(const uint8_t *raw, uint32_t raw_len, uint32_t i0) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (i0 < raw_len) v = *reinterpret_cast<const uint4 *>(raw + i0);
    return v;
}
/* called by all 256 lanes of a work-group, i0 = 16 bytes per lane in lane order: the byte after a
 * lane's 16 and the byte before them come from the neighbouring lanes, from memory only at the
 * ends of a wave */
__device__ __forceinline__ RawLane raw_lane_load(const uint4 v, const uint8_t *raw, uint32_t raw_len, uint32_t i0) {
    RawLane r;
    r.n = i0 < raw_len ? min(16u, raw_len - i0) : 0u;
    r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t nx = __shfl_down(v.x & 255u, 1);     /* a lane past the end of the scan holds zeros */
    uint32_t pv = __shfl_up(v.w >> 24, 1);
    if (lane == 63u) nx = (r.n && i0 + 16 < raw_len) ? raw[i0 + 16] : 0u;
    if (lane == 0u) pv = (r.n && i0 > 0) ? raw[i0 - 1] : 0u;
    r.nx = nx;
    /* parity of the 0xFF run that ends right before i0 (bounded walk: >=4 consecutive 0xFF give
     * 16 one-bits in the clean stream, which no Huffman table can decode, so the image fails
     * either way and the parity of a longer run cannot matter) */
    uint32_t run = 0;
    if (r.n && i0 > 0 && pv == 0xFFu) {
        run = 1;
        while (run < 1024u && run < i0 && raw[i0 - 1 - run] == 0xFF) ++run;
    }
    r.d0 = run & 1u;
    return r;
}

/* bit 7 of every byte of x that is 0xFF / that is 0xD0..0xD7 (no carry crosses a byte), and those
 * four bits gathered into bits 0..3 (the products land on distinct bits: no carry either) */
__device__ __forceinline__ uint32_t bytes_ff(uint32_t x) { return x & 0x80808080u & ((x & 0x7F7F7F7Fu) + 0x01010101u); }
__device__ __forceinline__ uint32_t bytes_rst(uint32_t x) {
    const uint32_t y = (x ^ 0xD0D0D0D0u) & 0xF8F8F8F8u;
    return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;
}
__device__ __forceinline__ uint32_t gather_bit7(uint32_t t) { return (((t >> 7) * 0x00204081u) >> 21) & 15u; }

/* classification of the lane's bytes: bit i of keep = byte i goes to the clean stream,
 * bit i of mark = byte i is the 0xFF of a restart marker.
 * Byte by byte (src/bitstream.c:7-18 and the restart extension), with d = "this byte is a dropped
 * follower", d of byte 0 = d0:  ff = exists & !d & byte == 0xFF;  mark = ff & use_rst & next byte in
 * 0xD0..0xD7 (past the end of the scan: zero padding, never a marker);  keep = exists & !d & !mark;
 * d of the next byte = ff.  A byte is dropped iff an odd number of 0xFF bytes, counted from the last
 * kept non-0xFF byte, stands right before it -- the "escaped character" problem of text parsers,
 * solved on the 16-bit mask of 0xFF bytes with one addition: the carry of a run of ones that starts
 * on an odd bit comes out on the other parity than that of a run that starts on an even bit. */
__device__ __forceinline__ void raw_lane_classify(const RawLane &r, bool use_rst, uint32_t &keep, uint32_t &mark) {
    const uint32_t ex = (1u << r.n) - 1u;   /* n <= 16 */
    const uint32_t B = (gather_bit7(bytes_ff(r.w[0])) | (gather_bit7(bytes_ff(r.w[1])) << 4) |
                        (gather_bit7(bytes_ff(r.w[2])) << 8) | (gather_bit7(bytes_ff(r.w[3])) << 12)) & ex;
    const uint32_t b = B & ~r.d0;                       /* a dropped first byte starts no run */
    const uint32_t follows = (b << 1) | r.d0;           /* bytes right behind a 0xFF of a run */
    const uint32_t even = 0x55555555u;
    const uint32_t odd_starts = b & ~even & ~follows;   /* runs that start on an odd bit */
    const uint32_t inv = (odd_starts + b) << 1;
    const uint32_t dropped = (even ^ inv) & follows;
    mark = 0;
    if (use_rst) {   /* per image: wave-uniform */
        const uint32_t R = gather_bit7(bytes_rst(r.w[0])) | (gather_bit7(bytes_rst(r.w[1])) << 4) |
                           (gather_bit7(bytes_rst(r.w[2])) << 8) | (gather_bit7(bytes_rst(r.w[3])) << 12) |
                           ((r.nx >= 0xD0u && r.nx <= 0xD7u) ? 1u << 16 : 0u);
        mark = B & ~dropped & (R >> 1) & ex;
    }
    keep = ex & ~dropped & ~mark;
}

/* exclusive prefix sums of (a, b) over the 256 lanes of a work-group; totals in ta, tb */
__device__ __forceinline__ void wg_exclusive_scan2(uint32_t a, uint32_t b, uint32_t *s_tmp /* [8] */,
                                                   uint32_t &ea, uint32_t &eb, uint32_t &ta, uint32_t &tb) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t ia = a, ib = b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t oa = __shfl_up(ia, d), ob = __shfl_up(ib, d);
        if ((int)lane >= d) { ia += oa; ib += ob; }
    }
    if (lane == 63) { s_tmp[wave] = ia; s_tmp[4 + wave] = ib; }
    __syncthreads();
    uint32_t pa = 0, pb = 0;
    ta = 0; tb = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
        if (w < wave) { pa += s_tmp[w]; pb += s_tmp[4 + w]; }
        ta += s_tmp[w]; tb += s_tmp[4 + w];
    }
    ea = pa + ia - a; eb = pb + ib - b;
    __syncthreads();
}

/* The same for the per-lane counts of a raw chunk (kept bytes <= 16, markers <= 8 per lane, 256
 * lanes: both sums fit 16 bits): one packed value, scanned inside the wave with DPP row shifts and
 * row broadcasts (6 adds instead of 12 LDS permutes with their selects). */
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x) {
#define JD_DPP_ADD(ctrl, row_mask) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, row_mask, 0xF, true)
    JD_DPP_ADD(0x111, 0xF);   /* row_shr:1 */
    JD_DPP_ADD(0x112, 0xF);   /* row_shr:2 */
    JD_DPP_ADD(0x114, 0xF);   /* row_shr:4 */
    JD_DPP_ADD(0x118, 0xF);   /* row_shr:8: every row of 16 lanes is scanned */
    JD_DPP_ADD(0x142, 0xA);   /* row_bcast:15 into rows 1 and 3 */
    JD_DPP_ADD(0x143, 0xC);   /* row_bcast:31 into rows 2 and 3 */
#undef JD_DPP_ADD
    return x;
}
__device__ __forceinline__ void wg_exclusive_scan_packed(uint32_t a, uint32_t b, uint32_t *s_tmp /* [4] */,
                                                         uint32_t &ea, uint32_t &eb, uint32_t &ta, uint32_t &tb) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t v = a | (b << 16);
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
        const uint32_t t = s_tmp[w];
        pre += w < wave ? t : 0u;
        tot += t;
    }
    const uint32_t ex = pre + inc - v;
    ea = ex & 0xFFFFu; eb = ex >> 16; ta = tot & 0xFFFFu; tb = tot >> 16;
    __syncthreads();
}

/* A work-group takes JD_RAW_PER_WG chunks of the work list, one after the other, and issues the
 * loads of all of them before it looks at the first: a chunk is one load, a few dozen
 * instructions, a barrier and one store per lane, so a work-group that takes a single chunk
 * spends its life waiting for that load, and the chip cannot hold enough of them to cover it. */
__global__ __launch_bounds__(256) void k_unstuff_count(const jd_image *__restrict__ imgs,
                                                       const jd_raw_wg *__restrict__ wgs, uint32_t n_wgs,
                                                       const uint8_t *__restrict__ raw,
                                                       uint2 *__restrict__ chunk_cnt,
                                                       uint32_t *__restrict__ lane_masks) {
    __shared__ uint32_t s_tmp[8];
    jd_raw_wg wg[JD_RAW_PER_WG];
    uint4 v[JD_RAW_PER_WG];
#pragma unroll
    for (uint32_t e = 0; e < JD_RAW_PER_WG; ++e) {
        const uint32_t idx = blockIdx.x * JD_RAW_PER_WG + e;
        wg[e] = wgs[min(idx, n_wgs - 1u)];
        const jd_image &im = imgs[wg[e].img];
        v[e] = raw_lane_fetch(raw + im.raw_off, idx < n_wgs ? im.raw_len : 0u, wg[e].chunk * JD_RAW_CHUNK + threadIdx.x * 16u);
    }
#pragma unroll
    for (uint32_t e = 0; e < JD_RAW_PER_WG; ++e) {
        const uint32_t idx = blockIdx.x * JD_RAW_PER_WG + e;
        if (idx >= n_wgs) break;   /* the same for the whole work-group */
        const jd_image &im = imgs[wg[e].img];
        RawLane r = raw_lane_load(v[e], raw + im.raw_off, im.raw_len, wg[e].chunk * JD_RAW_CHUNK + threadIdx.x * 16u);
        uint32_t keep, mark;
        raw_lane_classify(r, im.ri_blocks != 0, keep, mark);
        /* the classification is kept for k_unstuff_marks / k_unstuff_scatter (4 bytes per 16 of the scan) */
        lane_masks[(size_t)idx * 256u + threadIdx.x] = keep | (mark << 16);
        uint32_t ea, eb, ta, tb;
        wg_exclusive_scan_packed(__popc(keep), __popc(mark), s_tmp, ea, eb, ta, tb);
        if (threadIdx.x == 0) chunk_cnt[im.chunk_off + wg[e].chunk] = make_uint2(ta, tb);
    }
}

/* one work-group per image: chunk_base = exclusive scan of chunk_cnt; U[0] = 0 and, when the
 * stream holds fewer markers than the geometry allows, the end of the last interval */
__global__ __launch_bounds__(256) void k_unstuff_chunks(const jd_image *__restrict__ imgs,
                                                        const uint2 *__restrict__ chunk_cnt,
                                                        uint2 *__restrict__ chunk_base,
                                                        uint32_t *__restrict__ ivl_u) {
    __shared__ uint32_t s_tmp[8];
    const jd_image &im = imgs[blockIdx.x];
    if (!im.ncomp) return;
    uint32_t ca = 0, cb = 0;
    for (uint32_t base = 0; base < im.n_chunks; base += 256) {
        uint32_t c = base + threadIdx.x;
        uint2 v = c < im.n_chunks ? chunk_cnt[im.chunk_off + c] : make_uint2(0, 0);
        uint32_t ea, eb, ta, tb;
        wg_exclusive_scan2(v.x, v.y, s_tmp, ea, eb, ta, tb);
        if (c < im.n_chunks) chunk_base[im.chunk_off + c] = make_uint2(ca + ea, cb + eb);
        ca += ta; cb += tb;
    }
    if (threadIdx.x == 0) {
        ivl_u[im.u_off] = 0;
        if (cb < im.n_intervals) ivl_u[im.u_off + cb + 1] = ca;
    }
}

__global__ __launch_bounds__(256) void k_unstuff_marks(const jd_image *__restrict__ imgs,
                                                       const jd_raw_wg *__restrict__ wgs,
                                                       const uint8_t *__restrict__ raw,
                                                       const uint2 *__restrict__ chunk_cnt,
                                                       const uint2 *__restrict__ chunk_base,
                                                       const uint32_t *__restrict__ lane_masks,
                                                       uint32_t *__restrict__ ivl_u) {
    __shared__ uint32_t s_tmp[8];
    const jd_raw_wg wg = wgs[blockIdx.x];
    const jd_image &im = imgs[wg.img];
    if (chunk_cnt[im.chunk_off + wg.chunk].y == 0) return; /* no marker in this chunk */
    const uint32_t km = lane_masks[(size_t)blockIdx.x * 256u + threadIdx.x];
    uint32_t keep = km & 0xFFFFu, mark = km >> 16;
    uint32_t ea, eb, ta, tb;
    wg_exclusive_scan_packed(__popc(keep), __popc(mark), s_tmp, ea, eb, ta, tb);
    const uint2 cb = chunk_base[im.chunk_off + wg.chunk];
    uint32_t u = cb.x + ea, m = cb.y + eb;
    while (mark) {
        uint32_t i = __ffs(mark) - 1;
        mark &= mark - 1;
        uint32_t before = __popc(keep & ((1u << i) - 1u));
        if (m < im.n_intervals) ivl_u[im.u_off + m + 1] = u + before; /* interval m+1 starts here */
        ++m;
    }
}

/* one work-group per image: interval lengths -> first segment of each interval, n_segs, the
 * per-segment words and the 0xFF padding behind each interval */
__global__ __launch_bounds__(256) void k_unstuff_layout(jd_image *__restrict__ imgs,
                                                        const uint32_t *__restrict__ ivl_u,
                                                        uint32_t *__restrict__ ivl_seg,
                                                        uint32_t *__restrict__ seg_info,
                                                        uint8_t *__restrict__ clean) {
    __shared__ uint32_t s_tmp[8];
    jd_image &im = imgs[blockIdx.x];
    if (!im.ncomp) return;
    const uint32_t nI = im.n_intervals;
    const uint32_t *U = ivl_u + im.u_off;
    uint32_t *S = ivl_seg + im.ivl_off;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nI; base += 256) {
        uint32_t j = base + threadIdx.x, ns = 0;
        if (j < nI) {
            uint32_t a = U[j], b = U[j + 1];
            /* intervals the stream never reaches (fewer markers than the geometry allows) get
             * no segment; a reached but empty interval still gets one */
            if (a != ~0u && b != ~0u && b >= a) ns = max(1u, (b - a + JD_SEG_BYTES - 1) / JD_SEG_BYTES);
        }
        uint32_t ea, eb, ta, tb;
        wg_exclusive_scan2(ns, 0, s_tmp, ea, eb, ta, tb);
        if (j < nI) S[j] = carry + ea;
        carry += ta;
    }
    const uint32_t n_segs = min(carry, im.seg_cap);
    if (threadIdx.x == 0) im.n_segs = n_segs;
    __syncthreads(); /* S[] written above is read below by other lanes of this work-group */
    for (uint32_t s = threadIdx.x; s < n_segs; s += 256) {
        uint32_t lo = 0, hi = nI - 1; /* last interval whose first segment is <= s */
        while (lo < hi) {
            uint32_t mid = (lo + hi + 1) >> 1;
            if (S[mid] <= s) lo = mid; else hi = mid - 1;
        }
        uint32_t a = U[lo], b = U[lo + 1];
        uint32_t len = (a != ~0u && b != ~0u && b >= a) ? b - a : 0u;
        uint32_t k = s - S[lo];
        uint32_t bytes = len > k * JD_SEG_BYTES ? min(JD_SEG_BYTES, len - k * JD_SEG_BYTES) : 0u;
        seg_info[im.seg_off + s] = (bytes * 8u) | (k == 0 ? JD_SEG_FIRST : 0u);
        uint8_t *dst = clean + im.clean_off;
        uint32_t q = bytes;
        for (; q < JD_SEG_BYTES && (q & 15u); ++q) dst[clean_byte_addr(s, q)] = 0xFF;
        for (; q < JD_SEG_BYTES; q += 16)
            *reinterpret_cast<uint4 *>(dst + clean_byte_addr(s, q)) = make_uint4(~0u, ~0u, ~0u, ~0u);
    }
}

/* Kept bytes of one chunk go to their padded position.  Per restart interval present in the
 * chunk (one, unless markers fall inside it) the bytes are first compacted in LDS at the byte
 * alignment of their destination, then stored 16 bytes per lane; only the ragged first and last
 * 16 bytes of a run are stored byte by byte (the neighbouring chunk fills in the rest). */
__device__ __forceinline__ void scatter_chunk(const jd_image &im, const jd_raw_wg wg, const uint32_t km, const uint4 rv,
                                              const uint2 *__restrict__ chunk_base, const uint32_t *__restrict__ ivl_u,
                                              const uint32_t *__restrict__ ivl_seg, uint8_t *__restrict__ clean,
                                              uint32_t *s_tmp, uint8_t *s_stage) {
    const uint32_t keep = km & 0xFFFFu, mark = km >> 16;
    const uint32_t rw[4] = {rv.x, rv.y, rv.z, rv.w};
    uint32_t ea, eb, ta, tb;
    wg_exclusive_scan_packed(__popc(keep), __popc(mark), s_tmp, ea, eb, ta, tb);
    const uint2 cb = chunk_base[im.chunk_off + wg.chunk];
    const uint32_t u_lo = cb.x, u_hi = cb.x + ta;   /* unpadded clean offsets of the chunk's kept bytes */
    const uint32_t nI = im.n_intervals;
    const uint64_t cap = (uint64_t)im.seg_cap * JD_SEG_BYTES;
    const uint32_t *U = ivl_u + im.u_off;
    const uint32_t *S = ivl_seg + im.ivl_off;
    uint8_t *dst = clean + im.clean_off;
    /* A chunk without a marker (any chunk of an image without restart intervals) is one run, and
     * all its kept bytes belong to it: the lane drops its few dropped bytes in registers (one
     * 128-bit byte removal each; a lane has one with probability 1/16 in entropy-coded data) and
     * stores its bytes with at most four LDS stores at their byte offset (LDS takes any
     * alignment), instead of 16 conditional byte stores. */
    if (tb == 0 && cb.y < nI && U[cb.y] <= u_lo && U[cb.y + 1] >= u_hi && u_hi > u_lo) {
        const uint64_t shift = (uint64_t)S[cb.y] * JD_SEG_BYTES - U[cb.y];
        const uint64_t pos_a = u_lo + shift, pos_b = u_hi + shift, base = pos_a & ~(uint64_t)15;
        uint32_t w0 = rv.x, w1 = rv.y, w2 = rv.z, w3 = rv.w;
        const uint32_t cnt = __popc(keep);
        uint32_t off = (uint32_t)(pos_a - base) + ea;
        if (cnt == 16u) {
            struct __attribute__((packed)) P16 { uint32_t a, b, c, d; } v{w0, w1, w2, w3};
            __builtin_memcpy(s_stage + off, &v, 16);
        } else if (cnt) {
            uint32_t drop = ~keep & ((2u << (31 - __clz((int)keep))) - 1u);   /* below the last kept byte */
            while (drop) {   /* remove byte p, the highest first: the bytes above it move down by one */
                const uint32_t p = 31u - (uint32_t)__clz((int)drop);
                drop &= ~(1u << p);
                const uint32_t s0 = __builtin_amdgcn_alignbyte(w1, w0, 1), s1 = __builtin_amdgcn_alignbyte(w2, w1, 1),
                               s2 = __builtin_amdgcn_alignbyte(w3, w2, 1), s3 = w3 >> 8;
                auto low = [&](uint32_t j) -> uint32_t {   /* the bytes of word j below byte p */
                    const int c = (int)p - 4 * (int)j;
                    return c <= 0 ? 0u : c >= 4 ? ~0u : (1u << (8 * c)) - 1u;
                };
                const uint32_t m0 = low(0), m1 = low(1), m2 = low(2), m3 = low(3);
                w0 = (w0 & m0) | (s0 & ~m0); w1 = (w1 & m1) | (s1 & ~m1);
                w2 = (w2 & m2) | (s2 & ~m2); w3 = (w3 & m3) | (s3 & ~m3);
            }
            if (cnt & 8u) {
                struct __attribute__((packed)) P8 { uint32_t a, b; } v{w0, w1};
                __builtin_memcpy(s_stage + off, &v, 8);
                off += 8; w0 = w2; w1 = w3;
            }
            if (cnt & 4u) { __builtin_memcpy(s_stage + off, &w0, 4); off += 4; w0 = w1; }
            if (cnt & 2u) { const uint16_t h = (uint16_t)w0; __builtin_memcpy(s_stage + off, &h, 2); off += 2; w0 >>= 16; }
            if (cnt & 1u) s_stage[off] = (uint8_t)w0;
        }
        __syncthreads();
        /* The 16-byte pieces of four neighbouring segments share a 64-byte sector of the transposed
         * stream: four neighbouring lanes take the same piece of segments 4g .. 4g + 3, so that a
         * store instruction writes whole sectors (lane order: group of four segments, piece, segment) */
        const uint32_t pieces = (uint32_t)((pos_b - base + 15u) >> 4);
        const uint64_t G0 = base >> 4;                                   /* first piece, counted from the image's start */
        const uint32_t grp0 = (uint32_t)(G0 / (JD_SEG_BYTES / 16u)) >> 2;
        const uint32_t grp1 = (uint32_t)((G0 + pieces - 1u) / (JD_SEG_BYTES / 16u)) >> 2;
        const uint32_t total = (grp1 - grp0 + 1u) * 4u * (JD_SEG_BYTES / 16u);
        for (uint32_t t = threadIdx.x; t < total; t += 256) {
            const uint32_t seg = (grp0 + t / (4u * (JD_SEG_BYTES / 16u))) * 4u + (t & 3u), pc = (t >> 2) % (JD_SEG_BYTES / 16u);
            const uint64_t G = (uint64_t)seg * (JD_SEG_BYTES / 16u) + pc;
            if (G < G0 || G >= G0 + pieces) continue;
            const uint32_t q = (uint32_t)(G - G0);
            const uint64_t P = G << 4;
            if (P + 16u > cap) continue;
            uint8_t *d = dst + clean_byte_addr(seg, pc * 16u);
            if (P >= pos_a && P + 16u <= pos_b) {
                *reinterpret_cast<uint4 *>(d) = *reinterpret_cast<const uint4 *>(s_stage + 16u * q);
            } else {
                const uint32_t lo = P >= pos_a ? 0u : (uint32_t)(pos_a - P);
                const uint32_t hi = P + 16u <= pos_b ? 16u : (uint32_t)(pos_b - P);
                for (uint32_t k = lo; k < hi; ++k) d[k] = s_stage[16u * q + k];
            }
        }
        return;
    }
    for (uint32_t run = 0; run <= tb; ++run) {
        const uint32_t m_r = cb.y + run;              /* interval of this run */
        if (m_r >= nI) break;                         /* behind the marker that ends the data */
        const uint32_t Um = U[m_r];
        const uint32_t a = max(Um, u_lo), b = min(U[m_r + 1], u_hi);
        if (b <= a) continue;
        const uint64_t shift = (uint64_t)S[m_r] * JD_SEG_BYTES - Um;   /* padded position = u + shift */
        const uint64_t pos_a = a + shift, pos_b = b + shift, base = pos_a & ~(uint64_t)15;
        const uint32_t head = (uint32_t)(pos_a - base);
        {
            uint32_t u = u_lo + ea, m = cb.y + eb;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool kp = (keep >> i) & 1u;
                if (kp && m == m_r) s_stage[u - a + head] = (uint8_t)(rw[i >> 2] >> (8 * (i & 3)));
                u += kp ? 1u : 0u;
                m += (mark >> i) & 1u;
            }
        }
        __syncthreads();
        const uint32_t pieces = (uint32_t)((pos_b - base + 15u) >> 4);
        for (uint32_t q = threadIdx.x; q < pieces; q += 256) {
            const uint64_t P = base + 16u * (uint64_t)q;
            if (P + 16u > cap) continue;
            const uint32_t seg = (uint32_t)(P / JD_SEG_BYTES), o = (uint32_t)(P % JD_SEG_BYTES);
            uint8_t *d = dst + clean_byte_addr(seg, o);
            if (P >= pos_a && P + 16u <= pos_b) {
                *reinterpret_cast<uint4 *>(d) = *reinterpret_cast<const uint4 *>(s_stage + 16u * q);
            } else {
                const uint32_t lo = P >= pos_a ? 0u : (uint32_t)(pos_a - P);
                const uint32_t hi = P + 16u <= pos_b ? 16u : (uint32_t)(pos_b - P);
                for (uint32_t k = lo; k < hi; ++k) d[k] = s_stage[16u * q + k];
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_unstuff_scatter(const jd_image *__restrict__ imgs,
                                                         const jd_raw_wg *__restrict__ wgs, uint32_t n_wgs,
                                                         const uint8_t *__restrict__ raw,
                                                         const uint2 *__restrict__ chunk_base,
                                                         const uint32_t *__restrict__ ivl_u,
                                                         const uint32_t *__restrict__ ivl_seg,
                                                         const uint32_t *__restrict__ lane_masks,
                                                         uint8_t *__restrict__ clean) {
    __shared__ uint32_t s_tmp[8];
    __shared__ __align__(16) uint8_t s_stage[JD_RAW_CHUNK + 32];
    jd_raw_wg wg[JD_RAW_PER_WG];
    uint4 rv[JD_RAW_PER_WG];
    uint32_t km[JD_RAW_PER_WG];
    /* all loads first (see k_unstuff_count) */
#pragma unroll
    for (uint32_t e = 0; e < JD_RAW_PER_WG; ++e) {
        const uint32_t idx = min(blockIdx.x * JD_RAW_PER_WG + e, n_wgs - 1u);
        wg[e] = wgs[idx];
        const jd_image &im = imgs[wg[e].img];
        km[e] = lane_masks[(size_t)idx * 256u + threadIdx.x];
        rv[e] = raw_lane_fetch(raw + im.raw_off, im.raw_len, wg[e].chunk * JD_RAW_CHUNK + threadIdx.x * 16u);
    }
#pragma unroll
    for (uint32_t e = 0; e < JD_RAW_PER_WG; ++e) {
        if (blockIdx.x * JD_RAW_PER_WG + e >= n_wgs) break;   /* the same for the whole work-group */
        if (e) __syncthreads();   /* the staging buffer is free again */
        scatter_chunk(imgs[wg[e].img], wg[e], km[e], rv[e], chunk_base, ivl_u, ivl_seg, clean, s_tmp, s_stage);
    }
}

/* ============================================================================================
 * Huffman: per-lane sequential decoder over one segment of the clean stream
 * ========================================================================================== */

/*
 * Lookup tables in LDS.  The host flattens every Huffman table into jd_lut_hot (jdgpu_prep.c:
 * 10-bit primary + exact sub-tables, entries = code length, magnitude size, zero run).  When a
 * work-group stages the tables it rewrites the resolved entries into the form its decode loop
 * wants, so that a symbol step needs no case distinction between DC, AC, ZRL and EOB:
 *
 *   kinc = how far the symbol moves the zigzag index k (src/decoding.c:50-66: i += run, then the
 *          loop's ++i): DC 1 (k: 0 -> 1), AC run + 1, ZRL 16, any other (r, 0) 64 -- "end of block"
 *          is simply k >= 64, which also ends a unit whose last coefficient sits at index 63.
 *
 *   light entry (k_huff_first / k_huff_repair; AC magnitudes are skipped, not read):
 *     4:0 bits consumed (code + magnitude), 11:5 kinc, 12 an AC magnitude of 8 bits or more,
 *     13 no such code; in a DC table 11:8 = magnitude size instead (kinc is 1, and the DC
 *     difference is read: the segment's sums of differences are what k_seg_scan turns into the
 *     DC predictors at every segment start)
 *   write entry (k_huff_write):
 *     3:0 code length - 1, 7:4 magnitude size, 14:8 kinc (0 = no such code)
 *
 * Link entries (bit 15, see jd_lut_hot) stay as they are.
 */
#define LE_WIDE 0x1000u
#define LE_BAD  0x2000u
/* A light AC entry may stand for SEVERAL symbols (round 3): where the first symbol of a 10-bit window is an
 * ordinary AC symbol (not an end of block, no large magnitude) and the bits behind it hold further whole AC
 * symbols, the entry carries the bits and the index advance of all of them (JD_PAIR_MAX at most; the name
 * says pair because most are).  The light decodes read no AC magnitudes, so nothing else of the group is
 * needed -- as long as no symbol but the last ends the data unit.  The group is built so that all but its
 * last symbol advance k by at most 16 together; while k < JD_PAIR_KMAX = 64 - 17 that cannot reach 64, and
 * decoding the group at once IS decoding one symbol after the other, on any bit pattern.  At
 * k >= JD_PAIR_KMAX (a few per cent of the steps at quality 85) the step takes the single-symbol entry from
 * the table in global memory.  1080p 4:2:0 q85 has 5.2 bits per symbol. */
#define LE_PAIR 0x4000u
#define JD_PAIR_KMAX 47u
#ifndef JD_PAIR_MAX
#define JD_PAIR_MAX 3u
#endif

__device__ __forceinline__ uint32_t kinc_of(uint32_t size, uint32_t run, bool is_dc) {
    const uint32_t zero = run == 15u ? 16u : 64u;  /* ZRL skips 16, any other (r, 0) ends the block */
    const uint32_t ac = size ? run + 1u : zero;
    return is_dc ? 1u : ac;
}
/* hot entry (or the fields of a jd_sym) -> light / write entry */
__device__ __forceinline__ uint32_t light_entry(uint32_t len, uint32_t size, uint32_t run, bool bad, bool is_dc) {
    const uint32_t mid = is_dc ? (1u << 5) | (size << 8) : kinc_of(size, run, false) << 5;
    return (len + size) | mid | ((!is_dc && size >= 8u) ? LE_WIDE : 0u) | (bad ? LE_BAD : 0u);
}
__device__ __forceinline__ uint32_t write_entry(uint32_t len, uint32_t size, uint32_t run, bool bad, bool is_dc) {
    return ((len - 1u) & 15u) | (size << 4) | (bad ? 0u : kinc_of(size, run, is_dc) << 8);
}
template <bool LIGHT>
__device__ __forceinline__ uint32_t convert_entry(uint32_t e, bool is_dc) {
    if (e & JD_LUT_LINK) return e;
    const uint32_t len = e & 31u, size = (e >> 5) & 15u, run = (e >> 9) & 15u;
    const bool bad = (e & 0x2000u) != 0;
    return LIGHT ? light_entry(len, size, run, bad, is_dc) : write_entry(len, size, run, bad, is_dc);
}
/* light entry of primary index idx of an AC table, a pair where one fits (LE_PAIR above); e1: the single
 * entry of idx, primary: the table's first level in global memory (read for the second symbol) */
__device__ __forceinline__ uint32_t pair_entry(uint32_t e1, uint32_t idx, const uint16_t *primary) {
    if (e1 & (JD_LUT_LINK | LE_BAD | LE_WIDE)) return e1;
    uint32_t tot = e1 & 31u, kinc = (e1 >> 5) & 127u, n = 1;
    /* kinc <= 16 here means: no symbol of the group so far is an end of block, and all of them together
     * leave k below 64 when they start below JD_PAIR_KMAX */
    while (n < JD_PAIR_MAX && kinc <= 16u && tot < JD_LUT_BITS) {
        const uint32_t idx2 = (idx << tot) & ((1u << JD_LUT_BITS) - 1u);   /* the bits behind the group, zeros where the window ends */
        const uint32_t e2 = convert_entry<true>(primary[idx2], false);
        if (e2 & (JD_LUT_LINK | LE_BAD | LE_WIDE)) break;
        if (tot + (e2 & 31u) > JD_LUT_BITS) break;                /* the next symbol is not whole inside the window */
        tot += e2 & 31u; kinc += (e2 >> 5) & 127u; ++n;
    }
    return n == 1u ? e1 : (tot | (kinc << 5) | LE_PAIR);
}

/* per data unit of the MCU, in LDS: byte offsets of the unit's DC / AC table in the LDS copy,
 * slots = dc slot | ac slot << 4 (canonical fallback only), nb = the unit after it in the MCU;
 * e0..e2 select the unit's component: 1 / 0, multiplied */
struct __attribute__((aligned(16))) BlkDesc {
    uint32_t dc_off, ac_off, slots, nb;
    int e0, e1, e2;
    uint32_t pad;
};

struct SegTables {
    const uint8_t *lut;     /* LDS copy of the lookup part of the image's tables, converted */
    const jd_lut *glut;     /* the batch's distinct tables in global memory, whole (canonical fallback) ... */
    const uint16_t *gidx;   /* ... and which of them the image's slots are (jd_image::lut_idx) */
    const BlkDesc *blk;     /* LDS */
    uint32_t bpm;
};

/* one symbol lookup against the LDS tables: desc = the unit's BlkDesc, isdc selects its DC or AC
 * table.  Returns the resolved (light or write) entry. */
template <bool LIGHT, bool PAIRS = false>
__device__ __forceinline__ uint32_t lookup_entry(const SegTables &T, const BlkDesc &desc, bool isdc, uint32_t top, uint32_t kk = 0u /* PAIRS: the zigzag index */) {
    const uint32_t off = isdc ? desc.dc_off : desc.ac_off;
    /* v_bfe_u32 + v_lshl_add_u32 (through asm: the compiler folds the two shifts of the C form into
     * shift + mask + add); the entry is read zero-extended, link entries are the ones whose 16-bit
     * value has bit 15 set (an unsigned compare: a 16-bit signed one makes the compiler re-mask the entry) */
    uint32_t idx;
    asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(idx) : "v"(top), "n"(32 - JD_LUT_BITS), "n"(JD_LUT_BITS));
    uint32_t e = *reinterpret_cast<const uint16_t *>(T.lut + (idx * 2u + off));
    asm("" : "+v"(e));   /* hides the 16-bit origin, or the compare below becomes a 16-bit one plus a mask */
    if (PAIRS && __builtin_expect((e & LE_PAIR) != 0u && kk >= JD_PAIR_KMAX, 0)) {
        /* the first symbol of the pair may end the data unit: the single entry, from global memory (rare) */
        uint32_t slot = desc.slots >> 4;
        asm volatile("" : "+v"(slot));
        e = convert_entry<true>(T.glut[T.gidx[slot]].hot.primary[idx], false);
    }
    if (__builtin_expect(e > 0x7FFFu, 0)) {
        const uint32_t first = e & 0xFFFu;
        if (__builtin_expect(first == JD_LUT_NOSUB, 0)) {
            uint32_t slot = isdc ? (desc.slots & 15u) : (desc.slots >> 4);
            asm volatile("" : "+v"(slot));   /* keeps the address arithmetic of this (very rare) path inside it */
            const jd_sym s = jd_lut_lookup_canon(T.glut + T.gidx[slot], top >> 16);
            e = LIGHT ? light_entry(s.len, s.size, s.run, s.bad != 0, isdc) : write_entry(s.len, s.size, s.run, s.bad != 0, isdc);
        } else {
            const uint32_t rest = (top >> 16) & ((1u << (16 - JD_LUT_BITS)) - 1u);
            const uint32_t idx2 = first + (rest >> ((16 - JD_LUT_BITS) - ((e >> 12) & 7u)));
            e = *reinterpret_cast<const uint16_t *>(T.lut + off + sizeof(uint16_t) * ((1u << JD_LUT_BITS) + idx2));
        }
    }
    return e;
}

__device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; } /* v_mad_i32_i24 */

/* JPEG EXTEND (src/decoding.c:31-33) of the `size` magnitude bits that END at bit 32 - tot of the
 * 32-bit window `top` (i.e. the last `size` of the first `tot` bits); size 0 -> 0.  A magnitude
 * below 2^(size-1) stands for mag - (2^size - 1): with m = 2^size - 1 that is 2 mag < m. */
__device__ __forceinline__ int extend_at(uint32_t top, uint32_t tot, uint32_t size) {
    const uint32_t mag = __builtin_amdgcn_ubfe(top, 32u - tot, size);
    uint32_t m;                                    /* (1 << size) - 1 */
    asm("v_bfm_b32 %0, %1, 0" : "=v"(m) : "v"(size));
    return (int)mag - (int)(mag + mag < m ? m : 0u);
}

/* magnitude bits -> value (JPEG EXTEND, src/decoding.c:31-33); v = stream bits left aligned at
 * the first magnitude bit; size 0 -> 0 whatever v holds */
__device__ __forceinline__ int extend_top(uint32_t v, uint32_t size) {
    uint32_t mag = (v >> 1) >> (31u - size);
    int neg = (int)(0xFFFFFFFFu << size) + 1;
    return (int)mag + (((int)v < 0) ? 0 : neg);
}

/*
 * MSB-first bit reader over the lane's segment (and the first words of the next one: a symbol
 * may straddle the boundary), replacing src/bitstream.c:28-45.  The decode loops take
 * JD_RSTEPS = 2 symbols per iteration; the reader is built around that:
 *
 *   window  three 32-bit words.  The next 32 stream bits are the low r bits of w0 followed by
 *           the high 32 - r bits of w1 -- one v_alignbit_b32.  Consuming bits decrements r; when
 *           w0 is used up the window shifts by a word (two selects), which two symbols of at
 *           most 31 bits can do twice: w2 is what the second shift moves in.
 *   refill  after every second symbol step, for all lanes, outside any divergent code: the words
 *           the shifts used up come from (qa, qb) = words c and c + 1, loaded at an EARLIER
 *           refill, and the lanes whose window moved load the words the queue now lacks.  So no
 *           load is ever waited for right after its issue (the compiler places the wait at the
 *           first use), the L2 / Infinity Cache latency has two steps to pass, and there is no
 *           refill branch that some lane of the wave takes in every step anyway.
 */
#ifndef JD_RSTEPS
#define JD_RSTEPS 2
#endif
/* AC symbol steps per iteration of the light decodes, after the one step that takes any symbol.  Odd: the
 * steps pair up, a refill after each pair.  Two instantiations, picked per image by the plan
 * (jd_image::long_units): streams of short data units -- a dozen symbols, most of them in groups since round
 * 3 -- finish a unit in a few steps and waste the rest of a long iteration (1080p q85: 1 + 1 steps 3.47 ms,
 * 1 + 3: 3.64, 1 + 5: 3.80 for the sync stage); streams of long units (quality 100: 50 symbols a unit, no
 * groups) pay every iteration's fixed part, the refill and the descriptor reload (config 2: 1 + 3 steps
 * 1.36 ms, 1 + 1: 1.73). */
#ifndef JD_ASTEPS_SHORT
#define JD_ASTEPS_SHORT 1
#endif
#ifndef JD_ASTEPS_LONG
#define JD_ASTEPS_LONG 3
#endif
struct BitReader {
    const char *base;          /* the image's clean stream (uniform: a scalar base + 32-bit lane offsets; the plan keeps a
                                  stream below 4 GiB) */
    uint32_t b_own, b_next;    /* byte offset of word 0 of this segment; of the next one, biased (word_offset) */
    uint32_t w0, w1, w2;       /* window (byte-swapped to native order) */
    int r;                     /* unread bits of w0, 0..31 */
    uint32_t s;                /* window shifts since the last refill, 0..2 */
    uint32_t qa, qb;           /* the two words behind the window (raw, big-endian) */
    uint32_t c;                /* index of the word in qa */

    /* byte offset of word w: 16-byte pieces are 1024 bytes apart, so word w sits at
     * 4 w + 1008 (w >> 2) from word 0 of its segment (5 instructions, the mask-and-or form took 8).
     * For w >= JD_SEG_WORDS the same expression runs on from the NEXT segment's word 0 when it starts
     * at b_next = that word's offset - (4 + 1008 / 4) JD_SEG_WORDS. */
    __device__ __forceinline__ uint32_t word_offset(uint32_t w) const {
        const uint32_t sel = w >= JD_SEG_WORDS ? b_next : b_own;
        return __umul24(w >> 2, 1008u) + sel + (w << 2);
    }
    /* raw (big-endian) word w of this segment, w >= JD_SEG_WORDS: of the next; the byte swap is
     * done when the word is USED, so that the load stays in flight instead of being waited for */
    __device__ __forceinline__ uint32_t word(uint32_t w) const {
        return *reinterpret_cast<const uint32_t *>(base + word_offset(w));
    }
    /* start at bit p of the segment */
    __device__ __forceinline__ void init(uint32_t p) {
        const uint32_t W = (p + 31u) >> 5;          /* word that holds w1; w0 = the one before (unused when p % 32 == 0) */
        w0 = __builtin_bswap32(word(W ? W - 1u : 0u));
        w1 = __builtin_bswap32(word(W));
        w2 = __builtin_bswap32(word(W + 1u));
        r = (int)((32u - (p & 31u)) & 31u);
        s = 0;
        c = W + 2u; qa = word(c); qb = word(c + 1u);
    }
    __device__ __forceinline__ uint32_t top() const { return __builtin_amdgcn_alignbit(w0, w1, (uint32_t)r); }
    /* drop n <= 31 bits; at most JD_RSTEPS calls between two refills */
    __device__ __forceinline__ void consume(uint32_t n) {
        r -= (int)n;
        const bool need = r < 0;
        w0 = need ? w1 : w0;
        w1 = need ? w2 : w1;
        s += (uint32_t)r >> 31;
        r &= 31;                                   /* -31..-1 -> 1..31 */
    }
    __device__ __forceinline__ void refill() {
        const uint32_t a = __builtin_bswap32(qa), b = __builtin_bswap32(qb);
        w1 = s == 2u ? a : w1;                     /* two shifts left w1 = w2 = the old w2 */
        w2 = s == 2u ? b : (s == 1u ? a : w2);
        c += s;
        /* Only the lanes whose window moved load, and only the words they lack: an iteration consumes
         * a dozen bits on average, and the per-CU vector memory path (TA / L1) is what k_huff_write
         * runs into -- reloading both words in every iteration made 7 L2 read requests per load
         * instruction, 44 % of that path's traffic.  The old qb moves up BEFORE the loads are issued,
         * so nothing reads a register a load is in flight to until the next refill. */
        const uint32_t oa = word_offset(c), ob = word_offset(c + 1u);
        qa = s == 1u ? qb : qa;
        if (s == 2u) qa = *reinterpret_cast<const uint32_t *>(base + oa);
        if (s != 0u) qb = *reinterpret_cast<const uint32_t *>(base + ob);
        s = 0;
    }
};

__device__ __forceinline__ BitReader make_reader(const uint8_t *clean_img, uint32_t seg, uint32_t n_segs) {
    BitReader r;
    r.base = reinterpret_cast<const char *>(clean_img);
    r.b_own = (uint32_t)clean_word_index(seg, 0) << 2;
    /* no next segment: the words past the end are never used; keep the loads inside the buffer */
    r.b_next = ((seg + 1 < n_segs) ? (uint32_t)clean_word_index(seg + 1, 0) << 2 : r.b_own) - (4u + 252u) * JD_SEG_WORDS;
    return r;
}

/* what a light decode reports: the exit state, n = data units COMPLETED | JD_CNT_WIDE when an AC
 * value of 8 or more magnitude bits was met, and the sums of the DC differences per component */
struct LightResult { uint32_t exit_state; int n, dc0, dc1, dc2; };
#define JD_CNT_WIDE (1 << 30) /* in cp_val[].x and cnt[].x beside the unit count */

/* checkpoints inside a segment: at the first symbol boundary at or after bit (c+1) * JD_CP_BITS
 * a light decode records its state (cp_state) and what it counted since the previous checkpoint
 * (cp_val; entry JD_CP: from the last checkpoint to the end).  A later decode of the same segment
 * from another start state that arrives at a checkpoint in the recorded state is from there on
 * identical to the recorded one: it stops and adds up the recorded rest. */
#ifndef JD_CP
#define JD_CP       3u
#endif
#define JD_CP_BITS  (JD_SEG_BITS / (JD_CP + 1u))
#define JD_CP_NONE  0xFFFFFFFFu
/* bit position of checkpoint c (c == JD_CP: the end of the segment) */
__device__ __forceinline__ constexpr uint32_t cp_mark(uint32_t c) { return (c + 1u) * JD_CP_BITS; }

/*
 * Light decode (no coefficient output): from packed start state `in` run to the end of the
 * segment's valid bits; report the exit state, the number of data units COMPLETED and the sum
 * of DC differences seen per component.  COMPARE: stop early at a checkpoint that matches the
 * recorded decode (whose exit state is old_exit).
 */
template <bool COMPARE, bool LONG_UNITS>   /* LONG_UNITS: jd_image::long_units -- the longer iteration, no symbol groups in the tables */
__device__ __forceinline__ LightResult decode_light(BitReader R, const SegTables &T, uint32_t in, uint32_t valid,
                                                    uint32_t *__restrict__ cp_state, int4 *__restrict__ cp_val,
                                                    uint32_t old_exit) {
    constexpr int ASTEPS = LONG_UNITS ? JD_ASTEPS_LONG : JD_ASTEPS_SHORT;
    constexpr bool PAIRS = !LONG_UNITS;
    uint32_t p = in & 63u, k = (in >> 6) & 127u, b = JD_ST_UNIT(in);
    uint32_t fl = 0;             /* OR of the entries since the last checkpoint: LE_WIDE, LE_BAD */
    int n = 0, dc0 = 0, dc1 = 0, dc2 = 0;  /* units completed, DC difference sums since the last checkpoint */
    int tn = 0, t0 = 0, t1 = 0, t2 = 0;    /* ... up to the last checkpoint */
    uint32_t tf = 0;
    uint32_t old[JD_CP];
#pragma unroll
    for (uint32_t c = 0; c < JD_CP; ++c) old[c] = COMPARE ? cp_state[c] : JD_CP_NONE;
    bool matched = false;
    R.init(p);
    BlkDesc tb = T.blk[b];
#pragma unroll
    for (uint32_t c = 0; c <= JD_CP; ++c) {
        const uint32_t mark = cp_mark(c);
        const uint32_t lim = c < JD_CP ? min(valid, mark) : valid;
        while (p < lim) {
            /* one step for any symbol: the lanes at the start of a data unit take their DC symbol here
             * (the difference, src/decoding.c:44, goes to the sum of the unit's component) */
            {
                const bool isdc = k == 0u;
                const uint32_t top = R.top();
                const uint32_t e = lookup_entry<true, PAIRS>(T, tb, isdc, top, k);
                fl |= e;
                const uint32_t tot = e & 31u;
                int d = extend_at(top, tot, (e >> 8) & 15u);
                d = isdc ? d : 0;
                dc0 = mad24(d, tb.e0, dc0); dc1 = mad24(d, tb.e1, dc1); dc2 = mad24(d, tb.e2, dc2);
                p += tot;
                k = isdc ? 1u : k + ((e >> 5) & 127u);
                R.consume(tot);
                const bool end = k >= 64u;
                k = end ? 0u : k; n += end ? 1 : 0; b = end ? tb.nb : b;
            }
            /* ASTEPS AC symbols (or groups of them) of the lanes inside a unit; a lane whose unit ends waits for
             * the first step of the next iteration */
#pragma unroll
            for (int st = 0; st < ASTEPS; ++st) {
                if (k != 0u && p < lim) {
                    const uint32_t top = R.top();
                    const uint32_t e = lookup_entry<true, PAIRS>(T, tb, false, top, k);
                    fl |= e;
                    const uint32_t tot = e & 31u;
                    p += tot;
                    k += (e >> 5) & 127u;
                    R.consume(tot);
                    const bool end = k >= 64u;   /* end of the data unit, branch-free */
                    k = end ? 0u : k; n += end ? 1 : 0; b = end ? tb.nb : b;
                }
                if ((st & 1) == 0) R.refill();   /* after every second symbol step, all lanes */
            }
            tb = T.blk[b]; /* once per iteration, all lanes: no divergent merge copies the descriptor around */
        }
        if (c < JD_CP && !matched) {
            if (p >= mark) {
                const uint32_t st = p | (k << 16) | (b << 24);
                tn += n; t0 += dc0; t1 += dc1; t2 += dc2; tf |= fl;
                cp_val[c] = make_int4(n | ((fl & LE_WIDE) ? JD_CNT_WIDE : 0), dc0, dc1, dc2); /* also on a match: the way here was new */
                if (COMPARE && st == old[c]) {
#pragma unroll
                    for (uint32_t i = c + 1; i <= JD_CP; ++i) {
                        const int4 v = cp_val[i];
                        tn += v.x & (JD_CNT_WIDE - 1); t0 += v.y; t1 += v.z; t2 += v.w;
                        tf |= (v.x & JD_CNT_WIDE) ? LE_WIDE : 0u;
                    }
                    matched = true;
                    p = 0xFFFFFFFFu; /* no further step */
                } else {
                    cp_state[c] = st;
                }
                n = dc0 = dc1 = dc2 = 0; fl &= LE_BAD; /* the error flag belongs to the exit state of the whole decode */
            } else { /* the segment ends before this checkpoint */
                cp_state[c] = JD_CP_NONE;
                cp_val[c] = make_int4(0, 0, 0, 0);
            }
        }
    }
    LightResult r;
    if (matched) {
        r.exit_state = old_exit;
    } else {
        cp_val[JD_CP] = make_int4(n | ((fl & LE_WIDE) ? JD_CNT_WIDE : 0), dc0, dc1, dc2);
        tn += n; t0 += dc0; t1 += dc1; t2 += dc2; tf |= fl;
        const uint32_t ovs = p >= JD_SEG_BITS ? p - JD_SEG_BITS : 0u;
        r.exit_state = JD_ST_PACK(ovs, k, b) | ((tf & LE_BAD) ? JD_ST_BAD : 0u);
    }
    r.n = tn | ((tf & LE_WIDE) ? JD_CNT_WIDE : 0); r.dc0 = t0; r.dc1 = t1; r.dc2 = t2;
    return r;
}

/* state-only decode: from bit p0 of the reader's segment, expecting a DC symbol of data unit 0,
 * to the end of the segment; returns the packed state reached (no counting).  Used to guess the
 * start state of the NEXT segment from the last JD_OVERLAP bits of this one. */
#ifndef JD_OVERLAP
#define JD_OVERLAP 1024u
#endif
#ifndef JD_OVERLAP_SPAN
#define JD_OVERLAP_SPAN 1536u   /* ... where a lane takes several segments and guesses only for the first: the longer
                                   guess is paid by a third of the segments (C3 sync: 1024: 4.33, 1536: 4.21, 2048: 4.34,
                                   3072: 4.29, 4096: 4.48 ms; config 5 3.23 / 3.23 / 3.28 / 3.35 / 3.46) */
#endif
template <bool PAIRS>
__device__ __forceinline__ uint32_t decode_state_only(BitReader R, const SegTables &T, uint32_t p0, uint32_t valid) {
    uint32_t p = p0, k = 0, b = 0;
    R.init(p);
    BlkDesc tb = T.blk[b];
    while (p < valid) {
#pragma unroll
        for (int st = 0; st < JD_RSTEPS; ++st) {
            if (st == 0 || p < valid) {
                const bool isdc = k == 0;
                const uint32_t e = lookup_entry<true, PAIRS>(T, tb, isdc, R.top(), k);
                const uint32_t tot = e & 31u;
                p += tot;
                k = isdc ? 1u : k + ((e >> 5) & 127u);
                R.consume(tot);
                const bool end = k >= 64u;
                k = end ? 0u : k; b = end ? tb.nb : b;
            }
            tb = T.blk[b];
        }
        R.refill();
    }
    return JD_ST_PACK(p - valid, k, b);
}

/* copy the lookup part of the image's nslots tables (luts: the batch's distinct tables) into LDS,
 * resolved entries rewritten for the kernel's decode loop (LIGHT or write form), and build the per-unit
 * descriptors */
template <bool LIGHT>
__device__ __forceinline__ void stage_tables(jd_lut_hot *s_lut, BlkDesc *s_blk, const jd_lut *luts, const jd_image &im,
                                             uint32_t nslots = JD_LUT_SLOTS) {
    constexpr uint32_t HOT32 = sizeof(jd_lut_hot) / 4;
    uint32_t *dst = reinterpret_cast<uint32_t *>(s_lut);
    /* two entries per lane and round, not unrolled: this loop must not set the kernel's register count */
#pragma unroll 1
    for (uint32_t slot = 0; slot < nslots; ++slot) {
        const jd_lut &g = luts[im.lut_idx[slot]];
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&g);
        const bool is_dc = g.is_dc != 0;
#pragma unroll 1
        for (uint32_t i = threadIdx.x; i < HOT32; i += blockDim.x) {
            const uint32_t v = src[i];
            uint32_t lo = convert_entry<LIGHT>(v & 0xFFFFu, is_dc), hi = convert_entry<LIGHT>(v >> 16, is_dc);
            /* first level of an AC table: groups where they fit.  Not for streams of long units (quality 100):
             * their short symbols sit at every k, the single-symbol path from global memory would be taken all
             * the time, and their batches are a matter of one lane's latency (config 2: sync 1.36 -> 1.80 ms) */
            if (LIGHT && !is_dc && !im.long_units && i < (1u << JD_LUT_BITS) / 2u) {
                lo = pair_entry(lo, 2u * i, g.hot.primary);
                hi = pair_entry(hi, 2u * i + 1u, g.hot.primary);
            }
            dst[slot * HOT32 + i] = lo | (hi << 16);
        }
    }
    if (threadIdx.x < JD_MAX_BPM) {
        const uint32_t dcs = im.blk_dc[threadIdx.x], acs = im.blk_ac[threadIdx.x];
        const uint32_t c = im.blk_comp[threadIdx.x];
        const int sel = 1;
        BlkDesc d;
        d.dc_off = dcs * (uint32_t)sizeof(jd_lut_hot); d.ac_off = acs * (uint32_t)sizeof(jd_lut_hot);
        d.slots = dcs | (acs << 4);
        d.nb = threadIdx.x + 1u == im.bpm ? 0u : threadIdx.x + 1u;
        d.e0 = c == 0 ? sel : 0; d.e1 = c == 1 ? sel : 0; d.e2 = c == 2 ? sel : 0;
        d.pad = 0;
        s_blk[threadIdx.x] = d;
    }
}

/* the (image, first segment) descriptor of this wave, in scalar registers: everything derived
 * from it (image record, stream base) is then read and computed once per wave, not per lane */
/* Work-groups go to the 8 XCDs round robin (blockIdx.x % 8) and no kernel here shares data between
 * work-groups; giving every XCD a contiguous eighth of the work list instead was measured and lost
 * ("Work-group order and the 8 XCDs") */
__device__ __forceinline__ uint32_t wg_index() { return blockIdx.x; }
__device__ __forceinline__ jd_huff_wg wave_desc(const jd_huff_wg *__restrict__ wgs, uint32_t waves) {
    const jd_huff_wg w = wgs[wg_index() * waves + (threadIdx.x >> 6)];
    jd_huff_wg u;
    u.img = __builtin_amdgcn_readfirstlane(w.img);
    u.seg0 = __builtin_amdgcn_readfirstlane(w.seg0);
    return u;
}

/*
 * Start states.  Three kernels, all one lane per segment, work-groups of JD_WG_SEGS segments of
 * one image (they share the image's lookup tables in LDS):
 *
 *   k_huff_first   every segment that does not start a restart interval guesses its start state
 *                  by decoding the last JD_OVERLAP bits of its predecessor from "expecting DC,
 *                  data unit 0" (JPEG Huffman streams self-synchronise), then decodes itself
 *                  from that state: exit state, data units completed, DC difference sums.
 *   k_chain_check  a segment whose start state is not its predecessor's exit state goes on the
 *                  image's repair list.
 *   k_huff_repair  decodes the listed segments again, packed densely into lanes, from the
 *                  predecessor's exit state; stops early at a checkpoint the earlier decode
 *                  passed in the same state.
 *
 * check + repair run a fixed number of rounds (a repaired segment seldom ends in a new state, so
 * the lists shrink by ~50x per round); k_huff_write verifies the chain once more and the host
 * adds rounds in the rare case that it is still broken.  Repairs are a separate launch rather
 * than a loop inside the first kernel so that no wave sits at a barrier, holding its SIMD slot,
 * while a few lanes of its work-group decode again (measured: 5.6 -> 3 ms per 2048 images).
 *
 * st_in / st_exit / cnt: per segment the start state used, the exit state reached, and
 * (data units completed | JD_CNT_WIDE, DC difference sums per component).
 */
__global__ __launch_bounds__(JD_WG_SEGS) void k_huff_first(const jd_image *__restrict__ imgs,
                                                           const jd_lut *__restrict__ luts,
                                                           const uint8_t *__restrict__ clean,
                                                           const uint32_t *__restrict__ seg_info,
                                                           const jd_huff_wg *__restrict__ wgs,
                                                           uint32_t *__restrict__ st_in,
                                                           uint32_t *__restrict__ st_exit,
                                                           int4 *__restrict__ cnt,
                                                           uint32_t *__restrict__ cp_state,
                                                           int4 *__restrict__ cp_val) {
    __shared__ __attribute__((aligned(16))) jd_lut_hot s_lut[JD_LUT_SLOTS];
    __shared__ BlkDesc s_blk[JD_MAX_BPM];

    /* one descriptor per wave; the waves of a work-group may serve different images, whose
     * tables and unit layout the plan made sure are the same (staged from the first wave's) */
    const jd_huff_wg wg = wave_desc(wgs, JD_WG_WAVES);
    const jd_image &im = imgs[wg.img];
    const uint32_t n_segs = im.n_segs;
    if (!__syncthreads_or(wg.seg0 < n_segs)) return; /* seg_cap is an upper bound: trailing groups are empty */
    const jd_image &im0 = imgs[wgs[wg_index() * JD_WG_WAVES].img];
    stage_tables<true>(s_lut, s_blk, luts, im0);
    __syncthreads();
    if (wg.seg0 >= n_segs) return;
    /* a lane takes first_span consecutive segments: the first from a guessed start state, the
     * others from the exit state of the one before -- their start is then right unless the guess
     * was wrong AND the decode did not find back within a whole segment (~1 % on C3, against
     * ~30 % for a guess), and they need no overlap decode */
    const uint32_t span = im.first_span;   /* 1 in small images (jd_plan_build) */
    uint32_t seg = wg.seg0 + (threadIdx.x & 63u) * span;
    if (seg >= n_segs) return;
    const SegTables T{reinterpret_cast<const uint8_t *>(s_lut), luts, im.lut_idx, s_blk, im.bpm};
    const uint8_t *const cimg = clean + im.clean_off;
    uint32_t prev_exit = 0;
#pragma unroll 1
    for (uint32_t h = 0; h < span && seg < n_segs; ++h, ++seg) {
        const uint32_t gs = im.seg_off + seg;
        const uint32_t si = seg_info[gs];
        uint32_t in = 0; /* the true start state of a restart interval */
        if (!(si & JD_SEG_FIRST)) {
            if (h == 0) {
                const uint32_t pv = seg_info[gs - 1] & SEG_MASK; /* JD_SEG_BITS: only the last segment of an interval is short */
                const uint32_t ov = span > 1u ? JD_OVERLAP_SPAN : JD_OVERLAP;
                const uint32_t p0 = pv > ov ? pv - ov : 0u;
                in = im.long_units ? decode_state_only<false>(make_reader(cimg, seg - 1, n_segs), T, p0, pv)
                                   : decode_state_only<true>(make_reader(cimg, seg - 1, n_segs), T, p0, pv);
            } else {
                in = prev_exit & ~JD_ST_BAD;
            }
        }
        const LightResult r = im.long_units   /* wave-uniform: a scalar branch */
            ? decode_light<false, true>(make_reader(cimg, seg, n_segs), T, in, si & SEG_MASK,
                                                  cp_state + (size_t)gs * JD_CP, cp_val + (size_t)gs * (JD_CP + 1u), 0u)
            : decode_light<false, false>(make_reader(cimg, seg, n_segs), T, in, si & SEG_MASK,
                                                   cp_state + (size_t)gs * JD_CP, cp_val + (size_t)gs * (JD_CP + 1u), 0u);
        st_in[gs] = in;
        st_exit[gs] = r.exit_state;
        /* with the unit count: did the segment hold an AC coefficient of 8 or more magnitude bits
         * (|value| >= 128)?  Such an image's units do not fit the write kernel's int8 buffers.  Kept
         * per checkpoint interval, so that a repair that stops at a checkpoint takes the flags of
         * the part it did not decode again from the decode that did; k_seg_scan gathers them once
         * the start states are final (a decode from a wrong start state sees garbage symbols). */
        cnt[gs] = make_int4(r.n, r.dc0, r.dc1, r.dc2);
        prev_exit = r.exit_state;
    }
}

__global__ __launch_bounds__(JD_WG_SEGS) void k_chain_check(const jd_image *__restrict__ imgs,
                                                            const uint32_t *__restrict__ seg_info,
                                                            const jd_huff_wg *__restrict__ wgs,
                                                            const uint32_t *__restrict__ st_in,
                                                            const uint32_t *__restrict__ st_exit,
                                                            uint32_t *__restrict__ list,
                                                            uint32_t *__restrict__ list_cnt,
                                                            const uint32_t *__restrict__ prev_cnt /* null in the first round of a pass */) {
    const jd_huff_wg wg = wave_desc(wgs, JD_WG_WAVES);
    /* a segment's start can only break when its predecessor was decoded again, i.e. was on the
     * previous round's list: an image whose list was empty then has nothing to check now */
    if (prev_cnt != nullptr && prev_cnt[wg.img] == 0u) return;
    const jd_image &im = imgs[wg.img];
    const uint32_t seg = wg.seg0 + (threadIdx.x & 63u);
    bool broken = false;
    if (wg.seg0 < im.n_segs && seg < im.n_segs) {
        const uint32_t gs = im.seg_off + seg;
        const uint32_t want = (seg_info[gs] & JD_SEG_FIRST) ? 0u : (st_exit[gs - 1] & ~JD_ST_BAD);
        broken = st_in[gs] != want;
    }
    const uint64_t bm = __ballot(broken);
    if (bm) {
        const uint32_t wlane = threadIdx.x & 63u;
        uint32_t base = 0;
        if (wlane == (uint32_t)__ffsll((long long)bm) - 1u) base = atomicAdd(&list_cnt[wg.img], (uint32_t)__popcll(bm));
        base = __shfl(base, __ffsll((long long)bm) - 1);
        if (broken) list[im.seg_off + base + (uint32_t)__popcll(bm & ((1ull << wlane) - 1ull))] = seg;
    }
}

__global__ __launch_bounds__(JD_WG_SEGS) void k_huff_repair(const jd_image *__restrict__ imgs,
                                                            const jd_lut *__restrict__ luts,
                                                            const uint8_t *__restrict__ clean,
                                                            const uint32_t *__restrict__ seg_info,
                                                            const jd_huff_wg *__restrict__ wgs,
                                                            const uint32_t *__restrict__ list,
                                                            const uint32_t *__restrict__ list_cnt,
                                                            uint32_t *st_in,
                                                            uint32_t *st_exit, /* read (predecessors) and written: no __restrict__ */
                                                            int4 *__restrict__ cnt,
                                                            uint32_t *__restrict__ cp_state,
                                                            int4 *__restrict__ cp_val) {
    __shared__ __attribute__((aligned(16))) jd_lut_hot s_lut[JD_LUT_SLOTS];
    __shared__ BlkDesc s_blk[JD_MAX_BPM];

    const jd_huff_wg wg = wave_desc(wgs, JD_WG_WAVES); /* here: entries [seg0, seg0 + 64) of the image's list */
    const jd_image &im = imgs[wg.img];
    const uint32_t n_list = min(list_cnt[wg.img], im.n_segs);
    if (!__syncthreads_or(wg.seg0 < n_list)) return; /* the usual case from the second round on */
    const jd_image &im0 = imgs[wgs[wg_index() * JD_WG_WAVES].img];
    stage_tables<true>(s_lut, s_blk, luts, im0);
    __syncthreads();
    if (wg.seg0 >= n_list) return;
    const uint32_t idx = wg.seg0 + (threadIdx.x & 63u);
    if (idx >= n_list) return;
    const SegTables T{reinterpret_cast<const uint8_t *>(s_lut), luts, im.lut_idx, s_blk, im.bpm};
    const uint32_t seg = list[im.seg_off + idx];
    const uint32_t gs = im.seg_off + seg;
    /* a listed segment never starts an interval, so it has a predecessor.  That one may be on
     * the list too and change its exit state while this lane reads it: either value is a
     * consistent start, and the next check lists this segment again if it was the old one. */
    const uint32_t in = st_exit[gs - 1] & ~JD_ST_BAD;
    const uint32_t old_exit = st_exit[gs] & ~JD_ST_BAD;
    const LightResult r = im.long_units
        ? decode_light<true, true>(make_reader(clean + im.clean_off, seg, im.n_segs), T, in, seg_info[gs] & SEG_MASK,
                                             cp_state + (size_t)gs * JD_CP, cp_val + (size_t)gs * (JD_CP + 1u), old_exit)
        : decode_light<true, false>(make_reader(clean + im.clean_off, seg, im.n_segs), T, in, seg_info[gs] & SEG_MASK,
                                              cp_state + (size_t)gs * JD_CP, cp_val + (size_t)gs * (JD_CP + 1u), old_exit);
    st_in[gs] = in;
    st_exit[gs] = r.exit_state;
    cnt[gs] = make_int4(r.n, r.dc0, r.dc1, r.dc2);
}

/* ============================================================================================
 * scan: first data unit and DC predictors at every segment start
 * ========================================================================================== */

struct ScanVal { int n, d0, d1, d2; uint32_t f; }; /* f: number of interval starts covered */

__device__ __forceinline__ ScanVal scan_combine(const ScanVal &a, const ScanVal &b) {
    /* b follows a; an interval start inside b resets the running sums */
    ScanVal r;
    if (b.f) { r = b; r.f = a.f + b.f; }
    else { r.n = a.n + b.n; r.d0 = a.d0 + b.d0; r.d1 = a.d1 + b.d1; r.d2 = a.d2 + b.d2; r.f = a.f; }
    return r;
}

__device__ __forceinline__ ScanVal shfl_up_val(const ScanVal &v, int d) {
    ScanVal o;
    o.n = __shfl_up(v.n, d); o.d0 = __shfl_up(v.d0, d); o.d1 = __shfl_up(v.d1, d);
    o.d2 = __shfl_up(v.d2, d); o.f = __shfl_up(v.f, d);
    return o;
}

/* one work-group per image; inclusive segmented scan in chunks of 256 with a running carry */
__global__ __launch_bounds__(256) void k_seg_scan(const jd_image *__restrict__ imgs,
                                                  const uint32_t *__restrict__ seg_info,
                                                  const int4 *__restrict__ cnt,
                                                  int2 *__restrict__ start_blk,
                                                  int4 *__restrict__ start_dc,
                                                  uint32_t *__restrict__ status,
                                                  uint32_t *__restrict__ wide) {
    __shared__ ScanVal s_wave[4];
    __shared__ ScanVal s_carry;
    const jd_image &im = imgs[blockIdx.x];
    const uint32_t n_segs = im.n_segs, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_carry.n = s_carry.d0 = s_carry.d1 = s_carry.d2 = 0; s_carry.f = 0; }
    if (threadIdx.x == 0 && im.ncomp && n_segs == 0) atomicOr(&status[blockIdx.x], (uint32_t)JDGPU_IMG_EOF);
    __syncthreads();
    for (uint32_t base = 0; base < n_segs; base += 256) {
        uint32_t seg = base + threadIdx.x;
        bool act = seg < n_segs;
        ScanVal v; v.n = v.d0 = v.d1 = v.d2 = 0; v.f = 0;
        bool first = false, isw = false;
        if (act) {
            int4 c = cnt[im.seg_off + seg];
            first = (seg_info[im.seg_off + seg] & JD_SEG_FIRST) != 0;
            v.n = c.x & (JD_CNT_WIDE - 1); v.d0 = c.y; v.d1 = c.z; v.d2 = c.w; v.f = first ? 1u : 0u;
            isw = (c.x & JD_CNT_WIDE) != 0;
        }
        if ((__any(isw) || (base == 0 && wave == 0 && im.pre_wide)) && lane == 0) atomicOr(&wide[blockIdx.x], 1u); /* see k_huff_first */
        ScanVal own = v;
        /* wave-level inclusive scan (Hillis-Steele over 64 lanes) */
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            ScanVal o = shfl_up_val(v, d);
            if ((int)lane >= d) v = scan_combine(o, v);
        }
        if (lane == 63) s_wave[wave] = v;
        __syncthreads();
        ScanVal pre = s_carry;
        for (uint32_t w = 0; w < wave; ++w) pre = scan_combine(pre, s_wave[w]);
        ScanVal inc = scan_combine(pre, v); /* inclusive over everything up to this segment */
        __syncthreads();
        if (threadIdx.x == 255) s_carry = inc;
        if (act) {
            /* exclusive value: state before this segment's own contribution */
            int ivl = (int)inc.f - 1; /* interval index of this segment */
            int ri = (int)im.ri_blocks;
            int ibase = ri ? ivl * ri : 0;
            int limit = ri ? min(ibase + ri, (int)im.n_blocks) : (int)im.n_blocks;
            int b0, p0, p1, p2;
            if (first) { b0 = ibase; p0 = p1 = p2 = 0; }
            else { b0 = ibase + inc.n - own.n; p0 = inc.d0 - own.d0; p1 = inc.d1 - own.d1; p2 = inc.d2 - own.d2; }
            start_blk[im.seg_off + seg] = make_int2(b0, limit);
            start_dc[im.seg_off + seg] = make_int4(p0, p1, p2, 0);
            /* fewer restart markers in the stream than the geometry needs: the trailing
             * intervals have no segment and so no lane that could notice; their data units
             * are missing -- "Reached end of file." (src/bitstream.c:17) */
            if (seg + 1 == n_segs && inc.f < im.n_intervals) atomicOr(&status[blockIdx.x], (uint32_t)JDGPU_IMG_EOF);
        }
        __syncthreads();
    }
}

/* ============================================================================================
 * write: final decode, coefficients out
 * ========================================================================================== */

__constant__ uint8_t c_zz_cm[64];

/*
 * k_huff_write: one lane per segment again, now from the true start state.  A data unit is
 * written entirely by the lane in whose segment it STARTS (that lane runs on into the next
 * segment to finish its last unit; a lane that starts in the middle of a unit decodes the rest
 * of it without storing).  Coefficients are assembled in a per-lane LDS buffer (128 bytes of
 * int16, or 64 bytes of int8 in the NARROW instantiation, where the DC value travels through the
 * flush queue).  A loop iteration decodes up to JD_WSTEPS symbols per lane -- a lane that has
 * completed a unit it owns pauses for the rest of the iteration, its buffer still holds that
 * unit -- and then the wave flushes the units completed in the iteration together: 8 lanes per
 * wide unit, 4 per narrow one, write it with one 16-byte store each (whole 64-byte requests instead
 * of ~12 scattered 2-byte stores per unit) and clear the buffer.  A wide unit leaves as 128 bytes of int16; a narrow one as
 * JD_NARROW_UNIT = 64 bytes: int8 coefficients, the DC value in 12 bits (byte 0 and the low nibble
 * of byte 63) and the last AC coefficient in the 4 bits left of byte 63.  The flush lanes check
 * those two ranges; an image that breaks them is marked wide and the host runs the pass again,
 * like one whose AC values the light decodes failed to flag.  Half the bytes matter because the
 * kernel is limited by the per-CU path to L2 (TA / L1: TCP_PENDING_STALL_CYCLES 84 % of its
 * time), whose traffic was two 64-byte write requests per unit plus the stream reads.
 * (Measured and rejected in round 2: DC differences into a side array, one 4-byte store per unit,
 * resolved by a scan kernel afterwards so that no decode has to read DC magnitudes: the 200 M
 * partial-line stores of a 4096 x 1080p batch cost 4.4 ms.  4 lanes per unit with two stores
 * each: 9.9 against 8.6 ms.  Only whole passes of 8 units, the rest waiting an iteration: 9.0
 * against 8.6 ms.  Stream loads in inline assembly with a hand-counted s_waitcnt, so that the
 * refill does not wait for the flush stores' acknowledgements: no change.  16-byte and 8-byte
 * stream loads into a register queue: more registers, 8.9 and 8.3 against 7.9 and 7.3 ms.)
 */
#define WBUF_BYTES 128u
#define JD_NARROW_UNIT 64u   /* bytes of a data unit in the coefficient buffer of a narrow image (a wide one: 128) */
#ifndef JD_WSTEPS
#define JD_WSTEPS JD_RSTEPS   /* symbol steps per iteration of the write kernel; the reader is refilled after every second one */
#endif
__device__ __forceinline__ uint32_t wbuf_key(uint32_t lane) {
    /* 16-byte chunks permuted by the lane number (XOR on the chunk index): both the lanes'
     * scattered stores and the 8-lanes-per-unit flush spread over the LDS banks, and the address of
     * coefficient idx is ONE instruction: (idx * 2) ^ key */
    return lane * WBUF_BYTES + ((lane & 7u) << 4);
}

/* NARROW: 64-byte unit buffers of int8 (24 instead of 16 waves per CU with 8-wave work-groups)
 * for images none of whose AC coefficients needs more than 7 magnitude bits (the light decodes
 * tell, `wide`).  The wide instantiation is launched beside it and takes the other images; each
 * skips the waves that are not its own. */
__device__ __forceinline__ uint32_t nbuf_key(uint32_t lane) { return lane * 64u + ((lane & 7u) << 3); } /* address = idx ^ key */
/* SUB = JD_CP + 1 (small batches, jd_plan::write_sub): SUB lanes per segment.  Lane q > 0 starts where
 * the light decodes recorded checkpoint q - 1 -- state, and data units / DC differences counted up to
 * there (cp_state, cp_val: final once the chain of start states is) -- and owns the units that start
 * before checkpoint q; lane 0 those before checkpoint 0.  The write stage of a batch that leaves the
 * chip idle is as long as one lane's chain of symbols: a quarter of it this way. */
template <uint32_t WAVES, uint32_t SLOTS, bool NARROW, bool SPLIT, uint32_t SUB = 1u>
__global__ __launch_bounds__(64 * WAVES) void k_huff_write(const jd_image *__restrict__ imgs,
                                                           const jd_lut *__restrict__ luts,
                                                           const uint8_t *__restrict__ clean,
                                                           const uint32_t *__restrict__ seg_info,
                                                           const jd_huff_wg *__restrict__ wgs,
                                                           const uint32_t *__restrict__ st_in,
                                                           const uint32_t *__restrict__ st_exit,
                                                           const int2 *__restrict__ start_blk,
                                                           const int4 *__restrict__ start_dc,
                                                           int16_t *__restrict__ coef,
                                                           uint32_t *__restrict__ status,
                                                           uint32_t *__restrict__ wide,
                                                           const uint32_t *__restrict__ redo /* a later pass: the images to do again; null = all */,
                                                           const uint32_t *__restrict__ cp_state,
                                                           const int4 *__restrict__ cp_val) {
    static_assert(JD_SEG_BITS >= 2048, "a data unit (<= 64 * 26 bits) must fit in one segment");
    static_assert(SUB == 1u || SUB == JD_CP + 1u, "one lane per segment, or one per checkpoint interval");
    /* <4 waves, 6 table slots>: 3 work-groups = 12 waves per CU wide, 4 narrow; <8, 4> (no image of
     * the batch uses more than 4 slots): 2 work-groups = 16 waves per CU wide, 3 = 24 narrow */
    __shared__ __attribute__((aligned(16))) jd_lut_hot s_lut[SLOTS];
    constexpr uint32_t UB = NARROW ? 64u : WBUF_BYTES;
    __shared__ __attribute__((aligned(16))) uint8_t s_buf[64 * WAVES * UB];
    /* flush queue, per wave: the r-th finished unit = {global unit index, lane that holds it | DC << 16 (NARROW)} */
    __shared__ uint2 s_q[64 * WAVES];
    __shared__ BlkDesc s_blk[JD_MAX_BPM];
    __shared__ uint8_t s_cm[64];

    const jd_huff_wg wg = wave_desc(wgs, WAVES); /* per wave, see k_huff_first */
    const jd_image &im = imgs[wg.img];
    const uint32_t n_segs = im.n_segs;
    /* SPLIT: narrow and wide instantiation run side by side, each on its own images */
    const bool mine = (!SPLIT || ((wide[wg.img] != 0u) != NARROW)) && (!redo || redo[wg.img] != 0u);
    if (!__syncthreads_or(wg.seg0 < n_segs && mine)) return;
    const uint32_t lane = threadIdx.x, wlane = lane & 63u, wbase = lane & ~63u;
    const uint32_t seg = wg.seg0 + wlane / SUB, sub = wlane % SUB;
    {
        const jd_image &im0 = imgs[wgs[wg_index() * WAVES].img];
        stage_tables<false>(s_lut, s_blk, luts, im0, SLOTS);
    }
    if (lane < 64) s_cm[lane] = c_zz_cm[lane];
    {
        uint4 *z = reinterpret_cast<uint4 *>(s_buf + lane * UB);
#pragma unroll
        for (uint32_t q = 0; q < UB / 16u; ++q) z[q] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    bool active = wg.seg0 < n_segs && seg < n_segs && mine;
    if (!__any(active)) return; /* whole wave beyond the image: no further work-group barrier below */

    const uint32_t gs = im.seg_off + (active ? seg : wg.seg0);
    const uint32_t si = seg_info[gs];
    const bool first = (si & JD_SEG_FIRST) != 0;
    const uint32_t valid = si & SEG_MASK;
    /* bits this lane may consume: its own segment plus the next one of the same interval */
    uint32_t avail = valid;
    bool is_last = true; /* last segment of its restart interval */
    if (active && seg + 1 < n_segs) {
        uint32_t sn = seg_info[gs + 1];
        if (!(sn & JD_SEG_FIRST)) { is_last = false; avail += sn & SEG_MASK; }
    }
    const uint32_t in = st_in[gs];
    uint32_t flags = 0;
    /* chain check: the state this segment starts from must be the one its predecessor ended in
     * (it is not when a correction crossed a whole work-group: the host then stitches again) */
    if (active && sub == 0u && !first && seg > 0 && ((st_exit[gs - 1] ^ in) & ~JD_ST_BAD)) flags |= JDGPU_IMG_UNSYNCED;
    if (active && sub == 0u && first && in != 0) flags |= JDGPU_IMG_UNSYNCED;

    const int2 sb = start_blk[gs];
    const int4 sd = start_dc[gs];
    int B = sb.x;
    const int limit = sb.y;
    int pr0 = sd.x, pr1 = sd.y, pr2 = sd.z;

    uint32_t p = in & 63u, k = (in >> 6) & 127u, b = JD_ST_UNIT(in);
    uint32_t vend = valid;      /* units that start before this bit are the lane's */
    bool last_sub = true;       /* the lane that reaches the end of the segment's data */
    if (SUB > 1u) {
        const uint32_t *cs = cp_state + (size_t)gs * JD_CP;
        const int4 *cv = cp_val + (size_t)gs * (JD_CP + 1u);
        if (sub > 0u) {
            const uint32_t st = cs[sub - 1u];
            if (st == JD_CP_NONE) active = false;   /* the segment ends before this lane's checkpoint */
            else { p = st & 0xFFFFu; k = (st >> 16) & 255u; b = st >> 24; }
            for (uint32_t i = 0; i < sub; ++i) {
                const int4 v = cv[i];
                B += v.x & (JD_CNT_WIDE - 1); pr0 += v.y; pr1 += v.z; pr2 += v.w;
            }
        }
        if (sub + 1u < SUB) {
            const uint32_t nx = cs[sub];
            if (nx != JD_CP_NONE) { vend = min(valid, nx & 0xFFFFu); last_sub = false; }
        }
    }
    BitReader R = make_reader(clean + im.clean_off, active ? seg : wg.seg0, n_segs);
    R.init(active ? p : 0u);
    const SegTables T{reinterpret_cast<const uint8_t *>(s_lut), luts, im.lut_idx, s_blk, 0u};
    BlkDesc tb = s_blk[b];
    bool own = (k == 0); /* does this lane own the data unit it is in? */
    int dcv = 0;         /* NARROW: DC value of the unit in progress */
    uint32_t ovf = 0;    /* NARROW: an AC value that does not fit (cannot happen when `wide` is complete); wide: a DC value beyond int16 */
    uint32_t ovf2 = 0;   /* NARROW: a DC value beyond 12 bits or a last AC coefficient beyond 4, seen by a flush lane */
    int16_t *const cbase = coef + (size_t)im.blk_off * 64;
    const uint32_t bkey = NARROW ? nbuf_key(lane) : wbuf_key(lane);
    /* flush roles (wide units: eighth of the unit; narrow ones: a quarter, computed at the flush) */
    constexpr uint32_t FL = NARROW ? 4u : 8u;        /* flush lanes per unit, 16 bytes each */
    const uint32_t qsub = (wlane & 7u) * 16u, osub = (wlane & 7u) * 4u;   /* wide units: byte offset in the unit buffer, dword offset in the line */
    const uint32_t dc_on = (wlane & (FL - 1u)) == 0u ? ~0u : 0u, top_on = (wlane & (FL - 1u)) == FL - 1u ? ~0u : 0u;
    const uint32_t dcmask = dc_on & 0xFFu, topmask = top_on & 0xFF000000u;

    /* a unit is decoded while it STARTED before the end of the segment */
    active = active && (k != 0 || p < vend) && B < limit;
    /* Everything the loop reads that came from memory is "used" here, so that the compiler waits
     * for those loads BEFORE the loop.  gfx9 counts loads and stores in one in-order counter: a
     * wait for a pre-loop load placed inside the loop is a wait for all but the newest few memory
     * operations, i.e. for the flush stores of the previous iteration, in every iteration. */
    asm volatile("" ::"v"(avail), "v"(valid), "v"(limit), "v"(B), "v"(p), "v"(k), "v"(b), "v"(R.b_own), "v"(R.b_next), "v"(cbase), "v"(pr0), "v"(pr1), "v"(pr2));
    while (__any(active)) {
        bool finished = false; /* an OWNED unit was completed in this iteration */
        int fB = 0;            /* ... and this is its index */
#pragma unroll
        for (int s = 0; s < JD_WSTEPS; ++s) {
            /* the first step takes any symbol, the others AC symbols only (no predictors, no table
             * select): a lane that completes a unit it owns pauses until the flush anyway, so only the
             * lane that ends the unit it started in the middle of loses a step to this */
            if (active && !finished && (s == 0 || k != 0u)) {
                const uint32_t top = R.top();
                const bool isdc = s == 0 && k == 0u;
                const uint32_t e = lookup_entry<false>(T, tb, isdc, top);
                const uint32_t len = (e & 15u) + 1u, size = (e >> 4) & 15u, kinc = (e >> 8) & 127u;
                const uint32_t tot = len + size;
                /* src/bitstream.c:17 (the data ends inside this symbol) before src/hufftree.c:69,72
                 * (no such code): the reference reads bit by bit and meets the end first.  The flags
                 * are set in a branch that is almost never taken; `ok` gates the side effects. */
                const bool ok = p + tot <= avail && kinc != 0u;
                if (__builtin_expect(!ok, 0)) flags |= (p + tot > avail) ? (uint32_t)JDGPU_IMG_EOF : (uint32_t)JDGPU_IMG_NOSYMBOL;
                int val = extend_at(top, tot, size);
                /* DC: predictor of the unit's component (src/decoding.c:44-46; quantised, the
                 * multiply is in k_pixel); e0..e2 are 1 for the unit's component, else 0 */
                const int d = isdc ? val : 0;
                pr0 = mad24(d, tb.e0, pr0); pr1 = mad24(d, tb.e1, pr1); pr2 = mad24(d, tb.e2, pr2);
                const int pv = mad24(pr0, tb.e0, mad24(pr1, tb.e1, __mul24(pr2, tb.e2)));
                val = isdc ? pv : val;
                const uint32_t kk = k + kinc - 1u;            /* zigzag index of this symbol's coefficient (DC: 0) */
                if (NARROW) {
                    dcv = isdc ? val : dcv;
                    /* kk in 1..63: an AC coefficient inside the unit (src/decoding.c:65; a run past 63 is dropped) */
                    if (ok && own && size != 0u && kk - 1u < 63u) {
                        ovf |= (uint32_t)(val + 128);
                        *reinterpret_cast<int8_t *>(s_buf + (s_cm[kk] ^ bkey)) = (int8_t)val;
                    }
                } else {
                    /* a DC value that leaves int16 (the reference keeps it in an int, src/decoding.c:44-46; no
                     * stream of 8-bit samples gets there): reported, not wrapped -- JDGPU_IMG_RANGE below.  AC
                     * values have at most 15 magnitude bits and always fit */
                    ovf |= (isdc && ok) ? (uint32_t)(val + 32768) : 0u;
                    if (ok && (isdc || (own && size != 0u && kk < 64u)))
                        *reinterpret_cast<int16_t *>(s_buf + ((2u * s_cm[kk & 63u]) ^ bkey)) = (int16_t)val;
                }
                p += tot;
                k += kinc;
                R.consume(tot);
                const bool end = ok && k >= 64u;
                const uint32_t nb = tb.nb;
                finished = end && own;
                fB = B;
                B += end ? 1 : 0; own = own || end; k = end ? 0u : k; b = end ? nb : b;
                active = ok && (k != 0 || p < vend) && B < limit;
            }
            if ((s & 1) && s + 1 < JD_WSTEPS) R.refill();   /* more than two steps: the window holds two symbols */
        }
        tb = s_blk[b]; /* once per iteration, all lanes: no divergent merge copies the descriptor around */
        R.refill(); /* all lanes, outside the divergent steps */
        /* ---- flush the units completed in this iteration (wave-synchronous) ---- */
        const uint64_t fm = __ballot(finished);
        if (fm) {
            if (finished) {
                uint32_t r = __popcll(fm & ((1ull << wlane) - 1ull));
                s_q[wbase + r] = make_uint2((uint32_t)fB, wlane | (NARROW ? (uint32_t)dcv << 16 : 0u)); /* dcv is only replaced by the next unit's DC step */
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t nf = __popcll(fm);
            for (uint32_t g = wlane / FL; g < nf; g += 64u / FL) {   /* FL lanes per unit, one store each: a whole unit per group of lanes */
                const uint2 qe = s_q[wbase + g];
                const uint32_t src = wbase + (qe.y & 63u), blk = qe.x;
                if (NARROW) {
                    /* the unit leaves as 64 bytes (JD_NARROW_UNIT, see narrow_unit_load): int8 coefficients,
                     * the 12-bit DC value in byte 0 and the low nibble of byte 63, the last AC coefficient in
                     * the high nibble of byte 63.  Four lanes, 16 bytes each (round 3: half as many store
                     * instructions per unit as 8 x 8 bytes, write 7.35 -> 7.22 ms per 4096 x 1080p).  The
                     * unit buffer's 8-byte pieces are permuted by the low three bits of the owning lane (nbuf_key):
                     * pieces 2r and 2r + 1 are the aligned pair r ^ (m >> 1), swapped when m is odd */
                    const uint32_t r = wlane & 3u, m = src & 7u;
                    uint4 *lp = reinterpret_cast<uint4 *>(s_buf + (src * 64u + ((r ^ (m >> 1)) << 4)));
                    const uint4 v = *lp;
                    *lp = make_uint4(0, 0, 0, 0);
                    const bool sw = (m & 1u) != 0u;
                    uint4 n16 = make_uint4(sw ? v.z : v.x, sw ? v.w : v.y, sw ? v.x : v.z, sw ? v.y : v.w);
                    const int dc = (int)qe.y >> 16;
                    const int a63 = (int)n16.w >> 24;
                    ovf2 |= ((uint32_t)(dc + 2048) & ~4095u & dc_on) | ((uint32_t)(a63 + 8) & ~15u & top_on);
                    n16.x = (n16.x & ~dcmask) | ((uint32_t)dc & dcmask);
                    const uint32_t top = (((uint32_t)dc << 16) & 0x0F000000u) | ((n16.w << 4) & 0xF0000000u);
                    n16.w = (n16.w & ~topmask) | (top & topmask);
                    uint32_t *out = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(cbase) + (size_t)blk * JD_NARROW_UNIT) + r * 4u;
                    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store((u4){n16.x, n16.y, n16.z, n16.w}, reinterpret_cast<u4 *>(out));
                } else {
                    uint4 *lp = reinterpret_cast<uint4 *>(s_buf + (qsub ^ wbuf_key(src)));
                    const uint4 v = *lp;
                    *lp = make_uint4(0, 0, 0, 0);
                    /* streamed: nothing on this CU reads the unit again (6.18 -> 6.03 ms per 2048 images) */
                    uint32_t *out = reinterpret_cast<uint32_t *>(cbase + (size_t)blk * 64) + osub;
                    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store((u4){v.x, v.y, v.z, v.w}, reinterpret_cast<u4 *>(out));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    /* data ended before the interval's data units did */
    if (!flags && seg < n_segs && B < limit && is_last && last_sub && (SUB == 1u || sub == 0u || cp_state[(size_t)gs * JD_CP + sub - 1u] != JD_CP_NONE))
        flags |= JDGPU_IMG_EOF;
    if (NARROW && ((ovf & ~255u) | ovf2)) { /* safety net: mark the image wide and have the host run the pass again */
        atomicOr(&wide[wg.img], 1u);
        flags |= JDGPU_IMG_UNSYNCED;
    }
    if (!NARROW && (ovf & ~65535u)) flags |= JDGPU_IMG_RANGE;
    if (flags) atomicOr(&status[wg.img], flags);
}

/* ============================================================================================
 * pixel stage
 * ========================================================================================== */

/* src/decoding.c:20-29: C(u) cos((2x+1) u pi/16) rounded to six decimals */
#define T0 0.707107
#define T1 0.980785
#define T2 0.923880
#define T3 0.831470
#define T5 0.555570
#define T6 0.382683
#define T7 0.195090

/* 8-point inverse transform with the reference's table, even/odd split:
 * out[x] = S * sum_u in[u] * T[u][x];  T[u][7-x] = (-1)^u T[u][x].  S is a power of two, so
 * scaling the constants is exact and commutes with every rounding.  Both passes run with S = 1:
 * the reference's final "sum / 4" (src/decoding.c:83) is taken by the rounding add (to_int128) --
 * one set of constants in scalar registers, which ended the kernel's scalar register spills
 * (10.4 -> 10.0 ms). */
/* pix_real: the arithmetic type of the IDCT: double, byte-identical to the reference on every fixture
 * (an all-float build was measured in round 2: -3.9 % of the stage, 7e-6 of the samples off by one, some
 * pixels by two: DESIGN.md section 4) */
typedef double pix_real;
template <int QUARTER>
__device__ __forceinline__ void idct8(const pix_real in[8], pix_real out[8]) {
    constexpr pix_real S = QUARTER ? 0.25 : 1.0;
    pix_real e0 = in[0] * (pix_real)(T0 * S);
    pix_real a0 = fma(in[4], (pix_real)(T0 * S), e0), a1 = fma(in[4], (pix_real)(-T0 * S), e0);   /* u = 0, 4 */
    pix_real b0 = fma(in[2], (pix_real)(T2 * S), in[6] * (pix_real)(T6 * S));              /* u = 2, 6 at x = 0 */
    pix_real b1 = fma(in[2], (pix_real)(T6 * S), -(in[6] * (pix_real)(T2 * S)));           /* at x = 1 */
    pix_real E0 = a0 + b0, E3 = a0 - b0, E1 = a1 + b1, E2 = a1 - b1;
    pix_real O0 = fma(in[1], (pix_real)(T1 * S), fma(in[3], (pix_real)(T3 * S), fma(in[5], (pix_real)(T5 * S), in[7] * (pix_real)(T7 * S))));
    pix_real O1 = fma(in[1], (pix_real)(T3 * S), fma(in[3], (pix_real)(-T7 * S), fma(in[5], (pix_real)(-T1 * S), in[7] * (pix_real)(-T5 * S))));
    pix_real O2 = fma(in[1], (pix_real)(T5 * S), fma(in[3], (pix_real)(-T1 * S), fma(in[5], (pix_real)(T7 * S), in[7] * (pix_real)(T3 * S))));
    pix_real O3 = fma(in[1], (pix_real)(T7 * S), fma(in[3], (pix_real)(-T5 * S), fma(in[5], (pix_real)(T3 * S), in[7] * (pix_real)(-T1 * S))));
    out[0] = E0 + O0; out[7] = E0 - O0;
    out[1] = E1 + O1; out[6] = E1 - O1;
    out[2] = E2 + O2; out[5] = E2 - O2;
    out[3] = E3 + O3; out[4] = E3 - O3;
}

/* The same for in[4..7] == 0: the terms left out are exact zeros, so the result is bit-identical
 * to idct8 on such input (23 instead of 35 operations). */
template <int QUARTER>
__device__ __forceinline__ void idct8_low(const pix_real in[4], pix_real out[8]) {
    constexpr pix_real S = QUARTER ? 0.25 : 1.0;
    const pix_real a = in[0] * (pix_real)(T0 * S);
    const pix_real b0 = in[2] * (pix_real)(T2 * S), b1 = in[2] * (pix_real)(T6 * S);
    const pix_real E0 = a + b0, E3 = a - b0, E1 = a + b1, E2 = a - b1;
    const pix_real O0 = fma(in[1], (pix_real)(T1 * S), in[3] * (pix_real)(T3 * S));
    const pix_real O1 = fma(in[1], (pix_real)(T3 * S), in[3] * (pix_real)(-T7 * S));
    const pix_real O2 = fma(in[1], (pix_real)(T5 * S), in[3] * (pix_real)(-T1 * S));
    const pix_real O3 = fma(in[1], (pix_real)(T7 * S), in[3] * (pix_real)(-T5 * S));
    out[0] = E0 + O0; out[7] = E0 - O0;
    out[1] = E1 + O1; out[6] = E1 - O1;
    out[2] = E2 + O2; out[5] = E2 - O2;
    out[3] = E3 + O3; out[4] = E3 - O3;
}

/* src/decoding.c:99: clamp(lround(v) + 128, 0, 255).  Adding 1.5 * 2^52 + 128 to v leaves
 * RN(v) + 128 in the low mantissa word (|v| << 2^31): round-to-nearest, which equals lround
 * except on exact ties -- and a sum of products with the six-decimal table is never an exact
 * half-integer.  One f64 add, no conversion instruction; the clamp is done on pairs (below). */
__device__ __forceinline__ int to_int128(pix_real v) {
    /* v is FOUR times the sample (the reference's "sum / 4", src/decoding.c:83, is left to this
     * add: with 1.5 * 2^54 the unit in the last place is 4, so the mantissa's low word is
     * RN(v / 4) + 128 -- the same bits as scaling the second pass's constants by 1/4, a power of
     * two, and one set of seven constants in scalar registers instead of two) */
    return __double2loint(v + (27021597764222976.0 + 512.0));
}
__device__ __forceinline__ uint32_t cvt_pk_i16(int lo, int hi) {          /* v_cvt_pk_i16_i32: both saturated to 16 bits */
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(lo, hi));
}
__device__ __forceinline__ uint32_t sat_pk_u8(uint32_t two_i16) {         /* v_sat_pk_u8_i16: {clamp(hi), clamp(lo)} to 0..255 -> bytes 1, 0 */
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(two_i16));
    return r;
}

/*
 * Exact integer form of the reference colour transform (src/image.c:44-48: doubles, clamp,
 * truncating store).  For Y in 0..255 and cb, cr in -128..127:
 *   R = clamp(Y + floor(1.402 cr)), B = clamp(Y + floor(1.772 cb)),
 *   G = clamp(Y + floor(-(34414 cb + 71414 cr) / 100000))
 * The real-valued sums are at least 1e-5 away from an integer unless they ARE integers, so the
 * reference's double evaluation and the exact rational floor agree -- except for the one
 * (cb, cr) pair noted below.  R, B: multiply-shift floors (below); G: the floor is taken in f32
 * on a value shifted positive, with a proven margin (the quotient (n/32 + 0.5) / 3125 keeps
 * fractions inside [1.5e-4, 1 - 1.5e-4], f32 error < 5e-5).  The whole
 * function is checked over all 2^24 (Y, Cb, Cr) against the oracle (tests/test_gpu_parity.py).
 */
struct ChromaTerms { int dr, dg, db; uint32_t exc; };
__device__ __forceinline__ ChromaTerms chroma_terms(int cbv, int crv) {
    int cb = cbv - 128, cr = crv - 128;
    ChromaTerms t;
    /* floor(1.402 c) = (c * 91882) >> 16 and floor(1.772 c) = (c * 116130) >> 16 for every c in
     * -128..127: the multipliers exceed 1.402 * 2^16 and 1.772 * 2^16 by < 0.6, i.e. the product
     * is off by < 0.0012, while the fractions of 1.402 c and 1.772 c are multiples of 0.002 and
     * 0.004 (zero only at c = 0) */
    t.dr = __mul24(cr, 91882) >> 16;
    t.db = __mul24(cb, 116130) >> 16;
    /* floor(n / 100000) = floor(floor(n / 32) / 3125), n = 25600000 - (34414 cb + 71414 cr) > 0 */
    uint32_t m = (uint32_t)(25600000 - __mul24(34414, cb) - __mul24(71414, cr)) >> 5;
    const float inv = 1.0f / 3125.0f;
    t.dg = (int)fmaf((float)m, inv, 0.5f * inv) - 256;
    /* the one place where double rounding in src/image.c:46 shows: Cb = 28, Cr = 228 gives
     * exactly Y - 37 in real arithmetic, but (Y + 34.414) - 71.414 in doubles lands just below
     * the integer for Y in 94..165, and the truncating store then yields one less */
    t.exc = (cbv == 28 && crv == 228) ? 1u : 0u;
    return t;
}
__device__ __forceinline__ int green_fix(const ChromaTerms &t, int Y) {
    return (int)(t.exc & (uint32_t)((uint32_t)(Y - 94) <= 71u));
}

__device__ __forceinline__ uint32_t clamp255(int v) { return (uint32_t)min(max(v, 0), 255); }

/*
 * The same terms from two 256-entry LDS tables (k_pixel builds them once per work-group):
 * entry c of the Cb table (is_cr = false) / the Cr table = { G word, d | flag << 16 }
 *   d       floor(1.772 (c - 128)) resp. floor(1.402 (c - 128)), the multiply-shift floors above
 *   G word  n = -34414 (c - 128) resp. -71414 (c - 128), split n = q * 100000 + rem (0 <= rem < 100000),
 *           as q * 2^20 + rem; the Cr table adds 2^20 - 100000 to its rem.  The sum of the two G words
 *           then carries into bit 20 exactly when rem_b + rem_r >= 100000 (and never further), so
 *           floor((n_b + n_r) / 100000) = (G_b + G_r) >> 20, arithmetic: two instructions for dg
 *   flag    c == 28 (Cb) resp. c == 228 (Cr): both set = the exceptional pair of chroma_terms
 */
__device__ __forceinline__ uint2 ctab_entry(uint32_t c, bool is_cr) {
    const int v = (int)c - 128;
    const int d = is_cr ? (__mul24(v, 91882) >> 16) : (__mul24(v, 116130) >> 16);
    const int n = is_cr ? -71414 * v : -34414 * v;
    const int q = (n + 100000 * 128) / 100000 - 128;            /* floor(n / 100000) */
    const uint32_t rem = (uint32_t)(n - q * 100000);
    const uint32_t g = ((uint32_t)q << 20) + rem + (is_cr ? (1u << 20) - 100000u : 0u);
    const uint32_t flag = (is_cr ? c == 228u : c == 28u) ? 0x10000u : 0u;
    return make_uint2(g, ((uint32_t)d & 0xFFFFu) | flag);
}
#define JD_CTAB_CR 256u   /* the Cr table follows the Cb table */
__device__ __forceinline__ ChromaTerms chroma_lookup(const uint2 *ctab, uint32_t cbv, uint32_t crv) {
    const uint2 b = ctab[cbv], r = ctab[JD_CTAB_CR + crv];
    ChromaTerms t;
    t.dr = (int)r.y; t.db = (int)b.y;            /* low halves; the users take 16 bits */
    t.dg = (int)(b.x + r.x) >> 20;
    t.exc = (b.y & r.y) >> 16;
    return t;
}

/*
 * 4 pixels -> their 12 output bytes, two pixels per instruction.  Px4: per channel the bytes of
 * pixels (0, 2) and of pixels (1, 3), each pair in the low two bytes of a register -- what
 * v_sat_pk_u8_i16 leaves when it clamps two 16-bit sums to 0..255 and packs them, which is the
 * reference's clamp (src/image.c:34-36, 44-48) for two values at once.
 */
struct Px4 { uint32_t c02[3], c13[3]; };
typedef short jd_short2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add16(uint32_t a, uint32_t b) {   /* v_pk_add_u16 */
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(jd_short2, a) + __builtin_bit_cast(jd_short2, b));
}
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(jd_short2, a) - __builtin_bit_cast(jd_short2, b));
}
__device__ __forceinline__ uint32_t pack16(int lo, int hi) { return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u); }

/* y4: the luma bytes of the 4 pixels; cbv / crv: their chroma */
__device__ __forceinline__ Px4 ycc4(uint32_t y4, const uint32_t cbv[4], const uint32_t crv[4]) {
    Px4 r;
    r.c02[0] = __builtin_amdgcn_perm(0u, y4, 0x0C0C0200u); r.c13[0] = __builtin_amdgcn_perm(0u, y4, 0x0C0C0301u);
    r.c02[1] = cbv[0] | (cbv[2] << 8); r.c13[1] = cbv[1] | (cbv[3] << 8);
    r.c02[2] = crv[0] | (crv[2] << 8); r.c13[2] = crv[1] | (crv[3] << 8);
    return r;
}
/* ct: the chroma terms of the 4 pixels (with chroma shared by two columns, ct[1] = ct[0], ct[3] = ct[2]) */
__device__ __forceinline__ Px4 rgb4(uint32_t y4, const ChromaTerms ct[4]) {
    Px4 r;
    const uint32_t y02 = y4 & 0x00FF00FFu, y13 = (y4 >> 8) & 0x00FF00FFu;   /* two 16-bit lanes each */
    uint32_t s02[3] = {pk_add16(y02, pack16(ct[0].dr, ct[2].dr)), pk_add16(y02, pack16(ct[0].dg, ct[2].dg)),
                       pk_add16(y02, pack16(ct[0].db, ct[2].db))};
    uint32_t s13[3] = {pk_add16(y13, pack16(ct[1].dr, ct[3].dr)), pk_add16(y13, pack16(ct[1].dg, ct[3].dg)),
                       pk_add16(y13, pack16(ct[1].db, ct[3].db))};
    /* wave-uniform and almost never taken: a lane of the wave holds the exceptional chroma pair.
     * Kept a real branch (the empty asm stops the compiler from turning it into selects, which
     * cost every wave a dozen instructions per row). */
    if (__any((int)(ct[0].exc | ct[1].exc | ct[2].exc | ct[3].exc))) {
        asm volatile("" ::: "memory");
        int fix[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fix[i] = green_fix(ct[i], (int)((y4 >> (8 * i)) & 255u));
        s02[1] = pk_sub16(s02[1], pack16(fix[0], fix[2]));
        s13[1] = pk_sub16(s13[1], pack16(fix[1], fix[3]));
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { r.c02[c] = sat_pk_u8(s02[c]); r.c13[c] = sat_pk_u8(s13[c]); }
    return r;
}

/* output layouts (JDGPU_OUT_* in include/jpeg_gpu.h).  The lane's 4 pixels start at column x
 * (a multiple of 4) of row y; npx of them lie inside the image. */
#define JD_FMT_3 0u       /* pixel[] of src/image.h:4-12: 3 interleaved bytes */
#define JD_FMT_4 1u       /* 4 interleaved bytes, the fourth 255 */
#define JD_FMT_PLANAR 2u  /* three planes of width*height bytes */
template <uint32_t FMT>
__device__ __forceinline__ void store_px4(uint8_t *rows, size_t plane, uint32_t at, const Px4 &v, bool aligned,
                                          uint32_t npx) {
    /* rows: the image's output at the first pixel row of this MCU row (wave-uniform: a scalar base);
     * at = r * stride + x within it (< 2^22: r < 32, stride <= 2^16), so every address is the scalar
     * base plus a 32-bit lane offset; plane = width * height.  v_perm_b32 picks the output bytes:
     * selector 0..3 = bytes of the second operand, 4..7 = of the first, 0x0D = 0xFF */
    if (FMT == JD_FMT_PLANAR) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t w = __builtin_amdgcn_perm(v.c13[c], v.c02[c], 0x05010400u);   /* pixels 0, 1, 2, 3 */
            uint8_t *dst = rows + c * plane + at;
            if (aligned) *reinterpret_cast<uint32_t *>(dst) = w;       /* stride % 4 == 0, four pixels */
            else
                for (uint32_t i = 0; i < npx; ++i) dst[i] = (uint8_t)(w >> (8 * i));
        }
        return;
    }
    const uint32_t rg02 = v.c02[0] | (v.c02[1] << 16);    /* R0 R2 G0 G2 */
    const uint32_t br = v.c02[2] | (v.c13[0] << 16);      /* B0 B2 R1 R3 */
    const uint32_t gb13 = v.c13[1] | (v.c13[2] << 16);    /* G1 G3 B1 B3 */
    if (FMT == JD_FMT_4) {
        /* 16 bytes per lane, always aligned (x is a multiple of 4, images start on 256 bytes) */
        const uint32_t p[4] = {__builtin_amdgcn_perm(br, rg02, 0x0D040200u), __builtin_amdgcn_perm(gb13, br, 0x0D060402u),
                               __builtin_amdgcn_perm(br, rg02, 0x0D050301u), __builtin_amdgcn_perm(gb13, br, 0x0D070503u)};
        uint32_t *dst = reinterpret_cast<uint32_t *>(rows + at * 4u);
        if (npx == 4) *reinterpret_cast<uint4 *>(dst) = make_uint4(p[0], p[1], p[2], p[3]);
        else
            for (uint32_t i = 0; i < npx; ++i) dst[i] = p[i];
    } else {
        const uint32_t w[3] = {__builtin_amdgcn_perm(br, rg02, 0x06040200u),      /* R0 G0 B0 R1 */
                               __builtin_amdgcn_perm(rg02, gb13, 0x07050200u),    /* G1 B1 R2 G2 */
                               __builtin_amdgcn_perm(gb13, br, 0x07050301u)};     /* B2 R3 G3 B3 */
        uint8_t *dst = rows + at * 3u;
        if (aligned) {
            uint32_t *d32 = reinterpret_cast<uint32_t *>(dst);
            d32[0] = w[0]; d32[1] = w[1]; d32[2] = w[2];   /* streaming (nontemporal) stores: measured, no change */
        } else {
            for (uint32_t i = 0; i < npx * 3; ++i) dst[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
        }
    }
}

#define SUB_PX 64u                 /* pixels per wave: each wave owns a 64-pixel-wide part of the strip */
#define PIX_ROW (SUB_PX + 8u)      /* bytes per sample-tile row (+8 spreads the rows over banks) */
#define PLANE_BYTES (32u * PIX_ROW) /* generic instantiation: up to 4 units = 32 sample rows per MCU row */
#define TR_BLK 64u                 /* values per 8x8 block in the transpose buffer */

#define WAVE_SYNC()                                                   \
    do {                                                              \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        \
        __builtin_amdgcn_wave_barrier();                              \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");        \
    } while (0)

/*
 * k_pixel: one work-group per strip of an MCU row (JD_STRIP_PX pixels wide); each of its
 * waves owns a 64-pixel-wide part and runs both phases on it without any work-group barrier:
 *   phase 1: 8 lanes per data unit, 8 units at a time, luma units first, then Cb, then Cr.
 *            Lane u loads column u of the unit (coefficients are stored column-major: 16
 *            contiguous bytes of int16, or 8 of int8 in a narrow image), dequantises, does the
 *            vertical 8-point transform, transposes through LDS (within the wave), does the
 *            horizontal transform for row y, rounds and writes 8 samples of one row into the
 *            component's sample tile.  A group of units without a coefficient of frequency 4 or
 *            more in either direction takes both passes with half the terms.
 *   phase 2: 4 pixels per lane (x 2 rows when chroma is shared vertically): gather Y/Cb/Cr
 *            (replicated chroma), colour transform, one 12-byte store; 16 lanes cover 192
 *            contiguous bytes of an output row.
 * OUT_YCBCR leaves YCbCr in the buffer (what jpeg_read_image produces); FMT: output layout.
 */
template <bool OUT_YCBCR, uint32_t MODE, uint32_t FMT>
__global__ __launch_bounds__(64 * JD_PIX_WAVES) void k_pixel(const jd_image *__restrict__ imgs,
                                               const jd_pix_wg *__restrict__ wgs,
                                               const int16_t *__restrict__ coef,
                                               uint8_t *__restrict__ pix,
                                               const uint32_t *__restrict__ status,
                                               const uint32_t *__restrict__ wide,   /* per image: 0 = 64-byte units; null: none is */
                                               const uint32_t *__restrict__ redo,   /* a later pass: the images to do again; null = all */
                                               uint32_t pix_rows) {
    __shared__ __attribute__((aligned(16))) pix_real s_tr[8 * JD_PIX_WAVES * TR_BLK];
    /* sample tiles: per wave one plane per component, 8 rows per vertical sampling factor */
    /* the layouts with an instantiation of their own: luma HSH+1 x VSH+1 units per MCU, chroma 1x1 */
    constexpr bool FAST = MODE != JD_PIX_GENERIC, GRAY = MODE == JD_PIX_GRAY;
    constexpr uint32_t HSH = (MODE == JD_PIX_420 || MODE == JD_PIX_422) ? 1u : 0u;   /* chroma shared by 2 columns */
    constexpr uint32_t VSH = (MODE == JD_PIX_420 || MODE == JD_PIX_440) ? 1u : 0u;   /* ... by 2 rows */
    constexpr uint32_t PLANE1 = FAST ? (8u << VSH) * PIX_ROW : PLANE_BYTES;
    constexpr uint32_t PLANE2 = PLANE1 + (FAST ? (GRAY ? 0u : 8u * PIX_ROW) : PLANE_BYTES);
    constexpr uint32_t TILE_BYTES = PLANE2 + (FAST ? (GRAY ? 0u : 8u * PIX_ROW) : PLANE_BYTES);
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[JD_PIX_WAVES][TILE_BYTES];
    __shared__ __attribute__((aligned(16))) uint16_t s_q[4][64];
    constexpr uint32_t MAXB = FAST ? 6u : JD_MAX_BPM;      /* data units per MCU: the layouts with an instantiation have at most 2x2 + 1 + 1 */
    constexpr uint32_t NJMAP = FAST ? 8u * MAXB : JD_JMAP;
    __shared__ uint32_t s_meta[MAXB]; /* per data unit of the MCU: comp | tq << 2 | h << 4 | v << 6 | H << 8 (4 bits) | plane offset / 64 << 12 */
    __shared__ uint16_t s_jmap[NJMAP]; /* j-th data unit of a wave's part of the strip: MCU | unit << 8 */
    __shared__ uint2 s_ctab[OUT_YCBCR ? 1 : 512]; /* colour transform terms per Cb / Cr value (ctab_entry) */

    const jd_pix_wg wg = wgs[wg_index()];
    /* an image whose entropy decode failed keeps the zeros its output was cleared to at creation:
     * the data units it never reached hold whatever the (recycled) coefficient buffer held.  An image
     * the write kernel left UNSYNCED (broken segment chain, overflowing narrow unit) is done again by a
     * second pass (jdgpu_batch_sync): drawing it now would be wasted, and would stay in the buffer
     * under a failed status if that pass then meets the real end of data */
    if (status[wg.img] & (uint32_t)(JDGPU_IMG_EOF | JDGPU_IMG_NOSYMBOL | JDGPU_IMG_UNSYNCED | JDGPU_IMG_RANGE)) return;
    if (redo && redo[wg.img] == 0u) return;   /* finished by an earlier pass */
    const jd_image &im = imgs[wg.img];
    const bool nar = wide != nullptr && wide[wg.img] == 0u;   /* the image's units are in the narrow form (k_huff_write) */
    const uint32_t ub = nar ? JD_NARROW_UNIT : 128u;          /* bytes per unit */
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);   /* scalar: so is everything derived from it */
    /* one instantiation per pixel layout: the common ones get their geometry as constants */
    const uint32_t maxH = FAST ? 1u << HSH : im.maxH;
    const uint32_t maxV = FAST ? 1u << VSH : im.maxV;
    const uint32_t bpm = FAST ? (GRAY ? 1u : (1u << (HSH + VSH)) + 2u) : im.bpm;
    const uint32_t sub_mcus = SUB_PX / (8u * maxH);
    for (uint32_t i = tid; i < 256; i += 64 * JD_PIX_WAVES) s_q[i >> 6][i & 63u] = im.quant[i >> 6][i & 63u];
    if (tid < MAXB) {
        uint32_t c = im.blk_comp[tid];
        const uint32_t plane = c == 0 ? 0u : c == 1 ? PLANE1 : PLANE2;   /* multiples of 64 bytes */
        s_meta[tid] = c | ((uint32_t)im.comp_tq[c] << 2) | ((uint32_t)im.blk_h[tid] << 4) | ((uint32_t)im.blk_v[tid] << 6) |
                      ((uint32_t)im.comp_H[c] << 8) | ((plane >> 6) << 12);
    }
    if (tid < NJMAP) {
        /* order in which a wave takes the data units of its part of an MCU row, 8 at a time.  Generic:
         * as stored.  The layouts with an instantiation: all luma units first, then Cb, then Cr, so that
         * a group of 8 is of one kind -- chroma units seldom hold a coefficient beyond the low 4 x 4,
         * and a group in which none does takes the short transform (idct8_low) */
        uint32_t m = tid / bpm, b = tid % bpm;
        if (FAST && !GRAY) {
            const uint32_t nY = bpm - 2u, j2 = tid - sub_mcus * nY;
            if (tid < sub_mcus * nY) { m = tid / nY; b = tid % nY; }
            else { m = j2 % sub_mcus; b = nY + j2 / sub_mcus; }
        }
        s_jmap[tid] = (uint16_t)(m | (b << 8));
    }
    if (!OUT_YCBCR)
        for (uint32_t i = tid; i < 512; i += 64 * JD_PIX_WAVES) s_ctab[i] = ctab_entry(i & 255u, i >= JD_CTAB_CR);
    const uint32_t mcu_first = wg.mcu_col0 + wave * sub_mcus;          /* first MCU column of this wave */
    const uint32_t my_mcus = mcu_first < im.mcus_x ? min(sub_mcus, im.mcus_x - mcu_first) : 0u;
    const uint32_t nblk = my_mcus * bpm;
    const uint32_t ntot = FAST ? sub_mcus * bpm : nblk;   /* entries of s_jmap to walk; a narrow last part skips those of absent MCUs */

    /* ---- phase 1 ---- */
    const uint32_t l = lane & 7u, slot = lane >> 3;
    /* transpose buffer: element (y, u) of slot s at s*64 + (y ^ (s & 3))*8 + ((u + y) & 7).
     * ds_write_b64 is served 16 lanes at a time over 32 dword banks, ds_read_b64 32 lanes over 64:
     * the 16 lanes (2 slots) that write step y hit 16 different bank pairs and the 32 lanes
     * (4 slots) that read step u hit 32 (searched exhaustively; the earlier swizzle
     * (u + y) ^ (s & 3) in the column made every write 2-way: 36 % of the kernel's LDS cycles) */
    pix_real *tr = s_tr + (wave * 8u + slot) * TR_BLK;
    uint8_t *tile = s_tile[wave];
    const uint32_t sx = slot & 3u;
    __syncthreads();
    if (nblk == 0) return;
    const uint32_t row_end = min((uint32_t)wg.mcu_row + pix_rows, im.mcus_y);
    /* the wave's data units of one MCU row are consecutive in the coefficient buffer; lane: its
     * 16-byte column of unit `slot` */
    const size_t row_stride = (size_t)im.mcus_x * bpm * ub;   /* bytes */
    /* coefficients of the wave's part of the MCU row (wave-uniform: a scalar base, 32-bit lane offsets);
     * an image's units start at blk_off * 128 bytes in either form */
    const char *cbase = reinterpret_cast<const char *>(coef + (size_t)im.blk_off * 64) +
                        ((size_t)wg.mcu_row * im.mcus_x + mcu_first) * bpm * ub;
    const uint32_t W = im.width, H = im.height;
    const uint32_t S = im.stride;   /* pixels from one output row to the next (W unless the rows are pitched) */
    const uint32_t x0 = mcu_first * 8u * maxH;
    const uint32_t g = lane & 15u, rq = lane >> 4, x = x0 + 4u * g;
    uint8_t *out = pix + im.pix_off;
    const bool draws = x < W && 4u * g < my_mcus * 8u * maxH;   /* lane has pixels in phase 2 */
    const uint32_t npx = draws ? min(4u, W - x) : 0u;
    /* whole-dword stores: rows that start on a dword and a lane with all four pixels (pitched rows of a
     * ragged width: only the last lane of a row goes byte by byte, the padding stays zero) */
    const bool aligned = ((S & 3u) == 0) && npx == 4u;
    const uint8_t *tY = tile, *tB = tile + PLANE1, *tR = tile + PLANE2;
    const size_t plane = (size_t)S * H;

    /* A wave walks down pix_rows MCU rows.  All coefficient loads of the first 24 units of a
     * row (all of them for 4:2:0 and 4:4:4) are issued one row AHEAD: a wave that loaded, waited
     * out the HBM latency, computed and left spent a third of its life in that wait. */
    /* where the lane's row of unit j goes in the sample tile, and its row of the quantisation table */
    auto unit_geom = [&](uint32_t j, uint32_t &toff, uint32_t &qoff, uint32_t &uoff) -> bool {
        const uint32_t jm = s_jmap[j], m = jm & 255u, b = jm >> 8, meta = s_meta[b];
        const uint32_t row = ((meta >> 6) & 3u) * 8u + l;
        const uint32_t colpx = (m * ((meta >> 8) & 15u) + ((meta >> 4) & 3u)) * 8u;
        toff = (meta >> 12) * 64u + row * PIX_ROW + colpx;
        qoff = ((meta >> 2) & 3u) * 64u + l * 8u;
        uoff = (m * bpm + b) * ub + l * (ub >> 3);   /* byte offset of the lane's column in the row's coefficients */
        return m < my_mcus;
    };
    /* the lane's column of a unit: 8 int16 (raw.x .. raw.w) or, narrow, 8 int8 (raw.x, raw.y) */
    auto load_col = [&](const char *p) -> uint4 {
        if (nar) { const uint2 t = *reinterpret_cast<const uint2 *>(p); return make_uint4(t.x, t.y, 0u, 0u); }
        return *reinterpret_cast<const uint4 *>(p);
    };
    /* quantised coefficients v0 .. v0 + N - 1 of the column times their table entries (src/decoding.c:46,65).
     * Narrow: byte 0 of column 0 is the low byte of the 12-bit DC value, whose high nibble column 7
     * holds in the low half of its last byte (the high half: the last AC coefficient); dc12 / a63 are
     * those two values, put right before the call */
    auto dequant = [&](const uint4 &raw, uint32_t qoff, pix_real *col, int n /* 4 or 8 */) {
        const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w};
        uint32_t qw[4];
        if (n == 8) { const uint4 qv = *reinterpret_cast<const uint4 *>(&s_q[0][0] + qoff); qw[0] = qv.x; qw[1] = qv.y; qw[2] = qv.z; qw[3] = qv.w; }
        else { const uint2 qv = *reinterpret_cast<const uint2 *>(&s_q[0][0] + qoff); qw[0] = qv.x; qw[1] = qv.y; qw[2] = qw[3] = 0; }
        if (nar) {   /* wave-uniform: a real branch, not a select per coefficient */
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < n) col[k] = (pix_real)__mul24((int)(int8_t)(rw[k >> 2] >> (8 * (k & 3))), (int)((qw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu));
            /* column 0 cannot hold 12 bits in a byte: its first product is made from raw.w */
            col[0] = l == 0u ? (pix_real)__mul24((int)raw.w, (int)(qw[0] & 0xFFFFu)) : col[0];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < n) col[k] = (pix_real)__mul24((int)(int16_t)(rw[k >> 1] >> (16 * (k & 1))), (int)((qw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu));
        }
    };
    /* src/decoding.c:99 for two samples at a time: RN(v) + 128 as integers (to_int128), saturated to
     * 16 bits and then to 0..255 by the two packing instructions; the 8 samples of a row go to the tile */
    auto put_row = [&](const pix_real t[8], uint32_t toff) {
        const uint32_t p01 = sat_pk_u8(cvt_pk_i16(to_int128(t[0]), to_int128(t[1])));
        const uint32_t p23 = sat_pk_u8(cvt_pk_i16(to_int128(t[2]), to_int128(t[3])));
        const uint32_t p45 = sat_pk_u8(cvt_pk_i16(to_int128(t[4]), to_int128(t[5])));
        const uint32_t p67 = sat_pk_u8(cvt_pk_i16(to_int128(t[6]), to_int128(t[7])));
        *reinterpret_cast<uint2 *>(&tile[toff]) = make_uint2(pack16((int)p01, (int)p23), pack16((int)p45, (int)p67));
    };
    /* the first 24 units of a row (all of them in most layouts) sit in the same places in every MCU row */
    uint32_t g_toff[3] = {0, 0, 0}, g_qoff[3] = {0, 0, 0}, g_uoff[3] = {0, 0, 0};
    bool g_act[3] = {false, false, false}, g_any[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (8u * i + slot < ntot) g_act[i] = unit_geom(8u * i + slot, g_toff[i], g_qoff[i], g_uoff[i]);
        g_any[i] = __any(g_act[i]);   /* wave-uniform: false for a group of units of MCUs beyond the image's edge */
    }
    const uint32_t hi4 = l >= 4u ? ~0u : 0u;
    uint4 ahead[3]; /* single buffer: chunk i of the next row is loaded as soon as step i has used it */
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ahead[i] = make_uint4(0, 0, 0, 0);
        if (g_act[i]) ahead[i] = load_col(cbase + g_uoff[i]);
    }
    for (uint32_t mrow = wg.mcu_row; mrow < row_end; ++mrow, cbase += row_stride) {
        for (uint32_t r0 = 0; r0 < ntot; r0 += 24) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const uint32_t j = r0 + 8u * i + slot;
                if (r0 + 8u * i >= ntot) break; /* wave-uniform */
                bool act = g_act[i], any = g_any[i];
                uint32_t toff = g_toff[i], qoff = g_qoff[i], uoff = g_uoff[i];
                uint4 raw = ahead[i];   /* zero in a lane without a unit */
                if (r0) { /* layouts with more than 24 units per wave and row: loaded in place */
                    raw = make_uint4(0, 0, 0, 0);
                    act = j < ntot && unit_geom(j, toff, qoff, uoff);
                    any = __any(act);
                    if (act) raw = load_col(cbase + uoff);
                } else if (act && mrow + 1 < row_end) {
                    ahead[i] = load_col(cbase + row_stride + uoff);
                }
                if (!any) continue; /* wave-uniform */
                /* lane: column l of its unit = the coefficients of horizontal frequency l, vertical
                 * frequency 0..7 in raw.x .. raw.w.  No lane of the wave has one at a frequency of 4 or
                 * more in either direction (wave-uniform): both passes take half the terms */
                if (nar) {
                    /* put the two shared bytes right: column 7 gets its last coefficient as an int8 again,
                     * column 0 its DC value in raw.z (bits 0..11 = byte 0 | nibble << 8, sign-extended below) */
                    const uint32_t y7 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw.y, 0x107, 0xF, 0xF, true); /* row_shl:7: lane 8k <- lane 8k + 7 */
                    const uint32_t fixed = (raw.y & 0x00FFFFFFu) | ((uint32_t)((int)raw.y >> 28) << 24);
                    raw.y = l == 7u ? fixed : raw.y;
                    const int dc12 = __builtin_amdgcn_sbfe((int)((raw.x & 0xFFu) | ((y7 >> 16) & 0xF00u)), 0, 12);
                    /* column 0 cannot hold 12 bits in a byte: its first product is made here, from raw.w */
                    raw.w = (uint32_t)dc12;
                }
                const bool low = nar ? !__any((raw.y | ((raw.x | raw.y) & hi4)) != 0u)
                                     : !__any((raw.z | raw.w | ((raw.x | raw.y) & hi4)) != 0u);
                pix_real col[8], t[8];
                if (low) {
                    if (act) {
                        dequant(raw, qoff, col, 4);
                        idct8_low<0>(col, t);
#pragma unroll
                        for (int y = 0; y < 8; ++y) tr[((uint32_t)y ^ sx) * 8u + ((l + y) & 7u)] = t[y];
                    }
                    WAVE_SYNC();
                    if (act) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) col[u] = tr[(l ^ sx) * 8u + ((u + l) & 7u)];
                        idct8_low<0>(col, t);
                        put_row(t, toff);
                    }
                } else {
                    if (act) {
                        dequant(raw, qoff, col, 8);
                        idct8<0>(col, t); /* vertical: t[y] = sum_v M[v][u] T[v][y] */
#pragma unroll
                        for (int y = 0; y < 8; ++y) tr[((uint32_t)y ^ sx) * 8u + ((l + y) & 7u)] = t[y];
                    }
                    WAVE_SYNC(); /* the 8 lanes of a unit sit in one wave */
                    if (act) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) col[u] = tr[(l ^ sx) * 8u + ((u + l) & 7u)];
                        idct8<0>(col, t); /* horizontal: row y = l, t[x] = sample - 128 before rounding */
                        put_row(t, toff);
                    }
                }
                WAVE_SYNC();
            }
        }

        /* ---- phase 2 ---- */
        const uint32_t y0 = mrow * 8u * maxV;
        const uint32_t rows = min(8u * maxV, H - y0);
        /* output of pixel row y0 (wave-uniform): plane 0 of the planar layout, else the interleaved pixels */
        uint8_t *const orow = out + (size_t)y0 * S * (FMT == JD_FMT_PLANAR ? 1u : FMT == JD_FMT_4 ? 4u : 3u);
        if (draws) {
            /* 24-bit multiplies for the row offsets and the pixel index (rows, widths < 2^16): a
             * full 32-bit or 64-bit multiply costs as much as four plain instructions */
            if (FAST) {
                /* lane: 4 pixels of one chroma row = 1 or 2 (VSH) luma rows; two chroma rows per lane
                 * group rq.  With vertical sharing: {0,2}, {4,6}, {1,3}, {5,7} -- the two groups
                 * that share a 32-lane LDS access then read luma rows 8 apart, 16 banks: no conflict
                 * (2 apart would overlap on 12 of 16 banks).  offC is carried: the compiler makes
                 * rp * 72 a full multiply */
                const uint32_t rp0 = VSH ? (rq & 1u) * 4u + (rq >> 1) : rq;
                constexpr uint32_t STEP = VSH ? 2u : 4u;      /* without vertical sharing: rows rq, rq + 4 (1 apart in a group) */
                constexpr uint32_t CB = 4u >> HSH;            /* chroma bytes under the lane's 4 pixels */
                uint32_t offC = rp0 * PIX_ROW + CB * g;
                uint32_t at = __umul24(rp0 << VSH, S) + x;          /* pixel index in the MCU row's output; carried, like offC */
                const uint32_t at_step = __umul24(STEP << VSH, S);
                for (uint32_t rp = rp0; rp < rp0 + 2u * STEP && (rp << VSH) < rows; rp += STEP, offC += STEP * PIX_ROW, at += at_step) {
                    uint32_t cbv[4], crv[4];
                    if (GRAY) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) cbv[i] = crv[i] = 128u; /* absent channels: neutral */
                    } else if (HSH) {
                        uint32_t cbp = *reinterpret_cast<const uint16_t *>(&tB[offC]);
                        uint32_t crp = *reinterpret_cast<const uint16_t *>(&tR[offC]);
                        cbv[0] = cbv[1] = cbp & 255u; cbv[2] = cbv[3] = cbp >> 8;
                        crv[0] = crv[1] = crp & 255u; crv[2] = crv[3] = crp >> 8;
                    } else {
                        uint32_t cb4 = *reinterpret_cast<const uint32_t *>(&tB[offC]);
                        uint32_t cr4 = *reinterpret_cast<const uint32_t *>(&tR[offC]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) { cbv[i] = (cb4 >> (8 * i)) & 255u; crv[i] = (cr4 >> (8 * i)) & 255u; }
                    }
                    ChromaTerms ct[4];
                    if (!OUT_YCBCR && !GRAY) {
                        ct[0] = chroma_lookup(s_ctab, cbv[0], crv[0]);
                        ct[2] = chroma_lookup(s_ctab, cbv[2], crv[2]);
                        if (HSH) { ct[1] = ct[0]; ct[3] = ct[2]; }
                        else { ct[1] = chroma_lookup(s_ctab, cbv[1], crv[1]); ct[3] = chroma_lookup(s_ctab, cbv[3], crv[3]); }
                    } else if (!OUT_YCBCR) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) ct[i] = ChromaTerms{0, 0, 0, 0u};   /* neutral chroma: R = G = B = Y */
                    }
                    const uint32_t offY = ((offC - CB * g) << VSH) + 4u * g;   /* luma row (rp << VSH), column 4g */
#pragma unroll
                    for (uint32_t dy = 0; dy < (1u << VSH); ++dy) {
                        uint32_t r = (rp << VSH) + dy;
                        if (r < rows) {
                            uint32_t y4 = *reinterpret_cast<const uint32_t *>(&tY[offY + dy * PIX_ROW]);
                            Px4 v = OUT_YCBCR ? ycc4(y4, cbv, crv) : rgb4(y4, ct);
                            store_px4<FMT>(orow, plane, at + (dy ? S : 0u), v, aligned, npx);
                        }
                    }
                }
            } else {
                /* any mix of factors 1..4 whose replication ratios maxH/H, maxV/V are whole numbers
                 * (src/jpeg.c:82), and any channel order: per-sample gathers.  Pixel p of the MCU row takes
                 * sample p / ratio: a multiply-shift, floor(p * rcp / 512) with rcp = 512, 256, 171, 128 for
                 * ratio 1..4 (exact for p < 512; p < 64 here) */
                uint32_t hs[3], vs[3], ch[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    hs[c] = (512u + maxH / im.comp_H[c] - 1u) / (maxH / im.comp_H[c]);
                    vs[c] = (512u + maxV / im.comp_V[c] - 1u) / (maxV / im.comp_V[c]);
                    ch[c] = im.comp_chan[c];
                }
                for (uint32_t r = rq; r < rows; r += 4) {
                    uint32_t sm[3][4]; /* sm[channel][pixel]; channel = component id - 1 (src/decoding.c:107) */
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const uint8_t *tp = &tile[c * PLANE_BYTES + ((r * vs[c]) >> 9) * PIX_ROW];
                        const bool absent = (uint32_t)c >= im.ncomp; /* grayscale: neutral chroma */
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            uint32_t v = absent ? 128u : tp[((4u * g + i) * hs[c]) >> 9];
                            if (ch[c] == 0) sm[0][i] = v; else if (ch[c] == 1) sm[1][i] = v; else sm[2][i] = v;
                        }
                    }
                    uint32_t y4 = sm[0][0] | (sm[0][1] << 8) | (sm[0][2] << 16) | (sm[0][3] << 24);
                    Px4 v;
                    if (OUT_YCBCR) v = ycc4(y4, sm[1], sm[2]);
                    else {
                        ChromaTerms ct[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) ct[i] = chroma_lookup(s_ctab, sm[1][i], sm[2][i]);
                        v = rgb4(y4, ct);
                    }
                    store_px4<FMT>(orow, plane, __umul24(r, S) + x, v, aligned, npx);
                }
            }
        }
        WAVE_SYNC(); /* the tile is rewritten by the next row */
    }
}

/* image_ycbcr_to_rgb on a device buffer: src/image.c:39-51, same exact integer form.  4 pixels =
 * 12 bytes = three dwords per lane (the buffer is dword aligned); the last lane takes the ragged
 * rest byte by byte. */
__global__ __launch_bounds__(256) void k_ycbcr_to_rgb(uint8_t *px, size_t npix) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    if (i >= npix) return;
    if (i + 4u <= npix) {
        uint32_t *p = reinterpret_cast<uint32_t *>(px + 3u * i);
        const uint32_t w[3] = {p[0], p[1], p[2]};
        uint32_t o[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            auto byte = [&](int j) { return (int)((w[j >> 2] >> (8 * (j & 3))) & 255u); };
            const int Y = byte(3 * k), Cb = byte(3 * k + 1), Cr = byte(3 * k + 2);
            const ChromaTerms ct = chroma_terms(Cb, Cr);
            const uint32_t v[3] = {clamp255(Y + ct.dr), clamp255(Y + ct.dg - green_fix(ct, Y)), clamp255(Y + ct.db)};
#pragma unroll
            for (int c = 0; c < 3; ++c) o[(3 * k + c) >> 2] |= v[c] << (8 * ((3 * k + c) & 3));
        }
        p[0] = o[0]; p[1] = o[1]; p[2] = o[2];
    } else {
        for (size_t q = i; q < npix; ++q) {
            int Y = px[3 * q], Cb = px[3 * q + 1], Cr = px[3 * q + 2];
            ChromaTerms ct = chroma_terms(Cb, Cr);
            px[3 * q] = (uint8_t)clamp255(Y + ct.dr);
            px[3 * q + 1] = (uint8_t)clamp255(Y + ct.dg - green_fix(ct, Y));
            px[3 * q + 2] = (uint8_t)clamp255(Y + ct.db);
        }
    }
}

/* image_rgb_to_ycbcr on a device buffer: src/image.c:54-66 literally -- the same f64 products and
 * sums in the same order (the build disables contraction), clamp, truncating store.  The forward
 * transform has no integer shortcut worth proving: it is off the decode path. */
__device__ __forceinline__ uint8_t clamp_trunc(double v) { return (uint8_t)(v < 0 ? 0.0 : 255 < v ? 255.0 : v); }
__global__ __launch_bounds__(256) void k_rgb_to_ycbcr(uint8_t *px, size_t npix) {
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4u; /* 4 pixels per lane, as k_ycbcr_to_rgb */
    for (size_t i = i0; i < npix && i < i0 + 4u; ++i) {
        const double R = px[3 * i], G = px[3 * i + 1], B = px[3 * i + 2];
        px[3 * i] = clamp_trunc(0.299 * R + 0.587 * G + 0.114 * B);
        px[3 * i + 1] = clamp_trunc(-0.1687 * R - 0.3313 * G + 0.5 * B + 128.0);
        px[3 * i + 2] = clamp_trunc(0.5 * R - 0.4187 * G - 0.0813 * B + 128.0);
    }
}

/* ============================================================================================
 * host side of the C ABI
 * ========================================================================================== */

struct jdgpu_batch {
    jd_plan plan;
    unsigned flags;
    hipStream_t stream;
    uint8_t *d_raw, *d_clean, *d_pix;
    jd_image *d_img;
    jd_lut *d_luts;
    uint32_t *d_seg_info, *d_st_in, *d_st_exit, *d_status, *d_cp_state, *d_wide;
    uint32_t *d_list, *d_list_cnt;   /* repair lists: [segment capacity], [JD_REPAIR_ROUNDS][images] */
    int4 *d_cnt, *d_start_dc, *d_cp_val;
    int2 *d_start_blk;
    int16_t *d_coef;
    jd_huff_wg *d_hwg, *d_wwg, *d_rpl, *d_fwg;
    jd_pix_wg *d_pwg;
    jd_raw_wg *d_rwg;
    uint2 *d_chunk_cnt, *d_chunk_base;
    uint32_t *d_lane_masks;          /* per 16 raw bytes: kept-byte and marker bits (k_unstuff_count -> scatter) */
    uint32_t *d_ivl_u, *d_ivl_seg;
    bool any_rst;
    hipEvent_t ev[JDGPU_NSTAGES + 1];
    float ms[JDGPU_NSTAGES];
    std::vector<uint32_t> status;
    int retries;
    uint64_t *d_scan_src; /* JDGPU_SCAN_ON_DEVICE: where the scans lie (read by the gather kernel) */
    uint8_t *d_files; /* jdgpu_batch_create_from_files: the uploaded files, kept until the batch goes (the scan
                         gather runs on the batch's stream, in front of its first decode) */
    bool redo;       /* the current pass is a repeat: kernels skip the images whose word in the redo list is 0 */
    bool timed;
    uint8_t *d_meta; /* small batches: the plan's arrays (d_img .. d_wwg, d_scan_src) are parts of this one buffer */
    std::vector<uint8_t> meta_host;   /* ... and what their upload reads (it may still run when the creation returns) */
    uint32_t *h_status;   /* pinned, JD_HSTATUS_WORDS words, recycled with the stream: where a decode's status words
                             land without a copy call of their own (batches of at most that many images) */
    bool status_copied;   /* the decode in flight ends with that copy */
    hipEvent_t ev_created; /* the creation's GPU work (on g_hi) has ended: the batch's stream waits for it */
    bool hi_pending;       /* work of this batch may still run on g_hi (a creation that failed half way) */
    float create_ms[JDGPU_NCREATE];   /* where the creation of this batch spent its time (jdgpu_batch_create_ms) */
};

static std::mutex g_mu;
static int g_device = -1;

/* Scratch for jdgpu_ycbcr_to_rgb_noalloc: device staging memory that exists from module load to
 * process end, so that entry point never allocates.  g_scratch is its device address on the
 * bound device (hipGetSymbolAddress in jdgpu_init); host code never uses the array's name for
 * anything else.  One call at a time uses it (g_scratch_mu). */
#define JD_SCRATCH_BYTES (32u << 20)
#define JD_CLEARED_WORDS (1u + 3u + JD_REPAIR_ROUNDS)   /* per image: d_status, d_wide, d_list_cnt (one block) */
#define JD_META_PACK_MAX ((size_t)256 << 10)   /* plans up to this size go up as one buffer (jdgpu_batch_create) */
static __device__ __align__(256) uint8_t scratch[JD_SCRATCH_BYTES];
static uint8_t *g_scratch = nullptr;
static std::mutex g_scratch_mu;

/* The GPU work of a batch's CREATION (header parse, small uploads, fills, the scan gather: a millisecond of
 * kernels) runs on one stream of the highest priority.  On a batch's own stream, or on the null stream, such
 * a kernel waited for the work-groups of whatever batch was decoding at the time -- 8 to 19 ms instead of
 * 0.1 for the header parse under a running decode (rocprofv3 / JDGPU_TRACE) -- and a loader's creations,
 * which have to keep the PCIe link busy, lost that time.  Null when the device offers no priorities. */
static hipStream_t g_hi = nullptr;

extern "C" const char *jdgpu_last_error(void) { return g_err; }

extern "C" int jdgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int jdgpu_init(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        snprintf(g_err, sizeof g_err, "no HIP device visible: the MI355X backend has no CPU fallback");
        return JDGPU_ERR_NO_DEVICE;
    }
    /* One process drives one GPU (images are sharded over processes).  The buffer
     * cache, the pinned staging buffer, the constant tables and the static scratch array are
     * process-wide and belong to the device bound first: binding another one is refused rather
     * than handing device-A buffers to a device-B batch. */
    if (g_device >= 0 && device != g_device) {
        snprintf(g_err, sizeof g_err, "this process is bound to device %d: jdgpu_init(%d) refused (one process per GPU)", g_device, device);
        return JDGPU_ERR_ARG;
    }
    if (device < 0 || device >= n) { snprintf(g_err, sizeof g_err, "device %d out of range (0..%d)", device, n - 1); return JDGPU_ERR_ARG; }
    HIPCHK(hipSetDevice(device));
    if (g_device != device) {
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_zz_cm), jd_zigzag_to_cm, 64));
        /* the device address of the static scratch array: its name is a host-side stand-in */
        void *sp = nullptr;
        HIPCHK(hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)));
        g_scratch = static_cast<uint8_t *>(sp);
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least) {
            if (hipStreamCreateWithPriority(&g_hi, hipStreamNonBlocking, greatest) != hipSuccess) { (void)hipGetLastError(); g_hi = nullptr; }
        }
        g_device = device;
    }
    return JDGPU_OK;
}

static int ensure_device() {
    if (g_device >= 0) { hipError_t e = hipSetDevice(g_device); (void)e; return JDGPU_OK; }
    return jdgpu_init(0);
}

/* Device buffers are recycled.  A 4096 x 1080p batch holds ~56 GB; hipMalloc of that takes from
 * 2 ms to 0.5 s (measured, the latter while another batch decodes) and hipFree waits for the
 * whole device, so a loader that creates batch k+1 while batch k decodes would stall on both.
 * Destroyed batches hand their buffers to a process-wide cache (one process drives one GPU);
 * a creation takes the smallest cached buffer that is big enough and at most 25 % too big.
 * The cache is bounded (JDGPU_POOL_MB, default 131072; 0 turns it off), emptied when an
 * allocation fails, and by jdgpu_trim(). */
struct PoolBuf { void *p; size_t bytes; };
static std::mutex g_pool_mu;
static std::vector<PoolBuf> g_pool_free;
static std::vector<PoolBuf> g_pool_live;
static size_t g_pool_cached = 0;
static size_t pool_limit() {
    static size_t lim = [] {
        const char *e = getenv("JDGPU_POOL_MB");
        return (size_t)(e ? strtoull(e, nullptr, 10) : 131072ull) << 20;
    }();
    return lim;
}
static void pool_trim_locked(size_t keep) {
    while (g_pool_cached > keep && !g_pool_free.empty()) { /* oldest first */
        hipFree(g_pool_free.front().p);
        g_pool_cached -= g_pool_free.front().bytes;
        g_pool_free.erase(g_pool_free.begin());
    }
}
static int pool_alloc(void **out, size_t bytes) {
    *out = nullptr;
    bytes = (bytes + 255) & ~(size_t)255;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t best = g_pool_free.size();
        for (size_t i = 0; i < g_pool_free.size(); ++i) {
            const size_t c = g_pool_free[i].bytes;
            if (c >= bytes && c <= bytes + bytes / 4 + 4096 && (best == g_pool_free.size() || c < g_pool_free[best].bytes)) best = i;
        }
        if (best != g_pool_free.size()) {
            PoolBuf b = g_pool_free[best];
            g_pool_free.erase(g_pool_free.begin() + (long)best);
            g_pool_cached -= b.bytes;
            g_pool_live.push_back(b);
            *out = b.p;
            return JDGPU_OK;
        }
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) { /* memory held by the cache may be what is missing */
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lk(g_pool_mu);
        pool_trim_locked(0);
        e = hipMalloc(out, bytes);
    }
    if (e != hipSuccess) {
        *out = nullptr;
        snprintf(g_err, sizeof g_err, "hipMalloc(%zu bytes) -> %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? JDGPU_ERR_NOMEM : JDGPU_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_live.push_back({*out, bytes});
    return JDGPU_OK;
}
/* the caller has made sure that nothing in flight still uses the buffer */
static void pool_free(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool_live.size(); ++i)
        if (g_pool_live[i].p == p) {
            PoolBuf b = g_pool_live[i];
            g_pool_live[i] = g_pool_live.back();
            g_pool_live.pop_back();
            if (b.bytes > pool_limit()) { hipFree(b.p); return; }
            g_pool_free.push_back(b);
            g_pool_cached += b.bytes;
            pool_trim_locked(pool_limit());
            return;
        }
    hipFree(p); /* not ours: cannot happen, but never leak */
}
extern "C" void jdgpu_trim(void) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    pool_trim_locked(0);
}

/* A batch's stream and events are recycled as well: creating and destroying a stream costs 67 + 62 us, a tenth
 * of a single-image decode (rocprofv3 --hip-trace of bin/latency on birb.jpg). */
#define JD_HSTATUS_WORDS 4096u
struct StreamSet { hipStream_t stream; hipEvent_t ev[JDGPU_NSTAGES + 1]; hipEvent_t created; uint32_t *h_status; };
static std::mutex g_streams_mu;
static std::vector<StreamSet> g_streams_free;
static const size_t JD_STREAMS_KEPT = 16;
static hipError_t stream_set_take(StreamSet *out) {
    {
        std::lock_guard<std::mutex> lk(g_streams_mu);
        if (!g_streams_free.empty()) { *out = g_streams_free.back(); g_streams_free.pop_back(); return hipSuccess; }
    }
    StreamSet s{};
    hipError_t e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
    for (auto &ev : s.ev) if (e == hipSuccess) e = hipEventCreate(&ev);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.created, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&s.h_status), JD_HSTATUS_WORDS * 4u, hipHostMallocDefault);
    if (e != hipSuccess) {
        if (s.created) hipEventDestroy(s.created);
        for (auto &ev : s.ev) if (ev) hipEventDestroy(ev);
        if (s.stream) hipStreamDestroy(s.stream);
        return e;
    }
    *out = s;
    return hipSuccess;
}
/* the caller has synchronised the stream */
static void stream_set_give(const StreamSet &s) {
    {
        std::lock_guard<std::mutex> lk(g_streams_mu);
        if (g_streams_free.size() < JD_STREAMS_KEPT) { g_streams_free.push_back(s); return; }
    }
    for (auto &ev : s.ev) hipEventDestroy(ev);
    hipEventDestroy(s.created);
    hipStreamDestroy(s.stream);
    hipHostFree(s.h_status);
}

template <class T>
static int dalloc(T **p, size_t count) {
    return pool_alloc(reinterpret_cast<void **>(p), (count ? count : 1) * sizeof(T));
}

#define TRY(x) do { int rc_ = (x); if (rc_ != JDGPU_OK) { jdgpu_batch_destroy(b); return rc_; } } while (0)
#define HIPTRY(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            snprintf(g_err, sizeof g_err, "%s:%d %s -> %s", __FILE__, __LINE__, #call,            \
                     hipGetErrorString(e_));                                                      \
            jdgpu_batch_destroy(b);                                                               \
            return JDGPU_ERR_HIP;                                                                 \
        }                                                                                         \
    } while (0)

extern "C" void jdgpu_batch_destroy(jdgpu_batch *b) {
    if (!b) return;
    /* the buffers go back to the cache and may be handed out at once: nothing of this batch may
     * still be running (hipFree used to wait for the whole device here) */
    if (b->stream) hipStreamSynchronize(b->stream);
    if (b->hi_pending && g_hi) hipStreamSynchronize(g_hi);
    if (b->d_meta) { /* these are parts of d_meta */
        b->d_img = nullptr; b->d_luts = nullptr; b->d_hwg = b->d_rpl = b->d_fwg = b->d_wwg = nullptr;
        b->d_pwg = nullptr; b->d_rwg = nullptr; b->d_scan_src = nullptr;
    }
    void *bufs[] = {b->d_raw, b->d_clean, b->d_pix, b->d_img, b->d_luts, b->d_seg_info, b->d_st_in, b->d_st_exit, b->d_list,
                    b->d_status, b->d_cnt, b->d_cp_state, b->d_cp_val, b->d_start_dc, b->d_start_blk, b->d_coef,
                    b->d_hwg, b->d_pwg, b->d_rwg, b->d_wwg, b->d_rpl, b->d_chunk_cnt, b->d_chunk_base, b->d_ivl_u, b->d_ivl_seg,
                    b->d_lane_masks, b->d_files, b->d_scan_src, b->d_fwg, b->d_meta};
    for (void *p : bufs) pool_free(p);
    if (b->stream) {
        StreamSet s;
        s.stream = b->stream;
        for (int k = 0; k <= JDGPU_NSTAGES; ++k) s.ev[k] = b->ev[k];
        s.h_status = b->h_status;
        s.created = b->ev_created;
        stream_set_give(s);
    }
    jd_plan_free(&b->plan);
    delete b;
}

/* Pinned staging memory for the scan upload, kept for the life of the process and reused by every
 * batch creation (fresh pageable memory costs ~300 ms of page faults per 1.5 GB, and uploads from
 * it run at a third of the PCIe rate).  One creation at a time uploads. */
static std::mutex g_stage_mu;
static uint8_t *g_stage = nullptr;
static size_t g_stage_cap = 0;
static int stage_reserve(size_t bytes) {
    if (bytes <= g_stage_cap) return JDGPU_OK;
    if (g_stage) { hipHostFree(g_stage); g_stage = nullptr; g_stage_cap = 0; }
    size_t cap = (bytes + (64u << 20)) & ~(size_t)((1u << 20) - 1);
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&g_stage), cap, hipHostMallocDefault));
    g_stage_cap = cap;
    return JDGPU_OK;
}

/* Pinned host memory handed out to callers (jdgpu_pinned_alloc): a loader reads its files straight into
 * such a region, and batch creation then uploads from where they lie -- one DMA of the span that holds the
 * batch's scans (or whole files), no copy into the library's staging buffer.  That copy was 16 ms on 16
 * threads per 4096 x 1080p batch and the reason a loader could not feed the GPU with a few threads. */
struct PinnedRegion { uint8_t *base; size_t bytes; };
static std::mutex g_pin_mu;
static std::vector<PinnedRegion> g_pinned;

extern "C" void *jdgpu_pinned_alloc(size_t bytes) {
    if (ensure_device()) return nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        snprintf(g_err, sizeof g_err, "hipHostMalloc(%zu bytes) failed", bytes);
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pinned.push_back({static_cast<uint8_t *>(p), bytes ? bytes : 1});
    return p;
}
extern "C" void jdgpu_pinned_free(void *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (size_t i = 0; i < g_pinned.size(); ++i)
            if (g_pinned[i].base == p) { g_pinned[i] = g_pinned.back(); g_pinned.pop_back(); break; }
    }
    hipHostFree(p);
}
/* the registered region that holds [p, p + len), if any */
static bool pinned_region_of(const void *p, size_t len, PinnedRegion *out) {
    const uint8_t *q = static_cast<const uint8_t *>(p);
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (const PinnedRegion &r : g_pinned)
        if (q >= r.base && q + len <= r.base + r.bytes) { *out = r; return true; }
    return false;
}
/* Do the n byte ranges all lie in ONE pinned region, and is the span from the first to the last of them
 * worth a single upload (at most twice their sum: a batch that picks a few files out of a large arena
 * goes through the staging buffer instead)?  lo / hi: the span. */
template <class Range>
static bool pinned_span(size_t n, Range range, const uint8_t **lo, const uint8_t **hi) {
    const uint8_t *a = nullptr, *b = nullptr;
    PinnedRegion reg{nullptr, 0};
    size_t sum = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t *p; size_t len;
        if (!range(i, &p, &len)) continue;
        if (!reg.base && !pinned_region_of(p, len, &reg)) return false;
        if (p < reg.base || p + len > reg.base + reg.bytes) return false;
        if (!a || p < a) a = p;
        if (!b || p + len > b) b = p + len;
        sum += len;
    }
    if (!a || (size_t)(b - a) > 2 * sum + (1u << 20)) return false;
    *lo = a; *hi = b;
    return true;
}

/* set by jdgpu_batch_create_from_files around its call of jdgpu_batch_create: the device copy of
 * the files, which the new batch takes over (so that nothing has to wait for the gather kernel) */
static thread_local uint8_t *g_adopt_files = nullptr;
static thread_local float g_pre_ms[JDGPU_NCREATE] = {0, 0, 0, 0};   /* ... and what it spent before that call */
/* One bulk upload at a time: two that share the PCIe link both take twice as long and end together, and
 * the link then idles while both creations do their host work; one after the other, the second creation's
 * host work hides behind the first one's upload (pipelined loader: 32 -> 27 ms per 4096 x 1080p batch) */
static std::mutex g_upload_mu;

#include "jdgpu_headers.inc"   /* device-side header parsing and scan gather (row f1) */

extern "C" int jdgpu_batch_create(const jdgpu_image_desc *descs, size_t n, unsigned flags, jdgpu_batch **out) {
    if (!out || (!descs && n)) { snprintf(g_err, sizeof g_err, "null argument"); return JDGPU_ERR_ARG; }
    *out = nullptr;
    if ((flags & JDGPU_OUT_4BYTE) && (flags & JDGPU_OUT_PLANAR)) {
        snprintf(g_err, sizeof g_err, "JDGPU_OUT_4BYTE and JDGPU_OUT_PLANAR exclude each other");
        return JDGPU_ERR_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    jdgpu_batch *b = new jdgpu_batch();
    memset(static_cast<void *>(&b->plan), 0, sizeof b->plan);
    b->flags = flags;
    b->stream = nullptr; b->retries = 0; b->timed = false; b->redo = false;
    b->d_files = g_adopt_files; g_adopt_files = nullptr; b->d_scan_src = nullptr;
    b->d_raw = b->d_clean = b->d_pix = nullptr; b->d_img = nullptr; b->d_luts = nullptr;
    b->d_seg_info = b->d_st_in = b->d_st_exit = b->d_list = b->d_list_cnt = b->d_status = b->d_wide = nullptr;
    b->d_cp_state = nullptr; b->d_cp_val = nullptr;
    b->d_cnt = b->d_start_dc = nullptr; b->d_start_blk = nullptr; b->d_coef = nullptr; b->d_hwg = nullptr; b->d_pwg = nullptr;
    b->d_rwg = nullptr; b->d_wwg = nullptr; b->d_rpl = nullptr; b->d_fwg = nullptr; b->d_chunk_cnt = b->d_chunk_base = nullptr; b->d_ivl_u = b->d_ivl_seg = nullptr; b->d_lane_masks = nullptr; b->any_rst = false; b->d_meta = nullptr; b->h_status = nullptr; b->status_copied = false; b->ev_created = nullptr; b->hi_pending = false;
    for (auto &e : b->ev) e = nullptr;
    const bool trace = getenv("JDGPU_TRACE") != nullptr;   /* development aid: where batch creation spends its time */
    auto now = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    for (int k = 0; k < JDGPU_NCREATE; ++k) { b->create_ms[k] = g_pre_ms[k]; g_pre_ms[k] = 0; }
    double t0 = now();
    auto lap = [&](int part, const char *what) {
        const double t1 = now();
        b->create_ms[part] += (float)(t1 - t0);
        if (trace) fprintf(stderr, "[jdgpu] %s %.1f ms\n", what, t1 - t0);
        t0 = t1;
    };
    TRY(jd_plan_build(descs, n, (flags & JDGPU_OUT_4BYTE) ? 4u : 3u, (flags & JDGPU_OUT_PITCHED) ? 1u : 0u, &b->plan, g_err, sizeof g_err));
    jd_plan &P = b->plan;
    lap(JDGPU_CREATE_PLAN, "plan");
    b->status.assign(n, 0);
    for (size_t i = 0; i < n; ++i) if (!P.img[i].ncomp) b->status[i] = JDGPU_IMG_UNSUPPORTED;

    {
        StreamSet s;
        HIPTRY(stream_set_take(&s));
        b->stream = s.stream;
        for (int k = 0; k <= JDGPU_NSTAGES; ++k) b->ev[k] = s.ev[k];
        b->h_status = s.h_status;
        b->ev_created = s.created;
    }
    TRY(dalloc(&b->d_raw, P.raw_bytes));
    TRY(dalloc(&b->d_clean, P.clean_bytes));
    TRY(dalloc(&b->d_pix, (size_t)P.pix_bytes + 256));
    TRY(dalloc(&b->d_seg_info, (size_t)P.n_segs_cap + 1));
    TRY(dalloc(&b->d_st_in, (size_t)P.n_segs_cap));
    TRY(dalloc(&b->d_st_exit, (size_t)P.n_segs_cap));
    TRY(dalloc(&b->d_cnt, (size_t)P.n_segs_cap));
    TRY(dalloc(&b->d_cp_state, (size_t)P.n_segs_cap * JD_CP));
    TRY(dalloc(&b->d_cp_val, (size_t)P.n_segs_cap * (JD_CP + 1u)));
    TRY(dalloc(&b->d_start_dc, (size_t)P.n_segs_cap));
    TRY(dalloc(&b->d_start_blk, (size_t)P.n_segs_cap));
    TRY(dalloc(&b->d_list, (size_t)P.n_segs_cap));
    /* one block, so that a decode clears it with one fill: the status words [n], the wide flags [3n] (second
     * part: where a test sends the flags to lose them; third: the images a later pass has to do again, redo_of)
     * and the repair lists' counts [JD_REPAIR_ROUNDS][n] */
    TRY(dalloc(&b->d_status, (size_t)JD_CLEARED_WORDS * (n ? n : 1)));
    b->d_wide = b->d_status + (n ? n : 1);
    b->d_list_cnt = b->d_wide + 3 * (n ? n : 1);
    TRY(dalloc(&b->d_coef, (size_t)P.total_blocks * 64));
    TRY(dalloc(&b->d_chunk_cnt, (size_t)P.n_rwg));
    TRY(dalloc(&b->d_lane_masks, (size_t)P.n_rwg * 256u));
    TRY(dalloc(&b->d_chunk_base, (size_t)P.n_rwg));
    TRY(dalloc(&b->d_ivl_u, (size_t)P.n_u));
    TRY(dalloc(&b->d_ivl_seg, (size_t)P.n_intervals));
    for (size_t i = 0; i < n; ++i) b->any_rst |= P.img[i].ncomp && P.img[i].ri_blocks != 0;
    lap(JDGPU_CREATE_OTHER, "device allocations");

    /* the scans.  (a) They lie in device memory already (JDGPU_SCAN_ON_DEVICE).  (b) They lie in ONE region of
     * pinned host memory from jdgpu_pinned_alloc: the span that holds them goes up in one DMA from where it
     * is, and the scans are picked out of it on the device.  (c) Anywhere else: gathered by host threads into
     * the library's pinned staging buffer, one upload. */
    std::vector<uint64_t> src;
    bool on_device = (flags & JDGPU_SCAN_ON_DEVICE) != 0;
    /* held from the span upload to the submission of the plan's arrays: submitted after ANOTHER creation's bulk
     * upload, those few MB wait for all of it (28 ms instead of 2), and two creating threads that meet once that
     * way stay in step from then on, the link idle while both plan */
    std::unique_lock<std::mutex> up(g_upload_mu, std::defer_lock);
    if (on_device) {
        src.resize(n ? n : 1);
        for (size_t i = 0; i < n; ++i) src[i] = (uint64_t)(uintptr_t)descs[i].scan;
    } else {
        const uint8_t *lo = nullptr, *hi = nullptr;
        const bool span = pinned_span(n, [&](size_t i, const uint8_t **p, size_t *len) {
            if (!P.img[i].ncomp) return false;
            *p = descs[i].scan; *len = descs[i].scan_len;
            return true;
        }, &lo, &hi);
        if (span) {
            uint8_t *d_span = nullptr;
            TRY(pool_alloc(reinterpret_cast<void **>(&d_span), (size_t)(hi - lo) + 16));
            b->d_files = d_span;   /* released with the batch */
            up.lock();   /* released below, or by any return */
            HIPTRY(hipMemcpy(d_span, lo, (size_t)(hi - lo), hipMemcpyHostToDevice));
            lap(JDGPU_CREATE_UPLOAD, "span upload");
            src.resize(n ? n : 1);
            for (size_t i = 0; i < n; ++i) src[i] = P.img[i].ncomp ? (uint64_t)(uintptr_t)(d_span + (descs[i].scan - lo)) : 0;
            on_device = true;
        } else {
            std::lock_guard<std::mutex> lock(g_stage_mu);
            TRY(stage_reserve(P.raw_bytes));
            P.raw = g_stage; P.raw_owned = 0;
            jd_plan_gather(&P, descs, n);
            lap(JDGPU_CREATE_STAGE, "gather");
            HIPTRY(hipMemcpy(b->d_raw, P.raw, P.raw_bytes, hipMemcpyHostToDevice));
            P.raw = nullptr;
            lap(JDGPU_CREATE_UPLOAD, "scan upload");
        }
    }
    /* the plan's arrays.  A small batch (a single image above all) takes them up as ONE buffer and one copy:
     * eight allocations and nine copies of a few hundred bytes each cost ~10 us apiece, 0.1 ms of a
     * single-image decode's 1.2. */
    struct Part { void **dev; const void *host; size_t bytes; };
    const Part parts[] = {
        {reinterpret_cast<void **>(&b->d_img), P.img, n * sizeof(jd_image)},
        {reinterpret_cast<void **>(&b->d_luts), P.luts, (size_t)P.n_luts * sizeof(jd_lut)},
        {reinterpret_cast<void **>(&b->d_hwg), P.hwg, (size_t)P.n_hwg * JD_WG_WAVES * sizeof(jd_huff_wg)},
        {reinterpret_cast<void **>(&b->d_rpl), P.rpl, (size_t)P.n_hwg * JD_WG_WAVES * sizeof(jd_huff_wg)},
        {reinterpret_cast<void **>(&b->d_fwg), P.fwg, (size_t)P.n_fwg * JD_WG_WAVES * sizeof(jd_huff_wg)},
        {reinterpret_cast<void **>(&b->d_pwg), P.pwg, (size_t)P.n_pwg * sizeof(jd_pix_wg)},
        {reinterpret_cast<void **>(&b->d_rwg), P.rwg, (size_t)P.n_rwg * sizeof(jd_raw_wg)},
        {reinterpret_cast<void **>(&b->d_wwg), P.wwg, (size_t)P.n_wwg * P.wwg_waves * sizeof(jd_huff_wg)},
        {reinterpret_cast<void **>(&b->d_scan_src), src.data(), on_device ? (n ? n : 1) * sizeof(uint64_t) : 0},
    };
    size_t meta_bytes = 0;
    for (const Part &pt : parts) meta_bytes += (pt.bytes + 255) & ~(size_t)255;
    std::vector<uint8_t> &meta = b->meta_host;   /* read by the copy below: lives as long as the batch */
    const bool packed = meta_bytes <= JD_META_PACK_MAX;
    /* where the creation's GPU work goes: a large batch's on the high-priority stream (see g_hi), a small one's
     * on its own stream (no one to wait for, and no event between two streams in a single image's latency) */
    const hipStream_t cs = (g_hi && !packed) ? g_hi : b->stream;
    b->hi_pending = cs != b->stream;
    if (packed) {
        TRY(pool_alloc(reinterpret_cast<void **>(&b->d_meta), meta_bytes + 256));
        meta.assign(meta_bytes + 256, 0);
        size_t at = 0;
        for (const Part &pt : parts) {
            *pt.dev = b->d_meta + at;   /* an empty part points at the next one's start and is never read */
            if (pt.bytes) memcpy(meta.data() + at, pt.host, pt.bytes);
            at += (pt.bytes + 255) & ~(size_t)255;
        }
        HIPTRY(hipMemcpyAsync(b->d_meta, meta.data(), meta.size(), hipMemcpyHostToDevice, cs));
    } else {
        for (const Part &pt : parts) {
            if (!pt.bytes && pt.dev == reinterpret_cast<void **>(&b->d_scan_src)) continue;
            TRY(pool_alloc(pt.dev, pt.bytes ? pt.bytes : 1));
            if (pt.bytes) HIPTRY(hipMemcpyAsync(*pt.dev, pt.host, pt.bytes, hipMemcpyHostToDevice, cs));
        }
    }
    if (up.owns_lock()) up.unlock();
    if (on_device) {
        /* copied on the device into the raw buffer the unstuff stage reads (16-byte aligned starts, zeros after
         * each scan), on the batch's own stream */
        HIPTRY(hipMemsetAsync(b->d_raw, 0, P.raw_bytes, cs));
        if (P.n_rwg) {
            hipLaunchKernelGGL(k_gather_scans, dim3(P.n_rwg), dim3(256), 0, cs, b->d_img, b->d_rwg, b->d_scan_src, b->d_raw);
            HIPTRY(hipGetLastError());
        }
    }
    /* The output is cleared only where its content is not all written by a decode: padded rows
     * (JDGPU_OUT_PITCHED: the padding is zero and never written) and images the plan rejected.  A failed
     * image is zeroed when its status is known (jdgpu_batch_sync); clearing 25 GB per batch here cost
     * 4.3 ms of the GPU, a fifth of a decode. */
    if (flags & JDGPU_OUT_PITCHED) {
        HIPTRY(hipMemsetAsync(b->d_pix, 0, (size_t)P.pix_bytes + 256, cs));
    } else {
        for (size_t i = 0; i < n; ++i)
            if (!P.img[i].ncomp)
                HIPTRY(hipMemsetAsync(b->d_pix + P.img[i].pix_off, 0, (size_t)P.img[i].stride * P.img[i].height * P.px_bytes, cs));
    }
    HIPTRY(hipMemsetAsync(b->d_status, 0, (size_t)JD_CLEARED_WORDS * (n ? n : 1) * 4, cs));
    /* Everything above reads host memory of this call (the plan's arrays, the caller's scans) or, with
     * JDGPU_SCAN_ON_DEVICE, device memory the caller may release when this returns: wait for it.  The
     * exception: files the batch owns (jdgpu_batch_create_from_files adopted them) and plan arrays, which live
     * as long as the batch -- but the pageable plan arrays are staged by the runtime at enqueue time only when
     * small; so wait in every case, it is the upload's duration and nothing else. */
    /* ... except where nothing of that kind is left: a packed plan (its host copy belongs to the batch) whose
     * scans the library uploaded itself.  A single image's creation then ends without a wait. */
    if (cs != b->stream) {   /* the batch's stream starts behind the creation's work; the host waits for that work alone */
        HIPTRY(hipEventRecord(b->ev_created, cs));
        HIPTRY(hipStreamWaitEvent(b->stream, b->ev_created, 0));
        HIPTRY(hipEventSynchronize(b->ev_created));
        b->hi_pending = false;
    } else if (!packed || (flags & JDGPU_SCAN_ON_DEVICE)) {
        HIPTRY(hipStreamSynchronize(b->stream));
    }
    lap(JDGPU_CREATE_UPLOAD, "uploads, device gather");
    if (trace) fprintf(stderr, "[jdgpu]   (%.2f GB of scans, %u distinct Huffman tables)\n", P.raw_bytes / 1e9, P.n_luts);
    /* the host copies of the bulky arrays are not needed any more */
    free(P.luts); P.luts = nullptr;
    *out = b;
    return JDGPU_OK;
}

extern "C" int jdgpu_batch_create_ms(const jdgpu_batch *b, float ms[JDGPU_NCREATE]) {
    if (!b || !ms) return JDGPU_ERR_ARG;
    memcpy(ms, b->create_ms, sizeof b->create_ms);
    return JDGPU_OK;
}

/* JDGPU_NARROW=0 (tests, A/B timing): every image takes the wide write kernel and 128-byte units */
static bool narrow_units() {
    static const bool narrow = [] { const char *e = getenv("JDGPU_NARROW"); return !e || atoi(e) != 0; }();
    return narrow;
}

/* a pass after the first does only the images the first left unfinished (a broken segment chain, a
 * narrow unit that overflowed): the third part of d_wide holds one word per image, see jdgpu_batch_sync */
static const uint32_t *redo_of(const jdgpu_batch *b) {
    return b->redo ? b->d_wide + 2 * (b->plan.n ? b->plan.n : 1) : nullptr;
}

template <bool OUT_YCBCR, uint32_t MODE, uint32_t FMT>
static void launch_pixel_mode(jdgpu_batch *b, hipStream_t st) {
    const uint32_t *f = b->plan.pwg_mode_first;
    if (f[MODE + 1] > f[MODE])
        hipLaunchKernelGGL((k_pixel<OUT_YCBCR, MODE, FMT>), dim3(f[MODE + 1] - f[MODE]), dim3(64 * JD_PIX_WAVES), 0, st,
                           b->d_img, b->d_pwg + f[MODE], b->d_coef, b->d_pix, b->d_status, narrow_units() ? b->d_wide : nullptr,
                           redo_of(b), b->plan.pix_rows);
}
template <bool OUT_YCBCR, uint32_t FMT>
static void launch_pixel_modes(jdgpu_batch *b, hipStream_t st) {
    launch_pixel_mode<OUT_YCBCR, JD_PIX_GENERIC, FMT>(b, st);
    launch_pixel_mode<OUT_YCBCR, JD_PIX_420, FMT>(b, st);
    launch_pixel_mode<OUT_YCBCR, JD_PIX_444, FMT>(b, st);
    launch_pixel_mode<OUT_YCBCR, JD_PIX_422, FMT>(b, st);
    launch_pixel_mode<OUT_YCBCR, JD_PIX_440, FMT>(b, st);
    launch_pixel_mode<OUT_YCBCR, JD_PIX_GRAY, FMT>(b, st);
}
template <bool OUT_YCBCR>
static void launch_pixel_formats(jdgpu_batch *b, hipStream_t st) {
    if (b->flags & JDGPU_OUT_4BYTE) launch_pixel_modes<OUT_YCBCR, JD_FMT_4>(b, st);
    else if (b->flags & JDGPU_OUT_PLANAR) launch_pixel_modes<OUT_YCBCR, JD_FMT_PLANAR>(b, st);
    else launch_pixel_modes<OUT_YCBCR, JD_FMT_3>(b, st);
}
static void launch_pixel(jdgpu_batch *b, hipStream_t st) {
    if (b->flags & JDGPU_OUT_YCBCR) launch_pixel_formats<true>(b, st); else launch_pixel_formats<false>(b, st);
}

/* JDGPU_TEST_DROP_WIDE: the scan kernel's flags go nowhere, so that the narrow write kernel meets
 * values it cannot hold and its safety net (mark the image, run the pass again) is exercised */
static uint32_t *wide_flags_of(jdgpu_batch *b) {
    static const bool drop = getenv("JDGPU_TEST_DROP_WIDE") != nullptr;
    return drop ? b->d_wide + (b->plan.n ? b->plan.n : 1) : b->d_wide;
}

static void launch_write(jdgpu_batch *b, hipStream_t st) {
    jd_plan &P = b->plan;
#define WRITE_ARGS b->d_img, b->d_luts, b->d_clean, b->d_seg_info, b->d_wwg, b->d_st_in, b->d_st_exit, b->d_start_blk, \
                   b->d_start_dc, b->d_coef, b->d_status, b->d_wide, redo_of(b), b->d_cp_state, b->d_cp_val
    const bool narrow = narrow_units();
    if (P.wwg_waves == JD_WWG_WAVES_BIG) {
        const dim3 grid(P.n_wwg), block(64 * JD_WWG_WAVES_BIG);
        if (narrow) { /* int8 buffers where the image allows, the wide kernel beside it for the rest */
            hipLaunchKernelGGL((k_huff_write<JD_WWG_WAVES_BIG, JD_WWG_SLOTS_BIG, true, true>), grid, block, 0, st, WRITE_ARGS);
            hipLaunchKernelGGL((k_huff_write<JD_WWG_WAVES_BIG, JD_WWG_SLOTS_BIG, false, true>), grid, block, 0, st, WRITE_ARGS);
        } else {
            hipLaunchKernelGGL((k_huff_write<JD_WWG_WAVES_BIG, JD_WWG_SLOTS_BIG, false, false>), grid, block, 0, st, WRITE_ARGS);
        }
    } else {
        const dim3 grid(P.n_wwg), block(64 * JD_WG_WAVES);
        if (P.write_sub > 1u) { /* a small batch: JD_CP + 1 lanes per segment */
            if (narrow) {
                hipLaunchKernelGGL((k_huff_write<JD_WG_WAVES, JD_LUT_SLOTS, true, true, JD_CP + 1u>), grid, block, 0, st, WRITE_ARGS);
                hipLaunchKernelGGL((k_huff_write<JD_WG_WAVES, JD_LUT_SLOTS, false, true, JD_CP + 1u>), grid, block, 0, st, WRITE_ARGS);
            } else {
                hipLaunchKernelGGL((k_huff_write<JD_WG_WAVES, JD_LUT_SLOTS, false, false, JD_CP + 1u>), grid, block, 0, st, WRITE_ARGS);
            }
        } else if (narrow) { /* 35 instead of 51 KB per work-group: 4 instead of 3 per CU */
            hipLaunchKernelGGL((k_huff_write<JD_WG_WAVES, JD_LUT_SLOTS, true, true>), grid, block, 0, st, WRITE_ARGS);
            hipLaunchKernelGGL((k_huff_write<JD_WG_WAVES, JD_LUT_SLOTS, false, true>), grid, block, 0, st, WRITE_ARGS);
        } else {
            hipLaunchKernelGGL((k_huff_write<JD_WG_WAVES, JD_LUT_SLOTS, false, false>), grid, block, 0, st, WRITE_ARGS);
        }
    }
#undef WRITE_ARGS
}

static int enqueue_huffman(jdgpu_batch *b, bool first_pass) {
    jd_plan &P = b->plan;
    hipStream_t st = b->stream;
    const dim3 grid(P.n_hwg), block(JD_WG_SEGS);
    if (first_pass)
        hipLaunchKernelGGL(k_huff_first, dim3(P.n_fwg), block, 0, st, b->d_img, b->d_luts, b->d_clean, b->d_seg_info, b->d_fwg,
                           b->d_st_in, b->d_st_exit, b->d_cnt, b->d_cp_state, b->d_cp_val);
    /* chain check + repair; on a later pass (k_huff_write still found a break) the same again */
    if (!first_pass) HIPCHK(hipMemsetAsync(b->d_list_cnt, 0, (size_t)JD_REPAIR_ROUNDS * P.n * 4, st)); /* jdgpu_batch_decode has */
    /* JDGPU_TEST_REPAIR_ROUNDS (tests): fewer built-in rounds in the first pass, so that chains are still
     * broken when k_huff_write checks them and the second-pass machinery of jdgpu_batch_sync has work */
    uint32_t rounds = JD_REPAIR_ROUNDS;
    if (first_pass)
        if (const char *e = getenv("JDGPU_TEST_REPAIR_ROUNDS")) rounds = min(rounds, (uint32_t)strtoul(e, nullptr, 10));
    for (uint32_t r = 0; r < rounds; ++r) {
        uint32_t *lc = b->d_list_cnt + (size_t)r * P.n;
        hipLaunchKernelGGL(k_chain_check, grid, block, 0, st, b->d_img, b->d_seg_info, b->d_hwg, b->d_st_in, b->d_st_exit,
                           b->d_list, lc, r ? lc - P.n : (const uint32_t *)nullptr);
        hipLaunchKernelGGL(k_huff_repair, grid, block, 0, st, b->d_img, b->d_luts, b->d_clean, b->d_seg_info, b->d_rpl,
                           b->d_list, lc, b->d_st_in, b->d_st_exit, b->d_cnt, b->d_cp_state, b->d_cp_val);
    }
    return JDGPU_OK;
}

extern "C" int jdgpu_batch_decode(jdgpu_batch *b) {
    if (!b) return JDGPU_ERR_ARG;
    jd_plan &P = b->plan;
    if (P.n == 0) return JDGPU_OK;
    const bool prof = (b->flags & JDGPU_PROFILE) != 0;
    b->retries = 0;
    b->redo = false;
    hipStream_t st = b->stream;
    /* status, wide flags (set by the light decodes, kept over repair passes) and repair counts */
    HIPCHK(hipMemsetAsync(b->d_status, 0, (size_t)JD_CLEARED_WORDS * P.n * 4, st));
    if (prof) HIPCHK(hipEventRecord(b->ev[0], st));
    if (P.n_rwg) {
        HIPCHK(hipMemsetAsync(b->d_ivl_u, 0xFF, (size_t)P.n_u * 4, st));
        hipLaunchKernelGGL(k_unstuff_count, dim3((P.n_rwg + JD_RAW_PER_WG - 1u) / JD_RAW_PER_WG), dim3(256), 0, st, b->d_img, b->d_rwg, P.n_rwg, b->d_raw, b->d_chunk_cnt, b->d_lane_masks);
        hipLaunchKernelGGL(k_unstuff_chunks, dim3((unsigned)P.n), dim3(256), 0, st, b->d_img, b->d_chunk_cnt, b->d_chunk_base, b->d_ivl_u);
        if (b->any_rst)
            hipLaunchKernelGGL(k_unstuff_marks, dim3(P.n_rwg), dim3(256), 0, st, b->d_img, b->d_rwg, b->d_raw, b->d_chunk_cnt,
                               b->d_chunk_base, b->d_lane_masks, b->d_ivl_u);
        hipLaunchKernelGGL(k_unstuff_layout, dim3((unsigned)P.n), dim3(256), 0, st, b->d_img, b->d_ivl_u, b->d_ivl_seg,
                           b->d_seg_info, b->d_clean);
        hipLaunchKernelGGL(k_unstuff_scatter, dim3((P.n_rwg + JD_RAW_PER_WG - 1u) / JD_RAW_PER_WG), dim3(256), 0, st, b->d_img, b->d_rwg, P.n_rwg, b->d_raw, b->d_chunk_base,
                           b->d_ivl_u, b->d_ivl_seg, b->d_lane_masks, b->d_clean);
    }
    if (prof) HIPCHK(hipEventRecord(b->ev[1], st));
    if (P.n_hwg) { if (int rc = enqueue_huffman(b, true)) return rc; }
    if (prof) HIPCHK(hipEventRecord(b->ev[2], st));
    hipLaunchKernelGGL(k_seg_scan, dim3((unsigned)P.n), dim3(256), 0, st, b->d_img, b->d_seg_info, b->d_cnt, b->d_start_blk, b->d_start_dc, b->d_status, wide_flags_of(b));
    if (prof) HIPCHK(hipEventRecord(b->ev[3], st));
    if (P.n_hwg) launch_write(b, st);
    if (prof) HIPCHK(hipEventRecord(b->ev[4], st));
    if (P.n_pwg) {
        launch_pixel(b, st);
    }
    if (prof) HIPCHK(hipEventRecord(b->ev[5], st));
    HIPCHK(hipGetLastError());
    b->status_copied = b->h_status && P.n <= JD_HSTATUS_WORDS;
    if (b->status_copied) HIPCHK(hipMemcpyAsync(b->h_status, b->d_status, P.n * 4, hipMemcpyDeviceToHost, st));
    b->timed = prof;
    return JDGPU_OK;
}

extern "C" int jdgpu_batch_sync(jdgpu_batch *b) {
    if (!b) return JDGPU_ERR_ARG;
    jd_plan &P = b->plan;
    HIPCHK(hipStreamSynchronize(b->stream));
    if (P.n == 0) return JDGPU_OK;
    std::vector<uint32_t> st(P.n);
    for (int attempt = 0;; ++attempt) {
        if (attempt == 0 && b->status_copied) memcpy(st.data(), b->h_status, P.n * 4);   /* came with the decode */
        else HIPCHK(hipMemcpy(st.data(), b->d_status, P.n * 4, hipMemcpyDeviceToHost));
        b->status_copied = false;
        bool unsynced = false;
        for (size_t i = 0; i < P.n; ++i) unsynced |= (st[i] & JDGPU_IMG_UNSYNCED) != 0;
        if (!unsynced) break;
        if (getenv("JDGPU_TEST_NO_RETRY")) break;   /* development aid (tools/guess_stats.py): leave the first pass's states as they are */
        if (attempt > 8) { snprintf(g_err, sizeof g_err, "segment chain did not converge"); return JDGPU_ERR_NOT_CONVERGED; }
        /* the chain of start states was still broken after the built-in repair rounds: repeat
         * them until a round lists nothing (every round mends at least the first break of each
         * image, so this ends), then redo scan + write + pixel */
        ++b->retries;
        hipStream_t s = b->stream;
        /* only the images marked unfinished are done again: they start from a clean status word,
         * the others keep theirs (and their coefficients and pixels) */
        {
            std::vector<uint32_t> redo(P.n), keep(P.n);
            for (size_t i = 0; i < P.n; ++i) {
                redo[i] = (st[i] & JDGPU_IMG_UNSYNCED) ? 1u : 0u;
                keep[i] = redo[i] ? 0u : st[i];
            }
            HIPCHK(hipMemcpy(b->d_wide + 2 * P.n, redo.data(), P.n * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(b->d_status, keep.data(), P.n * 4, hipMemcpyHostToDevice));
            b->redo = true;
        }
        std::vector<uint32_t> lc(P.n);
        for (uint32_t pass = 0;; ++pass) {
            if (pass > P.n_segs_cap) { snprintf(g_err, sizeof g_err, "segment chain did not converge"); return JDGPU_ERR_NOT_CONVERGED; }
            if (int rc = enqueue_huffman(b, false)) return rc;
            HIPCHK(hipMemcpyAsync(lc.data(), b->d_list_cnt + (size_t)(JD_REPAIR_ROUNDS - 1) * P.n, P.n * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            bool clean = true;
            for (size_t i = 0; i < P.n; ++i) clean &= lc[i] == 0;
            if (clean) break;
        }
        hipLaunchKernelGGL(k_seg_scan, dim3((unsigned)P.n), dim3(256), 0, s, b->d_img, b->d_seg_info, b->d_cnt, b->d_start_blk, b->d_start_dc, b->d_status, wide_flags_of(b));
        launch_write(b, s);
        launch_pixel(b, s);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
    }
    for (size_t i = 0; i < P.n; ++i) b->status[i] = st[i] | (P.img[i].ncomp ? 0u : (uint32_t)JDGPU_IMG_UNSUPPORTED);
    /* a failed image reads all zero (include/jpeg_gpu.h): the pixel kernel skipped it, so its region holds
     * whatever the recycled buffer held; cleared here, where the failure is known, instead of clearing
     * every batch's whole output at creation */
    if (!(b->flags & JDGPU_OUT_PITCHED)) {
        bool any = false;
        for (size_t i = 0; i < P.n; ++i)
            if (st[i] & (JDGPU_IMG_EOF | JDGPU_IMG_NOSYMBOL | JDGPU_IMG_RANGE)) {
                HIPCHK(hipMemsetAsync(b->d_pix + P.img[i].pix_off, 0, (size_t)P.img[i].stride * P.img[i].height * P.px_bytes, b->stream));
                any = true;
            }
        if (any) HIPCHK(hipStreamSynchronize(b->stream));
    }
    if (b->timed) {
        for (int s = 0; s < 5; ++s) HIPCHK(hipEventElapsedTime(&b->ms[s], b->ev[s], b->ev[s + 1]));
        HIPCHK(hipEventElapsedTime(&b->ms[JDGPU_STAGE_TOTAL], b->ev[0], b->ev[5]));
    }
    return JDGPU_OK;
}

extern "C" size_t jdgpu_batch_size(const jdgpu_batch *b) { return b ? b->plan.n : 0; }
extern "C" uint32_t jdgpu_batch_status(const jdgpu_batch *b, size_t i) { return (b && i < b->plan.n) ? b->status[i] : ~0u; }
extern "C" uint64_t jdgpu_batch_total_pixels(const jdgpu_batch *b) { return b ? b->plan.total_pixels : 0; }
extern "C" uint64_t jdgpu_batch_total_blocks(const jdgpu_batch *b) { return b ? b->plan.total_blocks : 0; }
extern "C" uint64_t jdgpu_batch_total_scan_bytes(const jdgpu_batch *b) {
    uint64_t t = 0;
    if (b) for (size_t i = 0; i < b->plan.n; ++i) t += b->plan.img[i].raw_len;
    return t;
}
extern "C" size_t jdgpu_batch_image_blocks(const jdgpu_batch *b, size_t i) { return (b && i < b->plan.n) ? b->plan.img[i].n_blocks : 0; }
extern "C" int jdgpu_batch_retries(const jdgpu_batch *b) { return b ? b->retries : 0; }
/* development aid: segments on the repair lists of the last pass, per round, summed over the images */
extern "C" int jdgpu_batch_repair_counts(jdgpu_batch *b, uint32_t *out, size_t rounds) {
    if (!b || !out) return JDGPU_ERR_ARG;
    const size_t n = b->plan.n;
    std::vector<uint32_t> h((size_t)JD_REPAIR_ROUNDS * (n ? n : 1));
    HIPCHK(hipMemcpy(h.data(), b->d_list_cnt, h.size() * 4, hipMemcpyDeviceToHost));
    for (size_t r = 0; r < rounds; ++r) {
        out[r] = 0;
        for (size_t i = 0; r < JD_REPAIR_ROUNDS && i < n; ++i) out[r] += h[r * n + i];
    }
    return JDGPU_OK;
}

extern "C" void *jdgpu_batch_device_pixels(jdgpu_batch *b, size_t i) {
    if (!b || i >= b->plan.n) return nullptr;
    return b->d_pix + b->plan.img[i].pix_off;
}

extern "C" int jdgpu_batch_read_pixels(jdgpu_batch *b, size_t i, void *dst) {
    if (!b || i >= b->plan.n || !dst) return JDGPU_ERR_ARG;
    const jd_image &im = b->plan.img[i];
    HIPCHK(hipMemcpy(dst, b->d_pix + im.pix_off, (size_t)im.stride * im.height * b->plan.px_bytes, hipMemcpyDeviceToHost));
    return JDGPU_OK;
}

extern "C" size_t jdgpu_batch_pixel_bytes(const jdgpu_batch *b, size_t i) {
    if (!b || i >= b->plan.n) return 0;
    return (size_t)b->plan.img[i].stride * b->plan.img[i].height * b->plan.px_bytes;
}

extern "C" int jdgpu_batch_image_size(const jdgpu_batch *b, size_t i, unsigned *width, unsigned *height) {
    if (!b || i >= b->plan.n) return JDGPU_ERR_ARG;
    if (width) *width = b->plan.img[i].width;
    if (height) *height = b->plan.img[i].height;
    return JDGPU_OK;
}

extern "C" size_t jdgpu_batch_pixel_pitch(const jdgpu_batch *b, size_t i) {
    if (!b || i >= b->plan.n) return 0;
    return (size_t)b->plan.img[i].stride * ((b->flags & JDGPU_OUT_PLANAR) ? 1u : b->plan.px_bytes);
}

extern "C" int jdgpu_batch_read_coefficients(jdgpu_batch *b, size_t i, int16_t *dst) {
    if (!b || i >= b->plan.n || !dst) return JDGPU_ERR_ARG;
    const jd_image &im = b->plan.img[i];
    size_t nb = im.n_blocks;
    std::vector<int16_t> tmp(nb * 64);
    uint32_t is_wide = 1;
    if (narrow_units()) HIPCHK(hipMemcpy(&is_wide, b->d_wide + i, 4, hipMemcpyDeviceToHost));
    if (!is_wide) {
        /* narrow image: 64-byte units of int8, the 12-bit DC value and the 4-bit last AC coefficient
         * sharing bytes 0 and 63 (k_huff_write) */
        std::vector<int8_t> t8(nb * JD_NARROW_UNIT);
        HIPCHK(hipMemcpy(t8.data(), b->d_coef + (size_t)im.blk_off * 64, nb * JD_NARROW_UNIT, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < nb; ++k) {
            const int8_t *u = &t8[k * 64];
            for (int c = 0; c < 64; ++c) tmp[k * 64 + c] = u[c];
            const int dc = ((uint8_t)u[0] | (((uint8_t)u[63] & 15) << 8));
            tmp[k * 64] = (int16_t)((dc ^ 0x800) - 0x800);   /* sign-extend 12 bits */
            tmp[k * 64 + 63] = (int16_t)(u[63] >> 4);
        }
    } else {
        HIPCHK(hipMemcpy(tmp.data(), b->d_coef + (size_t)im.blk_off * 64, nb * 128, hipMemcpyDeviceToHost));
    }
    for (size_t k = 0; k < nb; ++k)
        for (int z = 0; z < 64; ++z) dst[k * 64 + z] = tmp[k * 64 + jd_zigzag_to_cm[z]];
    return JDGPU_OK;
}

extern "C" long jdgpu_batch_debug_segments(jdgpu_batch *b, size_t i, uint8_t *clean, size_t clean_cap,
                                           uint32_t *seg_info, size_t seg_cap) {
    if (!b || i >= b->plan.n) return JDGPU_ERR_ARG;
    jd_image im;
    HIPCHK(hipMemcpy(&im, b->d_img + i, sizeof im, hipMemcpyDeviceToHost));
    if ((size_t)im.n_segs > seg_cap || (size_t)im.n_segs * JD_SEG_BYTES > clean_cap) return JDGPU_ERR_ARG;
    {
        size_t groups = ((size_t)im.n_segs + 63) / 64, bytes = groups * 64 * JD_SEG_BYTES;
        std::vector<uint8_t> t(bytes ? bytes : 1);
        HIPCHK(hipMemcpy(t.data(), b->d_clean + im.clean_off, bytes, hipMemcpyDeviceToHost));
        for (uint32_t s = 0; s < im.n_segs; ++s)
            for (uint32_t o = 0; o < JD_SEG_BYTES; ++o) clean[(size_t)s * JD_SEG_BYTES + o] = t[clean_byte_addr(s, o)];
    }
    HIPCHK(hipMemcpy(seg_info, b->d_seg_info + im.seg_off, (size_t)im.n_segs * 4, hipMemcpyDeviceToHost));
    return (long)im.n_segs;
}

/* development aid: the start state each segment of image i was last decoded from and the exit state it
 * reached (packed: bits 5:0 overshoot, 12:6 zigzag index, 18:13 data unit of the MCU, 19 error seen) */
extern "C" long jdgpu_batch_debug_states(jdgpu_batch *b, size_t i, uint32_t *st_in, uint32_t *st_exit, size_t cap) {
    if (!b || i >= b->plan.n || !st_in || !st_exit) return JDGPU_ERR_ARG;
    jd_image im;
    HIPCHK(hipMemcpy(&im, b->d_img + i, sizeof im, hipMemcpyDeviceToHost));
    if ((size_t)im.n_segs > cap) return JDGPU_ERR_ARG;
    HIPCHK(hipMemcpy(st_in, b->d_st_in + im.seg_off, (size_t)im.n_segs * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(st_exit, b->d_st_exit + im.seg_off, (size_t)im.n_segs * 4, hipMemcpyDeviceToHost));
    return (long)im.n_segs;
}

extern "C" int jdgpu_batch_stage_ms(jdgpu_batch *b, float ms[JDGPU_NSTAGES]) {
    if (!b || !ms || !b->timed) return JDGPU_ERR_ARG;
    memcpy(ms, b->ms, sizeof b->ms);
    return JDGPU_OK;
}

/* in-place conversion of a host pixel[] buffer on the device: upload, kernel, download */
static int convert_host_pixels(void (*kernel)(uint8_t *, size_t), const char *what, void *host_pixels,
                               unsigned width, unsigned height) {
    int rc = ensure_device();
    if (rc) return rc;
    size_t npix = (size_t)width * height;
    if (!npix) return JDGPU_OK;
    if (!host_pixels) { snprintf(g_err, sizeof g_err, "%s: null buffer", what); return JDGPU_ERR_ARG; }
    uint8_t *d = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&d), npix * 3));
    hipError_t e = hipMemcpy(d, host_pixels, npix * 3, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)((npix + 1023) / 1024)), dim3(256), 0, 0, d, npix);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(host_pixels, d, npix * 3, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) { snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e)); return JDGPU_ERR_HIP; }
    return JDGPU_OK;
}

/* jdgpu_ycbcr_to_rgb without any allocation: the staging memory is the static scratch array */
extern "C" int jdgpu_ycbcr_to_rgb_noalloc(void *host_pixels, unsigned width, unsigned height) {
    int rc = ensure_device();
    if (rc) return rc;
    const size_t npix = (size_t)width * height;
    if (!npix) return JDGPU_OK;
    if (!host_pixels) { snprintf(g_err, sizeof g_err, "jdgpu_ycbcr_to_rgb_noalloc: null buffer"); return JDGPU_ERR_ARG; }
    if (npix * 3 > JD_SCRATCH_BYTES) {
        snprintf(g_err, sizeof g_err, "jdgpu_ycbcr_to_rgb_noalloc: %u x %u pixels need %zu bytes, the static scratch holds %u (%u pixels)",
                 width, height, npix * 3, JD_SCRATCH_BYTES, JD_SCRATCH_BYTES / 3u);
        return JDGPU_ERR_ARG;
    }
    if (!g_scratch) { snprintf(g_err, sizeof g_err, "jdgpu_ycbcr_to_rgb_noalloc: no scratch address (jdgpu_init failed?)"); return JDGPU_ERR_HIP; }
    /* the array is shared by every call of the process: one at a time.  Nothing has to start at
     * zero -- the kernel reads and writes exactly the 3 * npix bytes uploaded here. */
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    uint8_t *d = g_scratch;
    HIPCHK(hipMemcpy(d, host_pixels, npix * 3, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_ycbcr_to_rgb, dim3((unsigned)((npix + 1023) / 1024)), dim3(256), 0, 0, d, npix);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(host_pixels, d, npix * 3, hipMemcpyDeviceToHost));
    return JDGPU_OK;
}

extern "C" size_t jdgpu_noalloc_max_bytes(void) { return JD_SCRATCH_BYTES; }

extern "C" int jdgpu_ycbcr_to_rgb(void *host_pixels, unsigned width, unsigned height) {
    return convert_host_pixels(k_ycbcr_to_rgb, "jdgpu_ycbcr_to_rgb", host_pixels, width, height);
}

extern "C" int jdgpu_rgb_to_ycbcr(void *host_pixels, unsigned width, unsigned height) {
    return convert_host_pixels(k_rgb_to_ycbcr, "jdgpu_rgb_to_ycbcr", host_pixels, width, height);
}

#include "jdgpu_print.inc"   /* image_print's text formatted on the device (row f4) */
