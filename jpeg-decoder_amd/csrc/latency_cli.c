/*
 * latency_cli.c -- development aid: wall time of the reference's single-image call sequence
 * (src/test.c:23-40: jpeg_read_info, jpeg_read_image, image_ycbcr_to_rgb) through this library,
 * per call, in one warm process.  usage: latency <file.jpg> [repeats]
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "image.h"
#include "jpeg.h"

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int main(int argc, char **argv) {
    if (argc < 2) { puts("usage: latency <file.jpg> [repeats]"); return 1; }
    int reps = argc > 2 ? atoi(argv[2]) : 20;
    for (int r = 0; r < reps; ++r) {
        FILE *fp = fopen(argv[1], "rb");
        if (!fp) { puts("Could not open file."); return 1; }
        jpeg_info ji;
        scan_info si;
        double t0 = now_ms();
        if (jpeg_read_info(&ji, &si, fp)) { puts("Could not parse file."); return 1; }
        double t1 = now_ms();
        pixel *im = malloc(image_size(ji.width, ji.height));
        jpeg_read_image(im, &ji, &si, fp);
        double t2 = now_ms();
        image_ycbcr_to_rgb(im, ji.width, ji.height);
        double t3 = now_ms();
        printf("call %2d  %ux%u  info %.3f ms  read_image %.3f ms  ycbcr_to_rgb %.3f ms  total %.3f ms\n", r,
               ji.width, ji.height, t1 - t0, t2 - t1, t3 - t2, t3 - t0);
        free(im);
        jpeg_free(&ji);
        fclose(fp);
    }
    return 0;
}
