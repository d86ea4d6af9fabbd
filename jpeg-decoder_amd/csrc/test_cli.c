/*
 * test_cli.c -- the demo viewer (config 1 of BASELINE.json): decode argv[1] and dump it to the
 * terminal as ANSI colour cells.  Same observable behaviour as the reference's src/test.c:9-48
 * (messages, exit codes, print geometry 200 columns / 2 cells per pixel), built against this
 * repo's include/jpeg.h + libjpegamd.so, so the decode runs on the MI355X.
 */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "image.h"
#include "jpeg.h"

int main(int argc, char **argv) {
    if (argc < 2) { puts("No path given."); return EXIT_FAILURE; }
    FILE *fp = fopen(argv[1], "rb");
    if (!fp) { printf("Could not open file: %s\n", strerror(errno)); return EXIT_FAILURE; }

    jpeg_info info;
    scan_info scan;
    int status = EXIT_FAILURE;
    if (jpeg_read_info(&info, &scan, fp)) {
        puts("Could not parse file.");
    } else {
        pixel *im = malloc(image_size(info.width, info.height));
        if (!im) {
            puts("Could not allocate memory for the image.");
        } else {
            jpeg_read_image(im, &info, &scan, fp);
            image_ycbcr_to_rgb(im, info.width, info.height);
            image_print(im, info.width, info.height, 200, info.height, 2);
            free(im);
            status = EXIT_SUCCESS;
        }
    }
    jpeg_free(&info);
    fclose(fp);
    return status;
}
