/* Plain C11 client of the two C ABIs (include/sdfgpu.h, include/sdfgpu_multi.h): what a cgo / JNI / ctypes binding
 * sees.  Builds a small occupancy grid through the single-GPU entry point and through the multi-GPU one (all ranks on
 * GPU 0 when the box has one GPU) and checks that both give the same field.  Without a GPU both create calls fail with
 * a message -- there is no CPU fallback -- and the program exits 0 after printing it (argument "--no-gpu").
 * Build: see tests/test_cpp_example.py (gcc -std=c11 ... -lsdfgpu_multi -lsdfgpu). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sdfgpu.h"
#include "sdfgpu_multi.h"

int main(int argc, char** argv) {
    const int expect_no_gpu = argc > 1 && strcmp(argv[1], "--no-gpu") == 0;
    const int64_t nx = 24, ny = 20, nz = 32, n = nx * ny * nz;
    uint8_t* filled = (uint8_t*)calloc((size_t)n, 1);
    float* a = (float*)malloc((size_t)n * sizeof(float));
    float* b = (float*)malloc((size_t)n * sizeof(float));
    if (!filled || !a || !b) return 2;
    for (int64_t x = 4; x < 9; ++x)
        for (int64_t y = 10; y < 14; ++y)
            for (int64_t z = 0; z < 12; ++z) filled[(x * ny + y) * nz + z] = 1;      /* one box */
    filled[(20 * ny + 3) * nz + 30] = 1;                                             /* and one voxel far from it */

    sdfgpu_handle h = NULL;
    int rc = sdfgpu_create(0, &h);
    if (rc != SDFGPU_OK) {
        printf("sdfgpu_create: %d (%s)\n", rc, sdfgpu_last_error(NULL));
        sdfgpu_multi_handle mh0 = NULL;
        rc = sdfgpu_multi_create(2, NULL, &mh0);
        printf("sdfgpu_multi_create: %d (%s)\n", rc, sdfgpu_multi_last_error(NULL));
        return expect_no_gpu && rc != SDFGPU_OK ? 0 : 1;
    }
    double mx = 0, mn = 0, mx2 = 0, mn2 = 0;
    rc = sdfgpu_build(h, filled, nx, ny, nz, 0.05, 1, a, &mx, &mn);
    if (rc != SDFGPU_OK) { printf("sdfgpu_build: %s\n", sdfgpu_last_error(h)); return 1; }

    const int ndev = sdfgpu_device_count();
    const int ranks = 3;
    int devices[3];
    for (int r = 0; r < ranks; ++r) devices[r] = ndev >= ranks ? r : 0;              /* one GPU: all ranks share it */
    sdfgpu_multi_handle mh = NULL;
    rc = sdfgpu_multi_create(ranks, devices, &mh);
    if (rc != SDFGPU_OK) { printf("sdfgpu_multi_create: %s\n", sdfgpu_multi_last_error(NULL)); return 1; }
    rc = sdfgpu_multi_build(mh, filled, nx, ny, nz, 0.05, 1, b, &mx2, &mn2);
    if (rc != SDFGPU_OK) { printf("sdfgpu_multi_build: %s\n", sdfgpu_multi_last_error(mh)); return 1; }
    int path = 0;
    sdfgpu_multi_last_path(mh, &path);
    const int same = memcmp(a, b, (size_t)n * sizeof(float)) == 0 && mx == mx2 && mn == mn2;
    printf("single GPU extrema (%.6f, %.6f); %d ranks extrema (%.6f, %.6f), path bits %d: %s\n", mx, mn, ranks, mx2, mn2, path,
           same ? "fields identical" : "MISMATCH");
    sdfgpu_multi_destroy(mh);
    sdfgpu_destroy(h);
    free(filled); free(a); free(b);
    return same ? 0 : 1;
}
