/* Per-launch kernel timing of the engine from a plain C process (no Python, no torch): decomposes a synthetic 2^24-sample
 * float32 signal through the host API a few times and prints the average duration of the extraction launches.
 * build: gcc -O1 -I include tools/c_timing.c -o tools/c_timing -L pyitd_amd -lpyitd_hip -lm -Wl,-rpath,$PWD/pyitd_amd
 * (or point LD_LIBRARY_PATH at a directory holding a diagnostic build named libpyitd_hip.so) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "pyitd_hip.h"

int main(int argc, char **argv)
{
    const int64_t n = 1ll << 24;
    const int32_t m = 7, reps = argc > 1 ? atoi(argv[1]) : 4;
    float *x = (float *)malloc((size_t)n * sizeof(float));
    double *rows = (double *)malloc((size_t)(m + 2) * n * sizeof(double));
    unsigned s = 12345u;
    for (int64_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const double t = (double)i / 48000.0;
        x[i] = (float)(sin(6.283185307179586 * 110 * t) + 0.5 * sin(6.283185307179586 * 440 * t + 1.3) + 0.05 * ((double)(s >> 8) / 8388608.0 - 1.0));
    }
    itd_engine *e = NULL;
    if (itd_engine_create(&e, 0, n, 1) != ITD_OK) return 1;
    int32_t n_rows, n_base, stop;
    int64_t knots[ITD_MAX_ROWS + 1];
    if (itd_decompose_host_f32(e, x, n, m, rows, NULL, &n_rows, &n_base, &stop, knots) != ITD_OK) return 2;
    itd_set_kernel_timing(e, reps);
    itd_set_kernel_timing_stride(e, 1);
    for (int r = 0; r < reps; ++r)
        if (itd_decompose_host_f32(e, x, n, m, rows, NULL, &n_rows, &n_base, &stop, knots) != ITD_OK) return 3;
    const char *names[3] = {"levels>=1", "level 0", "FINAL"};
    for (int w = 0; w < 3; ++w) {
        double ms = 0; int32_t c = 0;
        itd_get_kernel_timing(e, w, &ms, &c);
        printf("%s: %.1f us over %d launches\n", names[w], c ? ms / c * 1e3 : 0.0, (int)c);
    }
    printf("rows %d stop %d\n", (int)n_rows, (int)stop);
    itd_engine_destroy(e);
    return 0;
}
