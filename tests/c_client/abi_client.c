/* Plain-C client of the drop-in boundary (include/pyitd_hip.h): what a non-Python host would link.
 * usage: abi_client <n> <max_iteration>  — reads n float64 samples from stdin (raw), decomposes them on GPU 0 through
 * itd_decompose_host_f64, writes "<n_rows> <stop>\n" then the rows (raw float64) to stdout.  tests/test_abi_c_client.py
 * builds it with gcc against libpyitd_hip.so and compares the rows with the oracle bit for bit. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "pyitd_hip.h"

int main(int argc, char **argv)
{
    if (argc != 3) return 2;
    const int64_t n = atoll(argv[1]);
    const int32_t m = atoi(argv[2]);
    if (itd_abi_version() != ITD_ABI_VERSION) return 3;
    double *x = (double *)malloc((size_t)n * sizeof(double));
    double *rows = (double *)malloc((size_t)(m + 2) * n * sizeof(double));
    int64_t knots[ITD_MAX_ROWS + 1];
    if (!x || !rows || fread(x, sizeof(double), (size_t)n, stdin) != (size_t)n) return 4;
    itd_engine *e = NULL;
    int rc = itd_engine_create(&e, 0, n, 1);
    if (rc != ITD_OK) { fprintf(stderr, "create: %s\n", itd_status_string(rc)); return 5; }
    int32_t n_rows = 0, n_base = 0, stop = 0;
    rc = itd_decompose_host_f64(e, x, n, m, rows, NULL, &n_rows, &n_base, &stop, knots);
    if (rc != ITD_OK) { fprintf(stderr, "decompose: %s (%s)\n", itd_status_string(rc), itd_last_error(e)); return 6; }
    /* bad arguments come back as status codes, never as a crash */
    if (itd_decompose_host_f64(e, x, 2, m, rows, NULL, &n_rows, &n_base, &stop, knots) != ITD_ERR_INVALID_ARG) return 7;
    if (itd_decompose_host_f64(e, x, n, m, rows, NULL, &n_rows, &n_base, &stop, knots) != ITD_OK) return 8;
    printf("%d %d\n", (int)n_rows, (int)stop);
    fwrite(rows, sizeof(double), (size_t)n_rows * (size_t)n, stdout);
    itd_engine_destroy(e);
    free(x);
    free(rows);
    return 0;
}
