// Host build of pyitd_amd/csrc/itd_fitpack.hpp for tests/test_fitpack_host.py: the restatement of FITPACK's curfit is held to
// scipy.interpolate.splrep on this image before it ever runs on a GPU.  Test infrastructure: nothing in pyitd_amd/ uses this.
#include <stdlib.h>
#include <string.h>
#include "../../pyitd_amd/csrc/itd_fitpack.hpp"

extern "C" int fitpack_host_splrep(const double *x, const double *y, int m, double s, double *t_out, double *c_out, int *n_out,
                                   double *fp_out)
{
    double *buf = (double *)calloc((size_t)itd_fp::work_doubles(m) + 2 * (size_t)(m + 1), sizeof(double));
    if (!buf) return 100;
    double *x1 = buf, *y1 = buf + (m + 1);
    for (int i = 0; i < m; ++i) { x1[i + 1] = x[i]; y1[i + 1] = y[i]; }
    itd_fp::Work w = itd_fp::work_carve(buf + 2 * (m + 1), m);
    int n = 0;
    double fp = 0.0;
    const int ier = itd_fp::curfit(x1, y1, m, s, w, n, fp);
    for (int i = 0; i < n; ++i) { t_out[i] = w.t[i + 1]; c_out[i] = (i < n - 4) ? w.c[i + 1] : 0.0; }
    *n_out = n;
    *fp_out = fp;
    free(buf);
    return ier;
}

extern "C" void fitpack_host_splev(const double *t, const double *c, int n, int equi, double dx, int count, double *out)
{
    int l = itd_fp::K1, l1 = l + 1;
    for (int i = 0; i < count; ++i) out[i] = itd_fp::splev1(t, c, n, (double)i, equi != 0, dx, l, l1);
}

// the GPU form (implicit knots from the int32 data sites, strided working arrays) on the host, stride 1
extern "C" void fitpack_host_interp(const int32_t *e, const double *y, int m, double *c_out /* m */, int count, int equi, double dx,
                                    double *eval_out /* count */)
{
    double *a = (double *)calloc((size_t)5 * (m + 1), sizeof(double));
    double *z = a + (size_t)4 * (m + 1);
    itd_fp::interp_fit(e, m, [&](int k) { return y[k]; }, a, z, 1, m + 1);
    for (int i = 0; i < m; ++i) c_out[i] = z[i + 1];
    for (int i = 0; i < count; ++i) eval_out[i] = itd_fp::spline_eval(e, m, z, 1, (double)i, equi != 0, dx);
    free(a);
}
