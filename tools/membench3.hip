// membench3.hip — does the workgroup WIDTH matter for the extraction's three streams in the pipeline's cache state?  (tools/membench2's
// nt=5 kernel: nontemporal read + nontemporal rotation store + cacheable baseline store, 3 rotating baseline slots, random data;
// every thread moves 4 x 16 B per stream, a wavefront a 512-sample tile.)  k_extract runs one wavefront per workgroup (32768 workgroups
// per 2^24-sample level); this asks whether 2 / 4 / 8 wavefronts per workgroup would stream faster.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_level(const double* __restrict__ in, double* __restrict__ rot, double* __restrict__ bas)
{
    using V = double __attribute__((ext_vector_type(2)));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const size_t base = ((size_t)blockIdx.x * wpb + wave) * 512;
    const V* vi = reinterpret_cast<const V*>(in + base);
    V* v1 = reinterpret_cast<V*>(rot + base);
    V* v2 = reinterpret_cast<V*>(bas + base);
    V x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = __builtin_nontemporal_load(&vi[lane + 64 * k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        __builtin_nontemporal_store(x[k], &v1[lane + 64 * k]);
        v2[lane + 64 * k] = -x[k];
    }
}

int main()
{
    const size_t n = 1ull << 24;
    const int L = 8;
    double *rows, *bases;
    CK(hipMalloc(&rows, (L + 1) * n * 8)); CK(hipMalloc(&bases, 3 * n * 8));
    {
        double* hostr = (double*)malloc(n * 8);
        unsigned long long st64 = 88172645463325252ull;
        for (size_t i = 0; i < n; ++i) { st64 ^= st64 << 13; st64 ^= st64 >> 7; st64 ^= st64 << 17; hostr[i] = (double)(st64 >> 11) * (1.0 / 9007199254740992.0) + 1.0; }
        for (int s = 0; s < 3; ++s) CK(hipMemcpy(bases + (size_t)s * n, hostr, n * 8, hipMemcpyHostToDevice));
        free(hostr);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int round = 0; round < 2; ++round)
        for (int thr = 64; thr <= 512; thr *= 2) {
            float best = 1e9, sum = 0; const int reps = 10;
            for (int r = 0; r < reps + 2; ++r) {
                CK(hipEventRecord(e0));
                for (int j = 1; j <= L; ++j)
                    k_level<<<(int)(n / 512 / (thr / 64)), thr>>>(bases + (size_t)((j - 1) % 3) * n, rows + (size_t)j * n, bases + (size_t)(j % 3) * n);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= L;
                if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
            }
            printf("%d wavefront(s) per workgroup: per level best %6.1f us  avg %6.1f us  -> %5.0f GB/s (avg)\n", thr / 64, best * 1e3, sum / reps * 1e3, 24.0 * n / (sum / reps * 1e-3) / 1e9);
        }
    return 0;
}
