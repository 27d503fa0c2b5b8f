// membench4.hip — what the memory system gives for the fused sample pass's traffic shape: one wavefront per 512-sample tile reads
// 8 B per sample (one row, nontemporal) and writes R rows of 8 B per sample (nontemporal, rows 2^24 x 8 B apart), nothing else.
// usage: hipcc --offload-arch=gfx950 -O3 -o tools/membench4 tools/membench4.hip && tools/membench4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int R, bool STAGED>
__global__ __launch_bounds__(64) void k_rows(const double *__restrict__ in, double *__restrict__ out, long n, long pitch)
{
    const int t = blockIdx.x, lane = threadIdx.x;
    using D2 = double __attribute__((ext_vector_type(2)));
    const D2 *src = reinterpret_cast<const D2 *>(in + (long)t * 512);
    D2 v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) v[g] = __builtin_nontemporal_load(&src[g * 64 + lane]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        D2 *dst = reinterpret_cast<D2 *>(out + (long)r * pitch + (long)t * 512);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            v[g] = v[g] * 0.999 + 1.0;
            __builtin_nontemporal_store(v[g], &dst[g * 64 + lane]);
        }
        if (STAGED) __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0): a level waits for the previous level's stores (what the table loads do)
    }
}

int main()
{
    const long n = 1l << 24;
    double *in, *out;
    CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&out, (n + 65536) * 8 * 8));
    CK(hipMemset(in, 0, n * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, int rows, auto launch) {
        float best = 1e9, sum = 0; const int reps = 20;
        for (int r = 0; r < reps + 3; ++r) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 3) { best = ms < best ? ms : best; sum += ms; }
        }
        const double bytes = (8.0 + 8.0 * rows) * n;
        printf("%-44s best %6.1f us  avg %6.1f us   %5.2f TB/s (avg), %.3f of 8 TB/s\n", name, best * 1e3, sum / reps * 1e3, bytes / (sum / reps * 1e-3) / 1e12, bytes / (sum / reps * 1e-3) / 8e12);
    };
    for (long pad : {0l, 512l, 4096l, 8192l + 512l, 65536l - 512l}) {
        char nm[96];
        snprintf(nm, sizeof nm, "1 read + 7 write rows, row pitch n + %ld", pad);
        run(nm, 7, [&] { k_rows<7, false><<<32768, 64>>>(in, out, n, n + pad); });
        snprintf(nm, sizeof nm, "   ... wait per row, row pitch n + %ld", pad);
        run(nm, 7, [&] { k_rows<7, true><<<32768, 64>>>(in, out, n, n + pad); });
    }
    run("1 read + 2 write rows", 2, [&] { k_rows<2, false><<<32768, 64>>>(in, out, n, n); });
    run("1 read + 2 write rows, pitch n + 512", 2, [&] { k_rows<2, false><<<32768, 64>>>(in, out, n, n + 512); });
    run("1 read + 1 write row", 1, [&] { k_rows<1, false><<<32768, 64>>>(in, out, n, n); });
    return 0;
}
