// dispatch_bench.hip — how fast can the GPU launch 64-thread workgroups with k_extract_r's footprint?
// grid = 32768 one-wavefront workgroups (the 2^24-sample launch), kernels that do almost nothing but reserve
// LDS / VGPRs; time per launch = pure dispatch cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int LDS_BYTES, int NREG>
__global__ __launch_bounds__(64) void k_empty(double* out, int never)
{
    __shared__ int lds[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    double r[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = (double)(threadIdx.x + i);
    if (never) {   // keep the registers and the LDS alive without running any of it
#pragma unroll
        for (int i = 0; i < NREG; ++i) { lds[(threadIdx.x + i) % (LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1)] = (int)r[i]; }
        __syncthreads();
        double acc = 0;
#pragma unroll
        for (int i = 0; i < NREG; ++i) acc += r[i] * lds[(threadIdx.x * 7 + i) % (LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1)];
        out[blockIdx.x * 64 + threadIdx.x] = acc;
    }
}

int main()
{
    double* out; CK(hipMalloc(&out, 1 << 24));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        float best = 1e9, sum = 0; const int reps = 20, per = 8;
        for (int r = 0; r < reps + 2; ++r) {
            CK(hipEventRecord(e0));
            for (int j = 0; j < per; ++j) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= per;
            if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("%-48s per launch: best %6.1f us  avg %6.1f us\n", name, best * 1e3, sum / reps * 1e3);
    };
    const int grid = 32768;
    run("LDS 0     few VGPRs   grid 32768 x 64", [&] { k_empty<0, 1><<<grid, 64>>>(out, 0); });
    run("LDS 3.7KB few VGPRs   grid 32768 x 64", [&] { k_empty<3712, 1><<<grid, 64>>>(out, 0); });
    run("LDS 0     ~70 VGPRs   grid 32768 x 64", [&] { k_empty<0, 32><<<grid, 64>>>(out, 0); });
    run("LDS 3.7KB ~70 VGPRs   grid 32768 x 64", [&] { k_empty<3712, 32><<<grid, 64>>>(out, 0); });
    run("LDS 6.4KB ~40 VGPRs   grid 32768 x 64", [&] { k_empty<6400, 16><<<grid, 64>>>(out, 0); });
    run("LDS 3.7KB ~70 VGPRs   grid 16384 x 64", [&] { k_empty<3712, 32><<<grid / 2, 64>>>(out, 0); });
    run("LDS 3.7KB ~70 VGPRs   grid 8192 x 256", [&] { k_empty<3712, 32><<<grid / 4, 256>>>(out, 0); });
    run("LDS 3.7KB ~70 VGPRs   grid (32768,1) y-batch 4", [&] { k_empty<3712, 32><<<dim3(grid / 4, 4), 64>>>(out, 0); });
    return 0;
}
