// cumask_probe.hip — can a latency-bound launch get compute units of its own next to memory-bound launches of another stream?
// hipExtStreamCreateWithCUMask: (1) which XCDs / CUs a masked stream's workgroups land on (HW_REG_XCC_ID, HW_REG_HW_ID), for a few
// mask layouts; (2) the three-stream level kernel of membench3 (24 B per sample) on all CUs and on the masks' complements;
// (3) a stand-in for the knot side (256-thread workgroups, 64 KB of LDS, a 50 us dependent spin) on a masked stream WHILE the level
// kernel streams on another: alone, side by side unmasked, side by side with disjoint masks.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/cumask_probe tools/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void k_where(unsigned *out)
{
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(8);      // stay resident so that the grid spreads out
}

__global__ void k_level(const double* __restrict__ in, double* __restrict__ rot, double* __restrict__ bas)
{
    using V = double __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const size_t base = (size_t)blockIdx.x * 512;
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

// the knot side's shape: 256 threads, 64 KB of LDS, ~50 us of dependent waiting
__global__ __launch_bounds__(256) void k_latency(unsigned *sink, int us)
{
    __shared__ double big[8192];
    big[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)us * 100) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) sink[blockIdx.x] = (unsigned)big[(blockIdx.x * 7) & 8191];
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("%s: %d CUs\n", prop.name, prop.multiProcessorCount);
    const int words = 8;                                              // 256 bits
    std::vector<std::vector<uint32_t>> masks;
    std::vector<const char *> names;
    { std::vector<uint32_t> m(words, 0); m[0] = m[1] = 0xffffffffu; masks.push_back(m); names.push_back("bits 0..63"); }
    { std::vector<uint32_t> m(words, 0); for (int w = 0; w < words; ++w) m[w] = 0x000000ffu; masks.push_back(m); names.push_back("low 8 bits of every 32"); }
    { std::vector<uint32_t> m(words, 0); for (int b = 0; b < 256; b += 4) m[b / 32] |= 1u << (b % 32); masks.push_back(m); names.push_back("every 4th bit"); }
    unsigned *d_out; CK(hipMalloc(&d_out, 2 * 4096 * sizeof(unsigned)));
    std::vector<unsigned> h(2 * 4096);
    std::vector<hipStream_t> ms(masks.size()), cs(masks.size());
    for (size_t k = 0; k < masks.size(); ++k) {
        hipError_t rc = hipExtStreamCreateWithCUMask(&ms[k], words, masks[k].data());
        if (rc != hipSuccess) { printf("hipExtStreamCreateWithCUMask(%s): %s\n", names[k], hipGetErrorString(rc)); return 0; }
        std::vector<uint32_t> c(words);
        for (int w = 0; w < words; ++w) c[w] = ~masks[k][w];
        CK(hipExtStreamCreateWithCUMask(&cs[k], words, c.data()));
        k_where<<<1024, 64, 0, ms[k]>>>(d_out);
        CK(hipStreamSynchronize(ms[k]));
        CK(hipMemcpy(h.data(), d_out, 2 * 1024 * sizeof(unsigned), hipMemcpyDeviceToHost));
        int per_xcc[8] = {0}; std::vector<int> seen(8 * 4096, 0); int distinct = 0;
        for (int b = 0; b < 1024; ++b) {
            const unsigned xcc = h[2 * b] & 7, hw = h[2 * b + 1];
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;     // GFX9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
            ++per_xcc[xcc];
            const int key = (int)(xcc * 256 + se * 32 + sh * 16 + cu);
            if (!seen[key]) { seen[key] = 1; ++distinct; }
        }
        printf("mask '%s': workgroups per XCC [%d %d %d %d %d %d %d %d], distinct (XCC, SE, SH, CU) = %d\n", names[k], per_xcc[0], per_xcc[1],
               per_xcc[2], per_xcc[3], per_xcc[4], per_xcc[5], per_xcc[6], per_xcc[7], distinct);
    }
    // bandwidth of the level kernel on all CUs and on the complements
    const size_t n = 1ull << 24;
    double *rows, *bases; CK(hipMalloc(&rows, 2 * n * 8)); CK(hipMalloc(&bases, 3 * n * 8));
    CK(hipMemset(bases, 0, 3 * n * 8));
    unsigned *sink; CK(hipMalloc(&sink, 4096 * sizeof(unsigned)));
    hipStream_t all; CK(hipStreamCreateWithFlags(&all, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_level = [&](hipStream_t s, const char *what) {
        float sum = 0;
        for (int r = 0; r < 12; ++r) {
            CK(hipEventRecord(e0, s));
            for (int j = 0; j < 4; ++j) k_level<<<(int)(n / 512), 64, 0, s>>>(bases + (size_t)(j % 3) * n, rows + (size_t)(j & 1) * n, bases + (size_t)((j + 1) % 3) * n);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (r >= 2) sum += t / 4;
        }
        printf("level kernel on %-34s %6.1f us per launch = %5.0f GB/s\n", what, sum / 10 * 1e3, 24.0 * n / (sum / 10 * 1e-3) / 1e9);
    };
    time_level(all, "all CUs");
    for (size_t k = 0; k < masks.size(); ++k) { char b[96]; snprintf(b, sizeof b, "the complement of '%s'", names[k]); time_level(cs[k], b); }
    // the latency-bound stand-in beside the level kernel
    auto side_by_side = [&](hipStream_t lat, hipStream_t bw, const char *what) {
        float sum = 0;
        for (int r = 0; r < 8; ++r) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, bw));
            for (int j = 0; j < 4; ++j) k_level<<<(int)(n / 512), 64, 0, bw>>>(bases + (size_t)(j % 3) * n, rows + (size_t)(j & 1) * n, bases + (size_t)((j + 1) % 3) * n);
            for (int j = 0; j < 4; ++j) k_latency<<<128, 256, 0, lat>>>(sink, 50);
            CK(hipEventRecord(e1, bw));
            CK(hipStreamSynchronize(lat)); CK(hipEventSynchronize(e1));
            hipEvent_t a, b2; CK(hipEventCreate(&a)); CK(hipEventCreate(&b2));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            // wall time until both are done
            CK(hipEventDestroy(a)); CK(hipEventDestroy(b2));
            if (r >= 2) sum += t;
        }
        printf("4 level launches (bandwidth stream's own time) beside 4 x 50 us latency launches, %-28s %7.1f us\n", what, sum / 6 * 1e3);
    };
    auto wall = [&](hipStream_t lat, hipStream_t bw, const char *what) {
        double sum = 0;
        for (int r = 0; r < 8; ++r) {
            CK(hipDeviceSynchronize());
            hipEvent_t s0, s1, s2; CK(hipEventCreate(&s0)); CK(hipEventCreate(&s1)); CK(hipEventCreate(&s2));
            CK(hipEventRecord(s0, bw)); CK(hipStreamWaitEvent(lat, s0, 0));
            for (int j = 0; j < 4; ++j) k_level<<<(int)(n / 512), 64, 0, bw>>>(bases + (size_t)(j % 3) * n, rows + (size_t)(j & 1) * n, bases + (size_t)((j + 1) % 3) * n);
            for (int j = 0; j < 4; ++j) k_latency<<<128, 256, 0, lat>>>(sink, 50);
            CK(hipEventRecord(s1, lat)); CK(hipStreamWaitEvent(bw, s1, 0));
            CK(hipEventRecord(s2, bw)); CK(hipEventSynchronize(s2));
            float t; CK(hipEventElapsedTime(&t, s0, s2));
            if (r >= 2) sum += t;
            CK(hipEventDestroy(s0)); CK(hipEventDestroy(s1)); CK(hipEventDestroy(s2));
        }
        printf("wall time of both (4 level launches + 4 x 50 us latency launches), %-30s %7.1f us\n", what, sum / 6 * 1e3);
    };
    hipStream_t lat_all; CK(hipStreamCreateWithFlags(&lat_all, hipStreamNonBlocking));
    side_by_side(lat_all, all, "no masks:");
    wall(lat_all, all, "no masks:");
    for (size_t k = 0; k < masks.size(); ++k) {
        char b[96]; snprintf(b, sizeof b, "'%s' / its complement:", names[k]);
        side_by_side(ms[k], cs[k], b);
        wall(ms[k], cs[k], b);
    }
    return 0;
}
