// anyorder_probe.hip — can ONE queue run two launches side by side on gfx950?
// hipExtLaunchKernel(..., flags = hipExtAnyOrderLaunch) leaves the dispatch packet's barrier bit clear: the packet processor may start
// the launch as soon as every workgroup of the launch in front has been DISPATCHED (not finished).  A latency-bound launch of a few
// large workgroups (the fused levels' knot side: 512 threads, 65 KB LDS, spinning on its neighbours) followed by a memory-bound launch
// of one-wavefront workgroups (the sample pass) would then share the device without a second queue — on two queues the large
// workgroups never find room beside the small ones (profiles/r04/experiments/README.md).
//   case A   spin ; stream            in order                        expected: t_spin + t_stream
//   case B   spin ; stream            stream launched "any order"     overlap: ~max(t_spin, t_stream)
//   case C   spin on queue 1, stream on queue 2, spin first
//   case D   stream ; spin            spin launched "any order" behind the memory-bound launch (the large workgroups come last)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(512) void k_spin(long long ticks, int *sink, unsigned long long *started)
{
    __shared__ int lds[65000 / 4];
    if (started && threadIdx.x == 0) atomicAdd(started, 1ull);
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x * 7) & 511] == -1) sink[0] = 1;
}

// the gate: one wavefront that returns when `started` has reached `target` (every workgroup of the latency-bound launch on the other queue is
// resident) or after `timeout` ticks of the 100 MHz clock
__global__ __launch_bounds__(64) void k_gate(const unsigned long long *started, unsigned long long target, long long timeout)
{
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && wall_clock64() - t0 < timeout) __builtin_amdgcn_s_sleep(4);
}
__global__ __launch_bounds__(64) void k_delay(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}

// one wavefront per 512 doubles: read 8 B, write `rows` x 8 B per sample (the sample pass's shape), 3.6 KB LDS per workgroup
__global__ __launch_bounds__(64) void k_stream(const double *__restrict__ x, double *__restrict__ out, long long n, int rows)
{
    __shared__ double lds[450];
    const long long i = (long long)blockIdx.x * 512 + threadIdx.x * 2;
    lds[threadIdx.x] = 0.0;
    double a[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) { a[g][0] = x[i + g * 128]; a[g][1] = x[i + g * 128 + 1]; }
    for (int r = 0; r < rows; ++r) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            a[g][0] = a[g][0] * 0.5 + lds[threadIdx.x]; a[g][1] = a[g][1] * 0.5;
            __builtin_nontemporal_store(a[g][0], out + (long long)r * n + i + g * 128);
            __builtin_nontemporal_store(a[g][1], out + (long long)r * n + i + g * 128 + 1);
        }
    }
}

int main()
{
    const long long n = 1ll << 23;
    const int rows = 7;
    double *x, *out; int *sink;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&out, n * 8 * rows)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(x, 0, n * 8));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    long long ticks = 6000;   // 60 us of the 100 MHz clock
    int spin_wgs = 256;
    unsigned long long *started; CK(hipMalloc(&started, 8)); CK(hipMemset(started, 0, 8));
    unsigned long long launched = 0; bool count = false;
    hipStream_t s3; { int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi)); CK(hipStreamCreateWithPriority(&s3, hipStreamNonBlocking, hi)); }
    hipEvent_t ea, ek; CK(hipEventCreateWithFlags(&ea, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ek, hipEventDisableTiming));
    auto spin = [&](hipStream_t s, int flags) {
        unsigned long long *a_s = count ? started : nullptr;
        if (count) launched += (unsigned long long)spin_wgs;
        void *args[] = {&ticks, &sink, &a_s};
        CK(hipExtLaunchKernel(reinterpret_cast<const void *>(&k_spin), dim3(spin_wgs), dim3(512), args, 0, s, nullptr, nullptr, flags));
    };
    auto stream = [&](hipStream_t s, int flags) {
        const double *a_x = x; double *a_o = out; long long a_n = n; int a_r = rows;
        void *args[] = {&a_x, &a_o, &a_n, &a_r};
        CK(hipExtLaunchKernel(reinterpret_cast<const void *>(&k_stream), dim3((unsigned)(n / 512)), dim3(64), args, 0, s, nullptr, nullptr, flags));
    };
    auto run = [&](const char *name, auto body) {
        float best = 1e9f, sum = 0.f; const int reps = 20, per = 10;
        for (int r = 0; r < reps + 3; ++r) {
            CK(hipEventRecord(e0, s1));
            for (int j = 0; j < per; ++j) body();
            CK(hipEventRecord(e1, s1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= per;
            if (r >= 3) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("%-72s best %7.1f us  avg %7.1f us\n", name, best * 1e3, sum / reps * 1e3);
    };
    for (int wgs : {256, 512}) {
        spin_wgs = wgs;
        printf("---- spin launch: %d workgroups x 512 threads x 65 KB LDS, 60 us; stream launch: 2^23 samples, 8 B in + %d x 8 B out ----\n", wgs, rows);
        run("spin alone", [&] { spin(s1, 0); });
        run("stream alone", [&] { stream(s1, 0); });
        run("A  spin ; stream (in order)", [&] { spin(s1, 0); stream(s1, 0); });
        run("B  spin ; stream (any order)", [&] { spin(s1, 0); stream(s1, hipExtAnyOrderLaunch); });
        run("C  spin on queue 1 | stream on queue 2", [&] {
            CK(hipEventRecord(e2, s1)); CK(hipStreamWaitEvent(s2, e2, 0));
            spin(s1, 0); stream(s2, 0);
            CK(hipEventRecord(e2, s2)); CK(hipStreamWaitEvent(s1, e2, 0));
        });
        run("D  stream ; spin (any order)", [&] { stream(s1, 0); spin(s1, hipExtAnyOrderLaunch); });
        run("E  spin ; stream (any order) ; stream (any order)", [&] { spin(s1, 0); stream(s1, hipExtAnyOrderLaunch); stream(s1, hipExtAnyOrderLaunch); });
        // the pipeline's shape: queue 1 runs memory-bound launches back to back; the latency-bound launch on queue 2 becomes ready when the
        // first of them ends (an event), i.e. at the same moment as queue 1's next launch
        for (hipStream_t q2 : {s2, s3}) {
            const char *qn = q2 == s2 ? "" : " (queue 2 of high priority)";
            char name[160];
            snprintf(name, sizeof name, "F  q1: stream E stream | q2: wait(E) spin%s   [sum of parts 226]", qn);
            run(name, [&] {
                stream(s1, 0); CK(hipEventRecord(ea, s1)); CK(hipStreamWaitEvent(q2, ea, 0));
                spin(q2, 0); stream(s1, 0);
                CK(hipEventRecord(ek, q2)); CK(hipStreamWaitEvent(s1, ek, 0));
            });
            for (long long d : {200ll, 400ll, 800ll}) {
                snprintf(name, sizeof name, "G  q1: stream E delay(%lld us) stream | q2: wait(E) spin%s", d / 100, qn);
                run(name, [&] {
                    stream(s1, 0); CK(hipEventRecord(ea, s1)); CK(hipStreamWaitEvent(q2, ea, 0));
                    spin(q2, 0);
                    k_delay<<<1, 64, 0, s1>>>(d); stream(s1, 0);
                    CK(hipEventRecord(ek, q2)); CK(hipStreamWaitEvent(s1, ek, 0));
                });
            }
            count = true;
            snprintf(name, sizeof name, "H  q1: stream E gate(all started) stream | q2: wait(E) spin%s", qn);
            run(name, [&] {
                stream(s1, 0); CK(hipEventRecord(ea, s1)); CK(hipStreamWaitEvent(q2, ea, 0));
                spin(q2, 0);
                k_gate<<<1, 64, 0, s1>>>(started, launched, 3000ll); stream(s1, 0);
                CK(hipEventRecord(ek, q2)); CK(hipStreamWaitEvent(s1, ek, 0));
            });
            count = false;
        }
    }
    return 0;
}
