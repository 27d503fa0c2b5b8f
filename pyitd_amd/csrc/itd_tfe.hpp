// itd_tfe.hpp — instantaneous amplitude, phase and frequency of a proper rotation.
//
// The reference describes this step but does not implement it (README.md:13-21, 41-55: "a sum of proper rotation
// components, for which instantaneous frequency and amplitude are well defined"); the definitions are those of the paper
// the README quotes (Frei & Osorio 2007, section on single-wave analysis): a proper rotation is cut into half waves at its
// zero crossings; with A the half wave's amplitude (its largest |x|),
//     amplitude(t) = A
//     phase(t)     = arcsin(x/A)          on the rising part of a positive half wave      [0, pi/2]
//                    pi - arcsin(x/A)     on the falling part of a positive half wave and the falling part of a negative one  [pi/2, 3pi/2]
//                    2 pi + arcsin(x/A)   on the rising part of a negative half wave     [3pi/2, 2pi]
//     frequency(t) = (phase(t+1) - phase(t)) mod 2 pi / (2 pi)      cycles per sample
// There is no upstream code, hence no parity target: tests check the definitions on signals with known answers.
//
// k_tfe_amplitude   half wave of every sample = number of zero crossings in front of it (the crossings come from the engine's
//                   ordered compaction, k_detect mode kZeroCross); max |x| per half wave by atomic max on the bit pattern of
//                   |x| (non-negative doubles order like unsigned integers), one atomic per wavefront where a wavefront lies
//                   inside one half wave;
// k_tfe_phase       amplitude, phase and frequency per sample.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace itd {

// zc: ordered indices i (m of them) with a sign change between x[i] and x[i+1]; sample j belongs to half wave
// hw(j) = number of crossings i with i < j.
__device__ __forceinline__ int64_t tfe_half_wave(const int32_t *__restrict__ zc, int64_t m, int64_t j)
{
    int64_t lo = 0, hi = m;      // first crossing with zc >= j
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)zc[mid] < j) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_tfe_amplitude(const double *__restrict__ x, int64_t n, const int32_t *__restrict__ zc,
                                                       int64_t m, unsigned long long *__restrict__ amp_bits /* m+1, zeroed */)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = j < n;
    const int64_t k = in ? tfe_half_wave(zc, m, j) : -1;
    const double a = in ? __builtin_fabs(x[j]) : 0.0;
    const unsigned long long bits = (a == a) ? __builtin_bit_cast(unsigned long long, a) : 0ull;   // NaN samples do not count
    const int64_t k0 = __shfl(k, 0);
    if (__all(k == k0)) {        // the whole wavefront inside one half wave: one atomic
        unsigned long long v = bits;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(v, d); v = o > v ? o : v; }
        if ((threadIdx.x & 63) == 0 && k0 >= 0) atomicMax(&amp_bits[k0], v);
    } else if (in) {
        atomicMax(&amp_bits[k], bits);
    }
}

__global__ __launch_bounds__(256) void k_tfe_phase(const double *__restrict__ x, int64_t n, const int32_t *__restrict__ zc, int64_t m,
                                                   const unsigned long long *__restrict__ amp_bits, double *__restrict__ amp_out,
                                                   double *__restrict__ phase_out, double *__restrict__ freq_out)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const double pi = 3.14159265358979323846;
    auto phase_of = [&](int64_t i, double *amp) {
        const int64_t k = tfe_half_wave(zc, m, i);
        const double A = __builtin_bit_cast(double, amp_bits[k]);
        const double xi = x[i];
        // rising or falling: the forward difference (the backward one at the last sample)
        const double slope = (i + 1 < n) ? x[i + 1] - xi : xi - x[i - 1];
        *amp = A;
        if (!(A > 0.0)) return 0.0;                     // an all-zero half wave
        const double r = xi / A;
        const double as = asin(r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r));
        if (xi >= 0.0) return slope >= 0.0 ? as : pi - as;
        return slope < 0.0 ? pi - as : 2.0 * pi + as;
    };
    double A, A1;
    const double ph = phase_of(j, &A);
    if (amp_out) amp_out[j] = A;
    if (phase_out) phase_out[j] = ph;
    if (freq_out) {
        double f = 0.0;
        if (j + 1 < n) {
            double dp = phase_of(j + 1, &A1) - ph;
            if (dp < 0.0) dp += 2.0 * pi;                // the phase wraps once per wave
            f = dp / (2.0 * pi);
        } else if (j >= 1) {
            double A0;
            double dp = ph - phase_of(j - 1, &A0);
            if (dp < 0.0) dp += 2.0 * pi;
            f = dp / (2.0 * pi);
        }
        freq_out[j] = f;
    }
}

}  // namespace itd
