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
// One wavefront per 512-sample tile, 64 samples per step.  The half wave of a sample = number of zero crossings in front of
// it = (crossings in front of the tile: k_compact's per-tile base of the engine's ordered compaction, k_detect mode kZeroCross)
// + (crossings of the tile in front of the sample: popcounts of the sign-change flags, evaluated here with the same predicate).
// k_tfe_amplitude   max |x| per half wave by atomic max on the bit pattern of |x| (non-negative doubles order like unsigned
//                   integers): a segmented max over the lanes of a step, one atomic per (step, half wave);
// k_tfe_phase       amplitude, phase and frequency per sample: one arcsine per sample, the neighbour's phase from the next lane.
// (The first version searched the crossing list per sample — three binary searches and up to three arcsines each — and issued
// one atomic per sample: 0.51 + 0.62 ms for 2^24 samples.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace itd {

constexpr int kTfeTile = 512;

// crossing index i: a sign change between x[i] and x[i+1] (find_extrema's test, itd_fourier_decomposition.py:23-27), 1 <= i <= n-2
// exactly as k_detect's kZeroCross mode flags it (first and last sample never flag) — the per-tile bases come from that scan
__device__ __forceinline__ bool tfe_crossing(int64_t i, int64_t n, double x0, double xp)
{
    return i >= 1 && i <= n - 2 && (((x0 > 0.0) && (0.0 > xp)) || ((x0 < 0.0) && (0.0 < xp)));
}

__global__ __launch_bounds__(64) void k_tfe_amplitude(const double *__restrict__ x, int64_t n, const int32_t *__restrict__ tile_base,
                                                      unsigned long long *__restrict__ amp_bits /* m+1, zeroed */)
{
    const int lane = threadIdx.x;
    const int64_t s = (int64_t)blockIdx.x * kTfeTile;
    int64_t hw0 = tile_base[blockIdx.x];            // half wave of the step's first sample
    for (int g = 0; g < kTfeTile / 64; ++g) {
        const int64_t j = s + g * 64 + lane;
        if (s + g * 64 >= n) break;
        const double xj = j < n ? x[j] : 0.0, xn = j + 1 < n ? x[j + 1] : 0.0;
        const bool cross = tfe_crossing(j, n, xj, xn);                   // crossing index j: samples j+1.. belong to the next half wave
        const unsigned long long cm = __ballot(cross);
        const int before = __popcll(cm & ((1ull << lane) - 1ull));      // crossings i < j inside the step
        const double a = j < n ? __builtin_fabs(xj) : 0.0;
        unsigned long long v = (a == a) ? __builtin_bit_cast(unsigned long long, a) : 0ull;   // NaN samples do not count
        // segmented max over runs of equal `before` (non-decreasing along the lanes): after the scan the last lane of a run holds it
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = __shfl_up(v, d);
            const int ob = __shfl_up(before, d);
            if (lane >= d && ob == before && o > v) v = o;
        }
        const int nxt = __shfl_down(before, 1);
        const bool last = lane == 63 || nxt != before || j + 1 >= n;
        if (last && j < n) atomicMax(&amp_bits[hw0 + before], v);
        hw0 += __popcll(cm);
    }
}

__global__ __launch_bounds__(64) void k_tfe_phase(const double *__restrict__ x, int64_t n, const int32_t *__restrict__ tile_base,
                                                  const unsigned long long *__restrict__ amp_bits, double *__restrict__ amp_out,
                                                  double *__restrict__ phase_out, double *__restrict__ freq_out)
{
    const double pi = 3.14159265358979323846;
    const int lane = threadIdx.x;
    const int64_t s = (int64_t)blockIdx.x * kTfeTile;
    int64_t hw0 = tile_base[blockIdx.x];
    if (s >= 64 && s + kTfeTile + 2 <= n) {
        // An inner tile (every sample has its two successors, none is the signal's first): everything is requested before anything
        // is used — the tile, the half waves' amplitudes of all eight 64-sample steps — and a step's last lane takes the next sample's
        // phase from the next step's first lane instead of a second arcsine.  (The generic loop below waits for its loads step by
        // step, two dependent round trips each: 171 us at 2^24 samples against 537 MB of traffic.)
        constexpr int G = kTfeTile / 64;
        double xr[G];
#pragma unroll
        for (int g = 0; g < G; ++g) xr[g] = x[s + g * 64 + lane];
        const double ext = lane < 2 ? x[s + kTfeTile + lane] : 0.0;
        const double ext0 = __shfl(ext, 0), ext1 = __shfl(ext, 1);
        double xn[G];
        unsigned long long cm[G];
        int before[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const double nxt = __shfl_down(xr[g], 1);
            const double first_next = g + 1 < G ? __shfl(xr[g + 1 < G ? g + 1 : g], 0) : ext0;
            xn[g] = lane == 63 ? first_next : nxt;
            const bool cross = ((xr[g] > 0.0) && (0.0 > xn[g])) || ((xr[g] < 0.0) && (0.0 < xn[g]));   // (1 <= j <= n-2 holds here)
            cm[g] = __ballot(cross);
            before[g] = __popcll(cm[g] & ((1ull << lane) - 1ull));
        }
        double A[G];
        int64_t hw = hw0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            A[g] = __builtin_bit_cast(double, amp_bits[hw + before[g]]);
            hw += __popcll(cm[g]);
        }
        const double A_ext = __builtin_bit_cast(double, amp_bits[hw]);       // the half wave of sample s + 512
        auto phase_in = [&](double xi, double slope, double Aa) {
            if (!(Aa > 0.0)) return 0.0;                    // an all-zero half wave
            const double r = xi / Aa;
            const double as = asin(r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r));
            if (xi >= 0.0) return slope >= 0.0 ? as : pi - as;
            return slope < 0.0 ? pi - as : 2.0 * pi + as;
        };
        double ph[G];
#pragma unroll
        for (int g = 0; g < G; ++g) ph[g] = phase_in(xr[g], xn[g] - xr[g], A[g]);
        const double ph_ext = phase_in(ext0, ext1 - ext0, A_ext);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int64_t j = s + g * 64 + lane;
            if (amp_out) __builtin_nontemporal_store(A[g], &amp_out[j]);
            if (phase_out) __builtin_nontemporal_store(ph[g], &phase_out[j]);
            if (freq_out) {
                const double nxt = __shfl_down(ph[g], 1);
                const double first_next = g + 1 < G ? __shfl(ph[g + 1 < G ? g + 1 : g], 0) : ph_ext;
                double dp = (lane == 63 ? first_next : nxt) - ph[g];
                if (dp < 0.0) dp += 2.0 * pi;            // the phase wraps once per wave
                __builtin_nontemporal_store(dp / (2.0 * pi), &freq_out[j]);
            }
        }
        return;
    }
    // phase of sample i in half wave k (amplitude A): rising or falling from the forward difference (the backward one at the last sample)
    auto phase_of = [&](int64_t i, double xi, double slope, double A) {
        if (!(A > 0.0)) return 0.0;                     // an all-zero half wave
        const double r = xi / A;
        const double as = asin(r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r));
        if (xi >= 0.0) return slope >= 0.0 ? as : pi - as;
        return slope < 0.0 ? pi - as : 2.0 * pi + as;
    };
    for (int g = 0; g < kTfeTile / 64; ++g) {
        const int64_t j0 = s + g * 64;
        if (j0 >= n) break;
        // lanes 0..63: samples j0 .. j0+63; the step also needs sample j0+64's phase (lane 63's neighbour): computed by lane 0 below
        const int64_t j = j0 + lane;
        const bool in = j < n;
        const double xj = in ? x[j] : 0.0, xn = j + 1 < n ? x[j + 1] : 0.0, xnn = j + 2 < n ? x[j + 2] : 0.0;
        const bool cross = tfe_crossing(j, n, xj, xn);
        const unsigned long long cm = __ballot(cross);
        const int before = __popcll(cm & ((1ull << lane) - 1ull));
        const int64_t k = hw0 + before;
        const double A = in ? __builtin_bit_cast(double, amp_bits[k]) : 0.0;
        const double xprev = (in && j >= 1 && j + 1 >= n) ? x[j - 1] : 0.0;        // only the signal's last sample needs it
        const double slope = (j + 1 < n) ? xn - xj : xj - xprev;
        const double ph = in ? phase_of(j, xj, slope, A) : 0.0;
        if (in && amp_out) amp_out[j] = A;
        if (in && phase_out) phase_out[j] = ph;
        if (freq_out) {
            // the next sample's phase: the next lane's, except for lane 63 (sample j0+64, in half wave k + cross)
            double ph_next = __shfl_down(ph, 1);
            const double ph_prev = __shfl_up(ph, 1);     // (shuffles outside the divergent branches below)
            if (lane == 63 && j + 1 < n) {
                const int64_t k1 = k + (cross ? 1 : 0);
                const double A1 = __builtin_bit_cast(double, amp_bits[k1]);
                const double slope1 = (j + 2 < n) ? xnn - xn : xn - xj;
                ph_next = phase_of(j + 1, xn, slope1, A1);
            }
            double f = 0.0;
            if (j + 1 < n) {
                double dp = ph_next - ph;
                if (dp < 0.0) dp += 2.0 * pi;            // the phase wraps once per wave
                f = dp / (2.0 * pi);
            } else if (in && j >= 1) {                   // the last sample: backward difference of the phases
                double dp = ph - ph_prev;
                if (lane == 0) {                         // its predecessor sits in the previous step
                    const int64_t kp = k - (tfe_crossing(j - 1, n, xprev, xj) ? 1 : 0);
                    const double Ap = __builtin_bit_cast(double, amp_bits[kp]);
                    dp = ph - phase_of(j - 1, xprev, xj - xprev, Ap);
                }
                if (dp < 0.0) dp += 2.0 * pi;
                f = dp / (2.0 * pi);
            }
            if (in) freq_out[j] = f;
        }
        hw0 += __popcll(cm);
    }
}

}  // namespace itd
