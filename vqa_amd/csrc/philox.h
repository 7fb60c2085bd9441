// Counter-based random numbers for the device-side shot statistics: Philox-4x32-10, uniform and
// standard normal deviates, and a binomial sampler (inversion for small n p, BTRS rejection
// otherwise).  Shared by the shot sampler (kern_sample.hip) and the estimator kernels
// (kern_stats.hip).  A stream is named by (seed, stream id, row): results do not depend on the
// launch shape.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vqa {

struct Philox {
    uint32_t key0, key1;
    uint32_t c0, c1, c2, c3;
    uint32_t draws;   // how many blocks this row has used
    uint32_t out[4];
    int have;         // unused 32-bit words left in out

    __device__ void init(uint64_t seed, uint64_t stream, uint64_t row) {
        // counter = (row low, row high ^ stream high, draw number, stream low): a call has fewer than
        // 2^32 rows, so the row's high word is free to carry the stream's, and two calls with
        // different stream ids never share a counter
        key0 = (uint32_t)seed;
        key1 = (uint32_t)(seed >> 32);
        c0 = (uint32_t)row;
        c1 = (uint32_t)(row >> 32) ^ (uint32_t)(stream >> 32);
        c3 = (uint32_t)stream;
        draws = 0;
        have = 0;
    }
    __device__ void block() {
        uint32_t x0 = c0, x1 = c1, x2 = draws, x3 = c3, k0 = key0, k1 = key1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint32_t hi0 = __umulhi(0xD2511F53u, x0), lo0 = 0xD2511F53u * x0;
            const uint32_t hi1 = __umulhi(0xCD9E8D57u, x2), lo1 = 0xCD9E8D57u * x2;
            const uint32_t y0 = hi1 ^ x1 ^ k0, y1 = lo1, y2 = hi0 ^ x3 ^ k1, y3 = lo0;
            x0 = y0; x1 = y1; x2 = y2; x3 = y3;
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        out[0] = x0; out[1] = x1; out[2] = x2; out[3] = x3;
        ++draws;
        have = 4;
    }
    // uniform in (0, 1): (m + 1/2) / 2^52 with 52 random bits m, exact in double, never 0 or 1
    __device__ double uniform() {
        if (have == 0) block();
        const bool first = have == 4;
        const uint32_t a = first ? out[0] : out[2], b = first ? out[1] : out[3];
        have -= 2;
        const uint64_t m = ((uint64_t)(a >> 6) << 26) | (uint64_t)(b >> 6);
        return ((double)m + 0.5) * (1.0 / 4503599627370496.0);
    }
    // two independent standard normal deviates (Box-Muller on two uniforms)
    __device__ void normal2(double& z0, double& z1) {
        const double u = uniform(), v = uniform();
        const double r = sqrt(-2.0 * log(u));
        double sn, cs;
        sincospi(2.0 * v, &sn, &cs);
        z0 = r * cs;
        z1 = r * sn;
    }
};

// log(k!) - [ (k + 1/2) log(k + 1) - (k + 1) + log(2 pi)/2 ]: the tail of Stirling's series
__device__ static const double kStirlingTail[10] = { 0.0810614667953272, 0.0413406959554092, 0.0276779256849983,
                                              0.02079067210376509, 0.0166446911898211, 0.0138761288230707,
                                              0.0118967099458917, 0.0104112652619720, 0.00925546218271273,
                                              0.00833056343336287 };
__device__ inline double stirling_tail(double k) {
    if (k <= 9.0) return kStirlingTail[(int)k];
    const double kp1sq = (k + 1.0) * (k + 1.0);
    return (1.0 / 12.0 - (1.0 / 360.0 - 1.0 / 1260.0 / kp1sq) / kp1sq) / (k + 1.0);
}

// k ~ Binomial(n, p), 0 <= p <= 1
__device__ inline int64_t binomial(Philox& rng, int64_t n, double p) {
    if (n <= 0 || !(p > 0.0)) return 0;
    if (p >= 1.0) return n;
    const bool flip = p > 0.5;
    if (flip) p = 1.0 - p;
    const double nd = (double)n, q = 1.0 - p;
    int64_t k;
    if (nd * p < 10.0) {
        // inversion: walk the probability mass function up from 0
        const double s = p / q, a = (nd + 1.0) * s;
        for (;;) {
            double r = exp(nd * log1p(-p)), u = rng.uniform();
            int64_t x = 0;
            bool ok = true;
            while (u > r) {
                u -= r;
                ++x;
                r *= a / (double)x - s;
                if (x > n || !(r > 0.0)) { ok = false; break; }   // rounding ran past the support: draw again
            }
            if (ok) { k = x; break; }
        }
    } else {
        const double spq = sqrt(nd * p * q), b = 1.15 + 2.53 * spq, a = -0.0873 + 0.0248 * b + 0.01 * p,
                     c = nd * p + 0.5, vr = 0.92 - 4.2 / b, r = p / q, alpha = (2.83 + 5.1 / b) * spq,
                     m = floor((nd + 1.0) * p);
        for (;;) {
            const double u = rng.uniform() - 0.5;
            double v = rng.uniform();
            const double us = 0.5 - fabs(u), kd = floor((2.0 * a / us + b) * u + c);
            if (us >= 0.07 && v <= vr) { k = (int64_t)kd; break; }
            if (kd < 0.0 || kd > nd) continue;
            v = log(v * alpha / (a / (us * us) + b));
            const double upper = (m + 0.5) * log((m + 1.0) / (r * (nd - m + 1.0))) +
                                 (nd + 1.0) * log((nd - m + 1.0) / (nd - kd + 1.0)) +
                                 (kd + 0.5) * log(r * (nd - kd + 1.0) / (kd + 1.0)) + stirling_tail(m) +
                                 stirling_tail(nd - m) - stirling_tail(kd) - stirling_tail(nd - kd);
            if (v <= upper) { k = (int64_t)kd; break; }
        }
    }
    return flip ? n - k : k;
}


}  // namespace vqa
