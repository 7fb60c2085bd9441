// Shot sampling on the device: multinomial outcome counts of the measured low bits.
//
// The reference draws n_shots basis states from a simulated state and keeps only
// `samples % 2` or `samples % 4` (vqa/QNN/myQNNs_base.py:725-727,
// vqa/QNN/myQNNs_classifier.py:613-621, 640-649), i.e. the counts of a multinomial over
// the marginal distribution of the lowest one or two qubits.  The engines return that
// marginal exactly; this kernel draws the counts from it, one row (circuit instance) per
// thread, as a chain of binomials: k_0 ~ Bin(n, p_0), k_1 ~ Bin(n - k_0, p_1 / (1 - p_0)), ...
//
// Random numbers: Philox-4x32-10 keyed by the caller's seed, counter = (row, draw number,
// stream), so a row's counts depend only on (seed, stream, row, its probabilities and
// shots), not on the launch shape.  Binomials: sequential inversion while n * min(p, 1-p)
// < 10, else Hoermann's transformed rejection with squeeze (BTRS, 1993); both exact.
#include "vqa_core.h"

namespace vqa {

namespace {

constexpr int kT = 256;

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
};

// log(k!) - [ (k + 1/2) log(k + 1) - (k + 1) + log(2 pi)/2 ]: the tail of Stirling's series
__device__ const double kStirlingTail[10] = { 0.0810614667953272, 0.0413406959554092, 0.0276779256849983,
                                              0.02079067210376509, 0.0166446911898211, 0.0138761288230707,
                                              0.0118967099458917, 0.0104112652619720, 0.00925546218271273,
                                              0.00833056343336287 };
__device__ double stirling_tail(double k) {
    if (k <= 9.0) return kStirlingTail[(int)k];
    const double kp1sq = (k + 1.0) * (k + 1.0);
    return (1.0 / 12.0 - (1.0 / 360.0 - 1.0 / 1260.0 / kp1sq) / kp1sq) / (k + 1.0);
}

// k ~ Binomial(n, p), 0 <= p <= 1
__device__ int64_t binomial(Philox& rng, int64_t n, double p) {
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

__global__ __launch_bounds__(kT) void k_sample_counts(const double* probs, int M, const int64_t* shots, int64_t E,
                                                      uint64_t seed, uint64_t stream_id, int64_t* counts) {
    const int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (e >= E) return;
    Philox rng;
    rng.init(seed, stream_id, (uint64_t)e);
    int64_t left = shots[e] > 0 ? shots[e] : 0;
    double rest = 1.0;
    const double* p = probs + e * M;
    int64_t* c = counts + e * M;
    for (int o = 0; o < M - 1; ++o) {
        double po = p[o];
        if (!(po > 0.0)) po = 0.0;              // also NaN -> 0
        const double q = rest > 0.0 ? fmin(po / rest, 1.0) : 0.0;
        const int64_t k = binomial(rng, left, q);
        c[o] = k;
        left -= k;
        rest -= po;
    }
    c[M - 1] = left;
}

// out[(j * K + 2 i + 1 | 2) * stride + col] = out[(j * K) * stride + col] for the listed parameters i
__global__ __launch_bounds__(kT) void k_fill_dead(double* out, int64_t n_samples, int K, int stride, const int32_t* list,
                                                  int n_list) {
    const int64_t idx = (int64_t)blockIdx.x * kT + threadIdx.x;
    const int64_t per = (int64_t)n_list * stride;
    if (idx >= n_samples * per) return;
    const int64_t j = idx / per;
    const int r = (int)(idx - j * per), u = r / stride, col = r - u * stride;
    const double v = out[(j * K) * stride + col];
    const int64_t rp = j * K + 2 * (int64_t)list[u] + 1;
    out[rp * stride + col] = v;
    out[(rp + 1) * stride + col] = v;
}

}  // namespace

int fill_dead_rows(hipStream_t stream, double* d_out, int64_t n_samples, int K, int stride, const int32_t* d_list,
                   int n_list) {
    const int64_t tot = n_samples * n_list * stride;
    if (tot <= 0 || !d_out) return VQA_OK;
    hipLaunchKernelGGL(k_fill_dead, dim3((unsigned)((tot + kT - 1) / kT)), dim3(kT), 0, stream, d_out, n_samples, K, stride,
                       d_list, n_list);
    VQA_HIP(hipGetLastError());
    return VQA_OK;
}

int sample_counts_launch(hipStream_t stream, const double* d_probs, int M, const int64_t* d_shots, int64_t E,
                         uint64_t seed, uint64_t stream_id, int64_t* d_counts) {
    if (E <= 0) return VQA_OK;
    hipLaunchKernelGGL(k_sample_counts, dim3((unsigned)((E + kT - 1) / kT)), dim3(kT), 0, stream, d_probs, M, d_shots, E,
                       seed, stream_id, d_counts);
    VQA_HIP(hipGetLastError());
    return VQA_OK;
}

}  // namespace vqa
