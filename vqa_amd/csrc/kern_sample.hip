// Shot sampling on the device: multinomial outcome counts of the measured low bits.
//
// The reference draws n_shots basis states from a simulated state and keeps only
// `samples % 2` or `samples % 4` (vqa/QNN/myQNNs_base.py:725-727,
// vqa/QNN/myQNNs_classifier.py:613-621, 640-649), i.e. the counts of a multinomial over
// the marginal distribution of the lowest one or two qubits.  The engines return that
// marginal exactly; this kernel draws the counts from it, one row (circuit instance) per
// thread, as a chain of binomials: k_0 ~ Bin(n, p_0), k_1 ~ Bin(n - k_0, p_1 / (1 - p_0)), ...
//
// Random numbers: Philox-4x32-10 keyed by the caller's seed, counter = (row + the caller's row
// offset, draw number, stream), so a row's counts depend only on (seed, stream, row, its probabilities and
// shots), not on the launch shape.  Binomials: sequential inversion while n * min(p, 1-p)
// < 10, else Hoermann's transformed rejection with squeeze (BTRS, 1993); both exact.
#include "philox.h"
#include "vqa_core.h"

namespace vqa {

namespace {

constexpr int kT = 256;

__global__ __launch_bounds__(kT) void k_sample_counts(const double* probs, int M, const int64_t* shots, int64_t E,
                                                      uint64_t seed, uint64_t stream_id, int64_t* counts, int64_t row0) {
    const int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (e >= E) return;
    Philox rng;
    rng.init(seed, stream_id, (uint64_t)(e + row0));
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
                         uint64_t seed, uint64_t stream_id, int64_t* d_counts, int64_t row_offset) {
    if (E <= 0) return VQA_OK;
    hipLaunchKernelGGL(k_sample_counts, dim3((unsigned)((E + kT - 1) / kT)), dim3(kT), 0, stream, d_probs, M, d_shots, E,
                       seed, stream_id, d_counts, row_offset);
    VQA_HIP(hipGetLastError());
    return VQA_OK;
}

}  // namespace vqa
