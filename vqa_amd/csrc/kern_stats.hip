// Estimator epilogue of the regressor's parameter-shift oracles on the device.
//
// Input: the matrix of exact expectation values of a sweep, E[b, k] (b = sample of the
// mini-batch, k = 0 unshifted, 2i+1 / 2i+2 = parameter i shifted by +-pi/2), the labels, the
// scale parameter w and the shot budget.  Output: per gradient component the batch sums the
// reference's oracles reduce to -- MSE_grad_eval (vqa/QNN/myQNNs_base.py:1346-1395) and MSE_iEval
// with its variance estimators (1421-1596, 1205-1316) -- so that only O(P) numbers leave the GPU.
// One hand-written pass over the matrix replaces the reference's (parameter) x (shift) x (sample)
// Python loop nest around its per-sample estimators (799-905):
//   mode 0  exact        every estimate is the exact value (min_approx_shots == -10, 874-882)
//   mode 1  CLT          estimate = E + z sqrt((1 - E^2) / n), z standard normal (892-905)
//   mode 2  finite shots estimate = (n - 2 k) / n, k ~ Binomial(n, (1 - E) / 2)   (715-738)
//   mode 3  given        the caller supplies the estimates (tests of the algebra)
// Shots of instance (b, i): n = floor(shots[i] * weight[b]) (1367, 1383); the unshifted
// evaluation uses n_shots0.  Random numbers: Philox streams named by (seed, (row0 + b) * K + column), so
// results do not depend on the launch shape.  Unbiased powers: 843-847.
//
// Layout of the work: a workgroup takes a tile of samples; its threads are gradient components
// (thread t: parameters t, t + 256, ...), so a wave reads consecutive (E+, E-) pairs of one row.
// Per-tile partial sums go to the caller's workspace and a second kernel adds the tiles in
// order: the result does not depend on scheduling.
#include "philox.h"
#include "vqa_core.h"

namespace vqa {

namespace {

constexpr int kT = 256;
constexpr int kTile = 32;   // samples per workgroup
constexpr int kQ = 8;       // sums per gradient component

struct StatsArgs {
    const double* E;        // [B, K]
    const double* est;      // mode 3: [B, K]
    const double* y;        // [B]
    const double* shots;    // [P + 1] (scale parameter last); unused in mode 0
    const double* weight;   // [B] or null (all ones)
    int64_t B;
    int64_t row0;           // number of the first row in the whole mini-batch (a shard of it): names the random streams
    int P;
    double w, n_shots0, n0;
    int mode, want_var, second_order;
    uint64_t seed;
    double* partial;        // [tiles, P + 1, kQ]
    double* m0;             // [B] estimates of the unshifted value, or null
    double* gd;             // [P + 1, B] per-sample gradient terms, or null
};

// per-sample quantities of the unshifted evaluation, shared by every component (LDS)
struct Base {
    double EX, resid, err2, alt_err2, var_err, wt, y, pad;
};

__device__ __forceinline__ double ub2(double M, double n) { return (n / (n - 1.0)) * M * M - 1.0 / (n - 1.0); }

__device__ __forceinline__ double estimate1(const StatsArgs& a, double e, double n, uint64_t row) {
    if (a.mode == 0) return e;
    Philox rng;
    rng.init(a.seed, 0, row);
    if (a.mode == 1) {
        double z0, z1;
        rng.normal2(z0, z1);
        return e + z0 * sqrt(fmax(1.0 - e * e, 0.0) / n);
    }
    const int64_t k = binomial(rng, (int64_t)n, fmin(fmax(0.5 * (1.0 - e), 0.0), 1.0));
    return (n - 2.0 * (double)k) / n;
}

__global__ __launch_bounds__(kT) void k_mse_stats(StatsArgs a) {
    __shared__ Base sb[kTile];
    const int K = 2 * a.P + 1;
    const int64_t b0 = (int64_t)blockIdx.x * kTile;
    const int nb = (int)min((int64_t)kTile, a.B - b0);
    const int tid = threadIdx.x;
    const double w = a.w, w2 = w * w;
    const bool so = a.second_order && a.n0 != 0.0;
    double* part = a.partial + (size_t)blockIdx.x * (a.P + 1) * kQ;
    // ---- the unshifted evaluation of every sample of the tile; the scale parameter's sums ----
    double q[kQ];
#pragma unroll
    for (int k = 0; k < kQ; ++k) q[k] = 0.0;
    if (tid < nb) {
        const int64_t b = b0 + tid;
        const double y = a.y[b], wt = a.weight ? a.weight[b] : 1.0;
        const double n = floor(a.n_shots0 * wt);
        const double e = a.E[b * K];
        const double EX = a.mode == 3 ? a.est[b * K] : estimate1(a, e, n, (uint64_t)((b + a.row0) * K));
        double EX_2, EX_3, EX_4, var_0;
        if (a.mode == 0) {
            EX_2 = EX * EX; EX_3 = EX_2 * EX; EX_4 = EX_2 * EX_2;
            var_0 = 1.0 - EX_2;
        } else {
            EX_2 = ub2(EX, n);
            EX_3 = (n / (n - 1.0)) * (n / (n - 2.0)) * EX * EX * EX - (3.0 * n - 2.0) / (n - 1.0) / (n - 2.0) * EX;
            EX_4 = (n / (n - 1.0)) * (n / (n - 2.0)) * (n / (n - 3.0)) * EX * EX * EX * EX -
                   ((n / (n - 1.0)) * ((6.0 * n - 8.0) / (n - 2.0)) * EX * EX - 3.0 / (n - 1.0)) / (n - 3.0);
            var_0 = n / (n - 1.0) * (1.0 - EX * EX);
        }
        Base s;
        s.EX = EX;
        s.y = y;
        s.wt = wt;
        s.resid = (w * EX - y) * w;
        s.err2 = y * y - 2.0 * w * y * EX + w2 * EX_2;
        s.alt_err2 = (y - w * EX) * (y - w * EX);
        s.var_err = w2 * var_0;
        s.pad = 0.0;
        sb[tid] = s;
        if (a.m0) a.m0[b] = EX;
        const double A = -2.0 * y, Bc = 2.0 * w;
        const double gd = 2.0 * (-y * EX + w * EX_2);
        q[0] = gd;
        q[1] = gd * gd;
        if (a.want_var) {
            double est = A * A + 4.0 * A * Bc * EX + (4.0 * Bc * Bc - A * A) * EX_2 - 4.0 * A * Bc * EX_3 - 4.0 * Bc * Bc * EX_4;
            double alt = (A + 2.0 * Bc * EX) * (A + 2.0 * Bc * EX) * var_0;
            if (so) {
                est += 2.0 * Bc * Bc * (1.0 - 2.0 * EX + EX_2) / (a.n0 - 1.0);
                alt += 2.0 * Bc * Bc * var_0 * var_0 / (a.n0 - 1.0);
            }
            q[2] = est;
            q[4] = alt;
            q[6] = A * A * EX_2 + 2.0 * A * Bc * EX_3 + Bc * Bc * EX_4;
        }
        if (a.gd) a.gd[(size_t)a.P * a.B + b] = gd;
    }
    __syncthreads();
    // the scale parameter's sums over the tile, in sample order (thread 0; nb <= 32 values each)
    {
        __shared__ double red[kTile][kQ];
        if (tid < nb) {
#pragma unroll
            for (int k = 0; k < kQ; ++k) red[tid][k] = q[k];
        }
        __syncthreads();
        if (tid < kQ) {
            double s = 0.0;
            for (int j = 0; j < nb; ++j) s += red[j][tid];
            part[(size_t)a.P * kQ + tid] = s;
        }
    }
    // ---- the circuit parameters: thread = component, loop over the tile's samples ----
    for (int i = tid; i < a.P; i += kT) {
#pragma unroll
        for (int k = 0; k < kQ; ++k) q[k] = 0.0;
        const double shots_i = a.mode == 0 ? 0.0 : a.shots[i];
        for (int j = 0; j < nb; ++j) {
            const int64_t b = b0 + j;
            const Base s = sb[j];
            const double* ep = a.E + b * K + 1 + 2 * i;
            const double2 e = make_double2(ep[0], ep[1]);   // (E+, E-)
            double Mp, Mm;
            const double n = floor(shots_i * s.wt);
            if (a.mode == 0) {
                Mp = e.x; Mm = e.y;
            } else if (a.mode == 3) {
                Mp = a.est[b * K + 1 + 2 * i];
                Mm = a.est[b * K + 2 + 2 * i];
            } else if (a.mode == 1) {
                Philox rng;
                rng.init(a.seed, 0, (uint64_t)((b + a.row0) * K + 1 + 2 * i));
                double z0, z1;
                rng.normal2(z0, z1);
                Mp = e.x + z0 * sqrt(fmax(1.0 - e.x * e.x, 0.0) / n);
                Mm = e.y + z1 * sqrt(fmax(1.0 - e.y * e.y, 0.0) / n);
            } else {
                Philox rng;
                rng.init(a.seed, 0, (uint64_t)((b + a.row0) * K + 1 + 2 * i));
                const int64_t kp = binomial(rng, (int64_t)n, fmin(fmax(0.5 * (1.0 - e.x), 0.0), 1.0));
                const int64_t km = binomial(rng, (int64_t)n, fmin(fmax(0.5 * (1.0 - e.y), 0.0), 1.0));
                Mp = (n - 2.0 * (double)kp) / n;
                Mm = (n - 2.0 * (double)km) / n;
            }
            const double d = Mp - Mm, gd = s.resid * d;
            q[0] += gd;
            q[1] += gd * gd;
            if (a.want_var) {
                const double EXp_2 = ub2(Mp, n), EXm_2 = ub2(Mm, n);
                const double var_p = n / (n - 1.0) * (1.0 - Mp * Mp), var_m = n / (n - 1.0) * (1.0 - Mm * Mm);
                const double dp2 = EXp_2 - 2.0 * Mp * Mm + EXm_2;
                const double var_dpdt = w2 * (var_p + var_m);
                const double extra = so ? s.var_err * var_dpdt / a.n0 : 0.0;
                q[2] += s.err2 * var_dpdt + extra;        // E_x[Var] main form, shifted part
                q[3] += s.var_err * w2 * dp2;             //               ... unshifted part
                q[4] += s.alt_err2 * var_dpdt + extra;    // the forms used when a mean comes out negative
                q[5] += s.var_err * w2 * d * d;
                q[6] += s.err2 * w2 * dp2;                // second moment of the per-sample gradient term
            }
            if (a.gd) a.gd[(size_t)i * a.B + b] = gd;
        }
#pragma unroll
        for (int k = 0; k < kQ; ++k) part[(size_t)i * kQ + k] = q[k];
    }
}

// out[c, k] = sum over tiles (in order) of partial[tile, c, k]
// Sums the tiles' partial results: out[idx] = sum over tiles, idx = (row, quantity).  A workgroup takes 16
// consecutive idx (128-byte reads) and deals the tiles to 16 groups of threads; each thread adds its
// tiles in ascending order and the 16 group sums meet in LDS in a fixed tree, so the result does not
// depend on timing (it does differ in the last bits from a single ascending sum).
__global__ __launch_bounds__(256) void k_mse_stats_finish(const double* partial, int n_tiles, int n_rows, double* out) {
    __shared__ double red[16][17];
    const int i = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int n = n_rows * kQ, idx = blockIdx.x * 16 + i;
    double s = 0.0;
    if (idx < n)
        for (int t = g; t < n_tiles; t += 16) s += partial[(size_t)t * n + idx];
    red[g][i] = s;
    __syncthreads();
    for (int half = 8; half >= 1; half >>= 1) {
        if (g < half) red[g][i] += red[g + half][i];
        __syncthreads();
    }
    if (g == 0 && idx < n) out[idx] = red[0][i];
}

// ---- independent random allocation (gCANS / Refoqus): MSE_iEval_iRandom ---------------------------
// vqa/QNN/myQNNs_base.py:1619-1666 (regressor) and vqa/QNN/myQNNs_classifier.py:1017-1053
// (classifier, partial MSE).  Per gradient component i the reference allocates shots to samples
// by a multinomial draw, evaluates the unshifted and the two shifted circuits of every sample
// that received shots, keeps the raw outcomes and forms (a) the gradient estimate from the
// outcome sums and (b) the sample variance of the per-shot products, the k-th shots of the three
// circuits taken together.  The three outcome sequences are independent, so the counts of the
// joint outcomes of a (component, sample) row are multinomial with product probabilities: the
// kernel draws those counts (a chain of binomials) instead of materialising shots, which gives
// the same distribution of every quantity the reference computes.
// A row is (component, sample): expect[(3 r + v) * n_obs + col], v = 0 unshifted, 1 / 2 shifted;
// rows of one component are contiguous, seg[i] .. seg[i + 1].  kind 0: Z0 outcomes +-1 of the
// regressor (col 0; the last component is the scale parameter, whose rows use twice their shots,
// split in halves: 1642-1653); kind 1: 'the true class was read' outcomes 1 / 0 of the classifier
// (col = the row's label).  One workgroup per component; fixed-order reductions.
struct IRandomArgs {
    const double* expect;
    int n_obs;
    const int32_t* label;     // [R] (kind 1)
    const double* y;          // [R] (kind 0)
    const int64_t* shots;     // [R] shots of the row (before doubling, scale rows)
    const int32_t* seg;       // [C + 1]
    const double* n_list;     // [C] total shots of each component
    int C, kind, has_scale;
    double w, M_tot;
    uint64_t seed;
    double* out;              // [C, 2]: gradient, variance
};

__global__ __launch_bounds__(kT) void k_irandom_stats(IRandomArgs a) {
    __shared__ double red[4][kT];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int r0 = a.seg[c], r1 = a.seg[c + 1];
    const double n = a.n_list[c], w = a.w;
    const bool scale = a.has_scale && c == a.C - 1;
    double acc[4] = { 0.0, 0.0, 0.0, 0.0 };   // gradient terms, sum g, sum g^2, shots
    for (int r = r0 + tid; r < r1; r += kT) {
        const int col = a.kind == 1 ? a.label[r] : 0;
        const double e0 = a.expect[(size_t)(3 * (size_t)r + 0) * a.n_obs + col];
        const double ep = a.expect[(size_t)(3 * (size_t)r + 1) * a.n_obs + col];
        const double em = a.expect[(size_t)(3 * (size_t)r + 2) * a.n_obs + col];
        const int64_t s = a.shots[r];
        Philox rng;
        rng.init(a.seed, 1, (uint64_t)r);
        // probability of the outcome that is counted: -1 (kind 0), success (kind 1)
        auto pr = [&](double e) { return fmin(fmax(a.kind == 0 ? 0.5 * (1.0 - e) : e, 0.0), 1.0); };
        const double p0 = pr(e0), pp = pr(ep), pm = pr(em);
        if (scale) {
            // 2 s outcomes of the unshifted circuit, halves paired shot by shot: g = 2 (w o1 - y) o2
            const double y = a.y[r];
            const int64_t k1 = binomial(rng, s, p0);                 // o1 = -1
            const int64_t k2a = binomial(rng, k1, p0);               // o2 = -1 where o1 = -1
            const int64_t k2b = binomial(rng, s - k1, p0);           // o2 = -1 where o1 = +1
            const double outsum = (double)(s - 2 * k1) + (double)(s - 2 * (k2a + k2b));
            const double n_eval = 2.0 * (double)s;
            acc[0] += w * (outsum * outsum - n_eval) / (2.0 * n * (n - 1.0) / a.M_tot + n) - y * outsum / n;
            const double gm = 2.0 * (-w - y), gp = 2.0 * (w - y);   // (w o1 - y) * 2 for o1 = -1 / +1
            // o2 = -1 flips the sign
            const double c_mm = (double)k2a, c_mp = (double)(k1 - k2a), c_pm = (double)k2b, c_pp = (double)(s - k1 - k2b);
            acc[1] += gm * (c_mp - c_mm) + gp * (c_pp - c_pm);
            acc[2] += gm * gm * (double)k1 + gp * gp * (double)(s - k1);
            acc[3] += (double)s;
        } else {
            const int64_t k0 = binomial(rng, s, p0);                 // counted outcome of the unshifted circuit
            const int64_t kpa = binomial(rng, k0, pp), kpb = binomial(rng, s - k0, pp);
            // of each (o0, o+) group, the shots whose o- is the counted outcome
            const int64_t kma = binomial(rng, kpa, pm), kmb = binomial(rng, k0 - kpa, pm);
            const int64_t kmc = binomial(rng, kpb, pm), kmd = binomial(rng, s - k0 - kpb, pm);
            const double sp = (double)(kpa + kpb), sm = (double)(kma + kmb + kmc + kmd), s0 = (double)k0, sd = (double)s;
            // shots with o+ != o-: (o+ counted, o- not) and (o+ not, o- counted), per value of o0
            const double d_a = (double)(kpa - kma), u_a = (double)kmb;      // o0 counted
            const double d_b = (double)(kpb - kmc), u_b = (double)kmd;      // o0 not counted
            if (a.kind == 0) {
                const double y = a.y[r];
                const double out0 = sd - 2.0 * s0, outp = sd - 2.0 * sp, outm = sd - 2.0 * sm;
                acc[0] += (w * out0 / (n + n * (n - 1.0) / a.M_tot) - y / n) * (outp - outm) * w;
                // g = (w o0 - y) w (o+ - o-); o+ - o- = -2 when only o+ is -1, +2 when only o- is -1
                const double fm = (-w - y) * w, fp = (w - y) * w;        // o0 = -1 / +1
                acc[1] += 2.0 * (fm * (u_a - d_a) + fp * (u_b - d_b));
                acc[2] += 4.0 * (fm * fm * (u_a + d_a) + fp * fp * (u_b + d_b));
            } else {
                acc[0] += (s0 / (1.0 + (n - 1.0) / a.M_tot) - 1.0) * (sp - sm) / n;
                // g = (o0 - 1)(o+ - o-): nonzero only where o0 = 0 (not counted): -(o+ - o-)
                acc[1] += -(d_b - u_b);
                acc[2] += d_b + u_b;
            }
            acc[3] += sd;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[k][tid] = acc[k];
    __syncthreads();
    for (int h = kT / 2; h > 0; h >>= 1) {
        if (tid < h) {
#pragma unroll
            for (int k = 0; k < 4; ++k) red[k][tid] += red[k][tid + h];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double N = red[3][0];
        a.out[2 * c] = red[0][0];
        a.out[2 * c + 1] = (red[2][0] - red[1][0] * red[1][0] / N) / (N - 1.0);   // np.var(..., ddof=1)
    }
}

// ---- classifier MSE gradient from drawn class counts ----------------------------------------------
// vqa/QNN/myQNNs_classifier.py:1055-1156 (MSE_grad_eval): with q = counts / shots the class
// frequencies of the unshifted (k = 0) and shifted (k = 1 + 2 i, 2 + 2 i) circuits of sample b,
//   partial MSE:  grad[i] = mean_b (q0[y_b] - 1) (q+[y_b] - q-[y_b])
//   full MSE:     grad[i] = mean_b sum_c (q0[c] - onehot_b[c]) (q+[c] - q-[c]).
// One workgroup per component, a thread takes every kT-th sample, fixed-order tree sum.
__global__ __launch_bounds__(kT) void k_class_grad(const int64_t* __restrict__ counts, const int64_t* __restrict__ shots,
                                                   int64_t B, int P, const int32_t* __restrict__ label,
                                                   const double* __restrict__ y1hot, double* __restrict__ grad) {
    __shared__ double red[kT];
    const int i = blockIdx.x, tid = threadIdx.x, K = 2 * P + 1;
    double acc = 0.0;
    for (int64_t b = tid; b < B; b += kT) {
        const int64_t r0 = b * K, rp = r0 + 1 + 2 * i, rm = rp + 1;
        const double n0 = (double)shots[r0], np_ = (double)shots[rp], nm = (double)shots[rm];
        if (label) {
            const int y = label[b];
            acc += ((double)counts[3 * r0 + y] / n0 - 1.0) * ((double)counts[3 * rp + y] / np_ - (double)counts[3 * rm + y] / nm);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                acc += ((double)counts[3 * r0 + c] / n0 - y1hot[3 * b + c]) *
                       ((double)counts[3 * rp + c] / np_ - (double)counts[3 * rm + c] / nm);
        }
    }
    red[tid] = acc;
    __syncthreads();
    for (int h = kT / 2; h > 0; h >>= 1) {
        if (tid < h) red[tid] += red[tid + h];
        __syncthreads();
    }
    if (tid == 0) grad[i] = red[0] / (double)B;
}

// ---- classifier: gradient terms and conditional-variance estimates from drawn class counts ---------
// vqa/QNN/myQNNs_classifier.py: MSE_iEval (888-999) with _estimate_MSEgrad_mean_condVar (861-886),
// CE_grad_eval / CE_iEval (1172-1269) with _estimate_CEgrad_mean_condVar (1158-1169).  From the class
// counts of the unshifted (k = 0) and shifted (k = 1 + 2 i, 2 + 2 i) circuits of every sample, a count c of
// n shots gives p = c / n, the unbiased p2 = n / (n - 1) p^2 - p / (n - 1) and the sample variance
// varp = c / (n - 1) - c^2 / (n (n - 1)) (623-627, 661-665).  Per component i the kernel sums over the samples
//   [0] the per-sample gradient term, [1] the conditional-variance estimate, [2] the form the reference
//   falls back to when the batch mean of [1] comes out negative:
//   kind 0, partial MSE (success = the label's class): gd = (p0 - 1)(p+ - p-);
//       [1] (p02 - 2 p0 + 1) vardp + varp0 (p+2 + p-2 - 2 p+ p-),  [2] (p0 - 1)^2 vardp + varp0 (p+ - p-)^2
//   kind 1, full MSE, summed over the three classes with y the one-hot label: gd = sum (p0 - y)(p+ - p-);
//       [1] sum (p02 + (1 - 2 p0) y) vardp + varp0 (p+2 + p-2 - 2 p+ p-),  [2] sum (p0 - y)^2 vardp + varp0 (p+ - p-)^2
//   kind 2, cross entropy from success counts: gd = -(c+ - c-) / (2 (c0 + 1)) (n0 + 1) / (n + 1);
//       with t = (n0 + 1) / (c0 + 1): [1] (t^2 (varp+ + varp-) + t^2 (t - 1)(p+2 + p-2 - 2 p+ p-)) / 4,
//       [2] the same with (p+ - p-)^2
// (vardp = varp+ + varp-).  Sums, not means: a batch that is split over several GPUs adds them.  One workgroup
// per component, a thread takes every kT-th sample, fixed-order tree sum.
struct Moments { double p, p2, varp; };
__device__ __forceinline__ Moments count_moments(double c, double n) {
    Moments m;
    m.p = c / n;
    m.varp = c / (n - 1.0) - c * c / (n * (n - 1.0));
    m.p2 = (n / (n - 1.0)) * m.p * m.p - m.p / (n - 1.0);
    return m;
}

__global__ __launch_bounds__(kT) void k_class_stats(const int64_t* __restrict__ counts, const int64_t* __restrict__ shots,
                                                    int64_t B, int P, const int32_t* __restrict__ label, int kind,
                                                    int want_var, double* __restrict__ out, double* __restrict__ gd_out) {
    __shared__ double red[3][kT];
    const int i = blockIdx.x, tid = threadIdx.x, K = 2 * P + 1;
    double acc[3] = { 0.0, 0.0, 0.0 };
    for (int64_t b = tid; b < B; b += kT) {
        const int64_t r0 = b * K, rp = r0 + 1 + 2 * i, rm = rp + 1;
        const double n0 = (double)shots[r0], n = (double)shots[rp];
        const int y = label[b];
        double gd = 0.0, main_v = 0.0, alt_v = 0.0;
        if (kind == 2) {
            const double c0 = (double)counts[3 * r0 + y], cp = (double)counts[3 * rp + y], cm = (double)counts[3 * rm + y];
            gd = -0.5 * (cp - cm) / (c0 + 1.0) * ((n0 + 1.0) / (n + 1.0));
            if (want_var) {
                const Moments mp = count_moments(cp, n), mm = count_moments(cm, n);
                const double t = (n0 + 1.0) / (c0 + 1.0), term1 = t * t * (mp.varp + mm.varp), vt = t * t * (t - 1.0);
                const double d = mp.p - mm.p;
                main_v = 0.25 * (term1 + vt * (mp.p2 + mm.p2 - 2.0 * mp.p * mm.p));
                alt_v = 0.25 * (term1 + vt * d * d);
            }
        } else {
            const int c_lo = kind == 0 ? y : 0, c_hi = kind == 0 ? y + 1 : 3;
            for (int c = c_lo; c < c_hi; ++c) {
                const Moments m0 = count_moments((double)counts[3 * r0 + c], n0);
                const Moments mp = count_moments((double)counts[3 * rp + c], n), mm = count_moments((double)counts[3 * rm + c], n);
                const double yc = (kind == 0 || c == y) ? 1.0 : 0.0;
                const double d = mp.p - mm.p;
                gd += (m0.p - yc) * d;
                if (want_var) {
                    const double vardp = mp.varp + mm.varp;
                    // (partial MSE: y = 1, so p02 + (1 - 2 p0) y = p02 - 2 p0 + 1)
                    main_v += (m0.p2 + (1.0 - 2.0 * m0.p) * yc) * vardp + m0.varp * (mp.p2 + mm.p2 - 2.0 * mp.p * mm.p);
                    alt_v += (m0.p - yc) * (m0.p - yc) * vardp + m0.varp * d * d;
                }
            }
        }
        acc[0] += gd; acc[1] += main_v; acc[2] += alt_v;
        if (gd_out) gd_out[(size_t)i * B + b] = gd;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) red[k][tid] = acc[k];
    __syncthreads();
    for (int h = kT / 2; h > 0; h >>= 1) {
        if (tid < h) {
#pragma unroll
            for (int k = 0; k < 3; ++k) red[k][tid] += red[k][tid + h];
        }
        __syncthreads();
    }
    if (tid < 4) out[4 * i + tid] = tid < 3 ? red[tid][0] : 0.0;
}

}  // namespace

int class_stats_launch(hipStream_t stream, const int64_t* d_counts, const int64_t* d_shots, int64_t B, int P,
                       const int32_t* d_label, int kind, int want_var, double* d_out, double* d_gd) {
    if (P <= 0) return VQA_OK;
    hipLaunchKernelGGL(k_class_stats, dim3(P), dim3(kT), 0, stream, d_counts, d_shots, B, P, d_label, kind, want_var, d_out, d_gd);
    VQA_HIP(hipGetLastError());
    return VQA_OK;
}

int class_grad_launch(hipStream_t stream, const int64_t* d_counts, const int64_t* d_shots, int64_t B, int P,
                      const int32_t* d_label, const double* d_y1hot, double* d_grad) {
    if (P <= 0 || B <= 0) return VQA_OK;
    hipLaunchKernelGGL(k_class_grad, dim3(P), dim3(kT), 0, stream, d_counts, d_shots, B, P, d_label, d_y1hot, d_grad);
    VQA_HIP(hipGetLastError());
    return VQA_OK;
}

size_t mse_stats_workspace_bytes(int64_t B, int P) {
    const int64_t tiles = (B + kTile - 1) / kTile;
    return (size_t)tiles * (size_t)(P + 1) * kQ * sizeof(double);
}

int mse_stats_launch(hipStream_t stream, const double* d_E, const double* d_est, int64_t B, int P, const double* d_y,
                     double w, const double* d_shots, const double* d_weight, double n_shots0, int mode, uint64_t seed,
                     int want_var, double n0, int second_order, double* d_workspace, double* d_out, double* d_m0,
                     double* d_gd, int64_t row_offset) {
    if (B <= 0) return VQA_OK;
    StatsArgs a;
    a.E = d_E; a.est = d_est; a.y = d_y; a.shots = d_shots; a.weight = d_weight;
    a.B = B; a.row0 = row_offset; a.P = P; a.w = w; a.n_shots0 = n_shots0; a.n0 = n0;
    a.mode = mode; a.want_var = want_var; a.second_order = second_order; a.seed = seed;
    a.partial = d_workspace; a.m0 = d_m0; a.gd = d_gd;
    const int64_t tiles = (B + kTile - 1) / kTile;
    hipLaunchKernelGGL(k_mse_stats, dim3((unsigned)tiles), dim3(kT), 0, stream, a);
    VQA_HIP(hipGetLastError());
    const int n = (P + 1) * kQ;
    hipLaunchKernelGGL(k_mse_stats_finish, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, stream, d_workspace, (int)tiles,
                       P + 1, d_out);
    VQA_HIP(hipGetLastError());
    return VQA_OK;
}

int irandom_stats_launch(hipStream_t stream, const double* d_expect, int n_obs, const int32_t* d_label, const double* d_y,
                         const int64_t* d_shots, const int32_t* d_seg, const double* d_n_list, int C, int kind,
                         int has_scale, double w, double M_tot, uint64_t seed, double* d_out) {
    if (C <= 0) return VQA_OK;
    IRandomArgs a;
    a.expect = d_expect; a.n_obs = n_obs; a.label = d_label; a.y = d_y; a.shots = d_shots; a.seg = d_seg;
    a.n_list = d_n_list; a.C = C; a.kind = kind; a.has_scale = has_scale; a.w = w; a.M_tot = M_tot; a.seed = seed;
    a.out = d_out;
    hipLaunchKernelGGL(k_irandom_stats, dim3(C), dim3(kT), 0, stream, a);
    VQA_HIP(hipGetLastError());
    return VQA_OK;
}

}  // namespace vqa
