/*
 * oracle.c -- CPU restatement of the statevector arithmetic on the hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import, link
 * or call this file; only tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py use it, as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED: the arithmetic restated here lives in two third-party
 * packages the reference imports but does not vendor or pin (qulacs: C++
 * statevector core; skqulacs: LearningCircuit).  Neither exists in this image
 * and the reference's own tests hold no golden vectors for this path
 * (tests/test_sample.py:1-2 is a placeholder).  This file therefore restates
 * the published qulacs conventions and is anchored on the reference's call
 * sites and on analytic known-answer cases (tests/test_known_answers.py):
 *
 *  - LearningCircuit.run(x): fresh |0...0>, input gates take their angle from
 *    input_func(x), then every gate in order
 *    (call sites vqa/QNN/myQNNs_base.py:745,913,1022,1032).
 *  - update_parameters(theta): theta[k] goes to every gate registered to
 *    learning parameter k (vqa/QNN/myQNNs_base.py:758,922,1056,1084,1110).
 *  - Observable.get_expectation_value(state) = sum coef <psi|P|psi>; the empty
 *    Pauli string is the identity (vqa/QNN/myQNNs_base.py:692-704,
 *    vqa/QNN/myQNNs_classifier.py:295-340).
 *  - state.sampling(n) draws basis indices from |amp|^2; the reference only
 *    uses samples % 2 and samples % 4 (vqa/QNN/myQNNs_base.py:725-727,
 *    vqa/QNN/myQNNs_classifier.py:613-619), i.e. the marginal of the lowest
 *    one or two qubits, which is what out_marginal holds.
 *  - qulacs conventions: qubit 0 = least significant index bit;
 *    R{X,Y,Z}(a) = exp(+i a/2 P); CNOT(control, target); CZ symmetric.
 *
 * One state per OpenMP thread, gates applied one by one in the submitted
 * order, no fusion: deliberately the plainest possible code path.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/vqa_hip.h"

typedef struct { double re, im; } c128;

static inline c128 cmul(c128 a, c128 b) {
    c128 r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re };
    return r;
}
static inline c128 cadd(c128 a, c128 b) { c128 r = { a.re + b.re, a.im + b.im }; return r; }

/* dense 2x2 on qubit q */
static void apply_1q(c128* s, int n, int q, const c128 m[4]) {
    const uint64_t D = 1ull << n, bit = 1ull << q;
    for (uint64_t k = 0; k < D; ++k) {
        if (k & bit) continue;
        c128 a0 = s[k], a1 = s[k | bit];
        s[k]       = cadd(cmul(m[0], a0), cmul(m[1], a1));
        s[k | bit] = cadd(cmul(m[2], a0), cmul(m[3], a1));
    }
}

static double op_angle(const vqa_op* op, const double* theta, const double* in_row,
                       int shift_param, double shift_val) {
    double v = 0.0;
    if (op->src == VQA_SRC_THETA) {
        v = theta[op->idx];
        if (op->idx == shift_param) v += shift_val;
    } else if (op->src == VQA_SRC_INPUT) {
        v = in_row[op->idx];
    }
    return op->coef * v + op->offset;
}

/* run the whole gate list on |0...0> */
static void run_circuit(c128* s, int n, const vqa_op* ops, int n_ops,
                        const double* theta, const double* in_row,
                        int shift_param, double shift_val) {
    const uint64_t D = 1ull << n;
    memset(s, 0, D * sizeof(c128));
    s[0].re = 1.0;
    const double r2 = 0.70710678118654752440;
    for (int g = 0; g < n_ops; ++g) {
        const vqa_op* op = &ops[g];
        c128 m[4];
        switch (op->kind) {
        case VQA_GATE_H:
            m[0] = (c128){ r2, 0 }; m[1] = (c128){ r2, 0 };
            m[2] = (c128){ r2, 0 }; m[3] = (c128){ -r2, 0 };
            apply_1q(s, n, op->q0, m);
            break;
        case VQA_GATE_X:
            m[0] = (c128){ 0, 0 }; m[1] = (c128){ 1, 0 };
            m[2] = (c128){ 1, 0 }; m[3] = (c128){ 0, 0 };
            apply_1q(s, n, op->q0, m);
            break;
        case VQA_GATE_Y:
            m[0] = (c128){ 0, 0 }; m[1] = (c128){ 0, -1 };
            m[2] = (c128){ 0, 1 }; m[3] = (c128){ 0, 0 };
            apply_1q(s, n, op->q0, m);
            break;
        case VQA_GATE_Z:
            m[0] = (c128){ 1, 0 }; m[1] = (c128){ 0, 0 };
            m[2] = (c128){ 0, 0 }; m[3] = (c128){ -1, 0 };
            apply_1q(s, n, op->q0, m);
            break;
        case VQA_GATE_RX: {
            double a = op_angle(op, theta, in_row, shift_param, shift_val);
            double c = cos(0.5 * a), sn = sin(0.5 * a);
            m[0] = (c128){ c, 0 }; m[1] = (c128){ 0, sn };
            m[2] = (c128){ 0, sn }; m[3] = (c128){ c, 0 };
            apply_1q(s, n, op->q0, m);
            break;
        }
        case VQA_GATE_RY: {
            double a = op_angle(op, theta, in_row, shift_param, shift_val);
            double c = cos(0.5 * a), sn = sin(0.5 * a);
            m[0] = (c128){ c, 0 }; m[1] = (c128){ sn, 0 };
            m[2] = (c128){ -sn, 0 }; m[3] = (c128){ c, 0 };
            apply_1q(s, n, op->q0, m);
            break;
        }
        case VQA_GATE_RZ: {
            double a = op_angle(op, theta, in_row, shift_param, shift_val);
            double c = cos(0.5 * a), sn = sin(0.5 * a);
            m[0] = (c128){ c, sn }; m[1] = (c128){ 0, 0 };
            m[2] = (c128){ 0, 0 }; m[3] = (c128){ c, -sn };
            apply_1q(s, n, op->q0, m);
            break;
        }
        case VQA_GATE_CNOT: {
            const uint64_t tb = 1ull << op->q0, cb = 1ull << op->q1;
            for (uint64_t k = 0; k < D; ++k) {
                if ((k & cb) && !(k & tb)) {
                    c128 t = s[k]; s[k] = s[k | tb]; s[k | tb] = t;
                }
            }
            break;
        }
        case VQA_GATE_CZ: {
            const uint64_t both = (1ull << op->q0) | (1ull << op->q1);
            for (uint64_t k = 0; k < D; ++k) {
                if ((k & both) == both) { s[k].re = -s[k].re; s[k].im = -s[k].im; }
            }
            break;
        }
        default: break;
        }
    }
}

/* Re <psi| sum_t coef_t P_t |psi>;  P|k> = i^{ny} (-1)^{popcount(k & z)} |k ^ x> */
static double expect_obs(const c128* s, int n, const double* coefs,
                         const uint64_t* zm, const uint64_t* xm, int n_terms) {
    const uint64_t D = 1ull << n;
    double total = 0.0;
    for (int t = 0; t < n_terms; ++t) {
        const uint64_t z = zm[t], x = xm ? xm[t] : 0ull;
        const int ny = __builtin_popcountll(x & z) & 3;
        double re = 0.0, im = 0.0;
        for (uint64_t k = 0; k < D; ++k) {
            c128 a = s[k], b = s[k ^ x];
            /* conj(b) * a */
            double pr = b.re * a.re + b.im * a.im;
            double pi = b.re * a.im - b.im * a.re;
            if (__builtin_popcountll(k & z) & 1) { pr = -pr; pi = -pi; }
            re += pr; im += pi;
        }
        /* multiply by i^ny and keep the real part */
        double v;
        switch (ny) { case 0: v = re; break; case 1: v = -im; break;
                      case 2: v = -re; break; default: v = im; break; }
        total += coefs[t] * v;
    }
    return total;
}

int oracle_version(void) { return VQA_ABI_VERSION; }

/*
 * CPU counterpart of vqa_eval_batch.  Observables are passed flattened:
 * observable o owns terms [obs_off[o], obs_off[o+1]).  n_threads <= 0: all
 * cores OpenMP reports.
 */
int oracle_eval_batch(int n_qubits, const vqa_op* ops, int n_ops, int n_theta, int n_input,
                      const double* theta, const double* input_angles, int B,
                      const int32_t* sample_idx, const int32_t* shift_param,
                      const double* shift_val, int64_t E,
                      const double* obs_coefs, const uint64_t* obs_zmasks,
                      const uint64_t* obs_xmasks, const int32_t* obs_off, int n_obs,
                      int n_marg_bits, double* out_expect, double* out_marginal,
                      int n_threads) {
    if (n_qubits < 1 || n_qubits > 30 || n_marg_bits < 0 || n_marg_bits > n_qubits) return VQA_EINVAL;
    (void)n_theta; (void)B;
    const uint64_t D = 1ull << n_qubits;
    const uint64_t M = 1ull << n_marg_bits;
    int rc = VQA_OK;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
#pragma omp parallel num_threads(n_threads)
    {
        c128* s = (c128*)malloc(D * sizeof(c128));
        if (!s) {
#pragma omp critical
            rc = VQA_ENOMEM;
        }
#pragma omp barrier
        if (rc == VQA_OK) {
#pragma omp for schedule(dynamic, 1)
            for (int64_t e = 0; e < E; ++e) {
                const int b = sample_idx ? sample_idx[e] : (int)e;
                const int sp = shift_param ? shift_param[e] : -1;
                const double sv = shift_val ? shift_val[e] : 0.0;
                const double* row = (n_input > 0 && input_angles) ? input_angles + (size_t)b * n_input : NULL;
                run_circuit(s, n_qubits, ops, n_ops, theta, row, sp, sv);
                for (int o = 0; o < n_obs; ++o) {
                    const int t0 = obs_off[o], t1 = obs_off[o + 1];
                    out_expect[e * n_obs + o] =
                        expect_obs(s, n_qubits, obs_coefs + t0, obs_zmasks + t0,
                                   obs_xmasks ? obs_xmasks + t0 : NULL, t1 - t0);
                }
                if (n_marg_bits > 0 && out_marginal) {
                    double* m = out_marginal + (size_t)e * M;
                    for (uint64_t j = 0; j < M; ++j) m[j] = 0.0;
                    for (uint64_t k = 0; k < D; ++k)
                        m[k & (M - 1)] += s[k].re * s[k].re + s[k].im * s[k].im;
                }
            }
        }
        free(s);
    }
    return rc;
}

/* Final state of one instance, interleaved (re, im). */
int oracle_statevector(int n_qubits, const vqa_op* ops, int n_ops,
                       const double* theta, const double* input_row,
                       int shift_param, double shift_val, double* out_state) {
    if (n_qubits < 1 || n_qubits > 30) return VQA_EINVAL;
    run_circuit((c128*)out_state, n_qubits, ops, n_ops, theta, input_row, shift_param, shift_val);
    return VQA_OK;
}
