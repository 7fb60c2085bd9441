/*
 * vqa_hip.h -- C ABI of libvqahip.so, the MI355X (gfx950) statevector engine
 * for batched evaluation of parameterized circuits and their parameter-shift
 * gradients.
 *
 * The reference (Qulacs-Osaka/VQA, pure Python) reaches this arithmetic through
 * a duck-typed skqulacs LearningCircuit and qulacs objects, one data sample at
 * a time.  Each entry point below names the reference call sites it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions (qulacs): basis index bit k <-> qubit k (qubit 0 is the least
 * significant bit); RX/RY/RZ(a) = exp(+i a/2 P):
 *   RX = [[c, i s], [i s, c]]   RY = [[c, s], [-s, c]]   RZ = diag(e^{+ia/2}, e^{-ia/2})
 * with c = cos(a/2), s = sin(a/2).  CNOT(control, target); CZ symmetric.
 * Every circuit starts from |0...0>.
 *
 * Error model: every function returns VQA_OK (0) or a negative VQA_E* code,
 * never throws or aborts across the boundary, and leaves outputs untouched on
 * error.  vqa_last_error() returns a thread-local message for the last
 * failure on the calling thread.
 *
 * Ownership: the caller owns every pointer it passes and keeps it alive until
 * the call returns (host-pointer entry points are synchronous) or until the
 * stream work has completed (device-pointer entry points are asynchronous on
 * the caller's stream).  The library owns the memory behind opaque handles;
 * handles are immutable after creation, may be shared between threads, and
 * are freed only by the matching *_destroy.
 */
#ifndef VQA_HIP_H
#define VQA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQA_ABI_VERSION 1

/* error codes */
#define VQA_OK            0
#define VQA_EINVAL       (-1)  /* bad argument (message says which) */
#define VQA_ENOMEM       (-2)  /* host or device allocation failed */
#define VQA_EHIP         (-3)  /* a HIP runtime call failed */
#define VQA_EUNSUPPORTED (-4)  /* valid request outside what the engine implements */
#define VQA_ENODEVICE    (-5)  /* no usable GPU */

/* gate kinds of the flat op table */
#define VQA_GATE_H     0
#define VQA_GATE_X     1
#define VQA_GATE_Y     2
#define VQA_GATE_Z     3
#define VQA_GATE_RX    4
#define VQA_GATE_RY    5
#define VQA_GATE_RZ    6
#define VQA_GATE_CNOT  7   /* q0 = target, q1 = control */
#define VQA_GATE_CZ    8   /* q0, q1 symmetric */

/* where a rotation takes its angle from */
#define VQA_SRC_CONST  0   /* angle = offset */
#define VQA_SRC_THETA  1   /* angle = coef * (theta[idx] + shift if idx is the shifted parameter) + offset */
#define VQA_SRC_INPUT  2   /* angle = coef * input_angles[sample, idx] + offset */

/*
 * One gate of the submitted (unfused) gate list.  Replaces the gate objects a
 * LearningCircuit accumulates through add_H_gate / add_CNOT_gate / add_gate(CZ)
 * / add_input_RZ_gate / add_parametric_R{X,Y,Z}_gate
 * (vqa/QNN/QML_utils.py:47-98, 114-163).
 */
typedef struct vqa_op {
    int32_t kind;     /* VQA_GATE_* */
    int32_t q0;       /* target qubit */
    int32_t q1;       /* control / second qubit, -1 when unused */
    int32_t src;      /* VQA_SRC_* (rotations only) */
    int32_t idx;      /* index into theta[] or into one row of input_angles */
    int32_t reserved; /* must be 0 */
    double  coef;     /* multiplies the sourced value */
    double  offset;   /* added to it (the whole angle for VQA_SRC_CONST) */
} vqa_op;

typedef struct vqa_circuit vqa_circuit;   /* opaque: gate list + compiled device programs */
typedef struct vqa_obs vqa_obs;           /* opaque: a real linear combination of Pauli strings */

/* ---- library ---------------------------------------------------------- */

/* ABI version (VQA_ABI_VERSION of the built library). */
int vqa_version(void);
/* Number of visible GPUs, or a negative VQA_E* code. */
int vqa_device_count(void);
/* Message for the last error on this thread ("" if none). Never NULL. */
const char* vqa_last_error(void);
/* Block until all work the library queued on its own stream of `device` is done. */
int vqa_sync(int device);

/* ---- circuits ---------------------------------------------------------- */

/*
 * Build a circuit from a flat op table.  n_theta is the number of trainable
 * parameters P (what len(circuit.get_parameters()) returns in the reference,
 * vqa/QNN/myQNNs_base.py:290), n_input the row length of input_angles.
 * Replaces LearningCircuit construction (vqa/QNN/QML_utils.py:33-164).
 */
int vqa_circuit_create(int n_qubits, const vqa_op* ops, int n_ops,
                       int n_theta, int n_input, vqa_circuit** out);
int vqa_circuit_destroy(vqa_circuit* c);
/* Introspection: qubits, ops, parameters, inputs of a circuit. */
int vqa_circuit_info(const vqa_circuit* c, int* n_qubits, int* n_ops,
                     int* n_theta, int* n_input);
/* The engine that serves this circuit's Z-basis calls (1..4, see vqa_set_engine), or a
 * negative error code.  No counterpart in the reference; the host layer uses it to decide
 * whether X/Y Pauli terms are measured behind basis-change gates on the same engine. */
int vqa_circuit_engine(const vqa_circuit* c);

/* ---- observables -------------------------------------------------------- */

/*
 * sum_t coefs[t] * P_t with P_t the Pauli string that has Z on the set bits of
 * zmasks[t] only, X on the set bits of xmasks[t] only, and Y where both are
 * set; both masks 0 = identity (contributes coefs[t] * <psi|psi>).  xmasks may
 * be NULL (all Z-strings).  Replaces qulacs Observable(n) + add_operator(coef,
 * "Z i Z j") (vqa/QNN/myQNNs_base.py:353-361, vqa/QNN/myQNNs_classifier.py:295-339).
 */
int vqa_observable_create(int n_qubits, const double* coefs,
                          const uint64_t* zmasks, const uint64_t* xmasks,
                          int n_terms, vqa_obs** out);
int vqa_observable_destroy(vqa_obs* o);

/* ---- batched evaluation ------------------------------------------------- */

/*
 * Evaluate E circuit instances.  Instance e runs the circuit on sample
 * sample_idx[e] (a row of input_angles[B, n_input], row-major) with parameters
 * theta[P], where theta[shift_param[e]] is replaced by theta[..] + shift_val[e]
 * (shift_param[e] = -1: no shift), and writes
 *   out_expect[e * n_obs + o]            = Re <psi_e| obs[o] |psi_e>
 *   out_marginal[e * 2^n_marg_bits + m]  = Prob(lowest n_marg_bits qubits read m)
 * out_marginal may be NULL when n_marg_bits == 0; out_expect may be NULL when
 * n_obs == 0.  sample_idx may be NULL when E == B (instance e uses sample e);
 * shift_param/shift_val may both be NULL (no shifts).
 *
 * One instance replaces circuit.update_parameters(theta) + circuit.run(x) +
 * Observable.get_expectation_value(state) (vqa/QNN/myQNNs_base.py:692-704,
 * 907-914, 1022-1034, 1110-1125) and, through the marginal, the only use the
 * reference makes of state.sampling(): samples % 2 / samples % 4
 * (vqa/QNN/myQNNs_base.py:725-727, vqa/QNN/myQNNs_classifier.py:613-619).
 *
 * Host pointers; synchronous; runs on GPU `device`.
 */
int vqa_eval_batch(vqa_circuit* c, int device,
                   const double* theta, const double* input_angles, int B,
                   const int32_t* sample_idx, const int32_t* shift_param,
                   const double* shift_val, int64_t E,
                   const vqa_obs* const* obs, int n_obs, int n_marg_bits,
                   double* out_expect, double* out_marginal);

/*
 * Same computation with every data pointer a DEVICE pointer on `device`, queued
 * on `stream` (a hipStream_t; NULL = HIP's default stream) and not waited for.
 * Inputs may stay resident in HBM between calls.  Instance lists cannot be
 * checked on the host here: an instance that names a sample >= B or a
 * parameter >= P gets NaN in all its outputs.
 */
int vqa_eval_batch_dev(vqa_circuit* c, int device, void* stream,
                       const double* d_theta, const double* d_input_angles, int B,
                       const int32_t* d_sample_idx, const int32_t* d_shift_param,
                       const double* d_shift_val, int64_t E,
                       const vqa_obs* const* obs, int n_obs, int n_marg_bits,
                       double* d_out_expect, double* d_out_marginal);

/*
 * Full parameter-shift sweep for n_sel samples: for sample s = sel[j] (sel NULL:
 * s = j, n_sel <= B) evaluates the 2P+1 instances
 *   k = 0        : theta
 *   k = 2i + 1   : theta + shift * e_i
 *   k = 2i + 2   : theta - shift * e_i          (i = 0..P-1)
 * and writes out_expect[(j * (2P+1) + k) * n_obs + o] and
 * out_marginal[(j * (2P+1) + k) * 2^n_marg_bits + m].  The instance list is
 * generated on the device.  Replaces the loop nest of the parameter-shift
 * oracles: base evaluation + for each parameter the +/- pi/2 pair over the
 * mini-batch (vqa/QNN/myQNNs_base.py:1370-1394, 1469-1530;
 * vqa/QNN/myQNNs_classifier.py:1116-1156).
 *
 * Sweeps (and every call served by the streaming engine, n >= 14) use a
 * workspace the library keeps per device and grows on demand.  Calls on one
 * device may come from different threads and streams: a call holds a
 * per-device lock while it queues its work, and a call that takes the
 * workspace makes its stream wait for the previous user's last kernel (an
 * event), so such calls run one after the other on the GPU.  Different devices
 * are independent.  A sweep of several samples on the streaming engine may
 * spread them over a few more streams of the library's own: they start behind
 * an event on the caller's stream, and the caller's stream waits for them
 * before the call returns, so the caller orders against `stream` alone.
 *
 * Light cones.  <0|U^+ P U|0> of a Pauli string P depends only on the gates inside the backward
 * light cone of P's qubits, and those gates touch the cone's qubits alone.  When the library chose
 * the engine itself (vqa_set_engine(0)) and the cones of a call's terms are narrower than the
 * register -- local terms behind a shallow circuit on many qubits, the energy of a variational
 * eigensolver -- vqa_eval_batch[_dev], vqa_psr_full[_dev] and vqa_psr_params_dev deal the terms
 * into groups and evaluate every group on the circuit reduced to its cone (cone qubits renumbered,
 * gates outside dropped, X / Y factors measured as Z behind appended basis gates; parameters and
 * input angles keep their numbers), then add the groups' results into the caller's output in a
 * fixed order.  The values are those of the whole register up to rounding.  A call of few
 * instances runs its groups on the caller's stream and on streams of the library's own, ordered
 * against `stream` as described above.  VQA_NO_CONE_SPLIT=1 in the environment switches it off.
 */
int vqa_psr_full(vqa_circuit* c, int device,
                 const double* theta, const double* input_angles, int B,
                 const int32_t* sel, int n_sel, double shift,
                 const vqa_obs* const* obs, int n_obs, int n_marg_bits,
                 double* out_expect, double* out_marginal);

int vqa_psr_full_dev(vqa_circuit* c, int device, void* stream,
                     const double* d_theta, const double* d_input_angles, int B,
                     const int32_t* d_sel, int n_sel, double shift,
                     const vqa_obs* const* obs, int n_obs, int n_marg_bits,
                     double* d_out_expect, double* d_out_marginal);

/*
 * vqa_psr_full_dev that allocates nothing.  No counterpart in the reference: it is for callers
 * that cannot allocate device memory when they call -- code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  Same arguments, same outputs.  Everything the
 * sweep needs besides the caller's inputs and outputs (the compiled program, the work counter,
 * the run lists, the state images and run records of the sweep's workgroups) lives in a static
 * device array of VQA_NOALLOC_SCRATCH_BYTES (32 MiB) that every GPU the library runs on holds until the
 * process ends; small host tables reach it as kernel arguments of a copy kernel, not through
 * hipMemcpy.  The call makes no hipMalloc / hipFree and uses no library workspace.
 *
 * Limits: circuits served by the register engine (4 to 14 qubits, Z-basis observables, at most
 * 6 marginal bits); the array must hold the program plus one workgroup's area, else
 * VQA_EUNSUPPORTED with the sizes in the message (14 qubits: up to about 120 diagonal stages).
 * The sweep launches as many workgroups as the array has areas for, so large circuits run on a
 * part of the GPU: this entry point trades speed for the guarantee.  All calls on one GPU share
 * that GPU's array: calls that overlap in time on one GPU race; order them on one stream.  On the
 * host the call holds the same per-device lock as every other entry point while it queues its
 * work (a static mutex: taking it allocates nothing), so host threads cannot interleave their
 * launches; the lock does not order what runs on the GPU.
 */
#define VQA_NOALLOC_SCRATCH_BYTES ((size_t)32 << 20)
int vqa_psr_full_noalloc_dev(vqa_circuit* c, int device, void* stream,
                             const double* d_theta, const double* d_input_angles, int B,
                             const int32_t* d_sel, int n_sel, double shift,
                             const vqa_obs* const* obs, int n_obs, int n_marg_bits,
                             double* d_out_expect, double* d_out_marginal);

/*
 * vqa_psr_full_dev for a SUBSET of the parameters: param_mask is a HOST array of P bytes
 * (NULL = all); only the unshifted row and the rows k = 2i+1, 2i+2 of parameters with
 * param_mask[i] != 0 are guaranteed to be written, the others may be left untouched.  For
 * sharding the shifted evaluations of ONE circuit over several GPUs (a VQE energy gradient:
 * the loop over parameters of the reference's vqe optimizers' targets,
 * vqa/VQA_optimizers/vqe_optimizers.py:207-242, 637), every rank asking for its own
 * parameters.  With a mask the call waits for its work on `stream` before it returns.
 */
int vqa_psr_params_dev(vqa_circuit* c, int device, void* stream,
                       const double* d_theta, const double* d_input_angles, int B,
                       const int32_t* d_sel, int n_sel, double shift, const uint8_t* param_mask,
                       const vqa_obs* const* obs, int n_obs, int n_marg_bits,
                       double* d_out_expect, double* d_out_marginal);

/*
 * Which parameters cannot change the observables and the marginal of a call: out_dead[i] = 1
 * when every rotation driven by theta[i] lies outside the backward light cone of the measured
 * qubits (e.g. the last-layer rotations on qubits the observable does not touch), so that both
 * shifted values of a parameter-shift sweep equal the unshifted one.  vqa_psr_full[_dev] copies
 * those rows instead of running the circuit; this call reports the set (host only, no GPU
 * work).  Returns the number of dead parameters or a negative VQA_E* code.  No counterpart in
 * the reference, which evaluates every shifted circuit (vqa/QNN/myQNNs_base.py:1381-1394).
 */
int vqa_dead_parameters(vqa_circuit* c, const vqa_obs* const* obs, int n_obs,
                        int n_marg_bits, uint8_t* out_dead);

/*
 * How vqa_eval_batch[_dev] / vqa_psr_full[_dev] / vqa_psr_params_dev would deal the Pauli terms of these
 * observables (and the marginal) into light-cone groups, each evaluated on the circuit reduced to the
 * group's backward light cone (see vqa_psr_full): returns the number of groups, 0 when the call runs on
 * the whole register, or a negative VQA_E* code; the groups' register sizes go to
 * out_qubits[0 .. min(groups, max_out)) when out_qubits is not NULL.  sweep != 0: n_instances samples
 * of a full parameter-shift sweep, else n_instances circuit evaluations of an explicit list (the choice
 * weighs the groups' launches against the work, which grows with the instances).  Host only, no GPU
 * work; the plan is cached on the circuit and reused by the calls.  No counterpart in the reference,
 * which runs every circuit on the whole register (vqa/QNN/myQNNs_base.py:692-704).
 */
int vqa_light_cone_groups(vqa_circuit* c, const vqa_obs* const* obs, int n_obs, int n_marg_bits,
                          int sweep, int64_t n_instances, int32_t* out_qubits, int max_out);

/*
 * Shot sampling.  For each of E rows draws counts[e, 0..n_outcomes) ~ Multinomial(shots[e],
 * probs[e, :]) (row-major; a row of probs is used as given, its last entry takes what the
 * others leave; negative or NaN entries count as 0).  With probs = the marginal of the
 * lowest one or two qubits (out_marginal above) this replaces state.sampling(n_shots)
 * followed by `samples % 2` / `samples % 4` and the bincount
 * (vqa/QNN/myQNNs_base.py:725-727, 749-765; vqa/QNN/myQNNs_classifier.py:559, 613-621,
 * 640-649): the reference keeps nothing else of its samples.  The stream of random numbers
 * is Philox-4x32-10 keyed by `seed`; row e of a call draws from counter (e, stream_id), so
 * results do not depend on the launch shape, and calls with different stream_id are
 * independent.  qulacs' own generator cannot be reproduced: agreement with the reference
 * is in distribution.  _dev: device pointers, queued on `stream`, not waited for.
 */
int vqa_sample_counts(int device, const double* probs, int n_outcomes,
                      const int64_t* shots, int64_t E, uint64_t seed, uint64_t stream_id,
                      int64_t* counts);
int vqa_sample_counts_dev(int device, void* stream, const double* d_probs, int n_outcomes,
                          const int64_t* d_shots, int64_t E, uint64_t seed, uint64_t stream_id,
                          int64_t* d_counts);
/* The same for one shard of a list of rows that several GPUs share: row e of the call draws from the
 * counter of row e + row_offset, so a row's counts do not depend on how the list was split. */
int vqa_sample_counts_shard_dev(int device, void* stream, const double* d_probs, int n_outcomes,
                                const int64_t* d_shots, int64_t E, uint64_t seed, uint64_t stream_id,
                                int64_t* d_counts, int64_t row_offset);

/*
 * Estimator epilogue of the regressor's gradient oracles, on the device.  From the matrix of exact
 * expectation values of a sweep, d_expect[b * (2P+1) + k] (out_expect of vqa_psr_full_dev with one
 * observable), the labels d_y[B] and the scale parameter w it computes, per gradient component
 * c = 0..P (P = the scale parameter), sums over the B samples of the per-sample terms the
 * reference reduces in NumPy after its Python loop nest:
 *   out_sums[c * 8 + 0]  the gradient term: (w M0 - y) w (M+ - M-); c = P: 2 (-y M0 + w M0^2)
 *                         (vqa/QNN/myQNNs_base.py:1378, 1393)
 *   out_sums[c * 8 + 1]  its square
 * and with want_variances (MSE_iEval, 1421-1596) the estimator polynomials of 1205-1316:
 *   [2], [3]  conditional-variance estimate, shifted and unshifted part (c = P: [2] only)
 *   [4], [5]  the forms taken when a batch mean comes out negative (1242-1248, 1274-1286)
 *   [6]       second moment of the gradient term (1313)            ([7] unused)
 * M are estimates of the exact values from n = floor(shots[c] * weight[b]) shots (weight NULL: 1;
 * the unshifted evaluation uses n_shots0):  mode 0 exact values (min_approx_shots = -10), 1 normal
 * approximation E + z sqrt((1 - E^2) / n) (892-905), 2 binomial finite-shot means (715-738), 3 the
 * caller's d_estimates[B, 2P+1].  Unbiased powers as 843-847.  Random numbers: Philox keyed by
 * `seed`, one stream per (sample, column); not the reference's NumPy sequence.
 * d_workspace: vqa_mse_stats_workspace(B, P) bytes of device memory, the caller's.  Optional
 * outputs: d_out_m0[B] (estimates of the unshifted value), d_out_grad_data[(P+1) * B] (the
 * per-sample gradient terms, 'grad_data_list' of MSE_iEval).  Deterministic (fixed summation
 * order); queued on `stream`, not waited for.
 */
int64_t vqa_mse_stats_workspace(int64_t B, int P);
int vqa_mse_stats_dev(int device, void* stream, const double* d_expect, const double* d_estimates,
                      int64_t B, int P, const double* d_y, double w, const double* d_shots,
                      const double* d_weight, double n_shots0, int mode, uint64_t seed,
                      int want_variances, double n0, int second_order, double* d_workspace,
                      double* d_out_sums, double* d_out_m0, double* d_out_grad_data);
/*
 * The same for ONE SHARD of a mini-batch that several GPUs share (one process per GPU, contiguous
 * blocks of samples): the B rows given are rows row_offset .. row_offset + B of the whole batch, and
 * the random streams are named by the row's number in the whole batch, so the estimates a sample
 * gets do not depend on how many ranks share the work.  The ranks add their out_sums (one
 * all-reduce of (P+1) * 8 doubles) and finish as one process would.  The sample loop this shards:
 * vqa/QNN/myQNNs_base.py:1114-1124 under 1370-1394 / 1469-1586.
 */
int vqa_mse_stats_shard_dev(int device, void* stream, const double* d_expect, const double* d_estimates,
                            int64_t B, int P, const double* d_y, double w, const double* d_shots,
                            const double* d_weight, double n_shots0, int mode, uint64_t seed,
                            int want_variances, double n0, int second_order, double* d_workspace,
                            double* d_out_sums, double* d_out_m0, double* d_out_grad_data,
                            int64_t row_offset);

/*
 * Gradient and variance estimates of the oracles with independent random shot allocation
 * (MSE_iEval_iRandom: vqa/QNN/myQNNs_base.py:1619-1666; classifier, partial MSE:
 * vqa/QNN/myQNNs_classifier.py:1017-1053), from the exact values of a ragged instance list.
 * A row r is one (gradient component, sample that received shots); rows of component i are
 * seg[i] .. seg[i+1]; d_expect[(3 r + v) * n_obs + col] holds the exact value of the row's
 * unshifted (v = 0) and shifted (v = 1: +pi/2, 2: -pi/2) circuit as computed by
 * vqa_eval_batch_dev on the list in that order.  kind 0: <Z0> (col 0), labels d_y[r], scale w;
 * with has_scale the last component is the scale parameter, whose rows use twice d_shots[r]
 * outcomes of the unshifted circuit (1642-1653).  kind 1: class probabilities, col = d_label[r].
 * The reference keeps the raw outcomes of d_shots[r] shots of each circuit; the kernel draws the
 * counts of their joint outcomes instead (same distribution; Philox keyed by seed, one stream
 * per row) and writes d_out[2 i] = the gradient estimate (1652, 1662; classifier 1049) and
 * d_out[2 i + 1] = the sample variance of the per-shot products (1599-1617; classifier 1002-1015).
 * d_n_list[i]: the component's total shots; M_tot: number of training samples.
 */
int vqa_irandom_stats_dev(int device, void* stream, const double* d_expect, int n_obs,
                          const int32_t* d_label, const double* d_y, const int64_t* d_shots,
                          const int32_t* d_seg, const double* d_n_list, int n_components, int kind,
                          int has_scale, double w, double M_tot, uint64_t seed, double* d_out);

/*
 * Classifier MSE gradient from drawn class counts (vqa/QNN/myQNNs_classifier.py:1055-1156,
 * MSE_grad_eval): d_counts[(b * (2P+1) + k) * 3 + c] = how often class c was read for sample b
 * in instance k of its parameter-shift sweep (k = 0 unshifted, 1 + 2 i / 2 + 2 i = parameter i
 * shifted up / down), d_shots[b * (2P+1) + k] = the shots of that instance.  With q = counts /
 * shots:  d_label given (partial MSE):  d_grad[i] = mean_b (q0[label_b] - 1) (q+[label_b] - q-[label_b]);
 * d_onehot given ([n_samples, 3], full MSE):  d_grad[i] = mean_b sum_c (q0[c] - onehot_b[c]) (q+[c] - q-[c]).
 * Exactly one of d_label and d_onehot is not NULL.  Device pointers; deterministic summation order.
 */
int vqa_class_grad_dev(int device, void* stream, const int64_t* d_counts, const int64_t* d_shots,
                       int64_t n_samples, int n_params, const int32_t* d_label, const double* d_onehot,
                       double* d_grad);

/*
 * The classifier's gradient oracles with variance estimates, from drawn class counts: per circuit
 * parameter i, out_sums[4 i + 0] = sum over the samples of the per-sample gradient term, [1] = sum of
 * the conditional-variance estimate, [2] = sum of the form the reference takes when the batch mean of
 * [1] comes out negative ([3] unused).  kind 0: MSE on the true class alone (partial_MSE), 1: MSE over
 * the three classes -- MSE_iEval, vqa/QNN/myQNNs_classifier.py:888-999 with
 * _estimate_MSEgrad_mean_condVar, 861-886 (without want_variances: MSE_grad_eval, 1055-1156) --
 * 2: cross entropy from the success counts -- CE_grad_eval / CE_iEval, 1172-1269 with
 * _estimate_CEgrad_mean_condVar, 1158-1169 (the L2 term lambda * theta_i is the caller's).  Counts and
 * shots as for vqa_class_grad_dev; labels[n_samples] are the class indices 0..2.  Sums, not means:
 * the ranks of a batch split over several GPUs add them and divide by the whole batch size.
 * d_out_grad_data: NULL or [n_params, n_samples], the per-sample gradient terms ('grad_data_list').
 * Deterministic; queued on `stream`, not waited for.
 */
int vqa_class_stats_dev(int device, void* stream, const int64_t* d_counts, const int64_t* d_shots,
                        int64_t n_samples, int n_params, const int32_t* d_label, int kind,
                        int want_variances, double* d_out_sums, double* d_out_grad_data);

/*
 * Final state of ONE instance, interleaved (re, im) doubles, 2 * 2^n values,
 * host pointer.  For plumbing tests of the per-sample circuit.run(x) shim
 * (vqa/QNN/myQNNs_base.py:913); not a hot-path call.
 */
int vqa_statevector(vqa_circuit* c, int device,
                    const double* theta, const double* input_row,
                    int shift_param, double shift_val, double* out_state);

/*
 * Engine selection for circuits created afterwards on this thread's process:
 * 0 = automatic, 1 = LDS gate-by-gate engine (n <= 13), 2 = register-blocked
 * fused engine (4 <= n <= 14), 3 = HBM streaming engine (any n; tiled passes from
 * 13 qubits on), 4 = HBM streaming engine with one sweep per gate.  For tests
 * and benchmarks that compare engines; not part of the reference interface.
 */
int vqa_set_engine(int engine);

/*
 * Measurement aid of the HBM streaming engine, not part of the reference interface: how many passes
 * over a whole state the library has launched in this process since the counters were last reset --
 * those that read and write a state, and those that only read one (measuring passes).  A pass moves
 * 16 * 2^n bytes each way; benchmarks price a call's HBM traffic with these counts.
 */
int vqa_state_pass_counters(int64_t* read_write, int64_t* read_only, int reset);

#ifdef __cplusplus
}
#endif
#endif /* VQA_HIP_H */
