// Internal declarations shared by the C-ABI layer and the three engines of
// libvqahip.so.  Not installed; the public interface is include/vqa_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vqa_hip.h"

namespace vqa {

// ---- errors ----------------------------------------------------------------
void set_error(const char* fmt, ...);
#define VQA_FAIL(code, ...)            \
    do {                               \
        ::vqa::set_error(__VA_ARGS__); \
        return (code);                 \
    } while (0)
#define VQA_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t _e = (call);                                                        \
        if (_e != hipSuccess)                                                          \
            VQA_FAIL(VQA_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), \
                     __FILE__, __LINE__);                                              \
    } while (0)

// ---- which instance does what ----------------------------------------------
// Either an explicit list (sample_idx / shift_param / shift_val, any of which
// may be null) or the implicit full parameter-shift sweep: instance e is
// sample sel[e / K] (or e / K), K = 2P+1, k = e % K: 0 = unshifted,
// 2i+1 = +shift on parameter i, 2i+2 = -shift.
struct EvalSpec {
    const int32_t* sample_idx;
    const int32_t* shift_param;
    const double* shift_val;
    const int32_t* sel;
    int psr;
    int P;
    double shift;
    int B;        // rows of input_angles (sample indices are checked against it on the device)
    int n_theta;  // parameters (shift_param is checked against it)
};

// Returns false (and a harmless b = 0, no shift) when the instance names a
// sample or parameter that does not exist; the caller then writes NaN for it.
__device__ __forceinline__ bool decode_eval(const EvalSpec& sp, int64_t e, int& b, int& sparam,
                                            double& sval) {
    if (sp.psr) {
        const int K = 2 * sp.P + 1;
        const int64_t j = e / K;
        const int k = (int)(e - j * K);
        b = sp.sel ? sp.sel[j] : (int)j;
        if (k == 0) {
            sparam = -1;
            sval = 0.0;
        } else {
            sparam = (k - 1) >> 1;
            sval = (k & 1) ? sp.shift : -sp.shift;
        }
    } else {
        b = sp.sample_idx ? sp.sample_idx[e] : (int)e;
        sparam = sp.shift_param ? sp.shift_param[e] : -1;
        sval = sp.shift_val ? sp.shift_val[e] : 0.0;
    }
    const bool ok = (sp.B == 0 || (unsigned)b < (unsigned)sp.B) && sparam >= -1 && sparam < sp.n_theta;
    if (!ok) {
        b = 0;
        sparam = -1;
        sval = 0.0;
    }
    return ok;
}

__device__ __forceinline__ double op_angle(const vqa_op& op, const double* theta,
                                           const double* in_row, int sparam, double sval) {
    double v = 0.0;
    if (op.src == VQA_SRC_THETA) {
        v = theta[op.idx];
        if (op.idx == sparam) v += sval;
    } else if (op.src == VQA_SRC_INPUT) {
        v = in_row[op.idx];
    }
    return op.coef * v + op.offset;
}

// ---- observables, passed to kernels by value --------------------------------
constexpr int kMaxObs = 64;
constexpr int kMaxTerms = 96;
struct ObsPack {
    int n_obs;
    int n_terms;
    int off[kMaxObs + 1];
    double coef[kMaxTerms];
    uint64_t z[kMaxTerms];
    uint64_t x[kMaxTerms];
};

// ---- handles ------------------------------------------------------------------
struct DeviceCtx {  // per (circuit, device)
    vqa_op* d_ops = nullptr;
    void* reg_prog = nullptr;  // engine-2 compiled program (device copy)
};

}  // namespace vqa

struct vqa_obs {
    int n_qubits;
    std::vector<double> coef;
    std::vector<uint64_t> z, x;
};

#include "reg_program.h"

namespace vqa {
// Which parameters cannot change the measured quantities of a call (their rotations lie outside
// the backward light cone of the observables' qubits): both shifted values equal the unshifted
// one, so a parameter-shift sweep copies it instead of running the circuit.  Per (circuit,
// measured qubits), with the engines' derived lists (built on first use, under the circuit's mutex).
struct ConeInfo {
    std::vector<uint8_t> dead;               // [n_theta]
    std::vector<uint8_t> gate_dead;          // [n_ops]: gates outside the cone (they cannot change the measured values either)
    int n_dead = 0, n_gate_dead = 0;
    // the circuit without its dead gates, compiled for the sweeps of this cone
    bool reg_tried = false;
    std::shared_ptr<RegProgram> reg;         // engine 2 (null: use the circuit's own program)
    std::map<int, void*> reg_dev;            // its device copies
    std::shared_ptr<void> hbm_plan;          // engine 3
    std::vector<int32_t> dead_list;          // the dead parameters, ascending
    bool reg_built = false;
    std::vector<int32_t> reg_live_runs;      // engine 2: runs of the live parameters, in run order
    bool reg_pairs_built = false;
    int reg_pairs = 0;                       // ... how many pairs the list below holds
    std::vector<int32_t> reg_live_items;     // ... the same runs as items of two (the second -1: a run on its own)
    bool group_built = false;
    std::vector<int32_t> group_live_k;       // engine 2, small circuits: runs of the live parameters, pair runs first
    bool group_late_built = false;
    std::vector<int32_t> group_late_k;       // engine 2, grouped sweeps with late resumption: one run per live parameter, by U-stage (empty: not available)
    // engine 2: device copies of a sweep's default run list and dead list, [runs, padded to 16 bytes | dead parameters],
    // by (device, kind of list); uploaded when first used, freed with the circuit (cone_lists_dev)
    std::map<std::pair<int, int>, void*> list_dev;
    bool hbm_built = false;
    std::vector<int32_t> hbm_ord, hbm_ord_pass, hbm_ord_sweep, hbm_unused;   // engine 3: the sweep plan's lists without the dead
    std::vector<double> hbm_coef;
};
}  // namespace vqa

struct vqa_circuit {
    int n_qubits, n_ops, n_theta, n_input;
    int engine;  // resolved engine: 1 LDS, 2 register-blocked, 3 HBM
    bool auto_engine;  // the engine was chosen by the library (vqa_set_engine(0) at creation)
    std::vector<vqa_op> ops;
    std::shared_ptr<vqa::RegProgram> reg;  // host copy of the engine-2 program (may be null)
    std::shared_ptr<void> hbm_plan;        // engine 3: the gate list cut into tiled passes (built on first use)
    int hbm_tiles = 0;                     // engine 3: the build whose plans this circuit holds (11 or 12; 0: not chosen yet)
    std::map<uint64_t, std::shared_ptr<void>> hbm_hplans;   // engine 3: Hadamard layers used to measure X-type terms, by qubit mask
    std::map<std::pair<uint64_t, uint64_t>, std::shared_ptr<vqa::ConeInfo>> cones;   // by (Z-type, general) measured qubits
    std::map<std::vector<uint64_t>, std::shared_ptr<void>> splits;   // light-cone splits of a call's terms, by the terms (vqa_abi.hip)
    std::mutex mu;
    std::map<int, vqa::DeviceCtx> dev;
};

namespace vqa {

struct EvalCall {  // one batched evaluation, all pointers on the device
    vqa_circuit* c;
    int device;
    hipStream_t stream;
    const vqa_op* d_ops;
    const double* d_theta;
    const double* d_input;
    EvalSpec spec;
    int64_t E;
    ObsPack obs;
    int n_marg_bits;
    int n_sweep;  // > 0: the instance list is a full parameter-shift sweep over this many samples
    const uint8_t* param_mask;  // sweeps: HOST array [n_theta], 0 = this parameter's two rows are not wanted (null: all)
    ConeInfo* cone;  // what cannot matter for this call's observables (null: the state itself is asked for)
    double* d_out_expect;
    double* d_out_marginal;
    struct FixedScratch* fixed = nullptr;  // non-null: the call must not allocate; its working memory is this region
    int ws_slice = 0;  // > 0: a group of a split call on side stream ws_slice - 1, with that stream's own workspace
};

// A fixed region of device memory that serves as a call's working memory (the allocation-free
// entry point): handed out front to back, 256-byte aligned.
struct FixedScratch {
    char* base;
    size_t bytes, used;
};

// engine 3: passes over a state launched since the process started (or the counters were last reset), those that read
// and write a state and those that only read one (the measuring passes); see vqa_state_pass_counters
void count_state_passes(int64_t read_write, int64_t read_only);

// engines: each returns VQA_OK or an error code with set_error() called
int lds_engine_run(const EvalCall& call);
int lds_engine_max_qubits();
int reg_engine_run(const EvalCall& call, const void* d_prog, const RegProgram& prog);
bool reg_engine_can_sweep(const RegProgram& prog);
int reg_engine_sweep(const EvalCall& call, const void* d_prog, const RegProgram& prog, int n_samples);
bool reg_engine_can_group(const RegProgram& prog);
int reg_engine_group(const EvalCall& call, const void* d_prog, const RegProgram& prog, int n_samples);
// engine 3 (kern_hbm.hip, built twice: 12-qubit tiles with four amplitude bits per thread, _t12, and
// 11-qubit tiles with three, _t11, which keeps four workgroups on a CU instead of two and is 6-15 % faster
// up to 24 qubits); these four choose the build by the register size (vqa_abi.hip)
int hbm_engine_run(const EvalCall& call);
bool hbm_engine_can_sweep(const EvalCall& call);
int hbm_engine_sweep(const EvalCall& call, int n_samples);
int hbm_engine_statevector(const EvalCall& call, double* d_out_state);
int hbm_engine_run_t12(const EvalCall& call);
bool hbm_engine_can_sweep_t12(const EvalCall& call);
int hbm_engine_sweep_t12(const EvalCall& call, int n_samples);
int hbm_engine_statevector_t12(const EvalCall& call, double* d_out_state);
int hbm_engine_run_t11(const EvalCall& call);
bool hbm_engine_can_sweep_t11(const EvalCall& call);
int hbm_engine_sweep_t11(const EvalCall& call, int n_samples);
int hbm_engine_statevector_t11(const EvalCall& call, double* d_out_state);

// rows 2i+1, 2i+2 of every sample <- its row 0, for the parameters i of d_list (kern_sample.hip)
int fill_dead_rows(hipStream_t stream, double* d_out, int64_t n_samples, int K, int stride, const int32_t* d_list, int n_list);

// multinomial outcome counts per row, chain of binomials (kern_sample.hip)
int sample_counts_launch(hipStream_t stream, const double* d_probs, int M, const int64_t* d_shots, int64_t E,
                         uint64_t seed, uint64_t stream_id, int64_t* d_counts, int64_t row_offset = 0);

// estimator epilogue of the regressor's oracles (kern_stats.hip)
size_t mse_stats_workspace_bytes(int64_t B, int P);
int mse_stats_launch(hipStream_t stream, const double* d_E, const double* d_est, int64_t B, int P, const double* d_y,
                     double w, const double* d_shots, const double* d_weight, double n_shots0, int mode, uint64_t seed,
                     int want_var, double n0, int second_order, double* d_workspace, double* d_out, double* d_m0,
                     double* d_gd, int64_t row_offset);

// the classifier's gradient terms and conditional-variance estimates from class counts (kern_stats.hip)
int class_stats_launch(hipStream_t stream, const int64_t* d_counts, const int64_t* d_shots, int64_t B, int P,
                       const int32_t* d_label, int kind, int want_var, double* d_out, double* d_gd);
int class_grad_launch(hipStream_t stream, const int64_t* d_counts, const int64_t* d_shots, int64_t B, int P,
                      const int32_t* d_label, const double* d_y1hot, double* d_grad);
int irandom_stats_launch(hipStream_t stream, const double* d_expect, int n_obs, const int32_t* d_label, const double* d_y,
                         const int64_t* d_shots, const int32_t* d_seg, const double* d_n_list, int C, int kind,
                         int has_scale, double w, double M_tot, uint64_t seed, double* d_out);

// Working memory of a call: the per-device workspace owned by the library (grown on demand, never
// shrunk; users on different streams are ordered by an event) or, for a call with a fixed region,
// the next `bytes` of that region (VQA_EUNSUPPORTED when it does not hold them).
int call_workspace(const EvalCall& call, size_t bytes, void** out);
// bytes a call may still ask for: SIZE_MAX for the growable workspace
size_t call_workspace_room(const EvalCall& call);
// bytes of the growable workspace the library holds on the call's device right now (0 for a fixed region)
size_t call_workspace_held(const EvalCall& call);
// host -> device copy of a small table on the call's stream: hipMemcpyAsync, or, for a call that must
// not allocate (pageable copies stage through buffers the runtime may allocate), chunks passed
// as kernel arguments to a copy kernel.  d_dst 16-byte aligned, with room for bytes rounded up to 16.
int call_upload(const EvalCall& call, void* d_dst, const void* src, size_t bytes);
// the second form on its own: the source has been read when the function returns, whatever memory it is
// in (a function-local table that must not outlive the call).  Same alignment and room as above.
int upload_by_value(hipStream_t stream, void* d_dst, const void* src, size_t bytes);
// Engine 2's sweeps read their run list and their dead list from the device.  The default lists of a cone (no caller's
// parameter subset) are uploaded once per device and kept: *d_live / *d_dead point into that copy.  kind tells the
// lists of the different sweep kernels apart (0: one run per workgroup, 1: items of two runs, 2: grouped sweeps,
// 3: grouped sweeps with late resumption).  Not for calls that must not allocate.
int cone_lists_dev(const EvalCall& call, int kind, const std::vector<int32_t>& live, const std::vector<int32_t>& dead,
                   const int32_t** d_live, const int32_t** d_dead);
// A work counter on the call's device that is never cleared: a sweep kernel's workgroups draw tickets from it with
// atomic increments and subtract *base, the count before this launch; `draws` is what the launch will add (every
// workgroup draws until a ticket lies past the end: items + workgroups), which the host adds to its own copy here.
// One counter per workspace slice, whose users the device runs in the order they were queued.  Null for a call that
// must not allocate (it clears a counter in its own region).  call_ticket_reset: after a launch that failed.
unsigned* call_ticket_counter(const EvalCall& call, unsigned draws, unsigned* base);
void call_ticket_reset(const EvalCall& call);
// Up to three more streams of the library's own on the call's device, for a call that spreads independent
// work over several hardware queues, with an event each and one event to start them from.  The call makes
// them wait for *fork (recorded on its own stream) before their first kernel and its own stream wait for
// done[i] (recorded behind their last kernel) before it returns, so that the call's stream still orders the
// whole call.  Returns how many it got (0 for a call that must not allocate).
int call_side_streams(const EvalCall& call, int want, hipStream_t* streams, hipEvent_t* done, hipEvent_t* fork);

}  // namespace vqa
