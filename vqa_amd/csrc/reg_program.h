// Program format of engine 2 (register-blocked fused engine), shared by the
// host-side circuit compiler (reg_compile.cpp) and the kernels (kern_reg.hip).
//
// A circuit is brought into a normal form of alternating stages:
//   U-stage: at most one 2x2 unitary per qubit (the product of the elementary
//            1-qubit gates the submitted list applies to that qubit there);
//   D-stage: a diagonal operator: single-qubit Z-phases, two-qubit ZZ-phases
//            (from CNOT-RZ-CNOT sandwiches), CZ and Z signs.
// Per evaluation every 2x2 is re-expressed on the device as
//   e^{i d} RZ(alpha) RY(beta) RZ(gamma),
// so the dense part applied to the amplitudes is a REAL rotation and the phases
// e^{i alpha/2}, e^{i gamma/2} are merged into the neighbouring D-stages.  The
// global phase is dropped.  On register bits the rotation by phi = beta/2 is
// three in-place shears (a0 += u a1; a1 -= s a0; a0 += u a1 with u = tan(phi/2)
// <= 1, s = sin(phi)); on lane bits, where each lane holds one amplitude of the
// pair, it is ONE fused multiply-add per amplitude with the common factor
// 1/cos or 1/sin left out, so states are carried unnormalised and the
// measurement divides by the norm.
//
// Amplitude index bits are "positions": 0..R-1 live in registers (2^R
// amplitudes per thread), R..R+L-1 are lane bits, the rest are wave bits.  A
// layout maps positions to qubits; transpositions through LDS change it.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/vqa_hip.h"

namespace vqa {

constexpr int kRegBits = 4;       // R: register-resident index bits
constexpr int kRegAmps = 1 << kRegBits;
constexpr int kMaxPos = 16;       // positions a layout can describe
constexpr int kMaxThreadBits = 10;
constexpr int kSignWords = (1 << kMaxThreadBits) / 32;

enum RegOpKind : int32_t {
    ROP_END = 0,
    ROP_UREG = 1,    // a = register bit j,  b = U-entry index
    ROP_ULANE = 2,   // a = lane bit l,      b = U-entry index
    ROP_DIAG = 3,    // a = D-descriptor index
    ROP_XPOSE = 4,   // a = transposition descriptor index
};

struct RegOp {
    int32_t kind, a, b, c;
};

// The kernels run PHASES: one loop iteration executes, in this fixed order and
// each only when its flag is set, a transposition, a rotation on each register
// bit, a rotation on each lane bit and a diagonal stage.  Every block updates
// the amplitudes in place behind a uniform branch, which is what lets the
// register allocator keep the state in one set of registers across the loop.
constexpr int32_t PH_XPOSE = 1 << 0;
constexpr int32_t PH_UREG0 = 1 << 1;    // .. PH_UREG0 << 3
constexpr int32_t PH_ULANE0 = 1 << 5;   // .. PH_ULANE0 << 5
constexpr int32_t PH_DIAG = 1 << 11;
constexpr int32_t PH_END = 1 << 12;
struct PhaseOp {
    int32_t flags;
    int32_t xdesc;    // transposition descriptor (PH_XPOSE)
    int32_t ddesc;    // D-stage descriptor (PH_DIAG)
    int32_t ent[10];  // U-entry of register bits 0..3, lane bits 0..5 (0 when unused)
    int32_t pad[3];
};

// an elementary 1-qubit gate inside a U-entry, or an angle term of a D-entry
struct ElemGate {
    int32_t kind;  // VQA_GATE_*
    int32_t src;   // VQA_SRC_*
    int32_t idx;
    int32_t pad;
    double coef, offset;
};

// one (U-stage, qubit): product of elems[begin, begin+count) in time order
struct UEntry {
    int32_t begin, count;
    int32_t d_before;  // ph1 slot that absorbs e^{i gamma/2} (-1: dropped)
    int32_t d_after;   // ph1 slot that absorbs e^{i alpha/2} (-1: dropped)
    int32_t parity;    // U-stage index & 1 (the two preparation phases)
    int32_t on_lane;   // 1: applied while the qubit sits on a lane bit, 0: on a register bit
    int32_t input_free;  // 1: no gate of the entry takes its angle from the input row (same 2x2 for every sample)
    int32_t pad;
};

// one single-qubit phase slot of a D-stage: exp(+i/2 * sum of angle terms) on
// |0>, conj on |1>.  Slots of a stage are ordered by POSITION in the layout the
// stage is applied in: slot = stage * n + position.
struct D1Entry {
    int32_t begin, count;  // into terms[]
};
// one ZZ phase of a D-stage: exp(+i/2 * sum) when the two bits agree, conj otherwise
struct D2Entry {
    int32_t begin, count;
};

struct PairRef {
    int32_t a, b;   // positions (meaning depends on the list it is in)
    int32_t ph2;    // index into the ph2 table
    int32_t pad;
};

// a D-stage as applied in the layout current at that point of the program
struct DDesc {
    int32_t ph1_base;                 // first ph1 slot of this stage (n consecutive slots, by position)
    int32_t tt_begin, tt_count;       // pairs with both qubits on thread bits: a, b = thread bit numbers
    int32_t rt_begin, rt_count;       // a = register bit, b = thread bit number
    int32_t rr_begin, rr_count;       // a, b = register bits
    int32_t sign_id;                  // index of the stage's CZ / Z sign table, -1: no signs
    int32_t pc;                       // index of the phase that applies it
    int32_t pad[3];
};

// CZ / Z signs of a D-stage in a given layout: sign(r, t) = tt_par[t] ^
// XOR_j (r_j & cross[j][t]) ^ rr_par[r].  Identical tables are shared.
struct SignTab {
    uint32_t tt_par[kSignWords];           // bit t: parity from thread-only terms
    uint32_t cross[kRegBits][kSignWords];  // bit t: parity flip when register bit j is set
    uint32_t rr_par;                       // bit r: parity from register-only terms
    int32_t pad[3];
};

// LDS transposition: the slot function is XOR-linear; wconst[p] = byte offset
// contributed by old-layout position bit p, rconst[p] = byte offset of the old
// position that new position p reads.
struct XDesc {
    int32_t wconst[kMaxPos];
    int32_t rconst[kMaxPos];
};

// What has to be recomputed when one trainable parameter is shifted
// (parameter-shift sweeps recompute only this and resume from a checkpoint).
struct PatchRef {
    int32_t kind;     // 0: U-entry `index`; 1: ph1 slot `index`; 2: ph2 entry `index`
    int32_t index;
    double coef_sum;  // kinds 1, 2: sum of the coefficients the parameter enters the slot with;
                      // kind 0: coefficient of the one gate of the entry it enters, 0 when several
};
struct ParamInfo {
    int32_t ref_begin, ref_count;  // into refs[]
    int32_t resume_diag;           // first D-stage whose tables change; -1: restart from |0...0>
    int32_t resume_pc;             // phase to resume at, starting with its diagonal stage (0 when restarting)
    // A parameter that drives one rotation of ONE fused 2x2 (U-stage u, D-stage u behind it) can also resume
    // BEHIND that U-stage: with M the unshifted 2x2 and M' the one with the generator inserted, the state
    // behind the stage is M' M^-1 applied to the unshifted one there (the other qubits' 2x2s commute with it),
    // so the run loads the checkpoint in front of D-stage u, applies that one-qubit correction and goes on
    // with the unshifted tables: one U-stage and one D-stage less per run, nothing to patch.
    int32_t late_diag;             // that D-stage, -1: not available (several references, last U-stage)
    int32_t late_pc;               // the phase that applies it
    int32_t late_mask;             // 1 << (position of the qubit in the layout of that D-stage)
    int32_t pad;
};

// One run of a parameter-shift sweep besides the unshifted one.  variant 0: the
// parameter's generator is inserted in front of its rotation and both shifted
// values come out of this one run; 1 / 2: the circuit at +shift / -shift.
struct RunInfo {
    int32_t param, variant;
};

// What a sweep workgroup needs for one run, worked out per sample (device side,
// 8 x 16 bytes): the patched table entries, the measurement mode and where to resume.
struct RunRec {
    double m[2];      // new rotation data of the fused 2x2 (table slot ids[0]);  late resumption: C00
    double fb[2];     // factor for the phase slot ids[1] (in front of the 2x2, or the parameter's own slot);  late: C01
    double fa[2];     // factor for the phase slot ids[2] (behind the 2x2);  late: C10
    double rho[2];    // pair: ratio of the unit factors left out of the two runs
    double cs[2];     // pair: cos, sin of half the shift of the rotation angle
    int32_t ids[4];   // table slots (index into the workgroup's tables, -1: none); ids[3]: late resumption: ParamInfo::late_mask, else 0
    int32_t out[4];   // mode (0: one value, 1: pair), output instance k, resume diagonal stage, resume phase
    double c11[2];    // late resumption: C11 (C = the one-qubit correction, in the frame of the checkpoint)
};

struct RegHeader {
    int32_t n, T, L, W;          // qubits, thread bits, lane bits, wave bits
    int32_t block;               // threads per workgroup
    int32_t S;                   // states per workgroup (explicit-list kernel)
    int32_t n_uent, n_d1, n_d2;  // per-evaluation table sizes (entries of 16 bytes)
    int32_t n_elems, n_terms, n_ops, n_ddesc, n_xdesc, n_pairs, n_sign, n_params, n_refs, n_phases;
    int32_t n_runs;
    int32_t has_xpose;
    int32_t n_dep;               // fused 2x2s that depend on the input row (dep[] lists them)
    int8_t final_qubit_at_pos[kMaxPos];
    // byte offsets of the arrays inside the device blob
    int64_t off_elems, off_uent, off_d1, off_d2, off_terms, off_ops, off_ddesc, off_xdesc, off_pairs,
        off_signs, off_params, off_refs, off_phases, off_runs, off_dep;
    int64_t total_bytes;
};

// host-side result of compiling a circuit for engine 2
struct RegProgram {
    RegHeader hdr;
    std::vector<ElemGate> elems;
    std::vector<UEntry> uent;
    std::vector<D1Entry> d1;
    std::vector<D2Entry> d2;
    std::vector<ElemGate> terms;
    std::vector<RegOp> ops;        // host-side op stream the phases are packed from
    std::vector<PhaseOp> phases;
    std::vector<DDesc> ddesc;
    std::vector<XDesc> xdesc;
    std::vector<PairRef> pairs;
    std::vector<SignTab> signs;
    std::vector<ParamInfo> params;
    std::vector<PatchRef> refs;
    std::vector<RunInfo> runs;
    std::vector<int32_t> dep;      // the U-entries with input_free == 0
    std::vector<char> blob;  // header + arrays, as uploaded
    size_t lds_bytes = 0;      // explicit-list kernel (S states per workgroup)
    size_t lds_bytes_psr = 0;  // parameter-shift kernel (one state per workgroup)
    size_t lds_bytes_ftab = 0; // ... of which the diagonal stages' thread-factor tables (the last part of the area)
    int n_dense_gates = 0, n_xpose = 0, n_diag = 0;
};

// Returns false when the circuit is outside what engine 2 handles (n <
// kRegBits, too many thread bits, tables that do not fit LDS); `why` says so.
bool reg_compile(int n_qubits, int n_theta, const std::vector<vqa_op>& ops, RegProgram* out,
                 std::string* why);

}  // namespace vqa
