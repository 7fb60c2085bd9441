// Engine 3: HBM-streaming statevector kernels (any n; used for n >= 14).
//
// States live in a library-owned HBM workspace, C instances at a time
// (C * 2^n * 16 bytes).  For n >= 13 the gate list is cut into PASSES: a pass
// names 12 "tile" qubits (the four lowest plus eight others); a workgroup
// gathers the 4096 amplitudes that differ only in those bits into LDS, applies
// every gate of the pass there (gates on the tile qubits; diagonal gates and
// controls may also sit on the other, "outer" qubits, whose bits are fixed
// for the workgroup) and writes the tile back: one read and one write of the
// state per pass instead of per gate (an IBM-HEA layer on 20 qubits is one
// pass instead of 80 gate sweeps).  Below 13 qubits, or with engine 4, every
// gate is one coalesced sweep over the chunk: dense 1-qubit gates as
// amplitude-pair sweeps, RZ / CZ / Z as in-place phase or sign sweeps, CNOT as
// a conditional pair swap.  The observables are reduced in two deterministic
// stages (per-block partials, then one finishing block per instance).
//
// Replaces, per instance, LearningCircuit.run(x) + Observable expectation
// (vqa/QNN/myQNNs_base.py:692-704, 907-914) for registers that do not fit a
// workgroup.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "vqa_core.h"

#ifndef VQA_TILE_GROUP
#define VQA_TILE_GROUP 4
#endif
#ifndef VQA_TILE_BITS
#define VQA_TILE_BITS 12
#endif

// The file is built twice with different tile shapes (Makefile): VQA_HBM_VARIANT is the suffix of this
// build's four entry points, between which hbm_engine_run / _can_sweep / _sweep / _statevector
// (vqa_abi.hip) choose by the register size.
#ifndef VQA_HBM_VARIANT
#define VQA_HBM_VARIANT _t12
#endif
#define VQA_HBM_CAT2(a, b) a##b
#define VQA_HBM_CAT(a, b) VQA_HBM_CAT2(a, b)
#define HBM_FN(name) VQA_HBM_CAT(name, VQA_HBM_VARIANT)

namespace vqa {

namespace {

constexpr int kT = 256;
constexpr int kZBatch = 32;  // Z-string terms (or marginal bins) reduced per pass

struct HbmCommon {
    int n, n_ops, n_input;
    const vqa_op* ops;
    const double* theta;
    const double* input;
    EvalSpec spec;
    int64_t e_base;  // first instance of this chunk
    int C;           // instances in this chunk
};

// Several samples of a sweep in one launch: blockIdx.z is the sample's lane, and a lane's states, tables,
// partial sums and output rows lie at fixed distances from lane 0's (all zero for a launch without lanes).
struct LaneStride {
    int64_t st;     // amplitudes between the states of consecutive lanes
    int64_t tab;    // entries between their angle tables
    int64_t part;   // doubles between their partial sums
    int64_t inst;   // bytes between their per-instance tables (of the plan the launch runs)
    int64_t rows;   // output rows (instances) between them; the sample number goes up by one per lane
};
// the lanes of the launches this thread issues: set by a sweep around a group of samples, one lane otherwise
struct LaneGrid {
    int n = 1;
    LaneStride ls = { 0, 0, 0, 0, 0 };
};
thread_local LaneGrid t_lanes;
inline dim3 lane_grid(dim3 g) {
    g.z = (unsigned)t_lanes.n;
    return g;
}
template <class T>
__device__ __forceinline__ T* lane_bytes(T* p, int64_t bytes) {
    return (T*)((char*)p + bytes * blockIdx.z);
}
template <class T>
__device__ __forceinline__ const T* lane_bytes(const T* p, int64_t bytes) {
    return (const T*)((const char*)p + bytes * blockIdx.z);
}

__global__ __launch_bounds__(kT) void k_init(double2* st, int64_t total, int64_t D, LaneStride ls) {
    st += ls.st * blockIdx.z;
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (i < total) st[i] = make_double2((i & (D - 1)) == 0 ? 1.0 : 0.0, 0.0);
}

// (cos, sin) of the half angle of every rotation gate of every instance in the chunk
__global__ __launch_bounds__(kT) void k_angles(HbmCommon h, double2* tab, LaneStride ls) {
    tab += ls.tab * blockIdx.z;
    h.e_base += ls.rows * blockIdx.z;
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (int64_t)h.C * h.n_ops) return;
    const int ci = (int)(i / h.n_ops), g = (int)(i - (int64_t)ci * h.n_ops);
    const vqa_op op = h.ops[g];
    double2 cs = make_double2(1.0, 0.0);
    if (op.kind >= VQA_GATE_RX && op.kind <= VQA_GATE_RZ) {
        int b, sparam;
        double sval;
        decode_eval(h.spec, h.e_base + ci, b, sparam, sval);
        const double ang = op_angle(op, h.theta, h.input + (size_t)b * h.n_input, sparam, sval);
        double sn, c;
        sincos(0.5 * ang, &sn, &c);
        cs = make_double2(c, sn);
    }
    tab[i] = cs;
}

// dense 1-qubit gate / CNOT: one amplitude pair per thread
__global__ __launch_bounds__(kT) void k_pair(double2* st, int n, int C, int kind, int q0, int q1,
                                             const double2* tab, int n_ops, int g) {
    const int64_t PP = (int64_t)1 << (n - 1);
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (i >= PP * C) return;
    const int ci = (int)(i >> (n - 1));
    const int64_t pi = i & (PP - 1);
    const int64_t bit = (int64_t)1 << q0;
    const int64_t i0 = ((pi >> q0) << (q0 + 1)) | (pi & (bit - 1));
    double2* s = st + ((int64_t)ci << n);
    const double2 a0 = s[i0], a1 = s[i0 | bit];
    const double2 cs = tab[(int64_t)ci * n_ops + g];
    const double c = cs.x, sn = cs.y;
    double2 r0, r1;
    switch (kind) {
    case VQA_GATE_H: {
        const double r2 = 0.70710678118654752440;
        r0 = make_double2(r2 * (a0.x + a1.x), r2 * (a0.y + a1.y));
        r1 = make_double2(r2 * (a0.x - a1.x), r2 * (a0.y - a1.y));
        break;
    }
    case VQA_GATE_X: r0 = a1; r1 = a0; break;
    case VQA_GATE_Y:
        r0 = make_double2(a1.y, -a1.x);
        r1 = make_double2(-a0.y, a0.x);
        break;
    case VQA_GATE_RX:
        r0 = make_double2(c * a0.x - sn * a1.y, c * a0.y + sn * a1.x);
        r1 = make_double2(c * a1.x - sn * a0.y, c * a1.y + sn * a0.x);
        break;
    case VQA_GATE_RY:
        r0 = make_double2(c * a0.x + sn * a1.x, c * a0.y + sn * a1.y);
        r1 = make_double2(c * a1.x - sn * a0.x, c * a1.y - sn * a0.y);
        break;
    case VQA_GATE_CNOT:
        if (i0 & ((int64_t)1 << q1)) { r0 = a1; r1 = a0; } else { return; }
        break;
    default: return;
    }
    s[i0] = r0;
    s[i0 | bit] = r1;
}

// diagonal gates: RZ, Z, CZ -- one amplitude per thread
__global__ __launch_bounds__(kT) void k_diag(double2* st, int n, int C, int kind, int q0, int q1,
                                             const double2* tab, int n_ops, int g) {
    const int64_t D = (int64_t)1 << n;
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (i >= D * C) return;
    const int ci = (int)(i >> n);
    const int64_t k = i & (D - 1);
    const int64_t bit = (int64_t)1 << q0;
    if (kind == VQA_GATE_RZ) {
        const double2 cs = tab[(int64_t)ci * n_ops + g];
        const double ss = (k & bit) ? -cs.y : cs.y;
        const double2 v = st[i];
        st[i] = make_double2(cs.x * v.x - ss * v.y, cs.x * v.y + ss * v.x);
    } else {
        const int64_t need = kind == VQA_GATE_CZ ? (bit | ((int64_t)1 << q1)) : bit;
        if ((k & need) == need) {
            const double2 v = st[i];
            st[i] = make_double2(-v.x, -v.y);
        }
    }
}


// ---- tiled passes -----------------------------------------------------------------
constexpr int kTileBits = VQA_TILE_BITS;      // qubits of a tile
constexpr int kTileAmps = 1 << kTileBits;
constexpr int kTileLow = 4;                   // the lowest qubits belong to every tile (256-byte runs in HBM)
constexpr int kGroup = VQA_TILE_GROUP;        // tile qubits held in a thread's registers during a sweep
constexpr int kTT = kTileAmps >> kGroup;      // threads of a tile workgroup
constexpr int kTA = 1 << kGroup;              // amplitudes per thread
constexpr int kThreadBits = kTileBits - kGroup;
#ifdef VQA_TILE_WAVES
constexpr int kTileWaves = VQA_TILE_WAVES;
#else
constexpr int kTileWaves = 2 * (kTT / 64) / 4; // waves per SIMD with two workgroups on a CU (what their LDS allows)
#endif
constexpr int kGateBatch = 256;               // most gates in one pass
constexpr int kMaxSweeps = 64;                // most sweeps in one pass
// internal gate kinds (beside VQA_GATE_*)
constexpr int kGateZZ = 9;                    // CNOT(c,t) RZ_t(a) CNOT(c,t) = exp(i a/2 Z_c Z_t), p0 = t, p1 = c
constexpr int kGateU = 10;                    // a fused run of 1-qubit gates on one qubit with a dense 2x2, g = run number
constexpr int kGateD1A = 11;                  // the diagonal factor behind the real rotation of fused 2x2 number g
constexpr int kGateD1B = 12;                  // the diagonal factor in front of it

// one gate of a pass: positions < 32 are tile positions, 32 + q is outer qubit q, -1 unused
struct TileGate {
    int32_t kind, p0, p1, g;                  // g = index into the angle table (D1A / D1B: into the plan's table of fused 2x2s)
};
// Normal form.  Every maximal run of 1-qubit gates on a qubit that holds a non-diagonal gate is
// multiplied out per instance (k_mats) and written as
//     M = diag(A0, A1) * R * diag(1, b1),   R = [[c, -s], [s, c]],  c = |M00|, s = |M10|,  |A0| = |A1| = |b1| = 1,
// and the real rotation R is applied as three shears  a0 -= u a1;  a1 += s a0;  a0 -= u a1  with
// u = tan(phi / 2) = s / (1 + c) <= 1 (stable for every angle in [0, pi/2]), so the dense part costs three
// fused multiply-adds per amplitude (a complex 2x2 costs eight) and
// the complex factors join the diagonal gates around
// the rotation, which are applied together, once per layer of the circuit, as a phase per amplitude.
//
// A pass is a list of SWEEPS over the tile in LDS.  In a sweep every thread holds the
// amplitudes that differ in the sweep's group positions, applies the sweep's register
// operations to them in order -- rotations on a group position and CNOTs whose target is a
// group position (the control may be any qubit) -- then the sweep's diagonal run, and puts
// them back.  A diagonal run is a phase per amplitude: the product of a 64-entry table for the
// low six tile positions (with the outer bits, fixed for the tile, folded in), one for the high
// six, and the few two-qubit gates that join the halves.  The tables are worked out per instance
// beforehand (k_prep_tab) up to the gates that involve outer qubits, which depend on the tile.
enum { OP_ROT = 0, OP_CNOT = 1 };
struct GroupOp {
    int32_t type, slot, a, b;                 // ROT: a = fused 2x2 number within the pass, b = 1: diag(1, b1) is applied here,
                                              // in front of the rotation (nothing earlier in the pass could take it); CNOT: a = control position
};
struct MatDesc {
    int32_t begin, count;                     // gates [begin, begin + count) of the plan's gate image, multiplied in order;
                                              // count | kMatPlain: keep the three-shear form (A0 = A1 = 1 is counted on)
};
constexpr int32_t kMatPlain = 1 << 30;
// a diagonal run: its gates inside the tile, except the two-qubit gates that join the halves
struct RunDesc {
    int32_t g1_begin, g1_count;               // single-qubit gates in the plan's gate image
    int32_t g2_begin, g2_count;               // two-qubit gates in the plan's gate image
};
struct SweepDesc {
    int32_t gpos[4];                          // group slot -> tile position
    int32_t tpos[kThreadBits];                // thread bit -> tile position
    int32_t op_begin, op_count;               // register operations (plan numbering) of a sweep with CNOTs, in order
    int32_t rot[4];                           // a sweep without CNOTs: the rotation on each group slot (plan numbering), -1: none
    int32_t run;                              // diagonal run (plan numbering), -1: none
    int32_t next_run;                         // the next diagonal run of the pass behind this sweep, -1: none
    int32_t x_begin, x_tab, x_count;          // late gates of the run (plan numbering): the first x_tab need the tile's outer
                                              // bits and go into the phase table, the others join the halves of the tile
    int32_t nx_begin, nx_tab;                 // x_begin, x_tab of next_run
};
// What the pass kernel reads per instance, written by the k_prep_* kernels: everything a
// workgroup needs besides the amplitudes is the same for all its threads, so it sits in
// global memory in the order of use and arrives through the scalar data cache.
struct OpRec {                                // one register operation
    double t, u;                              // rotation: tan(phi / 2), sin(phi) of the normal form
    int32_t code;                             // bit slot: rotation on that group slot, bit 4 + slot: CNOT onto it, bit 8: diag(1, b1) first, bit 9: the two-shear form, bit 10: the three-shear form
    int32_t ctrl;                             // CNOT: control position
    int32_t pad[2];
    double2 b1;
};
struct XRec {                                 // one late gate: the amplitude gets v1 when sel is set, else v0
    int32_t mode;                             // sel = bit(p0) (4), bit(p0) ^ bit(p1) (5), bit(p0) & bit(p1) with v = +-1 (2)
    int32_t p0, p1, pad;
    double2 v0, v1;
};
// one sweep as the pass kernel reads it (128 bytes, fetched a sweep ahead): the descriptor and the
// rotations of a sweep without CNOTs
struct SweepRec {
    uint32_t pos_lo, pos_hi;                  // 4 bits per entry: tpos[0 .. kThreadBits), then gpos[0 .. 4)
    int32_t flags;                            // bit s: rotation on group slot s; bit 4 + s: diag(1, b1) first; bit 8 + s: the
                                              // two-shear form; bit 12 + s: the three-shear form; bit 16: operations from the records;
                                              // bit 20 + g: g of the (ascending) group positions lie in the low half of the tile positions
    int32_t run, next_run, x_begin, x_tab, x_count, nx_begin, nx_tab, op_begin, op_count;
    int32_t rot[4];                           // the rotations' records (b1 is read there when it is needed)
    double tu[4][2];                          // their shears
};
static_assert(sizeof(SweepRec) == 128, "SweepRec is fetched as two 64-byte scalar loads");

struct TilePassArgs {
    int n;
    int tq[kTileBits];                        // tile position -> qubit (ascending)
    int oq[32];                               // outer bit number -> qubit (ascending)
    int n_outer;
    int n_sweeps;                             // this pass
    int n_ops_total, n_runs_total, n_x_total, n_sweeps_total; // row lengths of the per-instance tables
    int sweep_off;                            // where the pass's sweeps start in a row
#ifdef VQA_TILE_ABLATE
    int ablate;                               // timing experiments (libvqahip_abl.so): bit 0 no sweeps, 1 no LDS traffic in the sweeps, 2 no register operations, 3 no diagonal runs,
                                              // 4 no late gates in the tables, 5 no late gates that join the halves
#endif
};

struct c2 {
    double x, y;
};
__host__ __device__ __forceinline__ c2 cm(c2 a, c2 b) { return { a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x }; }
__host__ __device__ __forceinline__ c2 ca(c2 a, c2 b) { return { a.x + b.x, a.y + b.y }; }
__device__ __forceinline__ c2 cdiv(c2 a, c2 b) {
    const double d = 1.0 / (b.x * b.x + b.y * b.y);
    return { (a.x * b.x + a.y * b.y) * d, (a.y * b.x - a.x * b.y) * d };
}
__device__ __forceinline__ c2 c2of(double2 v) { return { v.x, v.y }; }
__device__ __forceinline__ double2 d2of(c2 v) { return make_double2(v.x, v.y); }

__device__ __forceinline__ bool gate_bit(int p, int64_t base, int i) {
    return p >= 32 ? ((base >> (p - 32)) & 1) != 0 : ((i >> p) & 1) != 0;
}

// LDS slot of tile index i: the low four bits are XORed with the two nibbles above, so that
// 16 indices that differ in one position of each residue class mod 4 land in 16 different
// 16-byte bank groups (the sweeps choose their thread bits accordingly)
__device__ __forceinline__ int tile_slot(int i) { return i ^ ((i >> 4) & 15) ^ ((i >> 8) & 15); }

// LDS slot of entry e of a 64-entry phase table: bits 4 and 5 are XORed onto bits 0 and 1.  The 16 lanes
// of a row differ in one tile position of each residue class mod 4, so within a half of the positions
// bit 4 never varies together with bit 0, nor bit 5 with bit 1: their entries land in different 16-byte
// bank groups.  XOR-linear: slot(a | b) = slot(a) ^ slot(b) for disjoint a, b.
__device__ __forceinline__ int tab_slot(int e) { return e ^ (e >> 4); }

// the product of the 1-qubit gates [begin, begin + count) for every instance, in normal form:
// mats[(ci * M + m) * 4 ..] = (tan(phi / 2), sin(phi)) or (-s / c, s c), A0, A1, b1
__global__ __launch_bounds__(kT) void k_mats(const TileGate* __restrict__ gates, const MatDesc* __restrict__ descs, int M,
                                             const double2* __restrict__ tab0, int n_ops, int C, double2* __restrict__ mats0,
                                             LaneStride ls) {
    const double2* __restrict__ tab = tab0 + ls.tab * blockIdx.z;
    double2* __restrict__ mats = lane_bytes(mats0, ls.inst);
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (int64_t)C * M) return;
    const int ci = (int)(i / M), k = (int)(i - (int64_t)ci * M);
    const MatDesc md = descs[k];
    c2 m00 = { 1, 0 }, m01 = { 0, 0 }, m10 = { 0, 0 }, m11 = { 1, 0 };
    for (int g = md.begin; g < md.begin + (md.count & ~kMatPlain); ++g) {
        const TileGate tg = gates[g];
        const bool rot = tg.kind >= VQA_GATE_RX && tg.kind <= VQA_GATE_RZ;   // the others have no table entry to read
        const double2 cs = rot ? tab[(int64_t)ci * n_ops + tg.g] : make_double2(1.0, 0.0);
        const double c = cs.x, sn = cs.y;
        const double r2 = 0.70710678118654752440;
        c2 g00, g01, g10, g11;
        switch (tg.kind) {
        case VQA_GATE_H: g00 = { r2, 0 }; g01 = { r2, 0 }; g10 = { r2, 0 }; g11 = { -r2, 0 }; break;
        case VQA_GATE_X: g00 = { 0, 0 }; g01 = { 1, 0 }; g10 = { 1, 0 }; g11 = { 0, 0 }; break;
        case VQA_GATE_Y: g00 = { 0, 0 }; g01 = { 0, -1 }; g10 = { 0, 1 }; g11 = { 0, 0 }; break;
        case VQA_GATE_Z: g00 = { 1, 0 }; g01 = { 0, 0 }; g10 = { 0, 0 }; g11 = { -1, 0 }; break;
        case VQA_GATE_RX: g00 = { c, 0 }; g01 = { 0, sn }; g10 = { 0, sn }; g11 = { c, 0 }; break;
        case VQA_GATE_RY: g00 = { c, 0 }; g01 = { sn, 0 }; g10 = { -sn, 0 }; g11 = { c, 0 }; break;
        default: /* RZ */ g00 = { c, sn }; g01 = { 0, 0 }; g10 = { 0, 0 }; g11 = { c, -sn }; break;
        }
        const c2 n00 = ca(cm(g00, m00), cm(g01, m10)), n01 = ca(cm(g00, m01), cm(g01, m11));
        const c2 n10 = ca(cm(g10, m00), cm(g11, m10)), n11 = ca(cm(g10, m01), cm(g11, m11));
        m00 = n00; m01 = n01; m10 = n10; m11 = n11;
    }
    // M00 (or M01), M10 (or M11) and one more entry are reproduced as they are (to rounding), the
    // fourth through the unitarity of M
    const double tiny = 1e-150;
    const double c = hypot(m00.x, m00.y), s = hypot(m10.x, m10.y);
    c2 A0, A1, b1;
    if (c >= s) {
        A0 = { m00.x / c, m00.y / c };
        if (s > tiny) {
            A1 = { m10.x / s, m10.y / s };
            b1 = cdiv(m11, { A1.x * c, A1.y * c });
        } else {
            A1 = { m11.x / c, m11.y / c };
            b1 = { 1.0, 0.0 };
        }
    } else {
        A1 = { m10.x / s, m10.y / s };
        if (c > tiny) {
            A0 = { m00.x / c, m00.y / c };
            b1 = cdiv({ -m01.x, -m01.y }, { A0.x * s, A0.y * s });
        } else {
            A0 = { -m01.x / s, -m01.y / s };
            b1 = { 1.0, 0.0 };
        }
    }
    double t = s / (1.0 + c), u = s;   // tan(phi / 2), sin(phi)
    if (c >= 0.5 && !(md.count & kMatPlain)) {
        // two shears are enough up to the scales 1 / c and c, which A0 and A1 take over: a1 += (s / c) a0,
        // a0 -= (s c) a1.  (The first component comes out of a difference that leaves c^2 of it, so
        // the form is kept to angles where that costs two bits at most.)  Marked by a negative t.
        t = -(s / c);
        u = s * c;
        A0 = { A0.x / c, A0.y / c };
        A1 = { A1.x * c, A1.y * c };
    }
    double2* o = mats + i * 4;
    o[0] = make_double2(t, u);
    o[1] = d2of(A0);
    o[2] = d2of(A1);
    o[3] = d2of(b1);
}

// the factor a single-qubit diagonal gate puts on an amplitude whose bit is `bit`
__device__ __forceinline__ c2 diag1_factor(const TileGate& tg, bool bit, const double2* __restrict__ tab_row,
                                           const double2* __restrict__ mats_row) {
    if (tg.kind == VQA_GATE_RZ) {
        const double2 cs = tab_row[tg.g];
        return { cs.x, bit ? -cs.y : cs.y };
    }
    if (tg.kind == VQA_GATE_Z) return { bit ? -1.0 : 1.0, 0.0 };
    const double2* m = mats_row + (int64_t)tg.g * 4;
    if (tg.kind == kGateD1A) return c2of(bit ? m[2] : m[1]);
    return bit ? c2of(m[3]) : c2{ 1.0, 0.0 };
}

// the register operations of every instance: oprec[ci * NO + o]
__global__ __launch_bounds__(kT) void k_prep_ops(const GroupOp* __restrict__ ops, const int32_t* __restrict__ op_mat0, int NO,
                                                 int C, const double2* __restrict__ mats0, int M, OpRec* __restrict__ oprec0,
                                                 LaneStride ls) {
    const double2* __restrict__ mats = lane_bytes(mats0, ls.inst);
    OpRec* __restrict__ oprec = lane_bytes(oprec0, ls.inst);
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (int64_t)C * NO) return;
    const int ci = (int)(i / NO), o = (int)(i - (int64_t)ci * NO);
    const GroupOp op = ops[o];
    OpRec r;
    r.t = r.u = 0.0;
    r.ctrl = r.pad[0] = r.pad[1] = 0;
    r.b1 = make_double2(1.0, 0.0);
    if (op.type == OP_ROT) {
        const double2* m = mats + ((int64_t)ci * M + op_mat0[o] + op.a) * 4;
        r.t = fabs(m[0].x);
        r.u = m[0].y;
        r.b1 = m[3];
        r.code = (1 << op.slot) | (op.b ? 256 : 0) | (m[0].x < 0.0 ? 512 : 1024);
    } else {
        r.code = 16 << op.slot;
        r.ctrl = op.a;
    }
    oprec[i] = r;
}

// the sweeps of every instance: swrec[ci * NS + k]
__global__ __launch_bounds__(kT) void k_prep_sweeps(const SweepDesc* __restrict__ sweeps, int NS, int C,
                                                    const OpRec* __restrict__ oprec0, int NO, SweepRec* __restrict__ swrec0,
                                                    LaneStride ls) {
    const OpRec* __restrict__ oprec = lane_bytes(oprec0, ls.inst);
    SweepRec* __restrict__ swrec = lane_bytes(swrec0, ls.inst);
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (int64_t)C * NS) return;
    const int ci = (int)(i / NS), k = (int)(i - (int64_t)ci * NS);
    const SweepDesc sd = sweeps[k];
    SweepRec r;
    uint64_t pos = 0;
    for (int b = 0; b < kThreadBits; ++b) pos |= (uint64_t)sd.tpos[b] << (4 * b);
    for (int j = 0; j < 4; ++j) pos |= (uint64_t)sd.gpos[j] << (4 * (kThreadBits + j));
    r.pos_lo = (uint32_t)pos;
    r.pos_hi = (uint32_t)(pos >> 32);
    r.flags = sd.op_count > 0 ? 1 << 16 : 0;
    int glo = 0;
    for (int j = 0; j < kGroup; ++j) glo += sd.gpos[j] < 6;
    r.flags |= 1 << (20 + glo);
    r.run = sd.run; r.next_run = sd.next_run;
    r.x_begin = sd.x_begin; r.x_tab = sd.x_tab; r.x_count = sd.x_count;
    r.nx_begin = sd.nx_begin; r.nx_tab = sd.nx_tab;
    r.op_begin = sd.op_begin; r.op_count = sd.op_count;
    for (int j = 0; j < 4; ++j) {
        r.rot[j] = sd.rot[j];
        r.tu[j][0] = r.tu[j][1] = 0.0;
        if (sd.rot[j] >= 0) {
            const OpRec o = oprec[(int64_t)ci * NO + sd.rot[j]];
            r.tu[j][0] = o.t;
            r.tu[j][1] = o.u;
            r.flags |= (1 << j) | ((o.code & 256) ? 16 << j : 0) | ((o.code & 512) ? 256 << j : 0) | ((o.code & 1024) ? 4096 << j : 0);
        }
    }
    swrec[i] = r;
}

// the two 64-entry phase tables of every diagonal run of every instance, without the late gates:
// pretab[(ci * NR + run) * 128 + e], e < 64: low six tile bits = e, e >= 64: high six = e - 64
__global__ __launch_bounds__(128) void k_prep_tab(const TileGate* __restrict__ gates, const RunDesc* __restrict__ runs, int NR,
                                                  const double2* __restrict__ tab0, int n_ops, const double2* __restrict__ mats0,
                                                  int M, double2* __restrict__ pretab0, LaneStride ls) {
    const double2* __restrict__ tab = tab0 + ls.tab * blockIdx.z;
    const double2* __restrict__ mats = lane_bytes(mats0, ls.inst);
    double2* __restrict__ pretab = lane_bytes(pretab0, ls.inst);
    const int run = blockIdx.x, ci = blockIdx.y, tid = threadIdx.x;
    const RunDesc rd = runs[run];
    const int half = tid >> 6, i = half ? (tid & 63) << 6 : (tid & 63);
    const double2* tab_row = tab + (int64_t)ci * n_ops;
    const double2* mats_row = mats + (int64_t)ci * M * 4;
    c2 ph = { 1.0, 0.0 };
    for (int g = rd.g1_begin; g < rd.g1_begin + rd.g1_count; ++g) {
        const TileGate tg = gates[g];
        if ((int)(tg.p0 >= 6) != half) continue;
        ph = cm(ph, diag1_factor(tg, (i >> tg.p0) & 1, tab_row, mats_row));
    }
    bool neg = false;
    for (int g = rd.g2_begin; g < rd.g2_begin + rd.g2_count; ++g) {
        const TileGate tg = gates[g];
        if ((int)(tg.p0 >= 6) != half) continue;
        const bool b0 = (i >> tg.p0) & 1, b1 = (i >> tg.p1) & 1;
        if (tg.kind == kGateZZ) {
            const double2 cs = tab_row[tg.g];
            ph = cm(ph, { cs.x, b0 != b1 ? -cs.y : cs.y });
        } else {
            neg ^= b0 && b1;
        }
    }
    if (neg) ph = { -ph.x, -ph.y };
    pretab[((int64_t)ci * NR + run) * 128 + tid] = d2of(ph);
}

// the late gates of every instance: xrec[ci * NX + k]
__global__ __launch_bounds__(kT) void k_prep_x(const TileGate* __restrict__ xg, int NX, int C, const double2* __restrict__ tab0,
                                               int n_ops, XRec* __restrict__ xrec0, LaneStride ls) {
    const double2* __restrict__ tab = tab0 + ls.tab * blockIdx.z;
    XRec* __restrict__ xrec = lane_bytes(xrec0, ls.inst);
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (int64_t)C * NX) return;
    const int ci = (int)(i / NX), k = (int)(i - (int64_t)ci * NX);
    const TileGate tg = xg[k];
    XRec r;
    r.p0 = tg.p0;
    r.p1 = tg.p1 < 0 ? tg.p0 : tg.p1;
    r.pad = 0;
    r.v0 = make_double2(1.0, 0.0);
    r.v1 = make_double2(-1.0, 0.0);
    if (tg.kind == VQA_GATE_RZ || tg.kind == kGateZZ) {
        const double2 cs = tab[(int64_t)ci * n_ops + tg.g];
        r.mode = tg.kind == kGateZZ ? 5 : 4;
        r.v0 = cs;
        r.v1 = make_double2(cs.x, -cs.y);
    } else {
        r.mode = tg.kind == VQA_GATE_CZ ? 2 : 4;   // Z: v = +-1 by its one bit
    }
    xrec[i] = r;
}

// The amplitude updates are written as inline assembly with tied operands so that they are in
// place by construction: behind the uniform branches of the loop over the operations anything
// else makes the register allocator keep a second copy of the state and move it back after
// every operation.
//   (a0, a1) <- (c a0 - s a1, s a0 + c a1) as three shears, t = tan(phi / 2), u = sin(phi), for the
// two components of two pairs at once (the steps of a pair depend on each other)
__device__ __forceinline__ void shear_quad(double& a0, double& a1, double& b0, double& b1, double& c0, double& c1,
                                           double& d0, double& d1, double t, double u) {
    asm("v_fma_f64 %0, -%8, %1, %0\n\tv_fma_f64 %2, -%8, %3, %2\n\tv_fma_f64 %4, -%8, %5, %4\n\tv_fma_f64 %6, -%8, %7, %6\n\t"
        "v_fma_f64 %1, %9, %0, %1\n\tv_fma_f64 %3, %9, %2, %3\n\tv_fma_f64 %5, %9, %4, %5\n\tv_fma_f64 %7, %9, %6, %7\n\t"
        "v_fma_f64 %0, -%8, %1, %0\n\tv_fma_f64 %2, -%8, %3, %2\n\tv_fma_f64 %4, -%8, %5, %4\n\tv_fma_f64 %6, -%8, %7, %6"
        : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1), "+v"(d0), "+v"(d1)
        : "s"(t), "s"(u));
}
//   the two-shear form: a1 += t a0, a0 -= u a1
__device__ __forceinline__ void shear2_quad(double& a0, double& a1, double& b0, double& b1, double& c0, double& c1,
                                            double& d0, double& d1, double t, double u) {
    asm("v_fma_f64 %1, %8, %0, %1\n\tv_fma_f64 %3, %8, %2, %3\n\tv_fma_f64 %5, %8, %4, %5\n\tv_fma_f64 %7, %8, %6, %7\n\t"
        "v_fma_f64 %0, -%9, %1, %0\n\tv_fma_f64 %2, -%9, %3, %2\n\tv_fma_f64 %4, -%9, %5, %4\n\tv_fma_f64 %6, -%9, %7, %6"
        : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1), "+v"(d0), "+v"(d1)
        : "s"(t), "s"(u));
}
//   (ax + i ay) <- (ax + i ay) (tx + i ty)
__device__ __forceinline__ void cmul_inplace(double& ax, double& ay, double tx, double ty) {
    double t1, t2;
    asm("v_mul_f64 %2, %1, %5\n\tv_mul_f64 %3, %1, %4\n\tv_fma_f64 %1, %0, %5, %3\n\tv_fma_f64 %0, %0, %4, -%2"
        : "+v"(ax), "+v"(ay), "=&v"(t1), "=&v"(t2)
        : "v"(tx), "v"(ty));
}
//   (a, b) <- (b, a) where m1 = 1, m0 = 0; unchanged where m1 = 0, m0 = 1 (exact: products with 0 and 1 of finite numbers)
__device__ __forceinline__ void cswap_inplace(double& a, double& b, double m0, double m1) {
    double t1, t2;
    asm("v_mul_f64 %2, %1, %5\n\tv_mul_f64 %3, %0, %5\n\tv_fma_f64 %0, %0, %4, %2\n\tv_fma_f64 %1, %1, %4, %3"
        : "+v"(a), "+v"(b), "=&v"(t1), "=&v"(t2)
        : "v"(m0), "v"(m1));
}

// the rotation of a fused 2x2 on group slot J (see "Normal form"); pre: diag(1, b1) first
template <int J>
__device__ __forceinline__ void apply_rot(double2 (&a)[kTA], double t, double u, bool pre, bool two, bool three, c2 b1) {
    if (pre) {
#pragma unroll
        for (int r = 0; r < kTA; ++r)
            if (r & (1 << J)) cmul_inplace(a[r].x, a[r].y, b1.x, b1.y);
    }
    // the pairs of slot J, two at a time; the two forms are separate blocks behind flags of their own (not an if / else)
    constexpr int lo = (1 << J) - 1;
    if (three) {
#pragma unroll
        for (int q = 0; q < kTA / 2; q += 2) {
            const int r0 = ((q & ~lo) << 1) | (q & lo), r1 = (((q + 1) & ~lo) << 1) | ((q + 1) & lo);
            shear_quad(a[r0].x, a[r0 | (1 << J)].x, a[r0].y, a[r0 | (1 << J)].y, a[r1].x, a[r1 | (1 << J)].x, a[r1].y,
                       a[r1 | (1 << J)].y, t, u);
        }
    }
    if (two) {
#pragma unroll
        for (int q = 0; q < kTA / 2; q += 2) {
            const int r0 = ((q & ~lo) << 1) | (q & lo), r1 = (((q + 1) & ~lo) << 1) | ((q + 1) & lo);
            shear2_quad(a[r0].x, a[r0 | (1 << J)].x, a[r0].y, a[r0 | (1 << J)].y, a[r1].x, a[r1 | (1 << J)].x, a[r1].y,
                        a[r1 | (1 << J)].y, t, u);
        }
    }
}

template <int J>
__device__ __forceinline__ void apply_cnot(double2 (&a)[kTA], int cpos, int64_t base, int ib, const int (&off)[kTA]) {
#pragma unroll
    for (int r = 0; r < kTA; ++r) {
        if (r & (1 << J)) continue;
        const bool on = gate_bit(cpos, base, ib | off[r]);
        const double m1 = on ? 1.0 : 0.0, m0 = on ? 0.0 : 1.0;
        cswap_inplace(a[r].x, a[r | (1 << J)].x, m0, m1);
        cswap_inplace(a[r].y, a[r | (1 << J)].y, m0, m1);
    }
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every
// outstanding global load, which would stall the sweeps on the prefetch of the next tile.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the outer bits of tile number w
__device__ __forceinline__ uint32_t tile_base(const TilePassArgs& a, uint32_t w) {
    uint32_t base = 0;
    for (int j = 0; j < a.n_outer; ++j) base |= ((w >> j) & 1u) << a.oq[j];
    return base;
}

// one register operation: the record has one bit per alternative, tested one after the other
// (blocks, not a switch: each block then updates the amplitudes in place, and they stay in one
// set of registers around the loop over the operations)
__device__ __forceinline__ void apply_op(double2 (&v)[kTA], const OpRec& r, int64_t base, int ib, const int (&off)[kTA]) {
    const c2 b1 = c2of(r.b1);
    const bool pre = (r.code & 256) != 0, two = (r.code & 512) != 0, three = (r.code & 1024) != 0;
    if (r.code & 1) apply_rot<0>(v, r.t, r.u, pre, two, three, b1);
    if (r.code & 2) apply_rot<1>(v, r.t, r.u, pre, two, three, b1);
    if (r.code & 4) apply_rot<2>(v, r.t, r.u, pre, two, three, b1);
    if (kGroup > 3 && (r.code & 8)) apply_rot<kGroup - 1>(v, r.t, r.u, pre, two, three, b1);
    if (r.code & 16) apply_cnot<0>(v, r.ctrl, base, ib, off);
    if (r.code & 32) apply_cnot<1>(v, r.ctrl, base, ib, off);
    if (r.code & 64) apply_cnot<2>(v, r.ctrl, base, ib, off);
    if (kGroup > 3 && (r.code & 128)) apply_cnot<kGroup - 1>(v, r.ctrl, base, ib, off);
}


// The table part of a diagonal run: amplitude r of the thread gets the product of the low table's entry
// for the low six positions of its index and the high table's entry for the high six.  The group
// positions ascend, GLO of them lie in the low half: the low entry then depends on the low GLO bits of
// r only and the high entry on the others, 2^GLO + 2^(kGroup - GLO) table reads instead of two per
// amplitude.  The side with fewer entries stays in registers, the other one is read entry by entry.
template <int GLO>
__device__ __forceinline__ void diag_tables(double2 (&v)[kTA], const double2* dt, int ib, const int (&off)[kTA]) {
    constexpr int NL = 1 << GLO, NH = kTA >> GLO;
    const int tlo = tab_slot(ib & 63), thi = 64 + tab_slot(ib >> 6);
    if (NL <= NH) {
        c2 lo[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) lo[k] = c2of(dt[tlo ^ tab_slot(off[k] & 63)]);
#pragma unroll
        for (int m = 0; m < NH; ++m) {
            const c2 hi = c2of(dt[thi ^ tab_slot(off[m << GLO] >> 6)]);
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const c2 ph = cm(lo[k], hi);
                cmul_inplace(v[(m << GLO) | k].x, v[(m << GLO) | k].y, ph.x, ph.y);
            }
            if ((((m + 1) * NL) & 3) == 0) __builtin_amdgcn_sched_barrier(0);   // a few table reads in flight, not all
        }
    } else {
        c2 hi[NH];
#pragma unroll
        for (int m = 0; m < NH; ++m) hi[m] = c2of(dt[thi ^ tab_slot(off[m << GLO] >> 6)]);
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const c2 lo = c2of(dt[tlo ^ tab_slot(off[k] & 63)]);
#pragma unroll
            for (int m = 0; m < NH; ++m) {
                const c2 ph = cm(lo, hi[m]);
                cmul_inplace(v[(m << GLO) | k].x, v[(m << GLO) | k].y, ph.x, ph.y);
            }
            if ((((k + 1) * NH) & 3) == 0) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Sum of v over the 64 lanes of a wave (every lane gets it): DPP inside the rows of 16 lanes, then
// the four row sums through scalar registers.
template <int CTRL>
__device__ __forceinline__ double row_step(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v = row_step<0xB1>(v);    // quad_perm [1,0,3,2]
    v = row_step<0x4E>(v);    // quad_perm [2,3,0,1]
    v = row_step<0x141>(v);   // row_half_mirror
    v = row_step<0x140>(v);   // row_mirror
    double s = 0.0;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), row * 16);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), row * 16);
        s += __hiloint2double(hi, lo);
    }
    return s;
}

// Where the states of a launch are read and written, and what the pass measures at its end.
// A state that branches off the running unshifted state inside a pass: where it starts.
struct BranchRec {
    int64_t src_off;          // its source, in amplitudes from PassIO::alt_src: the running state as it was in front of sweep `start`
    int32_t start, pad;       // the first sweep of the pass that the state's own parameter can change: it runs the pass from there
};
struct PassIO {
    const double2* alt_src;   // states ci >= alt_from are read from here: the running unshifted state they branch off
    int alt_from;
    const BranchRec* branch;  // not null: state ci >= alt_from reads alt_src + branch[ci].src_off and starts at sweep branch[ci].start
    // state 0 (the running state) leaves the tile as it is in front of sweep s, for every bit s of snap_mask, in
    // snap + (number of lower bits set) states: what the states that branch off in this pass start from
    uint64_t snap_mask;
    double2* snap;
    const double2* rd0;       // state 0 is read from here (nullptr: from its own place)
    double2* wr0;             // state 0 is written here (nullptr: to its own place)
    // MEAS: per Z-string (or marginal bin) the tile's sums of sign |phi|^2 and sign Re(conj(psi) phi), psi = F,
    // to partial[((ci * n_tiles + tile) * kZBatch + t) * 2 + {0, 1}] (what k_zpart2 writes per block)
    const double2* F;
    double* partial;
    int nz, write_state;
    int state_first, n_states;   // set by launch_pass: the order of the workgroups
    uint32_t z[kZBatch];
    int32_t bin[kZBatch];     // < 0: Z-string z (a sign); >= 0: the indicator of (index & z) == bin
};

// One workgroup runs the pass on one tile of one instance.  (LANES = 0: a launch without lanes keeps
// the kernel's code as it is without them.)
template <int MEAS, int LANES>
__global__ __launch_bounds__(kTT, kTileWaves) void k_tile_pass(double2* __restrict__ st0, const SweepRec* __restrict__ swrec0,
                                                   const OpRec* __restrict__ oprec0, const double2* __restrict__ pretab0,
                                                   const XRec* __restrict__ xrec0, TilePassArgs a, PassIO io, LaneStride ls) {
    // this workgroup's lane: its states, its tables, its partial sums
    double2* __restrict__ st = LANES ? st0 + ls.st * blockIdx.z : st0;
    const SweepRec* __restrict__ swrec = LANES ? lane_bytes(swrec0, ls.inst) : swrec0;
    const OpRec* __restrict__ oprec = LANES ? lane_bytes(oprec0, ls.inst) : oprec0;
    const double2* __restrict__ pretab = LANES ? lane_bytes(pretab0, ls.inst) : pretab0;
    const XRec* __restrict__ xrec = LANES ? lane_bytes(xrec0, ls.inst) : xrec0;
    const int64_t lane_st = LANES ? ls.st * blockIdx.z : 0;   // (added where a state of `io` is used: `io` itself stays as it came)
    const int64_t lane_part = LANES ? ls.part * blockIdx.z : 0;
    extern __shared__ double2 tile[];
    __shared__ double2 dtab[2][128];   // the phase tables of the current diagonal run and of the next one
    const int tid = threadIdx.x;
    // workgroup number -> (tile, state); states first when they read a common state (its tile then
    // comes from HBM once for all of them)
    const uint32_t n_tiles = 1u << (a.n - kTileBits);
    // (states first: workgroups go round-robin to the eight XCDs, each with an L2 of its own, so the states of
    // a tile take every eighth workgroup number: the tile of the common state then comes through one L2)
    const uint32_t chunk = blockIdx.x >> 3;
    const int ci = io.state_first ? (int)(chunk % (uint32_t)io.n_states) : (int)(blockIdx.x >> (a.n - kTileBits));
    const uint32_t tile_no = io.state_first ? ((chunk / (uint32_t)io.n_states) << 3) | (blockIdx.x & 7) : blockIdx.x & (n_tiles - 1);
    double2* own = st + ((int64_t)ci << a.n);
    const double2* s = ci >= io.alt_from ? io.alt_src + lane_st : (ci == 0 && io.rd0) ? io.rd0 + lane_st : own;
    int s0 = 0;   // the first sweep this state runs
    if (io.branch && ci >= io.alt_from) {
        const BranchRec br = io.branch[ci];
        s += br.src_off;
        s0 = br.start;
    }
    const uint64_t snap_mask = ci == 0 ? io.snap_mask : 0;
    double2* sout = (ci == 0 && io.wr0) ? io.wr0 + lane_st : own;
    const OpRec* __restrict__ my_ops = oprec + (int64_t)ci * a.n_ops_total;
    const double2* __restrict__ my_tab = pretab + (int64_t)ci * a.n_runs_total * 128;
    const XRec* __restrict__ my_x = xrec + (int64_t)ci * a.n_x_total;
    const SweepRec* __restrict__ my_sw = swrec + (int64_t)ci * a.n_sweeps_total + a.sweep_off;
    // local index l = k * kTT + tid (12 bits) -> global index; the low four bits are contiguous
    // (state indices fit 32 bits: at most 30 qubits)
    uint32_t g_tid = tid & ((1 << kTileLow) - 1);
#pragma unroll
    for (int j = kTileLow; j < kThreadBits; ++j) g_tid |= (uint32_t)((tid >> j) & 1) << a.tq[j];
    uint32_t g_k[kTA];
#pragma unroll
    for (int k = 0; k < kTA; ++k) {
        g_k[k] = 0;
#pragma unroll
        for (int j = kThreadBits; j < kTileBits; ++j) g_k[k] |= (uint32_t)((k >> (j - kThreadBits)) & 1) << a.tq[j];
    }
    const uint32_t base = tile_base(a, tile_no);
    double2 pv[kTA];
#pragma unroll
    for (int k = 0; k < kTA; ++k) pv[k] = s[base | g_tid | g_k[k]];
#ifdef VQA_TILE_ABLATE
    const int abl = a.ablate;
#else
    constexpr int abl = 0;
#endif
    {
        // The phase of amplitude i of a diagonal run factors into a part from its low six tile bits (and
        // the outer bits, fixed here), a part from its high six, and the gates that join the two halves.
        // The parts come from the instance's tables; the late gates that involve an outer qubit are
        // applied to them here.  The tables of a run are put together a run ahead: the entry is
        // fetched at the start of the sweep of the run before and finished at its end.
        auto finish_table = [&](double2 entry, int xb, int xt, double2* dst) {
            const int half = tid >> 6, i = half ? (tid & 63) << 6 : (tid & 63);
            c2 ph = c2of(entry);
            for (int k = xb; k < xb + ((abl & 16) ? 0 : xt); ++k) {
                const XRec x = my_x[k];
                const int tp = x.p0 < 32 ? x.p0 : x.p1;   // the tile position it acts on, if any
                if ((int)(tp < 32 && tp >= 6) != half) continue;
                const bool b0 = gate_bit(x.p0, base, i), b1 = gate_bit(x.p1, base, i);
                const bool sel = (x.mode & 2) ? (b0 && b1) : (x.mode & 1) ? (b0 != b1) : b0;
                ph = cm(ph, c2of(sel ? x.v1 : x.v0));
            }
            dst[(tid & 64) | tab_slot(tid & 63)] = d2of(ph);
        };
        double2 ahead = make_double2(1.0, 0.0);
        SweepRec sw_next = my_sw[s0];
        // the first diagonal run at or behind the first sweep
        const int run0 = sw_next.run >= 0 ? sw_next.run : sw_next.next_run;
        const int xb0 = sw_next.run >= 0 ? sw_next.x_begin : sw_next.nx_begin, xt0 = sw_next.run >= 0 ? sw_next.x_tab : sw_next.nx_tab;
        if (run0 >= 0 && tid < 128) ahead = my_tab[run0 * 128 + tid];
#pragma unroll
        for (int k = 0; k < kTA; ++k) tile[tile_slot(k * kTT + tid)] = pv[k];
        if (run0 >= 0 && tid < 128) finish_table(ahead, xb0, xt0, dtab[0]);
        int cur = 0;   // which table the next diagonal run reads
        for (int si = s0; si < ((abl & 1) ? 0 : a.n_sweeps); ++si) {
            lds_barrier();   // the previous sweep's amplitudes (or the gathered tile) and the next table are in LDS
            if ((snap_mask >> si) & 1) {
                double2* snap = io.snap + lane_st + ((int64_t)__popcll(snap_mask & (((uint64_t)1 << si) - 1)) << a.n);
#pragma unroll
                for (int k = 0; k < kTA; ++k) snap[base | g_tid | g_k[k]] = tile[tile_slot(k * kTT + tid)];
                lds_barrier();   // everybody has read the tile before the sweep changes it
            }
            const SweepRec sw = sw_next;
            sw_next = my_sw[min(si + 1, a.n_sweeps - 1)];   // on its way while this sweep runs
            const uint64_t pos = ((uint64_t)sw.pos_hi << 32) | sw.pos_lo;
            const bool table_next = sw.run >= 0 && sw.next_run >= 0 && tid < 128;
            if (table_next) ahead = my_tab[sw.next_run * 128 + tid];
            const double2* dt = dtab[cur];
            // this thread's amplitudes: thread bits on tpos, register bits on gpos
            int ib = 0;
#pragma unroll
            for (int b = 0; b < kThreadBits; ++b) ib |= ((tid >> b) & 1) << (int)((pos >> (4 * b)) & 15);
            int off[kTA];
#pragma unroll
            for (int r = 0; r < kTA; ++r) {
                off[r] = 0;
#pragma unroll
                for (int j = 0; j < kGroup; ++j) off[r] |= ((r >> j) & 1) << (int)((pos >> (4 * (kThreadBits + j))) & 15);
            }
            const int sb = tile_slot(ib);
            double2 v[kTA];
#pragma unroll
            for (int r = 0; r < kTA; ++r) v[r] = (abl & 2) ? make_double2(1.0 + r, tid) : tile[sb ^ tile_slot(off[r])];
            const int o_end = (abl & 4) ? sw.op_begin : sw.op_begin + sw.op_count;
            if (sw.op_begin < o_end) {
                // the next record is on its way while the current operation runs
                OpRec cur = my_ops[sw.op_begin];
                for (int o = sw.op_begin; o < o_end; ++o) {
                    const OpRec nxt = my_ops[min(o + 1, o_end - 1)];
                    apply_op(v, cur, base, ib, off);
                    cur = nxt;
                }
            }
            if (!(abl & 4)) {
                // at most one rotation per group slot and nothing else: they commute, and their
                // shears came with the sweep's record
                const int f = sw.flags;
#define VQA_ROT_SLOT(J)                                                                                         \
                if (f & (1 << J)) {                                                                             \
                    c2 b1 = { 1.0, 0.0 };                                                                       \
                    if (f & (16 << J)) b1 = c2of(my_ops[sw.rot[J]].b1);                                         \
                    apply_rot<J>(v, sw.tu[J][0], sw.tu[J][1], (f & (16 << J)) != 0, (f & (256 << J)) != 0,      \
                                 (f & (4096 << J)) != 0, b1);                                                   \
                }
                VQA_ROT_SLOT(0) VQA_ROT_SLOT(1) VQA_ROT_SLOT(2)
                constexpr int JL = kGroup - 1;
                if (kGroup > 3) { VQA_ROT_SLOT(JL) }
#undef VQA_ROT_SLOT
            }
            if (sw.run >= 0 && !(abl & 8)) {
                // (one block per number of group positions in the low half, each updating the amplitudes in place)
                // and behind a flag test of its own: tests the compiler can prove exclusive would cost a second copy of the state)
                const int gf = sw.flags >> 20;
                if (gf & 1) diag_tables<0>(v, dt, ib, off);
                if (gf & 2) diag_tables<1>(v, dt, ib, off);
                if (gf & 4) diag_tables<2>(v, dt, ib, off);
                if (gf & 8) diag_tables<3>(v, dt, ib, off);
                if (kGroup > 3 && (gf & 16)) diag_tables<kGroup>(v, dt, ib, off);
                for (int k = sw.x_begin + sw.x_tab; k < sw.x_begin + ((abl & 32) ? sw.x_tab : sw.x_count); ++k) {
                    const XRec x = my_x[k];
                    // (two blocks, not an if / else: each updates the amplitudes in place)
                    if (x.mode & 2) {
#pragma unroll
                        for (int r = 0; r < kTA; ++r) {
                            const int i = ib | off[r];
                            const double sg = ((i >> x.p0) & (i >> x.p1) & 1) ? -1.0 : 1.0;
                            asm("v_mul_f64 %0, %0, %2\n\tv_mul_f64 %1, %1, %2" : "+v"(v[r].x), "+v"(v[r].y) : "v"(sg));
                        }
                    }
                    if (x.mode & 4) {
#pragma unroll
                        for (int r = 0; r < kTA; ++r) {
                            const int i = ib | off[r];
                            const bool sel = ((i >> x.p0) ^ ((x.mode & 1) ? (i >> x.p1) : 0)) & 1;
                            const double2 f = sel ? x.v1 : x.v0;
                            cmul_inplace(v[r].x, v[r].y, f.x, f.y);
                        }
                    }
                }
            }
            if (abl & 2) {
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < kTA; ++r) acc += v[r].x + v[r].y;
                if (acc == 0.123) tile[tid] = make_double2(acc, acc);
            } else {
#pragma unroll
                for (int r = 0; r < kTA; ++r) tile[sb ^ tile_slot(off[r])] = v[r];
            }
            if (table_next) finish_table(ahead, sw.nx_begin, sw.nx_tab, dtab[cur ^ 1]);
            if (sw.run >= 0) cur ^= 1;
        }
        if (MEAS) {
#pragma unroll
            for (int k = 0; k < kTA; ++k) pv[k] = io.F[lane_st + (base | g_tid | g_k[k])];   // psi, on its way during the barrier
        }
        lds_barrier();
        if (!MEAS || io.write_state) {
#pragma unroll
            for (int k = 0; k < kTA; ++k) sout[base | g_tid | g_k[k]] = tile[tile_slot(k * kTT + tid)];
        }
        if (MEAS) {
            // The index bits that differ between a thread's amplitudes are the kGroup highest tile
            // positions.  A Walsh-Hadamard transform over them gives, per thread, the sums of |phi|^2 and of
            // Re(conj(psi) phi) under every sign pattern of those bits; they are parked in LDS (the tile is
            // not needed any more), and an entry is then the sum over the threads of its pattern's value
            // times the sign (or bin test) of the thread's own index bits.  A bin's mask must avoid the
            // varying bits (the low qubits of a marginal always do: they are thread bits).
            double wp[kTA], wc[kTA];
#pragma unroll
            for (int k = 0; k < kTA; ++k) {
                const double2 phi = tile[tile_slot(k * kTT + tid)], psi = pv[k];
                wp[k] = phi.x * phi.x + phi.y * phi.y;
                wc[k] = psi.x * phi.x + psi.y * phi.y;
            }
#pragma unroll
            for (int h = 1; h < kTA; h <<= 1) {
#pragma unroll
                for (int k = 0; k < kTA; ++k) {
                    if (k & h) continue;
                    const double a0 = wp[k], a1 = wp[k | h], c0 = wc[k], c1 = wc[k | h];
                    wp[k] = a0 + a1; wp[k | h] = a0 - a1;
                    wc[k] = c0 + c1; wc[k | h] = c0 - c1;
                }
            }
            __syncthreads();   // everybody has read the tile
            double* wl = reinterpret_cast<double*>(tile);   // [pattern][quantity][thread]
#pragma unroll
            for (int k = 0; k < kTA; ++k) {
                wl[(2 * k) * kTT + tid] = wp[k];
                wl[(2 * k + 1) * kTT + tid] = wc[k];
            }
            __syncthreads();
            const int wave = tid >> 6, lane = tid & 63;
            // the own index bits of the threads lane, lane + 64, ...
            uint32_t g_lane = lane & ((1 << kTileLow) - 1);
#pragma unroll
            for (int j = kTileLow; j < 6; ++j) g_lane |= (uint32_t)((lane >> j) & 1) << a.tq[j];
            for (int t = wave; t < io.nz; t += kTT / 64) {
                const uint32_t z = io.z[t];
                const int bin = io.bin[t];
                int pat = 0;
#pragma unroll
                for (int j = 0; j < kGroup; ++j) pat |= (int)((z >> a.tq[kThreadBits + j]) & 1) << j;
                if (bin >= 0) pat = 0;
                double va = 0.0, vx = 0.0;
#pragma unroll
                for (int w = 0; w < kTT / 64; ++w) {
                    uint32_t idx = base | g_lane;
#pragma unroll
                    for (int j = 6; j < kThreadBits; ++j) idx |= (uint32_t)((w >> (j - 6)) & 1) << a.tq[j];
                    const bool on = bin >= 0 ? (idx & z) == (uint32_t)bin : !(__popc(idx & z) & 1);
                    const double sg = bin >= 0 ? (on ? 1.0 : 0.0) : (on ? 1.0 : -1.0);
                    va += sg * wl[(2 * pat) * kTT + w * 64 + lane];
                    vx += sg * wl[(2 * pat + 1) * kTT + w * 64 + lane];
                }
                va = wave_sum(va);
                vx = wave_sum(vx);
                if (lane == 0) {
                    double* out = &io.partial[lane_part + (((int64_t)ci * n_tiles + tile_no) * kZBatch + t) * 2];
                    out[0] = va;
                    out[1] = vx;
                }
            }
        }
    }
}

// Cuts the gate list into passes.  First the list is brought into the normal form's units:
// CNOT - RZ(target) - CNOT sandwiches become one ZZ phase gate, and every maximal run of
// 1-qubit gates on a qubit (nothing else touches the qubit in between) that holds a
// non-diagonal gate becomes one unit U.  A unit can run in a pass when its target (for the
// non-diagonal kinds) is a tile qubit and none of its qubits is held back by an earlier unit
// that could not run; units that share no qubit commute, so later ones on other qubits may run
// ahead of a held-back one.  Inside a pass the units are grouped into sweeps with the same
// argument: a rotation or CNOT goes into the earliest sweep at or behind the last one that
// touches its qubits in which its target is (or can become) a group position.  Diagonal gates
// (the units' own diagonal factors among them) wait until an operation on one of their qubits
// needs them applied; they then join the run of the sweep the previous diagonal gates went to
// when that is late enough, so that a layer of the circuit ends up with one diagonal run.
struct TilePlan {
    struct Pass {
        int tq[kTileBits];
        std::vector<TileGate> gates;   // the fused 2x2s' gates, the CNOTs, then the diagonal runs' gates
        std::vector<TileGate> xgates;  // the late gates
        std::vector<GroupOp> ops;
        std::vector<MatDesc> mats;     // gate ranges relative to this pass's gates
        std::vector<int> mat_first;    // per fused 2x2: the first sweep of the pass whose outcome depends on it (the one whose
                                       // diagonal run holds the factor in front of its rotation, or the rotation's own)
        std::vector<RunDesc> runs;     // likewise
        std::vector<SweepDesc> sweeps; // operations, runs and late gates in the pass's numbering
    };
    std::vector<Pass> passes;
    // images copied to the device: the passes' tables one after the other, in the plan's numbering
    std::vector<TileGate> all_gates, all_xgates;
    std::vector<GroupOp> all_ops;
    std::vector<int32_t> op_mat0;      // per operation: the plan's number of its pass's first fused 2x2
    std::vector<MatDesc> all_mats;
    std::vector<RunDesc> all_runs;
    std::vector<SweepDesc> all_sweeps;
    std::vector<size_t> soff;          // where each pass's sweeps start
    std::vector<int> pass_of_gate;     // gate index -> the pass that runs it
    std::vector<int> first_sweep_of_gate;   // gate index -> the first sweep of that pass it can change (0 unless it sits in a fused 2x2)
    // parameter-shift sweeps (filled by plan_sweep): the parameters that drive exactly one
    // rotation, ordered by the pass of that rotation, and the ones that drive none
    bool sweepable = false;
    std::vector<int32_t> ord, ord_pass, ord_sweep, unused;
    std::vector<double> ord_coef;
};

static bool diag_kind(int kind) {
    return kind == VQA_GATE_RZ || kind == VQA_GATE_Z || kind == VQA_GATE_CZ || kind == kGateZZ || kind == kGateD1A ||
           kind == kGateD1B;
}

// false when the pass needs more descriptors than the kernel holds (the caller retries with fewer units).
// runs[u] = the gates of unit U number u (their p0 is filled in here).
static bool group_sweeps(TilePlan::Pass& pass, const std::vector<TileGate>& in_order,
                         const std::vector<std::vector<TileGate>>& runs) {
    struct HOp { int type, pos, ctrl, mat, pre; TileGate cnot; };
    struct HSweep { std::vector<int> gpos; std::vector<HOp> ops; std::vector<TileGate> diag; };
    std::vector<HSweep> sweeps;
    std::map<int, int> last_op, last_diag;   // position -> last sweep with a register operation on it / a diagonal gate on it
    auto lo = [&](int p) { auto it = last_op.find(p); return it == last_op.end() ? -1 : it->second; };
    auto ld = [&](int p) { auto it = last_diag.find(p); return it == last_diag.end() ? -1 : it->second; };
    std::vector<TileGate> pending;
    int last_flush = -1;
    auto fits = [&](const HSweep& sw, int t) {
        if (std::find(sw.gpos.begin(), sw.gpos.end(), t) != sw.gpos.end()) return true;
        if ((int)sw.gpos.size() >= kGroup) return false;
        // every residue class mod 4 keeps a position outside the group (see tile_slot)
        int same = 0, members = 0;
        for (int p : sw.gpos) same += (p & 3) == (t & 3);
        for (int p = 0; p < kTileBits; ++p) members += (p & 3) == (t & 3);
        return same + 1 < members;
    };
    // applies the waiting diagonal gates that touch position p (p < 0: all of them)
    auto flush = [&](int p) {
        std::vector<TileGate> take, keep;
        for (const TileGate& d : pending) ((p < 0 || d.p0 == p || d.p1 == p) ? take : keep).push_back(d);
        if (take.empty()) return;
        int f = 0;
        for (const TileGate& d : take) {
            f = std::max(f, lo(d.p0));
            if (d.p1 >= 0) f = std::max(f, lo(d.p1));
        }
        f = std::max(f, last_flush);
        if (sweeps.empty()) sweeps.emplace_back();
        last_flush = f;
        for (const TileGate& d : take) {
            sweeps[f].diag.push_back(d);
            last_diag[d.p0] = std::max(ld(d.p0), f);
            if (d.p1 >= 0) last_diag[d.p1] = std::max(ld(d.p1), f);
        }
        pending.swap(keep);
    };
    auto place = [&](int target, int ctrl) {
        flush(target);
        int s = std::max({ lo(target), ld(target) + 1, ctrl >= 0 ? lo(ctrl) : -1, 0 });
        while (s < (int)sweeps.size() && !fits(sweeps[s], target)) ++s;
        if (s >= (int)sweeps.size()) { sweeps.emplace_back(); s = (int)sweeps.size() - 1; }
        HSweep& sw = sweeps[s];
        if (std::find(sw.gpos.begin(), sw.gpos.end(), target) == sw.gpos.end()) sw.gpos.push_back(target);
        last_op[target] = s;
        if (ctrl >= 0) last_op[ctrl] = std::max(lo(ctrl), s);
        return s;
    };
    for (const TileGate& tg : in_order) {
        if (tg.kind == kGateU) {
            const int p = tg.p0, mat = (int)pass.mats.size();
            // A lone Hadamard (the layers that turn X factors into Z factors) is R(pi/4) diag(1, -1) exactly: in
            // the three-shear form nothing is left behind the rotation, and its sweep needs no diagonal run for it.
            const bool plain_h = runs[tg.g].size() == 1 && runs[tg.g][0].kind == VQA_GATE_H;
            pass.mats.push_back(MatDesc{ (int)pass.gates.size(), (int)runs[tg.g].size() | (plain_h ? kMatPlain : 0) });
            for (TileGate e : runs[tg.g]) {
                e.p0 = p;
                pass.gates.push_back(e);
            }
            bool touched = lo(p) >= 0 || ld(p) >= 0;
            for (const TileGate& d : pending) touched |= d.p0 == p || d.p1 == p;
            if (touched) pending.push_back(TileGate{ kGateD1B, p, -1, mat });
            const int s = place(p, -1);   // (flushes the factor in front of the rotation, if any, to sweep last_flush)
            pass.mat_first.push_back(touched ? std::min(last_flush, s) : s);
            sweeps[s].ops.push_back(HOp{ OP_ROT, p, -1, mat, touched ? 0 : 1, {} });
            if (!plain_h) pending.push_back(TileGate{ kGateD1A, p, -1, mat });
        } else if (tg.kind == VQA_GATE_CNOT) {
            const int s = place(tg.p0, tg.p1);
            sweeps[s].ops.push_back(HOp{ OP_CNOT, tg.p0, tg.p1, -1, 0, tg });
        } else {
            pending.push_back(tg);
        }
    }
    flush(-1);
    if ((int)sweeps.size() > kMaxSweeps) return false;
    std::vector<TileGate> run_gates, cnot_gates;
    for (HSweep& hs : sweeps) {
        SweepDesc sd;
        memset(&sd, 0, sizeof(sd));
        // fill the group with unused positions, then hand out the thread bits: bits 0..3 get one
        // position of each residue class mod 4 (classes 0, 1, 2 first), the rest ascending
        for (int p = kTileBits - 1; p >= 0 && (int)hs.gpos.size() < kGroup; --p)
            if (fits(hs, p) && std::find(hs.gpos.begin(), hs.gpos.end(), p) == hs.gpos.end()) hs.gpos.push_back(p);
        std::sort(hs.gpos.begin(), hs.gpos.end());   // ascending: the kernel's diagonal run counts on it
        std::vector<int> rest;
        for (int p = 0; p < kTileBits; ++p)
            if (std::find(hs.gpos.begin(), hs.gpos.end(), p) == hs.gpos.end()) rest.push_back(p);
        std::vector<int> tp;
        for (int cls = 0; cls < 4; ++cls) {
            auto it = std::find_if(rest.begin(), rest.end(), [&](int p) { return (p & 3) == cls; });
            tp.push_back(*it);   // every class keeps a position outside the group (fits())
            rest.erase(it);
        }
        // A 16-byte LDS read serves the lanes in four groups of 16 that are not contiguous: lane bits 0 and 1
        // and the even (or odd) patterns of bits 2, 3, 4.  With the classes 0, 1, 2, 3 on bits 0 .. 3 the 16
        // lanes of such a group land in 16 different bank groups exactly when bit 4 gets a position of
        // class 0 or 1 (the sums of the slot bits of lane bits 2 + 3 and 2 + 4 then complete bits 0 and 1);
        // 16-byte writes go by 8 contiguous lanes, which the classes 0, 1, 2 on bits 0 .. 2 keep apart.
        {
            auto it = std::find_if(rest.begin(), rest.end(), [](int p) { return (p & 3) < 2; });
            if (it != rest.end()) std::rotate(rest.begin(), it, it + 1);
        }
        tp.insert(tp.end(), rest.begin(), rest.end());
        for (int j = 0; j < kGroup; ++j) sd.gpos[j] = hs.gpos[j];
        for (int b = 0; b < kThreadBits; ++b) sd.tpos[b] = tp[b];
        bool has_cnot = false;
        for (const HOp& op : hs.ops) has_cnot |= op.type == OP_CNOT;
        sd.op_begin = (int)pass.ops.size();
        for (int j = 0; j < 4; ++j) sd.rot[j] = -1;
        for (const HOp& op : hs.ops) {
            const int slot = (int)(std::find(hs.gpos.begin(), hs.gpos.end(), op.pos) - hs.gpos.begin());
            if (op.type == OP_ROT) {
                if (!has_cnot) sd.rot[slot] = (int)pass.ops.size();   // one per slot: a second one waits for the first one's diagonal factor
                pass.ops.push_back(GroupOp{ OP_ROT, slot, op.mat, op.pre });
            } else {
                pass.ops.push_back(GroupOp{ OP_CNOT, slot, op.ctrl, 0 });
                cnot_gates.push_back(op.cnot);
            }
        }
        sd.op_count = has_cnot ? (int)pass.ops.size() - sd.op_begin : 0;
        sd.run = sd.next_run = -1;
        if (!hs.diag.empty()) {
            // the gates inside the tile that stay within a half go into the instance's tables;
            // late gates: first those with an outer qubit, then those that join the halves
            auto high = [](int p) { return p >= 6; };
            std::vector<TileGate> one, two, x_tab, x_cross;
            for (const TileGate& d : hs.diag) {
                const bool o0 = d.p0 >= 32, o1 = d.p1 >= 32;
                if (o0 || o1) x_tab.push_back(d);
                else if (d.p1 < 0) one.push_back(d);
                else if (high(d.p0) != high(d.p1)) x_cross.push_back(d);
                else two.push_back(d);
            }
            sd.run = (int)pass.runs.size();
            RunDesc rd;
            rd.g1_begin = (int)run_gates.size();
            rd.g1_count = (int)one.size();
            run_gates.insert(run_gates.end(), one.begin(), one.end());
            rd.g2_begin = (int)run_gates.size();
            rd.g2_count = (int)two.size();
            run_gates.insert(run_gates.end(), two.begin(), two.end());
            pass.runs.push_back(rd);
            sd.x_begin = (int)pass.xgates.size();
            sd.x_tab = (int)x_tab.size();
            pass.xgates.insert(pass.xgates.end(), x_tab.begin(), x_tab.end());
            pass.xgates.insert(pass.xgates.end(), x_cross.begin(), x_cross.end());
            sd.x_count = (int)pass.xgates.size() - sd.x_begin;
        }
        pass.sweeps.push_back(sd);
    }
    for (int si = (int)pass.sweeps.size() - 1, nxt = -1, nxb = 0, nxt_tab = 0; si >= 0; --si) {
        SweepDesc& sd = pass.sweeps[si];
        sd.next_run = nxt;
        sd.nx_begin = nxb;
        sd.nx_tab = nxt_tab;
        if (sd.run >= 0) {
            nxt = sd.run;
            nxb = sd.x_begin;
            nxt_tab = sd.x_tab;
        }
    }
    pass.gates.insert(pass.gates.end(), cnot_gates.begin(), cnot_gates.end());
    const int run_base = (int)pass.gates.size();
    for (RunDesc& rd : pass.runs) {
        rd.g1_begin += run_base;
        rd.g2_begin += run_base;
    }
    pass.gates.insert(pass.gates.end(), run_gates.begin(), run_gates.end());
    return true;
}

// gate_dead (may be null): gates to leave out (outside the light cone of what a call measures)
static TilePlan plan_tiles(int n, const std::vector<vqa_op>& ops, const std::vector<uint8_t>* gate_dead = nullptr) {
    TilePlan plan;
    auto is_dead = [&](size_t g) { return gate_dead && (*gate_dead)[g]; };
    // the gate list in units: ZZ phase gates for the sandwiches, runs of 1-qubit gates per qubit
    struct PG { int kind, q0, q1, g; };
    std::vector<PG> pg;
    std::vector<std::vector<TileGate>> runs;   // the 1-qubit gates of a run, in order
    std::vector<int> open(n, -1);              // the run a qubit's next 1-qubit gate joins (index into pg), -1: none
    // A trainable RZ at the end of a qubit's run commutes with the CZ / ZZ gates behind it: it is carried over them and
    // opens the qubit's next run, so that its parameter's state branches off a layer later in a parameter-shift sweep
    // (the register engine does the same, reg_compile.cpp).  VQA_NO_RZ_FORWARD=1: as listed.
    std::vector<std::vector<TileGate>> carry(n);
    const bool forward = !getenv("VQA_NO_RZ_FORWARD");
    auto close_run = [&](int q) {              // a CZ / ZZ gate on q: the run ends, its trailing trainable RZs travel on
        if (open[q] >= 0 && forward) {
            std::vector<TileGate>& run = runs[pg[open[q]].g];
            size_t keep = run.size();
            while (keep > 1 && run[keep - 1].kind == VQA_GATE_RZ && ops[run[keep - 1].g].src == VQA_SRC_THETA) --keep;
            carry[q].insert(carry[q].begin(), run.begin() + keep, run.end());
            run.resize(keep);
        }
        open[q] = -1;
    };
    auto open_run = [&](int q) {               // the run the next 1-qubit gate on q joins, the carried gates first
        if (open[q] < 0) {
            open[q] = (int)pg.size();
            pg.push_back(PG{ kGateU, q, -1, (int)runs.size() });
            runs.emplace_back(carry[q]);
            carry[q].clear();
        }
    };
    for (size_t i = 0; i < ops.size(); ++i) {
        const vqa_op& op = ops[i];
        if (is_dead(i)) continue;
        if (op.kind == VQA_GATE_CNOT && i + 2 < ops.size() && ops[i + 1].kind == VQA_GATE_RZ && ops[i + 1].q0 == op.q0 &&
            ops[i + 2].kind == VQA_GATE_CNOT && ops[i + 2].q0 == op.q0 && ops[i + 2].q1 == op.q1 &&
            !is_dead(i + 1) && !is_dead(i + 2)) {
            close_run(op.q0);
            close_run(op.q1);
            pg.push_back(PG{ kGateZZ, op.q0, op.q1, (int)i + 1 });
            i += 2;
        } else if (op.kind == VQA_GATE_CZ) {
            close_run(op.q0);
            close_run(op.q1);
            pg.push_back(PG{ op.kind, op.q0, op.q1, (int)i });
        } else if (op.kind == VQA_GATE_CNOT) {
            // (not diagonal: what was carried so far is applied in front of it, as a run of its own)
            for (int q : { op.q0, op.q1 }) {
                if (!carry[q].empty()) { open[q] = -1; open_run(q); }
                open[q] = -1;
            }
            pg.push_back(PG{ op.kind, op.q0, op.q1, (int)i });
        } else {
            open_run(op.q0);
            runs[pg[open[op.q0]].g].push_back(TileGate{ op.kind, -1, -1, (int)i });
        }
    }
    for (int q = 0; q < n; ++q)
        if (!carry[q].empty()) { open[q] = -1; open_run(q); }
    // runs of diagonal gates only stay single diagonal gates
    {
        std::vector<PG> out;
        for (const PG& u : pg) {
            bool dense = false;
            if (u.kind == kGateU)
                for (const TileGate& e : runs[u.g]) dense |= !(e.kind == VQA_GATE_RZ || e.kind == VQA_GATE_Z);
            if (u.kind != kGateU || dense) {
                out.push_back(u);
            } else {
                for (const TileGate& e : runs[u.g]) out.push_back(PG{ e.kind, u.q0, -1, e.g });
            }
        }
        pg.swap(out);
    }
    // The passes are cut from the END of the circuit backwards (the commutation rules are the same
    // in both directions): each pass then holds as many of the latest units as it can, which puts a
    // parameter's rotation into as late a pass as possible -- a parameter-shift sweep propagates one
    // extra state per parameter from the pass of its rotation to the last one.
    std::vector<int> todo(pg.size());
    for (size_t i = 0; i < pg.size(); ++i) todo[i] = (int)(pg.size() - 1 - i);
    // the units of `todo` that a pass on the tile can run (at most cap of them), in order; the others in `rest`
    auto take = [&](const int (&pos_of)[64], int cap, std::vector<TileGate>* taken, std::vector<int>* rest) {
        uint64_t held = 0;
        int n_taken = 0;
        for (int gi : todo) {
            const PG& op = pg[gi];
            const uint64_t qs = ((uint64_t)1 << op.q0) | (op.q1 >= 0 ? (uint64_t)1 << op.q1 : 0);
            const bool ok = !(held & qs) && (diag_kind(op.kind) || pos_of[op.q0] < 32) && n_taken < cap;
            if (!ok) {
                held |= qs;
                if (rest) rest->push_back(gi);
                continue;
            }
            ++n_taken;
            if (taken) taken->push_back(TileGate{ op.kind, pos_of[op.q0], op.q1 >= 0 ? pos_of[op.q1] : -1, op.g });
        }
        return n_taken;
    };
    while (!todo.empty()) {
        // tile: the low qubits plus the first others that pending units need as targets
        const int n_free = kTileBits - kTileLow;
        std::vector<int> tile;
        for (int gi : todo) {
            const PG& op = pg[gi];
            if (diag_kind(op.kind) || op.q0 < kTileLow) continue;
            if (std::find(tile.begin(), tile.end(), op.q0) == tile.end()) tile.push_back(op.q0);
            if ((int)tile.size() == n_free) break;
        }
        for (int q = kTileLow; q < n && (int)tile.size() < n_free; ++q)
            if (std::find(tile.begin(), tile.end(), q) == tile.end()) tile.push_back(q);
        for (int q = 0; q < kTileLow; ++q) tile.push_back(q);
        std::sort(tile.begin(), tile.end());
        int pos_of[64];
        for (int q = 0; q < 64; ++q) pos_of[q] = 32 + q;                 // outer by default
        for (int p = 0; p < kTileBits; ++p) pos_of[tile[p]] = p;
        for (int cap = kGateBatch;; cap /= 2) {
            TilePlan::Pass pass;
            for (int p = 0; p < kTileBits; ++p) pass.tq[p] = tile[p];
            std::vector<int> rest;
            std::vector<TileGate> taken;
            take(pos_of, cap, &taken, &rest);
            std::reverse(taken.begin(), taken.end());   // back into circuit order
            if (!group_sweeps(pass, taken, runs) && cap > 1) continue;
            todo.swap(rest);
            plan.passes.push_back(std::move(pass));
            break;
        }
    }
    std::reverse(plan.passes.begin(), plan.passes.end());
    plan.pass_of_gate.assign(ops.size(), -1);
    plan.first_sweep_of_gate.assign(ops.size(), 0);
    for (size_t k = 0; k < plan.passes.size(); ++k) {
        const auto& p = plan.passes[k];
        plan.soff.push_back(plan.all_sweeps.size());
        const int gbase = (int)plan.all_gates.size(), mbase = (int)plan.all_mats.size(), obase = (int)plan.all_ops.size(),
                  rbase = (int)plan.all_runs.size(), xbase = (int)plan.all_xgates.size();
        for (const MatDesc& md : p.mats) plan.all_mats.push_back(MatDesc{ md.begin + gbase, md.count });
        for (size_t mi = 0; mi < p.mats.size(); ++mi)
            for (int gi = p.mats[mi].begin; gi < p.mats[mi].begin + (p.mats[mi].count & ~kMatPlain); ++gi)
                plan.first_sweep_of_gate[p.gates[gi].g] = p.mat_first[mi];
        for (const RunDesc& rd : p.runs)
            plan.all_runs.push_back(RunDesc{ rd.g1_begin + gbase, rd.g1_count, rd.g2_begin + gbase, rd.g2_count });
        for (TileGate tg : p.gates) {
            if (tg.kind == kGateD1A || tg.kind == kGateD1B) tg.g += mbase;   // the plan's numbering of the fused 2x2s
            else plan.pass_of_gate[tg.g] = (int)k;
            plan.all_gates.push_back(tg);
        }
        for (const TileGate& tg : p.xgates) {
            plan.pass_of_gate[tg.g] = (int)k;
            plan.all_xgates.push_back(tg);
        }
        for (const GroupOp& op : p.ops) {
            plan.all_ops.push_back(op);
            plan.op_mat0.push_back(mbase);
        }
        for (SweepDesc sd : p.sweeps) {
            sd.op_begin += obase;
            for (int j = 0; j < 4; ++j)
                if (sd.rot[j] >= 0) sd.rot[j] += obase;
            if (sd.run >= 0) sd.run += rbase;
            if (sd.next_run >= 0) sd.next_run += rbase;
            sd.x_begin += xbase;
            sd.nx_begin += xbase;
            plan.all_sweeps.push_back(sd);
        }
    }
    return plan;
}

// Which parameters a sweep can serve with ONE extra state each: those that drive exactly
// one rotation (with a non-zero coefficient).  A circuit with a parameter in several
// gates is left to the explicit instance list.
static void plan_sweep(TilePlan& plan, int n_theta, const std::vector<vqa_op>& ops,
                       const std::vector<uint8_t>* gate_dead = nullptr) {
    std::vector<int> count(n_theta, 0), gate(n_theta, -1);
    for (size_t g = 0; g < ops.size(); ++g) {
        const vqa_op& op = ops[g];
        if (gate_dead && (*gate_dead)[g]) continue;
        if (op.kind >= VQA_GATE_RX && op.kind <= VQA_GATE_RZ && op.src == VQA_SRC_THETA) {
            ++count[op.idx];
            gate[op.idx] = (int)g;
        }
    }
    plan.sweepable = true;
    std::vector<int> with;
    for (int i = 0; i < n_theta; ++i) {
        if (count[i] == 0) plan.unused.push_back(i);
        else if (count[i] == 1 && ops[gate[i]].coef != 0.0) with.push_back(i);
        else plan.sweepable = false;
    }
    std::stable_sort(with.begin(), with.end(), [&](int a, int b) {
        const int pa = plan.pass_of_gate[gate[a]], pb = plan.pass_of_gate[gate[b]];
        return pa != pb ? pa < pb : plan.first_sweep_of_gate[gate[a]] < plan.first_sweep_of_gate[gate[b]];
    });
    for (int i : with) {
        plan.ord.push_back(i);
        plan.ord_pass.push_back(plan.pass_of_gate[gate[i]]);
        plan.ord_sweep.push_back(plan.first_sweep_of_gate[gate[i]]);
        plan.ord_coef.push_back(ops[gate[i]].coef);
    }
}

struct ZBatch {
    int nz;
    uint64_t z[kZBatch];
    double coef[kZBatch];
    int col[kZBatch];  // output column (observable) each term adds to
    int bin[kZBatch];  // < 0: Z-string z (a sign); >= 0: the indicator of (index & z) == bin (a marginal bin)
};

__device__ __forceinline__ double block_sum(double v, double* red) {
    const int tid = threadIdx.x;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double r = 0.0;
    if (tid == 0) r = red[0] + red[1] + red[2] + red[3];
    return r;
}

// stage 1: partial[(ci * nb + blk) * kZBatch + t] = sum over the block's slice.
// A block reads one contiguous slice of 2^n / nb amplitudes, a thread every kT-th of them: the
// index bits below log2(kT) are fixed per thread and those from log2(slice) on per block, so an
// entry whose mask avoids the bits in between has ONE sign (or bin test) per thread: it is applied
// once, to the thread's plain sum, and only the other entries cost work per amplitude.
// (NZ = accumulators kept per thread: batches of up to 8 entries run with a quarter of the registers)
template <int NZ>
__global__ __launch_bounds__(kT) void k_zpart(const double2* st, int n, ZBatch zb, double* partial, LaneStride ls) {
    st += ls.st * blockIdx.z;
    partial += ls.part * blockIdx.z;
    __shared__ double red[4];
    const int64_t D = (int64_t)1 << n;
    const int ci = blockIdx.x, blk = blockIdx.y, nb = gridDim.y;
    const double2* s = st + ((int64_t)ci << n);
    double acc[NZ];
#pragma unroll
    for (int t = 0; t < NZ; ++t) acc[t] = 0.0;
    const int64_t L = D / nb;
    const uint32_t vary = (uint32_t)((L - 1) & ~(int64_t)(kT - 1));
    const uint32_t k0 = (uint32_t)(blk * L + threadIdx.x);
    double tot = 0.0;
    for (int64_t k = blk * L + threadIdx.x; k < (blk + 1) * L; k += kT) {
        const double2 v = s[k];
        const double pr = v.x * v.x + v.y * v.y;
        const uint32_t k32 = (uint32_t)k;
        tot += pr;
#pragma unroll
        for (int t = 0; t < NZ; ++t) {
            if (t < zb.nz && ((uint32_t)zb.z[t] & vary)) {
                if (zb.bin[t] >= 0)
                    acc[t] += ((k32 & (uint32_t)zb.z[t]) == (uint32_t)zb.bin[t]) ? pr : 0.0;
                else
                    acc[t] += (__popc(k32 & (uint32_t)zb.z[t]) & 1) ? -pr : pr;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NZ; ++t) {
        if (t < zb.nz) {
            if (!((uint32_t)zb.z[t] & vary)) {
                if (zb.bin[t] >= 0) acc[t] = ((k0 & (uint32_t)zb.z[t]) == (uint32_t)zb.bin[t]) ? tot : 0.0;
                else acc[t] = (__popc(k0 & (uint32_t)zb.z[t]) & 1) ? -tot : tot;
            }
            const double r = block_sum(acc[t], red);
            if (threadIdx.x == 0) partial[((int64_t)ci * nb + blk) * kZBatch + t] = r;
        }
    }
}

// one Pauli string with X/Y factors: Re( i^ny * sum_k (-1)^{popc(k&z)} conj(a[k^x]) a[k] )
__global__ __launch_bounds__(kT) void k_xpart(const double2* st, int n, uint64_t z, uint64_t x,
                                              double* partial, LaneStride ls) {
    st += ls.st * blockIdx.z;
    partial += ls.part * blockIdx.z;
    __shared__ double red[4];
    const int64_t D = (int64_t)1 << n;
    const int ci = blockIdx.y, nb = gridDim.x;
    const double2* s = st + ((int64_t)ci << n);
    const int ny = __popcll(x & z) & 3;
    double acc = 0.0;
    for (int64_t k = (int64_t)blockIdx.x * kT + threadIdx.x; k < D; k += (int64_t)nb * kT) {
        const double2 v = s[k], w = s[k ^ (int64_t)x];
        const double pr = w.x * v.x + w.y * v.y, pim = w.x * v.y - w.y * v.x;
        const double val = ny == 0 ? pr : ny == 1 ? -pim : ny == 2 ? -pr : pim;
        acc += (__popcll((uint64_t)k & z) & 1) ? -val : val;
    }
    const double r = block_sum(acc, red);
    if (threadIdx.x == 0) partial[((int64_t)ci * nb + blockIdx.x) * kZBatch] = r;
}

// stage 2: out[(e_base + ci) * stride + col0 + t] (+)= coef_t * sum_blk partial
// expectation outputs of C instances start at 0 (NaN for an instance that names a sample or
// parameter that does not exist); the finishing kernels add their terms
__global__ __launch_bounds__(kT) void k_out_init(double* out, int64_t e_base, int C, int stride, EvalSpec spec, LaneStride ls) {
    e_base += ls.rows * blockIdx.z;
    const int i = blockIdx.x * kT + threadIdx.x;
    if (i >= C * stride) return;
    const int ci = i / stride;
    int b_, sp_;
    double sv_;
    const bool valid = decode_eval(spec, e_base + ci, b_, sp_, sv_);
    out[(e_base + ci) * stride + (i - ci * stride)] = valid ? 0.0 : __longlong_as_double(0x7ff8000000000000LL);
}

// stage 2: marginal bins (spread): out[(e_base + ci) * stride + col0 + t] = sum_blk partial;
// expectation terms: out[(e_base + ci) * stride + column of t] += coef_t * sum_blk partial
// (entries t0 .. t0 + nz - 1 of the batch)
__global__ __launch_bounds__(kT) void k_finish(const double* partial, int nb, int nz, ZBatch zb,
                                               int use_coef, double xcoef, double* out,
                                               int64_t e_base, int stride, int col0, int spread,
                                               EvalSpec spec, int t0, LaneStride ls) {
    partial += ls.part * blockIdx.z;
    e_base += ls.rows * blockIdx.z;
    __shared__ double red[4];
    const int ci = blockIdx.x;
    bool valid;
    {
        int b_, sp_;
        double sv_;
        valid = decode_eval(spec, e_base + ci, b_, sp_, sv_);
    }
    const double nanv = __longlong_as_double(0x7ff8000000000000LL);
    for (int t = 0; t < nz; ++t) {
        double v = 0.0;
        for (int b = threadIdx.x; b < nb; b += kT) v += partial[((int64_t)ci * nb + b) * kZBatch + t0 + t];
        const double r = block_sum(v, red);
        if (threadIdx.x == 0) {
            if (spread) {  // one output column per t (marginal bins)
                out[(e_base + ci) * stride + col0 + t] = valid ? r : nanv;
            } else {       // NaN from k_out_init stays NaN
                out[(e_base + ci) * stride + (use_coef ? zb.col[t0 + t] : col0)] += (use_coef ? zb.coef[t0 + t] : xcoef) * r;
            }
        }
    }
}

// ---- parameter-shift sweeps ---------------------------------------------------------
// With R(a) = exp(i a/2 P):  R(a +- d) = cos(d/2) R(a) +- sin(d/2) R(a + pi), so the two
// shifted final states of a parameter that drives one rotation are  c psi +- s phi  with
// psi the unshifted final state, phi the final state with that angle moved by pi, and
// (c, s) = (cos, sin)(coef * shift / 2).  For a Hermitian O
//     E+- = c^2 <psi|O|psi> + s^2 <phi|O|phi> +- 2 c s Re <psi|O|phi>,
// and the same holds bin by bin for the marginal.  One propagated state per parameter
// gives both shifted values; it starts from the unshifted state in front of the pass that
// holds the rotation, so the passes before it are shared by every parameter.

// (cos, sin) of the half angles for slots [0, n_slots): slots 0 and 1 unshifted, slot 2 + s
// has the rotation of parameter ord[s] moved by pi
__global__ __launch_bounds__(kT) void k_sweep_angles(HbmCommon h, int64_t j, const int32_t* ord, int n_slots,
                                                     double2* tab, LaneStride ls) {
    j += blockIdx.z;
    tab += ls.tab * blockIdx.z;
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (int64_t)n_slots * h.n_ops) return;
    const int sl = (int)(i / h.n_ops), g = (int)(i - (int64_t)sl * h.n_ops);
    const vqa_op op = h.ops[g];
    double2 cs = make_double2(1.0, 0.0);
    if (op.kind >= VQA_GATE_RX && op.kind <= VQA_GATE_RZ) {
        int b = h.spec.sel ? h.spec.sel[j] : (int)j;
        if (!(h.spec.B == 0 || (unsigned)b < (unsigned)h.spec.B)) b = 0;   // reported as NaN by the finishing kernels
        const int sparam = sl < 2 ? -1 : ord[sl - 2];
        const double sval = op.coef != 0.0 ? 3.14159265358979323846 / op.coef : 0.0;
        const double ang = op_angle(op, h.theta, h.input + (size_t)b * h.n_input, sparam, sval);
        double sn, c;
        sincos(0.5 * ang, &sn, &c);
        cs = make_double2(c, sn);
    }
    tab[i] = cs;
}

// dst[c][k] = src[k] for c < count: one read, count writes
__global__ __launch_bounds__(kT) void k_spawn(double2* dst, const double2* src, int64_t D, int count, LaneStride ls) {
    dst += ls.st * blockIdx.z;
    src += ls.st * blockIdx.z;
    const int64_t k = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (k >= D) return;
    const double2 v = src[k];
    for (int c = 0; c < count; ++c) dst[(int64_t)c * D + k] = v;
}

// stage 1 for state pairs (psi = f, phi = slot ci): per Z-string (or marginal bin) the sums
// of sign |phi|^2 and sign Re(conj(psi) phi); partial[((ci * nb + blk) * kZBatch + t) * 2 + {0, 1}]
// (slices and per-thread signs as in k_zpart; the slot is the fast grid index, so the blocks that
// run side by side read the same slice of psi)
template <int NZ>
__global__ __launch_bounds__(kT) void k_zpart2(const double2* f, const double2* st, int n, ZBatch zb,
                                               double* partial, LaneStride ls) {
    f += ls.st * blockIdx.z;
    st += ls.st * blockIdx.z;
    partial += ls.part * blockIdx.z;
    __shared__ double red[4];
    const int64_t D = (int64_t)1 << n;
    const int ci = blockIdx.x, blk = blockIdx.y, nb = gridDim.y;
    const double2* s = st + ((int64_t)ci << n);
    double acc[NZ], acx[NZ];
#pragma unroll
    for (int t = 0; t < NZ; ++t) { acc[t] = 0.0; acx[t] = 0.0; }
    const int64_t L = D / nb;
    const uint32_t vary = (uint32_t)((L - 1) & ~(int64_t)(kT - 1));
    const uint32_t k0 = (uint32_t)(blk * L + threadIdx.x);
    double tot = 0.0, totx = 0.0;
    for (int64_t k = blk * L + threadIdx.x; k < (blk + 1) * L; k += kT) {
        const double2 v = s[k], u = f[k];
        const double pr = v.x * v.x + v.y * v.y;
        const double cr = u.x * v.x + u.y * v.y;
        const uint32_t k32 = (uint32_t)k;
        tot += pr;
        totx += cr;
#pragma unroll
        for (int t = 0; t < NZ; ++t) {
            if (t < zb.nz && ((uint32_t)zb.z[t] & vary)) {
                if (zb.bin[t] >= 0) {
                    const bool in = (k32 & (uint32_t)zb.z[t]) == (uint32_t)zb.bin[t];
                    acc[t] += in ? pr : 0.0;
                    acx[t] += in ? cr : 0.0;
                } else {
                    const bool neg = __popc(k32 & (uint32_t)zb.z[t]) & 1;
                    acc[t] += neg ? -pr : pr;
                    acx[t] += neg ? -cr : cr;
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NZ; ++t) {
        if (t < zb.nz) {
            if (!((uint32_t)zb.z[t] & vary)) {
                if (zb.bin[t] >= 0) {
                    const bool in = (k0 & (uint32_t)zb.z[t]) == (uint32_t)zb.bin[t];
                    acc[t] = in ? tot : 0.0;
                    acx[t] = in ? totx : 0.0;
                } else {
                    const bool neg = __popc(k0 & (uint32_t)zb.z[t]) & 1;
                    acc[t] = neg ? -tot : tot;
                    acx[t] = neg ? -totx : totx;
                }
            }
            const double r = block_sum(acc[t], red);
            const double rx = block_sum(acx[t], red);
            if (threadIdx.x == 0) {
                double* o = &partial[(((int64_t)ci * nb + blk) * kZBatch + t) * 2];
                o[0] = r;
                o[1] = rx;
            }
        }
    }
}

// one Pauli string with X/Y factors for state pairs: <phi|P|phi> and Re <psi|P|phi>
__global__ __launch_bounds__(kT) void k_xpart2(const double2* f, const double2* st, int n, uint64_t z,
                                               uint64_t x, double* partial, LaneStride ls) {
    f += ls.st * blockIdx.z;
    st += ls.st * blockIdx.z;
    partial += ls.part * blockIdx.z;
    __shared__ double red[4];
    const int64_t D = (int64_t)1 << n;
    const int ci = blockIdx.y, nb = gridDim.x;
    const double2* s = st + ((int64_t)ci << n);
    const int ny = __popcll(x & z) & 3;
    double acc = 0.0, acx = 0.0;
    for (int64_t k = (int64_t)blockIdx.x * kT + threadIdx.x; k < D; k += (int64_t)nb * kT) {
        const double2 v = s[k], w = s[k ^ (int64_t)x], u = f[k ^ (int64_t)x];
        const double pr = w.x * v.x + w.y * v.y, pim = w.x * v.y - w.y * v.x;
        const double qr = u.x * v.x + u.y * v.y, qim = u.x * v.y - u.y * v.x;
        const double val = ny == 0 ? pr : ny == 1 ? -pim : ny == 2 ? -pr : pim;
        const double vax = ny == 0 ? qr : ny == 1 ? -qim : ny == 2 ? -qr : qim;
        const bool neg = __popcll((uint64_t)k & z) & 1;
        acc += neg ? -val : val;
        acx += neg ? -vax : vax;
    }
    const double r = block_sum(acc, red);
    const double rx = block_sum(acx, red);
    if (threadIdx.x == 0) {
        double* o = &partial[(((int64_t)ci * nb + blockIdx.x) * kZBatch) * 2];
        o[0] = r;
        o[1] = rx;
    }
}

// expectation rows 2i+1 (+) and 2i+2 (-) of a sample start at c^2 * (row 0 of the sample), i = ord[ci]
__global__ __launch_bounds__(kT) void k_out_init2(double* out, int64_t row0, const int32_t* ord, const double* ord_coef,
                                                  double shift, int C, int stride, LaneStride ls) {
    row0 += ls.rows * blockIdx.z;
    const int i = blockIdx.x * kT + threadIdx.x;
    if (i >= C * stride) return;
    const int ci = i / stride, col = i - ci * stride;
    const double c = cos(0.5 * ord_coef[ci] * shift);
    const int64_t rp = row0 + 2 * (int64_t)ord[ci] + 1;
    const double base = c * c * out[row0 * stride + col];
    out[rp * stride + col] = base;
    out[(rp + 1) * stride + col] = base;
}

// stage 2 for state pairs: rows 2i+1 (+) and 2i+2 (-) of sample j, i = ord[ci]:
//   marginal bins (spread):  c^2 * (row 0 of the sample) + s^2 A +- 2 c s X
//   expectation terms:       += coef (s^2 A +- 2 c s X) in the term's column
__global__ __launch_bounds__(kT) void k_finish2(const double* partial, int nb, int nz, ZBatch zb, int use_coef,
                                                double xcoef, double* out, int64_t row0, const int32_t* ord,
                                                const double* ord_coef, double shift, int stride, int col0,
                                                int spread, int t0, LaneStride ls) {
    partial += ls.part * blockIdx.z;
    row0 += ls.rows * blockIdx.z;
    __shared__ double red[4];
    const int ci = blockIdx.x;
    double sn, c;
    sincos(0.5 * ord_coef[ci] * shift, &sn, &c);
    const int64_t rp = row0 + 2 * (int64_t)ord[ci] + 1, rm = rp + 1;
    for (int t = 0; t < nz; ++t) {
        double va = 0.0, vx = 0.0;
        for (int b = threadIdx.x; b < nb; b += kT) {
            const double* q = &partial[(((int64_t)ci * nb + b) * kZBatch + t0 + t) * 2];
            va += q[0];
            vx += q[1];
        }
        const double a = block_sum(va, red);
        const double x = block_sum(vx, red);
        if (threadIdx.x == 0) {
            if (spread) {
                const double base = c * c * out[row0 * stride + col0 + t];
                out[rp * stride + col0 + t] = base + sn * sn * a + 2.0 * c * sn * x;
                out[rm * stride + col0 + t] = base + sn * sn * a - 2.0 * c * sn * x;
            } else {
                const double cf = use_coef ? zb.coef[t0 + t] : xcoef;
                const int col = use_coef ? zb.col[t0 + t] : col0;
                out[rp * stride + col] += cf * (sn * sn * a + 2.0 * c * sn * x);
                out[rm * stride + col] += cf * (sn * sn * a - 2.0 * c * sn * x);
            }
        }
    }
}

// parameters that drive no gate: both shifted rows equal the unshifted one
__global__ __launch_bounds__(kT) void k_copy_base(double* out, int64_t row0, const int32_t* unused, int n_unused,
                                                  int stride, LaneStride ls) {
    row0 += ls.rows * blockIdx.z;
    const int i = blockIdx.x * kT + threadIdx.x;
    if (i >= n_unused * stride) return;
    const int u = i / stride, col = i - u * stride;
    const double v = out[row0 * stride + col];
    const int64_t rp = row0 + 2 * (int64_t)unused[u] + 1;
    out[rp * stride + col] = v;
    out[(rp + 1) * stride + col] = v;
}

const TilePlan& plan_of(const EvalCall& call) {
    std::lock_guard<std::mutex> lk(call.c->mu);
    if (!call.c->hbm_plan) {
        auto plan = std::make_shared<TilePlan>(plan_tiles(call.c->n_qubits, call.c->ops));
        plan_sweep(*plan, call.c->n_theta, call.c->ops);
        if (getenv("VQA_PLAN_DEBUG")) {
            fprintf(stderr, "[vqa] tiled plan: %zu passes for %zu gates\n", plan->passes.size(), call.c->ops.size());
            for (size_t k = 0; k < plan->passes.size(); ++k) {
                const auto& p = plan->passes[k];
                size_t nd = 0;
                for (const auto& sd : p.sweeps) nd += sd.run >= 0;
                fprintf(stderr, "[vqa]   pass %2zu: %3zu gates, %2zu sweeps (%zu with a diagonal run), %2zu fused 2x2, %3zu register ops\n",
                        k, p.gates.size(), p.sweeps.size(), nd, p.mats.size(), p.ops.size());
            }
        }
        call.c->hbm_plan = plan;
    }
    return *static_cast<const TilePlan*>(call.c->hbm_plan.get());
}

// The plan of the circuit without the gates outside the light cone of what the call measures
// (parameter-shift sweeps; the circuit's own plan when there is nothing to leave out)
const TilePlan& cone_plan_of(const EvalCall& call) {
    ConeInfo& cone = *call.cone;
    if (cone.n_gate_dead == 0) return plan_of(call);
    std::lock_guard<std::mutex> lk(call.c->mu);
    if (!cone.hbm_plan) {
        auto plan = std::make_shared<TilePlan>(plan_tiles(call.c->n_qubits, call.c->ops, &cone.gate_dead));
        plan_sweep(*plan, call.c->n_theta, call.c->ops, &cone.gate_dead);
        if (getenv("VQA_PLAN_DEBUG"))
            fprintf(stderr, "[vqa] tiled plan without %d gates outside the light cone: %zu passes\n", cone.n_gate_dead,
                    plan->passes.size());
        cone.hbm_plan = plan;
    }
    return *static_cast<const TilePlan*>(cone.hbm_plan.get());
}

// The plan of a layer of Hadamards on the qubits of `mask`: it turns X factors of Pauli
// strings into Z factors, so that many X-type terms are measured in one pass over the state.
const TilePlan& hplan_of(const EvalCall& call, uint64_t mask) {
    std::lock_guard<std::mutex> lk(call.c->mu);
    auto it = call.c->hbm_hplans.find(mask);
    if (it == call.c->hbm_hplans.end()) {
        std::vector<vqa_op> ops;
        for (int q = 0; q < call.c->n_qubits; ++q)
            if ((mask >> q) & 1) ops.push_back(vqa_op{ VQA_GATE_H, q, -1, VQA_SRC_CONST, 0, 0, 0.0, 0.0 });
        it = call.c->hbm_hplans.emplace(mask, std::make_shared<TilePlan>(plan_tiles(call.c->n_qubits, ops))).first;
    }
    return *static_cast<const TilePlan*>(it->second.get());
}

bool tiled(const EvalCall& call) { return call.c->n_qubits >= kTileBits + 1 && call.c->engine != 4; }

// The qubits of the Hadamard layer for this call's observables: the union of the X masks of
// the pure-X terms, when there are enough of them to pay for the extra passes.
uint64_t hlayer_mask(const EvalCall& call) {
    if (!tiled(call)) return 0;
    uint64_t mask = 0;
    int count = 0;
    for (int t = 0; t < call.obs.n_terms; ++t)
        if (call.obs.x[t] != 0 && call.obs.z[t] == 0) { mask |= call.obs.x[t]; ++count; }
    return count >= 3 ? mask : 0;
}

// device images of a plan
struct PlanImage {
    TileGate* gates;
    TileGate* xgates;
    GroupOp* ops;
    int32_t* op_mat0;
    MatDesc* mats;
    RunDesc* runs;
    SweepDesc* sweeps;
};
static size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }
template <class T>
static size_t vec_bytes(const std::vector<T>& v) { return pad256(std::max<size_t>(v.size(), 1) * sizeof(T)); }

size_t plan_image_bytes(const TilePlan& p) {
    return vec_bytes(p.all_gates) + vec_bytes(p.all_xgates) + vec_bytes(p.all_ops) + vec_bytes(p.op_mat0) +
           vec_bytes(p.all_mats) + vec_bytes(p.all_runs) + vec_bytes(p.all_sweeps);
}

PlanImage plan_image(const TilePlan& p, void* buf) {
    char* w = (char*)buf;
    PlanImage im;
    im.gates = (TileGate*)w;    w += vec_bytes(p.all_gates);
    im.xgates = (TileGate*)w;   w += vec_bytes(p.all_xgates);
    im.ops = (GroupOp*)w;       w += vec_bytes(p.all_ops);
    im.op_mat0 = (int32_t*)w;   w += vec_bytes(p.op_mat0);
    im.mats = (MatDesc*)w;      w += vec_bytes(p.all_mats);
    im.runs = (RunDesc*)w;      w += vec_bytes(p.all_runs);
    im.sweeps = (SweepDesc*)w;
    return im;
}

// A plan with its device images and the per-instance tables of the instances it runs on.
struct PlanDev {
    const TilePlan* plan = nullptr;
    void* buf = nullptr;       // plan_image_bytes(*plan)
    char* inst = nullptr;      // inst_bytes(instances): mats | oprec | pretab | xrec | swrec, each [instances][row]
    int64_t cap = 0;           // instances the tables are carved for
    int64_t lane_inst = 0;     // bytes from one lane's tables to the next lane's (launches with lanes)
    size_t n_mats() const { return std::max<size_t>(plan->all_mats.size(), 1); }
    size_t n_ops() const { return std::max<size_t>(plan->all_ops.size(), 1); }
    size_t n_runs() const { return std::max<size_t>(plan->all_runs.size(), 1); }
    size_t n_x() const { return std::max<size_t>(plan->all_xgates.size(), 1); }
    size_t image_bytes() const { return plan_image_bytes(*plan); }
    size_t mats_bytes(int64_t C) const { return pad256((size_t)C * n_mats() * 4 * sizeof(double2)); }
    size_t ops_bytes(int64_t C) const { return pad256((size_t)C * n_ops() * sizeof(OpRec)); }
    size_t tab_bytes(int64_t C) const { return pad256((size_t)C * n_runs() * 128 * sizeof(double2)); }
    size_t x_bytes(int64_t C) const { return pad256((size_t)C * n_x() * sizeof(XRec)); }
    size_t n_sw() const { return std::max<size_t>(plan->all_sweeps.size(), 1); }
    size_t sw_bytes(int64_t C) const { return pad256((size_t)C * n_sw() * sizeof(SweepRec)); }
    size_t inst_bytes(int64_t C) const { return mats_bytes(C) + ops_bytes(C) + tab_bytes(C) + x_bytes(C) + sw_bytes(C); }
    void carve(char* p, int64_t C) { inst = p; cap = C; }
    double2* mats(int64_t slot) const { return (double2*)inst + slot * n_mats() * 4; }
    OpRec* oprec(int64_t slot) const { return (OpRec*)(inst + mats_bytes(cap)) + slot * n_ops(); }
    double2* pretab(int64_t slot) const { return (double2*)(inst + mats_bytes(cap) + ops_bytes(cap)) + slot * n_runs() * 128; }
    XRec* xrec(int64_t slot) const { return (XRec*)(inst + mats_bytes(cap) + ops_bytes(cap) + tab_bytes(cap)) + slot * n_x(); }
    SweepRec* swrec(int64_t slot) const {
        return (SweepRec*)(inst + mats_bytes(cap) + ops_bytes(cap) + tab_bytes(cap) + x_bytes(cap)) + slot * n_sw();
    }
};

// the strides of a launch that reads or writes the per-instance tables of pd
inline LaneStride lanes_of(const PlanDev& pd) {
    LaneStride ls = t_lanes.ls;
    ls.inst = pd.lane_inst;
    return ls;
}

// copies the plan's tables (a plan lives as long as its circuit, so the asynchronous
// copies may read its host images after this returns)
int upload_plan(const EvalCall& call, const PlanDev& pd) {
    const TilePlan& plan = *pd.plan;
    const PlanImage im = plan_image(plan, pd.buf);
    auto up = [&](void* dst, const void* src, size_t bytes) {
        return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, call.stream) : hipSuccess;
    };
    VQA_HIP(up(im.gates, plan.all_gates.data(), plan.all_gates.size() * sizeof(TileGate)));
    VQA_HIP(up(im.xgates, plan.all_xgates.data(), plan.all_xgates.size() * sizeof(TileGate)));
    VQA_HIP(up(im.ops, plan.all_ops.data(), plan.all_ops.size() * sizeof(GroupOp)));
    VQA_HIP(up(im.op_mat0, plan.op_mat0.data(), plan.op_mat0.size() * sizeof(int32_t)));
    VQA_HIP(up(im.mats, plan.all_mats.data(), plan.all_mats.size() * sizeof(MatDesc)));
    VQA_HIP(up(im.runs, plan.all_runs.data(), plan.all_runs.size() * sizeof(RunDesc)));
    VQA_HIP(up(im.sweeps, plan.all_sweeps.data(), plan.all_sweeps.size() * sizeof(SweepDesc)));
    for (const void* fn : { (const void*)k_tile_pass<0, 0>, (const void*)k_tile_pass<1, 0>, (const void*)k_tile_pass<0, 1>,
                            (const void*)k_tile_pass<1, 1> })
        VQA_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kTileAmps * (int)sizeof(double2)));
    return VQA_OK;
}

// the per-instance tables of C instances (table rows slot0 ..) from their angle tables (tab_stride entries each)
void launch_mats(const EvalCall& call, const PlanDev& pd, const double2* tab, int tab_stride, int slot0, int C) {
    const TilePlan& plan = *pd.plan;
    const int M = (int)plan.all_mats.size(), NO = (int)plan.all_ops.size(), NR = (int)plan.all_runs.size(),
              NX = (int)plan.all_xgates.size();
    if (C <= 0) return;
    const PlanImage im = plan_image(plan, pd.buf);
    auto grid = [](int64_t tot) { return dim3((unsigned)((tot + kT - 1) / kT)); };
    if (M > 0)
        hipLaunchKernelGGL(k_mats, lane_grid(grid((int64_t)C * M)), dim3(kT), 0, call.stream, im.gates, im.mats, M, tab, tab_stride, C,
                           pd.mats(slot0), lanes_of(pd));
    if (NO > 0)
        hipLaunchKernelGGL(k_prep_ops, lane_grid(grid((int64_t)C * NO)), dim3(kT), 0, call.stream, im.ops, im.op_mat0, NO, C,
                           pd.mats(slot0), (int)pd.n_mats(), pd.oprec(slot0), lanes_of(pd));
    if (NR > 0)
        hipLaunchKernelGGL(k_prep_tab, lane_grid(dim3(NR, C)), dim3(128), 0, call.stream, im.gates, im.runs, NR, tab, tab_stride,
                           pd.mats(slot0), (int)pd.n_mats(), pd.pretab(slot0), lanes_of(pd));
    if (NX > 0)
        hipLaunchKernelGGL(k_prep_x, lane_grid(grid((int64_t)C * NX)), dim3(kT), 0, call.stream, im.xgates, NX, C, tab, tab_stride,
                           pd.xrec(slot0), lanes_of(pd));
    const int NS = (int)plan.all_sweeps.size();
    if (NS > 0)
        hipLaunchKernelGGL(k_prep_sweeps, lane_grid(grid((int64_t)C * NS)), dim3(kT), 0, call.stream, im.sweeps, NS, C, pd.oprec(slot0),
                           (int)pd.n_ops(), pd.swrec(slot0), lanes_of(pd));
}

thread_local int64_t t_state_passes = 0;   // states x passes launched by this thread (VQA_PLAN_DEBUG report)

// pass k of the plan on C states side by side (st; table rows slot0 .. of the per-instance tables);
// io (may be null): other sources / destinations of some states, a measurement at the end
void launch_pass(const EvalCall& call, const PlanDev& pd, size_t k, double2* st, int slot0, int C, const PassIO* io = nullptr) {
    const TilePlan& plan = *pd.plan;
    const auto& p = plan.passes[k];
    if (p.sweeps.empty() || C <= 0) return;
    t_state_passes += (int64_t)C * t_lanes.n;
    if (io && io->partial && !io->write_state) count_state_passes(0, (int64_t)C * t_lanes.n);
    else count_state_passes((int64_t)C * t_lanes.n, 0);
    const int n = call.c->n_qubits;
    TilePassArgs a;
    a.n = n;
    a.n_outer = 0;
    for (int q = 0; q < n; ++q) {
        bool in_tile = false;
        for (int t = 0; t < kTileBits; ++t) in_tile |= p.tq[t] == q;
        if (!in_tile) a.oq[a.n_outer++] = q;
    }
    for (int t = 0; t < kTileBits; ++t) a.tq[t] = p.tq[t];
    a.n_sweeps = (int)p.sweeps.size();
    a.n_ops_total = (int)pd.n_ops();
    a.n_runs_total = (int)pd.n_runs();
    a.n_x_total = (int)pd.n_x();
    a.n_sweeps_total = (int)pd.n_sw();
    a.sweep_off = (int)plan.soff[k];
    const int64_t n_tiles = ((int64_t)1 << n) >> kTileBits;
#ifdef VQA_TILE_ABLATE
    a.ablate = getenv("VQA_TILE_ABLATE") ? atoi(getenv("VQA_TILE_ABLATE")) : 0;
#endif
    PassIO pio;
    if (io) {
        pio = *io;
    } else {
        memset(&pio, 0, sizeof(pio));
        pio.alt_from = 1 << 30;
    }
    // (large registers only: a small common state stays in the caches whatever the order, and with
    // the tiles first all workgroups in flight read the same instance's records)
    pio.n_states = C;
    pio.state_first = (pio.F || pio.alt_from < C) && n >= 24;
    const dim3 grid((unsigned)(n_tiles * C));
    const bool lk = t_lanes.n > 1;
    auto* kern = pio.nz > 0 ? (lk ? k_tile_pass<1, 1> : k_tile_pass<1, 0>)
                            : (lk ? k_tile_pass<0, 1> : k_tile_pass<0, 0>);
    hipLaunchKernelGGL(kern, lane_grid(grid), dim3(kTT), kTileAmps * sizeof(double2), call.stream, st, pd.swrec(slot0),
                       pd.oprec(slot0), pd.pretab(slot0), pd.xrec(slot0), a, pio, lanes_of(pd));
}

void launch_all_passes(const EvalCall& call, const PlanDev& pd, double2* st, int slot0, int C) {
    for (size_t k = 0; k < pd.plan->passes.size(); ++k) launch_pass(call, pd, k, st, slot0, C);
}

// |0..0>, the angle table and every gate for the h.C instances of a chunk; pd.plan == nullptr: one sweep per gate
int launch_gates(const EvalCall& call, const HbmCommon& h, double2* st, double2* tab, const PlanDev& pd) {
    const int n = h.n, C = h.C;
    const int64_t D = (int64_t)1 << n;
    const int64_t tot = D * C;
    hipLaunchKernelGGL(k_init, lane_grid(dim3((unsigned)((tot + kT - 1) / kT))), dim3(kT), 0, call.stream, st,
                       tot, D, t_lanes.ls);
    const int64_t na = (int64_t)C * h.n_ops;
    if (na > 0)
        hipLaunchKernelGGL(k_angles, lane_grid(dim3((unsigned)((na + kT - 1) / kT))), dim3(kT), 0, call.stream,
                           h, tab, t_lanes.ls);
    if (pd.plan) {
        launch_mats(call, pd, tab, h.n_ops, 0, C);
        launch_all_passes(call, pd, st, 0, C);
        VQA_HIP(hipGetLastError());
        return VQA_OK;
    }
    for (int g = 0; g < h.n_ops; ++g) {
        const vqa_op& op = call.c->ops[g];
        if (op.kind == VQA_GATE_RZ || op.kind == VQA_GATE_Z || op.kind == VQA_GATE_CZ) {
            hipLaunchKernelGGL(k_diag, dim3((unsigned)((tot + kT - 1) / kT)), dim3(kT), 0,
                               call.stream, st, n, C, op.kind, op.q0, op.q1, tab, h.n_ops, g);
        } else {
            const int64_t np = tot / 2;
            hipLaunchKernelGGL(k_pair, dim3((unsigned)((np + kT - 1) / kT)), dim3(kT), 0,
                               call.stream, st, n, C, op.kind, op.q0, op.q1, tab, h.n_ops, g);
        }
    }
    VQA_HIP(hipGetLastError());
    return VQA_OK;
}

// Observables and marginal of C states side by side -> output rows e0 .. e0 + C - 1.
// pair == nullptr: plain expectation values (k_zpart / k_xpart / k_finish).
// pair != nullptr: the states are the pi-moved states of a sweep, pair->f the unshifted
// final state; writes the two shifted rows of every parameter (k_*2 kernels).
// hl.plan != nullptr: the pure-X terms are measured last, as Z-strings behind a layer of
// Hadamards that is applied to the states in place (plain: to the copy hl.alt of the one
// state when that is given; pairs: pair->fh is the transformed unshifted state).
struct PairArgs {
    const double2* f;
    const double2* fh;
    int64_t row0;
    const int32_t* ord;
    const double* ord_coef;
    double shift;
};
struct HLayer {
    PlanDev pd;
    uint64_t mask = 0;
    double2* alt = nullptr;
};

// What the tiled passes of a sweep measure themselves (pairs only; see fuse_plan): the one batch of
// Z-strings (with the marginal bins) at the end of the circuit's last pass, the one batch of pure-X
// strings at the end of the Hadamard layer's last pass.  Their partial sums come one per tile.
struct FuseInfo {
    bool main = false, h = false;
    bool main_write = true;   // the circuit's last pass still writes the final states (something else reads them)
    int n_tiles = 0;
    // Every pure-X string lies within the qubits one pass of the Hadamard layer transforms: the passes
    // then run side by side on the final states, each measuring its strings against the unshifted state
    // behind that pass alone (hl.alt[k]), and none of them writes a state.
    bool h_indep = false;
    std::vector<int> term_hpass;   // per term: the pass of the layer that measures it, -1: none
};

int measure_states(const EvalCall& call, double2* st, int C, int64_t e0, int nb, double* partial,
                   const PairArgs* pair, const HLayer& hl, const FuseInfo* fu = nullptr) {
    const int n = call.c->n_qubits;
    const int n_obs = call.obs.n_obs;
    if (n_obs > 0) {
        const unsigned g = (unsigned)((C * n_obs + kT - 1) / kT);
        if (pair)
            hipLaunchKernelGGL(k_out_init2, lane_grid(dim3(g)), dim3(kT), 0, call.stream, call.d_out_expect, pair->row0, pair->ord,
                               pair->ord_coef, pair->shift, C, n_obs, t_lanes.ls);
        else
            hipLaunchKernelGGL(k_out_init, lane_grid(dim3(g)), dim3(kT), 0, call.stream, call.d_out_expect, e0, C, n_obs, call.spec, t_lanes.ls);
    }
    // Z-strings of all observables, kZBatch at a time; `f` = the unshifted state the pairs are taken with.
    // The bins of the low-bit marginal ride in the same pass over the states when they fit behind
    // the last Z-strings (entries nzs .. nz - 1 of the batch).
    const int M = call.n_marg_bits > 0 ? 1 << call.n_marg_bits : 0;
    bool marg_done = M == 0;
    auto zflush = [&](ZBatch& zb, const double2* states, const double2* f, bool last, bool fused = false) {
        const int nb_unfused = nb;
        const int nb = fused ? fu->n_tiles : nb_unfused;
        const int nzs = zb.nz;
        int nbins = 0;
        if (last && !marg_done && nzs + M <= kZBatch) {
            for (int m = 0; m < M; ++m) {
                zb.z[nzs + m] = (uint64_t)(M - 1);
                zb.bin[nzs + m] = m;
                zb.coef[nzs + m] = 0.0;
                zb.col[nzs + m] = 0;
            }
            nbins = M;
            zb.nz = nzs + M;
            marg_done = true;
        }
        if (zb.nz == 0) return;
        if (pair) {
            if (fused) {}
            else if (zb.nz <= 8) hipLaunchKernelGGL(k_zpart2<8>, lane_grid(dim3(C, nb)), dim3(kT), 0, call.stream, f, states, n, zb, partial, t_lanes.ls);
            else hipLaunchKernelGGL(k_zpart2<kZBatch>, lane_grid(dim3(C, nb)), dim3(kT), 0, call.stream, f, states, n, zb, partial, t_lanes.ls);
            if (nzs)
                hipLaunchKernelGGL(k_finish2, lane_grid(dim3(C)), dim3(kT), 0, call.stream, partial, nb, nzs, zb, 1, 0.0,
                                   call.d_out_expect, pair->row0, pair->ord, pair->ord_coef, pair->shift, n_obs, 0, 0, 0, t_lanes.ls);
            if (nbins)
                hipLaunchKernelGGL(k_finish2, lane_grid(dim3(C)), dim3(kT), 0, call.stream, partial, nb, nbins, zb, 1, 0.0,
                                   call.d_out_marginal, pair->row0, pair->ord, pair->ord_coef, pair->shift, M, 0, 1, nzs, t_lanes.ls);
        } else {
            if (zb.nz <= 8) hipLaunchKernelGGL(k_zpart<8>, lane_grid(dim3(C, nb)), dim3(kT), 0, call.stream, states, n, zb, partial, t_lanes.ls);
            else hipLaunchKernelGGL(k_zpart<kZBatch>, lane_grid(dim3(C, nb)), dim3(kT), 0, call.stream, states, n, zb, partial, t_lanes.ls);
            if (nzs)
                hipLaunchKernelGGL(k_finish, lane_grid(dim3(C)), dim3(kT), 0, call.stream, partial, nb, nzs, zb, 1, 0.0,
                                   call.d_out_expect, e0, n_obs, 0, 0, call.spec, 0, t_lanes.ls);
            if (nbins)
                hipLaunchKernelGGL(k_finish, lane_grid(dim3(C)), dim3(kT), 0, call.stream, partial, nb, nbins, zb, 0, 0.0,
                                   call.d_out_marginal, e0, M, 0, 1, call.spec, nzs, t_lanes.ls);
        }
        zb.nz = 0;
    };
    ZBatch zb;
    zb.nz = 0;
    for (int o = 0; o < n_obs; ++o) {
        for (int t = call.obs.off[o]; t < call.obs.off[o + 1]; ++t) {
            if (call.obs.x[t] != 0) continue;
            zb.z[zb.nz] = call.obs.z[t];
            zb.coef[zb.nz] = call.obs.coef[t];
            zb.col[zb.nz] = o;
            zb.bin[zb.nz] = -1;
            if (++zb.nz == kZBatch && !(fu && fu->main)) zflush(zb, st, pair ? pair->f : nullptr, false);
        }
    }
    zflush(zb, st, pair ? pair->f : nullptr, true, fu && fu->main);
    // X/Y strings one at a time, except the pure-X ones when the Hadamard layer is in use
    for (int o = 0; o < n_obs; ++o) {
        for (int t = call.obs.off[o]; t < call.obs.off[o + 1]; ++t) {
            if (call.obs.x[t] == 0 || (hl.pd.plan && call.obs.z[t] == 0)) continue;
            if (pair) {
                hipLaunchKernelGGL(k_xpart2, lane_grid(dim3(nb, C)), dim3(kT), 0, call.stream, pair->f, st, n,
                                   call.obs.z[t], call.obs.x[t], partial, t_lanes.ls);
                hipLaunchKernelGGL(k_finish2, lane_grid(dim3(C)), dim3(kT), 0, call.stream, partial, nb, 1, zb, 0,
                                   call.obs.coef[t], call.d_out_expect, pair->row0, pair->ord,
                                   pair->ord_coef, pair->shift, n_obs, o, 0, 0, t_lanes.ls);
            } else {
                hipLaunchKernelGGL(k_xpart, lane_grid(dim3(nb, C)), dim3(kT), 0, call.stream, st, n,
                                   call.obs.z[t], call.obs.x[t], partial, t_lanes.ls);
                hipLaunchKernelGGL(k_finish, lane_grid(dim3(C)), dim3(kT), 0, call.stream, partial, nb,
                                   1, zb, 0, call.obs.coef[t], call.d_out_expect, e0, n_obs, o, 0, call.spec, 0, t_lanes.ls);
            }
        }
    }
    if (!marg_done) {
        for (int m0 = 0; m0 < M; m0 += kZBatch) {
            ZBatch mb;
            mb.nz = std::min(kZBatch, M - m0);
            for (int m = 0; m < mb.nz; ++m) {
                mb.z[m] = (uint64_t)(M - 1);
                mb.bin[m] = m0 + m;
                mb.coef[m] = 0.0;
                mb.col[m] = 0;
            }
            if (pair) {
                if (mb.nz <= 8) hipLaunchKernelGGL(k_zpart2<8>, lane_grid(dim3(C, nb)), dim3(kT), 0, call.stream, pair->f, st, n, mb, partial, t_lanes.ls);
                else hipLaunchKernelGGL(k_zpart2<kZBatch>, lane_grid(dim3(C, nb)), dim3(kT), 0, call.stream, pair->f, st, n, mb, partial, t_lanes.ls);
                hipLaunchKernelGGL(k_finish2, lane_grid(dim3(C)), dim3(kT), 0, call.stream, partial, nb, mb.nz, mb, 1, 0.0,
                                   call.d_out_marginal, pair->row0, pair->ord, pair->ord_coef, pair->shift, M, m0, 1, 0, t_lanes.ls);
            } else {
                if (mb.nz <= 8) hipLaunchKernelGGL(k_zpart<8>, lane_grid(dim3(C, nb)), dim3(kT), 0, call.stream, st, n, mb, partial, t_lanes.ls);
                else hipLaunchKernelGGL(k_zpart<kZBatch>, lane_grid(dim3(C, nb)), dim3(kT), 0, call.stream, st, n, mb, partial, t_lanes.ls);
                hipLaunchKernelGGL(k_finish, lane_grid(dim3(C)), dim3(kT), 0, call.stream, partial, nb, mb.nz, mb, 0, 0.0,
                                   call.d_out_marginal, e0, M, m0, 1, call.spec, 0, t_lanes.ls);
            }
        }
    }
    if (hl.pd.plan) {
        // H X H = Z: behind the Hadamard layer the pure-X strings are Z-strings with z = x
        double2* hs = st;
        if (!pair && hl.alt) {
            const int64_t D = (int64_t)1 << n;
            hipLaunchKernelGGL(k_spawn, lane_grid(dim3((unsigned)((D + kT - 1) / kT))), dim3(kT), 0, call.stream, hl.alt, st, D, 1, t_lanes.ls);
            hs = hl.alt;
        }
        const bool hf = fu && fu->h;
        launch_mats(call, hl.pd, nullptr, 0, 0, C);
        if (hf && fu->h_indep) {
            const int64_t D = (int64_t)1 << n;
            for (size_t k = 0; k < hl.pd.plan->passes.size(); ++k) {
                PassIO io;
                memset(&io, 0, sizeof(io));
                io.alt_from = 1 << 30;
                io.F = pair->fh + (int64_t)k * D;
                io.partial = partial;
                zb.nz = 0;
                for (int o = 0; o < n_obs; ++o) {
                    for (int t = call.obs.off[o]; t < call.obs.off[o + 1]; ++t) {
                        if (fu->term_hpass[t] != (int)k) continue;
                        io.z[io.nz] = (uint32_t)call.obs.x[t];
                        io.bin[io.nz++] = -1;
                        zb.z[zb.nz] = call.obs.x[t];
                        zb.coef[zb.nz] = call.obs.coef[t];
                        zb.col[zb.nz] = o;
                        zb.bin[zb.nz++] = -1;
                    }
                }
                if (zb.nz == 0) continue;
                launch_pass(call, hl.pd, k, st, 0, C, &io);
                zflush(zb, st, pair->fh, false, true);
            }
            VQA_HIP(hipGetLastError());
            return VQA_OK;
        }
        if (hf) {
            // the layer's last pass measures the strings and leaves the states unwritten
            const size_t np = hl.pd.plan->passes.size();
            for (size_t k = 0; k + 1 < np; ++k) launch_pass(call, hl.pd, k, hs, 0, C);
            PassIO io;
            memset(&io, 0, sizeof(io));
            io.alt_from = 1 << 30;
            io.F = pair->fh;
            io.partial = partial;
            for (int t = 0; t < call.obs.n_terms; ++t) {
                if (call.obs.x[t] == 0 || call.obs.z[t] != 0) continue;
                io.z[io.nz] = (uint32_t)call.obs.x[t];
                io.bin[io.nz++] = -1;
            }
            launch_pass(call, hl.pd, np - 1, hs, 0, C, &io);
        } else {
            launch_all_passes(call, hl.pd, hs, 0, C);
        }
        zb.nz = 0;
        for (int o = 0; o < n_obs; ++o) {
            for (int t = call.obs.off[o]; t < call.obs.off[o + 1]; ++t) {
                if (call.obs.x[t] == 0 || call.obs.z[t] != 0) continue;
                zb.z[zb.nz] = call.obs.x[t];
                zb.coef[zb.nz] = call.obs.coef[t];
                zb.col[zb.nz] = o;
                zb.bin[zb.nz] = -1;
                if (++zb.nz == kZBatch && !hf) zflush(zb, hs, pair ? pair->fh : nullptr, false);
            }
        }
        zflush(zb, hs, pair ? pair->fh : nullptr, false, hf);
    }
    VQA_HIP(hipGetLastError());
    return VQA_OK;
}

// What the passes of a sweep can measure themselves.  io (filled when fu->main): the one batch of
// Z-strings and marginal bins, in the order measure_states() puts them together.
void fuse_plan(const EvalCall& call, const HLayer& hl, int64_t n_tiles, FuseInfo* fu, PassIO* io) {
    *fu = FuseInfo();
    fu->n_tiles = (int)n_tiles;
    if (getenv("VQA_NO_FUSED_MEASURE") || n_tiles > (1 << 18)) return;
    const int M = call.n_marg_bits > 0 ? 1 << call.n_marg_bits : 0;
    int nz = 0, nx = 0, nmixed = 0;
    for (int t = 0; t < call.obs.n_terms; ++t) {
        if (call.obs.x[t] == 0) {
            if (nz < kZBatch) {
                io->z[nz] = (uint32_t)call.obs.z[t];
                io->bin[nz] = -1;
            }
            ++nz;
        } else if (hl.pd.plan && call.obs.z[t] == 0) {
            ++nx;
        } else {
            ++nmixed;
        }
    }
    if (nz + M <= kZBatch && nz + M > 0 && M <= (1 << kTileLow)) {
        for (int m = 0; m < M; ++m) {
            io->z[nz + m] = (uint32_t)(M - 1);
            io->bin[nz + m] = m;
        }
        io->nz = nz + M;
        fu->main = true;
        fu->main_write = hl.pd.plan || nmixed > 0;
    }
    fu->h = hl.pd.plan && nx > 0 && nx <= kZBatch;
    if (fu->h) {
        // the layer's gates are the Hadamards of the mask's qubits in ascending order
        const TilePlan& hp = *hl.pd.plan;
        int pass_of_qubit[64];
        for (int q = 0, g = 0; q < 64; ++q) pass_of_qubit[q] = ((hl.mask >> q) & 1) ? hp.pass_of_gate[g++] : -1;
        fu->h_indep = !getenv("VQA_NO_INDEP_H");
        fu->term_hpass.assign(call.obs.n_terms, -1);
        for (int t = 0; t < call.obs.n_terms && fu->h_indep; ++t) {
            if (call.obs.x[t] == 0 || call.obs.z[t] != 0) continue;
            int pk = -1;
            for (int q = 0; q < 64; ++q) {
                if (!((call.obs.x[t] >> q) & 1)) continue;
                if (pk >= 0 && pass_of_qubit[q] != pk) fu->h_indep = false;
                pk = pass_of_qubit[q];
            }
            fu->term_hpass[t] = pk;
        }
    }
}

}  // namespace

static int chunk_plan(const EvalCall& call, double frac, size_t cap_bytes, int64_t& Cmax, int& nb, int64_t* fit = nullptr) {
    const int n = call.c->n_qubits;
    const int64_t D = (int64_t)1 << n;
    size_t free_b = 0, total_b = 0;
    VQA_HIP(hipMemGetInfo(&free_b, &total_b));
    // (the workspace the library holds already is reused by this call: it counts as free)
    free_b += call_workspace_held(call);
    // state budget: a share of what is free now
    size_t budget = std::min<size_t>(cap_bytes, (size_t)(frac * (double)free_b));
    // (A workspace that is nearly as large as the budget is kept as it is: growing it by a state would free and
    // allocate the whole of it again, and memory that has been in use before is cleared by the driver when it is
    // handed out -- seconds for a hundred GB.  A sixteenth of it is left for the tables beside the states.)
    const size_t held = call_workspace_held(call);
    if (held - held / 16 >= budget - budget / 8) budget = std::min(budget, held - held / 16);
    Cmax = (int64_t)(budget / ((size_t)D * sizeof(double2)));
    if (Cmax < 1) VQA_FAIL(VQA_ENOMEM, "not enough free device memory for one %d-qubit state", n);
    if (fit) *fit = Cmax;   // the states the budget holds; Cmax: the most a launch runs side by side
    Cmax = std::min<int64_t>(Cmax, 4096);
    // (tests: a small cap makes a sweep run in several chunks of parameters on registers whose states would all fit)
    if (const char* cap = getenv("VQA_HBM_MAX_STATES")) {
        Cmax = std::min<int64_t>(Cmax, std::max(atoi(cap), 1));
        if (fit) *fit = Cmax;
    }
    nb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, D / (kT * 4)));
    return VQA_OK;
}

// carves a workspace: every part 256-byte aligned
struct Carver {
    char* p = nullptr;
    size_t total = 0;
    size_t add(size_t bytes) { const size_t at = total; total += pad256(bytes); return at; }
};

int HBM_FN(hbm_engine_run)(const EvalCall& call) {
    const int n = call.c->n_qubits;
    if (n < 1 || n > 30) VQA_FAIL(VQA_EUNSUPPORTED, "HBM engine supports 1..30 qubits (got %d)", n);
    if (call.n_marg_bits > 10)
        VQA_FAIL(VQA_EUNSUPPORTED, "HBM engine: n_marg_bits <= 10 (got %d)", call.n_marg_bits);
    if (call.E <= 0) return VQA_OK;
    const int64_t D = (int64_t)1 << n;
    int64_t Cmax;
    int nb;
    int rc = chunk_plan(call, 0.4, (size_t)8 << 30, Cmax, nb);
    if (rc) return rc;
    Cmax = std::min<int64_t>(Cmax, call.E);
    PlanDev pd;
    HLayer hl;
    if (tiled(call)) {
        pd.plan = call.cone ? &cone_plan_of(call) : &plan_of(call);
        hl.mask = hlayer_mask(call);
        if (hl.mask) hl.pd.plan = &hplan_of(call, hl.mask);
    }
    Carver cv;
    const size_t o_st = cv.add((size_t)Cmax * D * sizeof(double2));
    const size_t o_tab = cv.add((size_t)Cmax * std::max(call.c->n_ops, 1) * sizeof(double2));
    const size_t o_part = cv.add((size_t)Cmax * nb * kZBatch * sizeof(double));
    const size_t o_gate = cv.add(pd.plan ? pd.image_bytes() : 0);
    const size_t o_inst = cv.add(pd.plan ? pd.inst_bytes(Cmax) : 0);
    const size_t o_hgate = cv.add(hl.pd.plan ? hl.pd.image_bytes() : 0);
    const size_t o_hinst = cv.add(hl.pd.plan ? hl.pd.inst_bytes(Cmax) : 0);
    void* ws = nullptr;
    rc = call_workspace(call, cv.total, &ws);
    if (rc) return rc;
    char* w = (char*)ws;
    double2* st = (double2*)(w + o_st);
    double2* tab = (double2*)(w + o_tab);
    double* partial = (double*)(w + o_part);
    if (pd.plan) {
        pd.buf = w + o_gate;
        pd.carve(w + o_inst, Cmax);
        rc = upload_plan(call, pd);
        if (rc) return rc;
    }
    if (hl.pd.plan) {
        hl.pd.buf = w + o_hgate;
        hl.pd.carve(w + o_hinst, Cmax);
        rc = upload_plan(call, hl.pd);
        if (rc) return rc;
    }
    for (int64_t e0 = 0; e0 < call.E; e0 += Cmax) {
        HbmCommon h;
        h.n = n;
        h.n_ops = call.c->n_ops;
        h.n_input = call.c->n_input;
        h.ops = call.d_ops;
        h.theta = call.d_theta;
        h.input = call.d_input;
        h.spec = call.spec;
        h.e_base = e0;
        h.C = (int)std::min<int64_t>(Cmax, call.E - e0);
        rc = launch_gates(call, h, st, tab, pd);
        if (rc) return rc;
        rc = measure_states(call, st, h.C, e0, nb, partial, nullptr, hl);   // the chunk's states are not needed afterwards
        if (rc) return rc;
    }
    return VQA_OK;
}

bool HBM_FN(hbm_engine_can_sweep)(const EvalCall& call) {
    if (!tiled(call) || !call.spec.psr || call.n_sweep <= 0 || call.c->n_theta == 0 || !call.cone) return false;
    return cone_plan_of(call).sweepable;
}

// Full parameter-shift sweeps of n_samples samples (rows j * (2P+1) + k of the outputs).
int HBM_FN(hbm_engine_sweep)(const EvalCall& call, int n_samples) {
    const int n = call.c->n_qubits;
    if (n > 30) VQA_FAIL(VQA_EUNSUPPORTED, "HBM engine supports 1..30 qubits (got %d)", n);
    if (call.n_marg_bits > 10)
        VQA_FAIL(VQA_EUNSUPPORTED, "HBM engine: n_marg_bits <= 10 (got %d)", call.n_marg_bits);
    PlanDev pd;
    pd.plan = &cone_plan_of(call);
    const TilePlan& plan = *pd.plan;
    HLayer hl;
    hl.mask = hlayer_mask(call);
    if (hl.mask) hl.pd.plan = &hplan_of(call, hl.mask);
    // the sweep plan's parameter lists without the parameters that cannot matter for this call's
    // observables (those join the ones that drive no gate: their rows are copies of row 0)
    ConeInfo& cone = *call.cone;
    {
        std::lock_guard<std::mutex> lk(call.c->mu);
        if (!cone.hbm_built) {
            cone.hbm_unused = plan.unused;
            for (size_t k = 0; k < plan.ord.size(); ++k) {
                if (cone.dead[plan.ord[k]]) {
                    cone.hbm_unused.push_back(plan.ord[k]);
                } else {
                    cone.hbm_ord.push_back(plan.ord[k]);
                    cone.hbm_ord_pass.push_back(plan.ord_pass[k]);
                    cone.hbm_ord_sweep.push_back(plan.ord_sweep[k]);
                    cone.hbm_coef.push_back(plan.ord_coef[k]);
                }
            }
            cone.hbm_built = true;
        }
    }
    // a caller's parameter subset: per-call copies of the lists (the call then waits for the stream
    // before it returns, so that the asynchronous copies have read them)
    std::vector<int32_t> m_ord, m_ord_pass, m_ord_sweep, m_unused;
    std::vector<double> m_coef;
    if (call.param_mask) {
        for (size_t k = 0; k < cone.hbm_ord.size(); ++k) {
            if (!call.param_mask[cone.hbm_ord[k]]) continue;
            m_ord.push_back(cone.hbm_ord[k]);
            m_ord_pass.push_back(cone.hbm_ord_pass[k]);
            m_ord_sweep.push_back(cone.hbm_ord_sweep[k]);
            m_coef.push_back(cone.hbm_coef[k]);
        }
        for (int32_t i : cone.hbm_unused)
            if (call.param_mask[i]) m_unused.push_back(i);
    }
    const std::vector<int32_t>& ord = call.param_mask ? m_ord : cone.hbm_ord;
    const std::vector<int32_t>& ord_pass = call.param_mask ? m_ord_pass : cone.hbm_ord_pass;
    const std::vector<int32_t>& ord_sweep = call.param_mask ? m_ord_sweep : cone.hbm_ord_sweep;
    const std::vector<int32_t>& unused = call.param_mask ? m_unused : cone.hbm_unused;
    const std::vector<double>& ord_coef = call.param_mask ? m_coef : cone.hbm_coef;
    const int64_t D = (int64_t)1 << n;
    const int P = call.c->n_theta, K = 2 * P + 1;
    const int m = (int)ord.size(), n_unused = (int)unused.size();
    const int n_ops = std::max(call.c->n_ops, 1);
    int64_t Cmax;
    int nb;
    // slot 0: unshifted final state, slot 1: running unshifted state, slots 2 ..: one per parameter,
    // behind them the unshifted final state with the Hadamard layer applied (when in use) and the
    // second place of the running state (a pass reads it in one place and writes it to the other,
    // so that the states that branch off in the same launch read it as it was)
    int64_t fit_states = 0;
    int rc = chunk_plan(call, 0.6, (size_t)200 << 30, Cmax, nb, &fit_states);
    if (rc) return rc;
    FuseInfo fu;
    PassIO fio;
    memset(&fio, 0, sizeof(fio));
    fuse_plan(call, hl, D >> kTileBits, &fu, &fio);
    const int n_alt = fu.h_indep ? (int)hl.pd.plan->passes.size() : 1;   // places behind the Hadamard layer
    if (Cmax < 4 + n_alt) VQA_FAIL(VQA_ENOMEM, "not enough free device memory for %d %d-qubit states", 4 + n_alt, n);
    // Late branching.  A parameter's state equals the running unshifted state up to the first sweep of its
    // pass that the parameter can change (ord_sweep).  The running state then runs the pass first and leaves
    // the tile as it is in front of every such sweep (one place per distinct sweep of the pass), and the
    // branching states start there instead of at the head of the pass.  Taken when it saves a tenth of
    // the sweeps and the places cost at most a quarter of the parameter slots.  When there is room for the
    // places of ALL passes at once, the run of the unshifted circuit that every sample starts with leaves
    // them on its way and no second unshifted state runs beside the parameters' (keep_all).
    const size_t n_pass = plan.passes.size();
    std::vector<uint64_t> snap_mask(n_pass, 0);
    std::vector<int> snap_base(n_pass, 0);   // keep_all: the first place of each pass
    int n_snap = 0;
    bool keep_all = false;
    {
        int64_t total = 0, saved = 0;
        std::vector<int64_t> tail(n_pass + 1, 0);   // sweeps from pass k to the end
        for (size_t k = n_pass; k-- > 0;) tail[k] = tail[k + 1] + (int64_t)plan.passes[k].sweeps.size();
        for (int q = 0; q < m; ++q) {
            total += tail[ord_pass[q]];
            saved += ord_sweep[q];
            snap_mask[ord_pass[q]] |= (uint64_t)1 << ord_sweep[q];
        }
        int n_sum = 0;
        for (size_t k = 0; k < n_pass; ++k) {
            n_snap = std::max(n_snap, __builtin_popcountll(snap_mask[k]));
            n_sum += __builtin_popcountll(snap_mask[k]);
        }
        auto fits = [&](int places) { return Cmax - 3 - n_alt - places >= std::min<int64_t>(m, 4 * (int64_t)places); };
        if (m == 0 || saved * 10 < total || !fits(n_snap) || getenv("VQA_NO_LATE_BRANCH")) {
            n_snap = 0;
        } else if (fits(n_sum) && !getenv("VQA_LATE_BRANCH_PER_PASS")) {
            keep_all = true;
            n_snap = n_sum;
            for (size_t k = 1; k < n_pass; ++k) snap_base[k] = snap_base[k - 1] + __builtin_popcountll(snap_mask[k - 1]);
        }
    }
    const bool late = n_snap > 0;
    const int C = (int)std::min<int64_t>(Cmax - 3 - n_alt - n_snap, std::max(m, 1));
    // Lanes.  The samples of a sweep are independent, and one sample is a chain of a few dozen launches,
    // many of them small (a single state, or the few states of an early pass).  When the states of
    // several samples fit side by side (every parameter in one chunk), a launch runs a GROUP of samples,
    // one per lane of its grid (blockIdx.z): a lane has its own states and tables at fixed distances from
    // lane 0's.  Groups in turn go round-robin to a few streams (stream 0 the call's), each with lanes of
    // its own, so that the small launches of one group run beside the large ones of another.
    const int64_t per_lane = (int64_t)C + 3 + n_alt + n_snap;
    int n_lanes = 1;   // streams
    int G = 1;         // samples per launch
    hipStream_t side_stream[3];
    hipEvent_t side_done[3], side_fork = nullptr;
    if (C >= m && n_samples > 1 && !getenv("VQA_HBM_ONE_LANE")) {
        const int64_t fit = fit_states / per_lane;
        // (measured at depth 8, 15 / 16 / 17 / 18 / 20 qubits: +40 / +19 / +10 / +5 / +1 % over one sample per launch on four streams)
        int want_g = n <= 17 ? 16 : n <= 20 ? 8 : 1, want_s = n <= 20 ? 2 : 1;
        if (const char* e = getenv("VQA_HBM_GRID_LANES")) want_g = std::max(atoi(e), 1);
        if (const char* e = getenv("VQA_HBM_STREAMS")) want_s = std::min(std::max(atoi(e), 1), 4);
        G = (int)std::max<int64_t>(1, std::min<int64_t>({ want_g, (int64_t)n_samples, fit }));
        const int64_t groups = ((int64_t)n_samples + G - 1) / G;
        const int want = (int)std::min<int64_t>({ want_s, groups, fit / G });
        if (want > 1) n_lanes = 1 + call_side_streams(call, want - 1, side_stream, side_done, &side_fork);
    }
    const int n_slots = n_lanes * G;   // lanes of states and tables in the workspace
    Carver cv;
    const size_t st_bytes = pad256((size_t)per_lane * D * sizeof(double2));
    const size_t o_st = cv.add(st_bytes * n_slots);
    const size_t o_branch = cv.add((size_t)std::max(m, 1) * sizeof(BranchRec));
    const size_t tab_bytes = pad256((size_t)(C + 2) * n_ops * sizeof(double2));
    const size_t o_tab = cv.add(tab_bytes * n_slots);
    const size_t nb_part = std::max<size_t>(nb, (fu.main || fu.h) ? (size_t)fu.n_tiles : 0);
    const size_t part_bytes = pad256((size_t)C * nb_part * kZBatch * 2 * sizeof(double));
    const size_t o_part = cv.add(part_bytes * n_slots);
    const size_t o_gate = cv.add(pd.image_bytes());
    const size_t inst_bytes = pad256(pd.inst_bytes(C + 2));
    const size_t o_inst = cv.add(inst_bytes * n_slots);
    const size_t o_ord = cv.add((size_t)std::max(m, 1) * sizeof(int32_t));
    const size_t o_coef = cv.add((size_t)std::max(m, 1) * sizeof(double));
    const size_t o_unused = cv.add((size_t)std::max(n_unused, 1) * sizeof(int32_t));
    const size_t o_hgate = cv.add(hl.pd.plan ? hl.pd.image_bytes() : 0);
    const size_t hinst_bytes = pad256(hl.pd.plan ? hl.pd.inst_bytes(C) : 0);
    const size_t o_hinst = cv.add(hinst_bytes * n_slots);
    void* ws = nullptr;
    rc = call_workspace(call, cv.total, &ws);
    if (rc) return rc;
    char* w = (char*)ws;
    int32_t* d_ord = (int32_t*)(w + o_ord);
    double* d_coef = (double*)(w + o_coef);
    int32_t* d_unused = (int32_t*)(w + o_unused);
    BranchRec* d_branch = (BranchRec*)(w + o_branch);
    std::vector<BranchRec> h_branch;   // (function-local: uploaded as kernel arguments, read before the upload returns)
    pd.buf = w + o_gate;
    pd.carve(w + o_inst, C + 2);
    rc = upload_plan(call, pd);
    if (rc) return rc;
    if (hl.pd.plan) {
        hl.pd.buf = w + o_hgate;
        hl.pd.carve(w + o_hinst, C);
        rc = upload_plan(call, hl.pd);
        if (rc) return rc;
    }
    const PlanDev pd0 = pd;
    const HLayer hl0 = hl;
    const EvalCall& call0 = call;
    if (m > 0) {
        VQA_HIP(hipMemcpyAsync(d_ord, ord.data(), (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, call.stream));
        VQA_HIP(hipMemcpyAsync(d_coef, ord_coef.data(), (size_t)m * sizeof(double), hipMemcpyHostToDevice, call.stream));
    }
    if (n_unused > 0)
        VQA_HIP(hipMemcpyAsync(d_unused, unused.data(), (size_t)n_unused * sizeof(int32_t), hipMemcpyHostToDevice, call.stream));
    if (late) {
        for (int q = 0; q < m; ++q) {
            const uint64_t below = snap_mask[ord_pass[q]] & (((uint64_t)1 << ord_sweep[q]) - 1);
            h_branch.push_back(BranchRec{ (int64_t)(snap_base[ord_pass[q]] + __builtin_popcountll(below)) * D, ord_sweep[q], 0 });
        }
        static_assert(sizeof(BranchRec) % 16 == 0, "upload_by_value writes whole 16-byte words");
        rc = upload_by_value(call.stream, d_branch, h_branch.data(), (size_t)m * sizeof(BranchRec));
        if (rc) return rc;
    }
    const unsigned gD = (unsigned)((D + kT - 1) / kT);
    // (on every way out the call's stream waits for the other lanes: it orders the workspace's next user)
    struct LaneJoin {
        int n = 0;
        hipStream_t main = nullptr;
        hipStream_t* side = nullptr;
        hipEvent_t* done = nullptr;
        void now() {
            for (int l = 0; l < n; ++l)
                if (hipEventRecord(done[l], side[l]) != hipSuccess || hipStreamWaitEvent(main, done[l], 0) != hipSuccess)
                    (void)hipStreamSynchronize(side[l]);
            n = 0;
        }
        ~LaneJoin() { now(); }
    } join;
    if (n_lanes > 1) {
        // the lanes start behind the uploads above (and behind the workspace's previous user, whom the call's stream waits for)
        VQA_HIP(hipEventRecord(side_fork, call0.stream));
        for (int l = 1; l < n_lanes; ++l) VQA_HIP(hipStreamWaitEvent(side_stream[l - 1], side_fork, 0));
        join.n = n_lanes - 1;
        join.main = call0.stream;
        join.side = side_stream;
        join.done = side_done;
    }
    HbmCommon h;
    h.n = n;
    h.n_ops = call.c->n_ops;
    h.n_input = call.c->n_input;
    h.ops = call.d_ops;
    h.theta = call.d_theta;
    h.input = call.d_input;
    h.spec = call.spec;
    h.C = 1;
    struct LaneReset {
        ~LaneReset() { t_lanes = LaneGrid(); }
    } lane_reset;
    for (int64_t j = 0, group = 0; j < n_samples; j += G, ++group) {
        // samples j .. j + lanes - 1 in one chain of launches; the pointers below are lane 0's
        const int64_t row0 = j * K;
        t_lanes.n = (int)std::min<int64_t>(G, n_samples - j);
        t_lanes.ls = LaneStride{ (int64_t)(st_bytes / sizeof(double2)), (int64_t)(tab_bytes / sizeof(double2)),
                                 (int64_t)(part_bytes / sizeof(double)), 0, (int64_t)K };
        // the group's stream, states and tables
        const int lane = (int)(group % n_lanes);
        const size_t slot = (size_t)lane * G;
        EvalCall call = call0;
        if (lane > 0) call.stream = side_stream[lane - 1];
        PlanDev pd = pd0;
        pd.carve(w + o_inst + inst_bytes * slot, C + 2);
        pd.lane_inst = (int64_t)inst_bytes;
        HLayer hl = hl0;
        if (hl.pd.plan) {
            hl.pd.carve(w + o_hinst + hinst_bytes * slot, C);
            hl.pd.lane_inst = (int64_t)hinst_bytes;
        }
        double2* st = (double2*)(w + o_st + st_bytes * slot);
        double2* tab = (double2*)(w + o_tab + tab_bytes * slot);
        double* partial = (double*)(w + o_part + part_bytes * slot);
        double2* F = st;
        double2* R = st + D;
        double2* S = st + 2 * D;
        double2* R2 = st + (int64_t)(C + 2) * D;
        hl.alt = st + (int64_t)(C + 3) * D;   // the unshifted final state behind the Hadamard layer (n_alt places)
        double2* snap = st + (int64_t)(C + 3 + n_alt) * D;   // late branching: the running state in front of the branching sweeps of a pass
        // the unshifted circuit: row 0 of the sample (NaN when the sample index is invalid)
        h.e_base = row0;
        hipLaunchKernelGGL(k_init, lane_grid(dim3(gD)), dim3(kT), 0, call.stream, F, D, D, t_lanes.ls);
        if (m > 0) {
            // the tables of the first chunk of parameters hold the unshifted circuit's in row 0
            const int cnt = std::min(C, m);
            const int64_t na = (int64_t)(cnt + 2) * h.n_ops;
            hipLaunchKernelGGL(k_sweep_angles, lane_grid(dim3((unsigned)((na + kT - 1) / kT))), dim3(kT), 0, call.stream, h, j, d_ord,
                               cnt + 2, tab, t_lanes.ls);
            launch_mats(call, pd, tab, h.n_ops, 0, cnt + 2);
        } else {
            if (h.n_ops > 0)
                hipLaunchKernelGGL(k_angles, lane_grid(dim3((unsigned)((h.n_ops + kT - 1) / kT))), dim3(kT), 0, call.stream, h, tab, t_lanes.ls);
            launch_mats(call, pd, tab, h.n_ops, 0, 1);
        }
        for (size_t k = 0; k < n_pass; ++k) {
            PassIO io;
            memset(&io, 0, sizeof(io));
            io.alt_from = 1 << 30;
            if (keep_all) {
                io.snap_mask = snap_mask[k];
                io.snap = snap + (int64_t)snap_base[k] * D;
            }
            launch_pass(call, pd, k, F, 0, 1, &io);
        }
        rc = measure_states(call, F, 1, row0, nb, partial, nullptr, hl);   // leaves F alone: the layer goes onto hl.alt
        if (rc) return rc;
        if (fu.h_indep) {
            // the unshifted final state behind each pass of the layer alone
            for (size_t k = 0; k < hl.pd.plan->passes.size(); ++k) {
                PassIO io;
                memset(&io, 0, sizeof(io));
                io.alt_from = 1 << 30;
                io.rd0 = F;
                io.wr0 = hl.alt + (int64_t)k * D;
                launch_pass(call, hl.pd, k, hl.alt, 0, 1, &io);
            }
        }
        if (n_unused > 0) {
            if (call.obs.n_obs > 0)
                hipLaunchKernelGGL(k_copy_base, lane_grid(dim3((unsigned)((n_unused * call.obs.n_obs + kT - 1) / kT))), dim3(kT), 0,
                                   call.stream, call.d_out_expect, row0, d_unused, n_unused, call.obs.n_obs, t_lanes.ls);
            if (call.n_marg_bits > 0) {
                const int M = 1 << call.n_marg_bits;
                hipLaunchKernelGGL(k_copy_base, lane_grid(dim3((unsigned)((n_unused * M + kT - 1) / kT))), dim3(kT), 0,
                                   call.stream, call.d_out_marginal, row0, d_unused, n_unused, M, t_lanes.ls);
            }
        }
        if (m == 0) continue;
        // the parameters, a chunk of slots at a time, in the order of the passes that hold their rotations
        if (!keep_all) hipLaunchKernelGGL(k_init, lane_grid(dim3(gD)), dim3(kT), 0, call.stream, R, D, D, t_lanes.ls);
        double2* Rcur = R;
        size_t rpass = 0;
        for (int a = 0; a < m; a += C) {
            const int cnt = std::min(C, m - a);
            const size_t k0 = (size_t)ord_pass[a], k1 = (size_t)ord_pass[a + cnt - 1];
            if (a > 0) {
                const int64_t na = (int64_t)(cnt + 2) * h.n_ops;
                hipLaunchKernelGGL(k_sweep_angles, lane_grid(dim3((unsigned)((na + kT - 1) / kT))), dim3(kT), 0, call.stream, h, j,
                                   d_ord + a, cnt + 2, tab, t_lanes.ls);
                launch_mats(call, pd, tab, h.n_ops, 0, cnt + 2);
            }
            for (; !keep_all && rpass < k0; ++rpass) launch_pass(call, pd, rpass, Rcur, 1, 1);
            int live = 0;
            for (size_t k = k0; late && k < n_pass; ++k) {
                // Late branching: the running state (in place, table row 1) runs pass k once, beside the chunk's
                // states that are under way when it has to move on with the chunk, and leaves the pass's
                // branching points; the states that branch here start from those in a launch behind it.
                // A later chunk that branches in the same pass finds the branching points still there.
                int fresh = 0;
                while (live + fresh < cnt && (size_t)ord_pass[a + live + fresh] == k) ++fresh;
                PassIO io;
                memset(&io, 0, sizeof(io));
                io.alt_from = 1 << 30;
                const bool run_r = !keep_all && rpass == k && (fresh > 0 || k < k1);
                if (run_r) {
                    io.snap_mask = snap_mask[k];
                    io.snap = snap;
                    rpass = k + 1;
                }
                PassIO bio;
                memset(&bio, 0, sizeof(bio));
                bio.alt_src = snap;
                if (!keep_all && k < k1) {
                    if (run_r) launch_pass(call, pd, k, R, 1, live + 1, &io);
                    else launch_pass(call, pd, k, S, 2, live, &io);
                    bio.alt_from = 0;
                    bio.branch = d_branch + a + live;
                    launch_pass(call, pd, k, S + (int64_t)live * D, 2 + live, fresh, &bio);
                    live += fresh;
                } else {
                    if (run_r) launch_pass(call, pd, k, R, 1, 1, &io);
                    bio.alt_from = live;
                    bio.branch = d_branch + a;
                    live += fresh;
                    if (k + 1 == n_pass && fu.main) {
                        memcpy(bio.z, fio.z, sizeof(bio.z));
                        memcpy(bio.bin, fio.bin, sizeof(bio.bin));
                        bio.nz = fio.nz;
                        bio.F = F;
                        bio.partial = partial;
                        bio.write_state = fu.main_write ? 1 : 0;
                    }
                    launch_pass(call, pd, k, S, 2, live, &bio);
                }
            }
            for (size_t k = k0; !late && k < n_pass; ++k) {
                // the states whose rotation sits in this pass branch off the running state: they read it
                int fresh = 0;
                while (live + fresh < cnt && (size_t)ord_pass[a + live + fresh] == k) ++fresh;
                PassIO io;
                memset(&io, 0, sizeof(io));
                io.alt_src = Rcur;
                if (k < k1) {
                    // the running state moves on with the chunk while later passes still branch off it
                    io.alt_from = 1 + live;
                    io.rd0 = Rcur == R ? nullptr : Rcur;
                    io.wr0 = Rcur == R ? R2 : nullptr;
                    live += fresh;
                    launch_pass(call, pd, k, R, 1, live + 1, &io);
                    Rcur = Rcur == R ? R2 : R;
                    rpass = k + 1;
                } else {
                    io.alt_from = live;
                    live += fresh;
                    if (k + 1 == n_pass && fu.main) {
                        memcpy(io.z, fio.z, sizeof(io.z));
                        memcpy(io.bin, fio.bin, sizeof(io.bin));
                        io.nz = fio.nz;
                        io.F = F;
                        io.partial = partial;
                        io.write_state = fu.main_write ? 1 : 0;
                    }
                    launch_pass(call, pd, k, S, 2, live, &io);
                }
            }
            PairArgs pa = { F, hl.alt, row0, d_ord + a, d_coef + a, call.spec.shift };
            rc = measure_states(call, S, cnt, 0, nb, partial, &pa, hl, &fu);
            if (rc) return rc;
        }
    }
    join.now();
    VQA_HIP(hipGetLastError());
    if (getenv("VQA_PLAN_DEBUG")) {
        fprintf(stderr, "[vqa] sweep of %d samples: %d live parameters, %zu passes, %lld state passes in all (%.1f per sample); "
                "%d parameter slots per chunk, %d places for branching points%s, %d samples per launch on %d streams\n",
                n_samples, m, n_pass, (long long)t_state_passes, (double)t_state_passes / std::max(n_samples, 1), C, n_snap,
                keep_all ? " (all passes)" : "", G, n_lanes);
        for (size_t k = 0; k < n_pass; ++k) {
            const auto& p = plan.passes[k];
            int nb_ = 0, skip = 0;
            std::vector<int> distinct;
            for (size_t q = 0; q < plan.ord.size(); ++q)
                if ((size_t)plan.ord_pass[q] == k && !cone.dead[plan.ord[q]]) {
                    ++nb_;
                    skip += plan.ord_sweep[q];
                    if (std::find(distinct.begin(), distinct.end(), plan.ord_sweep[q]) == distinct.end()) distinct.push_back(plan.ord_sweep[q]);
                }
            fprintf(stderr, "[vqa]   pass %2zu: %3zu gates, %2zu sweeps (%zu with a diagonal run), %2zu fused 2x2, %3zu register ops, %zu late gates; %d live parameters branch here, %d sweeps in front of them in all, %zu distinct first sweeps\n",
                    k, p.gates.size(), p.sweeps.size(), p.runs.size(), p.mats.size(), p.ops.size(), p.xgates.size(), nb_, skip, distinct.size());
        }
        t_state_passes = 0;
    }
    if (call.param_mask) VQA_HIP(hipStreamSynchronize(call.stream));
    return VQA_OK;
}

int HBM_FN(hbm_engine_statevector)(const EvalCall& call, double* d_out_state) {
    const int n = call.c->n_qubits;
    if (n < 1 || n > 30) VQA_FAIL(VQA_EUNSUPPORTED, "HBM engine supports 1..30 qubits (got %d)", n);
    PlanDev pd;
    if (tiled(call)) {
        pd.plan = &plan_of(call);
    }
    Carver cv;
    const size_t o_tab = cv.add((size_t)std::max(call.c->n_ops, 1) * sizeof(double2));
    const size_t o_gate = cv.add(pd.plan ? pd.image_bytes() : 0);
    const size_t o_inst = cv.add(pd.plan ? pd.inst_bytes(1) : 0);
    void* ws = nullptr;
    int rc = call_workspace(call, cv.total, &ws);
    if (rc) return rc;
    char* w = (char*)ws;
    if (pd.plan) {
        pd.buf = w + o_gate;
        pd.carve(w + o_inst, 1);
        rc = upload_plan(call, pd);
        if (rc) return rc;
    }
    HbmCommon h;
    h.n = n;
    h.n_ops = call.c->n_ops;
    h.n_input = call.c->n_input;
    h.ops = call.d_ops;
    h.theta = call.d_theta;
    h.input = call.d_input;
    h.spec = call.spec;
    h.e_base = 0;
    h.C = 1;
    return launch_gates(call, h, (double2*)d_out_state, (double2*)(w + o_tab), pd);
}

}  // namespace vqa
