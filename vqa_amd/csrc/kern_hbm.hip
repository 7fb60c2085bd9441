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
#ifndef VQA_TILES_PER_WG
#define VQA_TILES_PER_WG 1
#endif

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

__global__ __launch_bounds__(kT) void k_init(double2* st, int64_t total, int64_t D) {
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (i < total) st[i] = make_double2((i & (D - 1)) == 0 ? 1.0 : 0.0, 0.0);
}

// (cos, sin) of the half angle of every rotation gate of every instance in the chunk
__global__ __launch_bounds__(kT) void k_angles(HbmCommon h, double2* tab) {
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
constexpr int kMaxMats = 48;                  // most fused 2x2s in one pass (their matrices sit in LDS)
constexpr int kMaxOps = 96;                   // most register operations in one pass
constexpr int kMaxSweeps = 24;                // most sweeps in one pass
constexpr int kMaxPhaseGates = 64;            // most diagonal gates in the pass's phase-table runs
constexpr int kMaxDiagGates = 128;            // most diagonal gates in the pass's runs of both kinds
constexpr int kMaxShifts = 4;                 // most distinct CZ distances in a signs-only run
constexpr int kTilesPerWg = VQA_TILES_PER_WG;  // most tiles one workgroup runs one after the other
constexpr int kGateZZ = 9;                    // internal: CNOT(c,t) RZ_t(a) CNOT(c,t) = exp(i a/2 Z_c Z_t), p0 = t, p1 = c

// one gate of a pass: positions < 32 are tile positions, 32 + q is outer qubit q, -1 unused
struct TileGate {
    int32_t kind, p0, p1, g;                  // g = index into the angle table
};
// A pass is a list of SWEEPS over the tile in LDS.  In a sweep every thread holds the
// amplitudes that differ in the sweep's group positions, applies the sweep's register
// operations to them in order -- fused 2x2s on a group position (the product of the
// 1-qubit gates that follow each other there) and CNOTs whose target is a group position
// (the control may be any qubit) -- then the sweep's run of diagonal gates, and puts them
// back.  A diagonal run is either a phase per amplitude (two 64-entry tables for the low
// and the high six tile positions, plus the gates that join the halves) or, when it holds
// only Z and CZ, a sign per amplitude from a few bit masks.
enum { OP_MAT = 0, OP_CNOT = 1 };
enum { DG_NONE = 0, DG_PHASE = 1, DG_SIGN = 2 };
struct GroupOp {
    int32_t type, slot, a, b;                 // MAT: a = matrix number within the pass; CNOT: a = control position
};
struct MatDesc {
    int32_t begin, count;                     // gates [begin, begin + count) of the circuit's gate image, multiplied in order
};
struct SweepDesc {
    int32_t gpos[4];                          // group slot -> tile position
    int32_t tpos[kTileBits];                  // thread bit -> tile position
    int32_t op_begin, op_count;               // register operations of the pass
    int32_t dg_mode, dg_begin, dg_count, dg_cross;   // DG_PHASE: gates of the phase list, the first dg_cross join the halves;
                                                     // DG_SIGN: gates of the pass
    int32_t n_shift, shift[kMaxShifts];       // DG_SIGN: the distances |p0 - p1| of its CZs inside the tile
};
struct SignRun {                              // a signs-only run with this workgroup's outer bits folded in
    int32_t zmask, neg0, m[kMaxShifts];
};
struct TilePassArgs {
    int n, n_ops;
    int tq[kTileBits];                        // tile position -> qubit (ascending)
    int oq[32];                               // outer bit number -> qubit (ascending)
    int n_outer;
    int n_mats, n_pops, n_sweeps;             // this pass
    int pg_begin, pg_count, dg_total;         // its diagonal gates: [pg_begin, +dg_total), the first pg_count in phase-table runs
    int tiles_per_wg;
    int mat_off, n_mats_total;                // where its matrices start in an instance's row of the matrix table
};

struct c2 {
    double x, y;
};
__device__ __forceinline__ c2 cm(c2 a, c2 b) { return { a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x }; }
__device__ __forceinline__ c2 ca(c2 a, c2 b) { return { a.x + b.x, a.y + b.y }; }

__device__ __forceinline__ bool gate_bit(int p, int64_t base, int i) {
    return p >= 32 ? ((base >> (p - 32)) & 1) != 0 : ((i >> p) & 1) != 0;
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// LDS slot of tile index i: the low four bits are XORed with the two nibbles above, so that
// 16 indices that differ in one position of each residue class mod 4 land in 16 different
// 16-byte bank groups (the sweeps choose their thread bits accordingly)
__device__ __forceinline__ int tile_slot(int i) { return i ^ ((i >> 4) & 15) ^ ((i >> 8) & 15); }

// the product of the 1-qubit gates [begin, begin + count) for every instance: mats[(ci * M + m) * 4 ..]
__global__ __launch_bounds__(kT) void k_mats(const TileGate* __restrict__ gates, const MatDesc* __restrict__ descs, int M,
                                             const double2* __restrict__ tab, int n_ops, int C, double2* __restrict__ mats) {
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (int64_t)C * M) return;
    const int ci = (int)(i / M), k = (int)(i - (int64_t)ci * M);
    const MatDesc md = descs[k];
    c2 m00 = { 1, 0 }, m01 = { 0, 0 }, m10 = { 0, 0 }, m11 = { 1, 0 };
    for (int g = md.begin; g < md.begin + md.count; ++g) {
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
    double2* o = mats + i * 4;
    o[0] = make_double2(m00.x, m00.y);
    o[1] = make_double2(m01.x, m01.y);
    o[2] = make_double2(m10.x, m10.y);
    o[3] = make_double2(m11.x, m11.y);
}

template <int J>
__device__ __forceinline__ void apply_mat(double2 (&a)[kTA], const double2* m) {
    const double2 m00 = m[0], m01 = m[1], m10 = m[2], m11 = m[3];
#pragma unroll
    for (int r = 0; r < kTA; ++r) {
        if (r & (1 << J)) continue;
        const double2 a0 = a[r], a1 = a[r | (1 << J)];
        double2 r0, r1;
        r0.x = m00.x * a0.x - m00.y * a0.y + m01.x * a1.x - m01.y * a1.y;
        r0.y = m00.x * a0.y + m00.y * a0.x + m01.x * a1.y + m01.y * a1.x;
        r1.x = m10.x * a0.x - m10.y * a0.y + m11.x * a1.x - m11.y * a1.y;
        r1.y = m10.x * a0.y + m10.y * a0.x + m11.x * a1.y + m11.y * a1.x;
        a[r] = r0;
        a[r | (1 << J)] = r1;
        // two pairs at a time: all of them at once would double the live registers
        if (kTA > 8 && (r & ((2 << J) | 1 | (J == 0 ? 2 : 0))) != 0) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int J>
__device__ __forceinline__ void apply_cnot(double2 (&a)[kTA], int cpos, int64_t base, int ib, const int (&off)[kTA]) {
#pragma unroll
    for (int r = 0; r < kTA; ++r) {
        if (r & (1 << J)) continue;
        if (gate_bit(cpos, base, ib | off[r])) {
            const double2 t = a[r];
            a[r] = a[r | (1 << J)];
            a[r | (1 << J)] = t;
        }
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

// One workgroup runs the pass on `tiles_per_wg` tiles of one instance, one after the other:
// the next tile's amplitudes are fetched from HBM while the sweeps of the current one run.
__global__ __launch_bounds__(kTT, kTileWaves) void k_tile_pass(double2* __restrict__ st, const TileGate* __restrict__ gates,
                                                   const GroupOp* __restrict__ ops, const SweepDesc* __restrict__ sweeps,
                                                   const double2* __restrict__ tab, const double2* __restrict__ mats,
                                                   TilePassArgs a) {
    extern __shared__ double2 tile[];
    __shared__ double2 smat[kMaxMats * 4];
    __shared__ double2 dtab[2][128];
    __shared__ GroupOp s_ops[kMaxOps];
    __shared__ SweepDesc s_sw[kMaxSweeps];
    __shared__ SignRun s_sign[kMaxSweeps];
    __shared__ TileGate s_dg[kMaxDiagGates];
    __shared__ double2 s_dcs[kMaxPhaseGates];
    const int tid = threadIdx.x;
    const int ci = blockIdx.y;
    double2* s = st + ((int64_t)ci << a.n);
    tab += (int64_t)ci * a.n_ops;
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
    const uint32_t w0 = blockIdx.x * (uint32_t)a.tiles_per_wg;
    // the prefetched tile, as one vector value (an array carried around the tile loop would live in scratch)
    typedef double pvec_t __attribute__((ext_vector_type(2 * kTA)));
    pvec_t pv;
    {
        const uint32_t b0 = tile_base(a, w0);
#pragma unroll
        for (int k = 0; k < kTA; ++k) {
            const double2 t2 = s[b0 | g_tid | g_k[k]];
            pv[2 * k] = t2.x;
            pv[2 * k + 1] = t2.y;
        }
    }
    // the pass's tables, into LDS while the first amplitudes are on their way
    for (int k = tid; k < a.n_pops; k += kTT) s_ops[k] = ops[k];
    for (int k = tid; k < a.n_sweeps * (int)(sizeof(SweepDesc) / 4); k += kTT)
        reinterpret_cast<int32_t*>(s_sw)[k] = reinterpret_cast<const int32_t*>(sweeps)[k];
    for (int k = tid; k < 4 * a.n_mats; k += kTT)
        smat[k] = mats[((int64_t)ci * a.n_mats_total + a.mat_off) * 4 + k];
    for (int k = tid; k < a.dg_total; k += kTT) {
        const TileGate tg = gates[a.pg_begin + k];
        s_dg[k] = tg;
        if (k < a.pg_count) s_dcs[k] = tab[tg.g];
    }
    for (int t = 0; t < (kTilesPerWg > 1 ? a.tiles_per_wg : 1); ++t) {
        const uint32_t base = tile_base(a, w0 + t);
        lds_barrier();   // tables staged (t = 0) / the previous tile has left LDS
        if (tid < a.n_sweeps) {
            const SweepDesc& sd = s_sw[tid];
            if (sd.dg_mode == DG_SIGN) {
                // fold this tile's outer bits into the run's masks
                SignRun sr;
                sr.zmask = 0;
                sr.neg0 = 0;
#pragma unroll
                for (int j = 0; j < kMaxShifts; ++j) sr.m[j] = 0;
                for (int g = sd.dg_begin; g < sd.dg_begin + sd.dg_count; ++g) {
                    const TileGate tg = s_dg[g];
                    if (tg.kind == VQA_GATE_Z) {
                        if (tg.p0 < 32) sr.zmask ^= 1 << tg.p0;
                        else sr.neg0 ^= (int)((base >> (tg.p0 - 32)) & 1);
                    } else if (tg.p0 < 32 && tg.p1 < 32) {
                        const int lo = min(tg.p0, tg.p1), d = max(tg.p0, tg.p1) - lo;
#pragma unroll
                        for (int j = 0; j < kMaxShifts; ++j)
                            if (j < sd.n_shift && sd.shift[j] == d) sr.m[j] ^= 1 << lo;
                    } else if (tg.p0 < 32 || tg.p1 < 32) {
                        const int pt = tg.p0 < 32 ? tg.p0 : tg.p1, po = tg.p0 < 32 ? tg.p1 : tg.p0;
                        if ((base >> (po - 32)) & 1) sr.zmask ^= 1 << pt;
                    } else {
                        sr.neg0 ^= (int)((base >> (tg.p0 - 32)) & (base >> (tg.p1 - 32)) & 1);
                    }
                }
                s_sign[tid] = sr;
            }
        }
#pragma unroll
        for (int k = 0; k < kTA; ++k) tile[tile_slot(k * kTT + tid)] = make_double2(pv[2 * k], pv[2 * k + 1]);
        if (kTilesPerWg > 1 && t + 1 < a.tiles_per_wg) {
            const uint32_t b1 = tile_base(a, w0 + t + 1);
#pragma unroll
            for (int k = 0; k < kTA; ++k) {
                const double2 t2 = s[b1 | g_tid | g_k[k]];
                pv[2 * k] = t2.x;
                pv[2 * k + 1] = t2.y;
            }
        }
        for (int si = 0; si < a.n_sweeps; ++si) {
            lds_barrier();   // the previous sweep's amplitudes (or the gathered tile) are in LDS
            const SweepDesc& sw = s_sw[si];
            const int dg_mode = uni(sw.dg_mode), dg_begin = uni(sw.dg_begin), dg_count = uni(sw.dg_count),
                      dg_cross = uni(sw.dg_cross);
            double2* dt = dtab[si & 1];
            if (dg_mode == DG_PHASE) {
                if (tid < 128) {
                    // The phase of amplitude i factors into a part from its low six tile bits (and the
                    // outer bits, fixed here), a part from its high six, and the gates that join the
                    // two halves (the first dg_cross of the run).
                    const int half = tid >> 6, i = half ? (tid & 63) << 6 : (tid & 63);
                    c2 ph = { 1.0, 0.0 };
                    bool neg = false;
                    for (int g = dg_begin + dg_cross; g < dg_begin + dg_count; ++g) {
                        const TileGate tg = s_dg[g];
                        const bool hi = (tg.p0 < 32 && tg.p0 >= 6) || (tg.p1 >= 6 && tg.p1 < 32);
                        if ((int)hi != half) continue;
                        if (tg.kind == VQA_GATE_RZ) {
                            const double2 cs = s_dcs[g];
                            ph = cm(ph, { cs.x, gate_bit(tg.p0, base, i) ? -cs.y : cs.y });
                        } else if (tg.kind == kGateZZ) {
                            const double2 cs = s_dcs[g];
                            ph = cm(ph, { cs.x, gate_bit(tg.p0, base, i) != gate_bit(tg.p1, base, i) ? -cs.y : cs.y });
                        } else {
                            bool f = gate_bit(tg.p0, base, i);
                            if (tg.kind == VQA_GATE_CZ) f = f && gate_bit(tg.p1, base, i);
                            neg ^= f;
                        }
                    }
                    if (neg) ph = { -ph.x, -ph.y };
                    dt[tid] = make_double2(ph.x, ph.y);
                }
                lds_barrier();
            }
            // this thread's amplitudes: thread bits on tpos, register bits on gpos
            int ib = 0;
#pragma unroll
            for (int b = 0; b < kThreadBits; ++b) ib |= ((tid >> b) & 1) << uni(sw.tpos[b]);
            int off[kTA];
            {
                int gp[4];
#pragma unroll
                for (int j = 0; j < kGroup; ++j) gp[j] = uni(sw.gpos[j]);
#pragma unroll
                for (int r = 0; r < kTA; ++r) {
                    off[r] = 0;
#pragma unroll
                    for (int j = 0; j < kGroup; ++j) off[r] |= ((r >> j) & 1) << gp[j];
                }
            }
            const int sb = tile_slot(ib);
            double2 v[kTA];
#pragma unroll
            for (int r = 0; r < kTA; ++r) v[r] = tile[sb ^ tile_slot(off[r])];
            const int o_end = uni(sw.op_begin) + uni(sw.op_count);
            for (int o = uni(sw.op_begin); o < o_end; ++o) {
                const GroupOp op = s_ops[o];
                const int type = uni(op.type), slot = uni(op.slot), oa = uni(op.a);
                if (type == OP_MAT) {
                    const double2* m = &smat[4 * oa];
                    switch (slot) {
                    case 0: apply_mat<0>(v, m); break;
                    case 1: apply_mat<1>(v, m); break;
                    case 2: apply_mat<2>(v, m); break;
                    default: if (kGroup > 3) apply_mat<kGroup - 1>(v, m); break;
                    }
                } else {
                    switch (slot) {
                    case 0: apply_cnot<0>(v, oa, base, ib, off); break;
                    case 1: apply_cnot<1>(v, oa, base, ib, off); break;
                    case 2: apply_cnot<2>(v, oa, base, ib, off); break;
                    default: if (kGroup > 3) apply_cnot<kGroup - 1>(v, oa, base, ib, off); break;
                    }
                }
            }
            if (dg_mode == DG_PHASE) {
#pragma unroll
                for (int r = 0; r < kTA; ++r) {
                    const int i = ib | off[r];
                    const double2 lo = dt[i & 63], hi = dt[64 + (i >> 6)];
                    c2 ph = cm({ lo.x, lo.y }, { hi.x, hi.y });
                    for (int g = dg_begin; g < dg_begin + dg_cross; ++g) {
                        const TileGate tg = s_dg[g];
                        const int p0 = uni(tg.p0), p1 = uni(tg.p1);
                        if (uni(tg.kind) == VQA_GATE_CZ) {
                            if ((i >> p0) & (i >> p1) & 1) ph = { -ph.x, -ph.y };
                        } else {
                            const double2 cs = s_dcs[g];
                            ph = cm(ph, { cs.x, (((i >> p0) ^ (i >> p1)) & 1) ? -cs.y : cs.y });
                        }
                    }
                    const c2 rr = cm({ v[r].x, v[r].y }, ph);
                    v[r] = make_double2(rr.x, rr.y);
                    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four amplitudes' table reads in flight, not all
                }
            } else if (dg_mode == DG_SIGN) {
                const SignRun& sr = s_sign[si];
                const int zmask = uni(sr.zmask), neg0 = uni(sr.neg0), n_shift = uni(sw.n_shift);
                int sh[kMaxShifts], mk[kMaxShifts];
#pragma unroll
                for (int j = 0; j < kMaxShifts; ++j) {
                    sh[j] = uni(sw.shift[j]);
                    mk[j] = j < n_shift ? uni(sr.m[j]) : 0;
                }
#pragma unroll
                for (int r = 0; r < kTA; ++r) {
                    const int i = ib | off[r];
                    int x = i & zmask;
#pragma unroll
                    for (int j = 0; j < kMaxShifts; ++j) x ^= i & (i >> sh[j]) & mk[j];
                    if ((__popc(x) ^ neg0) & 1) v[r] = make_double2(-v[r].x, -v[r].y);
                }
            }
#pragma unroll
            for (int r = 0; r < kTA; ++r) tile[sb ^ tile_slot(off[r])] = v[r];
        }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < kTA; ++k) s[base | g_tid | g_k[k]] = tile[tile_slot(k * kTT + tid)];
    }
}

// Cuts the gate list into passes.  A gate can run in a pass when its target (for
// the non-diagonal kinds) is a tile qubit and none of its qubits is held back by
// an earlier gate that could not run; gates that share no qubit commute, so
// later gates on other qubits may run ahead of a held-back one.  A CNOT - RZ(target) -
// CNOT sandwich is one diagonal ZZ phase gate.  Inside a pass
// the gates are grouped into sweeps with the same argument: a gate goes into the
// earliest sweep at or behind the last one that touches its qubits -- a 1-qubit gate or
// CNOT into the sweep's register operations when its target is (or can become) one of the
// sweep's group positions and the sweep's diagonal run does not yet touch its qubits;
// a diagonal gate into the diagonal run of the last sweep that touches its qubits.
struct TilePlan {
    struct Pass {
        int tq[kTileBits];
        std::vector<TileGate> gates;   // register-operation gates, then the phase-list runs, then the signs-only runs
        std::vector<GroupOp> ops;
        std::vector<MatDesc> mats;     // gate ranges relative to this pass's gates
        std::vector<SweepDesc> sweeps;
        int pg_begin = 0, pg_count = 0, dg_total = 0;
    };
    std::vector<Pass> passes;
    // images copied to the device: the passes' tables one after the other
    std::vector<TileGate> all_gates;
    std::vector<GroupOp> all_ops;
    std::vector<MatDesc> all_mats;     // gate ranges in all_gates
    std::vector<SweepDesc> all_sweeps;
    std::vector<size_t> goff, ooff, moff, soff;   // where each pass starts in the images
    std::vector<int> pass_of_gate;     // gate index -> the pass that runs it
    // parameter-shift sweeps (filled by plan_sweep): the parameters that drive exactly one
    // rotation, ordered by the pass of that rotation, and the ones that drive none
    bool sweepable = false;
    std::vector<int32_t> ord, ord_pass, unused;
    std::vector<double> ord_coef;
};

// false when the pass needs more descriptors than the kernel holds (the caller retries with fewer gates)
static bool group_sweeps(TilePlan::Pass& pass, const std::vector<TileGate>& in_order) {
    struct HOp { int type, pos, ctrl; std::vector<TileGate> gates; };
    struct HSweep { std::vector<int> gpos; std::vector<HOp> ops; std::vector<TileGate> diag; };
    std::vector<HSweep> sweeps;
    std::map<int, std::pair<int, int>> last;   // position -> (sweep, 0 = register operations | 1 = diagonal run)
    auto last_of = [&](int p) { auto it = last.find(p); return it == last.end() ? std::make_pair(-1, 0) : it->second; };
    auto fits = [&](const HSweep& sw, int t) {
        if (std::find(sw.gpos.begin(), sw.gpos.end(), t) != sw.gpos.end()) return true;
        if ((int)sw.gpos.size() >= kGroup) return false;
        // every residue class mod 4 keeps a position outside the group (see tile_slot)
        int same = 0, members = 0;
        for (int p : sw.gpos) same += (p & 3) == (t & 3);
        for (int p = 0; p < kTileBits; ++p) members += (p & 3) == (t & 3);
        return same + 1 < members;
    };
    for (const TileGate& tg : in_order) {
        const bool diag = tg.kind == VQA_GATE_RZ || tg.kind == VQA_GATE_Z || tg.kind == VQA_GATE_CZ || tg.kind == kGateZZ;
        std::pair<int, int> dep = last_of(tg.p0);
        if (tg.p1 >= 0) dep = std::max(dep, last_of(tg.p1));
        bool as_op = !diag;
        if (diag && tg.p1 < 0 && tg.p0 < 32 && dep.first >= 0 && dep.second == 0) {
            // RZ / Z straight behind a fused 2x2 of its qubit joins that product
            const HSweep& sw = sweeps[dep.first];
            for (auto it = sw.ops.rbegin(); it != sw.ops.rend(); ++it) {
                if (it->pos == tg.p0 || it->ctrl == tg.p0) {
                    as_op = it->type == OP_MAT && it->pos == tg.p0;
                    break;
                }
            }
        }
        if (as_op) {
            int s = dep.first < 0 ? 0 : dep.second == 0 ? dep.first : dep.first + 1;
            while (s < (int)sweeps.size() && !fits(sweeps[s], tg.p0)) ++s;
            if (s >= (int)sweeps.size()) { sweeps.emplace_back(); s = (int)sweeps.size() - 1; }
            HSweep& sw = sweeps[s];
            if (std::find(sw.gpos.begin(), sw.gpos.end(), tg.p0) == sw.gpos.end()) sw.gpos.push_back(tg.p0);
            if (tg.kind == VQA_GATE_CNOT) {
                sw.ops.push_back(HOp{ OP_CNOT, tg.p0, tg.p1, { tg } });
                last[tg.p1] = std::max(last_of(tg.p1), std::make_pair(s, 0));
            } else {
                HOp* into = nullptr;
                for (auto it = sw.ops.rbegin(); it != sw.ops.rend(); ++it) {
                    if (it->pos == tg.p0 || it->ctrl == tg.p0) {
                        if (it->type == OP_MAT && it->pos == tg.p0) into = &*it;
                        break;
                    }
                }
                if (into) into->gates.push_back(tg);
                else sw.ops.push_back(HOp{ OP_MAT, tg.p0, -1, { tg } });
            }
            last[tg.p0] = std::max(last_of(tg.p0), std::make_pair(s, 0));
        } else {
            int s = std::max(dep.first, 0);
            if (sweeps.empty()) sweeps.emplace_back();
            sweeps[s].diag.push_back(tg);
            last[tg.p0] = std::max(last_of(tg.p0), std::make_pair(s, 1));
            if (tg.p1 >= 0) last[tg.p1] = std::max(last_of(tg.p1), std::make_pair(s, 1));
        }
    }
    if ((int)sweeps.size() > kMaxSweeps) return false;
    std::vector<TileGate> phase_list, sign_list;
    for (HSweep& hs : sweeps) {
        SweepDesc sd;
        memset(&sd, 0, sizeof(sd));
        // fill the group with unused positions, then hand out the thread bits: bits 0..3 get one
        // position of each residue class mod 4 (classes 0, 1, 2 first), the rest ascending
        for (int p = kTileBits - 1; p >= 0 && (int)hs.gpos.size() < kGroup; --p)
            if (fits(hs, p) && std::find(hs.gpos.begin(), hs.gpos.end(), p) == hs.gpos.end()) hs.gpos.push_back(p);
        std::vector<int> rest;
        for (int p = 0; p < kTileBits; ++p)
            if (std::find(hs.gpos.begin(), hs.gpos.end(), p) == hs.gpos.end()) rest.push_back(p);
        std::vector<int> tp;
        for (int cls = 0; cls < 4; ++cls) {
            auto it = std::find_if(rest.begin(), rest.end(), [&](int p) { return (p & 3) == cls; });
            tp.push_back(*it);   // every class keeps a position outside the group (fits())
            rest.erase(it);
        }
        tp.insert(tp.end(), rest.begin(), rest.end());
        for (int j = 0; j < kGroup; ++j) sd.gpos[j] = hs.gpos[j];
        for (int b = 0; b < kThreadBits; ++b) sd.tpos[b] = tp[b];
        sd.op_begin = (int)pass.ops.size();
        for (const HOp& op : hs.ops) {
            const int slot = (int)(std::find(hs.gpos.begin(), hs.gpos.end(), op.pos) - hs.gpos.begin());
            if (op.type == OP_MAT) {
                pass.ops.push_back(GroupOp{ OP_MAT, slot, (int)pass.mats.size(), 0 });
                pass.mats.push_back(MatDesc{ (int)pass.gates.size(), (int)op.gates.size() });
                pass.gates.insert(pass.gates.end(), op.gates.begin(), op.gates.end());
            } else {
                pass.ops.push_back(GroupOp{ OP_CNOT, slot, op.ctrl, 0 });
                pass.gates.push_back(op.gates[0]);   // kept for the gate -> pass map
            }
        }
        sd.op_count = (int)pass.ops.size() - sd.op_begin;
        sd.dg_count = (int)hs.diag.size();
        if (!hs.diag.empty()) {
            bool signs = true;
            std::vector<int> shifts;
            for (const TileGate& t : hs.diag) {
                signs &= t.kind == VQA_GATE_Z || t.kind == VQA_GATE_CZ;
                if (t.kind == VQA_GATE_CZ && t.p0 < 32 && t.p1 < 32) {
                    const int d = std::abs(t.p0 - t.p1);
                    if (std::find(shifts.begin(), shifts.end(), d) == shifts.end()) shifts.push_back(d);
                }
            }
            if (signs && (int)shifts.size() <= kMaxShifts) {
                sd.dg_mode = DG_SIGN;
                sd.n_shift = (int)shifts.size();
                for (size_t j = 0; j < shifts.size(); ++j) sd.shift[j] = shifts[j];
                sd.dg_begin = (int)sign_list.size();            // moved behind the phase list below
                sign_list.insert(sign_list.end(), hs.diag.begin(), hs.diag.end());
            } else {
                // gates between the low six and the high six tile bits go first
                auto cross = [](const TileGate& t) {
                    return (t.kind == VQA_GATE_CZ || t.kind == kGateZZ) && t.p0 < 32 && t.p1 < 32 && (t.p0 < 6) != (t.p1 < 6);
                };
                std::stable_partition(hs.diag.begin(), hs.diag.end(), cross);
                sd.dg_mode = DG_PHASE;
                sd.dg_cross = (int)std::count_if(hs.diag.begin(), hs.diag.end(), cross);
                sd.dg_begin = (int)phase_list.size();           // index into the phase list
                phase_list.insert(phase_list.end(), hs.diag.begin(), hs.diag.end());
            }
        }
        pass.sweeps.push_back(sd);
    }
    pass.pg_begin = (int)pass.gates.size();
    pass.pg_count = (int)phase_list.size();
    pass.gates.insert(pass.gates.end(), phase_list.begin(), phase_list.end());
    pass.gates.insert(pass.gates.end(), sign_list.begin(), sign_list.end());
    pass.dg_total = pass.pg_count + (int)sign_list.size();
    for (SweepDesc& sd : pass.sweeps)
        if (sd.dg_mode == DG_SIGN) sd.dg_begin += pass.pg_count;   // both kinds index the pass's diagonal list
    return (int)pass.mats.size() <= kMaxMats && (int)pass.ops.size() <= kMaxOps && pass.pg_count <= kMaxPhaseGates &&
           pass.dg_total <= kMaxDiagGates;
}

// gate_dead (may be null): gates to leave out (outside the light cone of what a call measures)
static TilePlan plan_tiles(int n, const std::vector<vqa_op>& ops, const std::vector<uint8_t>* gate_dead = nullptr) {
    TilePlan plan;
    auto is_dead = [&](size_t g) { return gate_dead && (*gate_dead)[g]; };
    // the gate list with CNOT(c,t) RZ_t CNOT(c,t) sandwiches as ZZ phase gates
    struct PG { int kind, q0, q1, g; };
    std::vector<PG> pg;
    for (size_t i = 0; i < ops.size(); ++i) {
        const vqa_op& op = ops[i];
        if (is_dead(i)) continue;
        if (op.kind == VQA_GATE_CNOT && i + 2 < ops.size() && ops[i + 1].kind == VQA_GATE_RZ && ops[i + 1].q0 == op.q0 &&
            ops[i + 2].kind == VQA_GATE_CNOT && ops[i + 2].q0 == op.q0 && ops[i + 2].q1 == op.q1 &&
            !is_dead(i + 1) && !is_dead(i + 2)) {
            pg.push_back(PG{ kGateZZ, op.q0, op.q1, (int)i + 1 });
            i += 2;
        } else {
            pg.push_back(PG{ op.kind, op.q0, op.q1, (int)i });
        }
    }
    std::vector<int> todo(pg.size());
    for (size_t i = 0; i < pg.size(); ++i) todo[i] = (int)i;
    auto diagonal = [](int kind) { return kind == VQA_GATE_RZ || kind == VQA_GATE_Z || kind == VQA_GATE_CZ || kind == kGateZZ; };
    while (!todo.empty()) {
        // tile: the low qubits plus the first eight others that pending gates need as targets
        std::vector<int> tile;
        for (int q = 0; q < kTileLow; ++q) tile.push_back(q);
        for (int gi : todo) {
            const PG& op = pg[gi];
            if (diagonal(op.kind) || op.q0 < kTileLow) continue;
            if (std::find(tile.begin(), tile.end(), op.q0) == tile.end()) tile.push_back(op.q0);
            if ((int)tile.size() == kTileBits) break;
        }
        for (int q = kTileLow; q < n && (int)tile.size() < kTileBits; ++q)
            if (std::find(tile.begin(), tile.end(), q) == tile.end()) tile.push_back(q);
        std::sort(tile.begin(), tile.end());
        int pos_of[64];
        for (int q = 0; q < 64; ++q) pos_of[q] = 32 + q;                 // outer by default
        for (int p = 0; p < kTileBits; ++p) pos_of[tile[p]] = p;
        for (int cap = kGateBatch;; cap /= 2) {
            TilePlan::Pass pass;
            for (int p = 0; p < kTileBits; ++p) pass.tq[p] = tile[p];
            uint64_t held = 0;
            std::vector<int> rest;
            std::vector<TileGate> taken;
            for (int gi : todo) {
                const PG& op = pg[gi];
                const uint64_t qs = ((uint64_t)1 << op.q0) | (op.q1 >= 0 ? (uint64_t)1 << op.q1 : 0);
                const bool ok = !(held & qs) && (diagonal(op.kind) || pos_of[op.q0] < 32) &&
                                (int)taken.size() < cap;
                if (!ok) {
                    held |= qs;
                    rest.push_back(gi);
                    continue;
                }
                taken.push_back(TileGate{ op.kind, pos_of[op.q0], op.q1 >= 0 ? pos_of[op.q1] : -1, op.g });
            }
            if (!group_sweeps(pass, taken) && cap > 1) continue;
            todo.swap(rest);
            plan.passes.push_back(std::move(pass));
            break;
        }
    }
    plan.pass_of_gate.assign(ops.size(), -1);
    for (size_t k = 0; k < plan.passes.size(); ++k) {
        const auto& p = plan.passes[k];
        plan.goff.push_back(plan.all_gates.size());
        plan.ooff.push_back(plan.all_ops.size());
        plan.moff.push_back(plan.all_mats.size());
        plan.soff.push_back(plan.all_sweeps.size());
        for (const MatDesc& md : p.mats) plan.all_mats.push_back(MatDesc{ md.begin + (int)plan.all_gates.size(), md.count });
        plan.all_gates.insert(plan.all_gates.end(), p.gates.begin(), p.gates.end());
        plan.all_ops.insert(plan.all_ops.end(), p.ops.begin(), p.ops.end());
        plan.all_sweeps.insert(plan.all_sweeps.end(), p.sweeps.begin(), p.sweeps.end());
        for (const TileGate& tg : p.gates) plan.pass_of_gate[tg.g] = (int)k;
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
        return plan.pass_of_gate[gate[a]] < plan.pass_of_gate[gate[b]];
    });
    for (int i : with) {
        plan.ord.push_back(i);
        plan.ord_pass.push_back(plan.pass_of_gate[gate[i]]);
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
__global__ __launch_bounds__(kT) void k_zpart(const double2* st, int n, ZBatch zb, double* partial) {
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
                                              double* partial) {
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
__global__ __launch_bounds__(kT) void k_out_init(double* out, int64_t e_base, int C, int stride, EvalSpec spec) {
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
                                               EvalSpec spec, int t0) {
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
                                                     double2* tab) {
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
__global__ __launch_bounds__(kT) void k_spawn(double2* dst, const double2* src, int64_t D, int count) {
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
                                               double* partial) {
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
                                               uint64_t x, double* partial) {
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
                                                  double shift, int C, int stride) {
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
                                                int spread, int t0) {
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
                                                  int stride) {
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
                for (const auto& sd : p.sweeps) nd += sd.dg_count > 0;
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
constexpr int kHGates = 32;
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

// device images of a plan: gates | ops | mats | sweeps, each sized for g gates
struct PlanImage {
    TileGate* gates;
    GroupOp* ops;
    MatDesc* mats;
    SweepDesc* sweeps;
};
static size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

size_t gate_buf_bytes(size_t g) {
    g = std::max<size_t>(g, 1);
    return pad256(g * sizeof(TileGate)) + pad256(g * sizeof(GroupOp)) + pad256(g * sizeof(MatDesc)) +
           pad256(g * sizeof(SweepDesc));
}

PlanImage plan_image(size_t g, void* gate_buf) {
    g = std::max<size_t>(g, 1);
    char* w = (char*)gate_buf;
    PlanImage im;
    im.gates = (TileGate*)w;   w += pad256(g * sizeof(TileGate));
    im.ops = (GroupOp*)w;      w += pad256(g * sizeof(GroupOp));
    im.mats = (MatDesc*)w;     w += pad256(g * sizeof(MatDesc));
    im.sweeps = (SweepDesc*)w;
    return im;
}

// A plan with its device images and the fused-2x2 table of the instances it runs on.
struct PlanDev {
    const TilePlan* plan = nullptr;
    size_t g = 0;            // gates the images are sized for
    void* buf = nullptr;     // gate_buf_bytes(g)
    double2* mats = nullptr; // [instances][all_mats][4]
    size_t m4() const { return std::max<size_t>(plan->all_mats.size(), 1) * 4; }   // double2 per instance
    size_t mats_bytes(int64_t C) const { return pad256((size_t)C * m4() * sizeof(double2)); }
};

// copies the plan's tables (a plan lives as long as its circuit, so the asynchronous
// copies may read its host images after this returns)
int upload_plan(const EvalCall& call, const PlanDev& pd) {
    const TilePlan& plan = *pd.plan;
    const PlanImage im = plan_image(pd.g, pd.buf);
    if (!plan.all_gates.empty())
        VQA_HIP(hipMemcpyAsync(im.gates, plan.all_gates.data(), plan.all_gates.size() * sizeof(TileGate),
                               hipMemcpyHostToDevice, call.stream));
    if (!plan.all_ops.empty())
        VQA_HIP(hipMemcpyAsync(im.ops, plan.all_ops.data(), plan.all_ops.size() * sizeof(GroupOp),
                               hipMemcpyHostToDevice, call.stream));
    if (!plan.all_mats.empty())
        VQA_HIP(hipMemcpyAsync(im.mats, plan.all_mats.data(), plan.all_mats.size() * sizeof(MatDesc),
                               hipMemcpyHostToDevice, call.stream));
    if (!plan.all_sweeps.empty())
        VQA_HIP(hipMemcpyAsync(im.sweeps, plan.all_sweeps.data(), plan.all_sweeps.size() * sizeof(SweepDesc),
                               hipMemcpyHostToDevice, call.stream));
    VQA_HIP(hipFuncSetAttribute((const void*)k_tile_pass, hipFuncAttributeMaxDynamicSharedMemorySize,
                                kTileAmps * (int)sizeof(double2)));
    return VQA_OK;
}

// the fused 2x2s of C instances (table rows slot0 ..) from their angle tables (tab_stride entries each)
void launch_mats(const EvalCall& call, const PlanDev& pd, const double2* tab, int tab_stride, int slot0, int C) {
    const int M = (int)pd.plan->all_mats.size();
    if (M == 0 || C <= 0) return;
    const PlanImage im = plan_image(pd.g, pd.buf);
    const int64_t tot = (int64_t)C * M;
    hipLaunchKernelGGL(k_mats, dim3((unsigned)((tot + kT - 1) / kT)), dim3(kT), 0, call.stream, im.gates, im.mats, M, tab,
                       tab_stride, C, pd.mats + (size_t)slot0 * pd.m4());
}

// pass k of the plan on C states side by side (st; their angle tables at tab, table rows slot0 .. of the fused 2x2s)
void launch_pass(const EvalCall& call, const PlanDev& pd, size_t k, double2* st, const double2* tab, int tab_stride,
                 int slot0, int C) {
    const TilePlan& plan = *pd.plan;
    const auto& p = plan.passes[k];
    if (p.gates.empty() || C <= 0) return;
    const int n = call.c->n_qubits;
    const PlanImage im = plan_image(pd.g, pd.buf);
    TilePassArgs a;
    a.n = n; a.n_ops = tab_stride;
    a.n_outer = 0;
    for (int q = 0; q < n; ++q) {
        bool in_tile = false;
        for (int t = 0; t < kTileBits; ++t) in_tile |= p.tq[t] == q;
        if (!in_tile) a.oq[a.n_outer++] = q;
    }
    for (int t = 0; t < kTileBits; ++t) a.tq[t] = p.tq[t];
    a.n_mats = (int)p.mats.size();
    a.n_pops = (int)p.ops.size();
    a.n_sweeps = (int)p.sweeps.size();
    a.pg_begin = p.pg_begin;
    a.pg_count = p.pg_count;
    a.dg_total = p.dg_total;
    a.mat_off = (int)plan.moff[k];
    a.n_mats_total = (int)plan.all_mats.size();
    const int64_t D = (int64_t)1 << n;
    // enough workgroups to fill the GPU a few times over, else several tiles per workgroup
    const int64_t n_tiles = D >> kTileBits;
    int tpw = 1;
    while (tpw < kTilesPerWg && n_tiles % (2 * tpw) == 0 && (n_tiles / (2 * tpw)) * C >= 2048) tpw *= 2;
    a.tiles_per_wg = tpw;
    hipLaunchKernelGGL(k_tile_pass, dim3((unsigned)(n_tiles / tpw), C), dim3(kTT), kTileAmps * sizeof(double2),
                       call.stream, st, im.gates + plan.goff[k], im.ops + plan.ooff[k], im.sweeps + plan.soff[k], tab,
                       pd.mats + (size_t)slot0 * pd.m4(), a);
}

void launch_all_passes(const EvalCall& call, const PlanDev& pd, double2* st, const double2* tab, int tab_stride,
                       int slot0, int C) {
    for (size_t k = 0; k < pd.plan->passes.size(); ++k) launch_pass(call, pd, k, st, tab, tab_stride, slot0, C);
}

// |0..0>, the angle table and every gate for the h.C instances of a chunk; pd.plan == nullptr: one sweep per gate
int launch_gates(const EvalCall& call, const HbmCommon& h, double2* st, double2* tab, const PlanDev& pd) {
    const int n = h.n, C = h.C;
    const int64_t D = (int64_t)1 << n;
    const int64_t tot = D * C;
    hipLaunchKernelGGL(k_init, dim3((unsigned)((tot + kT - 1) / kT)), dim3(kT), 0, call.stream, st,
                       tot, D);
    const int64_t na = (int64_t)C * h.n_ops;
    if (na > 0)
        hipLaunchKernelGGL(k_angles, dim3((unsigned)((na + kT - 1) / kT)), dim3(kT), 0, call.stream,
                           h, tab);
    if (pd.plan) {
        launch_mats(call, pd, tab, h.n_ops, 0, C);
        launch_all_passes(call, pd, st, tab, h.n_ops, 0, C);
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

int measure_states(const EvalCall& call, double2* st, int C, int64_t e0, int nb, double* partial,
                   const PairArgs* pair, const HLayer& hl) {
    const int n = call.c->n_qubits;
    const int n_obs = call.obs.n_obs;
    if (n_obs > 0) {
        const unsigned g = (unsigned)((C * n_obs + kT - 1) / kT);
        if (pair)
            hipLaunchKernelGGL(k_out_init2, dim3(g), dim3(kT), 0, call.stream, call.d_out_expect, pair->row0, pair->ord,
                               pair->ord_coef, pair->shift, C, n_obs);
        else
            hipLaunchKernelGGL(k_out_init, dim3(g), dim3(kT), 0, call.stream, call.d_out_expect, e0, C, n_obs, call.spec);
    }
    // Z-strings of all observables, kZBatch at a time; `f` = the unshifted state the pairs are taken with.
    // The bins of the low-bit marginal ride in the same pass over the states when they fit behind
    // the last Z-strings (entries nzs .. nz - 1 of the batch).
    const int M = call.n_marg_bits > 0 ? 1 << call.n_marg_bits : 0;
    bool marg_done = M == 0;
    auto zflush = [&](ZBatch& zb, const double2* states, const double2* f, bool last) {
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
            if (zb.nz <= 8) hipLaunchKernelGGL(k_zpart2<8>, dim3(C, nb), dim3(kT), 0, call.stream, f, states, n, zb, partial);
            else hipLaunchKernelGGL(k_zpart2<kZBatch>, dim3(C, nb), dim3(kT), 0, call.stream, f, states, n, zb, partial);
            if (nzs)
                hipLaunchKernelGGL(k_finish2, dim3(C), dim3(kT), 0, call.stream, partial, nb, nzs, zb, 1, 0.0,
                                   call.d_out_expect, pair->row0, pair->ord, pair->ord_coef, pair->shift, n_obs, 0, 0, 0);
            if (nbins)
                hipLaunchKernelGGL(k_finish2, dim3(C), dim3(kT), 0, call.stream, partial, nb, nbins, zb, 1, 0.0,
                                   call.d_out_marginal, pair->row0, pair->ord, pair->ord_coef, pair->shift, M, 0, 1, nzs);
        } else {
            if (zb.nz <= 8) hipLaunchKernelGGL(k_zpart<8>, dim3(C, nb), dim3(kT), 0, call.stream, states, n, zb, partial);
            else hipLaunchKernelGGL(k_zpart<kZBatch>, dim3(C, nb), dim3(kT), 0, call.stream, states, n, zb, partial);
            if (nzs)
                hipLaunchKernelGGL(k_finish, dim3(C), dim3(kT), 0, call.stream, partial, nb, nzs, zb, 1, 0.0,
                                   call.d_out_expect, e0, n_obs, 0, 0, call.spec, 0);
            if (nbins)
                hipLaunchKernelGGL(k_finish, dim3(C), dim3(kT), 0, call.stream, partial, nb, nbins, zb, 0, 0.0,
                                   call.d_out_marginal, e0, M, 0, 1, call.spec, nzs);
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
            if (++zb.nz == kZBatch) zflush(zb, st, pair ? pair->f : nullptr, false);
        }
    }
    zflush(zb, st, pair ? pair->f : nullptr, true);
    // X/Y strings one at a time, except the pure-X ones when the Hadamard layer is in use
    for (int o = 0; o < n_obs; ++o) {
        for (int t = call.obs.off[o]; t < call.obs.off[o + 1]; ++t) {
            if (call.obs.x[t] == 0 || (hl.pd.plan && call.obs.z[t] == 0)) continue;
            if (pair) {
                hipLaunchKernelGGL(k_xpart2, dim3(nb, C), dim3(kT), 0, call.stream, pair->f, st, n,
                                   call.obs.z[t], call.obs.x[t], partial);
                hipLaunchKernelGGL(k_finish2, dim3(C), dim3(kT), 0, call.stream, partial, nb, 1, zb, 0,
                                   call.obs.coef[t], call.d_out_expect, pair->row0, pair->ord,
                                   pair->ord_coef, pair->shift, n_obs, o, 0, 0);
            } else {
                hipLaunchKernelGGL(k_xpart, dim3(nb, C), dim3(kT), 0, call.stream, st, n,
                                   call.obs.z[t], call.obs.x[t], partial);
                hipLaunchKernelGGL(k_finish, dim3(C), dim3(kT), 0, call.stream, partial, nb,
                                   1, zb, 0, call.obs.coef[t], call.d_out_expect, e0, n_obs, o, 0, call.spec, 0);
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
                if (mb.nz <= 8) hipLaunchKernelGGL(k_zpart2<8>, dim3(C, nb), dim3(kT), 0, call.stream, pair->f, st, n, mb, partial);
                else hipLaunchKernelGGL(k_zpart2<kZBatch>, dim3(C, nb), dim3(kT), 0, call.stream, pair->f, st, n, mb, partial);
                hipLaunchKernelGGL(k_finish2, dim3(C), dim3(kT), 0, call.stream, partial, nb, mb.nz, mb, 1, 0.0,
                                   call.d_out_marginal, pair->row0, pair->ord, pair->ord_coef, pair->shift, M, m0, 1, 0);
            } else {
                if (mb.nz <= 8) hipLaunchKernelGGL(k_zpart<8>, dim3(C, nb), dim3(kT), 0, call.stream, st, n, mb, partial);
                else hipLaunchKernelGGL(k_zpart<kZBatch>, dim3(C, nb), dim3(kT), 0, call.stream, st, n, mb, partial);
                hipLaunchKernelGGL(k_finish, dim3(C), dim3(kT), 0, call.stream, partial, nb, mb.nz, mb, 0, 0.0,
                                   call.d_out_marginal, e0, M, m0, 1, call.spec, 0);
            }
        }
    }
    if (hl.pd.plan) {
        // H X H = Z: behind the Hadamard layer the pure-X strings are Z-strings with z = x
        double2* hs = st;
        if (!pair && hl.alt) {
            const int64_t D = (int64_t)1 << n;
            hipLaunchKernelGGL(k_spawn, dim3((unsigned)((D + kT - 1) / kT)), dim3(kT), 0, call.stream, hl.alt, st, D, 1);
            hs = hl.alt;
        }
        launch_mats(call, hl.pd, nullptr, 0, 0, C);
        launch_all_passes(call, hl.pd, hs, nullptr, 0, 0, C);
        zb.nz = 0;
        for (int o = 0; o < n_obs; ++o) {
            for (int t = call.obs.off[o]; t < call.obs.off[o + 1]; ++t) {
                if (call.obs.x[t] == 0 || call.obs.z[t] != 0) continue;
                zb.z[zb.nz] = call.obs.x[t];
                zb.coef[zb.nz] = call.obs.coef[t];
                zb.col[zb.nz] = o;
                zb.bin[zb.nz] = -1;
                if (++zb.nz == kZBatch) zflush(zb, hs, pair ? pair->fh : nullptr, false);
            }
        }
        zflush(zb, hs, pair ? pair->fh : nullptr, false);
    }
    VQA_HIP(hipGetLastError());
    return VQA_OK;
}

}  // namespace

static int chunk_plan(const EvalCall& call, double frac, size_t cap_bytes, int64_t& Cmax, int& nb) {
    const int n = call.c->n_qubits;
    const int64_t D = (int64_t)1 << n;
    size_t free_b = 0, total_b = 0;
    VQA_HIP(hipMemGetInfo(&free_b, &total_b));
    // state budget: a share of what is free now
    size_t budget = std::min<size_t>(cap_bytes, (size_t)(frac * (double)free_b));
    Cmax = (int64_t)(budget / ((size_t)D * sizeof(double2)));
    if (Cmax < 1) VQA_FAIL(VQA_ENOMEM, "not enough free device memory for one %d-qubit state", n);
    Cmax = std::min<int64_t>(Cmax, 4096);
    nb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, D / (kT * 4)));
    return VQA_OK;
}

// carves a workspace: every part 256-byte aligned
struct Carver {
    char* p = nullptr;
    size_t total = 0;
    size_t add(size_t bytes) { const size_t at = total; total += pad256(bytes); return at; }
};

int hbm_engine_run(const EvalCall& call) {
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
        pd.g = (size_t)call.c->n_ops;
        hl.mask = hlayer_mask(call);
        if (hl.mask) {
            hl.pd.plan = &hplan_of(call, hl.mask);
            hl.pd.g = kHGates;
        }
    }
    Carver cv;
    const size_t o_st = cv.add((size_t)Cmax * D * sizeof(double2));
    const size_t o_tab = cv.add((size_t)Cmax * std::max(call.c->n_ops, 1) * sizeof(double2));
    const size_t o_part = cv.add((size_t)Cmax * nb * kZBatch * sizeof(double));
    const size_t o_gate = cv.add(pd.plan ? gate_buf_bytes(pd.g) : 0);
    const size_t o_mats = cv.add(pd.plan ? pd.mats_bytes(Cmax) : 0);
    const size_t o_hgate = cv.add(hl.pd.plan ? gate_buf_bytes(hl.pd.g) : 0);
    const size_t o_hmats = cv.add(hl.pd.plan ? hl.pd.mats_bytes(Cmax) : 0);
    void* ws = nullptr;
    rc = call_workspace(call, cv.total, &ws);
    if (rc) return rc;
    char* w = (char*)ws;
    double2* st = (double2*)(w + o_st);
    double2* tab = (double2*)(w + o_tab);
    double* partial = (double*)(w + o_part);
    if (pd.plan) {
        pd.buf = w + o_gate;
        pd.mats = (double2*)(w + o_mats);
        rc = upload_plan(call, pd);
        if (rc) return rc;
    }
    if (hl.pd.plan) {
        hl.pd.buf = w + o_hgate;
        hl.pd.mats = (double2*)(w + o_hmats);
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

bool hbm_engine_can_sweep(const EvalCall& call) {
    if (!tiled(call) || !call.spec.psr || call.n_sweep <= 0 || call.c->n_theta == 0 || !call.cone) return false;
    return cone_plan_of(call).sweepable;
}

// Full parameter-shift sweeps of n_samples samples (rows j * (2P+1) + k of the outputs).
int hbm_engine_sweep(const EvalCall& call, int n_samples) {
    const int n = call.c->n_qubits;
    if (n > 30) VQA_FAIL(VQA_EUNSUPPORTED, "HBM engine supports 1..30 qubits (got %d)", n);
    if (call.n_marg_bits > 10)
        VQA_FAIL(VQA_EUNSUPPORTED, "HBM engine: n_marg_bits <= 10 (got %d)", call.n_marg_bits);
    PlanDev pd;
    pd.plan = &cone_plan_of(call);
    pd.g = (size_t)call.c->n_ops;
    const TilePlan& plan = *pd.plan;
    HLayer hl;
    hl.mask = hlayer_mask(call);
    if (hl.mask) {
        hl.pd.plan = &hplan_of(call, hl.mask);
        hl.pd.g = kHGates;
    }
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
                    cone.hbm_coef.push_back(plan.ord_coef[k]);
                }
            }
            cone.hbm_built = true;
        }
    }
    // a caller's parameter subset: per-call copies of the lists (the call then waits for the stream
    // before it returns, so that the asynchronous copies have read them)
    std::vector<int32_t> m_ord, m_ord_pass, m_unused;
    std::vector<double> m_coef;
    if (call.param_mask) {
        for (size_t k = 0; k < cone.hbm_ord.size(); ++k) {
            if (!call.param_mask[cone.hbm_ord[k]]) continue;
            m_ord.push_back(cone.hbm_ord[k]);
            m_ord_pass.push_back(cone.hbm_ord_pass[k]);
            m_coef.push_back(cone.hbm_coef[k]);
        }
        for (int32_t i : cone.hbm_unused)
            if (call.param_mask[i]) m_unused.push_back(i);
    }
    const std::vector<int32_t>& ord = call.param_mask ? m_ord : cone.hbm_ord;
    const std::vector<int32_t>& ord_pass = call.param_mask ? m_ord_pass : cone.hbm_ord_pass;
    const std::vector<int32_t>& unused = call.param_mask ? m_unused : cone.hbm_unused;
    const std::vector<double>& ord_coef = call.param_mask ? m_coef : cone.hbm_coef;
    const int64_t D = (int64_t)1 << n;
    const int P = call.c->n_theta, K = 2 * P + 1;
    const int m = (int)ord.size(), n_unused = (int)unused.size();
    const int n_ops = std::max(call.c->n_ops, 1);
    int64_t Cmax;
    int nb;
    // slot 0: unshifted final state, slot 1: running unshifted state, slots 2 ..: one per parameter,
    // behind them the unshifted final state with the Hadamard layer applied (when in use)
    int rc = chunk_plan(call, 0.6, (size_t)200 << 30, Cmax, nb);
    if (rc) return rc;
    if (Cmax < 4) VQA_FAIL(VQA_ENOMEM, "not enough free device memory for four %d-qubit states", n);
    const int C = (int)std::min<int64_t>(Cmax - 3, std::max(m, 1));
    Carver cv;
    const size_t o_st = cv.add((size_t)(C + 3) * D * sizeof(double2));
    const size_t o_tab = cv.add((size_t)(C + 2) * n_ops * sizeof(double2));
    const size_t o_part = cv.add((size_t)C * nb * kZBatch * 2 * sizeof(double));
    const size_t o_gate = cv.add(gate_buf_bytes(pd.g));
    const size_t o_mats = cv.add(pd.mats_bytes(C + 2));
    const size_t o_ord = cv.add((size_t)std::max(m, 1) * sizeof(int32_t));
    const size_t o_coef = cv.add((size_t)std::max(m, 1) * sizeof(double));
    const size_t o_unused = cv.add((size_t)std::max(n_unused, 1) * sizeof(int32_t));
    const size_t o_hgate = cv.add(hl.pd.plan ? gate_buf_bytes(hl.pd.g) : 0);
    const size_t o_hmats = cv.add(hl.pd.plan ? hl.pd.mats_bytes(C) : 0);
    void* ws = nullptr;
    rc = call_workspace(call, cv.total, &ws);
    if (rc) return rc;
    char* w = (char*)ws;
    double2* st = (double2*)(w + o_st);
    double2* tab = (double2*)(w + o_tab);
    double* partial = (double*)(w + o_part);
    int32_t* d_ord = (int32_t*)(w + o_ord);
    double* d_coef = (double*)(w + o_coef);
    int32_t* d_unused = (int32_t*)(w + o_unused);
    pd.buf = w + o_gate;
    pd.mats = (double2*)(w + o_mats);
    rc = upload_plan(call, pd);
    if (rc) return rc;
    if (hl.pd.plan) {
        hl.pd.buf = w + o_hgate;
        hl.pd.mats = (double2*)(w + o_hmats);
        rc = upload_plan(call, hl.pd);
        if (rc) return rc;
    }
    if (m > 0) {
        VQA_HIP(hipMemcpyAsync(d_ord, ord.data(), (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, call.stream));
        VQA_HIP(hipMemcpyAsync(d_coef, ord_coef.data(), (size_t)m * sizeof(double), hipMemcpyHostToDevice, call.stream));
    }
    if (n_unused > 0)
        VQA_HIP(hipMemcpyAsync(d_unused, unused.data(), (size_t)n_unused * sizeof(int32_t), hipMemcpyHostToDevice, call.stream));
    const size_t n_pass = plan.passes.size();
    double2* F = st;
    double2* R = st + D;
    double2* S = st + 2 * D;
    hl.alt = st + (int64_t)(C + 2) * D;   // the unshifted final state behind the Hadamard layer
    const unsigned gD = (unsigned)((D + kT - 1) / kT);
    HbmCommon h;
    h.n = n;
    h.n_ops = call.c->n_ops;
    h.n_input = call.c->n_input;
    h.ops = call.d_ops;
    h.theta = call.d_theta;
    h.input = call.d_input;
    h.spec = call.spec;
    h.C = 1;
    for (int64_t j = 0; j < n_samples; ++j) {
        const int64_t row0 = j * K;
        // the unshifted circuit: row 0 of the sample (NaN when the sample index is invalid)
        h.e_base = row0;
        hipLaunchKernelGGL(k_init, dim3(gD), dim3(kT), 0, call.stream, F, D, D);
        if (h.n_ops > 0)
            hipLaunchKernelGGL(k_angles, dim3((unsigned)((h.n_ops + kT - 1) / kT)), dim3(kT), 0, call.stream, h, tab);
        launch_mats(call, pd, tab, h.n_ops, 0, 1);
        launch_all_passes(call, pd, F, tab, h.n_ops, 0, 1);
        rc = measure_states(call, F, 1, row0, nb, partial, nullptr, hl);   // leaves F alone: the layer goes onto hl.alt
        if (rc) return rc;
        if (n_unused > 0) {
            if (call.obs.n_obs > 0)
                hipLaunchKernelGGL(k_copy_base, dim3((unsigned)((n_unused * call.obs.n_obs + kT - 1) / kT)), dim3(kT), 0,
                                   call.stream, call.d_out_expect, row0, d_unused, n_unused, call.obs.n_obs);
            if (call.n_marg_bits > 0) {
                const int M = 1 << call.n_marg_bits;
                hipLaunchKernelGGL(k_copy_base, dim3((unsigned)((n_unused * M + kT - 1) / kT)), dim3(kT), 0,
                                   call.stream, call.d_out_marginal, row0, d_unused, n_unused, M);
            }
        }
        if (m == 0) continue;
        // the parameters, a chunk of slots at a time, in the order of the passes that hold their rotations
        hipLaunchKernelGGL(k_init, dim3(gD), dim3(kT), 0, call.stream, R, D, D);
        size_t rpass = 0;
        for (int a = 0; a < m; a += C) {
            const int cnt = std::min(C, m - a);
            const size_t k0 = (size_t)ord_pass[a], k1 = (size_t)ord_pass[a + cnt - 1];
            const int64_t na = (int64_t)(cnt + 2) * h.n_ops;
            hipLaunchKernelGGL(k_sweep_angles, dim3((unsigned)((na + kT - 1) / kT)), dim3(kT), 0, call.stream, h, j,
                               d_ord + a, cnt + 2, tab);
            launch_mats(call, pd, tab, h.n_ops, 0, cnt + 2);
            for (; rpass < k0; ++rpass) launch_pass(call, pd, rpass, R, tab + n_ops, h.n_ops, 1, 1);
            int live = 0;
            for (size_t k = k0; k < n_pass; ++k) {
                int fresh = 0;
                while (live + fresh < cnt && (size_t)ord_pass[a + live + fresh] == k) ++fresh;
                if (fresh > 0)
                    hipLaunchKernelGGL(k_spawn, dim3(gD), dim3(kT), 0, call.stream, S + (int64_t)live * D, R, D, fresh);
                live += fresh;
                if (k < k1) {
                    // the running state moves on with the chunk while later passes still take copies of it
                    launch_pass(call, pd, k, R, tab + n_ops, h.n_ops, 1, live + 1);
                    rpass = k + 1;
                } else {
                    launch_pass(call, pd, k, S, tab + 2 * (size_t)n_ops, h.n_ops, 2, live);
                }
            }
            PairArgs pa = { F, hl.alt, row0, d_ord + a, d_coef + a, call.spec.shift };
            rc = measure_states(call, S, cnt, 0, nb, partial, &pa, hl);
            if (rc) return rc;
        }
    }
    VQA_HIP(hipGetLastError());
    if (call.param_mask) VQA_HIP(hipStreamSynchronize(call.stream));
    return VQA_OK;
}

int hbm_engine_statevector(const EvalCall& call, double* d_out_state) {
    const int n = call.c->n_qubits;
    if (n < 1 || n > 30) VQA_FAIL(VQA_EUNSUPPORTED, "HBM engine supports 1..30 qubits (got %d)", n);
    PlanDev pd;
    if (tiled(call)) {
        pd.plan = &plan_of(call);
        pd.g = (size_t)call.c->n_ops;
    }
    Carver cv;
    const size_t o_tab = cv.add((size_t)std::max(call.c->n_ops, 1) * sizeof(double2));
    const size_t o_gate = cv.add(pd.plan ? gate_buf_bytes(pd.g) : 0);
    const size_t o_mats = cv.add(pd.plan ? pd.mats_bytes(1) : 0);
    void* ws = nullptr;
    int rc = call_workspace(call, cv.total, &ws);
    if (rc) return rc;
    char* w = (char*)ws;
    if (pd.plan) {
        pd.buf = w + o_gate;
        pd.mats = (double2*)(w + o_mats);
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
