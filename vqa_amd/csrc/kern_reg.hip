// Engine 2: register-blocked fused statevector kernels (4 <= n <= 14).
//
// The state of one circuit instance lives in the VGPRs of 2^(n-4) threads, 16
// complex128 amplitudes per thread.  The program compiled by reg_compile.cpp is
// interpreted per instance:
//   - preparation: every fused 2x2 is multiplied out from its angle sources,
//     split into RZ RY RZ, and its phases are merged into the diagonal stages'
//     phase tables (all in LDS);
//   - real rotations on register bits are three in-place fused multiply-adds
//     per amplitude pair; on lane bits the partner amplitude comes through DPP /
//     swizzle moves and the update is one fused multiply-add (the common
//     1/cos or 1/sin factor is left out and divided away at measurement);
//     qubits on wave bits are brought into registers by a conflict-free
//     transposition through LDS;
//   - diagonal stages are one complex multiply per amplitude by a 16-entry
//     table built per thread, with CZ / Z signs folded in from per-thread masks;
//   - the Z-basis observables and the low-bit marginal are reduced in place.
//
// Two kernels share that code.  k_reg_eval runs an explicit instance list.
// k_reg_psr runs full parameter-shift sweeps: one workgroup takes a sample,
// runs the unshifted circuit once while saving the state in front of every
// diagonal stage, and then evaluates each +-shift by patching the few table
// entries that depend on the shifted parameter and resuming from the last
// checkpoint the shift cannot have changed.
//
// Replaces, per instance, LearningCircuit.run(x) + Observable expectation
// (vqa/QNN/myQNNs_base.py:692-704, 907-914), and, for sweeps, the loop nest of
// the parameter-shift oracles (vqa/QNN/myQNNs_base.py:1370-1394), for the
// ansaetze of vqa/QNN/QML_utils.py:33-164.
#include "reg_program.h"
#include "vqa_core.h"

namespace vqa {

// Ablation switches for tools/section_profile.py: only in the separately built
// libvqahip_prof.so (-DVQA_PROF).  VQA_ABLATE (environment, bit mask) switches
// parts of the sweep kernel off so that their cost shows as a time difference;
// results are then wrong on purpose.  bits 0..3: transpositions, register-bit
// rotations, lane-bit rotations, diagonal stages; 4: measurement; 5: recomputing
// the shifted 2x2; 6: checkpoint loads; 7: patching / restoring the tables; 8: the
// phase loop of shifted evaluations; 9: run the two shifts one by one instead of as a pair;
// 10: pair measurement without loading the unshifted state; 11: without its reductions.
// waves per SIMD the sweep kernel is register-budgeted for
// Timing experiments on the sweep kernel (make libvqahip_var.so VAR_FLAGS=-DVQA_EXP=<bits>; results are
// wrong on purpose): 1 no checkpoint loads, 2 no loads of the unshifted state in the pair measurement, 4 no
// phase loop in the runs, 8 no measurement in the runs, 16 no late correction, 32 no record prefetch;
// inside the phase loop: 64 no barriers in the transpositions, 128 no transpositions, 256 no register-bit
// rotations, 512 no DPP lane rotations, 1024 no matrix-unit lane rotations, 2048 no diagonal stages.
#ifndef VQA_EXP
#define VQA_EXP 0
#endif
#ifndef VQA_WAVES
#define VQA_WAVES 3
#endif
// the same for the sweep kernel's one- and two-wave workgroups (10 and 11 qubits)
// ... and for the sweep kernel that runs two runs side by side
#ifndef VQA_PSR2_WAVES
#define VQA_PSR2_WAVES 2
#endif
#ifndef VQA_SMALL_WAVES
#define VQA_SMALL_WAVES 3
#endif
// the same for the grouped sweeps of small circuits (one-wave workgroups).  Measured at 6 / 8 / 9
// qubits: 2 (no spills) 4.03M / 1.81M / 614k gradients/s, 3 (a few spills) 3.82M / 1.48M / 571k,
// 4 and 5 slower still.
#ifndef VQA_GROUP_WAVES
#define VQA_GROUP_WAVES 2
#endif

#ifdef VQA_PROF
#define VQA_ABL(a, bit) (((a).ablate >> (bit)) & 1)
// cycle stamps of wave 0 inside the pair measurement (read with vqa_debug_stamps)
// (summed per workgroup in LDS, added to the global totals once per sample)
__device__ unsigned long long g_stamp[16];
__shared__ unsigned long long s_stamp[16];
#define VQA_STAMP0 unsigned long long st_t = clock64();
#define VQA_STAMP(i)                                                       \
    {                                                                      \
        const unsigned long long st_n = clock64();                         \
        if (threadIdx.x == 0) s_stamp[i] += st_n - st_t;                   \
        st_t = st_n;                                                       \
    }
#else
#define VQA_ABL(a, bit) 0
#define VQA_STAMP0
#define VQA_STAMP(i)
#endif

namespace {

struct RegArgs {
    const char* prog;  // device blob (RegHeader + arrays)
    const double* theta;
    const double* input;
    int n_input;
    EvalSpec spec;
    int64_t E;         // instances (explicit list) or samples (sweep)
    ObsPack obs;
    int n_marg_bits;
    double* out_expect;
    double* out_marginal;
    double2* ckpt;     // sweep: per-workgroup checkpoint area
    unsigned* next;    // sweep: work counter; workgroups take samples from it (tickets count from next_base)
    unsigned next_base;   // ... what the counter holds when the launch starts (unsigned arithmetic, the counter may wrap)
    const int32_t* dead;  // sweep kernels: parameters outside the light cone of the call's observables; the workgroup that runs
    int n_dead;           //   slice 0 of a sample writes their rows (copies of the unshifted row)
    RunRec* recs;      // sweep: per-workgroup run records
    const int32_t* live;  // sweep: the runs to execute, ascending (the others' outputs are filled in by the caller)
    int n_live;
    int need_psi;      // grouped sweeps: the list holds pair runs (the unshifted state of every sample is kept)
    int split;         // sweep kernel: workgroups that share a sample's run list (>= 1)
    int use_ftab;      // sweep kernel: the diagonal stages' thread-factor tables are kept in LDS
#ifdef VQA_PROF
    int ablate;
#endif
};

struct cplx {
    double x, y;
};
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return { a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x }; }
__device__ __forceinline__ cplx cconj(cplx a) { return { a.x, -a.y }; }
// flip the sign of v when bit (0/1) is set, without a select
__device__ __forceinline__ double flip_if(double v, unsigned bit) {
    return __hiloint2double(__double2hiint(v) ^ (int)(bit << 31), __double2loint(v));
}
__device__ __forceinline__ cplx csel_conj(cplx a, unsigned bit) { return { a.x, flip_if(a.y, bit) }; }

// square root of a unit-modulus complex number (either root)
__device__ __forceinline__ cplx sqrt_phase(cplx z) {
    cplx w = z.x >= 0.0 ? cplx{ 1.0 + z.x, z.y } : cplx{ z.y, 1.0 - z.x };  // 1+z  or  i(1-z)
    const double inv = rsqrt(w.x * w.x + w.y * w.y);
    return { w.x * inv, w.y * inv };
}

// program arrays are read through the constant address space: a uniform index
// then becomes a scalar load (the program never changes during a launch)
#define VQA_CAS __attribute__((address_space(4)))
template <class Tp>
__device__ __forceinline__ const VQA_CAS Tp* to_const(const void* p) {
    return (const VQA_CAS Tp*)(uintptr_t)p;
}

struct Prog {
    const VQA_CAS RegHeader* H;
    const VQA_CAS ElemGate* elems;
    const VQA_CAS UEntry* uent;
    const VQA_CAS D1Entry* d1;
    const VQA_CAS D2Entry* d2;
    const VQA_CAS ElemGate* terms;
    const VQA_CAS PhaseOp* phases;
    const VQA_CAS DDesc* ddesc;
    const VQA_CAS XDesc* xdesc;
    const VQA_CAS PairRef* pairs;
    const VQA_CAS SignTab* signs;
    const VQA_CAS ParamInfo* params;
    const VQA_CAS PatchRef* refs;
    const VQA_CAS RunInfo* runs;
    const VQA_CAS int32_t* dep;
};

__device__ __forceinline__ Prog load_prog(const char* blob) {
    Prog p;
    p.H = to_const<RegHeader>(blob);
    p.elems = to_const<ElemGate>(blob + p.H->off_elems);
    p.uent = to_const<UEntry>(blob + p.H->off_uent);
    p.d1 = to_const<D1Entry>(blob + p.H->off_d1);
    p.d2 = to_const<D2Entry>(blob + p.H->off_d2);
    p.terms = to_const<ElemGate>(blob + p.H->off_terms);
    p.phases = to_const<PhaseOp>(blob + p.H->off_phases);
    p.ddesc = to_const<DDesc>(blob + p.H->off_ddesc);
    p.xdesc = to_const<XDesc>(blob + p.H->off_xdesc);
    p.pairs = to_const<PairRef>(blob + p.H->off_pairs);
    p.signs = to_const<SignTab>(blob + p.H->off_signs);
    p.params = to_const<ParamInfo>(blob + p.H->off_params);
    p.refs = to_const<PatchRef>(blob + p.H->off_refs);
    p.runs = to_const<RunInfo>(blob + p.H->off_runs);
    p.dep = to_const<int32_t>(blob + p.H->off_dep);
    return p;
}

__device__ __forceinline__ double elem_angle_c(const VQA_CAS ElemGate* g, const double* theta, const double* in_row,
                                               int sparam, double sval) {
    double v = 0.0;
    const int src = g->src, idx = g->idx;
    if (src == VQA_SRC_THETA) {
        v = theta[idx];
        if (idx == sparam) v += sval;
    } else if (src == VQA_SRC_INPUT) {
        v = in_row[idx];
    }
    return g->coef * v + g->offset;
}

// Fused 2x2 of U-entry u -> e^{i d} RZ(al) RY(be) RZ(ga) with c, s = (cos, sin)(be/2)
// >= 0, A = e^{i al/2}, G = e^{i ga/2}.  The table entry m the rotation code reads:
//   register bit:            (tan(be/4), s)   three shears, exact
//   lane bit, !SCALED, and lane bits 4 / 5 (on_lane == 2):  (c, s)   x <- c x +- s partner
//   lane bit, SCALED, c >= s: (s/c, 0)        x <- x +- (s/c) partner        (state / c)
//   lane bit, SCALED, c <  s: (c/s, 1)        x <- +-(c/s) x + partner       (state / s, and a
//       sign flip of the |1> half, which is a Z after the rotation: RZ(al) Z = i RZ(al - pi),
//       so A is multiplied by -i)
// pparam >= 0: the generator (X, Y or Z) of the rotation that parameter pparam drives is
// inserted in front of that rotation (derivative direction of the parameter).
// gp = the unit factor of the exact 2x2 that the engine leaves out: e^{i d}, times i
// in the second scaled lane form, times G when there is no stage in front to take it.
// the 2x2 of one gate of a fused entry
__device__ __forceinline__ void elem_matrix(const VQA_CAS ElemGate* g, const double* theta, const double* row, int sparam,
                                            double sval, cplx& g00, cplx& g01, cplx& g10, cplx& g11) {
    const int kind = g->kind;
    const double r2 = 0.70710678118654752440;
    double c = 1.0, sn = 0.0;
    if (kind >= VQA_GATE_RX) sincos(0.5 * elem_angle_c(g, theta, row, sparam, sval), &sn, &c);
    switch (kind) {
    case VQA_GATE_H: g00 = { r2, 0 }; g01 = { r2, 0 }; g10 = { r2, 0 }; g11 = { -r2, 0 }; break;
    case VQA_GATE_X: g00 = { 0, 0 }; g01 = { 1, 0 }; g10 = { 1, 0 }; g11 = { 0, 0 }; break;
    case VQA_GATE_Y: g00 = { 0, 0 }; g01 = { 0, -1 }; g10 = { 0, 1 }; g11 = { 0, 0 }; break;
    case VQA_GATE_Z: g00 = { 1, 0 }; g01 = { 0, 0 }; g10 = { 0, 0 }; g11 = { -1, 0 }; break;
    case VQA_GATE_RX: g00 = { c, 0 }; g01 = { 0, sn }; g10 = { 0, sn }; g11 = { c, 0 }; break;
    case VQA_GATE_RY: g00 = { c, 0 }; g01 = { sn, 0 }; g10 = { -sn, 0 }; g11 = { c, 0 }; break;
    default: /* RZ */ g00 = { c, sn }; g01 = { 0, 0 }; g10 = { 0, 0 }; g11 = { c, -sn }; break;
    }
}

// C = M' M^-1 for the rotation that parameter i drives inside fused entry u, M' being M with the rotation's
// generator P in front of it: with M = L R K (L the gates behind the rotation R), M' = L R P K and P
// commuting with R, C = L P L^-1 -- only the gates behind the rotation enter
__device__ __forceinline__ void late_generator(const Prog& pp, int u, const double* theta, const double* row, int i,
                                               cplx& c00, cplx& c01, cplx& c10, cplx& c11) {
    const int ubegin = pp.uent[u].begin, ucount = pp.uent[u].count;
    bool found = false;
    c00 = { 1, 0 }; c01 = { 0, 0 }; c10 = { 0, 0 }; c11 = { 1, 0 };
    for (int j = 0; j < ucount; ++j) {
        const VQA_CAS ElemGate* g = pp.elems + ubegin + j;
        const int kind = g->kind;
        if (!found) {
            if (kind >= VQA_GATE_RX && g->src == VQA_SRC_THETA && g->idx == i) {
                found = true;
                if (kind == VQA_GATE_RX) { c00 = { 0, 0 }; c01 = { 1, 0 }; c10 = { 1, 0 }; c11 = { 0, 0 }; }
                else if (kind == VQA_GATE_RY) { c00 = { 0, 0 }; c01 = { 0, -1 }; c10 = { 0, 1 }; c11 = { 0, 0 }; }
                else c11 = { -1, 0 };
            }
            continue;
        }
        cplx g00, g01, g10, g11;
        elem_matrix(g, theta, row, -1, 0.0, g00, g01, g10, g11);
        auto add = [](cplx x, cplx y) { return cplx{ x.x + y.x, x.y + y.y }; };
        // C <- g C g^-1 (g unitary)
        const cplx t00 = add(cmul(g00, c00), cmul(g01, c10)), t01 = add(cmul(g00, c01), cmul(g01, c11));
        const cplx t10 = add(cmul(g10, c00), cmul(g11, c10)), t11 = add(cmul(g10, c01), cmul(g11, c11));
        c00 = add(cmul(t00, cconj(g00)), cmul(t01, cconj(g01)));
        c01 = add(cmul(t00, cconj(g10)), cmul(t01, cconj(g11)));
        c10 = add(cmul(t10, cconj(g00)), cmul(t11, cconj(g01)));
        c11 = add(cmul(t10, cconj(g10)), cmul(t11, cconj(g11)));
    }
}

// the 2x2 of U-entry u multiplied out (see compute_uentry for the arguments)
__device__ __forceinline__ void uentry_product(const Prog& pp, int u, const double* theta, const double* row, int sparam,
                                               double sval, int pparam, cplx& m00, cplx& m01, cplx& m10, cplx& m11) {
    const int ubegin = pp.uent[u].begin, ucount = pp.uent[u].count;
    m00 = { 1, 0 }; m01 = { 0, 0 }; m10 = { 0, 0 }; m11 = { 1, 0 };
    for (int j = 0; j < ucount; ++j) {
        const VQA_CAS ElemGate* g = pp.elems + ubegin + j;
        const int kind = g->kind;
        cplx g00, g01, g10, g11;
        elem_matrix(g, theta, row, sparam, sval, g00, g01, g10, g11);
        if (kind >= VQA_GATE_RX && g->src == VQA_SRC_THETA && g->idx == pparam) {
            const cplx h00 = g00, h01 = g01, h10 = g10, h11 = g11;
            if (kind == VQA_GATE_RX) {          // g X
                g00 = h01; g01 = h00; g10 = h11; g11 = h10;
            } else if (kind == VQA_GATE_RY) {   // g Y,  Y = [[0, -i], [i, 0]]
                g00 = { -h01.y, h01.x }; g01 = { h00.y, -h00.x };
                g10 = { -h11.y, h11.x }; g11 = { h10.y, -h10.x };
            } else {                            // g Z
                g01 = { -h01.x, -h01.y }; g11 = { -h11.x, -h11.y };
            }
        }
        // M <- G * M
        const cplx a0 = cmul(g00, m00), a1 = cmul(g01, m10), b0 = cmul(g00, m01), b1 = cmul(g01, m11);
        const cplx c0 = cmul(g10, m00), c1 = cmul(g11, m10), d0 = cmul(g10, m01), d1 = cmul(g11, m11);
        m00 = { a0.x + a1.x, a0.y + a1.y };
        m01 = { b0.x + b1.x, b0.y + b1.y };
        m10 = { c0.x + c1.x, c0.y + c1.y };
        m11 = { d0.x + d1.x, d0.y + d1.y };
    }
}

// TWO: a rotation on a register bit whose cos is at least kTwoShearMin and that has a diagonal stage behind it
// is applied as TWO shears, R = diag(c, 1/c) L(-s c) U(s / c): the entry is (s / c, -s c) -- the sign bit of
// the second number tells the forms apart -- and the scales go into the stage's phase slot with A, which
// then has modulus c: a slot P stands for the factor P on |0> and conj(P) / |P|^2 on |1>.
constexpr double kTwoShearMin = 0.25;
template <bool SCALED, bool TWO = false>
__device__ __forceinline__ void compute_uentry(const Prog& pp, int u, const double* theta, const double* row,
                                               int sparam, double sval, int pparam, double2& m, cplx& A, cplx& G,
                                               cplx& gp) {
    cplx m00, m01, m10, m11;
    uentry_product(pp, u, theta, row, sparam, sval, pparam, m00, m01, m10, m11);
    const double a00 = sqrt(m00.x * m00.x + m00.y * m00.y), a01 = sqrt(m01.x * m01.x + m01.y * m01.y);
    const double inv = 1.0 / sqrt(a00 * a00 + a01 * a01);
    const cplx dA = cmul(m00, m11), dB = cmul(m01, m10);
    cplx det = { dA.x - dB.x, dA.y - dB.y };
    const double idet = rsqrt(det.x * det.x + det.y * det.y);
    det = { det.x * idet, det.y * idet };
    const cplx eid = cconj(sqrt_phase(det));
    const double tiny = 1e-150;
    const bool h0 = a00 > tiny, h1 = a01 > tiny;
    const cplx Xp = h0 ? cmul(cplx{ m00.x / a00, m00.y / a00 }, eid) : cplx{ 1, 0 };
    const cplx Yp = h1 ? cmul(cplx{ m01.x / a01, m01.y / a01 }, eid) : cplx{ 1, 0 };
    if (h0 && h1) {
        A = sqrt_phase(cmul(Xp, Yp));
        G = cmul(Xp, cconj(A));
    } else if (!h1) {
        A = Xp; G = { 1, 0 };
    } else {
        A = Yp; G = { 1, 0 };
    }
    const double c = a00 * inv, sn = a01 * inv;
    gp = cconj(eid);
    if (pp.uent[u].d_before < 0) gp = cmul(gp, G);
    if (!pp.uent[u].on_lane) {
        if (TWO && c >= kTwoShearMin && pp.uent[u].d_after >= 0) {
            m = make_double2(sn / c, -(sn * c));
            A = { A.x * c, A.y * c };
        } else {
            m = make_double2(sn / (1.0 + c), sn);
        }
    } else if (!SCALED || pp.uent[u].on_lane == 2) {   // (2: rotated by the matrix unit, exact form)
        m = make_double2(c, sn);
    } else if (c >= sn) {
        m = make_double2(sn / c, 0.0);
    } else {
        m = make_double2(c / sn, 1.0);
        A = { A.y, -A.x };
        gp = { -gp.y, gp.x };
    }
}

// Per-instance preparation of the LDS tables of the S states of a workgroup:
// mat[u] = rotation data of every fused 2x2 (see compute_uentry), ph1 / ph2 = phase tables of the
// diagonal stages with the RZ factors of the neighbouring 2x2s merged in.
// Instance of state s2 is e0 + s2 (explicit list) or the unshifted circuit of
// sample e0 + s2 (sweep).
// ctab (explicit lists; null otherwise): rotation data, A and G of every input-free fused 2x2 at
// the unshifted parameters, worked out once per launch.  The entries are then prepared in three
// homogeneous loops (a wave that mixed cached and computed entries would run the full
// computation for its few uncached lanes anyway): the input-dependent entries are computed,
// the input-free ones copied, and an instance whose shift touches an input-free entry
// recomputes that one entry and corrects the phase slots around it.
template <bool SWEEP, bool SCALED, bool TWO = false>
__device__ __forceinline__ void prep_tables(const RegArgs& a, const Prog& pp, double2* tabs, int64_t e0, int S,
                                            int NU, int N1, int N2, int tid, int nthr, const double2* ctab = nullptr) {
    const int tab_stride = NU + N1 + N2;
    auto decode = [&](int64_t e2, int& b, int& sparam, double& sval) {
        if (SWEEP) {
            b = a.spec.sel ? a.spec.sel[e2] : (int)e2;
            if (a.spec.B != 0 && (unsigned)b >= (unsigned)a.spec.B) b = 0;
            sparam = -1;
            sval = 0.0;
        } else {
            decode_eval(a.spec, e2, b, sparam, sval);
        }
    };
    // phase 0: phase tables from the explicit RZ / ZZ terms
    for (int i = tid; i < S * (N1 + N2); i += nthr) {
        const int s2 = i / (N1 + N2), k = i - s2 * (N1 + N2);
        // (a state slot behind the end of the list runs the last instance once more: its amplitudes must
        // be finite, because the matrix-unit lane rotations multiply them by zero into their wave-mates)
        const int64_t e2 = min(e0 + s2, a.E - 1);
        cplx ph = { 1.0, 0.0 };
        {
            const int begin = k < N1 ? pp.d1[k].begin : pp.d2[k - N1].begin;
            const int count = k < N1 ? pp.d1[k].count : pp.d2[k - N1].count;
            if (count > 0) {
                int b, sparam;
                double sval;
                decode(e2, b, sparam, sval);
                const double* row = a.input + (size_t)b * a.n_input;
                double ang = 0.0;
                for (int j = 0; j < count; ++j) ang += elem_angle_c(pp.terms + begin + j, a.theta, row, sparam, sval);
                double sn, c;
                sincos(0.5 * ang, &sn, &c);
                ph = { c, sn };
            }
        }
        tabs[(size_t)s2 * tab_stride + NU + k] = make_double2(ph.x, ph.y);
    }
    // phases 1, 2: even U-stages first, so that no two entries of one phase
    // update the same phase-table slot
    const int ND = ctab ? pp.H->n_dep : NU;   // entries to compute per state
    for (int parity = 0; parity < 2; ++parity) {
        __syncthreads();
        if (ctab) {
            // the input-free entries: copies (same parity rule, other entries than the loop below)
            for (int i = tid; i < S * NU; i += nthr) {
                const int s2 = i / NU, u = i - s2 * NU;
                if (pp.uent[u].parity != parity || !pp.uent[u].input_free) continue;
                const int d_before = pp.uent[u].d_before, d_after = pp.uent[u].d_after;
                double2* tb = tabs + (size_t)s2 * tab_stride;
                tb[u] = ctab[3 * u];
                if (d_before >= 0) {
                    const double2 p = tb[NU + d_before], g = ctab[3 * u + 2];
                    const cplx q = cmul({ p.x, p.y }, { g.x, g.y });
                    tb[NU + d_before] = make_double2(q.x, q.y);
                }
                if (d_after >= 0) {
                    const double2 p = tb[NU + d_after], g = ctab[3 * u + 1];
                    const cplx q = cmul({ p.x, p.y }, { g.x, g.y });
                    tb[NU + d_after] = make_double2(q.x, q.y);
                }
            }
        }
        for (int i = tid; i < S * ND; i += nthr) {
            const int s2 = i / ND, u = ctab ? pp.dep[i - s2 * ND] : i - s2 * ND;
            const int64_t e2 = min(e0 + s2, a.E - 1);
            if (pp.uent[u].parity != parity) continue;
            const int d_before = pp.uent[u].d_before, d_after = pp.uent[u].d_after;
            int b, sparam;
            double sval;
            decode(e2, b, sparam, sval);
            const double* row = a.input + (size_t)b * a.n_input;
            double2 cs;
            cplx A, G, gp;
            compute_uentry<SCALED, TWO>(pp, u, a.theta, row, sparam, sval, -1, cs, A, G, gp);
            double2* tb = tabs + (size_t)s2 * tab_stride;
            tb[u] = cs;
            if (d_before >= 0) {
                const double2 p = tb[NU + d_before];
                const cplx q = cmul({ p.x, p.y }, G);
                tb[NU + d_before] = make_double2(q.x, q.y);
            }
            if (d_after >= 0) {
                const double2 p = tb[NU + d_after];
                const cplx q = cmul({ p.x, p.y }, A);
                tb[NU + d_after] = make_double2(q.x, q.y);
            }
        }
    }
    if (ctab) {
        // shifted instances: the input-free entries their parameter enters (one thread per state)
        __syncthreads();
        for (int s2 = tid; s2 < S; s2 += nthr) {
            const int64_t e2 = e0 + s2;
            if (e2 >= a.E) continue;
            int b, sparam;
            double sval;
            decode(e2, b, sparam, sval);
            if (sparam < 0) continue;
            const double* row = a.input + (size_t)b * a.n_input;
            double2* tb = tabs + (size_t)s2 * tab_stride;
            const int rb = pp.params[sparam].ref_begin, rcnt = pp.params[sparam].ref_count;
            for (int q = 0; q < rcnt; ++q) {
                const int u = pp.refs[rb + q].index;
                if (pp.refs[rb + q].kind != 0 || !pp.uent[u].input_free) continue;
                double2 cs;
                cplx A, G, gp;
                compute_uentry<SCALED>(pp, u, a.theta, row, sparam, sval, -1, cs, A, G, gp);
                tb[u] = cs;
                const int d_before = pp.uent[u].d_before, d_after = pp.uent[u].d_after;
                if (d_before >= 0) {
                    const double2 g0 = ctab[3 * u + 2], p = tb[NU + d_before];
                    const cplx f = cmul(cconj({ g0.x, g0.y }), G), z = cmul({ p.x, p.y }, f);
                    tb[NU + d_before] = make_double2(z.x, z.y);
                }
                if (d_after >= 0) {
                    const double2 a0 = ctab[3 * u + 1], p = tb[NU + d_after];
                    const cplx f = cmul(cconj({ a0.x, a0.y }), A), z = cmul({ p.x, p.y }, f);
                    tb[NU + d_after] = make_double2(z.x, z.y);
                }
            }
        }
    }
}

// ---- cross-lane partner fetch: value of lane (lane ^ (1 << LB)) -----------------
// On gfx950 a DPP move costs the vector unit as much as an FP64 operation (about four
// cycles per wave on a SIMD; tools/micro/valu_rates.hip).  The LDS crossbar (ds_swizzle:
// no memory access, no vector-unit time) can do the same exchange for lane bits 0..4;
// VQA_SWZ_OF_16 of every 16 amplitudes go that way.  Measured on the 12-qubit sweep:
// 0 -> 32.0k, 4 -> 31.6k, 8 -> 31.3k, 16 -> 28.0k gradients/s, so the default is 0.
#ifndef VQA_SWZ_OF_16
#define VQA_SWZ_OF_16 0
#endif
template <int LB, bool SWZ>
__device__ __forceinline__ int xor_fetch32(int v) {
    if constexpr (LB == 5) return __builtin_amdgcn_ds_bpermute(((int)(threadIdx.x & 63) ^ 32) << 2, v);
    else if constexpr (SWZ || LB == 4) return __builtin_amdgcn_ds_swizzle(v, ((1 << LB) << 10) | 0x1F);   // lane ^ (1 << LB)
    else if constexpr (LB == 0) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (LB == 1) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (LB == 2) {
        // (the first move writes every lane, so that it needs no defined old value: one instruction less)
        int t = __builtin_amdgcn_mov_dpp(v, 0x104, 0xF, 0xF, true);         // row_shl:4: right for banks 0,2
        return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);    // row_shr:4 into banks 1,3
    } else return __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, true);  // row_ror:8
}
template <int LB, bool SWZ>
__device__ __forceinline__ double xor_fetch(double v) {
    const int lo = xor_fetch32<LB, SWZ>(__double2loint(v));
    const int hi = xor_fetch32<LB, SWZ>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// The three amplitude updates are written as inline assembly with tied
// operands so that they are in place by construction: behind the uniform
// branches of the phase loop anything else makes the register allocator keep
// a second copy of the state and move it back after every block.
//   (a0, a1) <- (c a0 + s a1, c a1 - s a0) as three shears, u = tan(phi/2), s = sin(phi),
// for four pairs at once: the three steps of a pair depend on each other, so four
// pairs are interleaved inside the block (blocks cannot be interleaved by the compiler)
__device__ __forceinline__ void rot_quad(double& a0, double& a1, double& b0, double& b1, double& c0, double& c1,
                                         double& d0, double& d1, double u, double s) {
    asm("v_fma_f64 %0, %8, %1, %0\n\tv_fma_f64 %2, %8, %3, %2\n\tv_fma_f64 %4, %8, %5, %4\n\tv_fma_f64 %6, %8, %7, %6\n\t"
        "v_fma_f64 %1, -%9, %0, %1\n\tv_fma_f64 %3, -%9, %2, %3\n\tv_fma_f64 %5, -%9, %4, %5\n\tv_fma_f64 %7, -%9, %6, %7\n\t"
        "v_fma_f64 %0, %8, %1, %0\n\tv_fma_f64 %2, %8, %3, %2\n\tv_fma_f64 %4, %8, %5, %4\n\tv_fma_f64 %6, %8, %7, %6"
        : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1), "+v"(d0), "+v"(d1)
        : "v"(u), "v"(s));
}
//   (a0, a1) <- (a0 + t a1, a1 + y (a0 + t a1)): the two-shear form (see compute_uentry), four pairs at once
__device__ __forceinline__ void rot2_quad(double& a0, double& a1, double& b0, double& b1, double& c0, double& c1,
                                          double& d0, double& d1, double t, double y) {
    asm("v_fma_f64 %0, %8, %1, %0\n\tv_fma_f64 %2, %8, %3, %2\n\tv_fma_f64 %4, %8, %5, %4\n\tv_fma_f64 %6, %8, %7, %6\n\t"
        "v_fma_f64 %1, %9, %0, %1\n\tv_fma_f64 %3, %9, %2, %3\n\tv_fma_f64 %5, %9, %4, %5\n\tv_fma_f64 %7, %9, %6, %7"
        : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1), "+v"(d0), "+v"(d1)
        : "v"(t), "v"(y));
}
//   x <- c x + se p
__device__ __forceinline__ void lane_upd(double& x, double p, double c, double se) {
    double t;
    asm("v_mul_f64 %1, %4, %2\n\tv_fma_f64 %0, %3, %0, %1" : "+v"(x), "=&v"(t) : "v"(p), "v"(c), "v"(se));
}
//   x <- x + te p
__device__ __forceinline__ void lane_upd_a(double& x, double p, double te) {
    asm("v_fma_f64 %0, %2, %1, %0" : "+v"(x) : "v"(p), "v"(te));
}
//   x <- ke x + p
__device__ __forceinline__ void lane_upd_b(double& x, double p, double ke) {
    asm("v_fma_f64 %0, %2, %0, %1" : "+v"(x) : "v"(p), "v"(ke));
}
//   (ax + i ay) <- (ax + i ay) (tx + i ty)
__device__ __forceinline__ void cmul_inplace(double& ax, double& ay, double tx, double ty) {
    double t1, t2;
    asm("v_mul_f64 %2, %1, %5\n\tv_mul_f64 %3, %1, %4\n\tv_fma_f64 %1, %0, %5, %3\n\tv_fma_f64 %0, %0, %4, -%2"
        : "+v"(ax), "+v"(ay), "=&v"(t1), "=&v"(t2)
        : "v"(tx), "v"(ty));
}

template <int J>
__device__ __forceinline__ void apply_ureg(double2 (&amp)[kRegAmps], double u, double s) {
    // the 8 amplitude pairs of register bit J, two pairs (re and im of each) per block
    constexpr int lo = (1 << J) - 1;
#pragma unroll
    for (int k = 0; k < kRegAmps / 2; k += 2) {
        const int ra = ((k & ~lo) << 1) | (k & lo), rb = (((k + 1) & ~lo) << 1) | ((k + 1) & lo);
        const int ra1 = ra | (1 << J), rb1 = rb | (1 << J);
        rot_quad(amp[ra].x, amp[ra1].x, amp[ra].y, amp[ra1].y, amp[rb].x, amp[rb1].x, amp[rb].y, amp[rb1].y, u, s);
    }
}

template <int J>
__device__ __forceinline__ void apply_ureg2(double2 (&amp)[kRegAmps], double t, double y) {
    constexpr int lo = (1 << J) - 1;
#pragma unroll
    for (int k = 0; k < kRegAmps / 2; k += 2) {
        const int ra = ((k & ~lo) << 1) | (k & lo), rb = (((k + 1) & ~lo) << 1) | ((k + 1) & lo);
        const int ra1 = ra | (1 << J), rb1 = rb | (1 << J);
        rot2_quad(amp[ra].x, amp[ra1].x, amp[ra].y, amp[ra1].y, amp[rb].x, amp[rb1].x, amp[rb].y, amp[rb1].y, t, y);
    }
}

// FORM: 0 = x <- m0 x +- m1 partner; 1 = x <- x +- m0 partner; 2 = x <- +-m0 x + partner
template <int LB, int FORM>
__device__ __forceinline__ void apply_ulane(double2 (&amp)[kRegAmps], double m0, double m1, int lane) {
    const unsigned bit = (unsigned)(lane >> LB) & 1u;
    const double ce = flip_if(FORM == 0 ? m1 : m0, bit);
    // the cross-lane moves below read registers that the previous block wrote
    // from inline assembly, which the compiler's hazard padding does not see
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
#define VQA_LANE_UPD(R, PR, PI)                                \
    {                                                          \
        if constexpr (FORM == 0) {                             \
            lane_upd(amp[R].x, PR, m0, ce);                    \
            lane_upd(amp[R].y, PI, m0, ce);                    \
        } else if constexpr (FORM == 1) {                      \
            lane_upd_a(amp[R].x, PR, ce);                      \
            lane_upd_a(amp[R].y, PI, ce);                      \
        } else {                                               \
            lane_upd_b(amp[R].x, PR, ce);                      \
            lane_upd_b(amp[R].y, PI, ce);                      \
        }                                                      \
    }
    // crossbar fetches first: they travel while the DPP part runs
    constexpr int NS = (LB < 5) ? VQA_SWZ_OF_16 : 0, ND = kRegAmps - NS;
    double sr[NS > 0 ? NS : 1], si[NS > 0 ? NS : 1];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        sr[k] = xor_fetch<LB, true>(amp[ND + k].x);
        si[k] = xor_fetch<LB, true>(amp[ND + k].y);
    }
#pragma unroll
    for (int r = 0; r < ND; ++r) {
        const double pr = xor_fetch<LB, false>(amp[r].x);
        const double pi = xor_fetch<LB, false>(amp[r].y);
        VQA_LANE_UPD(r, pr, pi)
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) VQA_LANE_UPD(ND + k, sr[k], si[k])
#undef VQA_LANE_UPD
}

// Rotations on lane bits 4 and 5 through the matrix unit.  v_mfma_f64_4x4x4_4b_f64 computes, for
// every lane l = 16 i + 4 b + j,  D[l] = sum_k A[16 k + 4 b + i] * B[16 k + 4 b + j]  (layout found
// by experiment, tools/micro/mfma_f64.hip): the contraction runs over lane bits 4 and 5 of B, and the
// result comes back in the same lanes.  With B = one register of the state and A = the 4x4 real
// matrix R5 (x) R4 of the two rotations, ONE instruction per register rotates both qubits, at the
// price of two register-bit rotations (16 cycles; two lane rotations through DPP cost 27, through
// the LDS crossbar more).  The FP64 matrix unit shares its arithmetic with the vector unit on
// gfx950 (measured: no overlap between an MFMA stream and an FMA stream on one SIMD), so this is
// fewer vector-unit cycles, not a second pipe.
// (c4, s4), (c5, s5): cos, sin of half the rotation angles; an absent rotation is (1, 0).
__device__ __forceinline__ void apply_ulane45(double2 (&amp)[kRegAmps], double c4, double s4, double c5, double s5, int lane) {
    const unsigned i0 = lane & 1u, i1 = (lane >> 1) & 1u, k0 = (lane >> 4) & 1u, k1 = (lane >> 5) & 1u;
    // R[out][in] = [[c, s], [-s, c]]
    const double r4 = (i0 ^ k0) ? flip_if(s4, i0) : c4;
    const double r5 = (i1 ^ k1) ? flip_if(s5, i1) : c5;
    const double A = r4 * r5;
    // the registers below were last written from inline assembly, which the compiler's hazard
    // padding does not see (same on the way out)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 4");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < kRegAmps; ++r) {
        amp[r].x = __builtin_amdgcn_mfma_f64_4x4x4f64(A, amp[r].x, 0.0, 0, 0, 0);
        amp[r].y = __builtin_amdgcn_mfma_f64_4x4x4f64(A, amp[r].y, 0.0, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15");
    __builtin_amdgcn_sched_barrier(0);
}

// State images in global memory are [amplitude r][thread] and are moved with buffer
// instructions: one resource for the workgroup's area, a uniform byte offset per
// image and amplitude, one 32-bit per-thread offset.  All sixteen accesses of a state
// are then in flight together (with per-amplitude 64-bit vector addresses the
// register budget makes the compiler reload an address between any two of them).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
// cache policy of the state-image traffic (aux operand of the buffer builtins).  The
// non-temporal hint (2) was measured and is slower: the images are re-read many times.
#ifndef VQA_IMAGE_AUX
#define VQA_IMAGE_AUX 0
#endif
__device__ __forceinline__ void load_state(double2 (&amp)[kRegAmps], __amdgpu_buffer_rsrc_t area, unsigned image_off,
                                           unsigned nthr, unsigned tid) {
    const unsigned toff = tid * (unsigned)sizeof(double2);
#pragma unroll
    for (int r = 0; r < kRegAmps; ++r) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(area, toff, image_off + r * nthr * (unsigned)sizeof(double2), VQA_IMAGE_AUX);
        amp[r] = *(const double2*)&v;
    }
}
__device__ __forceinline__ void store_state(const double2 (&amp)[kRegAmps], __amdgpu_buffer_rsrc_t area,
                                            unsigned image_off, unsigned nthr, unsigned tid) {
    const unsigned toff = tid * (unsigned)sizeof(double2);
#pragma unroll
    for (int r = 0; r < kRegAmps; ++r) {
        const double2 d = amp[r];
        __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4*)&d, area, toff,
                                               image_off + r * nthr * (unsigned)sizeof(double2), VQA_IMAGE_AUX);
    }
}

// everything the op interpreter needs besides the amplitudes
struct Ctx {
    const double2* mat;   // LDS tables of this state
    const double2* ph1;
    const double2* ph2;
    char* xbuf;           // LDS transposition buffer
    const unsigned short* smask;  // LDS: [sign table][thread] 16-bit sign masks
    const unsigned char* smask8 = nullptr;  // ... the same as bytes (two runs side by side: LDS is short there; five bits are used)
    // LDS (sweep kernel, may be null): per diagonal stage the thread-only phase factor in two tables, one by the
    // low half of the thread bits and one by the high half; valid while no phase slot is patched
    const double2* ftab;
    int T, nthr;
    int wave, tmask;   // uniform: wave of the workgroup, (1 << T) - 1
    // Thread ids are worked out where they are needed (two instructions) instead of
    // being carried through the phase loop in registers.
    __device__ __forceinline__ int lane() const {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    }
    __device__ __forceinline__ int tid() const { return wave * 64 + lane(); }
    __device__ __forceinline__ int t() const { return tid() & tmask; }
    int xc_id, xc_wt, xc_rt;      // cached per-thread offsets of one transposition descriptor
    __amdgpu_buffer_rsrc_t area;  // sweep: the workgroup's state images (checkpoints, final state)
    unsigned image_bytes;         // size of one image
#ifdef VQA_PROF
    int flag_mask;
#endif
};

// One transposition through LDS (descriptor xid).  Register bit 3 keeps its
// place, so the halves of the state with that bit 0 / 1 go through the same
// half-size buffer one after the other.
__device__ __forceinline__ void do_xpose(double2 (&amp)[kRegAmps], const Prog& pp, Ctx& cx, int xid) {
    const VQA_CAS XDesc* xd = pp.xdesc + xid;
    if (xid != cx.xc_id) {
        int wt = 0, rt = 0;
        const int t = cx.t();
        for (int i = 0; i < cx.T; ++i) {
            const int m = -((t >> i) & 1);
            wt ^= m & xd->wconst[kRegBits + i];
            rt ^= m & xd->rconst[kRegBits + i];
        }
        cx.xc_id = xid;
        cx.xc_wt = wt;
        cx.xc_rt = rt;
    }
    const int wt = cx.xc_wt, rt = cx.xc_rt;
    const int w0 = xd->wconst[0], w1 = xd->wconst[1], w2 = xd->wconst[2];
    const int r0 = xd->rconst[0], r1 = xd->rconst[1], r2 = xd->rconst[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int r = 0; r < kRegAmps / 2; ++r) {
            const int off = wt ^ ((r & 1) ? w0 : 0) ^ ((r & 2) ? w1 : 0) ^ ((r & 4) ? w2 : 0);
            *(double2*)(cx.xbuf + off) = amp[r + 8 * half];
        }
        if (!(VQA_EXP & 64)) __syncthreads();
#pragma unroll
        for (int r = 0; r < kRegAmps / 2; ++r) {
            const int off = rt ^ ((r & 1) ? r0 : 0) ^ ((r & 2) ? r1 : 0) ^ ((r & 4) ? r2 : 0);
            amp[r + 8 * half] = *(const double2*)(cx.xbuf + off);
        }
        if (!(VQA_EXP & 64)) __syncthreads();
    }
}

// One diagonal stage, the form for the 1024-thread workgroups (128 registers per thread): the thread-only
// factor bit by bit, the CZ / Z signs as one 16-bit mask per thread (bit r flips entry r).  The table form
// below needs a few registers more, which there end up in scratch inside the phase loop.
__device__ __forceinline__ void do_diag_masked(double2 (&amp)[kRegAmps], const Prog& pp, const Ctx& cx, int did) {
    // the whole descriptor in one go (field by field each scalar load is waited for where it is used)
    static_assert(sizeof(DDesc) == 48, "DDesc is fetched as three 16-byte scalar loads");
    const VQA_CAS i32x4* ddv = (const VQA_CAS i32x4*)(pp.ddesc + did);
    const i32x4 dd0 = ddv[0], dd1 = ddv[1], dd2 = ddv[2];
    struct { int ph1_base, tt_begin, tt_count, rt_begin, rt_count, rr_begin, rr_count, sign_id; } ddl =
        { dd0.x, dd0.y, dd0.z, dd0.w, dd1.x, dd1.y, dd1.z, dd1.w };
    (void)dd2;
    const auto* dd = &ddl;
    const double2* p1 = cx.ph1 + dd->ph1_base;
    const int t = cx.t();
    // thread-only factor
    cplx F = { 1.0, 0.0 };
    for (int i = 0; i < cx.T; ++i) {
        const double2 p = p1[kRegBits + i];
        F = cmul(F, csel_conj({ p.x, p.y }, (unsigned)(t >> i) & 1u));
    }
    const int tt_begin = dd->tt_begin, tt_count = dd->tt_count;
    for (int k = 0; k < tt_count; ++k) {
        const int pa = pp.pairs[tt_begin + k].a, pb = pp.pairs[tt_begin + k].b;
        const double2 z = cx.ph2[pp.pairs[tt_begin + k].ph2];
        F = cmul(F, csel_conj({ z.x, z.y }, (unsigned)((t >> pa) ^ (t >> pb)) & 1u));
    }
    // effective phase of each register bit
    cplx pj[kRegBits];
#pragma unroll
    for (int j = 0; j < kRegBits; ++j) {
        const double2 p = p1[j];
        pj[j] = { p.x, p.y };
    }
    const int rt_begin = dd->rt_begin, rt_count = dd->rt_count;
    for (int k = 0; k < rt_count; ++k) {
        const int pa = pp.pairs[rt_begin + k].a, pb = pp.pairs[rt_begin + k].b;
        const double2 z = cx.ph2[pp.pairs[rt_begin + k].ph2];
        const cplx ze = csel_conj({ z.x, z.y }, (unsigned)(t >> pb) & 1u);
#pragma unroll
        for (int j = 0; j < kRegBits; ++j)
            if (pa == j) pj[j] = cmul(pj[j], ze);
    }
    // phase table by doubling, T[r | 1<<j] = T[r] * conj(pj)^2: the lower 8 entries
    // are kept, the upper 8 are formed on the fly to save registers
    cplx tb[kRegAmps / 2];
    tb[0] = cmul(cmul(F, cmul(pj[0], pj[1])), cmul(pj[2], pj[3]));
#pragma unroll
    for (int j = 0; j < kRegBits - 1; ++j) {
        const cplx cj = cconj(pj[j]);
        const cplx w = cmul(cj, cj);
#pragma unroll
        for (int r = 0; r < (1 << j); ++r) tb[r | (1 << j)] = cmul(tb[r], w);
    }
    const cplx c3 = cconj(pj[kRegBits - 1]);
    const cplx w3 = cmul(c3, c3);
    // CZ / Z signs: bit r of the thread's mask flips entry r
    const int sign_id = dd->sign_id;
    unsigned sm = 0;
    if (sign_id >= 0) sm = cx.smask[sign_id * cx.nthr + cx.tid()];
#pragma unroll
    for (int r = 0; r < kRegAmps / 2; ++r) {
        const int flip_lo = (int)((sm << (31 - r)) & 0x80000000u);
        const int flip_hi = (int)((sm << (31 - (r + 8))) & 0x80000000u);
        const cplx hi = cmul(tb[r], w3);
        cmul_inplace(amp[r].x, amp[r].y,
                     __hiloint2double(__double2hiint(tb[r].x) ^ flip_lo, __double2loint(tb[r].x)),
                     __hiloint2double(__double2hiint(tb[r].y) ^ flip_lo, __double2loint(tb[r].y)));
        cmul_inplace(amp[r + 8].x, amp[r + 8].y,
                     __hiloint2double(__double2hiint(hi.x) ^ flip_hi, __double2loint(hi.x)),
                     __hiloint2double(__double2hiint(hi.y) ^ flip_hi, __double2loint(hi.y)));
    }
    // register-register ZZ phases
    const int rr_begin = dd->rr_begin, rr_count = dd->rr_count;
    for (int k = 0; k < rr_count; ++k) {
        const int pa = pp.pairs[rr_begin + k].a, pb = pp.pairs[rr_begin + k].b;
        const double2 z = cx.ph2[pp.pairs[rr_begin + k].ph2];
#pragma unroll
        for (int r = 0; r < kRegAmps; ++r) {
            const cplx ze = csel_conj({ z.x, z.y }, (unsigned)((r >> pa) ^ (r >> pb)) & 1u);
            cmul_inplace(amp[r].x, amp[r].y, ze.x, ze.y);
        }
    }
}


// One diagonal stage (descriptor did).
// SCALES: phase slots may carry the scales of two-shear rotations (see compute_uentry): a slot P stands for
// the factor P on |0> and conj(P) / |P|^2 on |1>; without, every slot has modulus one and the latter is conj(P).
template <bool SCALES = false>
__device__ __forceinline__ void do_diag(double2 (&amp)[kRegAmps], const Prog& pp, const Ctx& cx, int did, bool use_ftab) {
    // the whole descriptor in one go (field by field each scalar load is waited for where it is used)
    static_assert(sizeof(DDesc) == 48, "DDesc is fetched as three 16-byte scalar loads");
    const VQA_CAS i32x4* ddv = (const VQA_CAS i32x4*)(pp.ddesc + did);
    const i32x4 dd0 = ddv[0], dd1 = ddv[1], dd2 = ddv[2];
    struct { int ph1_base, tt_begin, tt_count, rt_begin, rt_count, rr_begin, rr_count, sign_id; } ddl =
        { dd0.x, dd0.y, dd0.z, dd0.w, dd1.x, dd1.y, dd1.z, dd1.w };
    (void)dd2;
    const auto* dd = &ddl;
    const double2* p1 = cx.ph1 + dd->ph1_base;
    const int t = cx.t();
    // CZ / Z signs: sign(r, t) = base(t) ^ XOR_j (r_j & cross_j(t)) ^ rr(r).  The thread's bits (base, cross_0..3)
    // go into the table below -- base flips the thread-only factor, cross_j the step of register bit j --
    // and cost ten sign flips; rr is the same for every thread and is applied at the end, when there is one
    const int sign_id = dd->sign_id;
    unsigned sm = 0, rr = 0;
    if (sign_id >= 0) {
        sm = cx.smask[sign_id * cx.nthr + cx.tid()];
        rr = pp.signs[sign_id].rr_par;
    }
    // The first table entry (thread-only factor times the register bits' phases) and the step of each register
    // bit, conj(p_j)^2: from the stage's tables (the thread-only factor in two halves, the low one with the
    // register bits' phases folded in, and the four steps), or worked out bit by bit when a slot is patched
    // or a ZZ phase between a register and a thread bit makes the steps differ from thread to thread
    const int tt_begin = dd->tt_begin, tt_count = dd->tt_count;
    const int rt_begin = dd->rt_begin, rt_count = dd->rt_count;
    // (two separate blocks up to the finished table: what is alive where they meet is the table alone)
    cplx tb[kRegAmps / 2], w3;
    auto step_of = [&](cplx w, int j) {   // the step of register bit j with the thread's flip
        const unsigned cr = (sm >> (1 + j)) & 1u;
        return cplx{ flip_if(w.x, cr), flip_if(w.y, cr) };
    };
    auto thread_pairs = [&](cplx T0) {    // ZZ phases between thread bits, the thread's own sign
        T0 = { flip_if(T0.x, sm & 1u), flip_if(T0.y, sm & 1u) };
        for (int k = 0; k < tt_count; ++k) {
            const int pa = pp.pairs[tt_begin + k].a, pb = pp.pairs[tt_begin + k].b;
            const double2 z = cx.ph2[pp.pairs[tt_begin + k].ph2];
            T0 = cmul(T0, csel_conj({ z.x, z.y }, (unsigned)((t >> pa) ^ (t >> pb)) & 1u));
        }
        return T0;
    };
    if (use_ftab && rt_count == 0) {
        const int TL = cx.T >> 1, FE = (1 << TL) + (1 << (cx.T - TL));
        const double2* ft = cx.ftab + (size_t)did * (FE + kRegBits);
        const double2 fa = ft[t & ((1 << TL) - 1)], fb = ft[(1 << TL) + (t >> TL)];
        tb[0] = thread_pairs(cmul({ fa.x, fa.y }, { fb.x, fb.y }));
        // phase table by doubling, T[r | 1<<j] = T[r] * conj(pj)^2: the lower 8 entries
        // are kept, the upper 8 are formed on the fly to save registers
#pragma unroll
        for (int j = 0; j < kRegBits - 1; ++j) {
            const double2 wj = ft[FE + j];
            const cplx w = step_of({ wj.x, wj.y }, j);
#pragma unroll
            for (int r = 0; r < (1 << j); ++r) tb[r | (1 << j)] = cmul(tb[r], w);
        }
        const double2 wl = ft[FE + kRegBits - 1];
        w3 = step_of({ wl.x, wl.y }, kRegBits - 1);
    } else {
        cplx F = { 1.0, 0.0 };
        for (int i = 0; i < cx.T; ++i) {
            const double2 p = p1[kRegBits + i];
            const unsigned bit = (unsigned)(t >> i) & 1u;
            cplx f = csel_conj({ p.x, p.y }, bit);
            if (SCALES) {
                const double sc = bit ? 1.0 / (p.x * p.x + p.y * p.y) : 1.0;
                f = { f.x * sc, f.y * sc };
            }
            F = cmul(F, f);
        }
        // effective phase of each register bit (with SCALES: and one over its squared modulus)
        cplx pj[kRegBits];
        double ij[kRegBits];
#pragma unroll
        for (int j = 0; j < kRegBits; ++j) {
            const double2 p = p1[j];
            pj[j] = { p.x, p.y };
            ij[j] = SCALES ? 1.0 / (p.x * p.x + p.y * p.y) : 1.0;
        }
        for (int k = 0; k < rt_count; ++k) {
            const int pa = pp.pairs[rt_begin + k].a, pb = pp.pairs[rt_begin + k].b;
            const double2 z = cx.ph2[pp.pairs[rt_begin + k].ph2];
            const cplx ze = csel_conj({ z.x, z.y }, (unsigned)(t >> pb) & 1u);
#pragma unroll
            for (int j = 0; j < kRegBits; ++j)
                if (pa == j) pj[j] = cmul(pj[j], ze);
        }
        tb[0] = thread_pairs(cmul(cmul(F, cmul(pj[0], pj[1])), cmul(pj[2], pj[3])));
        // the step of register bit j: (factor on |1>) / (factor on |0>) = (conj(p) / |p|^2)^2
        auto step_raw = [&](int j) {
            cplx cj = cconj(pj[j]);
            if (SCALES) cj = { cj.x * ij[j], cj.y * ij[j] };
            return cmul(cj, cj);
        };
#pragma unroll
        for (int j = 0; j < kRegBits - 1; ++j) {
            const cplx w = step_of(step_raw(j), j);
#pragma unroll
            for (int r = 0; r < (1 << j); ++r) tb[r | (1 << j)] = cmul(tb[r], w);
        }
        w3 = step_of(step_raw(kRegBits - 1), kRegBits - 1);
    }
#pragma unroll
    for (int r = 0; r < kRegAmps / 2; ++r) {
        const cplx hi = cmul(tb[r], w3);
        cmul_inplace(amp[r].x, amp[r].y, tb[r].x, tb[r].y);
        cmul_inplace(amp[r + 8].x, amp[r + 8].y, hi.x, hi.y);
    }
    if (rr) {
        // signs between register bits: a uniform +-1 per amplitude
#pragma unroll
        for (int r = 0; r < kRegAmps; ++r) {
            const double sg = ((rr >> r) & 1u) ? -1.0 : 1.0;
            asm("v_mul_f64 %0, %0, %2\n\tv_mul_f64 %1, %1, %2" : "+v"(amp[r].x), "+v"(amp[r].y) : "s"(sg));
        }
    }
    // register-register ZZ phases
    const int rr_begin = dd->rr_begin, rr_count = dd->rr_count;
    for (int k = 0; k < rr_count; ++k) {
        const int pa = pp.pairs[rr_begin + k].a, pb = pp.pairs[rr_begin + k].b;
        const double2 z = cx.ph2[pp.pairs[rr_begin + k].ph2];
#pragma unroll
        for (int r = 0; r < kRegAmps; ++r) {
            const cplx ze = csel_conj({ z.x, z.y }, (unsigned)((r >> pa) ^ (r >> pb)) & 1u);
            cmul_inplace(amp[r].x, amp[r].y, ze.x, ze.y);
        }
    }
}

// Runs phases[p0 ..] to PH_END.  diag_only_first: the first phase starts at its
// diagonal stage (resuming from a checkpoint taken in front of it).  When ckpt
// is set the state in front of every diagonal stage d is stored as image d of
// the workgroup's area.
struct NoHook {
    __device__ __forceinline__ void operator()(double2 (&)[kRegAmps], int) const {}
};
// hook(amp, d): called in front of diagonal stage d (grouped sweeps: checkpoints in LDS, late corrections)
// TWO: register-bit entries may be of the two-shear form, phase slots may carry their scales
template <bool SCALED, bool MASKED = false, class Hook = NoHook, bool TWO = false>
__device__ __forceinline__ void run_phases(double2 (&amp)[kRegAmps], const Prog& pp, Ctx& cx, int p0,
                                           bool diag_only_first, bool ckpt, bool use_ftab = false, Hook hook = Hook()) {
    const VQA_CAS PhaseOp* ph = pp.phases;
    int first_mask = diag_only_first ? (PH_DIAG | PH_END) : -1;
    VQA_STAMP0
    for (int p = p0;; ++p) {
        // the whole descriptor in one go (scalar loads issued together, one wait)
        const VQA_CAS i32x4* dsc = (const VQA_CAS i32x4*)(ph + p);
        const i32x4 d0 = dsc[0], d1 = dsc[1], d2 = dsc[2], d3 = dsc[3];
        const int ent[10] = { d0.w, d1.x, d1.y, d1.z, d1.w, d2.x, d2.y, d2.z, d2.w, d3.x };
#ifdef VQA_PROF
        const int flags = d0.x & first_mask & cx.flag_mask;
#else
        constexpr int exp_mask = ~(((VQA_EXP & 128) ? PH_XPOSE : 0) | ((VQA_EXP & 256) ? PH_UREG0 * 15 : 0) |
                                   ((VQA_EXP & 512) ? PH_ULANE0 * 15 : 0) | ((VQA_EXP & 1024) ? PH_ULANE0 * 0x30 : 0) |
                                   ((VQA_EXP & 2048) ? PH_DIAG : 0));
        const int flags = d0.x & first_mask & (ckpt ? -1 : exp_mask);
#endif
        first_mask = -1;
        if (flags & PH_END) break;
        VQA_STAMP(9)
        if (flags & PH_XPOSE) do_xpose(amp, pp, cx, d0.y);
        VQA_STAMP(10)
        // rotations: the table entry of the next slot is fetched while the current one runs
        double2 cs = cx.mat[ent[0]], csn;
#define VQA_REG_SLOT(J)                                           \
        csn = cx.mat[ent[J + 1]];                                 \
        if constexpr (!TWO) {                                     \
            if (flags & (PH_UREG0 << J)) apply_ureg<J>(amp, cs.x, cs.y); \
        } else {                                                  \
            /* (two optional blocks, not an if / else: each updates the amplitudes in place) */ \
            const int two = __builtin_amdgcn_readfirstlane(__double2hiint(cs.y)) < 0; \
            const int run = flags & (PH_UREG0 << J);              \
            if (run && !two) apply_ureg<J>(amp, cs.x, cs.y);      \
            if (run && two) apply_ureg2<J>(amp, cs.x, cs.y);      \
        }                                                         \
        cs = csn;
        VQA_REG_SLOT(0) VQA_REG_SLOT(1) VQA_REG_SLOT(2) VQA_REG_SLOT(3)
#undef VQA_REG_SLOT
        VQA_STAMP(11)
        // (with one state per workgroup the two scaled forms are separate optional
        // blocks, not an if / else: each block then updates the amplitudes in place)
#define VQA_LANE_SLOT(LB)                                                  \
        csn = cx.mat[ent[LB + 5 < 10 ? LB + 5 : 9]];                       \
        if constexpr (!SCALED) {                                           \
            if (flags & (PH_ULANE0 << LB)) apply_ulane<LB, 0>(amp, cs.x, cs.y, cx.lane()); \
        } else {                                                           \
            const int form_b = __builtin_amdgcn_readfirstlane(__double2hiint(cs.y)); \
            const int run = flags & (PH_ULANE0 << LB);                     \
            if (run && form_b == 0) apply_ulane<LB, 1>(amp, cs.x, cs.y, cx.lane()); \
            if (run && form_b != 0) apply_ulane<LB, 2>(amp, cs.x, cs.y, cx.lane()); \
        }                                                                  \
        cs = csn;
        VQA_LANE_SLOT(0) VQA_LANE_SLOT(1) VQA_LANE_SLOT(2) VQA_LANE_SLOT(3)
#undef VQA_LANE_SLOT
        // lane bits 4 and 5: both rotations in one pass through the matrix unit (cs = entry of bit 4)
        if (flags & (PH_ULANE0 * 0x30)) {
            const double2 c5 = cx.mat[ent[9]];
            const bool h4 = (flags & (PH_ULANE0 << 4)) != 0, h5 = (flags & (PH_ULANE0 << 5)) != 0;
            apply_ulane45(amp, h4 ? cs.x : 1.0, h4 ? cs.y : 0.0, h5 ? c5.x : 1.0, h5 ? c5.y : 0.0, cx.lane());
        }
        VQA_STAMP(12)
        if (flags & PH_DIAG) {
            const int did = d0.z;
            if (ckpt) store_state(amp, cx.area, (unsigned)did * cx.image_bytes, cx.nthr, cx.tid());
            hook(amp, did);
            if constexpr (MASKED) do_diag_masked(amp, pp, cx, did);
            else do_diag<TWO>(amp, pp, cx, did, use_ftab);
        }
        VQA_STAMP(13)
    }
}

// Sum of v over the 64 lanes of a wave (every lane gets it): DPP inside the rows
// of 16 lanes, then the four row sums through scalar registers.
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

// Sums NV per-thread values over the threads of a state.  T >= 6: a state is a whole
// workgroup; wave sums meet in LDS (red, two alternating halves so that one barrier
// per call is enough as long as calls are separated by another barrier).
template <int NV>
__device__ __forceinline__ void state_sum(double (&v)[NV], const Ctx& cx, double* red, unsigned& parity) {
    if (cx.T < 6) {
#pragma unroll
        for (int k = 0; k < NV; ++k)
            for (int i = 0; i < cx.T; ++i) v[k] += __shfl_xor(v[k], 1 << i, 64);
        return;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
    if (cx.T > 6) {
        const int wave = cx.wave, nwave = cx.nthr >> 6;
        double* buf = red + (parity & 1u) * 64;
        parity = __builtin_amdgcn_readfirstlane(parity ^ 1u);   // (uniform: lives in a scalar register)
        if (cx.lane() == 0) {
#pragma unroll
            for (int k = 0; k < NV; ++k) buf[k * 16 + wave] = v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            v[k] = 0.0;
            for (int w = 0; w < nwave; ++w) v[k] += buf[k * 16 + w];
        }
    }
}

// Qubit-index bits the thread / each register bit stands for at measurement time
// (worked out where needed: keeping them across the phase loop costs registers).
__device__ __forceinline__ void measure_masks(const Prog& pp, int T, int t, unsigned& kt, unsigned (&krb)[kRegBits]) {
    kt = 0;
    for (int i = 0; i < T; ++i)
        if ((t >> i) & 1) kt |= 1u << pp.H->final_qubit_at_pos[kRegBits + i];
#pragma unroll
    for (int j = 0; j < kRegBits; ++j) krb[j] = 1u << pp.H->final_qubit_at_pos[j];
}

// Z-basis observables and the low-bit marginal of the state held in amp, which
// may carry any common factor: every value is divided by the squared norm.
// Instance index e, written by the thread with t == 0 (and kept in basev when
// given).  Returns the squared norm of the state as held.
__device__ __forceinline__ double measure(const double2 (&amp)[kRegAmps], const RegArgs& a, const Prog& pp, const Ctx& cx,
                                          double* red, unsigned& parity, int64_t e, bool live, bool valid,
                                          double* basev) {
    const double nanv = __longlong_as_double(0x7ff8000000000000LL);
    unsigned kt, krb[kRegBits];
    const int t = cx.t();
    measure_masks(pp, cx.T, t, kt, krb);
    const int nq = a.obs.n_obs + (a.n_marg_bits > 0 ? (1 << a.n_marg_bits) : 0);
    double pr[kRegAmps];
#pragma unroll
    for (int r = 0; r < kRegAmps; ++r) pr[r] = amp[r].x * amp[r].x + amp[r].y * amp[r].y;
    double inv_norm = 1.0;
    for (int q = 0; q < nq; ++q) {
        double v[2] = { 0.0, 0.0 };   // the quantity; with the first one the norm
        if (q == 0) {
#pragma unroll
            for (int r = 0; r < kRegAmps; ++r) v[1] += pr[r];
        }
        if (q < a.obs.n_obs) {
            for (int tm = a.obs.off[q]; tm < a.obs.off[q + 1]; ++tm) {
                const unsigned z = (unsigned)a.obs.z[tm];
                const unsigned tpar = __popc(kt & z);
                double part = 0.0;
#pragma unroll
                for (int r = 0; r < kRegAmps; ++r) {
                    unsigned rp = tpar;
#pragma unroll
                    for (int j = 0; j < kRegBits; ++j)
                        if (r & (1 << j)) rp += __popc(krb[j] & z);
                    const double sg = __hiloint2double((int)(0x3FF00000u | ((rp & 1u) << 31)), 0);
                    part = fma(sg, pr[r], part);
                }
                v[0] += a.obs.coef[tm] * part;
            }
        } else {
            const unsigned m = (unsigned)(q - a.obs.n_obs), M1 = (1u << a.n_marg_bits) - 1;
#pragma unroll
            for (int r = 0; r < kRegAmps; ++r) {
                unsigned k = kt;
#pragma unroll
                for (int j = 0; j < kRegBits; ++j)
                    if (r & (1 << j)) k |= krb[j];
                v[0] = fma((k & M1) == m ? 1.0 : 0.0, pr[r], v[0]);
            }
        }
        if (q == 0) {
            state_sum<2>(v, cx, red, parity);
            inv_norm = 1.0 / v[1];
        } else {
            double w[1] = { v[0] };
            state_sum<1>(w, cx, red, parity);
            v[0] = w[0];
        }
        const double val = v[0] * inv_norm;
        if (t == 0 && live) {
            if (q < a.obs.n_obs) a.out_expect[e * a.obs.n_obs + q] = valid ? val : nanv;
            else a.out_marginal[(e << a.n_marg_bits) + (q - a.obs.n_obs)] = valid ? val : nanv;
            if (basev) basev[q] = val;
        }
    }
    return 1.0 / inv_norm;
}

// The kernel's argument block once more, for values that are needed once per sample: what is read through this
// pointer is loaded behind the call, not held in scalar registers around the run loop (the sweep kernels sit at
// the limit of those, and one more live value there costs 3 % of the headline rate in spills).
__device__ __forceinline__ const VQA_CAS RegArgs* late_args() {
    const VQA_CAS void* p = (const VQA_CAS void*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (const VQA_CAS RegArgs*)p;
}

// Rows 2i+1 and 2i+2 of the sample whose row 0 is instance row0, for the parameters i of the call's dead list:
// copies of the unshifted values in basev (what a separate pass over the output did before: one launch less per
// sweep).
__device__ __forceinline__ void fill_dead_params(const double* basev, int64_t row0, bool valid, int tid, int nthr) {
    const VQA_CAS RegArgs* la = late_args();
    const int n_dead = la->n_dead;
    if (n_dead <= 0) return;
    const double nanv = __longlong_as_double(0x7ff8000000000000LL);
    const int nmb = la->n_marg_bits, no = la->obs.n_obs, nq = no + (nmb > 0 ? (1 << nmb) : 0);
    const int32_t* dead = la->dead;
    double* oe = la->out_expect;
    double* om = la->out_marginal;
    __syncthreads();   // the unshifted values are thread 0's
    for (int idx = tid; idx < n_dead * nq; idx += nthr) {
        const int d = idx / nq, q = idx - d * nq;
        const int64_t e1 = row0 + 2 * (int64_t)dead[d] + 1;
        const double v = valid ? basev[q] : nanv;
        if (q < no) {
            oe[e1 * no + q] = v;
            oe[(e1 + 1) * no + q] = v;
        } else {
            om[(e1 << nmb) + (q - no)] = v;
            om[((e1 + 1) << nmb) + (q - no)] = v;
        }
    }
}

// Both shifted values of one parameter from ONE extra state.  With psi the
// unshifted final state and phi the final state of the circuit that has the
// parameter's generator inserted in front of its rotation, the states at
// +-kappa are cos(kappa/2) psi +- i sin(kappa/2) phi, so for a diagonal O
//   <O>(+-) = cos^2 <O>_psi + sin^2 <O>_phi -+ 2 cos sin Im <psi|O|phi>.
// amp holds phi (any scale), image psi_image of the area the stored psi (scale^2 =
// psi_norm), basev the unshifted values; rec (LDS) gives rho = the ratio of the
// unit factors the engine left out of the two runs and (cos, sin)(kappa/2);
// e1 = instance index of +kappa (-kappa is e1 + 1).
// psi_lds != nullptr: psi is read from LDS instead, amplitude r of the state's thread t at
// psi_lds[(r << T) + t] (small circuits, where the states of a workgroup share one sample's psi).
__device__ __forceinline__ void measure_pair(double2 (&amp)[kRegAmps], unsigned psi_image, double psi_norm,
                                             const double2* rec, const double* basev, const RegArgs& a, const Prog& pp,
                                             const Ctx& cx, double* red, unsigned& parity, int64_t e1, bool valid,
                                             const double2* psi_lds = nullptr) {
    const double nanv = __longlong_as_double(0x7ff8000000000000LL);
    unsigned kt, krb[kRegBits];
    const int t = cx.t();
    measure_masks(pp, cx.T, t, kt, krb);
    const int nq = a.obs.n_obs + (a.n_marg_bits > 0 ? (1 << a.n_marg_bits) : 0);
    const unsigned toff = cx.tid() * (unsigned)sizeof(double2), rstep = cx.nthr * (unsigned)sizeof(double2);
    double pr[kRegAmps];
    // amp <- conj(psi) phi, pr <- |phi|^2; the loads of psi run four ahead.  (The
    // scheduling fences keep that order: left alone, the compiler issues all sixteen
    // loads at once and spills the state to make room for them.)
    if (psi_lds) {
#pragma unroll
        for (int r = 0; r < kRegAmps; ++r) {
            const double2 ps = psi_lds[(r << cx.T) + t];
            const double2 ph = amp[r];
            pr[r] = ph.x * ph.x + ph.y * ph.y;
            amp[r] = make_double2(ps.x * ph.x + ps.y * ph.y, ps.x * ph.y - ps.y * ph.x);
        }
    }
    u32x4 pv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (!psi_lds && !(VQA_EXP & 2)) pv[r] = __builtin_amdgcn_raw_buffer_load_b128(cx.area, toff, psi_image + r * rstep, VQA_IMAGE_AUX);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < kRegAmps; ++r) {
        if (psi_lds) break;
        const u32x4 cur = pv[r & 3];
        const double2 ps = (VQA_ABL(a, 10) || (VQA_EXP & 2)) ? make_double2(1.0, 0.5) : *(const double2*)&cur;
        const double2 ph = amp[r];
        pr[r] = ph.x * ph.x + ph.y * ph.y;
        amp[r] = make_double2(ps.x * ph.x + ps.y * ph.y, ps.x * ph.y - ps.y * ph.x);
        if (r + 4 < kRegAmps && !(VQA_EXP & 2))
            pv[r & 3] = __builtin_amdgcn_raw_buffer_load_b128(cx.area, toff, psi_image + (r + 4) * rstep, VQA_IMAGE_AUX);
        __builtin_amdgcn_sched_barrier(0);
    }
    double inv_norm = 1.0, inv_cross = 1.0;
    for (int q = 0; q < nq; ++q) {
        double v[4] = { 0.0, 0.0, 0.0, 0.0 };   // sum o |phi|^2, Re / Im sum o conj(psi) phi, norm
        if (q == 0) {
#pragma unroll
            for (int r = 0; r < kRegAmps; ++r) v[3] += pr[r];
        }
        if (q < a.obs.n_obs) {
            for (int tm = a.obs.off[q]; tm < a.obs.off[q + 1]; ++tm) {
                const unsigned z = (unsigned)a.obs.z[tm];
                const unsigned tpar = __popc(kt & z);
                double p0 = 0.0, p1 = 0.0, p2 = 0.0;
#pragma unroll
                for (int r = 0; r < kRegAmps; ++r) {
                    unsigned rp = tpar;
#pragma unroll
                    for (int j = 0; j < kRegBits; ++j)
                        if (r & (1 << j)) rp += __popc(krb[j] & z);
                    // (a +-1 weight in a fused multiply-add: sign-flipped copies of the
                    // values would all be alive at once and spill)
                    const double sg = __hiloint2double((int)(0x3FF00000u | ((rp & 1u) << 31)), 0);
                    p0 = fma(sg, pr[r], p0);
                    p1 = fma(sg, amp[r].x, p1);
                    p2 = fma(sg, amp[r].y, p2);
                }
                const double cf = a.obs.coef[tm];
                v[0] += cf * p0; v[1] += cf * p1; v[2] += cf * p2;
            }
        } else {
            const unsigned m = (unsigned)(q - a.obs.n_obs), M1 = (1u << a.n_marg_bits) - 1;
#pragma unroll
            for (int r = 0; r < kRegAmps; ++r) {
                unsigned k = kt;
#pragma unroll
                for (int j = 0; j < kRegBits; ++j)
                    if (r & (1 << j)) k |= krb[j];
                const double w = (k & M1) == m ? 1.0 : 0.0;
                v[0] = fma(w, pr[r], v[0]);
                v[1] = fma(w, amp[r].x, v[1]);
                v[2] = fma(w, amp[r].y, v[2]);
            }
        }
        if (q == 0) {
            state_sum<4>(v, cx, red, parity);
            inv_norm = 1.0 / v[3];
            inv_cross = rsqrt(v[3] * psi_norm);
        } else {
            double w[3] = { v[0], v[1], v[2] };
            state_sum<3>(w, cx, red, parity);
            v[0] = w[0]; v[1] = w[1]; v[2] = w[2];
        }
        if (t == 0) {
            const bool is_obs = q < a.obs.n_obs;
            const size_t o1 = is_obs ? (size_t)e1 * a.obs.n_obs + q : ((size_t)e1 << a.n_marg_bits) + (q - a.obs.n_obs);
            const size_t step = is_obs ? (size_t)a.obs.n_obs : ((size_t)1 << a.n_marg_bits);
            double* out = is_obs ? a.out_expect : a.out_marginal;
            const double2 rho = rec[3], cs = rec[4];
            const double im = (rho.x * v[2] + rho.y * v[1]) * inv_cross;   // Im(rho * sum)
            const double mid = cs.x * cs.x * basev[q] + cs.y * cs.y * v[0] * inv_norm;
            const double dev = 2.0 * cs.x * cs.y * im;
            out[o1] = valid ? mid - dev : nanv;
            out[o1 + step] = valid ? mid + dev : nanv;
        }
    }
}

// per-thread constants of a launch: sign masks of every sign table (LDS)
template <int BLOCK>
__device__ __forceinline__ void thread_constants(const Prog& pp, unsigned short* smask, int T, int t, int tid) {
    const int n_sign = pp.H->n_sign;
    for (int sg = 0; sg < n_sign; ++sg) {
        const int wi = t >> 5, bi = t & 31;
        const unsigned base = (pp.signs[sg].tt_par[wi] >> bi) & 1u;
        unsigned c[kRegBits];
#pragma unroll
        for (int j = 0; j < kRegBits; ++j) c[j] = (pp.signs[sg].cross[j][wi] >> bi) & 1u;
        unsigned m = 0;
        if (BLOCK == 1024) {
            // the masked form (do_diag_masked): bit r = sign of entry r
            const unsigned rr = pp.signs[sg].rr_par;
#pragma unroll
            for (int r = 0; r < kRegAmps; ++r) {
                unsigned sb = base ^ ((rr >> r) & 1u);
#pragma unroll
                for (int j = 0; j < kRegBits; ++j)
                    if (r & (1 << j)) sb ^= c[j];
                m |= sb << r;
            }
        } else {
            // bit 0: the thread's own parity; bit 1 + j: the flip that register bit j brings (the signs between
            // register bits, rr_par, are the same for every thread and are read where they are applied)
            m = base;
#pragma unroll
            for (int j = 0; j < kRegBits; ++j) m |= c[j] << (1 + j);
        }
        smask[sg * BLOCK + tid] = (unsigned short)m;
    }
}

// ---- explicit instance list ----------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK, BLOCK == 1024 ? 4 : BLOCK == 512 ? 2 : VQA_WAVES) void k_reg_eval(RegArgs a) {
    // blocks above one wave hold a single state: the scaled lane rotations apply
    constexpr bool SCALED = BLOCK > 64;
    extern __shared__ double2 smem[];
    const Prog pp = load_prog(a.prog);
    // (above one wave a workgroup is one state and the thread bits are its size: a constant unrolls the loops over them)
    const int T = BLOCK == 128 ? 7 : BLOCK == 256 ? 8 : BLOCK == 512 ? 9 : BLOCK == 1024 ? 10 : pp.H->T;
    const int n = pp.H->n, S = pp.H->S;
    const int NU = pp.H->n_uent, N1 = pp.H->n_d1, N2 = pp.H->n_d2;
    const int tab_stride = NU + N1 + N2;
    const int tid = threadIdx.x;
    const int sl = tid >> T;              // state within the workgroup
    const int t = tid & ((1 << T) - 1);   // thread bits of the amplitude index

    double2* tabs = smem;                                 // S * tab_stride
    double2* xbuf = tabs + (size_t)S * tab_stride;        // 2^(n-1) slots when the program transposes
    double* red = (double*)(xbuf + (pp.H->has_xpose ? ((size_t)1 << (n - 1)) : 0));  // 128 doubles
    unsigned short* smask = (unsigned short*)(red + 128);  // n_sign * BLOCK

    Ctx cx;
    cx.mat = tabs + (size_t)sl * tab_stride;
    cx.ph1 = cx.mat + NU;
    cx.ph2 = cx.ph1 + N1;
    cx.xbuf = (char*)xbuf;
    cx.smask = smask;
    cx.ftab = nullptr;
    cx.T = T; cx.nthr = BLOCK;
    cx.wave = __builtin_amdgcn_readfirstlane(tid >> 6); cx.tmask = (1 << T) - 1;
    cx.xc_id = -1; cx.xc_wt = 0; cx.xc_rt = 0;
    cx.area = __builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0x00020000);
    cx.image_bytes = 0;
#ifdef VQA_PROF
    cx.flag_mask = -1;
#endif

    thread_constants<BLOCK>(pp, smask, T, t, tid);
    // the input-free fused 2x2s at the unshifted parameters, once per launch (several states per
    // workgroup only: with one state the preparation is a single round of the workgroup either way)
    double2* ctab = nullptr;
    if (BLOCK == 64 && S > 1) {
        ctab = (double2*)((char*)smask + (((size_t)pp.H->n_sign * BLOCK * 2 + 15) & ~(size_t)15));
        for (int u = tid; u < NU; u += BLOCK) {
            if (!pp.uent[u].input_free) continue;
            double2 cs;
            cplx A, G, gp;
            compute_uentry<SCALED>(pp, u, a.theta, a.input, -1, 0.0, -1, cs, A, G, gp);
            ctab[3 * u] = cs;
            ctab[3 * u + 1] = make_double2(A.x, A.y);
            ctab[3 * u + 2] = make_double2(G.x, G.y);
        }
    }

    unsigned parity = 0;
    for (int64_t e0 = (int64_t)blockIdx.x * S; e0 < a.E; e0 += (int64_t)gridDim.x * S) {
        const int64_t e = e0 + sl;
        const bool live = e < a.E;
        bool valid = true;
        if (live) {
            int b_, sp_;
            double sv_;
            valid = decode_eval(a.spec, e, b_, sp_, sv_);
        }
        __syncthreads();
        prep_tables<false, SCALED>(a, pp, tabs, e0, S, NU, N1, N2, tid, BLOCK, ctab);
        __syncthreads();

        double2 amp[kRegAmps];
#pragma unroll
        for (int r = 0; r < kRegAmps; ++r) amp[r] = make_double2(0.0, 0.0);
        if (t == 0) amp[0].x = 1.0;
        run_phases<SCALED, BLOCK == 1024>(amp, pp, cx, 0, false, false);
        measure(amp, a, pp, cx, red, parity, e, live, valid, nullptr);
    }
}

// A one-qubit 2x2 C (complex) on the qubit at one position of the layout, in place (late resumption of a
// sweep run, see ParamInfo).  Register bit: the pairs of the thread's own amplitudes; lane bit: each lane holds
// one amplitude of the pair and fetches the other.
//   (a0, a1) <- (c00 a0 + c01 a1, c10 a0 + c11 a1)
__device__ __forceinline__ void cpair_inplace(double& a0x, double& a0y, double& a1x, double& a1y, double2 c00, double2 c01,
                                              double2 c10, double2 c11) {
    double t0, t1, t2, t3;
    asm("v_mul_f64 %4, %8, %0\n\tv_mul_f64 %5, %8, %1\n\tv_mul_f64 %6, %12, %0\n\tv_mul_f64 %7, %12, %1\n\t"
        "v_fma_f64 %4, -%9, %1, %4\n\tv_fma_f64 %5, %9, %0, %5\n\tv_fma_f64 %6, -%13, %1, %6\n\tv_fma_f64 %7, %13, %0, %7\n\t"
        "v_fma_f64 %4, %10, %2, %4\n\tv_fma_f64 %5, %10, %3, %5\n\tv_fma_f64 %6, %14, %2, %6\n\tv_fma_f64 %7, %14, %3, %7\n\t"
        "v_fma_f64 %0, -%11, %3, %4\n\tv_fma_f64 %1, %11, %2, %5\n\tv_fma_f64 %4, -%15, %3, %6\n\tv_fma_f64 %3, %15, %2, %7\n\t"
        "v_mov_b64 %2, %4"
        : "+v"(a0x), "+v"(a0y), "+v"(a1x), "+v"(a1y), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(c00.x), "v"(c00.y), "v"(c01.x), "v"(c01.y), "v"(c10.x), "v"(c10.y), "v"(c11.x), "v"(c11.y));
}
template <int J>
__device__ __forceinline__ void apply_creg(double2 (&amp)[kRegAmps], double2 c00, double2 c01, double2 c10, double2 c11) {
    constexpr int lo = (1 << J) - 1;
#pragma unroll
    for (int k = 0; k < kRegAmps / 2; ++k) {
        const int r0 = ((k & ~lo) << 1) | (k & lo), r1 = r0 | (1 << J);
        cpair_inplace(amp[r0].x, amp[r0].y, amp[r1].x, amp[r1].y, c00, c01, c10, c11);
    }
}
//   a <- cs a + co p  (cs, co: the lane's own and the partner's coefficient)
__device__ __forceinline__ void clane_inplace(double& ax, double& ay, double px, double py, double csx, double csy, double cox,
                                              double coy) {
    double t0, t1;
    asm("v_mul_f64 %2, %6, %0\n\tv_mul_f64 %3, %6, %1\n\t"
        "v_fma_f64 %2, -%7, %1, %2\n\tv_fma_f64 %3, %7, %0, %3\n\t"
        "v_fma_f64 %2, %8, %4, %2\n\tv_fma_f64 %3, %8, %5, %3\n\t"
        "v_fma_f64 %0, -%9, %5, %2\n\tv_fma_f64 %1, %9, %4, %3"
        : "+v"(ax), "+v"(ay), "=&v"(t0), "=&v"(t1)
        : "v"(px), "v"(py), "v"(csx), "v"(csy), "v"(cox), "v"(coy));
}
template <int LB>
__device__ __forceinline__ void apply_clane(double2 (&amp)[kRegAmps], double2 c00, double2 c01, double2 c10, double2 c11,
                                            int lane) {
    const bool up = (lane >> LB) & 1;
    const double csx = up ? c11.x : c00.x, csy = up ? c11.y : c00.y, cox = up ? c10.x : c01.x, coy = up ? c10.y : c01.y;
    // the cross-lane moves read registers that inline assembly may have written last
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < kRegAmps; ++r) {
        const double px = xor_fetch<LB, false>(amp[r].x), py = xor_fetch<LB, false>(amp[r].y);
        clane_inplace(amp[r].x, amp[r].y, px, py, csx, csy, cox, coy);
    }
}
// Wave bit WB (thread bit 6 + WB): the partner amplitudes belong to another wave and come through the
// transposition buffer, half of the state at a time (slot = [amplitude][thread]: no bank conflicts either way).
// Every thread of the workgroup takes part.
template <int WB>
__device__ __forceinline__ void apply_cwave(double2 (&amp)[kRegAmps], double2 c00, double2 c01, double2 c10, double2 c11,
                                            char* xbuf, int tid, int nthr) {
    const bool up = (tid >> (6 + WB)) & 1;
    const double csx = up ? c11.x : c00.x, csy = up ? c11.y : c00.y, cox = up ? c10.x : c01.x, coy = up ? c10.y : c01.y;
    double2* buf = (double2*)xbuf;
    const int partner = tid ^ (64 << WB);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int r = 0; r < kRegAmps / 2; ++r) buf[r * nthr + tid] = amp[r + 8 * half];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kRegAmps / 2; ++r) {
            const double2 p = buf[r * nthr + partner];
            clane_inplace(amp[r + 8 * half].x, amp[r + 8 * half].y, p.x, p.y, csx, csy, cox, coy);
        }
        __syncthreads();
    }
}
// mask = 1 << position (one bit per alternative, tested one after the other: each block updates the amplitudes in place)
__device__ __forceinline__ void apply_correction(double2 (&amp)[kRegAmps], const double2* rec, int mask, const Ctx& cx) {
    const double2 c00 = rec[0], c01 = rec[1], c10 = rec[2], c11 = rec[7];
    const int lane = cx.lane();
    if (mask & (1 << 10)) apply_cwave<0>(amp, c00, c01, c10, c11, cx.xbuf, cx.tid(), cx.nthr);
    if (mask & (1 << 11)) apply_cwave<1>(amp, c00, c01, c10, c11, cx.xbuf, cx.tid(), cx.nthr);
    if (mask & (1 << 12)) apply_cwave<2>(amp, c00, c01, c10, c11, cx.xbuf, cx.tid(), cx.nthr);
    if (mask & (1 << 13)) apply_cwave<3>(amp, c00, c01, c10, c11, cx.xbuf, cx.tid(), cx.nthr);
    if (mask & 1) apply_creg<0>(amp, c00, c01, c10, c11);
    if (mask & 2) apply_creg<1>(amp, c00, c01, c10, c11);
    if (mask & 4) apply_creg<2>(amp, c00, c01, c10, c11);
    if (mask & 8) apply_creg<3>(amp, c00, c01, c10, c11);
    if (mask & 16) apply_clane<0>(amp, c00, c01, c10, c11, lane);
    if (mask & 32) apply_clane<1>(amp, c00, c01, c10, c11, lane);
    if (mask & 64) apply_clane<2>(amp, c00, c01, c10, c11, lane);
    if (mask & 128) apply_clane<3>(amp, c00, c01, c10, c11, lane);
    if (mask & 256) apply_clane<4>(amp, c00, c01, c10, c11, lane);
    if (mask & 512) apply_clane<5>(amp, c00, c01, c10, c11, lane);
}

// ---- full parameter-shift sweeps ------------------------------------------------------
// a.E = number of samples; sample j = sel[j] (or j); outputs at instance
// j * (2P+1) + k.  One state per workgroup (BLOCK = 2^T >= 64 threads).
// Workgroups draw samples from a shared counter until it passes a.E, so the
// launch may hold more workgroups than are resident at once.
//
// Per sample: the tables of the unshifted circuit; one record per run of the
// sweep (what to patch, where to resume, how to measure), worked out by all
// threads side by side and parked in global memory; the unshifted run, which
// leaves a checkpoint in front of every diagonal stage and the final state;
// then the runs one after the other, each: thread 0 undoes the previous patch
// and applies this run's, everybody resumes from the checkpoint and measures.
// The record of the next run is fetched while the current one computes.
// bag: the (A, G) of the sample's unshifted fused 2x2s where the caller keeps them (else A is worked out again).
// LATE: 0 = patched tables; 1 = late resumption of the pair runs (sweep kernel); 2 = of every parameter that
// allows it, the ones of the last U-stage included (grouped sweeps: the run named is the +shift one of the pair)
template <int BLOCK, bool SCALED = true, int LATE = 0, bool TWO = false>
__device__ __forceinline__ void make_run_record(const RegArgs& a, const Prog& pp, int r, const double* row, int NU, int N1,
                                                RunRec* out, const double2* bag = nullptr) {
    const int i = pp.runs[r].param, variant = pp.runs[r].variant;
    if (LATE && (variant == 0 || (LATE == 2 && variant == 1)) && pp.params[i].late_mask != 0) {
        // resumption behind the parameter's own U-stage (see ParamInfo): C = M' M^-1, taken into the frame of
        // the checkpoint in front of the diagonal stage, whose table still holds this 2x2's factor
        // diag(A, conj A): C' = diag(A, conj A)^-1 C diag(A, conj A)
        const int index = pp.refs[pp.params[i].ref_begin].index;
        const double coef = pp.refs[pp.params[i].ref_begin].coef_sum;
        cplx c00, c01, c10, c11;
        late_generator(pp, index, a.theta, row, i, c00, c01, c10, c11);
        cplx A;
        if (bag) {
            A = { bag[2 * index].x, bag[2 * index].y };
        } else {
            double2 mv;
            cplx G, gp;
            compute_uentry<SCALED, TWO>(pp, index, a.theta, row, -1, 0.0, -1, mv, A, G, gp);
        }
        // (A may carry the scale of a two-shear rotation: the frame is diag(A, conj(A) / |A|^2))
        const cplx A2 = cmul(A, A);
        const double i4 = TWO ? 1.0 / (A2.x * A2.x + A2.y * A2.y) : 1.0;
        c01 = cmul(c01, cplx{ A2.x * i4, -A2.y * i4 });
        c10 = cmul(c10, A2);
        double sk, ck;
        sincos(0.5 * coef * a.spec.shift, &sk, &ck);
        RunRec rec;
        rec.m[0] = c00.x; rec.m[1] = c00.y;
        rec.fb[0] = c01.x; rec.fb[1] = c01.y;
        rec.fa[0] = c10.x; rec.fa[1] = c10.y;
        rec.c11[0] = c11.x; rec.c11[1] = c11.y;
        rec.rho[0] = 1.0; rec.rho[1] = 0.0;
        rec.cs[0] = ck; rec.cs[1] = sk;
        rec.ids[0] = rec.ids[1] = rec.ids[2] = -1;
        rec.ids[3] = pp.params[i].late_mask;
        rec.out[0] = 1;
        rec.out[1] = 1 + 2 * i;
        rec.out[2] = pp.params[i].late_diag;
        rec.out[3] = pp.params[i].late_pc;
        *out = rec;
        return;
    }
    const int ref_count = pp.params[i].ref_count;
    const int kind = ref_count > 0 ? pp.refs[pp.params[i].ref_begin].kind : -1;
    const int index = ref_count > 0 ? pp.refs[pp.params[i].ref_begin].index : 0;
    const double coef_sum = ref_count > 0 ? pp.refs[pp.params[i].ref_begin].coef_sum : 0.0;
    const double shift = variant == 2 ? -a.spec.shift : a.spec.shift;
    double2 m = make_double2(0, 0);
    cplx fb = { 1, 0 }, fa = { 1, 0 }, rho = { 1, 0 };
    int id_m = -1, id_b = -1, id_a = -1;
    if (kind == 0) {
        // the fused 2x2 at theta, then with the generator inserted (variant 0) or the angle shifted
        double2 mv[2];
        cplx A[2], G[2], gp[2];
#pragma unroll 1
        for (int v = 0; v < 2; ++v)
            compute_uentry<SCALED, TWO>(pp, index, a.theta, row, (v == 1 && variant != 0) ? i : -1, shift,
                                      (v == 1 && variant == 0) ? i : -1, mv[v], A[v], G[v], gp[v]);
        m = mv[1];
        fb = cmul(cconj(G[0]), G[1]);
        fa = cmul(cconj(A[0]), A[1]);   // A[1] / A[0] (the moduli differ when a two-shear form carries its scale)
        if (TWO) {
            const double i0 = 1.0 / (A[0].x * A[0].x + A[0].y * A[0].y);
            fa = { fa.x * i0, fa.y * i0 };
        }
        rho = cmul(gp[1], cconj(gp[0]));
        id_m = index;
        if (pp.uent[index].d_before >= 0) id_b = NU + pp.uent[index].d_before;
        if (pp.uent[index].d_after >= 0) id_a = NU + pp.uent[index].d_after;
    } else if (kind > 0) {
        double sn, c;
        sincos(0.5 * coef_sum * shift, &sn, &c);
        fb = { c, sn };
        id_b = kind == 1 ? NU + index : NU + N1 + index;
    }
    double sk, ck;
    sincos(0.5 * coef_sum * a.spec.shift, &sk, &ck);
    RunRec rec;
    rec.m[0] = m.x; rec.m[1] = m.y;
    rec.fb[0] = fb.x; rec.fb[1] = fb.y;
    rec.fa[0] = fa.x; rec.fa[1] = fa.y;
    rec.rho[0] = rho.x; rec.rho[1] = rho.y;
    rec.cs[0] = ck; rec.cs[1] = sk;
    rec.ids[0] = id_m; rec.ids[1] = id_b; rec.ids[2] = id_a; rec.ids[3] = 0;
    rec.out[0] = variant == 0 ? 1 : 0;
    rec.out[1] = 1 + 2 * i + (variant == 2 ? 1 : 0);
    rec.out[2] = pp.params[i].resume_diag;
    rec.out[3] = pp.params[i].resume_pc;
    rec.c11[0] = rec.c11[1] = 0.0;
    *out = rec;
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK, BLOCK == 1024 ? 4 : BLOCK == 512 ? 2 : BLOCK <= 128 ? VQA_SMALL_WAVES : VQA_WAVES) void k_reg_psr(RegArgs a) {
    extern __shared__ double2 smem[];
    const Prog pp = load_prog(a.prog);
    // one state per workgroup: the thread bits are the workgroup's size (a constant lets the loops over
    // them unroll, so that their table reads are issued together)
    constexpr int T = BLOCK == 64 ? 6 : BLOCK == 128 ? 7 : BLOCK == 256 ? 8 : BLOCK == 512 ? 9 : 10;
    // two-shear register rotations (the 1024-thread workgroups keep the masked diagonal stages, which take unit slots)
    // (measured at 12 qubits, depth 10: a fifth of the kernel's FP64 multiply-adds less, and 2 % MORE time -- the test of
    // the form in front of every rotation, a readfirstlane and a scalar branch, costs each wave more of its own
    // serial time than the multiply-adds it saves; off unless built with -DVQA_TWO_SHEAR)
#ifdef VQA_TWO_SHEAR
    constexpr bool TWO = BLOCK < 1024;
#else
    constexpr bool TWO = false;
#endif
    const int n = pp.H->n;
    const int NU = pp.H->n_uent, N1 = pp.H->n_d1, N2 = pp.H->n_d2, ND = pp.H->n_ddesc;
    const int P = pp.H->n_params, NR = pp.H->n_runs;
    const int tid = threadIdx.x;
    const int t = tid;

    double2* tabs = smem;
    double2* xbuf = tabs + (size_t)(NU + N1 + N2);
    double* red = (double*)(xbuf + (pp.H->has_xpose ? ((size_t)1 << (n - 1)) : 0));
    double2* pz = (double2*)(red + 128);                   // the current run's record (8 entries)
    double2* sav = pz + 8;                                // what the current patch replaced (3 entries, thread 0's)
    double* basev = (double*)(pz + 16);                   // the unshifted values (79 entries), squared norm of psi
    unsigned short* smask = (unsigned short*)(basev + 80);
    constexpr int TL = T >> 1, FE = (1 << TL) + (1 << (T - TL));   // entries of a stage's two thread-factor tables
    double2* ftab = (double2*)((char*)smask + (((size_t)pp.H->n_sign * BLOCK * 2 + 15) & ~(size_t)15));   // ND * (FE + kRegBits)

    Ctx cx;
    cx.mat = tabs; cx.ph1 = tabs + NU; cx.ph2 = tabs + NU + N1;
    cx.xbuf = (char*)xbuf;
    cx.smask = smask;
    cx.ftab = ftab;
    cx.T = T; cx.nthr = BLOCK;
    cx.wave = __builtin_amdgcn_readfirstlane(tid >> 6); cx.tmask = (1 << T) - 1;
    cx.xc_id = -1; cx.xc_wt = 0; cx.xc_rt = 0;

#ifdef VQA_PROF
    cx.flag_mask = ~((a.ablate & 1 ? PH_XPOSE : 0) | (a.ablate & 2 ? PH_UREG0 * 15 : 0) |
                     (a.ablate & 4 ? PH_ULANE0 * 63 : 0) | (a.ablate & 8 ? PH_DIAG : 0));
#endif
    thread_constants<BLOCK>(pp, smask, T, t, tid);

    // the workgroup's area: ND checkpoints and the unshifted final state, 16 * 2^n bytes each
    cx.image_bytes = 16u << n;
    cx.area = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ckpt + (size_t)blockIdx.x * ((size_t)(ND + 1) << n)), 0,
                                                (ND + 1) * cx.image_bytes, 0x00020000);
    const double2* recs = (const double2*)(a.recs + (size_t)blockIdx.x * NR);
    const int K = 2 * P + 1;
    unsigned parity = 0;

    __shared__ unsigned s_next;
#ifdef VQA_PROF
    if (tid < 16) s_stamp[tid] = 0;
#endif
    for (;;) {
        __syncthreads();
        if (tid == 0) {
            const VQA_CAS RegArgs* la = late_args();
            s_next = atomicAdd(la->next, 1u) - la->next_base;
        }
        __syncthreads();
        // a work item is a sample, or (few samples) one of the a.split slices of a sample's run list
        const int64_t item = s_next;
        const int64_t j = item / a.split;
        const int slice = (int)(item - j * a.split);
#ifdef VQA_PROF
        if (tid < 16) {
            if (s_stamp[tid]) atomicAdd(&g_stamp[tid], s_stamp[tid]);
            s_stamp[tid] = 0;
        }
#endif
        if (j >= a.E) break;
        int b = a.spec.sel ? a.spec.sel[j] : (int)j;
        const bool valid = a.spec.B == 0 || (unsigned)b < (unsigned)a.spec.B;
        if (!valid) b = 0;
        const double* row = a.input + (size_t)b * a.n_input;
        VQA_STAMP0
        __syncthreads();
        prep_tables<true, true, TWO>(a, pp, tabs, j, 1, NU, N1, N2, tid, BLOCK);
        for (int r = tid; r < NR; r += BLOCK)
            make_run_record<BLOCK, true, 1, TWO>(a, pp, r, row, NU, N1, a.recs + (size_t)blockIdx.x * NR + r);
        __threadfence_block();
        // the thread-only factors of the diagonal stages, by halves of the thread bits (unpatched tables)
        __syncthreads();
        for (int i = tid; i < ((BLOCK == 1024 || !a.use_ftab) ? 0 : ND * (FE + kRegBits)); i += BLOCK) {
            const int d = i / (FE + kRegBits), e = i - d * (FE + kRegBits);
            const double2* p1 = cx.ph1 + pp.ddesc[d].ph1_base;
            cplx f = { 1.0, 0.0 };
            // (a slot P stands for the factor P on |0> and conj(P) / |P|^2 on |1>: two-shear rotations leave their scales there)
            if (e >= FE) {
                // the step of register bit e - FE
                const double2 p = p1[e - FE];
                const double inv = TWO ? 1.0 / (p.x * p.x + p.y * p.y) : 1.0;
                const cplx cj = { p.x * inv, -p.y * inv };
                f = cmul(cj, cj);
            } else {
                const bool high = e >= (1 << TL);
                const int bits = high ? e - (1 << TL) : e, b0 = high ? TL : 0, nb = high ? T - TL : TL;
                for (int k = 0; k < nb; ++k) {
                    const double2 p = p1[kRegBits + b0 + k];
                    const unsigned bit = (unsigned)(bits >> k) & 1u;
                    cplx g = csel_conj({ p.x, p.y }, bit);
                    if (TWO && bit) {
                        const double inv = 1.0 / (p.x * p.x + p.y * p.y);
                        g = { g.x * inv, g.y * inv };
                    }
                    f = cmul(f, g);
                }
                if (!high) {
                    // the register bits' own phases (all bits clear) ride in the low table
                    for (int j = 0; j < kRegBits; ++j) {
                        const double2 p = p1[j];
                        f = cmul(f, { p.x, p.y });
                    }
                }
            }
            ftab[i] = make_double2(f.x, f.y);
        }
        VQA_STAMP(0)

        // run -1 is the unshifted circuit; thread 0 keeps what the current patch replaced
        int old_m = -1, old_b = -1, old_a = -1;
        double2 nx = make_double2(0, 0);
        const int lo = (int)((int64_t)a.n_live * slice / a.split), NL = (int)((int64_t)a.n_live * (slice + 1) / a.split) - lo;
        const int32_t* live = a.live + lo;
        for (int ri = -1; ri <= NL; ++ri) {
            const int r = ri < 0 ? -1 : ri < NL ? live[ri] : NR;
            VQA_STAMP0
            __syncthreads();   // the previous run is over (tables free); its successor's record is in pz
            int id_m = -1, id_b = -1, id_a = -1, mode = 0, kout = 0, rdiag = -1, rpc = 0, late = 0;
            if (r >= 0 && r < NR) {
                const int4 ids = *(const int4*)&pz[5], o = *(const int4*)&pz[6];
                id_m = __builtin_amdgcn_readfirstlane(ids.x);
                id_b = __builtin_amdgcn_readfirstlane(ids.y);
                id_a = __builtin_amdgcn_readfirstlane(ids.z);
                late = __builtin_amdgcn_readfirstlane(ids.w);
                mode = __builtin_amdgcn_readfirstlane(o.x);
                kout = __builtin_amdgcn_readfirstlane(o.y);
                rdiag = __builtin_amdgcn_readfirstlane(o.z);
                rpc = __builtin_amdgcn_readfirstlane(o.w);
            }
            if (tid == 0 && !VQA_ABL(a, 7)) {
                if (old_a >= 0) tabs[old_a] = sav[2];
                if (old_b >= 0) tabs[old_b] = sav[1];
                if (old_m >= 0) tabs[old_m] = sav[0];
                if (id_m >= 0) { sav[0] = tabs[id_m]; tabs[id_m] = pz[0]; }
                if (id_b >= 0) {
                    const double2 o = tabs[id_b], f = pz[1];
                    sav[1] = o;
                    const cplx q = cmul({ o.x, o.y }, { f.x, f.y });
                    tabs[id_b] = make_double2(q.x, q.y);
                }
                if (id_a >= 0) {
                    const double2 o = tabs[id_a], f = pz[2];
                    sav[2] = o;
                    const cplx q = cmul({ o.x, o.y }, { f.x, f.y });
                    tabs[id_a] = make_double2(q.x, q.y);
                }
            }
            old_m = id_m; old_b = id_b; old_a = id_a;
            if (r == NR) break;   // (the last pass only undoes the last patch)
            // the checkpoint is on its way while thread 0's patch settles (the barrier orders LDS only)
            double2 amp[kRegAmps];
            int pc0 = 0;
            if (rdiag >= 0) {
                if ((VQA_EXP & 1) && r >= 0) {
#pragma unroll
                    for (int q = 0; q < kRegAmps; ++q) amp[q] = make_double2(0.015625 * (q + 1), 0.0078125);
                } else
                if (!VQA_ABL(a, 6)) load_state(amp, cx.area, (unsigned)rdiag * cx.image_bytes, BLOCK, tid);
                pc0 = rpc;
            } else {
#pragma unroll
                for (int q = 0; q < kRegAmps; ++q) amp[q] = make_double2(0.0, 0.0);
                if (t == 0) amp[0].x = 1.0;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // late resumption: the checkpoint is the unshifted state behind the parameter's U-stage; the
            // run's state there differs by a 2x2 on the one qubit (the record is not replaced before the end of the run)
            if (late && !(VQA_EXP & 16)) apply_correction(amp, pz, late, cx);
            VQA_STAMP(1)
            // the next run's record travels while this run computes
            if (!(VQA_EXP & 32) && ri + 1 < NL && (tid & 63) < 8) nx = recs[(size_t)live[ri + 1] * 8 + (tid & 63)];
            // (a run that patches a phase slot works the thread-only factors out bit by bit)
            if (!VQA_ABL(a, 8) && !((VQA_EXP & 4) && r >= 0))
                run_phases<true, BLOCK == 1024, NoHook, TWO>(amp, pp, cx, pc0, rdiag >= 0, r < 0, a.use_ftab && id_b < 0 && id_a < 0);
            VQA_STAMP(2)
            if (r < 0) store_state(amp, cx.area, (unsigned)ND * cx.image_bytes, BLOCK, tid);
            if (!VQA_ABL(a, 4) && !((VQA_EXP & 8) && r >= 0)) {
                if (mode == 1) {
                    measure_pair(amp, (unsigned)ND * cx.image_bytes, basev[79], pz, basev, a, pp, cx, red, parity,
                                 j * K + kout, valid);
                } else {
                    const double nrm = measure(amp, a, pp, cx, red, parity, j * K + kout, true, valid,
                                               r < 0 ? basev : nullptr);
                    if (r < 0 && tid == 0) basev[79] = nrm;
                }
            }
            VQA_STAMP(3)
            // hand the fetched record over (read by everybody after the barrier on top)
            if (tid < 8 && ri + 1 < NL) pz[tid] = nx;
        }
        // the rows of the parameters outside the light cone (behind the runs: the loop above stays as it was)
        if (slice == 0) fill_dead_params(basev, j * K, valid, tid, BLOCK);
    }
}

// ---- full parameter-shift sweeps, two runs of a sample side by side ----------------------------
// The sweep kernel above keeps three workgroups on a CU (three waves on a SIMD), and a wave spends two thirds
// of its time waiting -- for its workgroup's barriers, LDS round trips, descriptor fetches and state images --
// not issuing arithmetic.  Here a workgroup runs TWO runs of its sample in lock-step, sixteen amplitudes of each
// per thread (two workgroups on a CU, 256 registers per thread): every wait, barrier, scalar instruction,
// descriptor and table read serves both, the diagonal stage's table is built once, and runs that resume from
// the same checkpoint load it once and share the loads of the unshifted final state in the measurement.
// Items of the work list (a.live: two numbers per item) are pairs of late-resumption runs of one diagonal
// stage, or a single run with -1 beside it (patched tables, the runs left over at a stage's end); a single
// run takes a copy of itself along as the second state, which writes nothing.

// sums of NV values over the workgroup, every thread gets them (the layout of state_sum's buffer with up to
// eight waves: the two kinds of calls may alternate on the same two halves of red)
template <int NV>
__device__ __forceinline__ void state_sum2(double (&v)[NV], const Ctx& cx, double* red, unsigned& parity) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
    if (cx.T > 6) {
        const int wave = cx.wave, nwave = cx.nthr >> 6;
        double* buf = red + (parity & 1u) * 64;
        parity = __builtin_amdgcn_readfirstlane(parity ^ 1u);
        if (cx.lane() == 0) {
#pragma unroll
            for (int k = 0; k < NV; ++k) buf[k * 8 + wave] = v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            v[k] = 0.0;
            for (int w = 0; w < nwave; ++w) v[k] += buf[k * 8 + w];
        }
    }
}

// do_xpose for two states: state B goes through the second half-size buffer behind the first
__device__ __forceinline__ void do_xpose2(double2 (&ampA)[kRegAmps], double2 (&ampB)[kRegAmps], const Prog& pp, Ctx& cx, int xid,
                                          int half_bytes) {
    const VQA_CAS XDesc* xd = pp.xdesc + xid;
    if (xid != cx.xc_id) {
        int wt = 0, rt = 0;
        const int t = cx.t();
        for (int i = 0; i < cx.T; ++i) {
            const int m = -((t >> i) & 1);
            wt ^= m & xd->wconst[kRegBits + i];
            rt ^= m & xd->rconst[kRegBits + i];
        }
        cx.xc_id = xid;
        cx.xc_wt = wt;
        cx.xc_rt = rt;
    }
    const int wt = cx.xc_wt, rt = cx.xc_rt;
    const int w0 = xd->wconst[0], w1 = xd->wconst[1], w2 = xd->wconst[2];
    const int r0 = xd->rconst[0], r1 = xd->rconst[1], r2 = xd->rconst[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int r = 0; r < kRegAmps / 2; ++r) {
            const int off = wt ^ ((r & 1) ? w0 : 0) ^ ((r & 2) ? w1 : 0) ^ ((r & 4) ? w2 : 0);
            *(double2*)(cx.xbuf + off) = ampA[r + 8 * half];
            *(double2*)(cx.xbuf + half_bytes + off) = ampB[r + 8 * half];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kRegAmps / 2; ++r) {
            const int off = rt ^ ((r & 1) ? r0 : 0) ^ ((r & 2) ? r1 : 0) ^ ((r & 4) ? r2 : 0);
            ampA[r + 8 * half] = *(const double2*)(cx.xbuf + off);
            ampB[r + 8 * half] = *(const double2*)(cx.xbuf + half_bytes + off);
        }
        __syncthreads();
    }
}

// do_diag for two states that share the stage's tables: the thread's phase table is built once
__device__ __forceinline__ void do_diag2(double2 (&ampA)[kRegAmps], double2 (&ampB)[kRegAmps], const Prog& pp, const Ctx& cx,
                                         int did, bool use_ftab) {
    const VQA_CAS i32x4* ddv = (const VQA_CAS i32x4*)(pp.ddesc + did);
    const i32x4 dd0 = ddv[0], dd1 = ddv[1];
    struct { int ph1_base, tt_begin, tt_count, rt_begin, rt_count, rr_begin, rr_count, sign_id; } ddl =
        { dd0.x, dd0.y, dd0.z, dd0.w, dd1.x, dd1.y, dd1.z, dd1.w };
    const auto* dd = &ddl;
    const double2* p1 = cx.ph1 + dd->ph1_base;
    const int t = cx.t();
    const int sign_id = dd->sign_id;
    unsigned sm = 0, rr = 0;
    if (sign_id >= 0) {
        sm = cx.smask8[sign_id * cx.nthr + cx.tid()];
        rr = pp.signs[sign_id].rr_par;
    }
    const int tt_begin = dd->tt_begin, tt_count = dd->tt_count;
    const int rt_begin = dd->rt_begin, rt_count = dd->rt_count;
    cplx tb[kRegAmps / 2], w3;
    auto step_of = [&](cplx w, int j) {
        const unsigned cr = (sm >> (1 + j)) & 1u;
        return cplx{ flip_if(w.x, cr), flip_if(w.y, cr) };
    };
    auto thread_pairs = [&](cplx T0) {
        T0 = { flip_if(T0.x, sm & 1u), flip_if(T0.y, sm & 1u) };
        for (int k = 0; k < tt_count; ++k) {
            const int pa = pp.pairs[tt_begin + k].a, pb = pp.pairs[tt_begin + k].b;
            const double2 z = cx.ph2[pp.pairs[tt_begin + k].ph2];
            T0 = cmul(T0, csel_conj({ z.x, z.y }, (unsigned)((t >> pa) ^ (t >> pb)) & 1u));
        }
        return T0;
    };
    if (use_ftab && rt_count == 0) {
        const int TL = cx.T >> 1, FE = (1 << TL) + (1 << (cx.T - TL));
        const double2* ft = cx.ftab + (size_t)did * (FE + kRegBits);
        const double2 fa = ft[t & ((1 << TL) - 1)], fb = ft[(1 << TL) + (t >> TL)];
        tb[0] = thread_pairs(cmul({ fa.x, fa.y }, { fb.x, fb.y }));
#pragma unroll
        for (int j = 0; j < kRegBits - 1; ++j) {
            const double2 wj = ft[FE + j];
            const cplx w = step_of({ wj.x, wj.y }, j);
#pragma unroll
            for (int r = 0; r < (1 << j); ++r) tb[r | (1 << j)] = cmul(tb[r], w);
        }
        const double2 wl = ft[FE + kRegBits - 1];
        w3 = step_of({ wl.x, wl.y }, kRegBits - 1);
    } else {
        cplx F = { 1.0, 0.0 };
        for (int i = 0; i < cx.T; ++i) {
            const double2 p = p1[kRegBits + i];
            F = cmul(F, csel_conj({ p.x, p.y }, (unsigned)(t >> i) & 1u));
        }
        cplx pj[kRegBits];
#pragma unroll
        for (int j = 0; j < kRegBits; ++j) {
            const double2 p = p1[j];
            pj[j] = { p.x, p.y };
        }
        for (int k = 0; k < rt_count; ++k) {
            const int pa = pp.pairs[rt_begin + k].a, pb = pp.pairs[rt_begin + k].b;
            const double2 z = cx.ph2[pp.pairs[rt_begin + k].ph2];
            const cplx ze = csel_conj({ z.x, z.y }, (unsigned)(t >> pb) & 1u);
#pragma unroll
            for (int j = 0; j < kRegBits; ++j)
                if (pa == j) pj[j] = cmul(pj[j], ze);
        }
        tb[0] = thread_pairs(cmul(cmul(F, cmul(pj[0], pj[1])), cmul(pj[2], pj[3])));
#pragma unroll
        for (int j = 0; j < kRegBits - 1; ++j) {
            const cplx cj = cconj(pj[j]);
            const cplx w = step_of(cmul(cj, cj), j);
#pragma unroll
            for (int r = 0; r < (1 << j); ++r) tb[r | (1 << j)] = cmul(tb[r], w);
        }
        const cplx c3 = cconj(pj[kRegBits - 1]);
        w3 = step_of(cmul(c3, c3), kRegBits - 1);
    }
#pragma unroll
    for (int r = 0; r < kRegAmps / 2; ++r) {
        const cplx hi = cmul(tb[r], w3);
        cmul_inplace(ampA[r].x, ampA[r].y, tb[r].x, tb[r].y);
        cmul_inplace(ampB[r].x, ampB[r].y, tb[r].x, tb[r].y);
        cmul_inplace(ampA[r + 8].x, ampA[r + 8].y, hi.x, hi.y);
        cmul_inplace(ampB[r + 8].x, ampB[r + 8].y, hi.x, hi.y);
    }
    if (rr) {
#pragma unroll
        for (int r = 0; r < kRegAmps; ++r) {
            const double sg = ((rr >> r) & 1u) ? -1.0 : 1.0;
            asm("v_mul_f64 %0, %0, %2\n\tv_mul_f64 %1, %1, %2" : "+v"(ampA[r].x), "+v"(ampA[r].y) : "s"(sg));
            asm("v_mul_f64 %0, %0, %2\n\tv_mul_f64 %1, %1, %2" : "+v"(ampB[r].x), "+v"(ampB[r].y) : "s"(sg));
        }
    }
    const int rr_begin = dd->rr_begin, rr_count = dd->rr_count;
    for (int k = 0; k < rr_count; ++k) {
        const int pa = pp.pairs[rr_begin + k].a, pb = pp.pairs[rr_begin + k].b;
        const double2 z = cx.ph2[pp.pairs[rr_begin + k].ph2];
#pragma unroll
        for (int r = 0; r < kRegAmps; ++r) {
            const cplx ze = csel_conj({ z.x, z.y }, (unsigned)((r >> pa) ^ (r >> pb)) & 1u);
            cmul_inplace(ampA[r].x, ampA[r].y, ze.x, ze.y);
            cmul_inplace(ampB[r].x, ampB[r].y, ze.x, ze.y);
        }
    }
}

// run_phases for two states (the scaled lane forms; the table form of the diagonal stages); ckpt: state A's
// images in front of the diagonal stages are stored
__device__ __forceinline__ void run_phases2(double2 (&ampA)[kRegAmps], double2 (&ampB)[kRegAmps], const Prog& pp, Ctx& cx, int p0,
                                            bool diag_only_first, bool ckpt, bool use_ftab, int half_bytes) {
    const VQA_CAS PhaseOp* ph = pp.phases;
    int first_mask = diag_only_first ? (PH_DIAG | PH_END) : -1;
    for (int p = p0;; ++p) {
        const VQA_CAS i32x4* dsc = (const VQA_CAS i32x4*)(ph + p);
        const i32x4 d0 = dsc[0], d1 = dsc[1], d2 = dsc[2], d3 = dsc[3];
        const int ent[10] = { d0.w, d1.x, d1.y, d1.z, d1.w, d2.x, d2.y, d2.z, d2.w, d3.x };
        const int flags = d0.x & first_mask;
        first_mask = -1;
        if (flags & PH_END) break;
        if (flags & PH_XPOSE) do_xpose2(ampA, ampB, pp, cx, d0.y, half_bytes);
        double2 cs = cx.mat[ent[0]], csn;
#define VQA_REG_SLOT2(J)                                          \
        csn = cx.mat[ent[J + 1]];                                 \
        if (flags & (PH_UREG0 << J)) {                            \
            apply_ureg<J>(ampA, cs.x, cs.y);                      \
            apply_ureg<J>(ampB, cs.x, cs.y);                      \
        }                                                         \
        cs = csn;
        VQA_REG_SLOT2(0) VQA_REG_SLOT2(1) VQA_REG_SLOT2(2) VQA_REG_SLOT2(3)
#undef VQA_REG_SLOT2
#define VQA_LANE_SLOT2(LB)                                                 \
        csn = cx.mat[ent[LB + 5 < 10 ? LB + 5 : 9]];                       \
        {                                                                  \
            const int form_b = __builtin_amdgcn_readfirstlane(__double2hiint(cs.y)); \
            const int run = flags & (PH_ULANE0 << LB);                     \
            if (run && form_b == 0) {                                      \
                apply_ulane<LB, 1>(ampA, cs.x, cs.y, cx.lane());           \
                apply_ulane<LB, 1>(ampB, cs.x, cs.y, cx.lane());           \
            }                                                              \
            if (run && form_b != 0) {                                      \
                apply_ulane<LB, 2>(ampA, cs.x, cs.y, cx.lane());           \
                apply_ulane<LB, 2>(ampB, cs.x, cs.y, cx.lane());           \
            }                                                              \
        }                                                                  \
        cs = csn;
        VQA_LANE_SLOT2(0) VQA_LANE_SLOT2(1) VQA_LANE_SLOT2(2) VQA_LANE_SLOT2(3)
#undef VQA_LANE_SLOT2
        if (flags & (PH_ULANE0 * 0x30)) {
            const double2 c5 = cx.mat[ent[9]];
            const bool h4 = (flags & (PH_ULANE0 << 4)) != 0, h5 = (flags & (PH_ULANE0 << 5)) != 0;
            const double c4v = h4 ? cs.x : 1.0, s4v = h4 ? cs.y : 0.0, c5v = h5 ? c5.x : 1.0, s5v = h5 ? c5.y : 0.0;
            apply_ulane45(ampA, c4v, s4v, c5v, s5v, cx.lane());
            apply_ulane45(ampB, c4v, s4v, c5v, s5v, cx.lane());
        }
        if (flags & PH_DIAG) {
            const int did = d0.z;
            if (ckpt) store_state(ampA, cx.area, (unsigned)did * cx.image_bytes, cx.nthr, cx.tid());
            do_diag2(ampA, ampB, pp, cx, did, use_ftab);
        }
    }
}

// measure_pair for two pair runs of one sample: the unshifted final state is loaded once
__device__ __forceinline__ void measure_pair2(double2 (&ampA)[kRegAmps], double2 (&ampB)[kRegAmps], unsigned psi_image,
                                              double psi_norm, const double2* recA, const double2* recB, const double* basev,
                                              const RegArgs& a, const Prog& pp, const Ctx& cx, double* red, unsigned& parity,
                                              int64_t e1A, int64_t e1B, bool valid) {
    const double nanv = __longlong_as_double(0x7ff8000000000000LL);
    unsigned kt, krb[kRegBits];
    const int t = cx.t();
    measure_masks(pp, cx.T, t, kt, krb);
    const int nq = a.obs.n_obs + (a.n_marg_bits > 0 ? (1 << a.n_marg_bits) : 0);
    const unsigned toff = cx.tid() * (unsigned)sizeof(double2), rstep = cx.nthr * (unsigned)sizeof(double2);
    double prA[kRegAmps], prB[kRegAmps];
    // amp <- conj(psi) phi, pr <- |phi|^2 for both states; the loads of psi run eight ahead
    u32x4 pv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) pv[r] = __builtin_amdgcn_raw_buffer_load_b128(cx.area, toff, psi_image + r * rstep, VQA_IMAGE_AUX);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < kRegAmps; ++r) {
        const u32x4 cur = pv[r & 7];
        const double2 ps = *(const double2*)&cur;
        const double2 pa = ampA[r], pb = ampB[r];
        prA[r] = pa.x * pa.x + pa.y * pa.y;
        prB[r] = pb.x * pb.x + pb.y * pb.y;
        ampA[r] = make_double2(ps.x * pa.x + ps.y * pa.y, ps.x * pa.y - ps.y * pa.x);
        ampB[r] = make_double2(ps.x * pb.x + ps.y * pb.y, ps.x * pb.y - ps.y * pb.x);
        if (r + 8 < kRegAmps)
            pv[r & 7] = __builtin_amdgcn_raw_buffer_load_b128(cx.area, toff, psi_image + (r + 8) * rstep, VQA_IMAGE_AUX);
        __builtin_amdgcn_sched_barrier(0);
    }
    double inA = 1.0, icA = 1.0, inB = 1.0, icB = 1.0;
    for (int q = 0; q < nq; ++q) {
        // per state: sum o |phi|^2, Re / Im sum o conj(psi) phi; with the first quantity the norms
        double v[8] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
        if (q == 0) {
#pragma unroll
            for (int r = 0; r < kRegAmps; ++r) { v[6] += prA[r]; v[7] += prB[r]; }
        }
        if (q < a.obs.n_obs) {
            for (int tm = a.obs.off[q]; tm < a.obs.off[q + 1]; ++tm) {
                const unsigned z = (unsigned)a.obs.z[tm];
                const unsigned tpar = __popc(kt & z);
                double p[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                for (int r = 0; r < kRegAmps; ++r) {
                    unsigned rp = tpar;
#pragma unroll
                    for (int j = 0; j < kRegBits; ++j)
                        if (r & (1 << j)) rp += __popc(krb[j] & z);
                    const double sg = __hiloint2double((int)(0x3FF00000u | ((rp & 1u) << 31)), 0);
                    p[0] = fma(sg, prA[r], p[0]);
                    p[1] = fma(sg, ampA[r].x, p[1]);
                    p[2] = fma(sg, ampA[r].y, p[2]);
                    p[3] = fma(sg, prB[r], p[3]);
                    p[4] = fma(sg, ampB[r].x, p[4]);
                    p[5] = fma(sg, ampB[r].y, p[5]);
                }
                const double cf = a.obs.coef[tm];
#pragma unroll
                for (int k = 0; k < 6; ++k) v[k] += cf * p[k];
            }
        } else {
            const unsigned m = (unsigned)(q - a.obs.n_obs), M1 = (1u << a.n_marg_bits) - 1;
#pragma unroll
            for (int r = 0; r < kRegAmps; ++r) {
                unsigned k = kt;
#pragma unroll
                for (int j = 0; j < kRegBits; ++j)
                    if (r & (1 << j)) k |= krb[j];
                const double w = (k & M1) == m ? 1.0 : 0.0;
                v[0] = fma(w, prA[r], v[0]);
                v[1] = fma(w, ampA[r].x, v[1]);
                v[2] = fma(w, ampA[r].y, v[2]);
                v[3] = fma(w, prB[r], v[3]);
                v[4] = fma(w, ampB[r].x, v[4]);
                v[5] = fma(w, ampB[r].y, v[5]);
            }
        }
        if (q == 0) {
            state_sum2<8>(v, cx, red, parity);
            inA = 1.0 / v[6]; icA = rsqrt(v[6] * psi_norm);
            inB = 1.0 / v[7]; icB = rsqrt(v[7] * psi_norm);
        } else {
            double w[6] = { v[0], v[1], v[2], v[3], v[4], v[5] };
            state_sum2<6>(w, cx, red, parity);
#pragma unroll
            for (int k = 0; k < 6; ++k) v[k] = w[k];
        }
        // threads 0 and 1 write one run's two values each
        if (t < 2) {
            const bool is_obs = q < a.obs.n_obs;
            const int64_t e1 = t == 0 ? e1A : e1B;
            const size_t o1 = is_obs ? (size_t)e1 * a.obs.n_obs + q : ((size_t)e1 << a.n_marg_bits) + (q - a.obs.n_obs);
            const size_t step = is_obs ? (size_t)a.obs.n_obs : ((size_t)1 << a.n_marg_bits);
            double* out = is_obs ? a.out_expect : a.out_marginal;
            const double2* rec = t == 0 ? recA : recB;
            const double2 rho = rec[3], cs = rec[4];
            const double s0 = t == 0 ? v[0] : v[3], s1 = t == 0 ? v[1] : v[4], s2 = t == 0 ? v[2] : v[5];
            const double inv_norm = t == 0 ? inA : inB, inv_cross = t == 0 ? icA : icB;
            const double im = (rho.x * s2 + rho.y * s1) * inv_cross;
            const double mid = cs.x * cs.x * basev[q] + cs.y * cs.y * s0 * inv_norm;
            const double dev = 2.0 * cs.x * cs.y * im;
            out[o1] = valid ? mid - dev : nanv;
            out[o1 + step] = valid ? mid + dev : nanv;
        }
    }
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK, VQA_PSR2_WAVES) void k_reg_psr2(RegArgs a) {
    extern __shared__ double2 smem[];
    const Prog pp = load_prog(a.prog);
    constexpr int T = BLOCK == 64 ? 6 : BLOCK == 128 ? 7 : BLOCK == 256 ? 8 : 9;
    const int n = pp.H->n;
    const int NU = pp.H->n_uent, N1 = pp.H->n_d1, N2 = pp.H->n_d2, ND = pp.H->n_ddesc;
    const int P = pp.H->n_params, NR = pp.H->n_runs;
    const int tid = threadIdx.x;
    const int t = tid;
    const int half_bytes = pp.H->has_xpose ? (8 << n) : 0;   // one half-size transposition buffer per state

    // (the transposition buffers first: their addresses are then constants of the kernel)
    double2* xbuf = smem;
    double2* tabs = (double2*)((char*)smem + 2 * (size_t)half_bytes);
    double* red = (double*)(tabs + (size_t)(NU + N1 + N2));
    double2* pz = (double2*)(red + 128);                  // the records of the current item's two runs (2 x 8 entries)
    double2* sav = pz + 16;                               // what the current patch replaced (3 entries, thread 0's)
    double* basev = (double*)(sav + 4);                   // the unshifted values (79 entries), squared norm of psi
    unsigned char* smask8 = (unsigned char*)(basev + 80);
    constexpr int TL = T >> 1, FE = (1 << TL) + (1 << (T - TL));
    double2* ftab = (double2*)((char*)smask8 + (((size_t)pp.H->n_sign * BLOCK + 15) & ~(size_t)15));

    Ctx cx;
    cx.mat = tabs; cx.ph1 = tabs + NU; cx.ph2 = tabs + NU + N1;
    cx.xbuf = (char*)xbuf;
    cx.smask = nullptr;
    cx.smask8 = smask8;
    cx.ftab = ftab;
    cx.T = T; cx.nthr = BLOCK;
    cx.wave = __builtin_amdgcn_readfirstlane(tid >> 6); cx.tmask = (1 << T) - 1;
    cx.xc_id = -1; cx.xc_wt = 0; cx.xc_rt = 0;
#ifdef VQA_PROF
    cx.flag_mask = -1;
#endif
    // per-thread sign bits of every sign table: bit 0 the thread's own parity, bit 1 + j the flip that register bit j brings
    for (int sg = 0; sg < pp.H->n_sign; ++sg) {
        const int wi = t >> 5, bi = t & 31;
        unsigned m = (pp.signs[sg].tt_par[wi] >> bi) & 1u;
#pragma unroll
        for (int jj = 0; jj < kRegBits; ++jj) m |= ((pp.signs[sg].cross[jj][wi] >> bi) & 1u) << (1 + jj);
        smask8[sg * BLOCK + tid] = (unsigned char)m;
    }

    cx.image_bytes = 16u << n;
    cx.area = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ckpt + (size_t)blockIdx.x * ((size_t)(ND + 1) << n)), 0,
                                                (ND + 1) * cx.image_bytes, 0x00020000);
    const double2* recs = (const double2*)(a.recs + (size_t)blockIdx.x * NR);
    const int K = 2 * P + 1;
    unsigned parity = 0;

    __shared__ unsigned s_next2;
    for (;;) {
        __syncthreads();
        if (tid == 0) {
            const VQA_CAS RegArgs* la = late_args();
            s_next2 = atomicAdd(la->next, 1u) - la->next_base;
        }
        __syncthreads();
        const int64_t item = s_next2;
        const int64_t j = item / a.split;
        const int slice = (int)(item - j * a.split);
        if (j >= a.E) break;
        int b = a.spec.sel ? a.spec.sel[j] : (int)j;
        const bool valid = a.spec.B == 0 || (unsigned)b < (unsigned)a.spec.B;
        if (!valid) b = 0;
        const double* row = a.input + (size_t)b * a.n_input;
        __syncthreads();
        prep_tables<true, true>(a, pp, tabs, j, 1, NU, N1, N2, tid, BLOCK);
        for (int r = tid; r < NR; r += BLOCK)
            make_run_record<BLOCK, true, 1>(a, pp, r, row, NU, N1, a.recs + (size_t)blockIdx.x * NR + r);
        __threadfence_block();
        __syncthreads();
        for (int i = tid; i < ND * (FE + kRegBits); i += BLOCK) {
            const int d = i / (FE + kRegBits), e = i - d * (FE + kRegBits);
            const double2* p1 = cx.ph1 + pp.ddesc[d].ph1_base;
            cplx f = { 1.0, 0.0 };
            if (e >= FE) {
                const double2 p = p1[e - FE];
                const cplx cj = { p.x, -p.y };
                f = cmul(cj, cj);
            } else {
                const bool high = e >= (1 << TL);
                const int bits = high ? e - (1 << TL) : e, b0 = high ? TL : 0, nb = high ? T - TL : TL;
                for (int k = 0; k < nb; ++k) {
                    const double2 p = p1[kRegBits + b0 + k];
                    f = cmul(f, csel_conj({ p.x, p.y }, (unsigned)(bits >> k) & 1u));
                }
                if (!high) {
                    for (int jj = 0; jj < kRegBits; ++jj) {
                        const double2 p = p1[jj];
                        f = cmul(f, { p.x, p.y });
                    }
                }
            }
            ftab[i] = make_double2(f.x, f.y);
        }

        // item -1 is the unshifted circuit; thread 0 keeps what the current patch replaced
        int old_m = -1, old_b = -1, old_a = -1;
        double2 nx = make_double2(0, 0);
        const int lo = (int)((int64_t)a.n_live * slice / a.split), NI = (int)((int64_t)a.n_live * (slice + 1) / a.split) - lo;
        const int32_t* live = a.live + 2 * (size_t)lo;
        for (int it = -1; it <= NI; ++it) {
            const int rA = it < 0 ? -1 : it < NI ? live[2 * it] : NR;
            const int rB = (it >= 0 && it < NI) ? live[2 * it + 1] : -1;
            __syncthreads();   // the previous item is over (tables free); its successor's records are in pz
            int id_m = -1, id_b = -1, id_a = -1, mode = 0, kout = 0, rdiag = -1, rpc = 0, late = 0, koutB = 0, lateB = 0;
            if (rA >= 0 && rA < NR) {
                const int4 ids = *(const int4*)&pz[5], o = *(const int4*)&pz[6];
                id_m = __builtin_amdgcn_readfirstlane(ids.x);
                id_b = __builtin_amdgcn_readfirstlane(ids.y);
                id_a = __builtin_amdgcn_readfirstlane(ids.z);
                late = __builtin_amdgcn_readfirstlane(ids.w);
                mode = __builtin_amdgcn_readfirstlane(o.x);
                kout = __builtin_amdgcn_readfirstlane(o.y);
                rdiag = __builtin_amdgcn_readfirstlane(o.z);
                rpc = __builtin_amdgcn_readfirstlane(o.w);
            }
            if (rB >= 0) {
                const int4 ids = *(const int4*)&pz[8 + 5], o = *(const int4*)&pz[8 + 6];
                lateB = __builtin_amdgcn_readfirstlane(ids.w);
                koutB = __builtin_amdgcn_readfirstlane(o.y);
            }
            if (tid == 0) {
                if (old_a >= 0) tabs[old_a] = sav[2];
                if (old_b >= 0) tabs[old_b] = sav[1];
                if (old_m >= 0) tabs[old_m] = sav[0];
                if (id_m >= 0) { sav[0] = tabs[id_m]; tabs[id_m] = pz[0]; }
                if (id_b >= 0) {
                    const double2 o = tabs[id_b], f = pz[1];
                    sav[1] = o;
                    const cplx q = cmul({ o.x, o.y }, { f.x, f.y });
                    tabs[id_b] = make_double2(q.x, q.y);
                }
                if (id_a >= 0) {
                    const double2 o = tabs[id_a], f = pz[2];
                    sav[2] = o;
                    const cplx q = cmul({ o.x, o.y }, { f.x, f.y });
                    tabs[id_a] = make_double2(q.x, q.y);
                }
            }
            old_m = id_m; old_b = id_b; old_a = id_a;
            if (rA == NR) break;
            double2 ampA[kRegAmps], ampB[kRegAmps];
            int pc0 = 0;
            if (rdiag >= 0) {
                load_state(ampA, cx.area, (unsigned)rdiag * cx.image_bytes, BLOCK, tid);
                pc0 = rpc;
            } else {
#pragma unroll
                for (int q = 0; q < kRegAmps; ++q) ampA[q] = make_double2(0.0, 0.0);
                if (t == 0) ampA[0].x = 1.0;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // the second state starts from the same image (a single run: as a copy that writes nothing)
#pragma unroll
            for (int q = 0; q < kRegAmps; ++q) ampB[q] = ampA[q];
            if (late) apply_correction(ampA, pz, late, cx);
            if (rB >= 0) apply_correction(ampB, pz + 8, lateB, cx);
            // the next item's records travel while this one computes
            if (it + 1 < NI && (tid & 63) < 16) {
                const int rn = live[2 * (it + 1) + ((tid & 63) >> 3)];
                if (rn >= 0) nx = recs[(size_t)rn * 8 + (tid & 7)];
            }
            run_phases2(ampA, ampB, pp, cx, pc0, rdiag >= 0, rA < 0, id_b < 0 && id_a < 0, half_bytes);
            if (rA < 0) store_state(ampA, cx.area, (unsigned)ND * cx.image_bytes, BLOCK, tid);
            if (rB >= 0) {
                measure_pair2(ampA, ampB, (unsigned)ND * cx.image_bytes, basev[79], pz, pz + 8, basev, a, pp, cx, red, parity,
                              j * K + kout, j * K + koutB, valid);
            } else if (mode == 1) {
                measure_pair(ampA, (unsigned)ND * cx.image_bytes, basev[79], pz, basev, a, pp, cx, red, parity, j * K + kout, valid);
            } else {
                const double nrm = measure(ampA, a, pp, cx, red, parity, j * K + kout, true, valid, rA < 0 ? basev : nullptr);
                if (rA < 0 && tid == 0) basev[79] = nrm;
            }
            if (tid < 16 && it + 1 < NI) pz[tid] = nx;
        }
        // the rows of the parameters outside the light cone (behind the runs: the loop above stays as it was)
        if (slice == 0) fill_dead_params(basev, j * K, valid, tid, BLOCK);
    }
}

// ---- full parameter-shift sweeps of small circuits (several states per workgroup) ------
// Below 10 qubits a workgroup (one wave) holds S = 64 / 2^(n-4) states.  As an explicit list
// every instance would prepare all its fused 2x2s from scratch, which costs as much as the
// gates of so small a state.  Here the S states of a workgroup are S RUNS of ONE sample (the
// run list of the sweep kernel above: one run per parameter where both shifted values come out
// of one propagated state, else one per shift).  Per (workgroup, sample): the unshifted tables
// are prepared once; the unshifted circuit is run (by every state alike), its final state psi
// goes to LDS and its values to basev.  Per chunk of S runs: the tables are copied to every
// state, the first thread of each state patches the entries of its run -- the sweep kernel's
// run record for a pair run; for single shifts every entry the parameter enters: the fused 2x2
// recomputed at the shifted angle, the phase slots in front of and behind it multiplied by the
// ratio of new and old RZ factors, or the parameter's own phase slot turned by half the shift --
// and every state runs the circuit and measures (pairs against psi in LDS).
// a.E = number of samples; a.live[0 .. n_live) = what to run per sample: r >= 0 = run r of the
// run list, -(k + 1) = shift index k on its own; output instance j * (2P+1) + k.  With pair runs
// in the list (need_psi) row 0 of a sample is written by the unshifted run of every workgroup
// that works on the sample (same value); without, the list holds -(0 + 1) and nothing is kept
// per sample but the tables (the cheaper form when a sample's shifts fill only a chunk or two).
// Work items (sample, chunk of S entries) are dealt to the workgroups in contiguous ranges.
template <bool SCALED>
__device__ __forceinline__ void prep_base(const RegArgs& a, const Prog& pp, double2* base, double2* bag,
                                          const double* row, int NU, int N1, int N2, int tid, int nthr) {
    for (int k = tid; k < N1 + N2; k += nthr) {
        const int begin = k < N1 ? pp.d1[k].begin : pp.d2[k - N1].begin;
        const int count = k < N1 ? pp.d1[k].count : pp.d2[k - N1].count;
        cplx ph = { 1.0, 0.0 };
        if (count > 0) {
            double ang = 0.0;
            for (int j = 0; j < count; ++j) ang += elem_angle_c(pp.terms + begin + j, a.theta, row, -1, 0.0);
            double sn, c;
            sincos(0.5 * ang, &sn, &c);
            ph = { c, sn };
        }
        base[NU + k] = make_double2(ph.x, ph.y);
    }
    // every entry once; its factors join the phase slots around it in two rounds, even U-stages first:
    // no two entries of one round update the same slot
    for (int u0 = 0; u0 < NU; u0 += nthr) {
        const int u = u0 + tid;
        const bool have = u < NU;
        cplx A = { 1, 0 }, G = { 1, 0 };
        if (have) {
            double2 cs;
            cplx gp;
            compute_uentry<SCALED>(pp, u, a.theta, row, -1, 0.0, -1, cs, A, G, gp);
            base[u] = cs;
            bag[2 * u] = make_double2(A.x, A.y);
            bag[2 * u + 1] = make_double2(G.x, G.y);
        }
        for (int parity = 0; parity < 2; ++parity) {
            __syncthreads();
            if (!have || pp.uent[u].parity != parity) continue;
            const int d_before = pp.uent[u].d_before, d_after = pp.uent[u].d_after;
            if (d_before >= 0) {
                const double2 p = base[NU + d_before];
                const cplx q = cmul({ p.x, p.y }, G);
                base[NU + d_before] = make_double2(q.x, q.y);
            }
            if (d_after >= 0) {
                const double2 p = base[NU + d_after];
                const cplx q = cmul({ p.x, p.y }, A);
                base[NU + d_after] = make_double2(q.x, q.y);
            }
        }
    }
}

__global__ __launch_bounds__(64, VQA_GROUP_WAVES) void k_reg_group(RegArgs a) {
    constexpr int BLOCK = 64;
    constexpr bool SCALED = false;   // several states per wave: plain (cos, sin) lane rotations
    extern __shared__ double2 smem[];
    const Prog pp = load_prog(a.prog);
    const int n = pp.H->n, T = pp.H->T, S = pp.H->S;
    const int NU = pp.H->n_uent, N1 = pp.H->n_d1, N2 = pp.H->n_d2;
    const int tab_stride = NU + N1 + N2;
    const int tid = threadIdx.x;
    const int sl = tid >> T;
    const int t = tid & ((1 << T) - 1);

    double2* tabs = smem;                                  // S * tab_stride
    double2* base = tabs + (size_t)S * tab_stride;         // unshifted tables of the current sample
    double2* bag = base + tab_stride;                      // (A, G) of every fused 2x2 of the unshifted circuit
    double2* xbuf = bag + 2 * (size_t)NU;
    double* red = (double*)(xbuf + (pp.H->has_xpose ? ((size_t)1 << (n - 1)) : 0));
    unsigned short* smask = (unsigned short*)(red + 128);
    // behind the sign masks: psi (2^n amplitudes), the unshifted values, one run record per state
    double2* psi = (double2*)((char*)smask + (((size_t)pp.H->n_sign * BLOCK * 2 + 15) & ~(size_t)15));
    double* basev = (double*)(psi + ((size_t)1 << n));     // 79 values, squared norm of psi
    RunRec* srec = (RunRec*)(basev + 80) + sl;

    Ctx cx;
    cx.mat = tabs + (size_t)sl * tab_stride;
    cx.ph1 = cx.mat + NU;
    cx.ph2 = cx.ph1 + N1;
    cx.xbuf = (char*)xbuf;
    cx.smask = smask;
    cx.ftab = nullptr;
    cx.T = T; cx.nthr = BLOCK;
    cx.wave = 0; cx.tmask = (1 << T) - 1;
    cx.xc_id = -1; cx.xc_wt = 0; cx.xc_rt = 0;
    cx.area = __builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0x00020000);
    cx.image_bytes = 0;
#ifdef VQA_PROF
    cx.flag_mask = -1;
#endif
    thread_constants<BLOCK>(pp, smask, T, t, tid);

    const int K = 2 * a.spec.P + 1;
    const int NL = a.n_live;
    const int chunks = max((NL + S - 1) / S, 1);   // (no live run: one chunk, so that row 0 is still written)
    const int64_t total = a.E * chunks;
    const int64_t w0 = total * blockIdx.x / gridDim.x, w1 = total * (blockIdx.x + 1) / gridDim.x;
    int64_t cur = -1, cur_psi = -1;
    unsigned parity = 0;
    for (int64_t w = w0; w < w1; ++w) {
        const int64_t j = w / chunks;
        const int ch = (int)(w - j * chunks);
        int b = a.spec.sel ? a.spec.sel[j] : (int)j;
        const bool valid = a.spec.B == 0 || (unsigned)b < (unsigned)a.spec.B;
        if (!valid) b = 0;
        const double* row = a.input + (size_t)b * a.n_input;
        __syncthreads();   // the previous item is done with the tables
        if (j != cur) {
            prep_base<SCALED>(a, pp, base, bag, row, NU, N1, N2, tid, BLOCK);
            cur = j;
        }
        // pass 0 (first item of a sample when pair runs are listed): the unshifted circuit, by every
        // state alike; state 0 keeps psi and the values.  pass 1: the chunk's entries.
        const bool do_base = a.need_psi && j != cur_psi;
        cur_psi = j;
        for (int pass = do_base ? 0 : 1; pass < 2; ++pass) {
            __syncthreads();
            for (int i = tid; i < tab_stride; i += BLOCK) {
                const double2 v = base[i];
                for (int s2 = 0; s2 < S; ++s2) tabs[s2 * tab_stride + i] = v;
            }
            __syncthreads();
            // list entry: r >= 0 = run r of the run list; -(k + 1) = shift index k on its own (0 = unshifted)
            const int li = ch * S + sl;
            const bool live = pass == 0 ? sl == 0 : li < NL;
            const int r = (pass == 1 && live) ? a.live[li] : -1;
            int i = 0, variant = 3, kout = 0;   // variant 3: nothing to patch
            if (r >= 0) {
                i = pp.runs[r].param;
                variant = pp.runs[r].variant;
                kout = 1 + 2 * i + (variant == 2 ? 1 : 0);
            } else if (r < -1) {
                kout = -r - 1;
                i = (kout - 1) >> 1;
                variant = (kout & 1) ? 1 : 2;
            }
            if (live && t == 0 && variant != 3) {
                double2* tb = tabs + (size_t)sl * tab_stride;
                if (variant == 0) {
                    // both shifts from one run: the sweep kernel's record (generator inserted in front of the rotation)
                    make_run_record<BLOCK, SCALED>(a, pp, r, row, NU, N1, srec);
                    if (srec->ids[0] >= 0) tb[srec->ids[0]] = make_double2(srec->m[0], srec->m[1]);
                    if (srec->ids[1] >= 0) {
                        const double2 p = tb[srec->ids[1]];
                        const cplx q = cmul({ p.x, p.y }, { srec->fb[0], srec->fb[1] });
                        tb[srec->ids[1]] = make_double2(q.x, q.y);
                    }
                    if (srec->ids[2] >= 0) {
                        const double2 p = tb[srec->ids[2]];
                        const cplx q = cmul({ p.x, p.y }, { srec->fa[0], srec->fa[1] });
                        tb[srec->ids[2]] = make_double2(q.x, q.y);
                    }
                } else {
                    const double sval = variant == 1 ? a.spec.shift : -a.spec.shift;
                    const int rb = pp.params[i].ref_begin, rcnt = pp.params[i].ref_count;
                    for (int q = 0; q < rcnt; ++q) {
                        const int kind = pp.refs[rb + q].kind, index = pp.refs[rb + q].index;
                        if (kind == 0) {
                            double2 m;
                            cplx A, G, gp;
                            compute_uentry<SCALED>(pp, index, a.theta, row, i, sval, -1, m, A, G, gp);
                            tb[index] = m;
                            const int d_before = pp.uent[index].d_before, d_after = pp.uent[index].d_after;
                            if (d_before >= 0) {
                                const double2 g0 = bag[2 * index + 1], p = tb[NU + d_before];
                                const cplx f = cmul(cconj({ g0.x, g0.y }), G), z = cmul({ p.x, p.y }, f);
                                tb[NU + d_before] = make_double2(z.x, z.y);
                            }
                            if (d_after >= 0) {
                                const double2 a0 = bag[2 * index], p = tb[NU + d_after];
                                const cplx f = cmul(cconj({ a0.x, a0.y }), A), z = cmul({ p.x, p.y }, f);
                                tb[NU + d_after] = make_double2(z.x, z.y);
                            }
                        } else {
                            double sn, c;
                            sincos(0.5 * pp.refs[rb + q].coef_sum * sval, &sn, &c);
                            const int slot = kind == 1 ? NU + index : NU + N1 + index;
                            const double2 p = tb[slot];
                            const cplx z = cmul({ p.x, p.y }, { c, sn });
                            tb[slot] = make_double2(z.x, z.y);
                        }
                    }
                }
            }
            __syncthreads();

            double2 amp[kRegAmps];
#pragma unroll
            for (int q = 0; q < kRegAmps; ++q) amp[q] = make_double2(0.0, 0.0);
            if (t == 0) amp[0].x = 1.0;
            run_phases<SCALED>(amp, pp, cx, 0, false, false);
            if (pass == 0 && sl == 0) {
#pragma unroll
                for (int q = 0; q < kRegAmps; ++q) psi[(q << T) + t] = amp[q];
            }
            if (live && variant == 0) {
                measure_pair(amp, 0u, basev[79], (const double2*)srec, basev, a, pp, cx, red, parity, j * K + kout, valid, psi);
            } else {
                const double nrm = measure(amp, a, pp, cx, red, parity, j * K + kout, live, valid,
                                           (pass == 0 && sl == 0) ? basev : nullptr);
                if (pass == 0 && tid == 0) basev[79] = nrm;
            }
        }
    }
}


// ---- grouped sweeps with late resumption -------------------------------------------------
// The same division of work as k_reg_group (S states of a one-wave workgroup are S runs of one sample),
// for run lists in which every parameter drives one rotation of one fused 2x2 (a.live: one entry per live
// parameter, ordered by the U-stage of that 2x2; the kernel's own list has the unshifted circuit in front as
// entry 0).  Per (workgroup, sample) the unshifted circuit runs once -- as state 0 of the sample's first chunk,
// whose other states run from |0...0> beside it, or on its own when the workgroup's share of the sample starts
// at a later chunk -- and leaves the state in front of every diagonal stage and the final state psi.  A later
// chunk of S runs starts from the checkpoint behind the U-stage of its first run, with ONE copy of the
// unshifted tables for all states; in front of diagonal stage d the states whose parameter sits in U-stage d
// apply their one-qubit correction (see ParamInfo) -- partners through an exchange buffer in LDS, so that one
// code path serves whatever position the qubit has -- the states of the last U-stage behind the last phase;
// then every state measures both shifted values against psi.  Until its correction a state simply repeats
// the unshifted one.  Nothing is patched.  The run records are built 64 at a time, one per lane, in the
// workgroup's area of a.recs.
// GCK: the checkpoints (not psi) live in the workgroup's area of a.ckpt in global memory instead of LDS
// (seven qubits and more: in LDS they would leave room for too few workgroups on a CU)
template <bool GCK>
__global__ __launch_bounds__(64, VQA_GROUP_WAVES) void k_reg_group_late(RegArgs a) {
    constexpr int BLOCK = 64;
    constexpr bool SCALED = false;
    extern __shared__ double2 smem[];
    const Prog pp = load_prog(a.prog);
    const int n = pp.H->n, T = pp.H->T, S = pp.H->S;
    const int NU = pp.H->n_uent, N1 = pp.H->n_d1, N2 = pp.H->n_d2, ND = pp.H->n_ddesc;
    const int tab_stride = NU + N1 + N2;
    const int tid = threadIdx.x;
    const int sl = tid >> T;
    const int t = tid & ((1 << T) - 1);

    double2* base = smem;                                  // the sample's unshifted tables (all states read them)
    double2* bag = base + tab_stride;                      // (A, G) of every fused 2x2 (scratch of prep_base)
    double2* xbuf = bag + 2 * (size_t)NU;
    double* red = (double*)(xbuf + (pp.H->has_xpose ? ((size_t)1 << (n - 1)) : 0));
    unsigned short* smask = (unsigned short*)(red + 128);
    // the state in front of diagonal stage d as image d (LDS or global), the final state psi (LDS)
    double2* lds_img = (double2*)((char*)smask + (((size_t)pp.H->n_sign * BLOCK * 2 + 15) & ~(size_t)15));
    double2* ckpt = GCK ? a.ckpt + (size_t)blockIdx.x * ((size_t)ND << n) : lds_img;
    double2* psi = GCK ? lds_img : lds_img + ((size_t)ND << n);
    double* basev = (double*)(psi + ((size_t)1 << n));     // 79 values, squared norm of psi
    double2* xch = (double2*)(basev + 80);                 // exchange buffer: [state][amplitude r][thread]
    // run records, 64 at a time (one per lane; the chunks of a batch read theirs from the workgroup's area)
    RunRec* grec = a.recs + (size_t)blockIdx.x * 64;
    int rec_batch = -1;

    Ctx cx;
    cx.mat = base;
    cx.ph1 = base + NU;
    cx.ph2 = cx.ph1 + N1;
    cx.xbuf = (char*)xbuf;
    cx.smask = smask;
    cx.ftab = nullptr;
    cx.T = T; cx.nthr = BLOCK;
    cx.wave = 0; cx.tmask = (1 << T) - 1;
    cx.xc_id = -1; cx.xc_wt = 0; cx.xc_rt = 0;
    cx.area = __builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0x00020000);
    cx.image_bytes = 0;
#ifdef VQA_PROF
    cx.flag_mask = -1;
#endif
    thread_constants<BLOCK>(pp, smask, T, t, tid);

    const int K = 2 * a.spec.P + 1;
    const int NL = a.n_live;
    // entry 0 of a sample is the unshifted circuit itself, entry e > 0 run e - 1 of the list
    const int chunks = (NL + 1 + S - 1) / S;
    const int64_t total = a.E * chunks;
    const int64_t w0 = total * blockIdx.x / gridDim.x, w1 = total * (blockIdx.x + 1) / gridDim.x;
    const int end_pc = pp.H->n_phases - 1;
    int64_t cur = -1;
    unsigned parity = 0;
    for (int64_t w = w0; w < w1; ++w) {
        const int64_t j = w / chunks;
        const int ch = (int)(w - j * chunks);
        int b = a.spec.sel ? a.spec.sel[j] : (int)j;
        const bool valid = a.spec.B == 0 || (unsigned)b < (unsigned)a.spec.B;
        if (!valid) b = 0;
        const double* row = a.input + (size_t)b * a.n_input;
        __syncthreads();   // the previous item is done with the tables and the records
        const bool fresh = j != cur;
        if (fresh) {
            prep_base<SCALED>(a, pp, base, bag, row, NU, N1, N2, tid, BLOCK);
            cur = j;
            __syncthreads();
        }
        // the chunk: state sl takes entry ch * S + sl
        const int e = ch * S + sl;
        const bool live = e >= 1 && e <= NL;
        const int batch = (ch * S) >> 6;
        if (fresh || batch != rec_batch) {
            const int ej = (batch << 6) + tid;
            if (ej >= ch * S && ej >= 1 && ej <= NL)
                make_run_record<BLOCK, SCALED, 2>(a, pp, a.live[ej - 1], row, NU, N1, grec + tid, bag);
            rec_batch = batch;
            __threadfence_block();
            __syncthreads();
        }
        const RunRec* srec = grec + (e & 63);
        int run_stage = 1 << 30, my_mask = 0, kout = 0;
        if (live) {
            run_stage = srec->out[2];
            my_mask = srec->ids[3];
            kout = srec->out[1];
        }
        // A workgroup's first item of a sample also runs the unshifted circuit from |0...0>, which leaves the
        // checkpoints, psi and the unshifted values: as state 0 of chunk 0, side by side with the first S - 1
        // runs (which then start from |0...0> too), or as a pass of its own in front of a later chunk.
        for (int pass = fresh && ch != 0 ? 0 : 1; pass < 2; ++pass) {
            const bool zero = fresh && (pass == 0 || ch == 0);
            const int my_stage = pass == 1 ? run_stage : 1 << 30;
            const int u0 = __builtin_amdgcn_readfirstlane(my_stage);   // (the list is ordered by stage: state 0's is the first)
            double2 amp[kRegAmps];
            if (zero) {
#pragma unroll
                for (int q = 0; q < kRegAmps; ++q) amp[q] = make_double2(0.0, 0.0);
                if (t == 0) amp[0].x = 1.0;
            } else {
                const double2* img = u0 == ND ? psi : ckpt + ((size_t)u0 << n);
#pragma unroll
                for (int q = 0; q < kRegAmps; ++q) amp[q] = img[(q << T) + t];
            }
            // in front of diagonal stage d (d == ND: behind the last phase): the unshifted state leaves its image,
            // the states whose parameter sits in U-stage d apply their correction
            auto hook = [&](double2 (&v)[kRegAmps], int d) {
                if (zero && sl == 0) {
                    double2* img = d == ND ? psi : ckpt + ((size_t)d << n);
#pragma unroll
                    for (int q = 0; q < kRegAmps; ++q) img[(q << T) + t] = v[q];
                }
                const bool mine = my_stage == d;
                if (__ballot(mine) == 0) return;
                if (mine) {
                    const int p = __ffs(my_mask) - 1;
                    const double2* rec = (const double2*)srec;
                    const double2 c00 = rec[0], c01 = rec[1], c10 = rec[2], c11 = rec[7];
                    double2* mine_x = xch + ((size_t)sl << (kRegBits + T));
#pragma unroll
                    for (int q = 0; q < kRegAmps; ++q) mine_x[(q << T) + t] = v[q];
                    const int flip_q = p < kRegBits ? 1 << p : 0, flip_t = p < kRegBits ? 0 : 1 << (p - kRegBits);
#pragma unroll
                    for (int q = 0; q < kRegAmps; ++q) {
                        const double2 pa = mine_x[((q ^ flip_q) << T) + (t ^ flip_t)];
                        const bool up = (((q | (t << kRegBits)) >> p) & 1) != 0;
                        const double csx = up ? c11.x : c00.x, csy = up ? c11.y : c00.y;
                        const double cox = up ? c10.x : c01.x, coy = up ? c10.y : c01.y;
                        clane_inplace(v[q].x, v[q].y, pa.x, pa.y, csx, csy, cox, coy);
                    }
                }
            };
            const int pc0 = zero ? 0 : u0 < ND ? pp.ddesc[u0].pc : end_pc;
            run_phases<SCALED, false>(amp, pp, cx, pc0, !zero, false, false, hook);
            hook(amp, ND);
            if (zero) {
                if (GCK) __threadfence_block();
                const double nrm = measure(amp, a, pp, cx, red, parity, j * K, sl == 0, valid, sl == 0 ? basev : nullptr);
                if (tid == 0) basev[79] = nrm;
                __syncthreads();
                // (the workgroup that runs a sample's first chunk also writes the rows of the parameters outside the light cone)
                if (ch == 0) fill_dead_params(basev, j * K, valid, tid, BLOCK);
            }
            if (pass == 1 && live)
                measure_pair(amp, 0u, basev[79], (const double2*)srec, basev, a, pp, cx, red, parity, j * K + kout, valid, psi);
        }
    }
}

}  // namespace

static int check_reg_call(const EvalCall& call) {
    for (int t = 0; t < call.obs.n_terms; ++t)
        if (call.obs.x[t] != 0)
            VQA_FAIL(VQA_EUNSUPPORTED, "the register-blocked engine measures Z-basis observables only");
    if (call.n_marg_bits > 6)
        VQA_FAIL(VQA_EUNSUPPORTED, "register-blocked engine: n_marg_bits <= 6 (got %d)", call.n_marg_bits);
    return VQA_OK;
}

static RegArgs make_args(const EvalCall& call, const void* d_prog) {
    RegArgs a;
    a.prog = (const char*)d_prog;
    a.theta = call.d_theta;
    a.input = call.d_input;
    a.n_input = call.c->n_input;
    a.spec = call.spec;
    a.E = call.E;
    a.obs = call.obs;
    a.n_marg_bits = call.n_marg_bits;
    a.out_expect = call.d_out_expect;
    a.out_marginal = call.d_out_marginal;
    a.ckpt = nullptr;
    a.next = nullptr;
    a.next_base = 0;
    a.dead = nullptr;
    a.n_dead = 0;
    a.recs = nullptr;
    a.live = nullptr;
    a.n_live = 0;
    a.need_psi = 0;
    a.split = 1;
    a.use_ftab = 1;
#ifdef VQA_PROF
    const char* abl = getenv("VQA_ABLATE");
    a.ablate = abl ? atoi(abl) : 0;
#endif
    return a;
}

int reg_engine_run(const EvalCall& call, const void* d_prog, const RegProgram& P) {
    int rc = check_reg_call(call);
    if (rc) return rc;
    if (call.E <= 0) return VQA_OK;
    RegArgs a = make_args(call, d_prog);
    const int S = P.hdr.S;
    const void* fn = P.hdr.block == 64 ? (const void*)k_reg_eval<64>
                   : P.hdr.block == 128 ? (const void*)k_reg_eval<128>
                   : P.hdr.block == 256 ? (const void*)k_reg_eval<256>
                   : P.hdr.block == 512 ? (const void*)k_reg_eval<512>
                                        : (const void*)k_reg_eval<1024>;
    VQA_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.lds_bytes));
    const int64_t groups = (call.E + S - 1) / S;
    // resident workgroups per CU: LDS, and the register budget of the launch bounds
    const int by_regs = std::max(1, (P.hdr.block == 1024 ? 4 : P.hdr.block == 512 ? 2 : VQA_WAVES) * 4 / (P.hdr.block / 64));
    const int by_lds = (int)((160 * 1024) / std::max<size_t>(P.lds_bytes, 1));
    const int per_cu = std::max(1, std::min(by_regs, by_lds));
    const int grid = (int)std::min<int64_t>(groups, (int64_t)256 * per_cu);
    switch (P.hdr.block) {
    case 64: hipLaunchKernelGGL(k_reg_eval<64>, dim3(grid), dim3(64), P.lds_bytes, call.stream, a); break;
    case 128: hipLaunchKernelGGL(k_reg_eval<128>, dim3(grid), dim3(128), P.lds_bytes, call.stream, a); break;
    case 256: hipLaunchKernelGGL(k_reg_eval<256>, dim3(grid), dim3(256), P.lds_bytes, call.stream, a); break;
    case 512: hipLaunchKernelGGL(k_reg_eval<512>, dim3(grid), dim3(512), P.lds_bytes, call.stream, a); break;
    default: hipLaunchKernelGGL(k_reg_eval<1024>, dim3(grid), dim3(1024), P.lds_bytes, call.stream, a); break;
    }
    VQA_HIP(hipGetLastError());
    return VQA_OK;
}

// Sweep kernel: usable when every parameter enters at most one table entry and
// a state fills a workgroup (n >= 10); otherwise the caller runs the sweep as
// an explicit list.
bool reg_engine_can_sweep(const RegProgram& P) {
    if (P.hdr.S != 1) return false;
    for (const ParamInfo& pi : P.params)
        if (pi.ref_count > 1) return false;
    return true;
}

int reg_engine_sweep(const EvalCall& call, const void* d_prog, const RegProgram& P, int n_samples) {
    int rc = check_reg_call(call);
    if (rc) return rc;
    if (n_samples <= 0) return VQA_OK;
    RegArgs a = make_args(call, d_prog);
    a.E = n_samples;
    // the runs of the parameters that can matter for this call's observables; the rows of the
    // others are copies of the unshifted row (filled in below)
    ConeInfo& cone = *call.cone;
    {
        std::lock_guard<std::mutex> lk(call.c->mu);
        if (!cone.reg_built) {
            for (int r = 0; r < P.hdr.n_runs; ++r)
                if (!cone.dead[P.runs[r].param]) cone.reg_live_runs.push_back(r);
            // by the checkpoint a run starts from (late resumption: the diagonal stage behind its U-stage), so that
            // runs that load the same image follow one another (pair_up below; the image is warm in L2 for the next)
            auto stage_of = [&](int32_t r) {
                const ParamInfo& pi = P.params[P.runs[r].param];
                return (P.runs[r].variant == 0 && pi.late_mask != 0) ? pi.late_diag : pi.resume_diag;
            };
            std::stable_sort(cone.reg_live_runs.begin(), cone.reg_live_runs.end(),
                             [&](int32_t x, int32_t y) { return stage_of(x) < stage_of(y); });
            cone.reg_built = true;
        }
    }
    // a caller's parameter subset: per-call copies of the two lists (the call then waits for the
    // stream before it returns, so that the asynchronous copies have read them)
    std::vector<int32_t> m_live, m_dead;
    if (call.param_mask) {
        for (int32_t r : cone.reg_live_runs)
            if (call.param_mask[P.runs[r].param]) m_live.push_back(r);
        for (int32_t i : cone.dead_list)
            if (call.param_mask[i]) m_dead.push_back(i);
    }
    const std::vector<int32_t>& runs_1 = call.param_mask ? m_live : cone.reg_live_runs;
    const std::vector<int32_t>& dead_list = call.param_mask ? m_dead : cone.dead_list;
    const int n_dead = (int)dead_list.size();
    // Two runs side by side (k_reg_psr2): the list becomes items of two runs, pairs of late-resumption runs
    // that load the same checkpoint, or a run on its own with -1 beside it.  Taken when a state spans at most
    // eight waves, the two half-size transposition buffers fit LDS and at least a third of the runs find a partner.
    auto pair_up = [&](const std::vector<int32_t>& runs, std::vector<int32_t>& items) {
        int n_pairs = 0;
        auto late_stage = [&](int32_t r) {
            const ParamInfo& pi = P.params[P.runs[r].param];
            return (P.runs[r].variant == 0 && pi.late_mask != 0) ? pi.late_diag : -1;
        };
        for (size_t k = 0; k < runs.size(); ++k) {
            const int ls = late_stage(runs[k]);
            if (ls >= 0 && k + 1 < runs.size() && late_stage(runs[k + 1]) == ls) {
                items.push_back(runs[k]); items.push_back(runs[k + 1]);
                ++k; ++n_pairs;
            } else {
                items.push_back(runs[k]); items.push_back(-1);
            }
        }
        return n_pairs;
    };
    const size_t half_x = P.hdr.has_xpose ? ((size_t)8 << P.hdr.n) : 0;
    // + state B's buffer, the second record, the patch slots; the sign masks as bytes
    const size_t mask16 = ((size_t)P.hdr.n_sign * P.hdr.block * 2 + 15) & ~(size_t)15, mask8 = ((size_t)P.hdr.n_sign * P.hdr.block + 15) & ~(size_t)15;
    const size_t lds_psr2 = P.lds_bytes_psr + half_x + 8 * 16 + 4 * 16 - mask16 + mask8;
    // (workgroups a CU holds either way: the side-by-side form must keep more runs in flight)
    const int by_regs = P.hdr.block <= 128 ? VQA_SMALL_WAVES * 4 / (P.hdr.block / 64) : 8;
    const int per_cu_1 = std::max<int>(1, std::min<int>(std::min(by_regs, VQA_WAVES * 4 / std::max(P.hdr.block / 64, 1)),
                                                        (int)((160 * 1024) / std::max<size_t>(P.lds_bytes_psr, 1))));
    const int per_cu_2 = std::min<int>(8 / std::max(P.hdr.block / 64, 1), (int)((160 * 1024 - 64) / std::max<size_t>(lds_psr2, 1)));
    // (measured, IBM-HEA with <Z0>, gradients/s side by side against one run per workgroup: 10 qubits 608 k / 598 k,
    // 11 qubits 206 k / 226 k, 12 qubits 79.0 k / 78.3 k with half the state-image traffic, 13 qubits 47.8 k / 42.7 k:
    // taken from 12 qubits on, where a state spans four waves or more)
    bool two_runs = P.hdr.block >= 256 && P.hdr.block <= 512 && 2 * per_cu_2 > per_cu_1 && !getenv("VQA_NO_PSR2");
    if (getenv("VQA_PSR2")) two_runs = P.hdr.block <= 512 && per_cu_2 >= 1;   // (experiments: wherever it runs)
    std::vector<int32_t> m_items;
    if (two_runs) {
        if (call.param_mask) {
            two_runs = 3 * pair_up(m_live, m_items) >= (int)m_live.size() / 2 && !m_items.empty();
        } else {
            std::lock_guard<std::mutex> lk(call.c->mu);
            if (!cone.reg_pairs_built) {
                cone.reg_pairs = pair_up(cone.reg_live_runs, cone.reg_live_items);
                cone.reg_pairs_built = true;
            }
            two_runs = 3 * cone.reg_pairs >= (int)cone.reg_live_runs.size() / 2 && !cone.reg_live_items.empty();
        }
    }
    const std::vector<int32_t>& live_runs = !two_runs ? runs_1 : call.param_mask ? m_items : cone.reg_live_items;
    const size_t n_live = live_runs.size();   // entries of the uploaded list (two per item with two_runs)
    // (each list starts on a 16-byte boundary: the allocation-free upload stores 16 bytes at a time)
    const size_t live_part = (n_live * sizeof(int32_t) + 15) & ~(size_t)15;
    const size_t live_bytes = (live_part + (size_t)n_dead * sizeof(int32_t) + 255) & ~(size_t)255;
    // resident workgroups per CU: what LDS allows, and the register budget of the launch bounds for
    // the one- and two-wave workgroups of 10 and 11 qubits (VQA_WAVES waves per SIMD)
    // (timing experiments: VQA_NO_FTAB=1 leaves the diagonal stages' thread-factor tables out of LDS)
    a.use_ftab = getenv("VQA_NO_FTAB") ? 0 : 1;
    const size_t lds_psr = two_runs ? lds_psr2 : P.lds_bytes_psr - (a.use_ftab ? 0 : P.lds_bytes_ftab);
    // (two runs side by side: 256 registers per thread, two waves per SIMD)
    const int per_cu = two_runs ? per_cu_2 : std::max<int>(1, std::min<int>(by_regs, (int)((160 * 1024) / std::max<size_t>(lds_psr, 1))));
    // every workgroup of the launch owns an area of state images and run records; a call with a
    // fixed region of working memory launches as many workgroups as the region holds areas for
    // (workgroups draw samples from a counter, so any grid size completes the sweep)
    const size_t wg_bytes = (size_t)(P.hdr.n_ddesc + 1) * ((size_t)16 << P.hdr.n) +
                            (size_t)std::max(P.hdr.n_runs, 1) * sizeof(RunRec);
    int64_t slots = (int64_t)256 * per_cu;
    const size_t room = call_workspace_room(call);
    if (room != SIZE_MAX) {
        if (room < 256 + live_bytes + wg_bytes)
            VQA_FAIL(VQA_EUNSUPPORTED, "this sweep needs %zu bytes of working memory per workgroup, the fixed region has %zu left",
                     256 + live_bytes + wg_bytes, room);
        slots = std::min<int64_t>(slots, (int64_t)((room - 256 - live_bytes) / wg_bytes));
    }
    // Few samples: the run list of a sample is cut into slices for different workgroups, each of
    // which runs the unshifted circuit for its own checkpoints first (a single circuit, the VQE
    // case, would otherwise keep one workgroup busy and the rest of the GPU idle).
    const int64_t n_runs_max = std::max(P.hdr.n_runs, 1);
    // (no more slices than the list has entries: workgroups without work would still hold a CU's LDS while they
    // start and leave, which keeps launches of other streams off the GPU)
    const int64_t n_entries = std::max<int64_t>(1, (int64_t)(two_runs ? n_live / 2 : n_live));
    const int split = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(slots / n_samples, n_runs_max), n_entries));
    const int grid = (int)std::min<int64_t>((int64_t)n_samples * split, slots);
    const size_t ck_bytes = (size_t)grid * (P.hdr.n_ddesc + 1) * ((size_t)16 << P.hdr.n);
    const size_t rec_bytes = (size_t)grid * std::max(P.hdr.n_runs, 1) * sizeof(RunRec);
    if (getenv("VQA_PLAN_DEBUG"))
        fprintf(stderr, "[vqa] register sweep: %s, %zu list entries, LDS %zu bytes, %d workgroups per CU, grid %d, split %d\n",
                two_runs ? "two runs side by side" : "one run per workgroup", n_live, lds_psr, per_cu, grid, split);
    void* ws = nullptr;
    rc = call_workspace(call, 256 + live_bytes + ck_bytes + rec_bytes, &ws);
    if (rc) return rc;
    a.ckpt = (double2*)((char*)ws + 256 + live_bytes);
    a.recs = (RunRec*)((char*)ws + 256 + live_bytes + ck_bytes);
    a.n_live = (int)(two_runs ? n_live / 2 : n_live);
    a.split = (int)std::max<size_t>(1, std::min<size_t>((size_t)split, (size_t)a.n_live));
    // The lists: a cone's default lists stay on the device; a caller's parameter subset and a call that must not
    // allocate upload theirs into the workspace (the cached host lists live as long as the circuit: the
    // asynchronous copies may read them after this returns).
    const int32_t *d_live = nullptr, *d_dead = nullptr;
    if (!call.param_mask && !call.fixed) {
        if ((rc = cone_lists_dev(call, two_runs ? 1 : 0, live_runs, dead_list, &d_live, &d_dead))) return rc;
    } else {
        int32_t* w_live = (int32_t*)((char*)ws + 256);
        int32_t* w_dead = (int32_t*)((char*)ws + 256 + live_part);
        if (n_live && (rc = call_upload(call, w_live, live_runs.data(), n_live * sizeof(int32_t)))) return rc;
        if (n_dead && (rc = call_upload(call, w_dead, dead_list.data(), (size_t)n_dead * sizeof(int32_t)))) return rc;
        d_live = w_live;
        d_dead = w_dead;
    }
    a.live = d_live;
    a.dead = d_dead;
    a.n_dead = n_dead;
    const void* fn = two_runs ? (P.hdr.block == 64 ? (const void*)k_reg_psr2<64>
                               : P.hdr.block == 128 ? (const void*)k_reg_psr2<128>
                               : P.hdr.block == 256 ? (const void*)k_reg_psr2<256>
                                                    : (const void*)k_reg_psr2<512>)
                   : P.hdr.block == 64 ? (const void*)k_reg_psr<64>
                   : P.hdr.block == 128 ? (const void*)k_reg_psr<128>
                   : P.hdr.block == 256 ? (const void*)k_reg_psr<256>
                   : P.hdr.block == 512 ? (const void*)k_reg_psr<512>
                                        : (const void*)k_reg_psr<1024>;
    VQA_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_psr));
    // The work counter: the device's own, which is never cleared (the launch is told where it stands), or, for a
    // call that must not allocate, the head of its region, cleared here.
    const unsigned draws = (unsigned)((int64_t)n_samples * a.split + grid);
    a.next = call_ticket_counter(call, draws, &a.next_base);
    const bool own_counter = a.next == nullptr;
    if (own_counter) {
        a.next = (unsigned*)ws;
        a.next_base = 0;
        VQA_HIP(hipMemsetAsync(a.next, 0, 256, call.stream));
    }
    if (two_runs) {
        switch (P.hdr.block) {
        case 64: hipLaunchKernelGGL(k_reg_psr2<64>, dim3(grid), dim3(64), lds_psr, call.stream, a); break;
        case 128: hipLaunchKernelGGL(k_reg_psr2<128>, dim3(grid), dim3(128), lds_psr, call.stream, a); break;
        case 256: hipLaunchKernelGGL(k_reg_psr2<256>, dim3(grid), dim3(256), lds_psr, call.stream, a); break;
        default: hipLaunchKernelGGL(k_reg_psr2<512>, dim3(grid), dim3(512), lds_psr, call.stream, a); break;
        }
    } else
    switch (P.hdr.block) {
    case 64: hipLaunchKernelGGL(k_reg_psr<64>, dim3(grid), dim3(64), lds_psr, call.stream, a); break;
    case 128: hipLaunchKernelGGL(k_reg_psr<128>, dim3(grid), dim3(128), lds_psr, call.stream, a); break;
    case 256: hipLaunchKernelGGL(k_reg_psr<256>, dim3(grid), dim3(256), lds_psr, call.stream, a); break;
    case 512: hipLaunchKernelGGL(k_reg_psr<512>, dim3(grid), dim3(512), lds_psr, call.stream, a); break;
    default: hipLaunchKernelGGL(k_reg_psr<1024>, dim3(grid), dim3(1024), lds_psr, call.stream, a); break;
    }
    if (hipGetLastError() != hipSuccess) {
        if (!own_counter) call_ticket_reset(call);   // (the counter will not advance as the host assumed)
        VQA_FAIL(VQA_EHIP, "launch of the sweep kernel failed");
    }
    if (call.param_mask) VQA_HIP(hipStreamSynchronize(call.stream));
    return VQA_OK;
}

// Grouped sweeps of small circuits (k_reg_group): several states per workgroup and the tables fit.
static size_t group_lds_bytes(const RegProgram& P, bool need_psi) {
    // + the sample's unshifted tables and (A, G) pairs; with pair runs: psi, the unshifted values,
    // one run record per state
    return P.lds_bytes + ((size_t)P.hdr.n_d1 + P.hdr.n_d2) * 16 +   // (lds_bytes counts 3 entries per fused 2x2 already)
           (need_psi ? ((size_t)16 << P.hdr.n) + 80 * sizeof(double) + (size_t)P.hdr.S * sizeof(RunRec) : 0);
}
// k_reg_group_late: one copy of the tables, (A, G) scratch, the checkpoints and psi, the unshifted values, one
// run record per state, the exchange buffer (16 amplitudes of 64 threads)
static size_t group_late_lds_bytes(const RegProgram& P, bool global_ckpt) {
    const RegHeader& H = P.hdr;
    const size_t tab1 = ((size_t)H.n_uent + H.n_d1 + H.n_d2) * 16;
    const size_t xbuf = H.has_xpose ? ((size_t)8 << H.n) : 0;
    return tab1 + (size_t)H.n_uent * 32 + xbuf + 128 * sizeof(double) + (((size_t)H.n_sign * 64 * 2 + 15) & ~(size_t)15) +
           ((size_t)((global_ckpt ? 0 : H.n_ddesc) + 1) * 16 << H.n) + 80 * sizeof(double) + (size_t)64 * kRegAmps * 16 + 64;
}
// (the checkpoints in LDS must leave room for five workgroups on a CU; else they go to global memory)
static bool group_late_global_ckpt(const RegProgram& P) { return group_late_lds_bytes(P, false) > 32 * 1024; }
bool reg_engine_can_group(const RegProgram& P) {
    return P.hdr.S > 1 && P.hdr.block == 64 && group_lds_bytes(P, true) <= 160 * 1024;
}

int reg_engine_group(const EvalCall& call, const void* d_prog, const RegProgram& P, int n_samples) {
    int rc = check_reg_call(call);
    if (rc) return rc;
    if (n_samples <= 0) return VQA_OK;
    RegArgs a = make_args(call, d_prog);
    a.E = n_samples;
    // the runs to execute: those of the parameters that can matter for this call's observables (pair
    // runs first, so that the states of a chunk mostly measure the same way); the rows of the other
    // parameters are copies of the unshifted row
    ConeInfo& cone = *call.cone;
    const int S = P.hdr.S;
    auto list_runs = [&](std::vector<int32_t>& out, const uint8_t* mask) {
        for (int pass = 0; pass < 2; ++pass)
            for (int r = 0; r < P.hdr.n_runs; ++r) {
                const int i = P.runs[r].param;
                if (cone.dead[i] || (mask && !mask[i]) || (P.runs[r].variant == 0) != (pass == 0)) continue;
                out.push_back(r);
            }
        // every shift on its own when the pair runs (plus the unshifted run they need) save too few chunks
        int n_shifts = 1;
        for (int i = 0; i < call.c->n_theta; ++i)
            if (!cone.dead[i] && (!mask || mask[i])) n_shifts += 2;
        const size_t chunks_pair = (out.size() + S - 1) / S + 1, chunks_plain = ((size_t)n_shifts + S - 1) / S;
        if (out.empty() || 20 * chunks_pair > 17 * chunks_plain) {
            out.clear();
            out.push_back(-1);
            for (int i = 0; i < call.c->n_theta; ++i)
                if (!cone.dead[i] && (!mask || mask[i])) { out.push_back(-(2 * i + 1) - 1); out.push_back(-(2 * i + 2) - 1); }
        }
    };
    {
        std::lock_guard<std::mutex> lk(call.c->mu);
        if (!cone.group_built) {
            list_runs(cone.group_live_k, nullptr);
            cone.group_built = true;
        }
    }
    std::vector<int32_t> m_live, m_dead;
    if (call.param_mask) {
        list_runs(m_live, call.param_mask);
        for (int i = 0; i < call.c->n_theta; ++i)
            if (call.param_mask[i] && cone.dead[i]) m_dead.push_back(i);
    }
    // late resumption (k_reg_group_late) when every run of the list belongs to a parameter that drives one
    // rotation of one fused 2x2: one entry per parameter (the pair run, or the +shift run of a parameter of
    // the last U-stage), ordered by the U-stage
    auto late_list = [&](const uint8_t* mask, std::vector<int32_t>& out) {
        out.clear();
        if (getenv("VQA_NO_GROUP_LATE")) return;
        for (int r = 0; r < P.hdr.n_runs; ++r) {
            const int i = P.runs[r].param;
            if (cone.dead[i] || (mask && !mask[i])) continue;
            if (P.params[i].late_mask == 0) { out.clear(); return; }
            if (P.runs[r].variant != 2) out.push_back(r);
        }
        std::stable_sort(out.begin(), out.end(), [&](int32_t x, int32_t y) {
            return P.params[P.runs[x].param].late_diag < P.params[P.runs[y].param].late_diag;
        });
        if (group_late_lds_bytes(P, group_late_global_ckpt(P)) > 32 * 1024) out.clear();
    };
    {
        std::lock_guard<std::mutex> lk(call.c->mu);
        if (!cone.group_late_built) {
            late_list(nullptr, cone.group_late_k);
            cone.group_late_built = true;
        }
    }
    std::vector<int32_t> m_late;
    if (call.param_mask) late_list(call.param_mask, m_late);
    const std::vector<int32_t>& late_k = call.param_mask ? m_late : cone.group_late_k;
    bool late = !late_k.empty();
    const std::vector<int32_t>& dead_list = call.param_mask ? m_dead : cone.dead_list;
    const size_t n_dead = dead_list.size();
    const size_t dead_part = (n_dead * sizeof(int32_t) + 255) & ~(size_t)255;
    // launch shape and workspace (the late form's run records and global checkpoints are carved per workgroup)
    const std::vector<int32_t>* live_p = nullptr;
    bool gck = false;
    size_t lds = 0, live_part = 0, ck_bytes = 0, rec_bytes = 0;
    int grid = 0;
    auto shape = [&]() {
        live_p = late ? &late_k : call.param_mask ? &m_live : &cone.group_live_k;
        const size_t nl = live_p->size();
        gck = late && group_late_global_ckpt(P);
        lds = late ? group_late_lds_bytes(P, gck) : group_lds_bytes(P, (*live_p)[0] >= 0);
        // (the late form's list starts with the unshifted circuit itself)
        const int64_t items = (int64_t)n_samples * (int64_t)std::max<size_t>((nl + (late ? 1 : 0) + S - 1) / S, 1);
        const int by_lds = (int)((160 * 1024) / std::max<size_t>(lds, 1));
        const int per_cu = std::max(1, std::min(VQA_GROUP_WAVES * 4, by_lds));
        grid = (int)std::min<int64_t>(items, (int64_t)256 * per_cu);
        live_part = (nl * sizeof(int32_t) + 15) & ~(size_t)15;   // (16-byte aligned lists)
        ck_bytes = gck ? (size_t)grid * ((size_t)P.hdr.n_ddesc * 16 << P.hdr.n) : 0;
        rec_bytes = late ? (size_t)grid * 64 * sizeof(RunRec) : 0;
    };
    shape();
    // (a call that must not allocate has a fixed workspace: when the late form's areas do not fit it, the earlier form runs)
    if (late && call_workspace_room(call) < live_part + dead_part + ck_bytes + rec_bytes + 1024) {
        late = false;
        shape();
    }
    const std::vector<int32_t>& live_k = *live_p;
    const size_t n_live = live_k.size();
    void* ws = nullptr;
    rc = call_workspace(call, live_part + dead_part + ck_bytes + rec_bytes + 1024, &ws);
    if (rc) return rc;
    const int32_t *d_live = nullptr, *d_dead = nullptr;
    a.ckpt = (double2*)((char*)ws + ((live_part + dead_part + 255) & ~(size_t)255));
    a.recs = (RunRec*)((char*)a.ckpt + ((ck_bytes + 255) & ~(size_t)255));
    // a cone's default lists stay on the device; a caller's parameter subset and a call that must not allocate
    // upload theirs (the cached host lists live as long as the circuit; a per-call list is waited for below)
    if (!call.param_mask && !call.fixed) {
        if ((rc = cone_lists_dev(call, late ? 3 : 2, live_k, dead_list, &d_live, &d_dead))) return rc;
    } else {
        int32_t* w_live = (int32_t*)ws;
        int32_t* w_dead = (int32_t*)((char*)ws + live_part);
        if (n_live && (rc = call_upload(call, w_live, live_k.data(), n_live * sizeof(int32_t)))) return rc;
        if (n_dead && (rc = call_upload(call, w_dead, dead_list.data(), n_dead * sizeof(int32_t)))) return rc;
        d_live = w_live;
        d_dead = w_dead;
    }
    a.live = d_live;
    a.n_live = (int)n_live;
    a.need_psi = live_k[0] >= 0;
    if (late) {   // (the late form writes the dead parameters' rows itself)
        a.dead = d_dead;
        a.n_dead = (int)n_dead;
    }
    const void* fn = !late ? (const void*)k_reg_group : gck ? (const void*)k_reg_group_late<true> : (const void*)k_reg_group_late<false>;
    VQA_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (!late) hipLaunchKernelGGL(k_reg_group, dim3(grid), dim3(64), lds, call.stream, a);
    else if (gck) hipLaunchKernelGGL(k_reg_group_late<true>, dim3(grid), dim3(64), lds, call.stream, a);
    else hipLaunchKernelGGL(k_reg_group_late<false>, dim3(grid), dim3(64), lds, call.stream, a);
    VQA_HIP(hipGetLastError());
    if (n_dead && !late) {
        const int K = 2 * call.c->n_theta + 1;
        rc = fill_dead_rows(call.stream, call.d_out_expect, n_samples, K, call.obs.n_obs, d_dead, (int)n_dead);
        if (rc) return rc;
        if (call.n_marg_bits > 0)
            rc = fill_dead_rows(call.stream, call.d_out_marginal, n_samples, K, 1 << call.n_marg_bits, d_dead, (int)n_dead);
        if (rc) return rc;
    }
    if (call.param_mask) VQA_HIP(hipStreamSynchronize(call.stream));
    return VQA_OK;
}

#ifdef VQA_PROF
extern "C" int vqa_debug_stamps(unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = { 0 };
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

}  // namespace vqa
