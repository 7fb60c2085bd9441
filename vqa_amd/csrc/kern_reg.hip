// Engine 2: register-blocked fused statevector kernel (4 <= n <= 13).
//
// The state of one circuit instance lives in the VGPRs of 2^(n-4) threads, 16
// complex128 amplitudes per thread.  The program compiled by reg_compile.cpp is
// interpreted per instance:
//   - per-instance preparation: every fused 2x2 is multiplied out from its
//     angle sources, split into RZ RY RZ, and its phases are merged into the
//     diagonal stages' phase tables (all in LDS);
//   - real rotations on register bits are pure FMA work; on lane bits the
//     partner amplitude comes through DPP / swizzle moves; qubits on wave bits
//     are brought into registers by a conflict-free transposition through LDS;
//   - diagonal stages are one complex multiply per amplitude by a 16-entry
//     table built per thread, with CZ / Z signs folded in from bit tables;
//   - the Z-basis observables and the low-bit marginal are reduced in place.
//
// Replaces, per instance, LearningCircuit.run(x) + Observable expectation
// (vqa/QNN/myQNNs_base.py:692-704, 907-914) for the ansaetze of
// vqa/QNN/QML_utils.py:33-164.
#include "reg_program.h"
#include "vqa_core.h"

namespace vqa {

namespace {

struct RegArgs {
    const char* prog;  // device blob (RegHeader + arrays)
    const double* theta;
    const double* input;
    int n_input;
    EvalSpec spec;
    int64_t E;
    ObsPack obs;
    int n_marg_bits;
    double* out_expect;
    double* out_marginal;
};

struct cplx {
    double x, y;
};
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return { a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x }; }
__device__ __forceinline__ cplx cconj(cplx a) { return { a.x, -a.y }; }
__device__ __forceinline__ cplx csel_conj(cplx a, bool c) { return { a.x, c ? -a.y : a.y }; }

// square root of a unit-modulus complex number (either root)
__device__ __forceinline__ cplx sqrt_phase(cplx z) {
    cplx w = z.x >= 0.0 ? cplx{ 1.0 + z.x, z.y } : cplx{ z.y, 1.0 - z.x };  // 1+z  or  i(1-z)
    const double inv = rsqrt(w.x * w.x + w.y * w.y);
    return { w.x * inv, w.y * inv };
}

// ---- cross-lane partner fetch: value of lane (lane ^ (1 << LB)) -----------------
template <int LB>
__device__ __forceinline__ int xor_fetch32(int v) {
    if constexpr (LB == 0) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (LB == 1) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (LB == 2) {
        int t = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, false);  // row_shl:4 into banks 0,2
        return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);    // row_shr:4 into banks 1,3
    } else if constexpr (LB == 3) return __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, true);  // row_ror:8
    else if constexpr (LB == 4) return __builtin_amdgcn_ds_swizzle(v, 0x401F);                 // xor 16 within 32
    else return __builtin_amdgcn_ds_bpermute(((int)(threadIdx.x & 63) ^ 32) << 2, v);
}
template <int LB>
__device__ __forceinline__ double xor_fetch(double v) {
    const int lo = xor_fetch32<LB>(__double2loint(v));
    const int hi = xor_fetch32<LB>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <int J>
__device__ __forceinline__ void apply_ureg(double (&re)[kRegAmps], double (&im)[kRegAmps], double c, double s) {
#pragma unroll
    for (int r = 0; r < kRegAmps; ++r) {
        if (!(r & (1 << J))) {
            const int r1 = r | (1 << J);
            const double a0 = re[r], a1 = re[r1], b0 = im[r], b1 = im[r1];
            re[r] = c * a0 + s * a1;
            re[r1] = c * a1 - s * a0;
            im[r] = c * b0 + s * b1;
            im[r1] = c * b1 - s * b0;
        }
    }
}

template <int LB>
__device__ __forceinline__ void apply_ulane(double (&re)[kRegAmps], double (&im)[kRegAmps], double c, double s,
                                            int lane) {
    const double se = ((lane >> LB) & 1) ? -s : s;
#pragma unroll
    for (int r = 0; r < kRegAmps; ++r) {
        const double pr = xor_fetch<LB>(re[r]);
        const double pi = xor_fetch<LB>(im[r]);
        re[r] = c * re[r] + se * pr;
        im[r] = c * im[r] + se * pi;
    }
}

// program arrays are read through the constant address space: a uniform index
// then becomes a scalar load (the program never changes during a launch)
#define VQA_CAS __attribute__((address_space(4)))
template <class Tp>
__device__ __forceinline__ const VQA_CAS Tp* to_const(const void* p) {
    return (const VQA_CAS Tp*)(uintptr_t)p;
}

struct ProgPtrs {
    const VQA_CAS ElemGate* elems;
    const VQA_CAS UEntry* uent;
    const VQA_CAS D1Entry* d1;
    const VQA_CAS D2Entry* d2;
    const VQA_CAS ElemGate* terms;
};

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

// Per-instance preparation of the LDS tables of the S states of a workgroup:
// mat[u] = (cos, sin)(beta/2) of every fused 2x2, ph1 / ph2 = phase tables of the
// diagonal stages with the RZ factors of the neighbouring 2x2s merged in.
struct PrepArgs {
    const double* theta;
    const double* input;
    int n_input;
    EvalSpec spec;
    int64_t E;
};

__device__ __forceinline__ void prep_tables(const PrepArgs& a, const ProgPtrs& pp, double2* tabs, int64_t e0,
                                         int S, int NU, int N1, int N2, int tid, int nthr) {
    const int tab_stride = NU + N1 + N2;
    // phase 0: phase tables from the explicit RZ / ZZ terms
    for (int i = tid; i < S * (N1 + N2); i += nthr) {
        const int s2 = i / (N1 + N2), k = i - s2 * (N1 + N2);
        const int64_t e2 = e0 + s2;
        cplx ph = { 1.0, 0.0 };
        if (e2 < a.E) {
            const int begin = k < N1 ? pp.d1[k].begin : pp.d2[k - N1].begin;
            const int count = k < N1 ? pp.d1[k].count : pp.d2[k - N1].count;
            if (count > 0) {
                int b, sparam;
                double sval;
                decode_eval(a.spec, e2, b, sparam, sval);
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
    // phases 1, 2: fused 2x2 of every U-entry -> RZ RY RZ; even U-stages first,
    // so that no two entries of one phase update the same phase-table slot
    for (int parity = 0; parity < 2; ++parity) {
        __syncthreads();
        for (int i = tid; i < S * NU; i += nthr) {
            const int s2 = i / NU, u = i - s2 * NU;
            const int64_t e2 = e0 + s2;
            if (pp.uent[u].parity != parity || e2 >= a.E) continue;
            const int ubegin = pp.uent[u].begin, ucount = pp.uent[u].count;
            const int d_before = pp.uent[u].d_before, d_after = pp.uent[u].d_after;
            int b, sparam;
            double sval;
            decode_eval(a.spec, e2, b, sparam, sval);
            const double* row = a.input + (size_t)b * a.n_input;
            cplx m00 = { 1, 0 }, m01 = { 0, 0 }, m10 = { 0, 0 }, m11 = { 1, 0 };
            for (int j = 0; j < ucount; ++j) {
                const VQA_CAS ElemGate* g = pp.elems + ubegin + j;
                const int kind = g->kind;
                cplx g00, g01, g10, g11;
                const double r2 = 0.70710678118654752440;
                double c = 1.0, sn = 0.0;
                if (kind >= VQA_GATE_RX) sincos(0.5 * elem_angle_c(g, a.theta, row, sparam, sval), &sn, &c);
                switch (kind) {
                case VQA_GATE_H: g00 = { r2, 0 }; g01 = { r2, 0 }; g10 = { r2, 0 }; g11 = { -r2, 0 }; break;
                case VQA_GATE_X: g00 = { 0, 0 }; g01 = { 1, 0 }; g10 = { 1, 0 }; g11 = { 0, 0 }; break;
                case VQA_GATE_Y: g00 = { 0, 0 }; g01 = { 0, -1 }; g10 = { 0, 1 }; g11 = { 0, 0 }; break;
                case VQA_GATE_Z: g00 = { 1, 0 }; g01 = { 0, 0 }; g10 = { 0, 0 }; g11 = { -1, 0 }; break;
                case VQA_GATE_RX: g00 = { c, 0 }; g01 = { 0, sn }; g10 = { 0, sn }; g11 = { c, 0 }; break;
                case VQA_GATE_RY: g00 = { c, 0 }; g01 = { sn, 0 }; g10 = { -sn, 0 }; g11 = { c, 0 }; break;
                default: /* RZ */ g00 = { c, sn }; g01 = { 0, 0 }; g10 = { 0, 0 }; g11 = { c, -sn }; break;
                }
                // M <- G * M
                const cplx a0 = cmul(g00, m00), a1 = cmul(g01, m10), b0 = cmul(g00, m01), b1 = cmul(g01, m11);
                const cplx c0 = cmul(g10, m00), c1 = cmul(g11, m10), d0 = cmul(g10, m01), d1 = cmul(g11, m11);
                m00 = { a0.x + a1.x, a0.y + a1.y };
                m01 = { b0.x + b1.x, b0.y + b1.y };
                m10 = { c0.x + c1.x, c0.y + c1.y };
                m11 = { d0.x + d1.x, d0.y + d1.y };
            }
            // M = e^{i d} RZ(al) RY(be) RZ(ga):  |m00| = cos(be/2), |m01| = sin(be/2)
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
            cplx A, G;
            if (h0 && h1) {
                A = sqrt_phase(cmul(Xp, Yp));
                G = cmul(Xp, cconj(A));
            } else if (!h1) {
                A = Xp; G = { 1, 0 };
            } else {
                A = Yp; G = { 1, 0 };
            }
            double2* tb = tabs + (size_t)s2 * tab_stride;
            tb[u] = make_double2(a00 * inv, a01 * inv);
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
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_reg_eval(RegArgs a) {
    extern __shared__ double2 smem[];
    const VQA_CAS RegHeader* H = to_const<RegHeader>(a.prog);
    ProgPtrs pp;
    pp.elems = to_const<ElemGate>(a.prog + H->off_elems);
    pp.uent = to_const<UEntry>(a.prog + H->off_uent);
    pp.d1 = to_const<D1Entry>(a.prog + H->off_d1);
    pp.d2 = to_const<D2Entry>(a.prog + H->off_d2);
    pp.terms = to_const<ElemGate>(a.prog + H->off_terms);
    const VQA_CAS RegOp* ops = to_const<RegOp>(a.prog + H->off_ops);
    const VQA_CAS DDesc* ddesc = to_const<DDesc>(a.prog + H->off_ddesc);
    const VQA_CAS XDesc* xdesc = to_const<XDesc>(a.prog + H->off_xdesc);
    const VQA_CAS PairRef* pairs = to_const<PairRef>(a.prog + H->off_pairs);
    const VQA_CAS SignTab* signs = to_const<SignTab>(a.prog + H->off_signs);

    const int n = H->n, T = H->T, S = H->S;
    const int NU = H->n_uent, N1 = H->n_d1, N2 = H->n_d2;
    const int n_sign = H->n_sign;
    const int tab_stride = NU + N1 + N2;  // double2 entries per state
    const int tid = threadIdx.x, nthr = BLOCK;
    const int lane = tid & 63, wave = tid >> 6;
    const int sl = tid >> T;              // state within the workgroup
    const int t = tid & ((1 << T) - 1);   // thread bits of the amplitude index

    double2* tabs = smem;                             // S * tab_stride
    double2* xbuf = tabs + (size_t)S * tab_stride;    // 2^n slots when the program transposes
    double* red = (double*)(xbuf + (H->has_xpose ? ((size_t)1 << n) : 0));  // 64 doubles
    unsigned char* sbits = (unsigned char*)(red + 64);  // n_sign * BLOCK: per-thread CZ/Z sign bits

    double2* mat = tabs + (size_t)sl * tab_stride;
    double2* ph1 = mat + NU;
    double2* ph2 = ph1 + N1;

    // ---- per-thread constants of the whole launch ----
    for (int sg = 0; sg < n_sign; ++sg) {
        const int wi = t >> 5, bi = t & 31;
        unsigned v = (signs[sg].tt_par[wi] >> bi) & 1u;
#pragma unroll
        for (int j = 0; j < kRegBits; ++j) v |= ((signs[sg].cross[j][wi] >> bi) & 1u) << (1 + j);
        sbits[sg * BLOCK + tid] = (unsigned char)v;
    }
    unsigned kt = 0;  // qubit-index bits contributed by this thread at measurement time
    for (int i = 0; i < T; ++i)
        if ((t >> i) & 1) kt |= 1u << H->final_qubit_at_pos[kRegBits + i];
    unsigned krb[kRegBits];
#pragma unroll
    for (int j = 0; j < kRegBits; ++j) krb[j] = 1u << H->final_qubit_at_pos[j];

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
        prep_tables(PrepArgs{ a.theta, a.input, a.n_input, a.spec, a.E }, pp, tabs, e0, S, NU, N1, N2, tid, nthr);
        __syncthreads();

        // ---- state |0...0> ----
        double re[kRegAmps], im[kRegAmps];
#pragma unroll
        for (int r = 0; r < kRegAmps; ++r) { re[r] = 0.0; im[r] = 0.0; }
        if (t == 0) re[0] = 1.0;

        // ---- program (next op and its rotation are fetched while the current one runs) ----
        int pc = 0;
        int kind = ops[0].kind, oa = ops[0].a, ob = ops[0].b;
        double2 cs = make_double2(1.0, 0.0);
        if (kind == ROP_UREG || kind == ROP_ULANE) cs = mat[ob];
        while (kind != ROP_END) {
            const int nkind = ops[pc + 1].kind, na = ops[pc + 1].a, nb = ops[pc + 1].b;
            double2 ncs = make_double2(1.0, 0.0);
            if (nkind == ROP_UREG || nkind == ROP_ULANE) ncs = mat[nb];
            if (kind == ROP_UREG) {
                switch (oa) {
                case 0: apply_ureg<0>(re, im, cs.x, cs.y); break;
                case 1: apply_ureg<1>(re, im, cs.x, cs.y); break;
                case 2: apply_ureg<2>(re, im, cs.x, cs.y); break;
                default: apply_ureg<3>(re, im, cs.x, cs.y); break;
                }
            } else if (kind == ROP_ULANE) {
                switch (oa) {
                case 0: apply_ulane<0>(re, im, cs.x, cs.y, lane); break;
                case 1: apply_ulane<1>(re, im, cs.x, cs.y, lane); break;
                case 2: apply_ulane<2>(re, im, cs.x, cs.y, lane); break;
                case 3: apply_ulane<3>(re, im, cs.x, cs.y, lane); break;
                case 4: apply_ulane<4>(re, im, cs.x, cs.y, lane); break;
                default: apply_ulane<5>(re, im, cs.x, cs.y, lane); break;
                }
            } else if (kind == ROP_XPOSE) {
                const VQA_CAS XDesc* xd = xdesc + oa;
                int wt = 0, rt = 0;
                for (int i = 0; i < T; ++i) {
                    const int wc = xd->wconst[kRegBits + i], rc = xd->rconst[kRegBits + i];
                    if ((t >> i) & 1) { wt ^= wc; rt ^= rc; }
                }
                const int w0 = xd->wconst[0], w1 = xd->wconst[1], w2 = xd->wconst[2], w3 = xd->wconst[3];
                const int r0 = xd->rconst[0], r1 = xd->rconst[1], r2 = xd->rconst[2], r3 = xd->rconst[3];
#pragma unroll
                for (int r = 0; r < kRegAmps; ++r) {
                    const int slot = wt ^ ((r & 1) ? w0 : 0) ^ ((r & 2) ? w1 : 0) ^ ((r & 4) ? w2 : 0) ^ ((r & 8) ? w3 : 0);
                    xbuf[slot] = make_double2(re[r], im[r]);
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < kRegAmps; ++r) {
                    const int slot = rt ^ ((r & 1) ? r0 : 0) ^ ((r & 2) ? r1 : 0) ^ ((r & 4) ? r2 : 0) ^ ((r & 8) ? r3 : 0);
                    const double2 v = xbuf[slot];
                    re[r] = v.x;
                    im[r] = v.y;
                }
                __syncthreads();
            } else {  // ROP_DIAG
                const VQA_CAS DDesc* dd = ddesc + oa;
                const double2* p1 = ph1 + dd->ph1_base;
                // thread-only factor
                cplx F = { 1.0, 0.0 };
                for (int i = 0; i < T; ++i) {
                    const double2 p = p1[dd->qubit_at_pos[kRegBits + i]];
                    F = cmul(F, csel_conj({ p.x, p.y }, (t >> i) & 1));
                }
                const int tt_begin = dd->tt_begin, tt_count = dd->tt_count;
                for (int k = 0; k < tt_count; ++k) {
                    const int pa = pairs[tt_begin + k].a, pb = pairs[tt_begin + k].b;
                    const double2 z = ph2[pairs[tt_begin + k].ph2];
                    F = cmul(F, csel_conj({ z.x, z.y }, ((t >> pa) ^ (t >> pb)) & 1));
                }
                // effective phase of each register bit
                cplx pj[kRegBits];
#pragma unroll
                for (int j = 0; j < kRegBits; ++j) {
                    const double2 p = p1[dd->qubit_at_pos[j]];
                    pj[j] = { p.x, p.y };
                }
                const int rt_begin = dd->rt_begin, rt_count = dd->rt_count;
                for (int k = 0; k < rt_count; ++k) {
                    const int pa = pairs[rt_begin + k].a, pb = pairs[rt_begin + k].b;
                    const double2 z = ph2[pairs[rt_begin + k].ph2];
                    const cplx ze = csel_conj({ z.x, z.y }, (t >> pb) & 1);
#pragma unroll
                    for (int j = 0; j < kRegBits; ++j)
                        if (pa == j) pj[j] = cmul(pj[j], ze);
                }
                // 16-entry table by doubling: T[r | 1<<j] = T[r] * conj(pj)^2
                cplx tb[kRegAmps];
                tb[0] = cmul(cmul(F, cmul(pj[0], pj[1])), cmul(pj[2], pj[3]));
#pragma unroll
                for (int j = 0; j < kRegBits; ++j) {
                    const cplx cj = cconj(pj[j]);
                    const cplx w = cmul(cj, cj);
#pragma unroll
                    for (int r = 0; r < (1 << j); ++r) tb[r | (1 << j)] = cmul(tb[r], w);
                }
                // CZ / Z signs
                const int sign_id = dd->sign_id;
                if (sign_id >= 0) {
                    const unsigned sb = sbits[sign_id * BLOCK + tid];
                    const unsigned rr = signs[sign_id].rr_par;
#pragma unroll
                    for (int r = 0; r < kRegAmps; ++r) {
                        const unsigned sel = 1u | ((unsigned)r << 1);  // base bit + the cross bits of the set register bits
                        const unsigned sgn = (__popc(sb & sel) ^ (rr >> r)) & 1u;
                        if (sgn) { tb[r].x = -tb[r].x; tb[r].y = -tb[r].y; }
                    }
                }
#pragma unroll
                for (int r = 0; r < kRegAmps; ++r) {
                    const cplx v = cmul({ re[r], im[r] }, tb[r]);
                    re[r] = v.x;
                    im[r] = v.y;
                }
                // register-register ZZ phases
                const int rr_begin = dd->rr_begin, rr_count = dd->rr_count;
                for (int k = 0; k < rr_count; ++k) {
                    const int pa = pairs[rr_begin + k].a, pb = pairs[rr_begin + k].b;
                    const double2 z = ph2[pairs[rr_begin + k].ph2];
#pragma unroll
                    for (int r = 0; r < kRegAmps; ++r) {
                        const cplx ze = csel_conj({ z.x, z.y }, ((r >> pa) ^ (r >> pb)) & 1);
                        const cplx v = cmul({ re[r], im[r] }, ze);
                        re[r] = v.x;
                        im[r] = v.y;
                    }
                }
            }
            kind = nkind; oa = na; ob = nb; cs = ncs;
            ++pc;
        }

        // ---- measurement (Z basis) ----
        const double nanv = __longlong_as_double(0x7ff8000000000000LL);
        const int nq = a.obs.n_obs + (a.n_marg_bits > 0 ? (1 << a.n_marg_bits) : 0);
        for (int q = 0; q < nq; ++q) {
            double acc = 0.0;
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
                        const double pr = re[r] * re[r] + im[r] * im[r];
                        part += (rp & 1) ? -pr : pr;
                    }
                    acc += a.obs.coef[tm] * part;
                }
            } else {
                const unsigned m = (unsigned)(q - a.obs.n_obs), M1 = (1u << a.n_marg_bits) - 1;
#pragma unroll
                for (int r = 0; r < kRegAmps; ++r) {
                    unsigned k = kt;
#pragma unroll
                    for (int j = 0; j < kRegBits; ++j)
                        if (r & (1 << j)) k |= krb[j];
                    acc += ((k & M1) == m) ? re[r] * re[r] + im[r] * im[r] : 0.0;
                }
            }
            // sum over the threads of the state
            const int lb = T < 6 ? T : 6;
            for (int i = 0; i < lb; ++i) acc += __shfl_xor(acc, 1 << i, 64);
            if (T > 6) {
                __syncthreads();
                if (lane == 0) red[wave] = acc;
                __syncthreads();
                acc = 0.0;
                for (int w = 0; w < (BLOCK >> 6); ++w) acc += red[w];
            }
            if (t == 0 && live) {
                if (q < a.obs.n_obs) a.out_expect[e * a.obs.n_obs + q] = valid ? acc : nanv;
                else a.out_marginal[(e << a.n_marg_bits) + (q - a.obs.n_obs)] = valid ? acc : nanv;
            }
        }
    }
}

}  // namespace

int reg_engine_run(const EvalCall& call, const void* d_prog, const RegProgram& P) {
    for (int t = 0; t < call.obs.n_terms; ++t)
        if (call.obs.x[t] != 0)
            VQA_FAIL(VQA_EUNSUPPORTED, "the register-blocked engine measures Z-basis observables only");
    if (call.n_marg_bits > 6)
        VQA_FAIL(VQA_EUNSUPPORTED, "register-blocked engine: n_marg_bits <= 6 (got %d)", call.n_marg_bits);
    if (call.E <= 0) return VQA_OK;
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
    const int S = P.hdr.S;
    const void* fn = P.hdr.block == 64 ? (const void*)k_reg_eval<64>
                   : P.hdr.block == 128 ? (const void*)k_reg_eval<128>
                   : P.hdr.block == 256 ? (const void*)k_reg_eval<256>
                                        : (const void*)k_reg_eval<512>;
    VQA_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.lds_bytes));
    const int64_t groups = (call.E + S - 1) / S;
    const int per_cu = std::max<int>(1, std::min<int>(8, (int)((160 * 1024) / std::max<size_t>(P.lds_bytes, 1))));
    const int grid = (int)std::min<int64_t>(groups, (int64_t)256 * per_cu);
    switch (P.hdr.block) {
    case 64: hipLaunchKernelGGL(k_reg_eval<64>, dim3(grid), dim3(64), P.lds_bytes, call.stream, a); break;
    case 128: hipLaunchKernelGGL(k_reg_eval<128>, dim3(grid), dim3(128), P.lds_bytes, call.stream, a); break;
    case 256: hipLaunchKernelGGL(k_reg_eval<256>, dim3(grid), dim3(256), P.lds_bytes, call.stream, a); break;
    default: hipLaunchKernelGGL(k_reg_eval<512>, dim3(grid), dim3(512), P.lds_bytes, call.stream, a); break;
    }
    VQA_HIP(hipGetLastError());
    return VQA_OK;
}

}  // namespace vqa
