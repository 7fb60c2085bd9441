// Engine 1: LDS-resident, gate-by-gate statevector kernel (n <= 13).
//
// One 256-thread workgroup owns S complex128 states in LDS (S = 1 for n >= 9,
// S = 256 / 2^(n-1) below that, so small registers still fill the workgroup),
// applies the submitted gate list in order (LDS butterflies for dense gates,
// in-place phase/sign updates for diagonal ones) and reduces the observables
// and the low-bit marginal in place.  It executes the unfused gate list, which
// makes it the engine whose LDS traffic equals the algorithmic byte count
// (2 * 16 * 2^n bytes per gate), and the cross-check for the fused engine.
//
// Replaces, per instance, LearningCircuit.run(x) + Observable expectation
// (vqa/QNN/myQNNs_base.py:692-704, 907-914).
#include "vqa_core.h"

namespace vqa {

namespace {

constexpr int kThreads = 256;

struct LdsArgs {
    int n, n_ops, n_input, n_marg_bits;
    const vqa_op* ops;
    const double* theta;
    const double* input;
    EvalSpec spec;
    int64_t E;
    ObsPack obs;
    double* out_expect;
    double* out_marginal;
};

// sum `v` over the TPS consecutive threads of a state; result valid in the
// first thread of each segment.  TPS is a power of two <= 256.
__device__ __forceinline__ double seg_reduce(double v, double* red, int tid, int p, int TPS) {
    red[tid] = v;
    __syncthreads();
    for (int s = TPS >> 1; s > 0; s >>= 1) {
        if (p < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    double r = red[tid];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(kThreads) void k_lds_eval(LdsArgs a) {
    extern __shared__ double2 smem[];
    const int n = a.n;
    const int D = 1 << n;
    const int PP = D >> 1 ? D >> 1 : 1;              // pairs per state
    const int S = PP >= kThreads ? 1 : kThreads / PP;  // states per workgroup
    const int TPS = kThreads / S;                      // threads per state
    double2* st_all = smem;                            // S * D amplitudes
    double2* tab = st_all + (size_t)S * D;             // 256 (cos, sin) entries
    double* red = (double*)(tab + kThreads);           // 256 doubles

    const int tid = threadIdx.x;
    const int sidx = tid / TPS, p = tid - sidx * TPS;
    double2* st = st_all + (size_t)sidx * D;
    const int CH = TPS;  // gates per table chunk: S * CH = 256 entries

    for (int64_t e0 = (int64_t)blockIdx.x * S; e0 < a.E; e0 += (int64_t)gridDim.x * S) {
        const int64_t e = e0 + sidx;
        const bool live = e < a.E;
        bool valid = true;
        if (live) {
            int b_, sp_;
            double sv_;
            valid = decode_eval(a.spec, e, b_, sp_, sv_);
        }
        const double nanv = __longlong_as_double(0x7ff8000000000000LL);

        for (int k = p; k < D; k += TPS) st[k] = make_double2(k == 0 ? 1.0 : 0.0, 0.0);

        for (int gc = 0; gc < a.n_ops; gc += CH) {
            __syncthreads();
            {   // (cos, sin) of the half angle for (state ts, gate gg)
                const int ts = tid / CH, gg = gc + (tid - ts * CH);
                const int64_t e2 = e0 + ts;
                double2 cs = make_double2(1.0, 0.0);
                if (gg < a.n_ops && e2 < a.E) {
                    const vqa_op op = a.ops[gg];
                    if (op.kind >= VQA_GATE_RX && op.kind <= VQA_GATE_RZ) {
                        int b, sparam;
                        double sval;
                        decode_eval(a.spec, e2, b, sparam, sval);
                        const double ang =
                            op_angle(op, a.theta, a.input + (size_t)b * a.n_input, sparam, sval);
                        double sn, c;
                        sincos(0.5 * ang, &sn, &c);
                        cs = make_double2(c, sn);
                    }
                }
                tab[tid] = cs;
            }
            __syncthreads();
            const int gend = min(gc + CH, a.n_ops);
            for (int g = gc; g < gend; ++g) {
                const int kind = a.ops[g].kind, q0 = a.ops[g].q0, q1 = a.ops[g].q1;
                const double2 cs = tab[sidx * CH + (g - gc)];
                const double c = cs.x, s = cs.y;
                const int bit = 1 << q0;
                if (kind == VQA_GATE_RZ || kind == VQA_GATE_Z) {
                    for (int k = p; k < D; k += TPS) {
                        double2 v = st[k];
                        if (kind == VQA_GATE_Z) {
                            if (k & bit) v = make_double2(-v.x, -v.y);
                        } else {
                            const double ss = (k & bit) ? -s : s;  // e^{+ia/2} on |0>, e^{-ia/2} on |1>
                            v = make_double2(c * v.x - ss * v.y, c * v.y + ss * v.x);
                        }
                        st[k] = v;
                    }
                } else if (kind == VQA_GATE_CZ) {
                    const int both = bit | (1 << q1);
                    for (int k = p; k < D; k += TPS) {
                        if ((k & both) == both) {
                            double2 v = st[k];
                            st[k] = make_double2(-v.x, -v.y);
                        }
                    }
                } else {
                    const int cbit = kind == VQA_GATE_CNOT ? (1 << q1) : 0;
                    for (int pi = p; pi < PP; pi += TPS) {
                        const int i0 = ((pi >> q0) << (q0 + 1)) | (pi & (bit - 1));
                        const int i1 = i0 | bit;
                        const double2 a0 = st[i0], a1 = st[i1];
                        double2 r0, r1;
                        switch (kind) {
                        case VQA_GATE_H: {
                            const double r2 = 0.70710678118654752440;
                            r0 = make_double2(r2 * (a0.x + a1.x), r2 * (a0.y + a1.y));
                            r1 = make_double2(r2 * (a0.x - a1.x), r2 * (a0.y - a1.y));
                            break;
                        }
                        case VQA_GATE_X: r0 = a1; r1 = a0; break;
                        case VQA_GATE_Y:  // [[0,-i],[i,0]]
                            r0 = make_double2(a1.y, -a1.x);
                            r1 = make_double2(-a0.y, a0.x);
                            break;
                        case VQA_GATE_RX:  // [[c, is],[is, c]]
                            r0 = make_double2(c * a0.x - s * a1.y, c * a0.y + s * a1.x);
                            r1 = make_double2(c * a1.x - s * a0.y, c * a1.y + s * a0.x);
                            break;
                        case VQA_GATE_RY:  // [[c, s],[-s, c]]
                            r0 = make_double2(c * a0.x + s * a1.x, c * a0.y + s * a1.y);
                            r1 = make_double2(c * a1.x - s * a0.x, c * a1.y - s * a0.y);
                            break;
                        case VQA_GATE_CNOT:
                            if (i0 & cbit) { r0 = a1; r1 = a0; } else { r0 = a0; r1 = a1; }
                            break;
                        default: r0 = a0; r1 = a1; break;
                        }
                        st[i0] = r0;
                        st[i1] = r1;
                    }
                }
                __syncthreads();
            }
        }
        __syncthreads();

        // ---- measurement ----
        for (int o = 0; o < a.obs.n_obs; ++o) {
            double acc = 0.0;
            for (int t = a.obs.off[o]; t < a.obs.off[o + 1]; ++t) {
                const uint64_t z = a.obs.z[t], x = a.obs.x[t];
                const int ny = __popcll(x & z) & 3;
                double part = 0.0;
                for (int k = p; k < D; k += TPS) {
                    const double2 v = st[k];
                    double val;
                    if (x == 0) {
                        val = v.x * v.x + v.y * v.y;
                    } else {
                        const double2 w = st[k ^ (int)x];
                        const double pr = w.x * v.x + w.y * v.y;  // Re conj(w) v
                        const double pim = w.x * v.y - w.y * v.x; // Im conj(w) v
                        val = ny == 0 ? pr : ny == 1 ? -pim : ny == 2 ? -pr : pim;
                    }
                    part += (__popcll((uint64_t)k & z) & 1) ? -val : val;
                }
                acc += a.obs.coef[t] * part;
            }
            const double tot = seg_reduce(acc, red, tid, p, TPS);
            if (p == 0 && live) a.out_expect[e * a.obs.n_obs + o] = valid ? tot : nanv;
        }
        if (a.n_marg_bits > 0) {
            const int M = 1 << a.n_marg_bits;
            for (int m = 0; m < M; ++m) {
                double part = 0.0;
                for (int k = p; k < D; k += TPS) {
                    if ((k & (M - 1)) == m) {
                        const double2 v = st[k];
                        part += v.x * v.x + v.y * v.y;
                    }
                }
                const double tot = seg_reduce(part, red, tid, p, TPS);
                if (p == 0 && live) a.out_marginal[e * M + m] = valid ? tot : nanv;
            }
        }
        __syncthreads();
    }
}

}  // namespace

int lds_engine_max_qubits() { return 13; }

int lds_engine_run(const EvalCall& call) {
    const int n = call.c->n_qubits;
    if (n > lds_engine_max_qubits())
        VQA_FAIL(VQA_EUNSUPPORTED, "LDS engine holds at most %d qubits (got %d)",
                 lds_engine_max_qubits(), n);
    if (call.n_marg_bits > 10)
        VQA_FAIL(VQA_EUNSUPPORTED, "LDS engine: n_marg_bits <= 10 (got %d)", call.n_marg_bits);
    if (call.E <= 0) return VQA_OK;
    const int D = 1 << n;
    const int PP = D >> 1 ? D >> 1 : 1;
    const int S = PP >= kThreads ? 1 : kThreads / PP;
    const size_t lds = ((size_t)S * D + kThreads) * sizeof(double2) + kThreads * sizeof(double);
    LdsArgs a;
    a.n = n;
    a.n_ops = call.c->n_ops;
    a.n_input = call.c->n_input;
    a.n_marg_bits = call.n_marg_bits;
    a.ops = call.d_ops;
    a.theta = call.d_theta;
    a.input = call.d_input;
    a.spec = call.spec;
    a.E = call.E;
    a.obs = call.obs;
    a.out_expect = call.d_out_expect;
    a.out_marginal = call.d_out_marginal;
    VQA_HIP(hipFuncSetAttribute((const void*)k_lds_eval, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    const int64_t groups = (call.E + S - 1) / S;
    const int grid = (int)std::min<int64_t>(groups, 256 * 8);
    hipLaunchKernelGGL(k_lds_eval, dim3(grid), dim3(kThreads), lds, call.stream, a);
    VQA_HIP(hipGetLastError());
    return VQA_OK;
}

}  // namespace vqa
