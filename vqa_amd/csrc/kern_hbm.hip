// Engine 3: HBM-streaming statevector kernels (any n; used for n >= 14).
//
// States live in a library-owned HBM workspace, C instances at a time
// (C * 2^n * 16 bytes).  Every gate of the submitted list is one coalesced
// sweep over the chunk: dense 1-qubit gates as amplitude-pair sweeps, RZ / CZ /
// Z as in-place phase or sign sweeps, CNOT as a conditional pair swap.  The
// observables are reduced in two deterministic stages (per-block partials,
// then one finishing block per instance).
//
// Replaces, per instance, LearningCircuit.run(x) + Observable expectation
// (vqa/QNN/myQNNs_base.py:692-704, 907-914) for registers that do not fit a
// workgroup.
#include "vqa_core.h"

namespace vqa {

namespace {

constexpr int kT = 256;
constexpr int kZBatch = 16;  // Z-string terms (or marginal bins) reduced per pass

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

struct ZBatch {
    int nz;
    uint64_t z[kZBatch];
    double coef[kZBatch];
    int marg_bits;  // > 0: bins m0 .. m0+nz-1 of the low-bit marginal instead of Z-strings
    int m0;
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

// stage 1: partial[(ci * nb + blk) * kZBatch + t] = sum over the block's slice
__global__ __launch_bounds__(kT) void k_zpart(const double2* st, int n, ZBatch zb, double* partial) {
    __shared__ double red[4];
    const int64_t D = (int64_t)1 << n;
    const int ci = blockIdx.y, nb = gridDim.x;
    const double2* s = st + ((int64_t)ci << n);
    double acc[kZBatch];
#pragma unroll
    for (int t = 0; t < kZBatch; ++t) acc[t] = 0.0;
    const int M1 = (1 << zb.marg_bits) - 1;
    for (int64_t k = (int64_t)blockIdx.x * kT + threadIdx.x; k < D; k += (int64_t)nb * kT) {
        const double2 v = s[k];
        const double pr = v.x * v.x + v.y * v.y;
#pragma unroll
        for (int t = 0; t < kZBatch; ++t) {
            if (t < zb.nz) {
                if (zb.marg_bits > 0)
                    acc[t] += ((int)(k & M1) == zb.m0 + t) ? pr : 0.0;
                else
                    acc[t] += (__popcll((uint64_t)k & zb.z[t]) & 1) ? -pr : pr;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < kZBatch; ++t) {
        if (t < zb.nz) {
            const double r = block_sum(acc[t], red);
            if (threadIdx.x == 0) partial[((int64_t)ci * nb + blockIdx.x) * kZBatch + t] = r;
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
__global__ __launch_bounds__(kT) void k_finish(const double* partial, int nb, int nz, ZBatch zb,
                                               int use_coef, double xcoef, double* out,
                                               int64_t e_base, int stride, int col0, int spread,
                                               int accumulate, EvalSpec spec) {
    __shared__ double red[4];
    const int ci = blockIdx.x;
    double total = 0.0;
    bool valid;
    {
        int b_, sp_;
        double sv_;
        valid = decode_eval(spec, e_base + ci, b_, sp_, sv_);
    }
    const double nanv = __longlong_as_double(0x7ff8000000000000LL);
    for (int t = 0; t < nz; ++t) {
        double v = 0.0;
        for (int b = threadIdx.x; b < nb; b += kT) v += partial[((int64_t)ci * nb + b) * kZBatch + t];
        const double r = block_sum(v, red);
        if (threadIdx.x == 0) {
            if (spread) {  // one output column per t (marginal bins)
                out[(e_base + ci) * stride + col0 + t] = valid ? r : nanv;
            } else {
                total += (use_coef ? zb.coef[t] : xcoef) * r;
            }
        }
    }
    if (!spread && threadIdx.x == 0) {
        double* o = &out[(e_base + ci) * stride + col0];
        *o = !valid ? nanv : accumulate ? *o + total : total;
    }
}

int launch_gates(const EvalCall& call, const HbmCommon& h, double2* st, double2* tab) {
    const int n = h.n, C = h.C;
    const int64_t D = (int64_t)1 << n;
    const int64_t tot = D * C;
    hipLaunchKernelGGL(k_init, dim3((unsigned)((tot + kT - 1) / kT)), dim3(kT), 0, call.stream, st,
                       tot, D);
    const int64_t na = (int64_t)C * h.n_ops;
    if (na > 0)
        hipLaunchKernelGGL(k_angles, dim3((unsigned)((na + kT - 1) / kT)), dim3(kT), 0, call.stream,
                           h, tab);
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

}  // namespace

static int chunk_plan(const EvalCall& call, int64_t& Cmax, int& nb) {
    const int n = call.c->n_qubits;
    const int64_t D = (int64_t)1 << n;
    size_t free_b = 0, total_b = 0;
    VQA_HIP(hipMemGetInfo(&free_b, &total_b));
    // state chunk budget: at most 8 GiB and at most 40 % of what is free now
    size_t budget = std::min<size_t>((size_t)8 << 30, (size_t)(0.4 * (double)free_b));
    Cmax = (int64_t)(budget / ((size_t)D * sizeof(double2)));
    if (Cmax < 1) VQA_FAIL(VQA_ENOMEM, "not enough free device memory for one %d-qubit state", n);
    Cmax = std::min<int64_t>(Cmax, 4096);
    nb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, D / (kT * 4)));
    return VQA_OK;
}

int hbm_engine_run(const EvalCall& call) {
    const int n = call.c->n_qubits;
    if (n < 1 || n > 30) VQA_FAIL(VQA_EUNSUPPORTED, "HBM engine supports 1..30 qubits (got %d)", n);
    if (call.n_marg_bits > 10)
        VQA_FAIL(VQA_EUNSUPPORTED, "HBM engine: n_marg_bits <= 10 (got %d)", call.n_marg_bits);
    if (call.E <= 0) return VQA_OK;
    const int64_t D = (int64_t)1 << n;
    int64_t Cmax;
    int nb;
    int rc = chunk_plan(call, Cmax, nb);
    if (rc) return rc;
    Cmax = std::min<int64_t>(Cmax, call.E);
    const size_t st_bytes = (size_t)Cmax * D * sizeof(double2);
    const size_t tab_bytes = ((size_t)Cmax * std::max(call.c->n_ops, 1) * sizeof(double2) + 255) & ~(size_t)255;
    const size_t part_bytes = (size_t)Cmax * nb * kZBatch * sizeof(double);
    void* ws = nullptr;
    rc = workspace_get(call.device, st_bytes + tab_bytes + part_bytes, &ws);
    if (rc) return rc;
    double2* st = (double2*)ws;
    double2* tab = (double2*)((char*)ws + st_bytes);
    double* partial = (double*)((char*)ws + st_bytes + tab_bytes);

    const int n_obs = call.obs.n_obs;
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
        rc = launch_gates(call, h, st, tab);
        if (rc) return rc;
        // observables
        for (int o = 0; o < n_obs; ++o) {
            bool first = true;
            ZBatch zb;
            zb.marg_bits = 0;
            zb.m0 = 0;
            zb.nz = 0;
            auto flush = [&]() {
                if (zb.nz == 0) return;
                hipLaunchKernelGGL(k_zpart, dim3(nb, h.C), dim3(kT), 0, call.stream, st, n, zb,
                                   partial);
                hipLaunchKernelGGL(k_finish, dim3(h.C), dim3(kT), 0, call.stream, partial, nb,
                                   zb.nz, zb, 1, 0.0, call.d_out_expect, e0, n_obs, o, 0,
                                   first ? 0 : 1, call.spec);
                first = false;
                zb.nz = 0;
            };
            for (int t = call.obs.off[o]; t < call.obs.off[o + 1]; ++t) {
                if (call.obs.x[t] == 0) {
                    zb.z[zb.nz] = call.obs.z[t];
                    zb.coef[zb.nz] = call.obs.coef[t];
                    if (++zb.nz == kZBatch) flush();
                }
            }
            flush();
            for (int t = call.obs.off[o]; t < call.obs.off[o + 1]; ++t) {
                if (call.obs.x[t] != 0) {
                    hipLaunchKernelGGL(k_xpart, dim3(nb, h.C), dim3(kT), 0, call.stream, st, n,
                                       call.obs.z[t], call.obs.x[t], partial);
                    hipLaunchKernelGGL(k_finish, dim3(h.C), dim3(kT), 0, call.stream, partial, nb,
                                       1, zb, 0, call.obs.coef[t], call.d_out_expect, e0, n_obs, o,
                                       0, first ? 0 : 1, call.spec);
                    first = false;
                }
            }
            if (first) {  // observable without terms: expectation 0
                ZBatch z0;
                z0.nz = 0; z0.marg_bits = 0; z0.m0 = 0;
                hipLaunchKernelGGL(k_finish, dim3(h.C), dim3(kT), 0, call.stream, partial, nb, 0,
                                   z0, 1, 0.0, call.d_out_expect, e0, n_obs, o, 0, 0, call.spec);
            }
        }
        if (call.n_marg_bits > 0) {
            const int M = 1 << call.n_marg_bits;
            for (int m0 = 0; m0 < M; m0 += kZBatch) {
                ZBatch zb;
                zb.marg_bits = call.n_marg_bits;
                zb.m0 = m0;
                zb.nz = std::min(kZBatch, M - m0);
                hipLaunchKernelGGL(k_zpart, dim3(nb, h.C), dim3(kT), 0, call.stream, st, n, zb,
                                   partial);
                hipLaunchKernelGGL(k_finish, dim3(h.C), dim3(kT), 0, call.stream, partial, nb,
                                   zb.nz, zb, 0, 0.0, call.d_out_marginal, e0, M, m0, 1, 0, call.spec);
            }
        }
        VQA_HIP(hipGetLastError());
    }
    return VQA_OK;
}

int hbm_engine_statevector(const EvalCall& call, double* d_out_state) {
    const int n = call.c->n_qubits;
    if (n < 1 || n > 30) VQA_FAIL(VQA_EUNSUPPORTED, "HBM engine supports 1..30 qubits (got %d)", n);
    const size_t tab_bytes = ((size_t)std::max(call.c->n_ops, 1) * sizeof(double2) + 255) & ~(size_t)255;
    void* ws = nullptr;
    int rc = workspace_get(call.device, tab_bytes, &ws);
    if (rc) return rc;
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
    return launch_gates(call, h, (double2*)d_out_state, (double2*)ws);
}

}  // namespace vqa
