// Host-side circuit compiler of engine 2: gate list -> normal form of
// alternating U-/D-stages -> schedule of register / lane / LDS-transposition
// operations (format: reg_program.h).
//
// The gate lists it is tuned for are the reference's ansaetze
// (vqa/QNN/QML_utils.py:33-164): H layers, ZZ feature maps written as
// CNOT-RZ-CNOT sandwiches, RX/RY/RZ layers and CZ rings; any list of the
// supported gate kinds compiles.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <random>
#include <string>
#include <utility>

#include <stdio.h>

#include "reg_program.h"

namespace vqa {

namespace {

struct Stage {
    bool isU = true;
    // U: qubit -> elementary gates in time order
    std::map<int, std::vector<ElemGate>> g;
    // D: single-qubit RZ terms, ZZ terms, CZ pairs, Z gates
    std::map<int, std::vector<ElemGate>> rz;
    std::map<std::pair<int, int>, std::vector<ElemGate>> zz;
    std::vector<std::pair<int, int>> cz;
    std::vector<int> z;

    bool touches(int q) const {
        if (rz.count(q)) return true;
        for (int zq : z) if (zq == q) return true;
        for (auto& kv : zz) if (kv.first.first == q || kv.first.second == q) return true;
        for (auto& p : cz) if (p.first == q || p.second == q) return true;
        return false;
    }
};

ElemGate elem_of(const vqa_op& op) {
    ElemGate e;
    e.kind = op.kind;
    e.src = (op.kind >= VQA_GATE_RX && op.kind <= VQA_GATE_RZ) ? op.src : VQA_SRC_CONST;
    e.idx = op.idx;
    e.pad = 0;
    e.coef = op.coef;
    e.offset = op.offset;
    return e;
}

ElemGate fixed_gate(int kind) {
    ElemGate e;
    memset(&e, 0, sizeof(e));
    e.kind = kind;
    e.src = VQA_SRC_CONST;
    return e;
}

void add_1q(std::vector<Stage>& st, int q, const ElemGate& e) {
    const bool diag = e.kind == VQA_GATE_RZ || e.kind == VQA_GATE_Z;
    if (!st.empty() && st.back().isU) {
        st.back().g[q].push_back(e);
        return;
    }
    if (!st.empty()) {  // last stage is a D-stage
        Stage& d = st.back();
        if (diag) {
            if (e.kind == VQA_GATE_RZ) d.rz[q].push_back(e);
            else d.z.push_back(q);
            return;
        }
        if (!d.touches(q) && st.size() >= 2) {  // commutes with the D-stage: join the U-stage before it
            st[st.size() - 2].g[q].push_back(e);
            return;
        }
    }
    Stage u;
    u.isU = true;
    u.g[q].push_back(e);
    st.push_back(u);
}

Stage& last_d(std::vector<Stage>& st) {
    if (st.empty() || st.back().isU) {
        Stage d;
        d.isU = false;
        st.push_back(d);
    }
    return st.back();
}

std::vector<Stage> normal_form(const std::vector<vqa_op>& ops) {
    std::vector<Stage> st;
    const size_t N = ops.size();
    for (size_t i = 0; i < N; ++i) {
        const vqa_op& o = ops[i];
        if (o.kind == VQA_GATE_CNOT) {
            // CNOT(c,t) RZ_t(a) CNOT(c,t) = exp(i a/2 Z_c Z_t)
            if (i + 2 < N && ops[i + 1].kind == VQA_GATE_RZ && ops[i + 1].q0 == o.q0 &&
                ops[i + 2].kind == VQA_GATE_CNOT && ops[i + 2].q0 == o.q0 && ops[i + 2].q1 == o.q1) {
                Stage& d = last_d(st);
                d.zz[std::minmax(o.q0, o.q1)].push_back(elem_of(ops[i + 1]));
                i += 2;
                continue;
            }
            // lone CNOT = H_t CZ(c,t) H_t
            add_1q(st, o.q0, fixed_gate(VQA_GATE_H));
            last_d(st).cz.push_back({o.q0, o.q1});
            add_1q(st, o.q0, fixed_gate(VQA_GATE_H));
        } else if (o.kind == VQA_GATE_CZ) {
            last_d(st).cz.push_back({o.q0, o.q1});
        } else {
            add_1q(st, o.q0, elem_of(o));
        }
    }
    // a leading D-stage acts on |0...0> (global phase); a trailing one does not
    // change Z-basis probabilities, which is all this engine measures
    if (!st.empty() && !st.front().isU) st.erase(st.begin());
    if (!st.empty() && !st.back().isU) st.pop_back();
    // A trainable RZ at the end of a qubit's run commutes with the diagonal stage behind it: it opens the qubit's
    // run of the NEXT U-stage instead (RZ RY RX | CZ ring | RZ RY RX  ->  RY RX | CZ ring | RZ RY RX RZ).  The
    // fused 2x2s stay one per qubit and stage; a parameter-shift run of that parameter then resumes one diagonal
    // stage and one U-stage later (a third of the hardware-efficient ansatz' parameters).  VQA_NO_RZ_FORWARD=1: as listed.
    if (!getenv("VQA_NO_RZ_FORWARD")) {
        for (size_t i = 0; i + 2 < st.size(); ++i) {
            if (!st[i].isU || st[i + 1].isU || !st[i + 2].isU) continue;
            for (auto it = st[i].g.begin(); it != st[i].g.end();) {
                std::vector<ElemGate>& run = it->second;
                auto nx = st[i + 2].g.find(it->first);
                size_t moved = 0;
                while (nx != st[i + 2].g.end() && run.size() > 1 && run.back().kind == VQA_GATE_RZ && run.back().src == VQA_SRC_THETA) {
                    nx->second.insert(nx->second.begin(), run.back());
                    run.pop_back();
                    ++moved;
                }
                (void)moved;
                ++it;
            }
        }
    }
    return st;
}

// ---- GF(2) helpers -------------------------------------------------------------
int gf2_rank(std::vector<uint32_t> rows) {
    int rank = 0;
    for (int bit = 0; bit < 32; ++bit) {
        int piv = -1;
        for (size_t i = rank; i < rows.size(); ++i)
            if (rows[i] >> bit & 1) { piv = (int)i; break; }
        if (piv < 0) continue;
        std::swap(rows[rank], rows[piv]);
        for (size_t i = 0; i < rows.size(); ++i)
            if ((int)i != rank && (rows[i] >> bit & 1)) rows[i] ^= rows[rank];
        ++rank;
    }
    return rank;
}

inline int par(uint32_t x) { return __builtin_popcount(x) & 1; }

// Find an invertible XOR-linear slot function (masks f[k]: slot bit k =
// parity(index & f[k])) whose low bits are conflict-free for the 16-byte LDS
// writes in the old layout (8 consecutive lanes -> 8 distinct slots mod 8) and
// for the 16-byte reads in the new layout (the 16-lane read groups of the LDS
// -> 16 distinct slots mod 16).  sigma[p'] = old position read by new position p'.
// Register bit R-1 never moves (sigma[R-1] = R-1).
std::vector<uint32_t> find_swizzle(int n, const int* sigma) {
    const int R = kRegBits;
    const uint32_t w[3] = { 1u << (R + 0), 1u << (R + 1), 1u << (R + 2) };
    const uint32_t v[4] = { 1u << sigma[R + 0], 1u << sigma[R + 1],
                            (1u << sigma[R + 2]) ^ (1u << sigma[R + 3]),
                            (1u << sigma[R + 2]) ^ (1u << sigma[R + 4]) };
    uint32_t involved = 0;
    for (auto x : w) involved |= x;
    for (auto x : v) involved |= x;
    std::vector<int> bits;
    for (int p = 0; p < n; ++p) if (involved >> p & 1) bits.push_back(p);
    std::mt19937 rng(12345);
    std::vector<uint32_t> f(4, 0);
    bool ok = false;
    for (int attempt = 0; attempt < 200000 && !ok; ++attempt) {
        for (int k = 0; k < 4; ++k) {
            uint32_t m = 0;
            const int cnt = 1 + (int)(rng() % 3);
            for (int c = 0; c < cnt; ++c) m |= 1u << bits[rng() % bits.size()];
            f[k] = m;
        }
        std::vector<uint32_t> mw(3), mr(4);
        for (int i = 0; i < 3; ++i) { mw[i] = 0; for (int k = 0; k < 3; ++k) mw[i] |= (uint32_t)par(w[i] & f[k]) << k; }
        for (int i = 0; i < 4; ++i) { mr[i] = 0; for (int k = 0; k < 4; ++k) mr[i] |= (uint32_t)par(v[i] & f[k]) << k; }
        if (gf2_rank(mw) == 3 && gf2_rank(mr) == 4 && gf2_rank(f) == 4) ok = true;
    }
    if (!ok) {  // correct but possibly conflicting: identity slots, register bit R-1 on top
        std::vector<uint32_t> id;
        for (int p = 0; p < n; ++p) if (p != R - 1) id.push_back(1u << p);
        id.push_back(1u << (R - 1));
        return id;
    }
    // complete to an invertible map with unit vectors; register bit R-1 comes
    // last, so it alone makes the top slot bit (the two halves of the state
    // then use the same half-size buffer one after the other)
    for (int i = 0; i < n && (int)f.size() < n; ++i) {
        const int p = i < R - 1 ? i : (i < n - 1 ? i + 1 : R - 1);
        std::vector<uint32_t> t = f;
        t.push_back(1u << p);
        if (gf2_rank(t) == (int)t.size()) f = t;
    }
    return f;
}

template <class T>
int64_t place(std::vector<char>& blob, const std::vector<T>& v) {
    while (blob.size() % 16) blob.push_back(0);
    const int64_t off = (int64_t)blob.size();
    const char* p = (const char*)v.data();
    blob.insert(blob.end(), p, p + v.size() * sizeof(T));
    return off;
}

struct Layout {
    int q_at[kMaxPos], pos_of[kMaxPos];
};

}  // namespace

bool reg_compile(int n, int n_theta, const std::vector<vqa_op>& gate_list, RegProgram* out,
                 std::string* why) {
    const int R = kRegBits;
    if (n < R) { *why = "fewer qubits than register bits"; return false; }
    const int T = n - R;
    if (T > kMaxThreadBits) { *why = "too many qubits for one workgroup"; return false; }
    const int L = std::min(T, 6), W = T - L;
    RegProgram& P = *out;
    P = RegProgram();
    RegHeader& H = P.hdr;
    memset(&H, 0, sizeof(H));
    H.n = n; H.T = T; H.L = L; H.W = W;
    H.block = std::max(64, 1 << T);
    H.S = H.block >> T;

    std::vector<Stage> st = normal_form(gate_list);
    // strict alternation U D U ... U is guaranteed by construction
    std::vector<int> ustages, dstages;
    for (size_t i = 0; i < st.size(); ++i) (st[i].isU ? ustages : dstages).push_back((int)i);
    const int NU = (int)ustages.size(), ND = (int)dstages.size();
    if (NU > 0 && ND != NU - 1) { *why = "internal: stages do not alternate"; return false; }

    // ---- U-entries (tables indexed by entry; phase slots are assigned after scheduling) ----
    std::vector<std::map<int, int>> uidx(NU);  // U-stage -> qubit -> U-entry index
    for (int u = 0; u < NU; ++u) {
        for (auto& kv : st[ustages[u]].g) {
            UEntry e;
            memset(&e, 0, sizeof(e));
            e.begin = (int)P.elems.size();
            e.count = (int)kv.second.size();
            P.elems.insert(P.elems.end(), kv.second.begin(), kv.second.end());
            e.d_before = e.d_after = -1;
            e.parity = u & 1;
            e.input_free = 1;
            for (const ElemGate& g : kv.second)
                if (g.src == VQA_SRC_INPUT) e.input_free = 0;
            if (!e.input_free) P.dep.push_back((int32_t)P.uent.size());
            uidx[u][kv.first] = (int)P.uent.size();
            P.uent.push_back(e);
            ++P.n_dense_gates;
        }
    }

    // ---- schedule: ops + the layout every D-stage is applied in ----
    // (run twice: dry, for every candidate of the initial layout, to see the layouts the diagonal stages
    // would get; then for the chosen one)
    Layout lay;
    std::vector<Layout> dlayout(ND);
    bool dry = false;
    bool sched_failed = false;
    auto emit = [&](int kind, int a, int b) { if (!dry) P.ops.push_back(RegOp{ kind, a, b, 0 }); };
    auto emit_direct = [&](int q, int ue) {
        const int p = lay.pos_of[q];
        if (dry) return;
        // 1: rotated through DPP partner fetches; 2: lane bits 4 / 5, rotated by the matrix unit
        P.uent[ue].on_lane = p < R ? 0 : (p - R >= 4 ? 2 : 1);
        if (p < R) emit(ROP_UREG, p, ue);
        else emit(ROP_ULANE, p - R, ue);
    };
    // VQA_REG_SCHED (environment, experiments): 1 = rotate on register bits only (more
    // transpositions, no lane rotations); default: register bits and the cheap lane bits
    const char* sched_env = getenv("VQA_REG_SCHED");
    const int sched = sched_env ? atoi(sched_env) : 0;
    // Cost classes of a position (vector-unit cycles per rotation of a 16-amplitude thread, measured):
    // register bit 216; lane bits 4 and 5 together through the matrix unit 522 for the pair; lane
    // bits 0, 1, 3 through DPP 464 each; lane bit 2 (two DPP moves per word) more; wave bits: not at all.
    // sched 0: register bits + the matrix-unit pair are "cheap"; DPP lanes only take what a
    // transposition's slots cannot hold.  2: as before the matrix unit was used (DPP lanes cheap).
    const bool mfma_pair = sched == 0 && L >= 6;
    auto cheap = [&](int p) {
        if (p < R) return true;
        if (sched == 1) return false;
        if (mfma_pair) return p == R + 4 || p == R + 5;
        return p == R + 0 || p == R + 1 || p == R + 3;
    };
    auto dpp_lane = [&](int p) { return p == R + 0 || p == R + 1 || p == R + 3; };
    // slots a transposition may bring a qubit to: every cheap position but register bit R-1,
    // which stays where it is so that a transposition can run in two halves
    const int n_swap_slots = sched == 1 ? 3 : mfma_pair ? 5 : 6;
    const int slots_dpp[6] = { 0, 1, 2, R + 0, R + 1, R + 3 };
    const int slots_mfma[6] = { 0, 1, 2, R + 4, R + 5, -1 };
    const int* cheap_slots = mfma_pair ? slots_mfma : slots_dpp;

    auto run_schedule = [&]() {
    for (int u = 0; u < NU; ++u) {
        std::vector<std::pair<int, int>> byq;  // (position, qubit)
        for (auto& kv : uidx[u]) byq.push_back({ lay.pos_of[kv.first], kv.first });
        std::sort(byq.begin(), byq.end());
        if (W == 0) {
            for (auto& pq : byq) emit_direct(pq.second, uidx[u][pq.second]);
        } else {
            std::vector<int> remaining, direct;
            for (auto& pq : byq) {
                if (cheap(pq.first)) direct.push_back(pq.second);
                else remaining.push_back(pq.second);
            }
            // what the slots of the transpositions to come cannot hold goes through DPP now, where it
            // sits on a lane bit that allows it (one DPP rotation is cheaper than another transposition)
            if (mfma_pair) {
                for (size_t i = 0; i < remaining.size() && remaining.size() % n_swap_slots != 0 && remaining.size() > (size_t)n_swap_slots;) {
                    if (dpp_lane(lay.pos_of[remaining[i]])) {
                        direct.push_back(remaining[i]);
                        remaining.erase(remaining.begin() + i);
                    } else ++i;
                }
            }
            // in position order: one phase of the kernel's loop takes them all
            std::sort(direct.begin(), direct.end(), [&](int a, int b) { return lay.pos_of[a] < lay.pos_of[b]; });
            for (int q : direct) emit_direct(q, uidx[u][q]);
            while (!remaining.empty()) {
                bool any_wave = false;
                for (int q : remaining) any_wave |= lay.pos_of[q] >= R + L;
                if (!any_wave && remaining.size() <= 1) {
                    for (int q : remaining) emit_direct(q, uidx[u][q]);
                    break;
                }
                const int nb = (int)std::min<size_t>(n_swap_slots, remaining.size());
                int old_pos_of[kMaxPos];
                memcpy(old_pos_of, lay.pos_of, sizeof(old_pos_of));
                for (int i = 0; i < nb; ++i) {
                    const int qa = remaining[i], pa = lay.pos_of[qa], pb = cheap_slots[i], qb = lay.q_at[pb];
                    lay.q_at[pa] = qb; lay.pos_of[qb] = pa;
                    lay.q_at[pb] = qa; lay.pos_of[qa] = pb;
                }
                if (dry) {
                    remaining.erase(remaining.begin(), remaining.begin() + nb);
                    continue;
                }
                int sigma[kMaxPos];
                for (int p = 0; p < n; ++p) sigma[p] = old_pos_of[lay.q_at[p]];
                std::vector<uint32_t> f = find_swizzle(n, sigma);
                XDesc xd;
                memset(&xd, 0, sizeof(xd));
                for (int p = 0; p < n; ++p) {
                    int slot = 0;
                    for (int k = 0; k < n; ++k) slot |= (int)((f[k] >> p) & 1) << k;
                    xd.wconst[p] = slot << 4;  // byte offset of a 16-byte slot
                }
                for (int p = 0; p < n; ++p) xd.rconst[p] = xd.wconst[sigma[p]];
                if (sigma[R - 1] != R - 1 || xd.wconst[R - 1] != (16 << (n - 1))) {
                    *why = "internal: transposition does not split into halves";
                    sched_failed = true;
                    return;
                }
                for (int p = 0; p < n; ++p)
                    if (p != R - 1 && (xd.wconst[p] & (16 << (n - 1)))) {
                        *why = "internal: transposition slot function mixes the halves";
                        sched_failed = true;
                        return;
                    }
                int xid = -1;
                for (size_t k = 0; k < P.xdesc.size() && xid < 0; ++k)
                    if (!memcmp(&P.xdesc[k], &xd, sizeof(xd))) xid = (int)k;
                if (xid < 0) { xid = (int)P.xdesc.size(); P.xdesc.push_back(xd); }
                emit(ROP_XPOSE, xid, 0);
                ++P.n_xpose;
                for (int i = 0; i < nb; ++i) emit_direct(remaining[i], uidx[u][remaining[i]]);
                remaining.erase(remaining.begin(), remaining.begin() + nb);
            }
        }
        if (u < NU - 1) {
            dlayout[u] = lay;
            emit(ROP_DIAG, u, 0);
            if (!dry) ++P.n_diag;
        }
    }
    };
    // The initial layout (which qubit starts at which position) is free: |0...0> looks the same in all of
    // them.  A diagonal stage is cheapest when no CZ / Z sign and no ZZ phase sits between two register
    // bits (see do_diag): the identity, the strides coprime to n and a few shuffles are tried dry, and the
    // one whose diagonal stages cost least is taken (first best: the identity when nothing is gained).
    auto stage_cost = [&]() {
        int cost = 0;
        for (int d = 0; d < ND; ++d) {
            const Stage& sd = st[dstages[d]];
            const Layout& ly = dlayout[d];
            bool rr = false;
            for (auto& pq : sd.cz) rr |= ly.pos_of[pq.first] < R && ly.pos_of[pq.second] < R;
            for (int q : sd.z) rr |= ly.pos_of[q] < R;
            cost += rr ? 2 : 0;
            for (auto& kv : sd.zz) {
                const bool a = ly.pos_of[kv.first.first] < R, b = ly.pos_of[kv.first.second] < R;
                cost += (a && b) ? 4 : (a || b) ? 1 : 0;
            }
        }
        return cost;
    };
    std::vector<std::vector<int>> candidates;
    {
        std::vector<int> id(n);
        for (int p = 0; p < n; ++p) id[p] = p;
        candidates.push_back(id);
        // (the 1024-thread workgroups apply the signs as a mask per thread, whatever the layout)
        if (n > R && H.block < 1024) {
            for (int s = 2; s < n; ++s) {
                int g = s, h = n;
                while (h) { const int t = g % h; g = h; h = t; }
                if (g != 1) continue;
                std::vector<int> c(n);
                for (int p = 0; p < n; ++p) c[p] = (p * s) % n;
                candidates.push_back(c);
            }
            uint64_t rng = 0x9E3779B97F4A7C15ull;
            for (int k = 0; k < 24; ++k) {
                std::vector<int> c = id;
                for (int p = n - 1; p > 0; --p) {
                    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                    std::swap(c[p], c[(int)((rng >> 33) % (uint64_t)(p + 1))]);
                }
                candidates.push_back(c);
            }
        }
    }
    auto set_layout = [&](const std::vector<int>& c) {
        for (int p = 0; p < kMaxPos; ++p) { lay.q_at[p] = p < n ? c[p] : -1; lay.pos_of[p] = p; }
        for (int p = 0; p < n; ++p) lay.pos_of[c[p]] = p;
    };
    size_t best = 0;
    if (candidates.size() > 1 && !getenv("VQA_REG_NO_LAYOUT_SEARCH")) {
        int best_cost = 1 << 30;
        dry = true;
        for (size_t k = 0; k < candidates.size(); ++k) {
            set_layout(candidates[k]);
            run_schedule();
            const int c = stage_cost();
            if (c < best_cost) { best_cost = c; best = k; }
            if (c == 0) break;
        }
        dry = false;
    }
    set_layout(candidates[best]);
    run_schedule();
    if (sched_failed) return false;
    if (getenv("VQA_PLAN_DEBUG"))
        fprintf(stderr, "[vqa] register program: initial layout number %zu of %zu, diagonal-stage cost %d\n", best,
                candidates.size(), stage_cost());
    emit(ROP_END, 0, 0);
    for (int p = 0; p < kMaxPos; ++p) H.final_qubit_at_pos[p] = (int8_t)(p < n ? lay.q_at[p] : -1);

    // ---- phase tables: slot = stage * n + POSITION in that stage's layout ----
    for (int u = 0; u < NU; ++u) {
        for (auto& kv : uidx[u]) {
            UEntry& e = P.uent[kv.second];
            if (u > 0) e.d_before = (u - 1) * n + dlayout[u - 1].pos_of[kv.first];
            if (u < NU - 1) e.d_after = u * n + dlayout[u].pos_of[kv.first];
        }
    }
    std::vector<std::map<std::pair<int, int>, int>> d2idx(ND);
    for (int d = 0; d < ND; ++d) {
        const Stage& s = st[dstages[d]];
        const Layout& ly = dlayout[d];
        for (int p = 0; p < n; ++p) {
            D1Entry e;
            e.begin = (int)P.terms.size();
            auto it = s.rz.find(ly.q_at[p]);
            if (it != s.rz.end()) P.terms.insert(P.terms.end(), it->second.begin(), it->second.end());
            e.count = (int)P.terms.size() - e.begin;
            P.d1.push_back(e);
        }
        for (auto& kv : s.zz) {
            D2Entry e;
            e.begin = (int)P.terms.size();
            P.terms.insert(P.terms.end(), kv.second.begin(), kv.second.end());
            e.count = (int)P.terms.size() - e.begin;
            d2idx[d][kv.first] = (int)P.d2.size();
            P.d2.push_back(e);
        }
    }

    // ---- D-stage descriptors ----
    for (int d = 0; d < ND; ++d) {
        const Stage& s = st[dstages[d]];
        const Layout& ly = dlayout[d];
        DDesc dd;
        memset(&dd, 0, sizeof(dd));
        dd.ph1_base = d * n;
        std::vector<PairRef> tt, rt, rr;
        for (auto& kv : s.zz) {
            int pa = ly.pos_of[kv.first.first], pb = ly.pos_of[kv.first.second];
            if (pa > pb) std::swap(pa, pb);
            const int id = d2idx[d][kv.first];
            if (pa >= R) tt.push_back(PairRef{ pa - R, pb - R, id, 0 });
            else if (pb >= R) rt.push_back(PairRef{ pa, pb - R, id, 0 });
            else rr.push_back(PairRef{ pa, pb, id, 0 });
        }
        dd.tt_begin = (int)P.pairs.size(); dd.tt_count = (int)tt.size();
        P.pairs.insert(P.pairs.end(), tt.begin(), tt.end());
        dd.rt_begin = (int)P.pairs.size(); dd.rt_count = (int)rt.size();
        P.pairs.insert(P.pairs.end(), rt.begin(), rt.end());
        dd.rr_begin = (int)P.pairs.size(); dd.rr_count = (int)rr.size();
        P.pairs.insert(P.pairs.end(), rr.begin(), rr.end());
        dd.sign_id = -1;
        if (!s.cz.empty() || !s.z.empty()) {
            SignTab sg;
            memset(&sg, 0, sizeof(sg));
            for (int t = 0; t < (1 << T); ++t) {
                int base = 0, cr[kRegBits] = { 0, 0, 0, 0 };
                for (auto& p : s.cz) {
                    int pa = ly.pos_of[p.first], pb = ly.pos_of[p.second];
                    if (pa > pb) std::swap(pa, pb);
                    if (pa >= R) base ^= (t >> (pa - R)) & (t >> (pb - R)) & 1;
                    else if (pb >= R) cr[pa] ^= (t >> (pb - R)) & 1;
                }
                for (int q : s.z) {
                    const int p = ly.pos_of[q];
                    if (p >= R) base ^= (t >> (p - R)) & 1;
                }
                if (base) sg.tt_par[t >> 5] |= 1u << (t & 31);
                for (int j = 0; j < R; ++j) if (cr[j]) sg.cross[j][t >> 5] |= 1u << (t & 31);
            }
            for (int r = 0; r < kRegAmps; ++r) {
                int b = 0;
                for (auto& p : s.cz) {
                    const int pa = ly.pos_of[p.first], pb = ly.pos_of[p.second];
                    if (pa < R && pb < R) b ^= (r >> pa) & (r >> pb) & 1;
                }
                for (int q : s.z) {
                    const int p = ly.pos_of[q];
                    if (p < R) b ^= (r >> p) & 1;
                }
                if (b) sg.rr_par |= 1u << r;
            }
            for (size_t k = 0; k < P.signs.size() && dd.sign_id < 0; ++k)
                if (!memcmp(&P.signs[k], &sg, sizeof(sg))) dd.sign_id = (int)k;
            if (dd.sign_id < 0) { dd.sign_id = (int)P.signs.size(); P.signs.push_back(sg); }
        }
        P.ddesc.push_back(dd);
    }
    // ---- pack the op stream into phases ----
    {
        PhaseOp cur;
        memset(&cur, 0, sizeof(cur));
        int last_rank = -1;
        auto flush = [&]() {
            P.phases.push_back(cur);
            memset(&cur, 0, sizeof(cur));
            last_rank = -1;
        };
        for (const RegOp& op : P.ops) {
            if (op.kind == ROP_END) break;
            int rank, flag;
            if (op.kind == ROP_XPOSE) { rank = 0; flag = PH_XPOSE; }
            else if (op.kind == ROP_UREG) { rank = 1 + op.a; flag = PH_UREG0 << op.a; }
            else if (op.kind == ROP_ULANE) { rank = 5 + op.a; flag = PH_ULANE0 << op.a; }
            else { rank = 11; flag = PH_DIAG; }
            if (rank <= last_rank) flush();
            cur.flags |= flag;
            last_rank = rank;
            if (op.kind == ROP_XPOSE) cur.xdesc = op.a;
            else if (op.kind == ROP_DIAG) { cur.ddesc = op.a; P.ddesc[op.a].pc = (int)P.phases.size(); }
            else cur.ent[rank - 1] = op.b;
        }
        if (cur.flags) flush();
        cur.flags = PH_END;
        P.phases.push_back(cur);
    }

    // ---- what each trainable parameter touches (parameter-shift sweeps) ----
    P.params.resize(n_theta);
    {
        std::vector<std::vector<PatchRef>> per(n_theta);
        std::vector<int> first_stage(n_theta, 1 << 30);  // 2u for U-stage u, 2d+1 for D-stage d
        for (int u = 0; u < NU; ++u) {
            for (auto& kv : uidx[u]) {
                const UEntry& e = P.uent[kv.second];
                std::vector<int> seen;
                for (int j = 0; j < e.count; ++j) {
                    const ElemGate& g = P.elems[e.begin + j];
                    if (g.src != VQA_SRC_THETA) continue;
                    if (std::find(seen.begin(), seen.end(), g.idx) != seen.end()) continue;
                    seen.push_back(g.idx);
                    // coef_sum: the coefficient the parameter enters its ONE gate of this entry
                    // with; 0 when it enters several (no single generator to insert then)
                    int uses = 0;
                    double coef = 0.0;
                    for (int k = 0; k < e.count; ++k) {
                        const ElemGate& h = P.elems[e.begin + k];
                        if (h.src == VQA_SRC_THETA && h.idx == g.idx) { ++uses; coef = h.coef; }
                    }
                    per[g.idx].push_back(PatchRef{ 0, kv.second, uses == 1 ? coef : 0.0 });
                    first_stage[g.idx] = std::min(first_stage[g.idx], 2 * u);
                }
            }
        }
        for (int d = 0; d < ND; ++d) {
            auto scan = [&](int kind, int index, int begin, int count) {
                std::map<int, double> sums;
                for (int j = 0; j < count; ++j) {
                    const ElemGate& g = P.terms[begin + j];
                    if (g.src == VQA_SRC_THETA) sums[g.idx] += g.coef;
                }
                for (auto& kv : sums) {
                    per[kv.first].push_back(PatchRef{ kind, index, kv.second });
                    first_stage[kv.first] = std::min(first_stage[kv.first], 2 * d + 1);
                }
            };
            for (int p = 0; p < n; ++p) scan(1, d * n + p, P.d1[d * n + p].begin, P.d1[d * n + p].count);
            for (auto& kv : d2idx[d]) scan(2, kv.second, P.d2[kv.second].begin, P.d2[kv.second].count);
        }
        for (int i = 0; i < n_theta; ++i) {
            ParamInfo& pi = P.params[i];
            pi.ref_begin = (int)P.refs.size();
            pi.ref_count = (int)per[i].size();
            P.refs.insert(P.refs.end(), per[i].begin(), per[i].end());
            // U-stage u changes the tables of D-stage u-1 (its RZ(gamma)); D-stage d its own
            int rd;
            if (first_stage[i] == (1 << 30)) rd = ND;             // parameter not used: nothing to redo
            else if (first_stage[i] & 1) rd = first_stage[i] / 2;
            else rd = first_stage[i] / 2 - 1;
            if (rd >= ND) {            // only the last U-stage (or nothing) changes
                pi.resume_diag = ND > 0 ? ND - 1 : -1;
                if (first_stage[i] != (1 << 30) && ND == 0) pi.resume_diag = -1;
            } else {
                pi.resume_diag = rd;   // -1: the first U-stage changes -> restart
            }
            pi.resume_pc = pi.resume_diag >= 0 ? P.ddesc[pi.resume_diag].pc : 0;
            pi.late_diag = -1;
            pi.late_pc = 0;
            pi.late_mask = 0;
            pi.pad = 0;
        }
        // resumption behind the parameter's own U-stage (see ParamInfo); behind the last U-stage there is no
        // diagonal stage: late_diag = ND names the final state, late_pc the closing phase (grouped sweeps)
        for (int u = 0; u < NU; ++u) {
            const bool last = u + 1 == NU;
            for (auto& kv : uidx[u]) {
                const int pos = last ? lay.pos_of[kv.first] : dlayout[u].pos_of[kv.first];
                // (a wave bit: the partner amplitudes come through the transposition buffer, when the program has one)
                if (pos >= R + L && P.n_xpose == 0) continue;
                for (int i = 0; i < n_theta; ++i) {
                    ParamInfo& pi = P.params[i];
                    if (pi.ref_count != 1) continue;
                    const PatchRef& ref = P.refs[pi.ref_begin];
                    if (ref.kind != 0 || ref.index != kv.second || ref.coef_sum == 0.0) continue;
                    pi.late_diag = last ? ND : u;
                    pi.late_pc = last ? (int)P.phases.size() - 1 : P.ddesc[u].pc;
                    pi.late_mask = 1 << pos;
                }
            }
        }
    }

    // runs of a sweep: one per parameter where both shifts can come from one run
    // (the parameter drives exactly one rotation of a fused 2x2 whose trailing phase
    // goes into a later diagonal stage, so that both runs end in the same frame),
    // otherwise one per shift
    for (int i = 0; i < n_theta; ++i) {
        const ParamInfo& pi = P.params[i];
        bool pair = false;
        if (pi.ref_count == 1) {
            const PatchRef& ref = P.refs[pi.ref_begin];
            pair = ref.kind == 0 && ref.coef_sum != 0.0 && P.uent[ref.index].d_after >= 0;
        }
        if (pair) P.runs.push_back(RunInfo{ i, 0 });
        else { P.runs.push_back(RunInfo{ i, 1 }); P.runs.push_back(RunInfo{ i, 2 }); }
    }
    H.n_runs = (int)P.runs.size();

    if (getenv("VQA_PLAN_DEBUG") && atoi(getenv("VQA_PLAN_DEBUG")) >= 2) {
        // the phases: x = transposition, rJ / lJ = rotation on register / lane bit J, D = diagonal stage
        for (size_t k = 0; k < P.phases.size(); ++k) {
            const PhaseOp& ph = P.phases[k];
            fprintf(stderr, "[vqa]   phase %2zu:", k);
            if (ph.flags & PH_XPOSE) fprintf(stderr, " x%d", ph.xdesc);
            for (int j = 0; j < 4; ++j) if (ph.flags & (PH_UREG0 << j)) fprintf(stderr, " r%d", j);
            for (int j = 0; j < 6; ++j) if (ph.flags & (PH_ULANE0 << j)) fprintf(stderr, " l%d", j);
            if (ph.flags & PH_DIAG) fprintf(stderr, " D%d", ph.ddesc);
            if (ph.flags & PH_END) fprintf(stderr, " end");
            fprintf(stderr, "\n");
        }
    }
    H.n_uent = (int)P.uent.size();
    H.n_d1 = (int)P.d1.size();
    H.n_d2 = (int)P.d2.size();
    H.n_elems = (int)P.elems.size();
    H.n_terms = (int)P.terms.size();
    H.n_ops = (int)P.ops.size();
    H.n_ddesc = (int)P.ddesc.size();
    H.n_xdesc = (int)P.xdesc.size();
    H.n_pairs = (int)P.pairs.size();
    H.n_sign = (int)P.signs.size();
    H.n_params = n_theta;
    H.n_refs = (int)P.refs.size();
    H.n_phases = (int)P.phases.size();
    H.has_xpose = P.n_xpose > 0;
    if (getenv("VQA_PLAN_DEBUG"))
        fprintf(stderr, "[vqa] register program: %d qubits, %d U-stages, %d phases, %d transpositions (%d distinct), %d sign tables\n",
                n, NU, H.n_phases, P.n_xpose, H.n_xdesc, H.n_sign);

    const size_t tab1 = ((size_t)H.n_uent + H.n_d1 + H.n_d2) * 16;
    const size_t xbuf = H.has_xpose ? ((size_t)8 << n) : 0;  // half of the state at a time
    const size_t fixed = 128 * sizeof(double) + (((size_t)H.n_sign * H.block * 2 + 15) & ~(size_t)15);
    H.n_dep = (int)P.dep.size();
    // (explicit-list kernel: + the launch-wide cache of the input-free fused 2x2s, three entries each)
    P.lds_bytes = (size_t)H.S * tab1 + xbuf + fixed + (size_t)H.n_uent * 48;
    // + run record, unshifted values, the diagonal stages' thread-factor tables (two per stage, by halves of the thread bits)
    // (the 1024-thread workgroups keep the diagonal stages' earlier form and have no such tables)
    P.lds_bytes_ftab = H.block >= 1024 ? 0 : (size_t)H.n_ddesc * ((1 << (T >> 1)) + (1 << (T - (T >> 1))) + kRegBits) * 16 + 16;
    P.lds_bytes_psr = tab1 + xbuf + fixed + 16 * 16 + 80 * sizeof(double) + P.lds_bytes_ftab;
    // (64 bytes of margin for the kernels' few statically allocated words)
    if (getenv("VQA_PLAN_DEBUG"))
        fprintf(stderr, "[vqa] register program: LDS %zu bytes (explicit lists), %zu (sweeps): tables %zu, transposition buffer %zu, sign masks %zu\n",
                P.lds_bytes, P.lds_bytes_psr, tab1, xbuf, (size_t)H.n_sign * H.block * 2);
    if (std::max(P.lds_bytes, P.lds_bytes_psr) + 64 > 160 * 1024) { *why = "per-evaluation tables do not fit LDS"; return false; }

    // ---- device blob ----
    P.blob.assign(sizeof(RegHeader), 0);
    H.off_elems = place(P.blob, P.elems);
    H.off_uent = place(P.blob, P.uent);
    H.off_d1 = place(P.blob, P.d1);
    H.off_d2 = place(P.blob, P.d2);
    H.off_terms = place(P.blob, P.terms);
    H.off_ops = place(P.blob, P.ops);
    H.off_ddesc = place(P.blob, P.ddesc);
    H.off_xdesc = place(P.blob, P.xdesc);
    H.off_pairs = place(P.blob, P.pairs);
    H.off_signs = place(P.blob, P.signs);
    H.off_params = place(P.blob, P.params);
    H.off_refs = place(P.blob, P.refs);
    while (P.blob.size() % 64) P.blob.push_back(0);   // a phase descriptor is one 64-byte scalar load
    H.off_phases = place(P.blob, P.phases);
    H.off_runs = place(P.blob, P.runs);
    H.off_dep = place(P.blob, P.dep);
    while (P.blob.size() % 16) P.blob.push_back(0);
    H.total_bytes = (int64_t)P.blob.size();
    memcpy(P.blob.data(), &H, sizeof(H));
    return true;
}

}  // namespace vqa
