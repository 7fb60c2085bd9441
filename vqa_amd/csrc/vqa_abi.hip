// C-ABI layer of libvqahip.so: handles, argument checks, per-device streams and
// workspaces, host<->device staging for the host-pointer entry points, and the
// dispatch to the engines.  Declarations and the reference interfaces each
// entry point replaces: include/vqa_hip.h.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>

#include "vqa_core.h"

namespace vqa {

static thread_local char g_err[512] = "";
static int g_engine = 0;  // 0 auto

static std::mutex g_count_mu;
static int64_t g_pass_rw = 0, g_pass_ro = 0;
void count_state_passes(int64_t read_write, int64_t read_only) {
    std::lock_guard<std::mutex> lk(g_count_mu);
    g_pass_rw += read_write;
    g_pass_ro += read_only;
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- per-device state ------------------------------------------------------
// One stream and one growable workspace per device.  The workspace is shared by every circuit and
// every caller's stream, so its users are ordered: a call locks the device for the time it queues
// its work (DeviceGuard; two host threads cannot interleave their launches on one device), and a
// call that takes the workspace first makes its stream wait for the event the previous user
// recorded behind its last kernel.
struct DeviceState {
    hipStream_t stream = nullptr;
    void* ws = nullptr;
    size_t ws_bytes = 0;
    std::recursive_mutex* call_mu = nullptr;   // held while a call queues its work (g_call_mu[device])
    hipEvent_t ws_event = nullptr;    // recorded behind the last user of the workspace
    hipStream_t ws_stream = nullptr;  // that user's stream
    bool ws_recorded = false;
    bool ws_touched = false;          // the call in progress has taken the workspace
    hipStream_t touch_stream = nullptr;
    hipStream_t side[3] = { nullptr, nullptr, nullptr };   // call_side_streams (created on first use)
    hipEvent_t side_done[3] = { nullptr, nullptr, nullptr };
    hipEvent_t side_fork = nullptr;
    hipEvent_t own_done = nullptr;    // the library's own stream handed out as a side stream (call_side_streams)
    void* split_buf = nullptr;        // results of a split call's groups (split_call)
    size_t split_bytes = 0;
    void* ws_side[3] = { nullptr, nullptr, nullptr };   // workspaces of the groups that run on the side streams (slice i: side[i - 1] only)
    size_t ws_side_bytes[3] = { 0, 0, 0 };
    unsigned* tickets = nullptr;      // call_ticket_counter: one counter per workspace slice, 256 bytes apart, never cleared
    unsigned ticket_base[4] = { 0, 0, 0, 0 };   // what the counters will hold when everything queued so far has run
};
static std::mutex g_dev_mu;
static std::map<int, std::unique_ptr<DeviceState>> g_dev;
// One lock per device number, static so that taking it allocates nothing and needs no DeviceState:
// the allocation-free entry point orders its launches against every other call with it.
constexpr int kMaxDevices = 64;
static std::recursive_mutex g_call_mu[kMaxDevices];

static int device_state(int device, DeviceState** out) {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    auto it = g_dev.find(device);
    if (it == g_dev.end()) {
        int count = 0;
        VQA_HIP(hipGetDeviceCount(&count));
        if (device < 0 || device >= count || device >= kMaxDevices)
            VQA_FAIL(VQA_ENODEVICE, "device %d out of range (%d visible)", device, count);
        VQA_HIP(hipSetDevice(device));
        std::unique_ptr<DeviceState> ds(new DeviceState());
        ds->call_mu = &g_call_mu[device];
        VQA_HIP(hipStreamCreateWithFlags(&ds->stream, hipStreamNonBlocking));
        VQA_HIP(hipEventCreateWithFlags(&ds->ws_event, hipEventDisableTiming));
        it = g_dev.emplace(device, std::move(ds)).first;
    }
    *out = it->second.get();
    return VQA_OK;
}

// Locks a device for one call; on the way out, when the call took the workspace, records the event
// the next user on another stream waits for.
struct DeviceGuard {
    DeviceState* ds = nullptr;
    int lock(int device) {
        DeviceState* d;
        int rc = device_state(device, &d);
        if (rc) return rc;
        d->call_mu->lock();
        ds = d;
        return VQA_OK;
    }
    ~DeviceGuard() {
        if (!ds) return;
        if (ds->ws_touched) {
            ds->ws_touched = false;
            if (hipEventRecord(ds->ws_event, ds->touch_stream) == hipSuccess) {
                ds->ws_recorded = true;
                ds->ws_stream = ds->touch_stream;
            }
        }
        ds->call_mu->unlock();
    }
};

static int workspace_get(int device, hipStream_t stream, size_t bytes, void** out) {
    DeviceState* ds;
    int rc = device_state(device, &ds);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(*ds->call_mu);
    VQA_HIP(hipSetDevice(device));
    if (ds->ws_bytes < bytes) {
        if (ds->ws) {
            VQA_HIP(hipDeviceSynchronize());
            VQA_HIP(hipFree(ds->ws));
            ds->ws = nullptr;
            ds->ws_bytes = 0;
            ds->ws_recorded = false;
        }
        const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        hipError_t e = hipMalloc(&ds->ws, want);
        if (e != hipSuccess) VQA_FAIL(VQA_ENOMEM, "hipMalloc of %zu workspace bytes failed: %s", want, hipGetErrorString(e));
        ds->ws_bytes = want;
    }
    if (ds->ws_recorded && ds->ws_stream != stream) VQA_HIP(hipStreamWaitEvent(stream, ds->ws_event, 0));
    ds->ws_touched = true;
    ds->touch_stream = stream;
    *out = ds->ws;
    return VQA_OK;
}

// Which build of the tiled engine a circuit runs on.  11-qubit tiles from 13 to 25 qubits (below 13 nothing is
// tiled and the builds do not differ).  From 26 qubits on it depends on what binds the passes: a deep circuit
// (IBM-HEA, depth 8: some 20 non-diagonal gates per qubit, a dozen sweeps per pass) is bound by the work on the
// tile and keeps the 11-qubit tiles (26 / 27 qubits: 3.5 against 3.0, 1.5 against 1.3 gradients/s); a shallow one
// (RY + CZ ring, depth 6: 7 per qubit, 2-5 sweeps per pass) is bound by HBM and takes the 12-qubit tiles, which
// need fewer passes over the state (28 qubits: 3.97 against 4.16 s per gradient).  VQA_TILE12=1: always the
// latter.  Decided when a circuit first runs on the engine: its cached plans belong to one build.
static bool small_tiles(const EvalCall& call) {
    std::lock_guard<std::mutex> lk(call.c->mu);
    if (call.c->hbm_tiles == 0) {
        const int n = call.c->n_qubits;
        int64_t dense = 0;
        for (const vqa_op& op : call.c->ops)
            dense += op.kind == VQA_GATE_H || op.kind == VQA_GATE_X || op.kind == VQA_GATE_Y || op.kind == VQA_GATE_RX ||
                     op.kind == VQA_GATE_RY;
        const bool small = n >= 13 && (n <= 25 || dense >= 12 * (int64_t)n);
        call.c->hbm_tiles = (small && !getenv("VQA_TILE12")) ? 11 : 12;
    }
    return call.c->hbm_tiles == 11;
}
int hbm_engine_run(const EvalCall& call) { return small_tiles(call) ? hbm_engine_run_t11(call) : hbm_engine_run_t12(call); }
bool hbm_engine_can_sweep(const EvalCall& call) {
    return small_tiles(call) ? hbm_engine_can_sweep_t11(call) : hbm_engine_can_sweep_t12(call);
}
int hbm_engine_sweep(const EvalCall& call, int n_samples) {
    return small_tiles(call) ? hbm_engine_sweep_t11(call, n_samples) : hbm_engine_sweep_t12(call, n_samples);
}
int hbm_engine_statevector(const EvalCall& call, double* d_out_state) {
    return small_tiles(call) ? hbm_engine_statevector_t11(call, d_out_state) : hbm_engine_statevector_t12(call, d_out_state);
}

// The workspace of a group of a split call that runs on side stream `slice - 1`: that stream is its only user, so
// the stream's own order is all the ordering it needs.
static int workspace_slice(int device, int slice, size_t bytes, void** out) {
    DeviceState* ds;
    int rc = device_state(device, &ds);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(*ds->call_mu);
    VQA_HIP(hipSetDevice(device));
    void*& ws = ds->ws_side[slice - 1];
    size_t& have = ds->ws_side_bytes[slice - 1];
    if (have < bytes) {
        if (ws) {
            VQA_HIP(hipDeviceSynchronize());
            VQA_HIP(hipFree(ws));
            ws = nullptr;
            have = 0;
        }
        const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        hipError_t e = hipMalloc(&ws, want);
        if (e != hipSuccess) VQA_FAIL(VQA_ENOMEM, "hipMalloc of %zu workspace bytes failed: %s", want, hipGetErrorString(e));
        have = want;
    }
    *out = ws;
    return VQA_OK;
}

int call_workspace(const EvalCall& call, size_t bytes, void** out) {
    if (!call.fixed && call.ws_slice > 0) return workspace_slice(call.device, call.ws_slice, bytes, out);
    if (!call.fixed) return workspace_get(call.device, call.stream, bytes, out);
    FixedScratch& f = *call.fixed;
    const size_t at = (f.used + 255) & ~(size_t)255;
    if (at > f.bytes || bytes > f.bytes - at)
        VQA_FAIL(VQA_EUNSUPPORTED, "this call needs %zu bytes of working memory, the fixed region has %zu left", bytes,
                 at > f.bytes ? (size_t)0 : f.bytes - at);
    *out = f.base + at;
    f.used = at + bytes;
    return VQA_OK;
}

int call_side_streams(const EvalCall& call, int want, hipStream_t* streams, hipEvent_t* done, hipEvent_t* fork) {
    if (call.fixed || want <= 0) return 0;
    DeviceState* ds;
    if (device_state(call.device, &ds) != VQA_OK) return 0;
    std::lock_guard<std::recursive_mutex> lk(*ds->call_mu);
    if (hipSetDevice(call.device) != hipSuccess) return 0;
    if (!ds->side_fork && hipEventCreateWithFlags(&ds->side_fork, hipEventDisableTiming) != hipSuccess) return 0;
    int got = 0;
    want = std::min(want, 3);
    // The library's own stream first, unless the call runs on it: it exists anyway and holds a hardware queue (a
    // process gets four; with it left out the third side stream shared a queue with the second -- kernel trace of a
    // 28-qubit gradient's groups: two streams on one queue, a third of the launches waiting for their turn).
    if (ds->stream != call.stream && (ds->own_done || hipEventCreateWithFlags(&ds->own_done, hipEventDisableTiming) == hipSuccess)) {
        streams[got] = ds->stream;
        done[got] = ds->own_done;
        ++got;
    }
    for (int k = 0; got < want && k < 3; ++k, ++got) {
        if (!ds->side[k]) {
            hipStream_t st = nullptr;
            hipEvent_t ev = nullptr;
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
                (void)hipStreamDestroy(st);
                break;
            }
            ds->side[k] = st;
            ds->side_done[k] = ev;
        }
        streams[got] = ds->side[k];
        done[got] = ds->side_done[k];
    }
    *fork = ds->side_fork;
    return got;
}

unsigned* call_ticket_counter(const EvalCall& call, unsigned draws, unsigned* base) {
    if (call.fixed || call.ws_slice < 0 || call.ws_slice > 3) return nullptr;
    DeviceState* ds;
    if (device_state(call.device, &ds) != VQA_OK) return nullptr;
    std::lock_guard<std::recursive_mutex> lk(*ds->call_mu);
    if (!ds->tickets) {
        if (hipSetDevice(call.device) != hipSuccess) return nullptr;
        void* p = nullptr;
        if (hipMalloc(&p, 4 * 256) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, 4 * 256) != hipSuccess) {
            (void)hipFree(p);
            return nullptr;
        }
        ds->tickets = (unsigned*)p;
    }
    *base = ds->ticket_base[call.ws_slice];
    ds->ticket_base[call.ws_slice] += draws;   // (unsigned arithmetic on both sides: the counters may wrap)
    return ds->tickets + 64 * call.ws_slice;
}

void call_ticket_reset(const EvalCall& call) {
    DeviceState* ds;
    if (device_state(call.device, &ds) != VQA_OK) return;
    std::lock_guard<std::recursive_mutex> lk(*ds->call_mu);
    if (!ds->tickets) return;
    (void)hipDeviceSynchronize();
    (void)hipMemset(ds->tickets, 0, 4 * 256);
    for (unsigned& b : ds->ticket_base) b = 0;
}

int cone_lists_dev(const EvalCall& call, int kind, const std::vector<int32_t>& live, const std::vector<int32_t>& dead,
                   const int32_t** d_live, const int32_t** d_dead) {
    ConeInfo& cone = *call.cone;
    std::lock_guard<std::mutex> lk(call.c->mu);
    const size_t live_part = (live.size() * sizeof(int32_t) + 15) & ~(size_t)15;
    auto it = cone.list_dev.find({ call.device, kind });
    if (it == cone.list_dev.end()) {
        VQA_HIP(hipSetDevice(call.device));
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, std::max<size_t>(live_part + dead.size() * sizeof(int32_t), 16));
        if (e != hipSuccess) VQA_FAIL(VQA_ENOMEM, "hipMalloc of a sweep's run list failed: %s", hipGetErrorString(e));
        hipError_t e1 = live.empty() ? hipSuccess : hipMemcpy(d, live.data(), live.size() * sizeof(int32_t), hipMemcpyHostToDevice);
        hipError_t e2 = dead.empty() ? hipSuccess
                                     : hipMemcpy((char*)d + live_part, dead.data(), dead.size() * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            (void)hipFree(d);
            VQA_FAIL(VQA_EHIP, "upload of a sweep's run list failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        }
        it = cone.list_dev.emplace(std::make_pair(call.device, kind), d).first;
    }
    *d_live = (const int32_t*)it->second;
    *d_dead = (const int32_t*)((const char*)it->second + live_part);
    return VQA_OK;
}

size_t call_workspace_room(const EvalCall& call) {
    if (!call.fixed) return SIZE_MAX;
    const size_t at = (call.fixed->used + 255) & ~(size_t)255;
    return at > call.fixed->bytes ? 0 : call.fixed->bytes - at;
}

size_t call_workspace_held(const EvalCall& call) {
    if (call.fixed) return 0;
    DeviceState* ds;
    if (device_state(call.device, &ds) != VQA_OK) return 0;
    std::lock_guard<std::recursive_mutex> lk(*ds->call_mu);
    return ds->ws_bytes;
}

// copy kernel of call_upload: the bytes travel as a kernel argument
struct UploadChunk {
    uint4 v[192];   // 3 KiB of the 4 KiB a kernel's arguments may take
};
__global__ void k_upload(UploadChunk c, uint4* dst, int n16) {
    if ((int)threadIdx.x < n16) dst[threadIdx.x] = c.v[threadIdx.x];
}

int upload_by_value(hipStream_t stream, void* d_dst, const void* src, size_t bytes) {
    for (size_t off = 0; off < bytes; off += sizeof(UploadChunk)) {
        UploadChunk c;
        const size_t nb = std::min(sizeof(UploadChunk), bytes - off);
        memset(&c, 0, sizeof(c));
        memcpy(&c, (const char*)src + off, nb);
        hipLaunchKernelGGL(k_upload, dim3(1), dim3(192), 0, stream, c, (uint4*)((char*)d_dst + off), (int)((nb + 15) / 16));
    }
    VQA_HIP(hipGetLastError());
    return VQA_OK;
}

int call_upload(const EvalCall& call, void* d_dst, const void* src, size_t bytes) {
    // (a table of one chunk travels as a kernel argument either way: queuing a copy from pageable memory costs the
    // host some 50 us, a launch a tenth of it)
    if (!call.fixed && bytes > sizeof(UploadChunk)) {
        VQA_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, call.stream));
        return VQA_OK;
    }
    if (bytes == 0) return VQA_OK;
    return upload_by_value(call.stream, d_dst, src, bytes);
}

// ---- circuit helpers ---------------------------------------------------------
static int circuit_device_ctx(vqa_circuit* c, int device, DeviceCtx** out) {
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->dev.find(device);
    if (it == c->dev.end()) {
        DeviceCtx ctx;
        VQA_HIP(hipSetDevice(device));
        const size_t bytes = std::max<size_t>(1, c->ops.size()) * sizeof(vqa_op);
        hipError_t e = hipMalloc((void**)&ctx.d_ops, bytes);
        if (e != hipSuccess) VQA_FAIL(VQA_ENOMEM, "hipMalloc of the op table failed: %s", hipGetErrorString(e));
        if (!c->ops.empty())
            VQA_HIP(hipMemcpy(ctx.d_ops, c->ops.data(), c->ops.size() * sizeof(vqa_op), hipMemcpyHostToDevice));
        if (c->reg) {
            e = hipMalloc(&ctx.reg_prog, c->reg->blob.size());
            if (e != hipSuccess) VQA_FAIL(VQA_ENOMEM, "hipMalloc of the compiled program failed: %s", hipGetErrorString(e));
            VQA_HIP(hipMemcpy(ctx.reg_prog, c->reg->blob.data(), c->reg->blob.size(), hipMemcpyHostToDevice));
        }
        it = c->dev.emplace(device, ctx).first;
    }
    *out = &it->second;
    return VQA_OK;
}

static int pack_obs(const vqa_circuit* c, const vqa_obs* const* obs, int n_obs, ObsPack* p) {
    if (n_obs < 0 || n_obs > kMaxObs) VQA_FAIL(VQA_EINVAL, "n_obs must be in [0, %d] (got %d)", kMaxObs, n_obs);
    if (n_obs > 0 && !obs) VQA_FAIL(VQA_EINVAL, "obs is NULL with n_obs = %d", n_obs);
    memset(p, 0, sizeof(*p));
    p->n_obs = n_obs;
    int nt = 0;
    for (int o = 0; o < n_obs; ++o) {
        if (!obs[o]) VQA_FAIL(VQA_EINVAL, "obs[%d] is NULL", o);
        if (obs[o]->n_qubits != c->n_qubits)
            VQA_FAIL(VQA_EINVAL, "obs[%d] is on %d qubits, the circuit on %d", o, obs[o]->n_qubits, c->n_qubits);
        p->off[o] = nt;
        for (size_t t = 0; t < obs[o]->coef.size(); ++t) {
            if (nt >= kMaxTerms)
                VQA_FAIL(VQA_EUNSUPPORTED, "more than %d Pauli terms in one call", kMaxTerms);
            p->coef[nt] = obs[o]->coef[t];
            p->z[nt] = obs[o]->z[t];
            p->x[nt] = obs[o]->x[t];
            ++nt;
        }
    }
    p->off[n_obs] = nt;
    p->n_terms = nt;
    return VQA_OK;
}

// Backward light cone of the measured qubits through the gate list.  `full`: qubits that carry a
// general operator of the back-propagated observable, `zonly`: qubits that carry I or Z only.
// Going backwards, a diagonal gate commutes with Z-type factors; a non-diagonal 1-qubit gate
// turns a Z-type qubit into a general one; CZ / CNOT hand Z-type (CNOT from its control: general)
// factors to their partner.  A rotation is live when its qubit carries something it does not
// commute with; a parameter is dead when none of its rotations is live.
static std::vector<uint8_t> dead_parameters(const vqa_circuit& c, uint64_t zsupp, uint64_t xsupp,
                                            std::vector<uint8_t>* gate_dead) {
    uint64_t full = xsupp, zonly = zsupp & ~xsupp;
    std::vector<uint8_t> dead((size_t)c.n_theta, 1);
    gate_dead->assign((size_t)c.n_ops, 0);
    for (int g = c.n_ops - 1; g >= 0; --g) {
        const vqa_op& op = c.ops[g];
        const uint64_t b0 = (uint64_t)1 << op.q0, b1 = op.q1 >= 0 ? (uint64_t)1 << op.q1 : 0;
        bool live = false;
        switch (op.kind) {
        case VQA_GATE_H: case VQA_GATE_RX: case VQA_GATE_RY:
            live = ((full | zonly) & b0) != 0;
            if (live) { full |= b0; zonly &= ~b0; }
            break;
        case VQA_GATE_Z: case VQA_GATE_RZ:
            live = (full & b0) != 0;
            break;
        case VQA_GATE_X: case VQA_GATE_Y:   // Paulis map Z to -Z: types unchanged
            live = ((full | zonly) & b0) != 0;
            break;
        case VQA_GATE_CZ: {
            const uint64_t f = full;
            live = (f & (b0 | b1)) != 0;
            if ((f & b0) && !(f & b1)) zonly |= b1;
            if ((f & b1) && !(f & b0)) zonly |= b0;
            break;
        }
        case VQA_GATE_CNOT: {   // q0 = target, q1 = control
            // CNOT - RZ(target) - CNOT is the diagonal gate exp(i a/2 Z_c Z_t) (the ZZ feature maps): it commutes
            // with Z-type factors on both qubits and hands a Z-type factor to the partner of a general one, like CZ
            // (gate by gate the first CNOT would spread a Z-type factor from its target to its control for good)
            if (g >= 2 && c.ops[g - 1].kind == VQA_GATE_RZ && c.ops[g - 1].q0 == op.q0 && c.ops[g - 2].kind == VQA_GATE_CNOT &&
                c.ops[g - 2].q0 == op.q0 && c.ops[g - 2].q1 == op.q1) {
                const uint64_t f = full;
                live = (f & (b0 | b1)) != 0;
                if ((f & b0) && !(f & b1)) zonly |= b1;
                if ((f & b1) && !(f & b0)) zonly |= b0;
                (*gate_dead)[g] = (*gate_dead)[g - 1] = (*gate_dead)[g - 2] = live ? 0 : 1;
                if (live && c.ops[g - 1].src == VQA_SRC_THETA) dead[c.ops[g - 1].idx] = 0;
                g -= 2;
                continue;
            }
            const uint64_t f = full, z = zonly;
            live = ((f | z) & b0) != 0 || (f & b1) != 0;
            if (((f | z) & b0) && !(f & b1)) zonly |= b1;          // Z_t -> Z_c Z_t
            if (f & b1) { full |= b0; zonly &= ~b0; }              // X_c -> X_c X_t
            break;
        }
        default: break;
        }
        (*gate_dead)[g] = live ? 0 : 1;
        if (live && op.kind >= VQA_GATE_RX && op.kind <= VQA_GATE_RZ && op.src == VQA_SRC_THETA) dead[op.idx] = 0;
    }
    return dead;
}

static ConeInfo* cone_of(vqa_circuit* c, const ObsPack& obs, int n_marg_bits) {
    uint64_t zs = n_marg_bits > 0 ? (((uint64_t)1 << n_marg_bits) - 1) : 0, xs = 0;
    for (int t = 0; t < obs.n_terms; ++t) { zs |= obs.z[t]; xs |= obs.x[t]; }
    std::lock_guard<std::mutex> lk(c->mu);
    auto key = std::make_pair(zs & ~xs, xs);
    auto it = c->cones.find(key);
    if (it == c->cones.end()) {
        auto ci = std::make_shared<ConeInfo>();
        ci->dead = dead_parameters(*c, zs, xs, &ci->gate_dead);
        if (getenv("VQA_NO_CONE")) {
            ci->dead.assign((size_t)c->n_theta, 0);
            ci->gate_dead.assign((size_t)c->n_ops, 0);
        }
        for (uint8_t d : ci->gate_dead) ci->n_gate_dead += d;
        for (int i = 0; i < c->n_theta; ++i)
            if (ci->dead[i]) ci->dead_list.push_back(i);
        ci->n_dead = (int)ci->dead_list.size();
        it = c->cones.emplace(key, ci).first;
    }
    return it->second.get();
}

static int resolve_engine(const vqa_circuit* c) { return c->engine; }

// Engine 2 runs the circuit WITHOUT the gates outside the call's light cone: the program of
// that pruned gate list (compiled on first use; the circuit's own program when it has nothing to
// prune or the pruned list does not compile for sweeps).  Caller holds the circuit's mutex.
static void cone_reg_program_host(const EvalCall& call) {
    ConeInfo& cone = *call.cone;
    if (cone.reg_tried) return;
    cone.reg_tried = true;
    if (cone.n_gate_dead > 0) {
        std::vector<vqa_op> live;
        for (int g = 0; g < call.c->n_ops; ++g)
            if (!cone.gate_dead[g]) live.push_back(call.c->ops[g]);
        auto rp = std::make_shared<RegProgram>();
        std::string why;
        if (reg_compile(call.c->n_qubits, call.c->n_theta, live, rp.get(), &why) &&
            reg_engine_can_sweep(*rp) == reg_engine_can_sweep(*call.c->reg))
            cone.reg = rp;
        if (getenv("VQA_PLAN_DEBUG") && cone.reg) {
            fprintf(stderr, "[vqa] register program without %d gates: %d phases, %d transpositions, final layout (position: qubit)",
                    cone.n_gate_dead, rp->hdr.n_phases, rp->n_xpose);
            for (int p = 0; p < call.c->n_qubits; ++p) fprintf(stderr, " %d:%d", p, (int)rp->hdr.final_qubit_at_pos[p]);
            fprintf(stderr, "\n");
        }
    }
}

// ... and its copy on the call's device.
static int cone_reg_program(const EvalCall& call, const RegProgram** prog, void** d_prog) {
    ConeInfo& cone = *call.cone;
    std::lock_guard<std::mutex> lk(call.c->mu);
    cone_reg_program_host(call);
    if (!cone.reg) return VQA_OK;
    auto it = cone.reg_dev.find(call.device);
    if (it == cone.reg_dev.end()) {
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, cone.reg->blob.size());
        if (e != hipSuccess) VQA_FAIL(VQA_ENOMEM, "hipMalloc of the compiled program failed: %s", hipGetErrorString(e));
        VQA_HIP(hipMemcpy(d, cone.reg->blob.data(), cone.reg->blob.size(), hipMemcpyHostToDevice));
        it = cone.reg_dev.emplace(call.device, d).first;
    }
    *prog = cone.reg.get();
    *d_prog = it->second;
    return VQA_OK;
}

static int run_call(EvalCall& call, DeviceCtx* ctx) {
    VQA_HIP(hipSetDevice(call.device));
    int engine = resolve_engine(call.c);
    if (engine == 2) {
        // engine 2 measures in the Z basis only; X/Y Pauli terms and wide marginals go to the
        // gate-by-gate LDS engine (13 qubits: to the streaming engine); same results
        bool z_only = call.n_marg_bits <= 6;
        for (int t = 0; t < call.obs.n_terms; ++t) z_only &= call.obs.x[t] == 0;
        if (!z_only) engine = call.c->n_qubits <= 12 ? 1 : 3;   // (13, 14 qubits: the streaming engine)
        // automatic choice at 4 and 5 qubits (a state is a thread or two): sweeps stay here (their
        // groups share the prepared tables: 2x to 6x the LDS engine), explicit lists, where every
        // instance prepares all its fused 2x2s, go to the LDS engine (2x to 4x this one)
        else if (call.c->auto_engine && call.c->n_qubits <= 5 && !(call.spec.psr && call.n_sweep > 0)) engine = 1;
    }
    switch (engine) {
    case 1: return lds_engine_run(call);
    case 2:
        if (call.spec.psr && call.n_sweep > 0 && reg_engine_can_sweep(*call.c->reg)) {
            const RegProgram* prog = call.c->reg.get();
            void* d_prog = ctx->reg_prog;
            int rc = cone_reg_program(call, &prog, &d_prog);
            if (rc) return rc;
            if (reg_engine_can_sweep(*prog)) return reg_engine_sweep(call, d_prog, *prog, call.n_sweep);
        }
        {
            const RegProgram* prog = call.c->reg.get();
            void* d_prog = ctx->reg_prog;
            int rc = cone_reg_program(call, &prog, &d_prog);
            if (rc) return rc;
            if (call.spec.psr && call.n_sweep > 0 && call.cone && reg_engine_can_group(*prog) && !getenv("VQA_NO_GROUP"))
                return reg_engine_group(call, d_prog, *prog, call.n_sweep);
            return reg_engine_run(call, d_prog, *prog);
        }
    case 3:
        if (hbm_engine_can_sweep(call)) return hbm_engine_sweep(call, call.n_sweep);
        return hbm_engine_run(call);
    case 4: return hbm_engine_run(call);
    default: VQA_FAIL(VQA_EUNSUPPORTED, "engine %d is not available in this build", call.c->engine);
    }
}

// ---- light-cone splitting ---------------------------------------------------------------------
// <0|U^+ P U|0> of a Pauli string P depends only on the gates inside the backward light cone of P's
// qubits, and those gates touch the cone's qubits alone: the value is that of the circuit reduced to the
// cone, on as many qubits as the cone has.  A call whose terms have small cones (a sum of local terms
// behind a shallow circuit on a wide register: the variational eigensolver's energy) is therefore dealt
// into groups of terms, each evaluated on its own compacted circuit -- cone qubits renumbered in ascending
// order, gates outside the cone dropped, the parameters and input angles under their old numbers -- and
// the groups' results are added into the caller's output.  X / Y factors are measured as Z behind H or
// RZ(pi/2), H appended to the compacted circuit (RZ(a) = diag(e^{ia/2}, e^{-ia/2}): RZ(pi/2) is S^+ up
// to a phase), so the terms of a group agree qubit by qubit on the basis.  Exact, like the pruning of
// single gates.  Taken when the library chose the engine itself and the groups together cost less than
// the whole register; VQA_NO_CONE_SPLIT=1 (or VQA_NO_CONE=1) switches it off.
static int create_circuit(int engine_pref, int n_qubits, const vqa_op* ops, int n_ops, int n_theta, int n_input,
                          vqa_circuit** out);
static int run_call(EvalCall& call, DeviceCtx* ctx);

struct SplitGroup {
    uint64_t zs = 0, xs = 0;           // union of the terms' masks
    uint64_t xb = 0, yb = 0, zb = 0;   // measurement basis by qubit
    uint64_t qmask = 0;                // the cone's qubits
    int live = 0;                      // parameters inside the cone
    bool marg = false;                 // the call's marginal is measured here
    std::vector<int> terms;            // indices into the call's term list
    vqa_circuit* sub = nullptr;        // the compacted circuit
    std::vector<std::unique_ptr<vqa_obs>> obs;   // Z-only observables on its qubits, one per observable of the call present
    std::vector<const vqa_obs*> obs_ptr;
    std::vector<int32_t> parent;       // ... and which observable of the call each one adds to
};
struct SplitPlan {
    bool use = false;
    std::vector<SplitGroup> groups;
    std::vector<int32_t> gather;   // tables of k_split_gather: obs_off[n_obs + 1], then the entries (built on first use)
    ~SplitPlan() {
        for (SplitGroup& g : groups)
            if (g.sub) vqa_circuit_destroy(g.sub);
    }
};

static int popcount64(uint64_t v) { return __builtin_popcountll(v); }

// what a group costs, in amplitudes touched per gate layer: the register engine up to 14 qubits, the
// streaming engine beyond (slower per amplitude, and a chain of launches per sample)
static double split_cost(int nq, int live, bool sweep, double n_inst) {
    const bool reg = nq <= kRegBits + kMaxThreadBits;
    return ldexp(1.0, nq) * (sweep ? live + 1 : 1) * n_inst * (reg ? 1.0 : 8.0) + (reg ? 0x1p23 : 0x1p27);
}

static void split_walk(const vqa_circuit& c, uint64_t zs, uint64_t xs, uint64_t* qmask, int* live, std::vector<uint8_t>* gate_dead) {
    std::vector<uint8_t> gd;
    const std::vector<uint8_t> dead = dead_parameters(c, zs, xs, &gd);
    uint64_t q = zs | xs;
    for (int g = 0; g < c.n_ops; ++g)
        if (!gd[g]) {
            q |= (uint64_t)1 << c.ops[g].q0;
            if (c.ops[g].q1 >= 0 && (c.ops[g].kind == VQA_GATE_CNOT || c.ops[g].kind == VQA_GATE_CZ)) q |= (uint64_t)1 << c.ops[g].q1;
        }
    int n = 0;
    for (uint8_t d : dead) n += !d;
    *qmask = q ? q : 1;   // (a call with identity terms only: one qubit, no gate)
    *live = n;
    if (gate_dead) gate_dead->swap(gd);
}

static uint64_t compact_mask(uint64_t m, uint64_t qmask) {
    uint64_t out = 0;
    int pos = 0;
    for (int q = 0; q < 64; ++q)
        if ((qmask >> q) & 1) {
            if ((m >> q) & 1) out |= (uint64_t)1 << pos;
            ++pos;
        }
    return out;
}

// The plan of a call's observables (cached on the circuit; the caller holds no lock).  Null: run the call whole.
static SplitPlan* split_plan(vqa_circuit* c, const ObsPack& obs, int n_marg_bits, bool sweep, int64_t n_inst) {
    if (!c->auto_engine || getenv("VQA_NO_CONE_SPLIT") || getenv("VQA_NO_CONE")) return nullptr;
    if (obs.n_terms == 0 && n_marg_bits == 0) return nullptr;
    int bucket = 0;
    while (bucket < 16 && ((int64_t)1 << (bucket + 1)) <= n_inst) ++bucket;
    std::vector<uint64_t> key = { (uint64_t)obs.n_obs, (uint64_t)n_marg_bits, (uint64_t)sweep, (uint64_t)bucket };
    for (int o = 0; o < obs.n_obs; ++o)
        for (int t = obs.off[o]; t < obs.off[o + 1]; ++t) {
            uint64_t cb;
            memcpy(&cb, &obs.coef[t], 8);
            key.push_back((uint64_t)o); key.push_back(obs.z[t]); key.push_back(obs.x[t]); key.push_back(cb);
        }
    {
        std::lock_guard<std::mutex> lk(c->mu);
        auto it = c->splits.find(key);
        if (it != c->splits.end()) {
            SplitPlan* sp = (SplitPlan*)it->second.get();
            return sp->use ? sp : nullptr;
        }
    }
    const double ninst = (double)((int64_t)1 << bucket);
    auto plan = std::make_shared<SplitPlan>();
    // the whole call
    uint64_t zs_all = n_marg_bits > 0 ? (((uint64_t)1 << n_marg_bits) - 1) : 0, xs_all = 0;
    for (int t = 0; t < obs.n_terms; ++t) { zs_all |= obs.z[t]; xs_all |= obs.x[t]; }
    uint64_t q_all;
    int live_all;
    split_walk(*c, zs_all, xs_all, &q_all, &live_all, nullptr);
    const double whole = split_cost(c->n_qubits, live_all, sweep, ninst);
    // greedy grouping: a term joins the group whose cost it raises least, when that is less than a group of its own costs
    std::vector<int> term_obs(obs.n_terms, 0);
    for (int o = 0; o < obs.n_obs; ++o)
        for (int t = obs.off[o]; t < obs.off[o + 1]; ++t) term_obs[t] = o;
    std::vector<SplitGroup>& G = plan->groups;
    auto group_cost = [&](const SplitGroup& g) { return split_cost(popcount64(g.qmask), g.live, sweep, ninst); };
    if (n_marg_bits > 0) {
        SplitGroup g;
        g.marg = true;
        g.zs = g.zb = ((uint64_t)1 << n_marg_bits) - 1;
        split_walk(*c, g.zs, g.xs, &g.qmask, &g.live, nullptr);
        G.push_back(std::move(g));
    }
    for (int t = 0; t < obs.n_terms; ++t) {
        const uint64_t tx = obs.x[t] & ~obs.z[t], ty = obs.x[t] & obs.z[t], tz = obs.z[t] & ~obs.x[t];
        uint64_t q1;
        int l1;
        split_walk(*c, obs.z[t], obs.x[t], &q1, &l1, nullptr);
        const double alone = split_cost(popcount64(q1), l1, sweep, ninst);
        int best = -1;
        double best_delta = alone;
        uint64_t best_q = 0;
        int best_live = 0;
        for (size_t k = 0; k < G.size(); ++k) {
            const SplitGroup& g = G[k];
            if ((tx & (g.yb | g.zb)) || (ty & (g.xb | g.zb)) || (tz & (g.xb | g.yb))) continue;
            uint64_t qu;
            int lu;
            if ((q1 & ~g.qmask) == 0) { qu = g.qmask; lu = g.live; }   // (inside the group's cone already)
            else split_walk(*c, g.zs | obs.z[t], g.xs | obs.x[t], &qu, &lu, nullptr);
            const double delta = split_cost(popcount64(qu), lu, sweep, ninst) - group_cost(g);
            if (delta < best_delta) { best = (int)k; best_delta = delta; best_q = qu; best_live = lu; }
        }
        if (best < 0) {
            SplitGroup g;
            g.qmask = q1;
            g.live = l1;
            G.push_back(std::move(g));
            best = (int)G.size() - 1;
        } else {
            G[best].qmask = best_q;
            G[best].live = best_live;
        }
        SplitGroup& g = G[best];
        g.zs |= obs.z[t]; g.xs |= obs.x[t];
        g.xb |= tx; g.yb |= ty; g.zb |= tz;
        g.terms.push_back(t);
    }
    double total = 0.0;
    for (const SplitGroup& g : G) total += group_cost(g);
    plan->use = total < 0.7 * whole && G.size() <= 4096;
    if (getenv("VQA_PLAN_DEBUG")) {
        fprintf(stderr, "[vqa] light-cone split: %zu groups (qubits:", G.size());
        for (size_t k = 0; k < G.size() && k < 64; ++k) fprintf(stderr, " %d", popcount64(G[k].qmask));
        fprintf(stderr, "), cost %.3g against %.3g whole: %s\n", total, whole, plan->use ? "taken" : "not taken");
    }
    if (plan->use) {
        for (SplitGroup& g : G) {
            // the compacted circuit
            std::vector<uint8_t> gate_dead;
            split_walk(*c, g.zs, g.xs, &g.qmask, &g.live, &gate_dead);
            int pos_of[64];
            int nq = 0;
            for (int q = 0; q < 64; ++q) pos_of[q] = ((g.qmask >> q) & 1) ? nq++ : -1;
            std::vector<vqa_op> ops;
            for (int i = 0; i < c->n_ops; ++i) {
                if (gate_dead[i]) continue;
                vqa_op op = c->ops[i];
                op.q0 = pos_of[op.q0];
                if (op.kind == VQA_GATE_CNOT || op.kind == VQA_GATE_CZ) op.q1 = pos_of[op.q1];
                ops.push_back(op);
            }
            for (int q = 0; q < 64; ++q) {
                const uint64_t b = (uint64_t)1 << q;
                if (!((g.xb | g.yb) & b)) continue;
                vqa_op op;
                memset(&op, 0, sizeof(op));
                op.q0 = pos_of[q];
                op.q1 = -1;
                op.src = VQA_SRC_CONST;
                if (g.yb & b) {
                    op.kind = VQA_GATE_RZ;
                    op.coef = 1.0;
                    op.offset = 1.5707963267948966;
                    ops.push_back(op);
                    op.coef = op.offset = 0.0;
                }
                op.kind = VQA_GATE_H;
                ops.push_back(op);
            }
            if (create_circuit(0, nq, ops.data(), (int)ops.size(), c->n_theta, c->n_input, &g.sub) != VQA_OK) {
                plan->use = false;
                break;
            }
            // the group's observables: one per observable of the call that has terms here
            for (int t : g.terms) {
                const int o = term_obs[t];
                size_t k = 0;
                while (k < g.parent.size() && g.parent[k] != o) ++k;
                if (k == g.parent.size()) {
                    g.parent.push_back(o);
                    g.obs.emplace_back(new vqa_obs());
                    g.obs.back()->n_qubits = nq;
                }
                g.obs[k]->coef.push_back(obs.coef[t]);
                g.obs[k]->z.push_back(compact_mask(obs.z[t] | obs.x[t], g.qmask));
                g.obs[k]->x.push_back(0);
            }
            for (auto& o : g.obs) g.obs_ptr.push_back(o.get());
            if ((int)g.obs.size() > kMaxObs) { plan->use = false; break; }
        }
    }
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->splits.emplace(key, plan).first;
    SplitPlan* sp = (SplitPlan*)it->second.get();
    return sp->use ? sp : nullptr;
}

// out[row, o] = sum of the groups' columns that belong to observable o, in the plan's fixed order.
// part: the groups' blocks one after the other, block g = [E, width_g]; entries: (first column of the group's
// block, its width, the column inside it), obs_off[o] .. obs_off[o + 1] those of observable o.
__global__ void k_split_gather(const double* __restrict__ part, int64_t E, const int32_t* __restrict__ obs_off,
                               const int32_t* __restrict__ entries, double* __restrict__ out, int n_obs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E * n_obs) return;
    const int64_t row = i / n_obs;
    const int o = (int)(i - row * n_obs);
    double acc = 0.0;
    for (int k = obs_off[o]; k < obs_off[o + 1]; ++k) {
        const int32_t first = entries[3 * k], width = entries[3 * k + 1], col = entries[3 * k + 2];
        acc += part[E * first + row * width + col];
    }
    out[i] = acc;
}

// Runs a call group by group (the device is locked by the caller).  Every group writes its [E, width] block of
// the part buffer; one kernel on the call's stream adds the blocks into the caller's output at the end.  A call of
// few instances leaves most of the GPU idle per group (one sample's runs on a few dozen workgroups), so its
// groups go round-robin to the call's stream and the library's side streams, each with a workspace of its own.
static int split_call(const EvalCall& whole, SplitPlan& plan, DeviceState* ds) {
    const int64_t E = whole.E;
    if (E <= 0) return VQA_OK;
    const int n_obs = whole.obs.n_obs;
    // the gather tables (built once per plan) and the blocks' first columns
    std::vector<int32_t> first(plan.groups.size(), 0);
    int n_cols = 0;
    for (size_t g = 0; g < plan.groups.size(); ++g) {
        first[g] = n_cols;
        n_cols += (int)plan.groups[g].obs.size();
    }
    {
        std::lock_guard<std::mutex> lk(whole.c->mu);
        if (plan.gather.empty()) {
            std::vector<int32_t> off(n_obs + 1, 0), ent;
            for (int o = 0; o < n_obs; ++o) {
                for (size_t g = 0; g < plan.groups.size(); ++g)
                    for (size_t j = 0; j < plan.groups[g].parent.size(); ++j)
                        if (plan.groups[g].parent[j] == o) {
                            ent.push_back(first[g]); ent.push_back((int32_t)plan.groups[g].obs.size()); ent.push_back((int32_t)j);
                        }
                off[o + 1] = (int32_t)(ent.size() / 3);
            }
            plan.gather = off;
            plan.gather.insert(plan.gather.end(), ent.begin(), ent.end());
        }
    }
    const size_t tab_bytes = (plan.gather.size() * sizeof(int32_t) + 255) & ~(size_t)255;
    const size_t need = tab_bytes + (size_t)E * n_cols * sizeof(double);
    VQA_HIP(hipSetDevice(whole.device));
    if (ds->split_bytes < need) {
        if (ds->split_buf) {
            VQA_HIP(hipDeviceSynchronize());
            VQA_HIP(hipFree(ds->split_buf));
            ds->split_buf = nullptr;
            ds->split_bytes = 0;
        }
        const size_t want = (need + 65535) & ~(size_t)65535;
        hipError_t e = hipMalloc(&ds->split_buf, want);
        if (e != hipSuccess) VQA_FAIL(VQA_ENOMEM, "hipMalloc of %zu bytes for the groups' results failed: %s", want, hipGetErrorString(e));
        ds->split_bytes = want;
    }
    // (the part buffer is shared by all calls on the device, like the workspace: taking the workspace orders this
    // call's stream behind the previous user's)
    void* unused;
    int rc = workspace_get(whole.device, whole.stream, 1, &unused);
    if (rc) return rc;
    int32_t* d_tab = (int32_t*)ds->split_buf;
    double* part = (double*)((char*)ds->split_buf + tab_bytes);
    if (n_obs > 0) {
        rc = plan.gather.size() * sizeof(int32_t) <= 3072
                 ? upload_by_value(whole.stream, d_tab, plan.gather.data(), plan.gather.size() * sizeof(int32_t))
                 : (hipMemcpyAsync(d_tab, plan.gather.data(), plan.gather.size() * sizeof(int32_t), hipMemcpyHostToDevice,
                                   whole.stream) == hipSuccess ? VQA_OK : VQA_EHIP);
        if (rc) VQA_FAIL(rc, "upload of the gather tables failed");
    }
    // (a caller's parameter subset leaves the other rows of the blocks unwritten)
    if (whole.param_mask && n_cols > 0) VQA_HIP(hipMemsetAsync(part, 0, (size_t)E * n_cols * sizeof(double), whole.stream));
    // side streams: when a group cannot fill the GPU and none of the groups runs on the streaming engine (which
    // spreads its own work over the same streams)
    hipStream_t side[3];
    hipEvent_t done[3], fork;
    int n_side = 0;
    const bool few = whole.spec.psr && whole.n_sweep > 0 ? whole.n_sweep <= 64 : E <= 4096;
    bool small_engines = true;
    for (const SplitGroup& g : plan.groups) small_engines &= g.sub && g.sub->engine != 3 && g.sub->engine != 4;
    if (few && small_engines && plan.groups.size() >= 2 && !getenv("VQA_SPLIT_ONE_STREAM")) {
        n_side = call_side_streams(whole, (int)std::min<size_t>(3, plan.groups.size() - 1), side, done, &fork);
        if (n_side > 0) {
            VQA_HIP(hipEventRecord(fork, whole.stream));
            for (int i = 0; i < n_side; ++i) VQA_HIP(hipStreamWaitEvent(side[i], fork, 0));
        }
    }
    int turn = 0;
    rc = VQA_OK;
    for (size_t gi = 0; gi < plan.groups.size() && rc == VQA_OK; ++gi) {
        SplitGroup& g = plan.groups[gi];
        if (g.obs.empty() && !g.marg) continue;
        EvalCall call = whole;
        call.c = g.sub;
        call.ws_slice = turn % (n_side + 1);
        call.stream = call.ws_slice == 0 ? whole.stream : side[call.ws_slice - 1];
        ++turn;
        rc = pack_obs(g.sub, g.obs_ptr.data(), (int)g.obs_ptr.size(), &call.obs);
        if (rc) break;
        DeviceCtx* ctx;
        rc = circuit_device_ctx(g.sub, whole.device, &ctx);
        if (rc) break;
        call.d_ops = ctx->d_ops;
        call.n_marg_bits = g.marg ? whole.n_marg_bits : 0;
        call.d_out_marginal = g.marg ? whole.d_out_marginal : nullptr;
        call.cone = cone_of(g.sub, call.obs, call.n_marg_bits);
        call.d_out_expect = part + (size_t)E * first[gi];
        rc = run_call(call, ctx);
    }
    // the call's stream waits for the side streams whatever happened above
    for (int i = 0; i < n_side; ++i) {
        if (hipEventRecord(done[i], side[i]) != hipSuccess || hipStreamWaitEvent(whole.stream, done[i], 0) != hipSuccess) {
            (void)hipDeviceSynchronize();
            if (rc == VQA_OK) VQA_FAIL(VQA_EHIP, "joining the side streams of a split call failed");
        }
    }
    if (rc) return rc;
    if (n_obs > 0) {
        const int64_t total = E * n_obs;
        hipLaunchKernelGGL(k_split_gather, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, whole.stream, part, E, d_tab,
                           d_tab + (n_obs + 1), whole.d_out_expect, n_obs);
        VQA_HIP(hipGetLastError());
    }
    return VQA_OK;
}

static int check_common(vqa_circuit* c, int device, const double* theta, const double* input, int B,
                        int n_obs, int n_marg_bits, const double* out_e, const double* out_m) {
    if (!c) VQA_FAIL(VQA_EINVAL, "circuit handle is NULL");
    (void)device;
    if (c->n_theta > 0 && !theta) VQA_FAIL(VQA_EINVAL, "theta is NULL but the circuit has %d parameters", c->n_theta);
    if (c->n_input > 0 && (!input || B <= 0))
        VQA_FAIL(VQA_EINVAL, "input_angles is NULL or B <= 0 but the circuit has %d input gates", c->n_input);
    if (B < 0) VQA_FAIL(VQA_EINVAL, "B must be >= 0 (got %d)", B);
    if (n_marg_bits < 0 || n_marg_bits > c->n_qubits)
        VQA_FAIL(VQA_EINVAL, "n_marg_bits must be in [0, n_qubits] (got %d)", n_marg_bits);
    if (n_obs > 0 && !out_e) VQA_FAIL(VQA_EINVAL, "out_expect is NULL with n_obs = %d", n_obs);
    if (n_marg_bits > 0 && !out_m) VQA_FAIL(VQA_EINVAL, "out_marginal is NULL with n_marg_bits = %d", n_marg_bits);
    return VQA_OK;
}

// small RAII device buffer for the host-pointer entry points
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        hipError_t e = hipMalloc(&p, std::max<size_t>(bytes, 8));
        if (e != hipSuccess) VQA_FAIL(VQA_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return VQA_OK;
    }
    int upload(const void* src, size_t bytes, hipStream_t s) {
        int rc = alloc(bytes);
        if (rc) return rc;
        if (bytes) VQA_HIP(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, s));
        return VQA_OK;
    }
};

}  // namespace vqa

using namespace vqa;

extern "C" {

int vqa_version(void) { return VQA_ABI_VERSION; }

const char* vqa_last_error(void) { return g_err; }

int vqa_device_count(void) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) {
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return e == hipErrorNoDevice ? 0 : VQA_EHIP;
    }
    return count;
}

int vqa_state_pass_counters(int64_t* read_write, int64_t* read_only, int reset) {
    std::lock_guard<std::mutex> lk(g_count_mu);
    if (read_write) *read_write = g_pass_rw;
    if (read_only) *read_only = g_pass_ro;
    if (reset) g_pass_rw = g_pass_ro = 0;
    return VQA_OK;
}

int vqa_sync(int device) {
    DeviceState* ds;
    int rc = device_state(device, &ds);
    if (rc) return rc;
    VQA_HIP(hipStreamSynchronize(ds->stream));
    return VQA_OK;
}

int vqa_set_engine(int engine) {
    if (engine < 0 || engine > 4) VQA_FAIL(VQA_EINVAL, "engine must be 0..4 (got %d)", engine);
    g_engine = engine;
    return VQA_OK;
}

int vqa_circuit_create(int n_qubits, const vqa_op* ops, int n_ops, int n_theta, int n_input,
                       vqa_circuit** out) {
    return create_circuit(g_engine, n_qubits, ops, n_ops, n_theta, n_input, out);
}

}  // extern "C"

namespace vqa {
static int create_circuit(int engine_pref, int n_qubits, const vqa_op* ops, int n_ops, int n_theta, int n_input,
                          vqa_circuit** out) {
    if (!out) VQA_FAIL(VQA_EINVAL, "out is NULL");
    if (n_qubits < 1 || n_qubits > 30) VQA_FAIL(VQA_EINVAL, "n_qubits must be in [1, 30] (got %d)", n_qubits);
    if (n_ops < 0 || (n_ops > 0 && !ops)) VQA_FAIL(VQA_EINVAL, "ops is NULL or n_ops < 0");
    if (n_theta < 0 || n_input < 0) VQA_FAIL(VQA_EINVAL, "n_theta and n_input must be >= 0");
    for (int g = 0; g < n_ops; ++g) {
        const vqa_op& op = ops[g];
        if (op.kind < VQA_GATE_H || op.kind > VQA_GATE_CZ) VQA_FAIL(VQA_EINVAL, "op %d: unknown gate kind %d", g, op.kind);
        if (op.q0 < 0 || op.q0 >= n_qubits) VQA_FAIL(VQA_EINVAL, "op %d: qubit %d out of range", g, op.q0);
        const bool two = op.kind == VQA_GATE_CNOT || op.kind == VQA_GATE_CZ;
        if (two && (op.q1 < 0 || op.q1 >= n_qubits || op.q1 == op.q0))
            VQA_FAIL(VQA_EINVAL, "op %d: second qubit %d invalid", g, op.q1);
        if (op.kind >= VQA_GATE_RX && op.kind <= VQA_GATE_RZ) {
            if (op.src < VQA_SRC_CONST || op.src > VQA_SRC_INPUT) VQA_FAIL(VQA_EINVAL, "op %d: unknown angle source %d", g, op.src);
            if (op.src == VQA_SRC_THETA && (op.idx < 0 || op.idx >= n_theta))
                VQA_FAIL(VQA_EINVAL, "op %d: theta index %d out of range [0, %d)", g, op.idx, n_theta);
            if (op.src == VQA_SRC_INPUT && (op.idx < 0 || op.idx >= n_input))
                VQA_FAIL(VQA_EINVAL, "op %d: input index %d out of range [0, %d)", g, op.idx, n_input);
        }
    }
    int engine = engine_pref;
    std::shared_ptr<RegProgram> reg;
    if ((engine == 0 && n_qubits >= kRegBits && n_qubits <= kRegBits + kMaxThreadBits) || engine == 2) {
        reg = std::make_shared<RegProgram>();
        std::string why;
        std::vector<vqa_op> v(ops, ops + n_ops);
        if (!reg_compile(n_qubits, n_theta, v, reg.get(), &why)) {
            if (engine == 2) VQA_FAIL(VQA_EUNSUPPORTED, "the register-blocked engine cannot run this circuit: %s", why.c_str());
            reg.reset();
        } else {
            engine = 2;
        }
    }
    // automatic, when the register engine does not take the circuit: the LDS engine below the
    // streaming engine's tiles (at 13 qubits its sweeps are 12x the LDS engine's)
    if (engine == 0) engine = n_qubits <= 12 ? 1 : 3;
    if (engine == 1 && n_qubits > lds_engine_max_qubits())
        VQA_FAIL(VQA_EUNSUPPORTED, "the LDS engine holds at most %d qubits", lds_engine_max_qubits());
    vqa_circuit* c = new (std::nothrow) vqa_circuit();
    if (!c) VQA_FAIL(VQA_ENOMEM, "out of host memory");
    c->reg = reg;
    c->n_qubits = n_qubits;
    c->n_ops = n_ops;
    c->n_theta = n_theta;
    c->n_input = n_input;
    c->engine = engine;
    c->auto_engine = engine_pref == 0;
    c->ops.assign(ops, ops + n_ops);
    *out = c;
    return VQA_OK;
}
}  // namespace vqa

extern "C" {

int vqa_circuit_engine(const vqa_circuit* c) {
    if (!c) VQA_FAIL(VQA_EINVAL, "circuit handle is NULL");
    return c->engine;
}

int vqa_circuit_destroy(vqa_circuit* c) {
    if (!c) return VQA_OK;
    for (auto& kv : c->dev) {
        (void)hipSetDevice(kv.first);
        if (kv.second.d_ops) (void)hipFree(kv.second.d_ops);
        if (kv.second.reg_prog) (void)hipFree(kv.second.reg_prog);
    }
    for (auto& kv : c->cones) {
        for (auto& dv : kv.second->reg_dev) {
            (void)hipSetDevice(dv.first);
            (void)hipFree(dv.second);
        }
        for (auto& dv : kv.second->list_dev) {
            (void)hipSetDevice(dv.first.first);
            (void)hipFree(dv.second);
        }
    }
    delete c;
    return VQA_OK;
}

int vqa_circuit_info(const vqa_circuit* c, int* n_qubits, int* n_ops, int* n_theta, int* n_input) {
    if (!c) VQA_FAIL(VQA_EINVAL, "circuit handle is NULL");
    if (n_qubits) *n_qubits = c->n_qubits;
    if (n_ops) *n_ops = c->n_ops;
    if (n_theta) *n_theta = c->n_theta;
    if (n_input) *n_input = c->n_input;
    return VQA_OK;
}

int vqa_observable_create(int n_qubits, const double* coefs, const uint64_t* zmasks,
                          const uint64_t* xmasks, int n_terms, vqa_obs** out) {
    if (!out) VQA_FAIL(VQA_EINVAL, "out is NULL");
    if (n_qubits < 1 || n_qubits > 30) VQA_FAIL(VQA_EINVAL, "n_qubits must be in [1, 30] (got %d)", n_qubits);
    if (n_terms < 0 || (n_terms > 0 && (!coefs || !zmasks))) VQA_FAIL(VQA_EINVAL, "coefs/zmasks NULL or n_terms < 0");
    const uint64_t lim = (uint64_t)1 << n_qubits;
    for (int t = 0; t < n_terms; ++t) {
        if (zmasks[t] >= lim || (xmasks && xmasks[t] >= lim))
            VQA_FAIL(VQA_EINVAL, "term %d: Pauli mask has bits beyond qubit %d", t, n_qubits - 1);
    }
    vqa_obs* o = new (std::nothrow) vqa_obs();
    if (!o) VQA_FAIL(VQA_ENOMEM, "out of host memory");
    o->n_qubits = n_qubits;
    o->coef.assign(coefs, coefs + n_terms);
    o->z.assign(zmasks, zmasks + n_terms);
    if (xmasks) o->x.assign(xmasks, xmasks + n_terms);
    else o->x.assign(n_terms, 0);
    *out = o;
    return VQA_OK;
}

int vqa_observable_destroy(vqa_obs* o) {
    delete o;
    return VQA_OK;
}

static int eval_dev_common(vqa_circuit* c, int device, void* stream, const double* d_theta,
                           const double* d_input, int B, const EvalSpec& spec, int64_t E,
                           const vqa_obs* const* obs, int n_obs, int n_marg_bits, double* d_out_e,
                           double* d_out_m, const uint8_t* param_mask = nullptr) {
    int rc = check_common(c, device, d_theta, d_input, B, n_obs, n_marg_bits, d_out_e, d_out_m);
    if (rc) return rc;
    if (E < 0) VQA_FAIL(VQA_EINVAL, "E must be >= 0");
    EvalCall call;
    call.c = c;
    call.device = device;
    rc = pack_obs(c, obs, n_obs, &call.obs);
    if (rc) return rc;
    DeviceGuard guard;
    rc = guard.lock(device);
    if (rc) return rc;
    DeviceCtx* ctx;
    rc = circuit_device_ctx(c, device, &ctx);
    if (rc) return rc;
    call.stream = (hipStream_t)stream;
    call.d_ops = ctx->d_ops;
    call.d_theta = d_theta;
    call.d_input = d_input;
    call.spec = spec;
    call.E = E;
    call.n_marg_bits = n_marg_bits;
    call.n_sweep = spec.psr ? (int)(E / (2 * (int64_t)c->n_theta + 1)) : 0;
    call.cone = cone_of(c, call.obs, n_marg_bits);
    call.param_mask = param_mask;
    call.d_out_expect = d_out_e;
    call.d_out_marginal = d_out_m;
    if (SplitPlan* sp = split_plan(c, call.obs, n_marg_bits, spec.psr && call.n_sweep > 0,
                                   spec.psr && call.n_sweep > 0 ? call.n_sweep : E))
        return split_call(call, *sp, guard.ds);
    return run_call(call, ctx);
}

int vqa_eval_batch_dev(vqa_circuit* c, int device, void* stream, const double* d_theta,
                       const double* d_input_angles, int B, const int32_t* d_sample_idx,
                       const int32_t* d_shift_param, const double* d_shift_val, int64_t E,
                       const vqa_obs* const* obs, int n_obs, int n_marg_bits,
                       double* d_out_expect, double* d_out_marginal) {
    if ((d_shift_param == nullptr) != (d_shift_val == nullptr))
        VQA_FAIL(VQA_EINVAL, "shift_param and shift_val must both be given or both be NULL");
    if (!d_sample_idx && c && c->n_input > 0 && E > B)
        VQA_FAIL(VQA_EINVAL, "sample_idx is NULL but E (%lld) > B (%d)", (long long)E, B);
    EvalSpec spec = { d_sample_idx, d_shift_param, d_shift_val, nullptr, 0, 0, 0.0, c ? (c->n_input > 0 ? B : 0) : 0, c ? c->n_theta : 0 };
    return eval_dev_common(c, device, stream, d_theta, d_input_angles, B, spec, E, obs, n_obs,
                           n_marg_bits, d_out_expect, d_out_marginal);
}

int vqa_psr_full_dev(vqa_circuit* c, int device, void* stream, const double* d_theta,
                     const double* d_input_angles, int B, const int32_t* d_sel, int n_sel,
                     double shift, const vqa_obs* const* obs, int n_obs, int n_marg_bits,
                     double* d_out_expect, double* d_out_marginal) {
    if (!c) VQA_FAIL(VQA_EINVAL, "circuit handle is NULL");
    if (n_sel < 0) VQA_FAIL(VQA_EINVAL, "n_sel must be >= 0");
    if (!d_sel && c->n_input > 0 && n_sel > B)
        VQA_FAIL(VQA_EINVAL, "sel is NULL but n_sel (%d) > B (%d)", n_sel, B);
    EvalSpec spec = { nullptr, nullptr, nullptr, d_sel, 1, c->n_theta, shift, c->n_input > 0 ? B : 0, c->n_theta };
    const int64_t E = (int64_t)n_sel * (2 * (int64_t)c->n_theta + 1);
    return eval_dev_common(c, device, stream, d_theta, d_input_angles, B, spec, E, obs, n_obs,
                           n_marg_bits, d_out_expect, d_out_marginal);
}

int vqa_psr_params_dev(vqa_circuit* c, int device, void* stream, const double* d_theta,
                       const double* d_input_angles, int B, const int32_t* d_sel, int n_sel,
                       double shift, const uint8_t* param_mask, const vqa_obs* const* obs, int n_obs,
                       int n_marg_bits, double* d_out_expect, double* d_out_marginal) {
    if (!c) VQA_FAIL(VQA_EINVAL, "circuit handle is NULL");
    if (n_sel < 0) VQA_FAIL(VQA_EINVAL, "n_sel must be >= 0");
    if (!d_sel && c->n_input > 0 && n_sel > B)
        VQA_FAIL(VQA_EINVAL, "sel is NULL but n_sel (%d) > B (%d)", n_sel, B);
    EvalSpec spec = { nullptr, nullptr, nullptr, d_sel, 1, c->n_theta, shift, c->n_input > 0 ? B : 0, c->n_theta };
    const int64_t E = (int64_t)n_sel * (2 * (int64_t)c->n_theta + 1);
    return eval_dev_common(c, device, stream, d_theta, d_input_angles, B, spec, E, obs, n_obs,
                           n_marg_bits, d_out_expect, d_out_marginal, param_mask);
}

static int check_indices(const vqa_circuit* c, int B, const int32_t* idx, int64_t n, const char* what) {
    if (!idx || c->n_input == 0) return VQA_OK;
    for (int64_t i = 0; i < n; ++i)
        if (idx[i] < 0 || idx[i] >= B) VQA_FAIL(VQA_EINVAL, "%s[%lld] = %d out of range [0, %d)", what, (long long)i, idx[i], B);
    return VQA_OK;
}

int vqa_eval_batch(vqa_circuit* c, int device, const double* theta, const double* input_angles,
                   int B, const int32_t* sample_idx, const int32_t* shift_param,
                   const double* shift_val, int64_t E, const vqa_obs* const* obs, int n_obs,
                   int n_marg_bits, double* out_expect, double* out_marginal) {
    int rc = check_common(c, device, theta, input_angles, B, n_obs, n_marg_bits, out_expect, out_marginal);
    if (rc) return rc;
    if (E < 0) VQA_FAIL(VQA_EINVAL, "E must be >= 0");
    if (E == 0) return VQA_OK;
    rc = check_indices(c, B, sample_idx, E, "sample_idx");
    if (rc) return rc;
    if (shift_param)
        for (int64_t i = 0; i < E; ++i)
            if (shift_param[i] < -1 || shift_param[i] >= c->n_theta)
                VQA_FAIL(VQA_EINVAL, "shift_param[%lld] = %d out of range [-1, %d)", (long long)i, shift_param[i], c->n_theta);
    DeviceState* ds;
    rc = device_state(device, &ds);
    if (rc) return rc;
    VQA_HIP(hipSetDevice(device));
    hipStream_t s = ds->stream;
    const size_t M = (size_t)1 << n_marg_bits;
    DevBuf dth, din, dsi, dsp, dsv, doe, dom;
    if ((rc = dth.upload(theta, (size_t)c->n_theta * 8, s))) return rc;
    if ((rc = din.upload(input_angles, (size_t)B * c->n_input * 8, s))) return rc;
    if (sample_idx && (rc = dsi.upload(sample_idx, (size_t)E * 4, s))) return rc;
    if (shift_param && (rc = dsp.upload(shift_param, (size_t)E * 4, s))) return rc;
    if (shift_val && (rc = dsv.upload(shift_val, (size_t)E * 8, s))) return rc;
    if (n_obs > 0 && (rc = doe.alloc((size_t)E * n_obs * 8))) return rc;
    if (n_marg_bits > 0 && (rc = dom.alloc((size_t)E * M * 8))) return rc;
    rc = vqa_eval_batch_dev(c, device, s, (double*)dth.p, (double*)din.p, B, (int32_t*)dsi.p,
                            (int32_t*)dsp.p, (double*)dsv.p, E, obs, n_obs, n_marg_bits,
                            (double*)doe.p, (double*)dom.p);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    if (n_obs > 0) VQA_HIP(hipMemcpyAsync(out_expect, doe.p, (size_t)E * n_obs * 8, hipMemcpyDeviceToHost, s));
    if (n_marg_bits > 0) VQA_HIP(hipMemcpyAsync(out_marginal, dom.p, (size_t)E * M * 8, hipMemcpyDeviceToHost, s));
    VQA_HIP(hipStreamSynchronize(s));
    return VQA_OK;
}

int vqa_psr_full(vqa_circuit* c, int device, const double* theta, const double* input_angles, int B,
                 const int32_t* sel, int n_sel, double shift, const vqa_obs* const* obs, int n_obs,
                 int n_marg_bits, double* out_expect, double* out_marginal) {
    int rc = check_common(c, device, theta, input_angles, B, n_obs, n_marg_bits, out_expect, out_marginal);
    if (rc) return rc;
    if (n_sel < 0) VQA_FAIL(VQA_EINVAL, "n_sel must be >= 0");
    if (n_sel == 0) return VQA_OK;
    rc = check_indices(c, B, sel, n_sel, "sel");
    if (rc) return rc;
    DeviceState* ds;
    rc = device_state(device, &ds);
    if (rc) return rc;
    VQA_HIP(hipSetDevice(device));
    hipStream_t s = ds->stream;
    const size_t M = (size_t)1 << n_marg_bits;
    const size_t E = (size_t)n_sel * (2 * (size_t)c->n_theta + 1);
    DevBuf dth, din, dsel, doe, dom;
    if ((rc = dth.upload(theta, (size_t)c->n_theta * 8, s))) return rc;
    if ((rc = din.upload(input_angles, (size_t)B * c->n_input * 8, s))) return rc;
    if (sel && (rc = dsel.upload(sel, (size_t)n_sel * 4, s))) return rc;
    if (n_obs > 0 && (rc = doe.alloc(E * n_obs * 8))) return rc;
    if (n_marg_bits > 0 && (rc = dom.alloc(E * M * 8))) return rc;
    rc = vqa_psr_full_dev(c, device, s, (double*)dth.p, (double*)din.p, B, (int32_t*)dsel.p, n_sel,
                          shift, obs, n_obs, n_marg_bits, (double*)doe.p, (double*)dom.p);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    if (n_obs > 0) VQA_HIP(hipMemcpyAsync(out_expect, doe.p, E * n_obs * 8, hipMemcpyDeviceToHost, s));
    if (n_marg_bits > 0) VQA_HIP(hipMemcpyAsync(out_marginal, dom.p, E * M * 8, hipMemcpyDeviceToHost, s));
    VQA_HIP(hipStreamSynchronize(s));
    return VQA_OK;
}

// ---- the sweep entry point that allocates nothing ------------------------------------------------
// Working memory of vqa_psr_full_noalloc_dev: the compiled program, the work counter, the run
// lists, and per workgroup the state images and run records of a sweep, all inside this array.
// Every GPU the library runs on holds its own copy until the process ends.  (Size:
// VQA_NOALLOC_SCRATCH_BYTES of include/vqa_hip.h.)
static __device__ __align__(256) unsigned char scratch[VQA_NOALLOC_SCRATCH_BYTES];
constexpr int kMaxScratchDevices = kMaxDevices;
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[kMaxScratchDevices];   // device address of `scratch` on each GPU (null: not asked yet)

// The array's name is a host-side stand-in, not a device address: the address on `device` comes
// from hipGetSymbolAddress, once per device, with that device current.
static int scratch_address(int device, void** out) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        VQA_HIP(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) VQA_FAIL(VQA_EHIP, "hipGetSymbolAddress returned a null address for the scratch array");
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return VQA_OK;
}

int vqa_psr_full_noalloc_dev(vqa_circuit* c, int device, void* stream, const double* d_theta,
                             const double* d_input_angles, int B, const int32_t* d_sel, int n_sel,
                             double shift, const vqa_obs* const* obs, int n_obs, int n_marg_bits,
                             double* d_out_expect, double* d_out_marginal) {
    if (!c) VQA_FAIL(VQA_EINVAL, "circuit handle is NULL");
    if (n_sel < 0) VQA_FAIL(VQA_EINVAL, "n_sel must be >= 0");
    if (!d_sel && c->n_input > 0 && n_sel > B)
        VQA_FAIL(VQA_EINVAL, "sel is NULL but n_sel (%d) > B (%d)", n_sel, B);
    int rc = check_common(c, device, d_theta, d_input_angles, B, n_obs, n_marg_bits, d_out_expect, d_out_marginal);
    if (rc) return rc;
    if (c->engine != 2 || !c->reg)
        VQA_FAIL(VQA_EUNSUPPORTED, "the allocation-free sweep serves circuits of the register engine (4 to 14 qubits); "
                                   "this circuit runs on engine %d", c->engine);
    int count = 0;
    VQA_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count || device >= kMaxScratchDevices)
        VQA_FAIL(VQA_ENODEVICE, "device %d out of range (%d visible)", device, count);
    // the device's lock for the time the call queues its work, like every other entry point (a static
    // mutex: nothing is allocated); what runs on the GPU is ordered by the caller's streams alone
    std::lock_guard<std::recursive_mutex> device_lock(g_call_mu[device]);
    VQA_HIP(hipSetDevice(device));
    void* base = nullptr;
    if ((rc = scratch_address(device, &base))) return rc;
    if (n_sel == 0) return VQA_OK;
    FixedScratch fx = { (char*)base, VQA_NOALLOC_SCRATCH_BYTES, 0 };
    EvalCall call;
    call.c = c;
    call.device = device;
    rc = pack_obs(c, obs, n_obs, &call.obs);
    if (rc) return rc;
    call.stream = (hipStream_t)stream;
    call.d_ops = nullptr;   // (the register engine reads its compiled program, not the op table)
    call.d_theta = d_theta;
    call.d_input = d_input_angles;
    call.spec = EvalSpec{ nullptr, nullptr, nullptr, d_sel, 1, c->n_theta, shift, c->n_input > 0 ? B : 0, c->n_theta };
    call.E = (int64_t)n_sel * (2 * (int64_t)c->n_theta + 1);
    call.n_marg_bits = n_marg_bits;
    call.n_sweep = n_sel;
    call.cone = cone_of(c, call.obs, n_marg_bits);
    call.param_mask = nullptr;
    call.d_out_expect = d_out_expect;
    call.d_out_marginal = d_out_marginal;
    call.fixed = &fx;
    // the program of the gate list without the gates outside the light cone (host copy; compiled
    // when the cone is first seen), brought to the front of the scratch array on every call: the
    // array keeps whatever the last call left there, possibly another circuit's program
    const RegProgram* prog = c->reg.get();
    {
        std::lock_guard<std::mutex> lk(c->mu);
        cone_reg_program_host(call);
        if (call.cone->reg) prog = call.cone->reg.get();
    }
    void* d_prog = nullptr;
    if ((rc = call_workspace(call, prog->blob.size(), &d_prog))) return rc;
    if ((rc = call_upload(call, d_prog, prog->blob.data(), prog->blob.size()))) return rc;
    if (reg_engine_can_sweep(*c->reg) && reg_engine_can_sweep(*prog)) return reg_engine_sweep(call, d_prog, *prog, n_sel);
    if (reg_engine_can_group(*prog)) return reg_engine_group(call, d_prog, *prog, n_sel);
    return reg_engine_run(call, d_prog, *prog);
}

int vqa_light_cone_groups(vqa_circuit* c, const vqa_obs* const* obs, int n_obs, int n_marg_bits, int sweep,
                          int64_t n_instances, int32_t* out_qubits, int max_out) {
    if (!c) VQA_FAIL(VQA_EINVAL, "circuit handle is NULL");
    if (n_marg_bits < 0 || n_marg_bits > c->n_qubits) VQA_FAIL(VQA_EINVAL, "n_marg_bits must be in [0, n_qubits] (got %d)", n_marg_bits);
    if (n_instances < 1) VQA_FAIL(VQA_EINVAL, "n_instances must be >= 1");
    ObsPack pack;
    int rc = pack_obs(c, obs, n_obs, &pack);
    if (rc) return rc;
    SplitPlan* sp = split_plan(c, pack, n_marg_bits, sweep != 0, n_instances);
    if (!sp) return 0;
    int n = 0;
    for (const SplitGroup& g : sp->groups) {
        if (g.obs.empty() && !g.marg) continue;
        if (out_qubits && n < max_out) out_qubits[n] = g.sub ? g.sub->n_qubits : 0;
        ++n;
    }
    return n;
}

int vqa_dead_parameters(vqa_circuit* c, const vqa_obs* const* obs, int n_obs, int n_marg_bits, uint8_t* out_dead) {
    if (!c) VQA_FAIL(VQA_EINVAL, "circuit handle is NULL");
    if (!out_dead && c->n_theta > 0) VQA_FAIL(VQA_EINVAL, "out_dead is NULL");
    if (n_marg_bits < 0 || n_marg_bits > c->n_qubits)
        VQA_FAIL(VQA_EINVAL, "n_marg_bits must be in [0, n_qubits] (got %d)", n_marg_bits);
    ObsPack pack;
    int rc = pack_obs(c, obs, n_obs, &pack);
    if (rc) return rc;
    const ConeInfo* ci = cone_of(c, pack, n_marg_bits);
    for (int i = 0; i < c->n_theta; ++i) out_dead[i] = ci->dead[i];
    return ci->n_dead;
}

int vqa_sample_counts_dev(int device, void* stream, const double* d_probs, int n_outcomes,
                          const int64_t* d_shots, int64_t E, uint64_t seed, uint64_t stream_id,
                          int64_t* d_counts) {
    if (n_outcomes < 1 || n_outcomes > 1024) VQA_FAIL(VQA_EINVAL, "n_outcomes must be in [1, 1024] (got %d)", n_outcomes);
    if (E < 0) VQA_FAIL(VQA_EINVAL, "E must be >= 0");
    if (E == 0) return VQA_OK;
    if (!d_probs || !d_shots || !d_counts) VQA_FAIL(VQA_EINVAL, "probs, shots and counts must not be NULL");
    DeviceState* ds;
    int rc = device_state(device, &ds);
    if (rc) return rc;
    VQA_HIP(hipSetDevice(device));
    return sample_counts_launch((hipStream_t)stream, d_probs, n_outcomes, d_shots, E, seed, stream_id, d_counts);
}

int vqa_sample_counts_shard_dev(int device, void* stream, const double* d_probs, int n_outcomes,
                                const int64_t* d_shots, int64_t E, uint64_t seed, uint64_t stream_id,
                                int64_t* d_counts, int64_t row_offset) {
    if (n_outcomes < 1 || n_outcomes > 1024) VQA_FAIL(VQA_EINVAL, "n_outcomes must be in [1, 1024] (got %d)", n_outcomes);
    if (E < 0 || row_offset < 0) VQA_FAIL(VQA_EINVAL, "E and row_offset must be >= 0");
    if (E == 0) return VQA_OK;
    if (!d_probs || !d_shots || !d_counts) VQA_FAIL(VQA_EINVAL, "probs, shots and counts must not be NULL");
    DeviceState* ds;
    int rc = device_state(device, &ds);
    if (rc) return rc;
    VQA_HIP(hipSetDevice(device));
    return sample_counts_launch((hipStream_t)stream, d_probs, n_outcomes, d_shots, E, seed, stream_id, d_counts, row_offset);
}

int vqa_class_stats_dev(int device, void* stream, const int64_t* d_counts, const int64_t* d_shots, int64_t n_samples,
                        int n_params, const int32_t* d_label, int kind, int want_variances, double* d_out_sums,
                        double* d_out_grad_data) {
    if (n_samples < 0 || n_params < 0) VQA_FAIL(VQA_EINVAL, "n_samples and n_params must be >= 0");
    if (kind < 0 || kind > 2) VQA_FAIL(VQA_EINVAL, "kind must be 0 (partial MSE), 1 (full MSE) or 2 (cross entropy)");
    if (n_params == 0) return VQA_OK;
    if (!d_out_sums) VQA_FAIL(VQA_EINVAL, "out_sums is NULL");
    if (n_samples > 0 && (!d_counts || !d_shots || !d_label)) VQA_FAIL(VQA_EINVAL, "counts, shots and labels must not be NULL");
    int count = 0;
    VQA_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) VQA_FAIL(VQA_ENODEVICE, "device %d out of range (%d visible)", device, count);
    VQA_HIP(hipSetDevice(device));
    return class_stats_launch((hipStream_t)stream, d_counts, d_shots, n_samples, n_params, d_label, kind, want_variances,
                              d_out_sums, d_out_grad_data);
}

int vqa_sample_counts(int device, const double* probs, int n_outcomes, const int64_t* shots, int64_t E,
                      uint64_t seed, uint64_t stream_id, int64_t* counts) {
    if (n_outcomes < 1 || n_outcomes > 1024) VQA_FAIL(VQA_EINVAL, "n_outcomes must be in [1, 1024] (got %d)", n_outcomes);
    if (E < 0) VQA_FAIL(VQA_EINVAL, "E must be >= 0");
    if (E == 0) return VQA_OK;
    if (!probs || !shots || !counts) VQA_FAIL(VQA_EINVAL, "probs, shots and counts must not be NULL");
    DeviceState* ds;
    int rc = device_state(device, &ds);
    if (rc) return rc;
    VQA_HIP(hipSetDevice(device));
    hipStream_t s = ds->stream;
    DevBuf dp, dn, dc;
    if ((rc = dp.upload(probs, (size_t)E * n_outcomes * 8, s))) return rc;
    if ((rc = dn.upload(shots, (size_t)E * 8, s))) return rc;
    if ((rc = dc.alloc((size_t)E * n_outcomes * 8))) return rc;
    rc = sample_counts_launch(s, (const double*)dp.p, n_outcomes, (const int64_t*)dn.p, E, seed, stream_id, (int64_t*)dc.p);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    VQA_HIP(hipMemcpyAsync(counts, dc.p, (size_t)E * n_outcomes * 8, hipMemcpyDeviceToHost, s));
    VQA_HIP(hipStreamSynchronize(s));
    return VQA_OK;
}

int64_t vqa_mse_stats_workspace(int64_t B, int P) {
    if (B < 0 || P < 0) return 0;
    return (int64_t)mse_stats_workspace_bytes(B, P);
}

int vqa_mse_stats_dev(int device, void* stream, const double* d_expect, const double* d_estimates, int64_t B, int P,
                      const double* d_y, double w, const double* d_shots, const double* d_weight, double n_shots0,
                      int mode, uint64_t seed, int want_variances, double n0, int second_order, double* d_workspace,
                      double* d_out_sums, double* d_out_m0, double* d_out_grad_data) {
    return vqa_mse_stats_shard_dev(device, stream, d_expect, d_estimates, B, P, d_y, w, d_shots, d_weight, n_shots0, mode, seed,
                                   want_variances, n0, second_order, d_workspace, d_out_sums, d_out_m0, d_out_grad_data, 0);
}

int vqa_mse_stats_shard_dev(int device, void* stream, const double* d_expect, const double* d_estimates, int64_t B, int P,
                            const double* d_y, double w, const double* d_shots, const double* d_weight, double n_shots0,
                            int mode, uint64_t seed, int want_variances, double n0, int second_order, double* d_workspace,
                            double* d_out_sums, double* d_out_m0, double* d_out_grad_data, int64_t row_offset) {
    if (row_offset < 0) VQA_FAIL(VQA_EINVAL, "row_offset must be >= 0");
    if (B < 0 || P < 0) VQA_FAIL(VQA_EINVAL, "B and P must be >= 0");
    if (mode < 0 || mode > 3) VQA_FAIL(VQA_EINVAL, "mode must be 0 (exact), 1 (CLT), 2 (finite shots) or 3 (given estimates)");
    if (!d_out_sums) VQA_FAIL(VQA_EINVAL, "out_sums is NULL");
    if (B == 0) return VQA_OK;
    if (!d_expect || !d_y || !d_workspace) VQA_FAIL(VQA_EINVAL, "expect, y and workspace must not be NULL");
    if (mode == 3 && !d_estimates) VQA_FAIL(VQA_EINVAL, "mode 3 needs the estimates");
    if (mode != 0 && !d_shots) VQA_FAIL(VQA_EINVAL, "modes 1 to 3 need the shot list");
    if (mode == 0 && want_variances) VQA_FAIL(VQA_EINVAL, "the variance estimators need estimates from shots (modes 1 to 3)");
    int count = 0;
    VQA_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) VQA_FAIL(VQA_ENODEVICE, "device %d out of range (%d visible)", device, count);
    VQA_HIP(hipSetDevice(device));
    return mse_stats_launch((hipStream_t)stream, d_expect, d_estimates, B, P, d_y, w, d_shots, d_weight, n_shots0, mode, seed,
                            want_variances, n0, second_order, d_workspace, d_out_sums, d_out_m0, d_out_grad_data, row_offset);
}

int vqa_irandom_stats_dev(int device, void* stream, const double* d_expect, int n_obs, const int32_t* d_label,
                          const double* d_y, const int64_t* d_shots, const int32_t* d_seg, const double* d_n_list,
                          int n_components, int kind, int has_scale, double w, double M_tot, uint64_t seed,
                          double* d_out) {
    if (n_components < 0) VQA_FAIL(VQA_EINVAL, "n_components must be >= 0");
    if (n_components == 0) return VQA_OK;
    if (kind != 0 && kind != 1) VQA_FAIL(VQA_EINVAL, "kind must be 0 (Z0 outcomes) or 1 (class successes)");
    if (n_obs < 1) VQA_FAIL(VQA_EINVAL, "n_obs must be >= 1");
    if (!d_expect || !d_shots || !d_seg || !d_n_list || !d_out) VQA_FAIL(VQA_EINVAL, "expect, shots, seg, n_list and out must not be NULL");
    if (kind == 0 && !d_y) VQA_FAIL(VQA_EINVAL, "kind 0 needs the labels y");
    if (kind == 1 && !d_label) VQA_FAIL(VQA_EINVAL, "kind 1 needs the class labels");
    if (kind == 1 && has_scale) VQA_FAIL(VQA_EINVAL, "the classifier has no scale parameter");
    int count = 0;
    VQA_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) VQA_FAIL(VQA_ENODEVICE, "device %d out of range (%d visible)", device, count);
    VQA_HIP(hipSetDevice(device));
    return irandom_stats_launch((hipStream_t)stream, d_expect, n_obs, d_label, d_y, d_shots, d_seg, d_n_list, n_components,
                                kind, has_scale, w, M_tot, seed, d_out);
}

int vqa_class_grad_dev(int device, void* stream, const int64_t* d_counts, const int64_t* d_shots, int64_t n_samples,
                       int n_params, const int32_t* d_label, const double* d_onehot, double* d_grad) {
    if (n_samples < 0 || n_params < 0) VQA_FAIL(VQA_EINVAL, "n_samples and n_params must be >= 0");
    if (n_samples == 0 || n_params == 0) return VQA_OK;
    if (!d_counts || !d_shots || !d_grad) VQA_FAIL(VQA_EINVAL, "counts, shots and grad must not be NULL");
    if ((d_label != nullptr) == (d_onehot != nullptr))
        VQA_FAIL(VQA_EINVAL, "exactly one of the class labels (partial MSE) and the one-hot labels (full MSE) is given");
    int count = 0;
    VQA_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) VQA_FAIL(VQA_ENODEVICE, "device %d out of range (%d visible)", device, count);
    VQA_HIP(hipSetDevice(device));
    return class_grad_launch((hipStream_t)stream, d_counts, d_shots, n_samples, n_params, d_label, d_onehot, d_grad);
}

int vqa_statevector(vqa_circuit* c, int device, const double* theta, const double* input_row,
                    int shift_param, double shift_val, double* out_state) {
    if (!c) VQA_FAIL(VQA_EINVAL, "circuit handle is NULL");
    if (!out_state) VQA_FAIL(VQA_EINVAL, "out_state is NULL");
    if (c->n_theta > 0 && !theta) VQA_FAIL(VQA_EINVAL, "theta is NULL");
    if (c->n_input > 0 && !input_row) VQA_FAIL(VQA_EINVAL, "input_row is NULL");
    if (shift_param < -1 || shift_param >= c->n_theta) VQA_FAIL(VQA_EINVAL, "shift_param out of range");
    DeviceGuard guard;   // (the streaming engine's workspace: ordered against its other users)
    int rc = guard.lock(device);
    if (rc) return rc;
    DeviceState* ds = guard.ds;
    DeviceCtx* ctx;
    rc = circuit_device_ctx(c, device, &ctx);
    if (rc) return rc;
    VQA_HIP(hipSetDevice(device));
    hipStream_t s = ds->stream;
    const size_t D = (size_t)1 << c->n_qubits;
    DevBuf dth, din, dsp, dsv, dst;
    if ((rc = dth.upload(theta, (size_t)c->n_theta * 8, s))) return rc;
    if ((rc = din.upload(input_row, (size_t)c->n_input * 8, s))) return rc;
    const int32_t sp = shift_param;
    if ((rc = dsp.upload(&sp, 4, s))) return rc;
    if ((rc = dsv.upload(&shift_val, 8, s))) return rc;
    if ((rc = dst.alloc(D * 16))) return rc;
    EvalCall call;
    call.c = c;
    call.device = device;
    call.stream = s;
    call.d_ops = ctx->d_ops;
    call.d_theta = (double*)dth.p;
    call.d_input = (double*)din.p;
    call.spec = EvalSpec{ nullptr, (int32_t*)dsp.p, (double*)dsv.p, nullptr, 0, 0, 0.0, 0, c->n_theta };
    call.E = 1;
    memset(&call.obs, 0, sizeof(call.obs));
    call.n_marg_bits = 0;
    call.n_sweep = 0;
    call.cone = nullptr;
    call.param_mask = nullptr;
    call.d_out_expect = nullptr;
    call.d_out_marginal = nullptr;
    rc = hbm_engine_statevector(call, (double*)dst.p);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    VQA_HIP(hipMemcpyAsync(out_state, dst.p, D * 16, hipMemcpyDeviceToHost, s));
    VQA_HIP(hipStreamSynchronize(s));
    return VQA_OK;
}

}  // extern "C"
