#!/usr/bin/env python3
"""Benchmark of the hot path: full parameter-shift gradient evaluation of the
12-qubit hardware-efficient QNN (IBM feature map + depth-10 trainable block,
636 gates, 396 circuit parameters) over a batch of 65536 samples per GPU.

One "step" = one pass of the hot path over the batch: B * (2P+1) circuit
evaluations on the GPU (vqa_psr_full_dev, inputs resident in HBM), the
per-sample products and batch means that turn them into the regressor's MSE
gradient (vqa/QNN/myQNNs_base.py:1378-1394), and for N > 1 one all-reduce of the
P+1 gradient sums over RCCL.  Every one of the B * (2P+1) expectation values is
computed and written each step; the sweep kernel shares the unshifted prefix of
a sample's circuit between its shifted evaluations ("prefix_reuse").

The roofline object places the sweep kernel on the FP64 vector unit, the
resource rocprof shows it bound by (state vectors of 12 qubits live in
registers and LDS).  "achieved" / "frac" are the FP64 work the kernel EXECUTES
(PMC instruction counts of the committed profile of this same command, per
evaluation) at the launch time measured live; the ALGORITHMIC work of the
submitted, unfused gate list is reported beside it ("alg_*"): it exceeds the
peak because the engine fuses gates, drops redundant phases, reuses each
sample's unshifted prefix and leaves out gates outside the light cone.

Usage: python bench.py --gpus N --steps K --warmup W
(N > 1: either launched by torch.distributed.run, one rank per GPU, or started
plainly: the parent then spawns its N ranks itself before it touches the GPU.)
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# vendor peaks (MI355X): FP64 vector 78.6 TFLOP/s; HBM3E 8 TB/s; LDS 128 B/clk/CU
# x 256 CUs x 2.4 GHz.  The same three ceilings are measured on the box by
# vqa_amd/csrc/peaks.hip and printed beside these.
FP64_PEAK_TFLOPS = 78.6
HBM_PEAK_GBS = 8000.0
LDS_PEAK_GBS = 128 * 256 * 2.4

WORKLOADS = {
    # name: (n_qubits, depth, batch per GPU)
    "hea12_d10_b65536": (12, 10, 65536),
    "hea6_d5_b4096": (6, 5, 4096),
}


def alg_work(n, d):
    """Algorithmic, unfused work per circuit evaluation of the IBM-HEA circuit:
    every gate = one read + one write of the state (2*16*2^n bytes) plus one
    final read for the observable; flops: dense 1q gate 14*2^n, diagonal 1q
    6*2^n, CNOT/CZ 0, Z-string expectation 3*2^n."""
    G = 13 * n + 4 * n * d
    dense = 2 * n + 2 * n * (d + 1)
    diag = 4 * n + n * (d + 1)
    D = 1 << n
    return G, (G * 32 + 16) * D, (dense * 14 + diag * 6 + 3) * D


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="hea12_d10_b65536", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="override the per-GPU batch (not a valid headline number)")
    ap.add_argument("--engine", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short single-GPU measurements of the 6- and 20-qubit configurations")
    ap.add_argument("--backend", default="nccl", help="process-group backend (gloo: rehearsals of the N > 1 path)")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the process group even for one rank (exercises the RCCL calls on a one-GPU box)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the workload's batch on every GPU; strong: that batch split over the GPUs (contiguous blocks)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="every rank on cuda:0 (rehearsal of the N > 1 path on a one-GPU box; not a valid number)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly with --gpus N: this process becomes the launcher of N fresh rank processes
        # (it has made no GPU call and imports nothing that does)
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    if os.environ.get("VQA_BENCH_WATCHDOG"):
        # rehearsals: Python stacks of every thread, then exit, if the run is still going after this many seconds
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["VQA_BENCH_WATCHDOG"]), exit=True)

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if rank == 0:
            print(f"WORLD_SIZE={world} does not match --gpus {args.gpus}", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the HIP engine has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(args.backend)

    from vqa_amd import LearningCircuit, Observable, _lib, create_ibm_HEA_ansatz
    from vqa_amd.distributed import sharded_mse_gradient

    n, d, B = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    B_config = B
    if args.scaling == "strong":
        # the same total batch on any number of GPUs: rank r takes the r-th contiguous block
        from vqa_amd.distributed import shard_bounds

        lo, hi = shard_bounds(B, rank, world)
        B = hi - lo
    if args.engine:
        _lib.check(_lib.lib().vqa_set_engine(args.engine))
    circ = create_ibm_HEA_ansatz(n, d, seed=0, circuit_cls=lambda nq: LearningCircuit(nq, device=local_rank))
    P = circ.n_theta
    K = 2 * P + 1
    z0 = Observable(n)
    z0.add_operator(1.0, "Z 0")

    # synthetic data of the benchmark's shape: x ~ U(0, pi)^n, y = a smooth target
    rng = np.random.default_rng(2 + rank)
    x = rng.uniform(0.0, np.pi, (B, n))
    y = np.sin(x.sum(axis=1))
    theta = np.random.default_rng(0).uniform(0.0, 2 * np.pi, P)
    w = 1.0
    # vectorised restatement of the ansatz' input functions (one row per sample)
    xi = x
    zz = (np.pi - x) * (np.pi - x[:, :1])
    one = np.empty((B, 2 * n))
    one[:, 0::2] = xi
    one[:, 1::2] = zz
    ia = np.concatenate([one, one], axis=1)
    chk = circ.input_angles(x[:4])
    assert np.allclose(chk, ia[:4]), "vectorised input angles disagree with the circuit's input functions"

    dev = torch.device(f"cuda:{local_rank}")
    d_in = torch.from_numpy(np.ascontiguousarray(ia)).to(dev)
    d_theta = torch.from_numpy(theta).to(dev)
    d_y = torch.from_numpy(y).to(dev)
    d_out = torch.empty((B, K, 1), dtype=torch.float64, device=dev)
    measured = measured_peaks(local_rank) if rank == 0 else {}

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    class Timed:
        """the circuit with its sweep bracketed by events on torch's current stream
        (the stream the library launches on)"""

        def __init__(self):
            self.i = None

        def psr_full(self, *a, **k):
            if self.i is not None:
                ev0[self.i].record()
            r = circ.psr_full(*a, **k)
            if self.i is not None:
                ev1[self.i].record()
            return r

    timed = Timed()
    B_global = B_config if args.scaling == "strong" else B * world

    def step(i=None):
        timed.i = i
        return sharded_mse_gradient(timed, d_theta, d_in, d_y, w, z0, n_global=B_global, out_expect=d_out)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        g = step(i)
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    assert bool(torch.isfinite(g).all()), "gradient is not finite"

    if rank == 0:
        evals = B * K
        G, bytes_alg, flops_alg = alg_work(n, d)
        evals_per_s_kernel = evals / (kern_ms * 1e-3)
        grads_per_s = B_global * args.steps / dt
        res = {
            "metric": "parameter-shift gradients/sec",
            "value": grads_per_s,
            "unit": "gradients/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "circuit_evals_per_sec": B_global * K * args.steps / dt,
            "config": {
                "workload": f"{n}-qubit IBM-HEA QNN regressor, depth {d}, {G} gates, P={P}, "
                            + (f"batch {B} samples per GPU" if args.scaling == "weak" else f"batch {B_global} samples split over the GPUs")
                            + f", full parameter shift ({K} evaluations per sample)",
                "n_qubits": n, "depth": d, "batch_per_gpu": B, "global_batch": B_global, "n_params": P,
                "evals_per_step_per_gpu": evals, "engine": args.engine or "auto",
                "prefix_reuse": True,
                # parameters whose rotations lie outside the backward light cone of the measured qubit:
                # their shifted values equal the unshifted one exactly and are copied, not recomputed
                "light_cone_dead_parameters": int(circ.dead_parameters([z0], 0).sum()),
            },
            "roofline": roofline(bytes_alg, flops_alg, evals, kern_ms, measured),
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(circ, theta, ia, P, args.cpu_seconds)
        if not args.no_other_configs and world == 1 and args.workload == "hea12_d10_b65536" and not args.batch:
            # the same step with the light-cone pruning switched off (every gate of every shifted circuit
            # runs, nothing copied): what the sweep kernel does without that piece of circuit analysis
            res["without_light_cone_pruning"] = unpruned_step(local_rank, n, d, d_theta, d_in, d_y, w, z0, B, d_out)
            d_out = d_in = None   # (the 3.3 GB of headline outputs: the 20-qubit sweep wants the memory)
            torch.cuda.empty_cache()
            res["other_configs"] = other_configs(local_rank)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


def measured_peaks(device):
    """This GPU's own ceilings from the micro-benchmarks in vqa_amd/csrc/peaks.hip."""
    import ctypes as C

    path = os.path.join(ROOT, "vqa_amd", "csrc", "libvqapeaks.so")
    if not os.path.exists(path):
        return {}
    lib = C.CDLL(path)
    out = {}
    for key, fn in (("fp64_fma_tflops", "vqa_peak_fp64_fma"), ("lds_gbs", "vqa_peak_lds"), ("hbm_copy_gbs", "vqa_peak_hbm_copy")):
        v = C.c_double(0.0)
        if getattr(lib, fn)(C.c_int(device), C.byref(v)) == 0:
            out[key] = v.value
    return out


def profiled(key):
    """A per-evaluation figure of the sweep kernel from the committed rocprofv3 PMC passes
    of this same command (profiles/hbm_traffic.json, written by tools/hbm_traffic.py);
    None when absent."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            return float(json.load(f)[key])
    except (OSError, KeyError, ValueError):
        return None


def hbm_traffic_per_launch(evals):
    """HBM bytes per launch (PMC FETCH_SIZE / WRITE_SIZE, corrected as the micro-architecture
    guide prescribes)."""
    per_eval = profiled("hbm_bytes_per_eval")
    return None if per_eval is None else per_eval * evals


def roofline(bytes_alg, flops_alg, evals, kern_ms, measured):
    """Position of the sweep kernel on the FP64 roofline.  achieved / frac: the FP64 flops the kernel
    executes (SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 and SQ_INSTS_VALU_MFMA_MOPS_F64 of the committed profile,
    per evaluation) at the launch duration measured now.  alg_*: the unfused gate list's work at the
    same duration."""
    per_s = evals / (kern_ms * 1e-3)
    alg_tflops = flops_alg * per_s / 1e12
    alg_gbs = bytes_alg * per_s / 1e9
    ex_valu, ex_mfma = profiled("fp64_flops_per_eval"), profiled("mfma_f64_flops_per_eval") or 0.0
    ex = None if ex_valu is None else ex_valu + ex_mfma
    ex_tflops = None if ex is None else ex * per_s / 1e12
    r = {
        "kernel": profiled_str("kernel") or "k_reg_psr",
        # the FP64 pipe: v_fma_f64 / v_mul_f64 and, for the rotations on lane bits 4 and 5, v_mfma_f64_4x4x4
        # (on gfx950 the FP64 matrix peak equals the FP64 vector peak, 78.6 TFLOP/s, and the two share the pipe)
        "bound": "valu_fp64",
        "achieved": ex_tflops,
        "peak": FP64_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": None if ex_tflops is None else ex_tflops / FP64_PEAK_TFLOPS,
        "traffic": hbm_traffic_per_launch(evals),
        "kernel_ms_per_launch": kern_ms,
        "evals_per_launch": evals,
        "executed_fp64_flops_per_eval": ex,
        "executed_fp64_flops_per_eval_matrix_unit": ex_mfma,
        # share of the launch the FP64 pipe is issuing: 4 cycles per vector instruction, 16 per matrix instruction
        "fp64_issue_frac": None if ex is None else (profiled("fp64_wave_instr_per_eval") * 4.0 +
                                                    (profiled("mfma_f64_wave_instr_per_eval") or 0.0) * 16.0) * per_s / (1024 * 2.4e9),
        "profile_source": profiled_str("source"),
        # the submitted, unfused gate list priced the same way (ratios above 1: work the engine never does)
        "alg_flops_per_eval": flops_alg,
        "alg_bytes_per_eval": bytes_alg,
        "alg_achieved": alg_tflops,
        "alg_frac": alg_tflops / FP64_PEAK_TFLOPS,
        "alg_over_executed": None if ex is None else flops_alg / ex,
        "alg_lds_frac": alg_gbs / LDS_PEAK_GBS,
        "alg_hbm_frac": alg_gbs / HBM_PEAK_GBS,
        "hbm_traffic_frac": None if hbm_traffic_per_launch(evals) is None
        else hbm_traffic_per_launch(evals) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
    }
    if measured:
        r["measured_peaks"] = measured
        if "fp64_fma_tflops" in measured and ex_tflops is not None:
            r["frac_of_measured_peak"] = ex_tflops / measured["fp64_fma_tflops"]
    return r


def state_passes(reset=False):
    """(read-and-write, read-only) passes over a state that the HBM engine has launched since the last reset."""
    import ctypes as C

    from vqa_amd import _lib

    rw, ro = C.c_int64(0), C.c_int64(0)
    _lib.check(_lib.lib().vqa_state_pass_counters(C.byref(rw), C.byref(ro), int(reset)))
    return rw.value, ro.value


def hbm_profile(name):
    """Per-sample figures of an HBM-engine configuration from the committed profile, or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_configs.json")) as f:
            return json.load(f)[name]
    except (OSError, KeyError, ValueError):
        return None


def profiled_str(key):
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            return str(json.load(f)[key])
    except (OSError, KeyError, ValueError):
        return None


def unpruned_step(device, n, d, d_theta, d_in, d_y, w, z0, B, d_out):
    """One timed step on a fresh circuit object created with VQA_NO_CONE set (the library reads the
    switch when it first analyses a circuit for a set of measured qubits)."""
    import torch

    from vqa_amd import LearningCircuit, create_ibm_HEA_ansatz
    from vqa_amd.distributed import sharded_mse_gradient

    os.environ["VQA_NO_CONE"] = "1"
    try:
        c = create_ibm_HEA_ansatz(n, d, seed=0, circuit_cls=lambda nq: LearningCircuit(nq, device=device))
        sharded_mse_gradient(c, d_theta, d_in[:512], d_y[:512], w, z0, n_global=512, out_expect=d_out[:512])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sharded_mse_gradient(c, d_theta, d_in, d_y, w, z0, n_global=B, out_expect=d_out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        del os.environ["VQA_NO_CONE"]
    return {"gradients_per_sec": B / dt, "ms_per_step": 1e3 * dt, "steps": 1}


def spawn_ranks(n):
    """`python bench.py --gpus N` started plainly: N child processes of this same command, one rank
    per GPU, rendezvous on 127.0.0.1; rank 0's JSON line goes to the inherited stdout.  Returns the
    largest exit code."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    return max(p.wait() for p in procs)


def other_configs(device):
    """Full parameter-shift sweeps of the two other QNN configurations that fit one GPU in seconds,
    measured after the timed region (one GPU, <Z0>, inputs resident): not the headline value."""
    import torch

    from vqa_amd import LearningCircuit, create_ibm_HEA_ansatz, z_observable

    out = {}
    for name, n, d, B, n_feat, reps in (("sin(x) regressor: 6 qubits, depth 5, batch 4096", 6, 5, 4096, 1, 5),
                                        ("20-qubit QNN: depth 8 (HBM engine), batch 1024", 20, 8, 1024, 20, 1)):
        rng = np.random.default_rng(1)
        c = create_ibm_HEA_ansatz(n, d, seed=0, circuit_cls=lambda nq: LearningCircuit(nq, device=device))
        x = c.upload_inputs(rng.uniform(0.0, np.pi, (B, n_feat)))
        th = torch.from_numpy(rng.uniform(0.0, 2 * np.pi, c.n_theta)).to(x.device)
        obs = [z_observable(n, 0)]
        c.psr_full(th, x, n_sel=min(B, 2), obs_list=obs)
        torch.cuda.synchronize()
        state_passes(reset=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            c.psr_full(th, x, n_sel=B, obs_list=obs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        rw, ro = state_passes()
        K = 2 * c.n_theta + 1
        out[name] = {"gradients_per_sec": B / dt, "circuit_evals_per_sec": B * K / dt, "n_params": c.n_theta,
                     "ms_per_call": 1e3 * dt}
        if n == 20:
            # FP64 flops the tiled passes execute per sample of this configuration (PMC SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 of
            # tools/bench_hbm_batch.py 20 8 16, 11-qubit tiles) at the rate measured here, against the vector FP64 peak
            # ... the flops per state pass come from the committed profile (profiles/hbm_configs.json, written by
            # tools/hbm_configs_profile.py from the PMC passes of tools/prof_hbm_batch.sh); the passes over a state that
            # this call made are counted by the library now: a plan that changed since the profile shows as a
            # different count, and the figure is then left out
            prof = hbm_profile("hea20_d8")
            nbytes = (2 * rw + ro) * 16.0 * 2 ** n / reps
            roof = {"kernel": "k_tile_pass", "state_passes_per_sample": (rw + ro) / reps / B,
                    "traffic": nbytes, "hbm_achieved_GBs": nbytes / dt / 1e9, "hbm_frac": nbytes / dt / 8e12}
            if prof and abs(prof["state_passes_per_sample"] - (rw + ro) / reps / B) < 0.5:
                flops = prof["fp64_flops_per_sample"]
                roof.update({"bound": "valu_fp64", "achieved": flops * B / dt / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                             "frac": flops * B / dt / 78.6e12, "executed_fp64_flops_per_sample": flops,
                             "profile_source": prof.get("source")})
                if prof.get("hbm_bytes_per_sample"):
                    # HBM bytes of the same profile (FETCH_SIZE + WRITE_SIZE passes) at the rate measured here
                    roof.update({"traffic_pmc": prof["hbm_bytes_per_sample"] * B,
                                 "hbm_pmc_GBs": prof["hbm_bytes_per_sample"] * B / dt / 1e9,
                                 "hbm_pmc_frac": prof["hbm_bytes_per_sample"] * B / dt / 8e12})
            else:
                roof.update({"bound": "valu_fp64", "achieved": None, "peak": 78.6, "unit": "TFLOP/s", "frac": None,
                             "profile_source": "no committed profile of this plan"})
            out[name]["roofline"] = roof
        del c, x
        torch.cuda.empty_cache()
    out["28-qubit VQE: RY + CZ ring depth 6, Ising ring (56 Pauli terms), full gradient"] = vqe_config(device)
    out["Iris classifier: 4 qubits, depth 3 (RY/RZ + CZ ring), 105 training samples, two-bit marginal"] = iris_config(device)
    return out


def iris_config(device, B=105, reps=50):
    """The classifier's circuit at the size of the Iris training set: one full parameter-shift sweep with the
    marginal of qubits 0 and 1 (what the class probabilities are read from) per call; a call this small
    is bound by the launch, not by the arithmetic."""
    import torch

    from vqa_amd import LearningCircuit
    from vqa_amd.qml_utils import create_ry_rz_cz_ansatz

    rng = np.random.default_rng(0)
    c = create_ry_rz_cz_ansatz(4, 3, seed=0, circuit_cls=lambda nq: LearningCircuit(nq, device=device))
    x = c.upload_inputs(rng.uniform(0.0, np.pi, (B, 4)))
    th = torch.from_numpy(rng.uniform(0.0, 2 * np.pi, c.n_theta)).to(x.device)
    c.psr_full(th, x, n_sel=B, n_marg_bits=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        c.psr_full(th, x, n_sel=B, n_marg_bits=2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    K = 2 * c.n_theta + 1
    return {"gradients_per_sec": B / dt, "circuit_evals_per_sec": B * K / dt, "n_params": c.n_theta, "ms_per_call": 1e3 * dt}


def vqe_config(device, n=28, d=6):
    """The VQE-shaped configuration: one full parameter-shift gradient of a 2n-term Pauli sum (exact
    expectation values) through the Pauli-sum target object; the first call grows the library's
    workspace and is reported beside the timed second one."""
    import torch

    from vqa_amd import LearningCircuit, Observable
    from vqa_amd.qml_utils import create_ry_cz_ansatz
    from vqa_amd.vqe_target import PauliSumVQE

    c = create_ry_cz_ansatz(n, d, seed=4, circuit_cls=lambda nq: LearningCircuit(nq, device=device))
    h = Observable(n)
    for i in range(n):
        h.add_operator(1.0, f"Z {i} Z {(i + 1) % n}")
        h.add_operator(1.0, f"X {i}")
    theta = np.random.default_rng(4).uniform(-np.pi, np.pi, c.n_theta)
    shots = np.full(c.n_theta, 1000)

    def gradient_times(circuit, calls):
        vqe = PauliSumVQE(n, circuit, h, min_approx_shots=-10)
        times, g = [], None
        for _ in range(calls):
            state_passes(reset=True)
            t0 = time.perf_counter()
            g = vqe.iEval(theta, shots)["grad"]
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        return times, g

    # (1) as the library runs it by itself: the 56 terms dealt into groups by their backward light cones (13 and 14
    # qubits wide behind six CZ rings), every group on its own reduced circuit on the register engine
    times, g = gradient_times(c, 6)
    rw0, ro0 = state_passes()
    res = {"seconds_per_gradient": float(np.median(times[1:])), "first_call_seconds": times[0], "n_params": int(len(theta)),
           "shifted_evaluations_per_sec": 2 * len(theta) / float(np.median(times[1:])), "grad_norm": float(np.linalg.norm(g)),
           "path": "light-cone groups on the register engine" if rw0 + ro0 == 0 else "whole register"}
    # (2) the same gradient on the whole 28-qubit register (light-cone splitting switched off): the streaming
    # engine's tiled passes over 4 GiB states, priced against HBM.  Passes over a state that the timed gradient made,
    # counted by the library (those that read and write a state, and the measuring ones that only read) x 16 bytes
    # x 2^n amplitudes, against the 8 TB/s of HBM
    os.environ["VQA_NO_CONE_SPLIT"] = "1"
    try:
        c2 = create_ry_cz_ansatz(n, d, seed=4, circuit_cls=lambda nq: LearningCircuit(nq, device=device))
        wtimes, gw = gradient_times(c2, 2)
        rw, ro = state_passes()
    finally:
        del os.environ["VQA_NO_CONE_SPLIT"]
    del c, c2
    torch.cuda.empty_cache()
    nbytes = (rw * 2 + ro) * 16.0 * 2 ** n
    res["whole_register"] = {
        "seconds_per_gradient": wtimes[1], "first_call_seconds": wtimes[0],
        "max_abs_difference_of_the_two_gradients": float(np.abs(g - gw).max()),
        "roofline": {"kernel": "k_tile_pass", "bound": "hbm", "achieved": nbytes / wtimes[1] / 1e9, "peak": 8000.0,
                     "unit": "GB/s", "frac": nbytes / wtimes[1] / 8e12, "traffic": nbytes, "state_passes_read_write": rw,
                     "state_passes_read_only": ro}}
    return res


def cpu_baseline(circ, theta, ia, P, seconds):
    """The CPU oracle (a port of the qulacs-convention arithmetic; the reference's
    own CPU path, qulacs, is not installed) timed on this host's cores on a
    bounded sample of the same workload."""
    from oracle import cpu_oracle

    K = 2 * P + 1
    threads = int(os.environ.get("OMP_NUM_THREADS", 0)) or min(os.cpu_count() or 1, 16)
    z0 = ([1.0], [1])
    si, sp, sv = cpu_oracle.psr_lists(1, P)
    probe = 4 * threads
    t0 = time.perf_counter()
    cpu_oracle.eval_batch(circ, theta, ia, si[:probe], sp[:probe], sv[:probe], obs_list=[z0], n_threads=threads)
    per = (time.perf_counter() - t0) / probe
    n_eval = int(max(probe, min(seconds / per, 64 * K)))
    n_s = (n_eval + K - 1) // K
    si, sp, sv = cpu_oracle.psr_lists(n_s, P)
    si, sp, sv = si[:n_eval], sp[:n_eval], sv[:n_eval]
    t0 = time.perf_counter()
    cpu_oracle.eval_batch(circ, theta, ia, si, sp, sv, obs_list=[z0], n_threads=threads)
    dt = time.perf_counter() - t0
    # the reference's own CPU path is qulacs + skqulacs: timed when this box happens to have them (never installed here)
    ref_timing = None
    try:
        import qulacs  # noqa: F401
        import skqulacs  # noqa: F401

        try:
            ref_timing = reference_loop(circ, theta, P, min(seconds, 10.0))
            ref_note = "qulacs %s present: the reference's own loop is timed beside the port" % getattr(qulacs, "__version__", "?")
        except Exception as e:   # an API this script does not know: say so, keep the port's number
            ref_note = f"qulacs present but its loop could not be run ({type(e).__name__}: {e}); the baseline below is the port"
    except ImportError:
        ref_note = "qulacs / skqulacs not present on this box: the reference's CPU path cannot be timed, the port is"
    return {
        "reference_loop": ref_timing,
        "value": n_eval / dt / K,
        "unit": "gradients/s",
        "circuit_evals_per_sec": n_eval / dt,
        "cores": threads,
        "kind": "port",
        "reference_cpu_path": ref_note,
        "sample": f"{n_eval} circuit evaluations (first {n_s} samples of the same batch, same shifts), "
                  f"{threads} OpenMP threads, {dt:.1f} s",
    }


def reference_loop(circ, theta, P, seconds):
    """The reference's own evaluation loop -- skqulacs LearningCircuit.update_parameters, then per sample run(x) and
    Observable.get_expectation_value (vqa/QNN/myQNNs_base.py:916-924 under 1381-1394) -- on the same circuit, one
    process, for about `seconds`; only reached when qulacs and skqulacs are importable on this box."""
    from qulacs import Observable as QObservable
    from qulacs.gate import CZ
    from skqulacs.circuit import LearningCircuit as SkCircuit

    from vqa_amd import create_ibm_HEA_ansatz

    class Adapter(SkCircuit):
        def add_CZ_gate(self, a, b):
            self.add_gate(CZ(a, b))

    n = circ.n_qubit
    d = P // (3 * n) - 1
    ref = create_ibm_HEA_ansatz(n, d, seed=0, circuit_cls=Adapter)
    obs = QObservable(n)
    obs.add_operator(1.0, "Z 0")
    rng = np.random.default_rng(2)
    x = rng.uniform(0.0, np.pi, (64, n))
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        k = done % (2 * P + 1)
        th = np.array(theta, dtype=float)
        if k:
            th[(k - 1) // 2] += 0.5 * np.pi if k % 2 else -0.5 * np.pi
        ref.update_parameters(th)
        for xi in x:
            obs.get_expectation_value(ref.run(xi))
        done += len(x)
    dt = time.perf_counter() - t0
    return {"circuit_evals_per_sec": done / dt, "gradients_per_sec": done / dt / (2 * P + 1), "cores": 1,
            "sample": f"{done} circuit evaluations in {dt:.1f} s, one process"}


if __name__ == "__main__":
    main()
