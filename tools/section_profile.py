#!/usr/bin/env python3
"""Where the sweep kernel's time goes, by ablation: runs the same parameter-shift
sweep on the ablation build (make -C vqa_amd/csrc libvqahip_prof.so) with parts
of the kernel switched off (VQA_ABLATE bit mask; results are wrong on purpose)
and prints the time each part accounts for.

usage: python tools/section_profile.py [n_qubits depth batch]
(spawns one child process per variant; needs a GPU)
"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
VARIANTS = [(0, "full kernel"), (1, "no transpositions"), (2, "no register-bit rotations"), (4, "no lane-bit rotations"),
            (8, "no diagonal stages"), (16, "no measurement"), (64, "no checkpoint loads"),
            (128, "no table patching"), (1024, "pair measurement: no state load"), (15, "no gates at all"), (95, "gates, measurement, loads off"),
            (223, "... and no patching"), (479, "... and no phase loop")]

if os.environ.get("VQA_SECTION_CHILD"):
    import numpy as np
    import torch

    sys.path.insert(0, ROOT)
    from vqa_amd import LearningCircuit, create_ibm_HEA_ansatz, z_observable

    n, d, B = (int(v) for v in sys.argv[1:4])
    circ = create_ibm_HEA_ansatz(n, d, circuit_cls=LearningCircuit)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(circ.input_angles(rng.uniform(0, np.pi, (B, n)))).cuda()
    theta = torch.from_numpy(rng.uniform(0, 2 * np.pi, circ.n_theta)).cuda()
    obs = [z_observable(n, 0)]
    circ.psr_full(theta, x, n_sel=B, obs_list=obs)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    circ.psr_full(theta, x, n_sel=B, obs_list=obs)
    ev1.record()
    torch.cuda.synchronize()
    if os.environ.get("VQA_STAMPS"):
        import ctypes as C
        from vqa_amd import _lib
        buf = (C.c_ulonglong * 16)()
        _lib.lib().vqa_debug_stamps(buf, 0)
        v = [float(b) for b in buf]
        runs = B * circ.n_theta * 2  # two sweeps were run
        names = ["per sample: tables + run records", "patch + barriers", "state load + phase loop", "measurement",
                 "", "", "", "", "", "phase: head", "phase: transposition", "phase: register-bit rotations",
                 "phase: lane-bit rotations", "phase: diagonal stage"]
        for nm, c in zip(names, v):
            if nm:
                print(f"   {nm:34s} {c / runs:9.0f} cycles per run")
    print(ev0.elapsed_time(ev1))
    sys.exit(0)

args = sys.argv[1:4] or ["12", "10", "4096"]
if os.environ.get("VQA_STAMPS_ONLY"):
    VARIANTS = VARIANTS[:1]
base = None
for mask, name in VARIANTS:
    env = dict(os.environ, VQA_SECTION_CHILD="1", VQA_ABLATE=str(mask),
               VQA_HIP_LIBRARY=os.path.join(ROOT, "vqa_amd", "csrc", "libvqahip_prof.so"))
    out = subprocess.run([sys.executable, os.path.abspath(__file__)] + args, env=env, capture_output=True, text=True, timeout=300)
    if out.returncode != 0:
        print(name, "FAILED", out.stderr[-400:])
        sys.exit(1)
    if mask == 0:
        env2 = dict(env, VQA_STAMPS="1")
        o2 = subprocess.run([sys.executable, os.path.abspath(__file__)] + args, env=env2, capture_output=True, text=True, timeout=300)
        print("wave-0 clock stamps (100 MHz counter units x clock ratio; relative sizes matter):")
        print("\n".join(o2.stdout.strip().split("\n")[:-1]))
    ms = float(out.stdout.strip().split()[-1])
    base = base or ms
    print(f"{name:32s} {ms:9.1f} ms   saves {100 * (base - ms) / base:5.1f} %", flush=True)
