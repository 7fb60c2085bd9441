#!/usr/bin/env python3
"""Where the host spends an oracle call of the Iris classifier (cProfile of 300 MSE_grad_eval calls with device statistics).
usage: python tools/prof_iris_host.py"""
import os, sys, cProfile, pstats
import numpy as np, torch
from sklearn.datasets import load_iris
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit
from vqa_amd.qml_utils import create_ry_rz_cz_ansatz
from vqa_amd.qnn_classifier import myQNN_classifier
data = load_iris()
x = (data.data - data.data.min(0)) / (data.data.max(0) - data.data.min(0)) * np.pi
rng = np.random.default_rng(0)
perm = rng.permutation(len(x)); tr, te = perm[:105], perm[105:]
c = create_ry_rz_cz_ansatz(4, 3, seed=0, circuit_cls=LearningCircuit)
q = myQNN_classifier(4, c, x[tr], data.target[tr], x[te], data.target[te])
p = rng.uniform(0, 2 * np.pi, q.n_params)
q.device_statistics = True
B = 32
shots = np.full(q.n_params, 200 * B)
for _ in range(5): q.MSE_grad_eval(p, shots, mini_batch_size=B)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): q.MSE_grad_eval(p, shots, mini_batch_size=B)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
