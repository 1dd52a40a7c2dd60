"""Shared helpers for the tests: golden loading and config reconstruction (data only)."""
import os
from argparse import Namespace

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_config(G):
    cfg = {}
    for k in G.files:
        if k.startswith("config/"):
            v = G[k]
            v = v.item() if v.shape == () else v
            cfg[k[len("config/"):]] = v
    return Namespace(**cfg)


def golden_params(G, prefix="sd/", requires_grad=False):
    P = {}
    for k in G.files:
        if k.startswith(prefix):
            t = torch.from_numpy(np.array(G[k]))
            P[k[len(prefix):]] = t
    # decoder embeddings alias the shared table (sent_models.py:70)
    for k in list(P):
        if k.endswith("decoder.embedding.weight"):
            P[k] = P["embedding.weight"]
    if requires_grad:
        for k in P:
            if not k.endswith("decoder.embedding.weight"):
                P[k].requires_grad_(True)
    return P


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))
