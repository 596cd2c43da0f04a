#!/usr/bin/env python3
"""Interleaved A/B of engine variants WITH the somatosensory models (Paule's sizes): soma_bench.py VAR=a,b [rounds] [iters]; env AB_BATCH, AB_DTYPE"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

var, vals = sys.argv[1].split("=")
vals = vals.split(",")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
B, dt = int(os.environ.get("AB_BATCH", 1)), os.environ.get("AB_DTYPE", "f32")
wl = synthetic.make_workload(B, 300, "A")
tube = synthetic.make_tube_models()
engines = []
for v in vals:
    os.environ[var] = v
    e = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=300, objective="acoustic_semvec", dtype=dt, tube_models=tube)
    e.set_targets(wl.target_mel, wl.target_semvec)
    e.set_cp(wl.cp0)
    e.step(2, return_loss=False)
    e.synchronize()
    engines.append(e)
times = [[] for _ in vals]
for r in range(rounds):
    for k, e in enumerate(engines):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.step(iters, return_loss=False)
        torch.cuda.synchronize()
        times[k].append((time.perf_counter() - t0) / iters * 1e3)
for v, t in zip(vals, times):
    print(f"{var}={v}: median {np.median(t):.3f} ms/iter  min {np.min(t):.3f}  max {np.max(t):.3f}")
finals = [e.get_cp().cpu().numpy() for e in engines]
for v, f in zip(vals[1:], finals[1:]):
    print(f"final CP {var}={vals[0]} vs {v}: max|diff| {np.abs(f - finals[0]).max():.3e}, identical {np.array_equal(f, finals[0])}")
