#!/usr/bin/env python3
"""Interleaved A/B timing of engine variants in ONE process on ONE device (MI355X guide, methodology rule 24).
usage: ab_bench.py VAR=a,b[,c] [rounds] [iters]      e.g.  ab_bench.py PAULE_HIP_XCD_FAST=0,1 8 10
       ab_bench.py "A=0,B=2/A=1,B=3" [rounds] [iters]  variants separated by "/", each a list of assignments
env: AB_BATCH (256), AB_FRAMES (300), AB_DTYPE (bf16), AB_GRAPH (1), AB_SET (A), AB_OBJECTIVE (acoustic_semvec)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

if "/" in sys.argv[1]:
    variants = [dict(a.split("=") for a in v.split(",")) for v in sys.argv[1].split("/")]
    var, vals = "variant", [",".join(f"{k}={x}" for k, x in v.items()) for v in variants]
else:
    var, vals = sys.argv[1].split("=")
    vals = vals.split(",")
    variants = [{var: v} for v in vals]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cfg = dict(batch=int(os.environ.get("AB_BATCH", 256)), frames=int(os.environ.get("AB_FRAMES", 300)), objective=os.environ.get("AB_OBJECTIVE", "acoustic_semvec"), dtype=os.environ.get("AB_DTYPE", "bf16"))
wl = synthetic.make_workload(cfg["batch"], cfg["frames"], os.environ.get("AB_SET", "A"))
engines = []
for v in variants:
    os.environ.update(v)
    e = HipPlanner(wl.pred_sd, wl.emb_sd, batch=cfg["batch"], n_frames=cfg["frames"], objective=cfg["objective"], dtype=cfg["dtype"],
                   use_graph=os.environ.get("AB_GRAPH", "1") != "0")
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
for e in engines:
    e.synchronize()
for v, t in zip(vals, times):
    print(f"{var}={v}: median {np.median(t):.3f} ms/iter  min {np.min(t):.3f}  max {np.max(t):.3f}  ({rounds} rounds x {iters} iters)")
finals = [e.get_cp().cpu().numpy() for e in engines]
for v, f in zip(vals[1:], finals[1:]):
    print(f"final CP {var}={vals[0]} vs {v}: max|diff| {np.abs(f - finals[0]).max():.3e}, identical {np.array_equal(f, finals[0])}")
