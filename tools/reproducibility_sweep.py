#!/usr/bin/env python3
"""Run-to-run reproducibility across the schedules the planner picks: for every case N fresh engines run the same plan and the final
CP, the last losses and dL/dCP are hashed; anything that differs between two engines is reported.  (tools/reproducibility_check.py
locates a difference inside the buffers; this one only finds out WHERE to look.)   usage: reproducibility_sweep.py [N=5]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
CASES = [
    ("cfg5_128 (32-row fused launches, T = 2000)", dict(B=128, T=2000, set="A", dtype="bf16", objective="acoustic_semvec", iters=2)),
    ("B = 64 bf16 (32-row fused launches)", dict(B=64, T=300, set="A", dtype="bf16", objective="acoustic_semvec", iters=3)),
    ("cfg3 (fused forward + streamed sweeps)", dict(B=256, T=300, set="A", dtype="bf16", objective="acoustic_semvec", iters=3)),
    ("cfg3_setB", dict(B=256, T=300, set="B", dtype="bf16", objective="acoustic_semvec", iters=3)),
    ("B = 40 bf16 (chunk pipelines)", dict(B=40, T=300, set="A", dtype="bf16", objective="acoustic_semvec", iters=3)),
    ("B = 16 x 2000 set B (chunk pipelines, stacked)", dict(B=16, T=2000, set="B", dtype="bf16", objective="acoustic_semvec", iters=2)),
    ("cfg5 (16-row fused launches)", dict(B=16, T=2000, set="A", dtype="bf16", objective="acoustic_semvec", iters=3)),
    ("cfg2 (f32 sweeps)", dict(B=64, T=300, set="A", dtype="f32", objective="acoustic", iters=3)),
    ("cfg1 (f32, one row)", dict(B=1, T=300, set="A", dtype="f32", objective="acoustic_semvec", iters=3)),
    ("B = 256 f32 (chains kernels)", dict(B=256, T=100, set="A", dtype="f32", objective="acoustic_semvec", iters=2)),
    ("B = 700 bf16 (groups in passes)", dict(B=700, T=100, set="A", dtype="bf16", objective="acoustic_semvec", iters=2)),
]
bad = 0
for name, c in CASES:
    wl = synthetic.make_workload(c["B"], c["T"], c["set"])
    seen = {"cp": set(), "loss": set(), "dX": set()}
    for _ in range(N):
        eng = HipPlanner(wl.pred_sd, wl.emb_sd if c["objective"] != "acoustic" else None, batch=c["B"], n_frames=c["T"], objective=c["objective"], dtype=c["dtype"])
        eng.set_targets(wl.target_mel, wl.target_semvec if c["objective"] != "acoustic" else None)
        eng.set_cp(wl.cp0)
        loss = eng.step(c["iters"]).cpu().numpy()
        eng.synchronize()
        seen["cp"].add(hashlib.md5(eng.get_cp().cpu().numpy().tobytes()).hexdigest())
        seen["loss"].add(hashlib.md5(loss.tobytes()).hexdigest())
        seen["dX"].add(hashlib.md5(eng.debug_read("dX").float().cpu().numpy().tobytes()).hexdigest())
        eng.close()
    ok = all(len(v) == 1 for v in seen.values())
    bad += 0 if ok else 1
    print(f"{'reproducible    ' if ok else 'NOT REPRODUCIBLE'}  {name}: distinct results over {N} engines: " + ", ".join(f"{k} {len(v)}" for k, v in seen.items()), flush=True)
sys.exit(1 if bad else 0)
