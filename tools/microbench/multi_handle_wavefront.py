import os, sys
sys.path.insert(0, "/root/repo")
import torch
from paule_amd import synthetic
from paule_amd.engine import HipPlanner
keep = []
for i, (B, T, st, dt) in enumerate([(5, 61, "B", "bf16"), (2, 64, "A", "f32"), (40, 36, "B", "bf16"), (3, 37, "B", "f32"), (140, 24, "B", "bf16"), (5, 61, "B", "bf16")]):
    wl = synthetic.make_workload(B, T, st)
    e = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype=dt)
    e.set_targets(wl.target_mel, wl.target_semvec); e.set_cp(wl.cp0)
    e.step(3); e.synchronize()
    keep.append(e)
    print("engine", i, "ok", flush=True)
for e in keep:
    e.step(2); e.synchronize()
print("all ok")
