#!/usr/bin/env python3
"""Diagnostic (-DPL_STAMPS build): phases of the LSTM roles of the two-per-CU forward launch (lstm_fused2.hip), blocks 0..255."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PAULE_HIP_LIB"] = os.path.join(ROOT, "paule_amd", "csrc", "libpaule_hip_stamps.so")
os.environ.setdefault("PL_STAMP_FILE", os.path.join(ROOT, "gpurun_out", "stamps"))
os.environ["PAULE_HIP_FUSED_OCC2"] = os.environ.get("PAULE_HIP_FUSED_OCC2", "1")
os.environ["PAULE_HIP_STOP_AFTER_FWD"] = "1"
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
import numpy as np
from paule_amd import synthetic
from paule_amd.engine import HipPlanner
B, T = 256, 300
wl = synthetic.make_workload(B, T, "A")
eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", use_graph=False)
print("plan:", eng.plan_info())
eng.set_targets(wl.target_mel, wl.target_semvec)
eng.set_cp(wl.cp0)
eng.step(3)
eng.synchronize()
raw = np.fromfile(os.environ["PL_STAMP_FILE"] + ".sweep", dtype=np.uint64).reshape(2, 256, 8).astype(np.float64) * 0.01
labels = ["wait flags", "operands", "MFMA", "cell+staging", "h store issue", "drain+flag", "stash+landing", "-"]
blk = raw[0]
tot = blk.sum(axis=1)
used = np.flatnonzero(tot > 0)
print(f"{used.size} stamped workgroups among blocks 0..255; total per workgroup min {tot[used].min():.0f} / median {np.median(tot[used]):.0f} / max {tot[used].max():.0f} us")
key = np.round(blk[used] / tot[used, None], 1)
_, inv = np.unique(key, axis=0, return_inverse=True)
for k in np.unique(inv):
    sel = used[inv == k]
    med = np.median(blk[sel], axis=0)
    print(f"  {sel.size:3d} workgroups (blocks {sel[:6].tolist()}...): " + "  ".join(f"{lab} {v:.0f}" for lab, v in zip(labels, med) if lab != "-") + f"  | sum {med.sum():.0f} us")
