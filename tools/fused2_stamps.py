#!/usr/bin/env python3
"""Diagnostic (-DPL_STAMPS build): phases of the LSTM roles of the two-per-CU forward launch (lstm_fused2.hip), blocks 0..255."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PAULE_HIP_LIB"] = os.path.join(ROOT, "paule_amd", "csrc", "libpaule_hip_stamps.so")
os.environ.setdefault("PL_STAMP_FILE", os.path.join(ROOT, "gpurun_out", "stamps"))
os.environ["PAULE_HIP_FUSED_OCC2"] = os.environ.get("PAULE_HIP_FUSED_OCC2", "1")
os.environ["PAULE_HIP_DEBUG"] = "stop_after_fwd"
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
raw = np.fromfile(os.environ["PL_STAMP_FILE"] + ".sweep", dtype=np.uint64).reshape(512, 8)   # the forward launch's blocks 0 .. 511 (its stamps run into the backward half: no backward launch here)
tag = raw[:, 7].astype(np.int64)
blk = raw[:, :7].astype(np.float64) * 0.01
labels = ["wait flags", "operands", "MFMA", "cell+staging", "h store issue", "drain+flag", "stash+landing"]
role, rset = (tag & 255) - 1, (tag >> 8) & 255
tot = blk.sum(axis=1)
names = {0: "predictor recurrence", 1: "mel head", 2: "embedder layer 1", 3: "projection for layer 2", 4: "embedder layer 2"}
for r in sorted(set(role[role >= 0].tolist())):
    sel = np.flatnonzero((role == r) & (tot > 0))
    if sel.size == 0:
        print(f"role {r} ({names.get(r, '?')}): {int((role == r).sum())} workgroups, no stamps (product role)")
        continue
    med = np.median(blk[sel], axis=0)
    print(f"role {r} ({names.get(r, '?')}): {sel.size} workgroups, total min {tot[sel].min():.0f} / median {np.median(tot[sel]):.0f} / max {tot[sel].max():.0f} us: " +
          "  ".join(f"{lab} {v:.0f}" for lab, v in zip(labels, med)))
    for s_ in sorted(set(rset[sel].tolist())):
        ss = sel[rset[sel] == s_]
        print(f"     set {s_}: {ss.size} workgroups, total median {np.median(tot[ss]):.0f} us, wait flags {np.median(blk[ss, 0]):.0f}, operands {np.median(blk[ss, 1]):.0f}, MFMA {np.median(blk[ss, 2]):.0f}")
