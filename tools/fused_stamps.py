#!/usr/bin/env python3
"""Diagnostic (-DPL_STAMPS build): where a chain-step of each role of the fused forward launch goes (lstm_fused.hip).
usage: fused_stamps.py [B [T]]   (reads the block -> role table the way the host builds it: roles by the stamp pattern)
B <= 16 (the 16-row roles of lstm_fused16.h): the LSTM roles' phases are wait / operands->LDS / MFMA / cell / store issue / drain+flag
(forward) and wait / ingest / cell+dA image / tiles / drain+flag (backward); the product roles keep the labels below."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PAULE_HIP_LIB"] = os.path.join(ROOT, "paule_amd", "csrc", "libpaule_hip_stamps.so")
os.environ.setdefault("PL_STAMP_FILE", os.path.join(ROOT, "gpurun_out", "stamps"))
os.environ["PAULE_HIP_FUSED"] = os.environ.get("PAULE_HIP_FUSED", "1")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
import numpy as np  # noqa: E402
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 300
if B <= 16:
    os.environ.pop("PAULE_HIP_FUSED", None)
wl = synthetic.make_workload(B, T, "A")
eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
print("plan:", eng.plan_info())
eng.set_targets(wl.target_mel, wl.target_semvec)
eng.set_cp(wl.cp0)
eng.step(3)
eng.synchronize()
raw = np.fromfile(os.environ["PL_STAMP_FILE"] + ".sweep", dtype=np.uint64).reshape(2, 256, 8).astype(np.float64) * 0.01
labels = ["operands->LDS", "MFMA", "flag answer+prefetch", "cell/epilogue+stores", "drain+flag", "blocking wait", "prefetch issue (bwd)", "-"]
for d, name in ((0, "forward"), (1, "backward")):
    blk = raw[d]
    tot = blk.sum(axis=1)
    used = np.flatnonzero(tot > 0)
    if used.size == 0:
        continue
    print(f"=== {name} launch: {used.size} stamped workgroups; total stamped time per workgroup min {tot[used].min():.0f} / median {np.median(tot[used]):.0f} / max {tot[used].max():.0f} us")
    # group workgroups by their total time profile: roles differ in the share of phase 3 / 1
    key = np.round(blk[used] / tot[used, None], 1)
    if B <= 16:
        print("   (16-row LSTM roles: phases 0..5 = " + ("wait / operands->LDS / MFMA / cell / store issue / drain+flag" if d == 0 else "wait / ingest / cell+dA image / tiles / drain+flag") + ")")
    _, inv = np.unique(key, axis=0, return_inverse=True)
    for k in np.unique(inv):
        sel = used[inv == k]
        med = np.median(blk[sel], axis=0)
        print(f"  {sel.size:3d} workgroups (blocks {sel[:6].tolist()}...): " + "  ".join(f"{lab} {v:.0f}" for lab, v in zip(labels, med) if lab != "-") + f"  | sum {med.sum():.0f} us")
