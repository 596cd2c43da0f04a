#!/usr/bin/env python3
"""Diagnostic (-DPL_STAMPS build): phase times of the trailing update role of the predictor's backward sweep (lstm_persist_rs.hip: rs_tail_role),
per pair of frames, from thread 0 of each role workgroup (cfg3: workgroups 240 .. 255 of the launch)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PAULE_HIP_LIB"] = os.path.join(ROOT, "paule_amd", "csrc", "libpaule_hip_stamps.so")
os.environ.setdefault("PL_STAMP_FILE", os.path.join(ROOT, "gpurun_out", "stamps"))
import numpy as np  # noqa: E402
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

B, T = 256, 300
wl = synthetic.make_workload(B, T, "A")
eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
info = eng.plan_info()
eng.set_targets(wl.target_mel, wl.target_semvec)
eng.set_cp(wl.cp0)
eng.step(3)
eng.synchronize()
raw = np.fromfile(os.environ["PL_STAMP_FILE"] + ".sweep", dtype=np.uint64).reshape(2, 256, 8).astype(np.float64) * 0.01
n_tail, n_pf = info["bwd_tail"], info["bwd_prefetchers"]
first = 184 + n_pf
blk = raw[1][first:first + n_tail]
pairs = (T + 1) // 2 / max(1, n_tail // 8)
print(f"trailing update role: {n_tail} workgroups behind {n_pf} prefetchers; {pairs:.0f} pairs of frames each; us per pair, median / max over workgroups")
for i, lab in enumerate(["wait for the marks", "partial tiles loaded + summed", "barrier", "update (smoothness, Adam, projection)"]):
    v = blk[:, i] / pairs
    print(f"  {lab:40s} {np.median(v):7.2f} {v.max():7.2f}")
print(f"  {'sum':40s} {np.median(blk[:, :4].sum(axis=1) / pairs):7.2f}   total per workgroup {np.median(blk[:, :4].sum(axis=1)):.0f} us")
