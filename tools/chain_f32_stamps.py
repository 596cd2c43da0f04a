#!/usr/bin/env python3
"""Diagnostic (-DPL_STAMPS build): where a chain-step of the f32 chains kernels goes (lstm_chain_f32.hip); objective `acoustic`,
so the last forward / backward launch is the predictive model's.  usage: chain_f32_stamps.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PAULE_HIP_LIB"] = os.path.join(ROOT, "paule_amd", "csrc", "libpaule_hip_stamps.so")
os.environ.setdefault("PL_STAMP_FILE", os.path.join(ROOT, "gpurun_out", "stamps"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
import numpy as np  # noqa: E402
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = 300
wl = synthetic.make_workload(B, T, "A")
eng = HipPlanner(wl.pred_sd, None, batch=B, n_frames=T, objective="acoustic", dtype="f32")
eng.set_targets(wl.target_mel, None)
eng.set_cp(wl.cp0)
eng.step(3)
eng.synchronize()
raw = np.fromfile(os.environ["PL_STAMP_FILE"] + ".sweep", dtype=np.uint64).reshape(2, 256, 8).astype(np.float64) * 0.01
groups = (B + 15) // 16
for d, name, labels in ((0, "forward", ["operands->LDS", "MFMA chain", "answer+prefetch issue", "cell+store issue", "drain+flag", "blocking wait"]),
                        (1, "backward", ["partial sum", "cell+dA+answer", "prefetch issue", "MFMA+tile stores", "drain+flag", "blocking wait"])):
    blk = raw[d]
    used = blk[blk.sum(axis=1) > 0]
    if not len(used):
        continue
    sets = len(used) // 46
    chain_steps = T * -(-groups // max(sets, 1))
    print(f"--- {name}: {len(used)} workgroups, ~{chain_steps} chain-steps each; per chain-step (us), median / max over workgroups")
    tot = 0.0
    for i, lab in enumerate(labels):
        v = used[:, i] / chain_steps
        tot += np.median(v)
        print(f"  {lab:24s} {np.median(v):6.2f} {v.max():6.2f}")
    print(f"  {'sum':24s} {tot:6.2f}")
