#!/usr/bin/env python3
"""Stress the in-launch exchange of the persistent LSTM sweeps: many forward(+backward) sweeps back to back,
counting bounded-wait timeouts.  usage: sweep_stress.py [n_rounds] [mode: pred|step|burst] [batch] [dtype]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PAULE_HIP_SPIN_MS", "50")
import torch  # noqa: E402
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
mode = sys.argv[2] if len(sys.argv) > 2 else "pred"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
wl = synthetic.make_workload(B, 300, "A")
eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=300, objective="acoustic_semvec", dtype=dtype)
eng.set_targets(wl.target_mel, wl.target_semvec)
eng.set_cp(wl.cp0)
fails, t0 = 0, time.time()
for i in range(n):
    if mode == "pred":
        eng.get_pred()
    elif mode == "burst":
        eng.step(10, return_loss=False)
    else:
        eng.step(1, return_loss=False)
    try:
        eng.synchronize()
    except ValueError:
        fails += 1
print(f"zero_mode={os.environ.get('PAULE_HIP_ZERO_MODE', '0')} mode={mode} batch={B} dtype={dtype}: {fails} timeouts in {n} rounds "
      f"({(time.time() - t0) / n * 1e3:.2f} ms per round)", flush=True)
