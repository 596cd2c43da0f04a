#!/usr/bin/env python3
"""Times pl_train_pred_step (continued learning of the predictive model, paule/paule.py:1372-1377) on the GPU.
usage: train_bench.py [engine_batch] [n_rows] [dtype] [steps]     defaults: 8 8 bf16 30 (reference: batch_size = 8)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
T = 300
wl = synthetic.make_workload(max(B, n), T, "A")
eng = HipPlanner(wl.pred_sd, None, batch=B, n_frames=T, objective="acoustic", dtype=dtype)
cp = wl.cp0[:n].float().cuda()
mel = wl.target_mel[:n].float().cuda()
for _ in range(3):
    eng.train_pred_step(cp, mel)
torch.cuda.synchronize()
t0 = time.perf_counter()
losses = [eng.train_pred_step(cp, mel) for _ in range(steps)]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
eng.synchronize()
H, I, M = 720, 30, 60
flops_fwd_bwd = 2 * 2 * n * T * (4 * H * (I + H) + H * M)          # forward + backward-data
flops_dw = 2 * n * T * (4 * H * (I + H) + H * M)                    # weight gradients
print(f"engine batch {B}, mini-batch {n} x {T} frames, set A, {dtype}: {dt * 1e3:.3f} ms / optimiser step "
      f"({(flops_fwd_bwd + flops_dw) / dt / 1e12:.2f} TFLOP/s algorithmic); loss {float(losses[0]):.5f} -> {float(losses[-1]):.5f}")
