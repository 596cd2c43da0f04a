#!/usr/bin/env python3
"""End-to-end wall time of Paule.plan_resynth at the reference's own operating point: ONE utterance, 5 outer x 24 inner
iterations (paule/paule.py:399-400), Paule's default model shapes, log_ii = n_inner, no synthesis.
usage: plan_e2e.py [batch] [dtype]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from paule_amd import paule as pp, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
wl = synthetic.make_workload(B, 300, "A")
model = pp.Paule(pred_model=wl.pred_sd, embedder=wl.emb_sd, compute_dtype=dtype, device=torch.device("cuda"))
kw = dict(target_acoustic=wl.target_mel.numpy(), target_semvec=wl.target_semvec.numpy(), initial_cp=wl.cp0.numpy(), initialize_from=None,
          objective="acoustic_semvec", n_outer=5, n_inner=24, log_ii=24, continue_learning=False, verbose=False)
for k in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = model.plan_resynth(**kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"run {k}: plan_resynth(B={B}, {dtype}, 5 x 24 iterations) {dt * 1e3:.1f} ms wall = {dt / 120 * 1e3:.2f} ms per inner iteration; "
          f"loss {float(np.mean(res.planned_loss_steps[0])):.3f} -> {float(np.mean(res.planned_loss_steps[-1])):.3f}", flush=True)
