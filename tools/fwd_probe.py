#!/usr/bin/env python3
"""Time of the fused forward launch alone (pl_bench_kernel: hipEvents around `reps` launches) for engine variants.
usage: fwd_probe.py "B:VAR=a,VAR2=b" ...     e.g.  fwd_probe.py 256: 256:PAULE_HIP_FUSED_CP=4,PAULE_HIP_FUSED_CE=4 128:
(written for the forward-pipeline experiment of round 3, tools/experiments/lstm_fused8.hip, whose PAULE_HIP_FUSED_W8 switch no longer exists)
env: FP_FRAMES (300), FP_REPS (20)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

T = int(os.environ.get("FP_FRAMES", 300))
reps = int(os.environ.get("FP_REPS", 20))
keys = set()
for spec in sys.argv[1:]:
    b, _, env = spec.partition(":")
    B = int(b)
    assign = dict(a.split("=") for a in env.split(",") if a)
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(assign)
    os.environ.setdefault("PAULE_HIP_FUSED_MIN_B", "1")
    keys |= set(assign)
    wl = synthetic.make_workload(B, T, "A")
    e = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
    e.set_targets(wl.target_mel, wl.target_semvec)
    e.set_cp(wl.cp0)
    e.step(2, return_loss=False)
    e.synchronize()
    ms, fl = e.bench_kernel("fused_fwd", reps=reps)
    pi = e.plan_info()
    print(f"B={B:4d} T={T} {env or '(defaults)':60s} fwd launch {ms:7.3f} ms = {ms * 1e3 / T:6.2f} us per predictor step; "
          f"plan fused_fwd={pi['fused_fwd']} Cp={pi['fwd_chains_pred']} Ce={pi['fwd_chains_emb']} workgroups={pi['fwd_workgroups']}", flush=True)
    e.close()
