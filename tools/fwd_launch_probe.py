#!/usr/bin/env python3
"""Forward launch alone (pl_bench_kernel) under a list of environment variants.  usage: fwd_launch_probe.py "A=1,B=2/A=0" [B] [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paule_amd import synthetic
from paule_amd.engine import HipPlanner
variants = [dict(a.split("=") for a in v.split(",") if a) for v in sys.argv[1].split("/")]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = int(sys.argv[3]) if len(sys.argv) > 3 else 300
wl = synthetic.make_workload(B, T, os.environ.get("AB_SET", "A"))
keys = sorted({k for v in variants for k in v})
for v in variants:
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(v)
    e = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", use_graph=True)
    e.set_targets(wl.target_mel, wl.target_semvec)
    e.set_cp(wl.cp0)
    e.step(2, return_loss=False)
    e.synchronize()
    p = e.plan_info()
    ms = min(e.bench_kernel("fused_fwd", reps=10)[0] for _ in range(3)) if p["fused_fwd"] else float("nan")
    print(f"{v}: fused_fwd {ms*1e3:.0f} us  chains {p['fwd_chains_pred']}/{p['fwd_chains_emb']}  workgroups {p['fwd_workgroups']}", flush=True)
    del e
