#!/usr/bin/env python3
"""Runs a few planning iterations of a small batch (for rocprofv3 --kernel-trace): wf_probe.py [batch] [dtype] [graph 0/1] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
graph = (sys.argv[3] if len(sys.argv) > 3 else "1") != "0"
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
wl = synthetic.make_workload(B, 300, os.environ.get("AB_SET", "A"))
e = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=300, objective="acoustic_semvec", dtype=dtype, use_graph=graph)
e.set_targets(wl.target_mel, wl.target_semvec)
e.set_cp(wl.cp0)
e.step(iters, return_loss=False)
e.synchronize()
print("done")
