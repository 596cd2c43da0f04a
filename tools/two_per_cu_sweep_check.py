#!/usr/bin/env python3
"""One-off safety sweep (round 4): the two-per-CU forward launch / forward sweep against the one-per-CU kernels over ragged shapes:
losses and CP after 3 iterations must be bit-identical."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from paule_amd import synthetic
from paule_amd.engine import HipPlanner
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = [("A", b, random.randint(14, 70)) for b in (193, 200, 223, 241, 255)] + \
        [("B", b, random.randint(14, 70)) for b in (49, 50, 63, 97, 129, 191, 250)] + \
        [("A", b, random.randint(14, 30)) for b in (353, 400, 777)]
bad = 0
for mset, B, T in cases:
    wl = synthetic.make_workload(B, T, mset)
    res = {}
    for occ in ("0", "-1"):
        os.environ["PAULE_HIP_FUSED_OCC2"] = occ
        os.environ["PAULE_HIP_SWEEP2"] = occ
        e = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", use_graph=(B % 2 == 0))
        e.set_targets(wl.target_mel, wl.target_semvec)
        e.set_cp(wl.cp0)
        losses = e.step(3).cpu().numpy()
        e.synchronize()
        res[occ] = (e.plan_info(), losses, e.get_cp().cpu().numpy())
        del e
    same = np.array_equal(res["0"][1], res["-1"][1]) and np.array_equal(res["0"][2], res["-1"][2])
    bad += 0 if same else 1
    print(f"set {mset} B={B} T={T}: fwd_per_cu {res['0'][0]['fwd_per_cu']} -> {res['-1'][0]['fwd_per_cu']}, fused_fwd {res['-1'][0]['fused_fwd']}, identical {same}", flush=True)
print("MISMATCHES:", bad)
sys.exit(1 if bad else 0)
