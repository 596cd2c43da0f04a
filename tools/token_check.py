#!/usr/bin/env python3
"""Token form of the reduce-scatter backward sweep (PAULE_HIP_BWD_STREAM=2) against the flag form (=1): dA of every layer, dL/dCP,
losses and the plan after a few iterations; run-to-run reproducibility of the token form (fresh engines, bit-equal).
usage: token_check.py [B T [set]] ...   (needs a library built with make EXTRA=-DPL_EXPERIMENTS: the token form is not in the shipped
binary -- profiles/r04_token_handoff.txt)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402


def n(x):
    return x.detach().cpu().double().numpy() if hasattr(x, "detach") else np.asarray(x, dtype=np.float64)


def run(wl, B, T, form, iters=4):
    os.environ["PAULE_HIP_BWD_STREAM"] = form
    os.environ["PAULE_HIP_FUSED"] = "1"
    e = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
    e.set_targets(wl.target_mel, wl.target_semvec)
    e.set_cp(wl.cp0)
    l1 = n(e.step(1))
    e.synchronize()
    bufs = {k: n(e.debug_read(k)) for k in ("emb.G1", "emb.G0", "pred.G0", "dX")}
    l2 = n(e.step(iters))
    e.synchronize()
    out = dict(l1=l1, l2=l2, cp=n(e.get_cp()), **bufs)
    e.close()
    return out


shapes = [(256, 300, "A"), (144, 61, "A"), (270, 17, "A"), (64, 15, "A"), (40, 14, "A")]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else "A")]
bad = 0
for B, T, st in shapes:
    wl = synthetic.make_workload(B, T, st)
    a, b, c = run(wl, B, T, "1"), run(wl, B, T, "2"), run(wl, B, T, "2")
    print(f"== B={B} T={T} set {st}")
    for k in ("emb.G1", "emb.G0", "pred.G0", "dX"):
        if k not in a:
            continue
        d = np.linalg.norm(a[k] - b[k]) / max(np.linalg.norm(a[k]), 1e-30)
        cos = float((a[k] * b[k]).sum() / max(np.linalg.norm(a[k]) * np.linalg.norm(b[k]), 1e-30))
        print(f"   {k:8s} flag vs token: rel diff {d:.3e}  cos {cos:.8f}   token run-to-run identical {np.array_equal(b[k], c[k])}")
        bad |= (not np.array_equal(b[k], c[k])) or not (cos > 0.9999) or not np.isfinite(b[k]).all()
    print(f"   loss first iteration max rel diff {np.abs(a['l1'] - b['l1']).max() / np.abs(a['l1']).max():.3e}; after: {np.abs(a['l2'] - b['l2']).max() / np.abs(a['l2']).max():.3e};"
          f" CP max |diff| {np.abs(a['cp'] - b['cp']).max():.3e} mean {np.abs(a['cp'] - b['cp']).mean():.3e}; token CP run-to-run identical {np.array_equal(b['cp'], c['cp'])}")
    bad |= not np.array_equal(b["cp"], c["cp"])
print("FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
