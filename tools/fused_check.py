#!/usr/bin/env python3
"""Fused acoustic launches (lstm_fused.hip) against the per-layer path, buffer by buffer, bit for bit.
usage: fused_check.py [B ...]     env: FC_STAGE=fwd|iter (fwd: stop after the forward and compare its stashes)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

stage = os.environ.get("FC_STAGE", "fwd")
frames = int(os.environ.get("FC_FRAMES", 300))
model_set = os.environ.get("FC_SET", "A")
hidden = os.environ.get("FC_HIDDEN")
batches = [int(x) for x in sys.argv[1:]] or [256]
if stage == "fwd":
    os.environ["PAULE_HIP_DEBUG"] = "stop_after_fwd"
os.environ.setdefault("PAULE_HIP_SPIN_MS", "300")
FWD = ["X0", "pred.h0", "pred.c0", "pred.G0", "mel", "mel_tm", "emb.h0", "emb.c0", "emb.G0", "emb.h1", "emb.c1", "emb.G1"]
ALL = FWD + ["sem", "pred.dh_ext", "dX", "grad", "x"]
bad = 0
for B in batches:
    if hidden:
        H = int(hidden)
        wl = synthetic.make_workload(B, frames, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
    else:
        wl = synthetic.make_workload(B, frames, model_set)
    out = []
    for fused in ("0", os.environ.get("FC_MODE", "3")):
        os.environ["PAULE_HIP_FUSED"] = fused
        os.environ["PAULE_HIP_FUSED_MIN_B"] = "1"
        os.environ["PAULE_HIP_SWEEP16"] = "0"   # the per-layer reference on the 32-row kernels whatever the batch
        e = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=frames, objective="acoustic_semvec", dtype="bf16",
                       use_graph=os.environ.get("FC_GRAPH", "0") != "0")
        e.set_targets(wl.target_mel, wl.target_semvec)
        e.set_cp(wl.cp0)
        try:
            e.step(int(os.environ.get("FC_ITERS", 1)), return_loss=False)
            e.synchronize()
        except Exception as ex:  # noqa: BLE001
            print(f"B={B} fused={fused}: FAILED {ex}")
            bad += 1
            out.append(None)
            continue
        names = FWD if stage == "fwd" else ALL
        out.append({n: e.debug_read(n).float().cpu().numpy() for n in names})
        e.close()
    if out[0] is None or out[1] is None:
        continue
    for n in out[0]:
        a, b = out[0][n], out[1][n]
        same = np.array_equal(a, b, equal_nan=True)
        msg = f"B={B} {n:8s} identical={same}"
        if not same:
            bad += 1
            d = np.abs(a.astype(np.float64) - b.astype(np.float64))
            idx = np.flatnonzero(d.ravel() > 0)
            if os.environ.get("FC_BRIEF"):
                print(msg, flush=True)
                continue
            msg += f"  max|diff| {np.nanmax(d):.3e}  n_diff {idx.size}/{d.size}  first flat index {idx[0] if idx.size else -1}  nan_a {np.isnan(a).sum()} nan_b {np.isnan(b).sum()}"
            af, bf = a.astype(np.float64).ravel(), b.astype(np.float64).ravel()
            msg += f"  cos {af @ bf / (np.linalg.norm(af) * np.linalg.norm(bf) + 1e-300):.8f}  rel.l2 {np.linalg.norm(af - bf) / (np.linalg.norm(af) + 1e-300):.3e}"
            Bp = (B + 15) // 16 * 16
            if n.startswith(("pred.", "emb.")):
                Tl = frames if n.startswith("pred.") else frames // 2
                width = a.size // (Tl * Bp)
                tt, bb, cc = np.unravel_index(idx, (Tl, Bp, width))
                msg += "\n    samples (t, b, col: unfused, fused): " + "; ".join(
                    f"({tt[k]},{bb[k]},{cc[k]}: {a.ravel()[idx[k]]:.6g}, {b.ravel()[idx[k]]:.6g})" for k in range(0, min(idx.size, 4000), 800))
                msg += f"\n    diffs at t=0..3: {[int((tt == q).sum()) for q in range(4)]}  by quarter of the columns: {[int((cc * 4 // width == q).sum()) for q in range(4)]}"
                msg += f"\n    largest |a| among differing elements: {np.abs(a.ravel()[idx]).max():.4g}"
        print(msg, flush=True)
print("FUSED_CHECK", "OK" if bad == 0 else f"FAILED ({bad})")
sys.exit(0 if bad == 0 else 1)
