#!/usr/bin/env python3
"""Which buffer / which rows of the fused backward launch differ from the per-layer path (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from paule_amd import synthetic
from paule_amd.engine import HipPlanner

H = int(os.environ.get("DIAG_H", 720))
for B, T in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]:
    wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
    out = {}
    for mode in ("1", "3"):
        os.environ["PAULE_HIP_FUSED"] = mode
        os.environ["PAULE_HIP_FUSED_MIN_B"] = "1"
        os.environ["PAULE_HIP_SWEEP16"] = "0"
        e = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", use_graph=False)
        e.set_targets(wl.target_mel, wl.target_semvec)
        e.set_cp(wl.cp0)
        e.step(1, return_loss=False)
        e.synchronize()
        out[mode] = {n: e.debug_read(n).cpu().numpy() for n in ("emb.G1", "emb.dh_ext", "emb.G0", "pred.G0", "dX")}
        print(f"B={B} T={T} mode {mode}: plan {e.plan_info()}")
        e.close()
    Bp = (B + 15) // 16 * 16
    for n in ("emb.G1", "emb.dh_ext", "emb.G0", "pred.G0", "dX"):
        a, b = out["1"][n], out["3"][n]
        Tl = a.size // Bp // (a.size // Bp // (T if n in ("pred.G0", "dX") else T // 2)) if False else None
        rows_T = T if n in ("pred.G0", "dX") else T // 2
        a2, b2 = a.reshape(rows_T, Bp, -1), b.reshape(rows_T, Bp, -1)
        d = np.abs(a2 - b2)
        rel = d.max() / (np.abs(a2).max() + 1e-30)
        bad_rows = np.where(d.max(axis=(0, 2)) > 0.05 * np.abs(a2).max())[0]
        bad_t = np.where(d.max(axis=(1, 2)) > 0.05 * np.abs(a2).max())[0]
        bad_c = np.where(d.max(axis=(0, 1)) > 0.05 * np.abs(a2).max())[0]
        if len(bad_rows) and n == "emb.G0":
            idx = np.argwhere(d > 0.05 * np.abs(a2).max())[:12]
            for (tt, rr, cc) in idx:
                print(f"      t={tt} row={rr} col={cc}: per-layer {a2[tt, rr, cc]: .6e}  fused {b2[tt, rr, cc]: .6e}   neighbours fused {b2[tt, rr, cc-1]: .3e} {b2[tt, rr, cc+1]: .3e} per-layer {a2[tt, rr, cc-1]: .3e} {a2[tt, rr, cc+1]: .3e}")
        print(f"   {n:8s} max|diff|/max {rel:9.2e}  nan {np.isnan(b2).sum()}  bad batch rows {bad_rows[:12]}{'...' if len(bad_rows) > 12 else ''} ({len(bad_rows)})  "
              f"bad t {bad_t[:8]}{'...' if len(bad_t) > 8 else ''} ({len(bad_t)})  bad cols {bad_c[:8]} ({len(bad_c)})")
