#!/usr/bin/env python3
"""Diagnostic: the bf16 device path against oracle/bf16_emul.py stage by stage (relative L2 error of every backward buffer).
usage: python tests/emul_stages.py B T H   (kept under tests/: only tests may import oracle/)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from oracle import bf16_emul as be  # noqa: E402
from oracle import manual as mo  # noqa: E402
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

B, T, H = (int(v) for v in sys.argv[1:4])
wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
Bp, Hp, Tp = -(-B // 16) * 16, -(-H // 32) * 32, T // 2
n = lambda t: t.detach().cpu().double().numpy()
tm = lambda a, T_, Fp, F: n(a).reshape(T_, Bp, Fp)[:, :B, :F].transpose(1, 0, 2)
gates = lambda a, T_: n(a).reshape(T_, Bp, 4, Hp)[:, :B, :, :H].transpose(1, 0, 2, 3).reshape(B, T_, 4 * H)
em = be.EmulPlanner(wl.pred_sd, wl.emb_sd)
em.set_targets(wl.target_mel.numpy(), wl.target_semvec.numpy())
em.set_cp(wl.cp0.numpy())
ex = mo.ManualPlanner(wl.pred_sd, wl.emb_sd, objective="acoustic_semvec")
ex.set_targets(wl.target_mel.numpy(), wl.target_semvec.numpy())
ex.set_cp(wl.cp0.numpy())
eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", use_graph=False)
eng.set_targets(wl.target_mel, wl.target_semvec)
eng.set_cp(wl.cp0)
eng.step(1, return_loss=False)
eng.synchronize()
_, _, pe = be.loss_and_grad(em.models, "acoustic_semvec", em.x, em.target_mel, em.target_semvec)
_, _, px = mo.loss_and_grad(ex.models, "acoustic_semvec", ex.x, ex.target_mel, ex.target_semvec)
rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
print("sem      ", rel(n(eng.debug_read("sem")).reshape(Bp, -1)[:B, :300], pe["sem"]))
print("emb dA l1", rel(gates(eng.debug_read("emb.G1"), Tp), pe["emb_dA"][1]))
print("emb dA l0", rel(gates(eng.debug_read("emb.G0"), Tp), pe["emb_dA"][0]))
print("dmel_e   ", rel(tm(eng.debug_read("dmel_e"), Tp, 64, 60), pe["dmel_e"]), " exact:", rel(tm(eng.debug_read("dmel_e"), Tp, 64, 60), px["dmel_from_embedder"]))
print("dY       ", rel(tm(eng.debug_read("dY"), T, 64, 60), pe["dY"]))
print("pred dA  ", rel(gates(eng.debug_read("pred.G0"), T), pe["pred_dA"][0]))
print("dX       ", rel(tm(eng.debug_read("dX"), T, 32, 30), pe["dX"]), " exact:", rel(tm(eng.debug_read("dX"), T, 32, 30), px["grad_model"]))
print("|dX| device / emul / exact", np.linalg.norm(tm(eng.debug_read("dX"), T, 32, 30)), np.linalg.norm(pe["dX"]), np.linalg.norm(px["grad_model"]))
