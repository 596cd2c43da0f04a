"""HIP path (through the C-ABI) against the CPU oracle and the reference-generated fixtures.  GPU only.

Tolerances (written here): f32 engine vs float64 oracle over 20 iterations:
  CP atol 1e-5, loss curve rtol 1e-5, forward outputs atol 2e-5;
bf16 engine: loss curve rtol 2e-2, per-term gradient cosine >= 0.999, CP atol 2 * lr * n_iters * 1 %... stated below.
"""
import os
import numpy as np
import pytest
import torch

from conftest import state_dict_from
from oracle import manual as om
from oracle import planner as op
import oracle_golden as og
from paule_amd import synthetic

pytestmark = pytest.mark.gpu
_ORACLE_CACHE = {}

CP_ATOL_F32, LOSS_RTOL_F32, FWD_ATOL_F32 = 1e-5, 1e-5, 2e-5
LOSS_RTOL_BF16, COS_BF16 = 2e-2, 0.999
CASES = {"acoustic": dict(objective="acoustic"), "acoustic_semvec": dict(objective="acoustic_semvec"),
         "semvec": dict(objective="semvec"), "smiling": dict(objective="acoustic_semvec", smiling=True),
         "past_cp": dict(objective="acoustic"), "classifier": dict(objective="acoustic_semvec")}


@pytest.fixture(scope="module")
def HipPlanner():
    from paule_amd.engine import HipPlanner
    return HipPlanner


def _n(t):
    return t.detach().cpu().double().numpy() if isinstance(t, torch.Tensor) else np.asarray(t, dtype=np.float64)


def _cos(a, b):
    a, b = _n(a).ravel(), _n(b).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def _engine(HipPlanner, g, name, dtype="f32", **extra):
    kw = dict(CASES[name])
    eng = HipPlanner(state_dict_from(g, "pred"), state_dict_from(g, "emb"), batch=int(g["B"]), n_frames=int(g["T"]),
                     dtype=dtype, **kw, **extra)
    eng.set_targets(g["target_mel"], g["target_semvec"])
    eng.set_cp(g["cp0"])
    if name == "past_cp":
        eng.set_past_cp(g["past_cp"])
    if name == "classifier":
        eng.set_speech_classifier(state_dict_from(g, "clf"))
    return eng


def test_forward_matches_reference_fixture(HipPlanner, golden_small):
    g = golden_small
    eng = _engine(HipPlanner, g, "acoustic_semvec")
    mel, sem = eng.get_pred()
    np.testing.assert_allclose(_n(mel), g["fwd/pred_mel"], atol=FWD_ATOL_F32, rtol=0)
    np.testing.assert_allclose(_n(sem), g["fwd/pred_semvec"], atol=FWD_ATOL_F32, rtol=0)
    sem_l = eng.embed_mel(g["target_mel"], lens=g["fwd/embed_lens"])
    np.testing.assert_allclose(_n(sem_l), g["fwd/embed_semvec_lens"], atol=FWD_ATOL_F32, rtol=0)


def test_forward_stash_matches_manual(HipPlanner, golden_small):
    """gate stash / h / c of every layer against the explicit numpy forward (localises kernel bugs)."""
    g = golden_small
    eng = _engine(HipPlanner, g, "acoustic_semvec")
    eng.get_pred()
    B, T = int(g["B"]), int(g["T"])
    mm = om.ManualModels(state_dict_from(g, "pred"), state_dict_from(g, "emb"))
    mel, st_p = mm.pred_forward(g["cp0"])
    _, st_e = mm.emb_forward(mel)
    Bp = 16
    for tag, stashes, Tl in (("pred", st_p, T), ("emb", st_e, T // 2)):
        for l, st in enumerate(stashes):
            H = st["c"].shape[2]
            Hp = (H + 31) // 32 * 32
            c = _n(eng.debug_read(f"{tag}.c{l}")).reshape(Tl, Bp, Hp)[:, :B, :H].transpose(1, 0, 2)
            np.testing.assert_allclose(c, st["c"], atol=2e-5, rtol=0, err_msg=f"{tag}.c{l}")
            G = _n(eng.debug_read(f"{tag}.G{l}")).reshape(Tl, Bp, 4, Hp)[:, :B, :, :H].transpose(1, 0, 2, 3)
            for gi, k in enumerate("ifgo"):
                np.testing.assert_allclose(G[:, :, gi], st[k], atol=2e-5, rtol=0, err_msg=f"{tag}.G{l}.{k}")


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("name", sorted(CASES))
def test_trajectory_f32_vs_reference_fixture(HipPlanner, golden_small, name, use_graph):
    g = golden_small
    eng = _engine(HipPlanner, g, name, use_graph=use_graph)
    logs, done = [], 0
    for k in (1, 5, 20):
        loss, grad = eng.step(k - done, return_grad=True)
        logs.append(_n(loss))
        done = k
        np.testing.assert_allclose(_n(eng.get_cp()), g[f"{name}/cp_after_{k}"], atol=CP_ATOL_F32, rtol=0,
                                   err_msg=f"cp after {k}")
        ref_g = g[f"{name}/grad_at_{k}"]
        np.testing.assert_allclose(_n(grad), ref_g, atol=1e-5 * max(1.0, np.abs(ref_g).max()), rtol=0,
                                   err_msg=f"grad at {k}")
    np.testing.assert_allclose(np.concatenate(logs), g[f"{name}/loss_log"], rtol=LOSS_RTOL_F32, atol=1e-7)
    mel, sem = eng.get_pred()
    np.testing.assert_allclose(_n(mel), g[f"{name}/final_pred_mel"], atol=FWD_ATOL_F32, rtol=0)
    np.testing.assert_allclose(_n(sem), g[f"{name}/final_pred_semvec"], atol=FWD_ATOL_F32, rtol=0)


def test_gradient_terms_in_isolation_f32(HipPlanner, golden_small):
    """Each loss term's gradient alone (the 1e5-weighted local-linear term otherwise hides the model gradients)."""
    g = golden_small
    for term, key, obj in (("w_mel", "terms/mel_grad", "acoustic"), ("w_sem", "terms/sem_grad", "semvec"),
                           ("w_vel", "terms/vel_grad", "acoustic"), ("w_jerk", "terms/jerk_grad", "acoustic"),
                           ("w_ll", "terms/ll_grad", "acoustic")):
        w = dict(w_mel=0.0, w_sem=0.0, w_vel=0.0, w_jerk=0.0, w_ll=0.0)
        w[term] = 1.0
        eng = HipPlanner(state_dict_from(g, "pred"), state_dict_from(g, "emb"), batch=int(g["B"]), n_frames=int(g["T"]),
                         objective=obj, weights=w)
        eng.set_targets(g["target_mel"], g["target_semvec"])
        eng.set_cp(g["cp0"])
        _, grad = eng.step(1, return_grad=True)
        ref = g[key]
        got = _n(grad)[0]
        assert _cos(got, ref) > 1 - 1e-9, term
        np.testing.assert_allclose(got, ref, atol=2e-5 * np.abs(ref).max(), rtol=0, err_msg=term)


@pytest.mark.parametrize("name", ["acoustic", "acoustic_semvec"])
def test_trajectory_f32_set_a(HipPlanner, golden_set_a, name):
    """Paule's default architecture (ForwardModel L1/H720, EmbeddingModel L2/H720) against the reference fixture."""
    g = golden_set_a
    pred_sd, emb_sd = synthetic.make_models_frozen("A")   # numpy's frozen RandomState stream: the fixture's weights on any torch version
    chk = np.array([float(sum(v.double().abs().sum() for v in pred_sd.values())), float(sum(v.double().abs().sum() for v in emb_sd.values()))])
    np.testing.assert_allclose(chk, g["weights_checksum"], rtol=1e-13)
    wl = synthetic.Workload(pred_sd, emb_sd, None, None, None, int(g["B"]), int(g["T"]))
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=int(g["B"]), n_frames=int(g["T"]), objective=name)
    eng.set_targets(g["target_mel"], g["target_semvec"])
    eng.set_cp(g["cp0"])
    mel, sem = eng.get_pred()
    np.testing.assert_allclose(_n(mel), g["fwd/pred_mel"], atol=FWD_ATOL_F32, rtol=0)
    np.testing.assert_allclose(_n(sem), g["fwd/pred_semvec"], atol=FWD_ATOL_F32, rtol=0)
    loss = eng.step(20)
    np.testing.assert_allclose(_n(loss), g[f"{name}/loss_log"], rtol=LOSS_RTOL_F32, atol=1e-7)
    np.testing.assert_allclose(_n(eng.get_cp()), g[f"{name}/cp_after_20"], atol=CP_ATOL_F32, rtol=0)


@pytest.mark.parametrize("shape", [dict(B=1, T=46, set="B"), dict(B=19, T=33, set="B"), dict(B=5, T=64, set="A")])
def test_ragged_shapes_vs_oracle_f32(HipPlanner, shape):
    """B = 1 (the reference's only case), batches that are not a multiple of the tile, odd T (last frame dropped),
    the stacked class-default models (set B: L = 4, H = 180 -> padded hidden size)."""
    wl = synthetic.make_workload(shape["B"], shape["T"], shape["set"])
    orc = op.OraclePlanner(op.forward_model_from_state_dict(wl.pred_sd), op.embedding_model_from_state_dict(wl.emb_sd),
                           objective="acoustic_semvec")
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=shape["B"], n_frames=shape["T"], objective="acoustic_semvec")
    for pl in (orc, eng):
        pl.set_targets(wl.target_mel, wl.target_semvec)
        pl.set_cp(wl.cp0)
    lo, lh = _n(orc.step(8)), _n(eng.step(8))
    np.testing.assert_allclose(lh, lo, rtol=LOSS_RTOL_F32, atol=1e-7)
    np.testing.assert_allclose(_n(eng.get_cp()), _n(orc.get_cp()), atol=CP_ATOL_F32, rtol=0)


@pytest.mark.parametrize("shape", [dict(B=100, T=21, set="A"), dict(B=37, T=26, set="B")])
def test_f32_sweep_many_groups_vs_oracle_and_per_step(shape, monkeypatch):
    """f32 persistent sweeps (groups of 16 rows, Hp / 16 workgroups per group: 46 at H = 720, so 5 groups are resident
    on 256 CUs and B = 100 -> 7 groups makes the same workgroups sweep a second group), against the oracle at the f32
    bar, and against the launch-per-step kernels (PAULE_HIP_F32_SWEEP=0; only the dh summation order differs)."""
    from paule_amd.engine import HipPlanner
    wl = synthetic.make_workload(shape["B"], shape["T"], shape["set"])
    orc = op.OraclePlanner(op.forward_model_from_state_dict(wl.pred_sd), op.embedding_model_from_state_dict(wl.emb_sd),
                           objective="acoustic_semvec")
    orc.set_targets(wl.target_mel, wl.target_semvec)
    orc.set_cp(wl.cp0)
    lo, co = _n(orc.step(6)), _n(orc.get_cp())
    outs = []
    for sweep in ("1", "0"):
        monkeypatch.setenv("PAULE_HIP_F32_SWEEP", sweep)
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=shape["B"], n_frames=shape["T"], objective="acoustic_semvec")
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        lh = _n(eng.step(6))
        eng.synchronize()
        np.testing.assert_allclose(lh, lo, rtol=LOSS_RTOL_F32, atol=1e-7, err_msg=f"sweep={sweep}")
        np.testing.assert_allclose(_n(eng.get_cp()), co, atol=CP_ATOL_F32, rtol=0, err_msg=f"sweep={sweep}")
        outs.append(lh)
    np.testing.assert_allclose(outs[0], outs[1], rtol=LOSS_RTOL_F32, atol=1e-7)


@pytest.mark.parametrize("shape", [dict(B=100, T=21, H=720, chains=("0", "-1", "3")), dict(B=150, T=30, H=96, chains=("0", "2", "5")),
                                   dict(B=256, T=16, H=720, chains=("-1",))])
def test_f32_chains_equal_groups_taking_turns(shape, monkeypatch):
    """f32 batches of more 16-row groups than the chip holds at once (lstm_chain_f32.hip): a workgroup serves several groups in
    turn with one copy of its weights, per time step, instead of sweeping them one after the other.  Same MFMA order per output
    element and the same fixed-order sum of the partial tiles as lstm_persist_f32.hip, so the plans are bit-identical to
    PAULE_HIP_F32_CHAINS=0 (at B = 256, where that means the launch-per-step kernels, to the f32 bars); forced chain counts
    cover ragged sets (7 groups in sets of 3, 10 groups in sets of 5 and 2)."""
    from paule_amd.engine import HipPlanner
    B, T, H = shape["B"], shape["T"], shape["H"]
    wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
    outs = {}
    for ch in ("0",) + tuple(c for c in shape["chains"] if c != "0"):
        monkeypatch.setenv("PAULE_HIP_F32_CHAINS", ch)
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec")
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        loss = _n(eng.step(4))
        eng.synchronize()
        outs[ch] = (loss, _n(eng.get_cp()))
    for ch, (loss, cp) in outs.items():
        if ch == "0":
            continue
        if B <= 240:   # the reference run is the sweeps with the groups taking turns: same bits
            np.testing.assert_array_equal(loss, outs["0"][0], err_msg=f"chains={ch}")
            np.testing.assert_array_equal(cp, outs["0"][1], err_msg=f"chains={ch}")
        else:          # the reference run is the launch-per-step kernels: dh summation order differs
            np.testing.assert_allclose(loss, outs["0"][0], rtol=LOSS_RTOL_F32, atol=1e-7)
            np.testing.assert_allclose(cp, outs["0"][1], atol=CP_ATOL_F32, rtol=0)


@pytest.mark.parametrize("shape", [dict(B=64, T=31, H=720), dict(B=20, T=40, H=720), dict(B=96, T=21, H=720, chains="0"), dict(B=150, T=30, H=96),
                                   dict(B=7, T=24, H=96)])
def test_f32_streamed_backward_is_bit_identical(shape, monkeypatch):
    """Round 5: the whole-sequence f32 backward sweeps hand every 16 x 16 partial tile over on its own (lstm_bwd_stream_f32_kernel: the
    producing wave's counted wait and the tile's own flag, rotated destination order, each ingest wave polling the flags of the sources it
    sums) instead of one flag per workgroup and step.  Same tiles, same MFMAs, same fixed order of every sum: against
    PAULE_HIP_F32_STREAM=0 the loss log, every layer's dA, dL/dCP and CP are the same bits.  cfg2's shape (4 groups x 46 workgroups), a
    ragged group, groups taking turns (two passes with the chains off), a narrow model (6 workgroups a group: the same-XCD hand-off,
    odd and even T), a batch smaller than a group."""
    from paule_amd.engine import HipPlanner
    B, T, H = shape["B"], shape["T"], shape["H"]
    wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
    if "chains" in shape:
        monkeypatch.setenv("PAULE_HIP_F32_CHAINS", shape["chains"])
    monkeypatch.setenv("PAULE_HIP_F32_VALU", "0")
    outs = {}
    for st in ("0", "1"):
        monkeypatch.setenv("PAULE_HIP_F32_STREAM", st)
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec")
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        l1 = _n(eng.step(1))
        eng.synchronize()
        bufs = {k: _n(eng.debug_read(k)) for k in ("emb.G1", "emb.G0", "pred.G0", "dX")}
        l4 = _n(eng.step(3))
        eng.synchronize()
        outs[st] = (l1, bufs, l4, _n(eng.get_cp()))
        eng.close()
    a, b = outs["1"], outs["0"]
    np.testing.assert_array_equal(a[0], b[0])
    for k in b[1]:
        assert np.isfinite(a[1][k]).all() and np.abs(a[1][k]).max() > 0, k
        np.testing.assert_array_equal(a[1][k], b[1][k], err_msg=k)
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(a[3], b[3])


def test_optimizer_state_persists_and_resets(HipPlanner, golden_small):
    """Adam state continues across pl_step calls (= outer iterations, paule/paule.py:797) and resets on request."""
    g = golden_small
    a = _engine(HipPlanner, g, "acoustic")
    a.step(20)
    b = _engine(HipPlanner, g, "acoustic")
    for _ in range(4):
        b.step(5)
    np.testing.assert_array_equal(_n(a.get_cp()), _n(b.get_cp()))
    b.set_cp(g["cp0"])
    b.reset_optimizer()
    b.step(20)
    np.testing.assert_array_equal(_n(a.get_cp()), _n(b.get_cp()))


def test_weight_reupload(HipPlanner, golden_small):
    g = golden_small
    eng = _engine(HipPlanner, g, "acoustic")
    mel0, _ = eng.get_pred(with_semvec=False)
    sd = {k: v * 0.5 for k, v in state_dict_from(g, "pred").items()}
    eng.set_weights(pred_model=sd)
    mel1, _ = eng.get_pred(with_semvec=False)
    ref = op.forward_model_from_state_dict(sd)(torch.from_numpy(g["cp0"]))
    np.testing.assert_allclose(_n(mel1), _n(ref), atol=FWD_ATOL_F32, rtol=0)
    assert np.abs(_n(mel0) - _n(mel1)).max() > 1e-3


@pytest.mark.parametrize("name", ["acoustic", "acoustic_semvec"])
def test_bf16_against_oracle(HipPlanner, golden_small, name):
    g = golden_small
    eng = _engine(HipPlanner, g, name, dtype="bf16")
    loss, grad = eng.step(1, return_grad=True)
    assert _cos(grad, g[f"{name}/grad_at_1"]) >= COS_BF16
    more = eng.step(19)
    full = np.concatenate([_n(loss), _n(more)])
    np.testing.assert_allclose(full, g[f"{name}/loss_log"], rtol=LOSS_RTOL_BF16, atol=1e-4)
    # lr * n_iters bounds how far Adam can move a coordinate; bf16 must stay within 5 % of that budget
    np.testing.assert_allclose(_n(eng.get_cp()), g[f"{name}/cp_after_20"], atol=0.05 * 0.01 * 20, rtol=0)


def test_bf16_sweep_equals_per_step_kernels(golden_small, monkeypatch):
    """The persistent LSTM sweeps (one launch per layer, in-launch exchange) and the launch-per-step kernels run the
    same arithmetic (MFMA shape and k order differ: 32x32x16 vs 16x16x32), so they agree to f32 accumulation noise;
    and no bounded wait times out."""
    from paule_amd.engine import HipPlanner
    g = golden_small
    outs = []
    for no_sweep in ("0", "1"):
        monkeypatch.setenv("PAULE_HIP_NO_SWEEP", no_sweep)
        eng = _engine(HipPlanner, g, "acoustic_semvec", dtype="bf16")
        loss = _n(eng.step(5))
        eng.synchronize()
        outs.append((loss, _n(eng.get_cp())))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=2e-3, atol=1e-5)
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=0, atol=2e-3)


@pytest.mark.parametrize("bwd_mode", ["0", "1"])
@pytest.mark.parametrize("shape", [dict(B=40, T=30, set="B"), dict(B=70, T=24, set="A"), dict(B=270, T=17, set="A")])
def test_bf16_sweep_ragged_groups_vs_oracle(HipPlanner, shape, bwd_mode, monkeypatch):
    """Persistent sweeps with batch groups that are not full (B % 32 != 0), several groups, the stacked class-default
    models (set B: 4 x H180 -> 6 workgroups per group) and Paule's default H = 720 (23 workgroups per group); more groups
    than fit on the chip at once (B = 270 -> 9 groups, 8 resident: the ninth is swept by the same workgroups afterwards),
    odd T (last frame dropped); both backward exchange forms (1: reduce-scatter of partial dh tiles, the default;
    0: all-gather of dA)."""
    monkeypatch.setenv("PAULE_HIP_BWD_MODE", bwd_mode)
    wl = synthetic.make_workload(shape["B"], shape["T"], shape["set"])
    orc = op.OraclePlanner(op.forward_model_from_state_dict(wl.pred_sd), op.embedding_model_from_state_dict(wl.emb_sd),
                           objective="acoustic_semvec")
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=shape["B"], n_frames=shape["T"], objective="acoustic_semvec",
                     dtype="bf16")
    if shape["B"] == 70 and bwd_mode == "1":   # 49 ... 128 rows of set A: the library's default is the fused forward + backward launches
        plan = eng.plan_info()
        assert plan["fused_fwd"] == 1 and plan["fused_bwd"] == 1, plan
    for pl in (orc, eng):
        pl.set_targets(wl.target_mel, wl.target_semvec)
        pl.set_cp(wl.cp0)
    lo, lh = _n(orc.step(4)), _n(eng.step(4))
    eng.synchronize()
    # small weighted terms (velocity ~0.05) carry bf16 noise of the same absolute size as the large ones
    np.testing.assert_allclose(lh, lo, rtol=LOSS_RTOL_BF16, atol=5e-3)
    # Adam is sign-like where the gradient is bf16 noise: single coordinates may flip (bounded by lr per step), the bulk agrees
    dcp = np.abs(_n(eng.get_cp()) - _n(orc.get_cp()))
    assert dcp.max() <= 0.5 * 0.01 * 4 and dcp.mean() <= 1e-4, (dcp.max(), dcp.mean())


def test_bf16_model_gradients_in_isolation(HipPlanner, golden_small):
    g = golden_small
    for term, key, obj in (("w_mel", "terms/mel_grad", "acoustic"), ("w_sem", "terms/sem_grad", "semvec")):
        w = dict(w_mel=0.0, w_sem=0.0, w_vel=0.0, w_jerk=0.0, w_ll=0.0)
        w[term] = 1.0
        eng = HipPlanner(state_dict_from(g, "pred"), state_dict_from(g, "emb"), batch=int(g["B"]), n_frames=int(g["T"]),
                         objective=obj, weights=w, dtype="bf16")
        eng.set_targets(g["target_mel"], g["target_semvec"])
        eng.set_cp(g["cp0"])
        _, grad = eng.step(1, return_grad=True)
        assert _cos(_n(grad)[0], g[key]) >= COS_BF16, term


def test_full_size_properties_bf16(HipPlanner, monkeypatch):
    """BASELINE.json's headline shape (B = 256, T = 300, set A, acoustic_semvec, bf16): size-independent properties.
    (1) batch rows are independent: utterance b of the big batch == the same utterance planned in a batch of 16;
    (2) the loss decreases; (3) CP stays inside the clamp; (4) replaying from the same start is bit-identical."""
    B, T = 256, 300
    wl = synthetic.make_workload(B, T, "A")
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    loss = _n(eng.step(6))
    cp = _n(eng.get_cp())
    assert np.isfinite(loss).all() and np.isfinite(cp).all()
    assert (loss[-1, :, 0] < loss[0, :, 0]).all()
    assert np.abs(cp).max() <= 1.05 + 1e-6
    sl = slice(100, 116)
    for sweep16 in ("0", "1"):   # 0: the same 32-row kernels as the big batch -> bit-equal; 1: the 16-row kernels a batch of 16 gets by default
        monkeypatch.setenv("PAULE_HIP_SWEEP16", sweep16)
        sub = HipPlanner(wl.pred_sd, wl.emb_sd, batch=16, n_frames=T, objective="acoustic_semvec", dtype="bf16")
        sub.set_targets(wl.target_mel[sl], wl.target_semvec[sl])
        sub.set_cp(wl.cp0[sl])
        loss_s = _n(sub.step(6))
        if sweep16 == "0":
            np.testing.assert_array_equal(loss_s, loss[:, sl])
            np.testing.assert_array_equal(_n(sub.get_cp()), cp[sl])
        else:
            np.testing.assert_allclose(loss_s, loss[:, sl], rtol=2e-3, atol=2e-4)
            assert np.abs(_n(sub.get_cp()) - cp[sl]).mean() <= 1e-4
    monkeypatch.delenv("PAULE_HIP_SWEEP16")
    eng.set_cp(wl.cp0)
    eng.reset_optimizer()
    np.testing.assert_array_equal(_n(eng.step(6)), loss)
    eng.synchronize()


@pytest.mark.parametrize("shape", [dict(B=144, T=61, set="A"), dict(B=270, T=17, set="A"), dict(B=256, T=24, set="B")])
def test_backward_sweep_forms_are_bit_identical(HipPlanner, monkeypatch, shape):
    """Round 3's two changes of the 32-row reduce-scatter backward sweep change no bit: tile products on four waves (round 2's kernel),
    on eight waves with one hand-off per step, and on eight waves with per-tile flags, rotated tile order and streamed ingest (the
    default) produce the same dA of every layer, the same dL/dCP, losses and plan.  Ragged last group (B = 144), more groups than
    the chip holds at once (B = 270: a workgroup sweeps a second group), the stacked predictor's set (B: only the 720-wide embedder
    runs this kernel)."""
    B, T = shape["B"], shape["T"]
    wl = synthetic.make_workload(B, T, shape["set"])
    monkeypatch.setenv("PAULE_HIP_FUSED", "1")   # fused forward launch + per-layer backward sweeps, as cfg3 runs
    monkeypatch.setenv("PAULE_HIP_BWD_XT", "0")  # dL/dCP by the batched product in all three forms (round 4's ride-along tile sums it in another order)
    out = {}
    # round 5: the streamed form without its prefetcher workgroups (they only warm the L2: speed, never bits), and the chained form -- a
    # workgroup serves two / three groups in turn with one copy of its weights (PAULE_HIP_BWD_CHAINS; a probe, lstm_bwd_rs_chain_kernel)
    for form, env in (("w4", dict(PAULE_HIP_BWD_WAVES="4")), ("w8", dict(PAULE_HIP_BWD_WAVES="8", PAULE_HIP_BWD_STREAM="0")),
                      ("stream", dict(PAULE_HIP_BWD_WAVES="8", PAULE_HIP_BWD_STREAM="1")),
                      ("stream_nopf", dict(PAULE_HIP_BWD_WAVES="8", PAULE_HIP_BWD_STREAM="1", PAULE_HIP_BWD_PF="0")),
                      ("chain2", dict(PAULE_HIP_BWD_WAVES="8", PAULE_HIP_BWD_STREAM="1", PAULE_HIP_BWD_CHAINS="2")),
                      ("chain3", dict(PAULE_HIP_BWD_WAVES="8", PAULE_HIP_BWD_STREAM="1", PAULE_HIP_BWD_CHAINS="3"))):
        for k in ("PAULE_HIP_BWD_WAVES", "PAULE_HIP_BWD_STREAM", "PAULE_HIP_BWD_PF", "PAULE_HIP_BWD_CHAINS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
        assert eng.plan_info()["bwd_waves"] == (4 if form == "w4" else 8)
        if shape["set"] == "A":   # the predictor's own sweep: prefetchers beside the streamed form only (every shape here leaves CUs idle)
            assert (eng.plan_info()["bwd_prefetchers"] > 0) == (form == "stream"), (form, eng.plan_info())
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        loss = _n(eng.step(3))
        eng.synchronize()
        out[form] = dict(loss=loss, cp=_n(eng.get_cp()), dX=_n(eng.debug_read("dX")), G=_n(eng.debug_read("emb.G0")))
        eng.close()
    for form in ("w8", "stream", "stream_nopf", "chain2", "chain3"):
        for k in ("loss", "cp", "dX", "G"):
            np.testing.assert_array_equal(out[form][k], out["w4"][k], err_msg=f"{form}: {k}")


@pytest.mark.parametrize("shape", [dict(B=256, T=300, fused="1"), dict(B=270, T=120, fused="0")])
def test_big_gemm_matches_tile_gemm_bit_for_bit(HipPlanner, monkeypatch, shape):
    """Round 5: the large bf16 products run on gemm_big.hip's 256 x 256 tiles (eight waves, LDS-DMA double buffer).  Same MFMA, same k order,
    same epilogue arithmetic as gemm.hip's tiles: with PAULE_HIP_GEMM_BIG=0 (every product on gemm.hip) the iteration produces the same
    bits -- stashes, dL/dCP, losses, final CP.  Shapes: cfg3 (the dL/dh product of embedder layer 1: K = 2944, a partial last N tile);
    the per-layer path at 270 rows (ragged last M tile; the input projections' K = 736 ends on a HALF stage)."""
    B, T = shape["B"], shape["T"]
    wl = synthetic.make_workload(B, T, "A")
    monkeypatch.setenv("PAULE_HIP_FUSED", shape["fused"])
    out = {}
    for big in ("0", "1"):
        monkeypatch.setenv("PAULE_HIP_GEMM_BIG", big)
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        loss = _n(eng.step(3))
        eng.synchronize()
        out[big] = dict(loss=loss, cp=_n(eng.get_cp()), dX=_n(eng.debug_read("dX")), G0=_n(eng.debug_read("emb.G0")), G1=_n(eng.debug_read("emb.G1")))
        eng.close()
    for k in out["0"]:
        np.testing.assert_array_equal(out["1"][k], out["0"][k], err_msg=k)


@pytest.mark.parametrize("shape", [dict(B=256, T=300), dict(B=144, T=61), dict(B=270, T=17), dict(B=16, T=40)])
def test_ride_along_input_gradient_equals_the_batched_product(HipPlanner, monkeypatch, shape):
    """Round 4: in the predictor's streamed backward sweep the workgroup's partial dL/dCP = dA_t[its 128 gate rows] W_ih rides along as
    one more tile per step (the last wave's free tile slot) and a reduce kernel sums the 23 partials in a fixed order; the batched
    product dA W_ih -- which re-read the whole dA stash -- is gone.  Against PAULE_HIP_BWD_XT=0 (the product): the recurrence is
    untouched (every layer's dA and the losses bit-identical), dL/dCP agrees to f32 summation order (both accumulate the same bf16
    products in f32), the plans stay together; full, ragged and multi-pass groups, and the 32-row kernel on a 16-row batch."""
    B, T = shape["B"], shape["T"]
    wl = synthetic.make_workload(B, T, "A")
    monkeypatch.setenv("PAULE_HIP_FUSED", "1")
    monkeypatch.setenv("PAULE_HIP_FUSED_MIN_B", "1")
    monkeypatch.setenv("PAULE_HIP_SWEEP16", "0")
    out = {}
    for xt in ("0", "1"):
        monkeypatch.setenv("PAULE_HIP_BWD_XT", xt)
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        l1 = _n(eng.step(1))
        eng.synchronize()
        bufs = {k: _n(eng.debug_read(k)) for k in ("emb.G1", "emb.G0", "pred.G0", "dX")}
        l5 = _n(eng.step(4))
        eng.synchronize()
        out[xt] = (l1, bufs, l5, _n(eng.get_cp()))
        eng.close()
    np.testing.assert_array_equal(out["1"][0], out["0"][0])
    for k in ("emb.G1", "emb.G0", "pred.G0"):
        np.testing.assert_array_equal(out["1"][1][k], out["0"][1][k], err_msg=k)
    a, b = out["1"][1]["dX"], out["0"][1]["dX"]
    assert np.isfinite(a).all() and np.abs(a - b).max() <= 2e-6 * np.abs(b).max(), np.abs(a - b).max() / np.abs(b).max()
    # The default since round 5 (PAULE_HIP_BWD_XT=2): the same sweep WITHOUT its stores of dA over the predictor's gate stash -- nothing in a
    # planning iteration reads them once dL/dCP rides along.  Same bits everywhere else; the debug read of that stash says so instead of handing
    # out forward gates as dA (a 16-row batch has no ride-along: its dA is there).
    monkeypatch.setenv("PAULE_HIP_BWD_XT", "2")
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    l1 = _n(eng.step(1))
    eng.synchronize()
    np.testing.assert_array_equal(l1, out["1"][0])
    for k in ("emb.G1", "emb.G0", "dX"):
        np.testing.assert_array_equal(_n(eng.debug_read(k)), out["1"][1][k], err_msg=k)
    try:
        g0 = _n(eng.debug_read("pred.G0"))
    except ValueError as e:
        assert "did not keep the predictor's dA" in str(e), e
        g0 = None
    assert B != 256 or g0 is None            # cfg3's shape rides along for certain
    if g0 is not None:                       # (a shape whose backward pass another kernel carries keeps dA as before)
        np.testing.assert_array_equal(g0, out["1"][1]["pred.G0"])
    l5 = _n(eng.step(4))
    eng.synchronize()
    np.testing.assert_array_equal(l5, out["1"][2])
    np.testing.assert_array_equal(_n(eng.get_cp()), out["1"][3])
    eng.close()
    monkeypatch.delenv("PAULE_HIP_BWD_XT")
    np.testing.assert_allclose(out["1"][2], out["0"][2], rtol=1e-6, atol=1e-9)
    assert np.abs(out["1"][3] - out["0"][3]).max() <= 2e-5   # Adam: where |g| ~ eps a last-bit difference of g moves the update by a fraction of lr


def test_fused_launch_census_late_sign_in_keeps_the_first_cause(HipPlanner, monkeypatch):
    """the failure the census exists for is a workgroup that gets its CU LATE (a second process held it).  The resident
    workgroups give up after the census bound (status 2) and leave; the late one then signs in, completes the count -- and must
    not walk into its role as if the launch were whole, nor overwrite the status with the generic time-out of a role wait:
    pl_synchronize still reports the residency error (PL_ERR_STATE, "not all resident").  Test hook: workgroup 0 signs in late."""
    B, T, H = 40, 30, 96
    wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
    monkeypatch.setenv("PAULE_HIP_FUSED", "1")
    monkeypatch.setenv("PAULE_HIP_FUSED_MIN_B", "1")
    monkeypatch.setenv("PAULE_HIP_DEBUG", "census_ms=10,census_late_ms=40")
    monkeypatch.setenv("PAULE_HIP_SPIN_MS", "300")
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", use_graph=False)
    assert eng.plan_info()["fused_fwd"] == 1
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    eng.step(1, return_loss=False)
    with pytest.raises(ValueError, match="not all resident"):
        eng.synchronize()
    monkeypatch.setenv("PAULE_HIP_DEBUG", "census_ms=10")
    eng.set_cp(wl.cp0)
    eng.reset_optimizer()
    assert np.isfinite(_n(eng.step(2))).all()
    eng.synchronize()


# ---- fused acoustic launches (lstm_fused.hip): one persistent launch per direction, roles by workgroup ---------------------
FWD_BUFFERS = ["pred.h0", "pred.c0", "pred.G0", "mel", "mel_tm", "emb.h0", "emb.c0", "emb.G0", "emb.h1", "emb.c1", "emb.G1"]


def _fused_pair(HipPlanner, monkeypatch, wl, B, T, modes, iters, use_graph, extra_env=None):
    """The same plan on the per-layer path (PAULE_HIP_FUSED=0) and with the fused launches; returns {mode: engine}."""
    out = {}
    for mode in modes:
        monkeypatch.setenv("PAULE_HIP_FUSED", mode)
        monkeypatch.setenv("PAULE_HIP_FUSED_MIN_B", "1")
        monkeypatch.setenv("PAULE_HIP_SWEEP16", "0")   # small batches: the per-layer reference on the 32-row kernels too (the 16-row ones sum in another order)
        for k, v in (extra_env or {}).items():
            monkeypatch.setenv(k, v)
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", use_graph=use_graph)
        # the planner declines the fused launches silently (CU budget, widths, crossover ...): a comparison "fused vs per layer" of two
        # per-layer engines would pass vacuously -- the plan the library made is part of what is tested
        plan = eng.plan_info()
        assert plan["fused_fwd"] == (1 if int(mode) & 1 else 0) and plan["fused_bwd"] == (1 if int(mode) & 2 else 0), (mode, plan)
        if int(mode) & 1 and extra_env and "PAULE_HIP_FUSED_CP" in extra_env:
            assert plan["fwd_chains_pred"] == int(extra_env["PAULE_HIP_FUSED_CP"]) and plan["fwd_chains_emb"] == int(extra_env["PAULE_HIP_FUSED_CE"]), plan
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        eng.losses = _n(eng.step(iters))
        eng.synchronize()
        out[mode] = eng
    return out


@pytest.mark.parametrize("shape", [dict(B=256, T=300, H=720, graph=True), dict(B=144, T=61, H=720, graph=False),
                                   dict(B=40, T=50, H=96, graph=False), dict(B=33, T=31, H=96, graph=True, chains=dict(PAULE_HIP_FUSED_CP="1", PAULE_HIP_FUSED_CE="2"))])
def test_fused_forward_is_bit_identical(HipPlanner, monkeypatch, shape):
    """The fused forward launch (predictor recurrence, mel head + pooling, embedder layer 1, layer-2 projection, layer 2 as
    roles of one grid, several batch groups per workgroup) computes what the per-layer sweeps and GEMMs compute, bit for bit:
    every forward stash, the pooled mel, and with them the losses and the CP trajectory.  Ragged batches (a last group of 16
    rows, fewer groups than chains), odd T, a one-group-per-workgroup table."""
    B, T, H = shape["B"], shape["T"], shape["H"]
    wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
    monkeypatch.setenv("PAULE_HIP_FUSED_OCC2", "0")   # this test is lstm_fused.hip's (one workgroup per CU); the two-per-CU launch has its own below
    monkeypatch.setenv("PAULE_HIP_DEBUG", "stop_after_fwd")
    e = _fused_pair(HipPlanner, monkeypatch, wl, B, T, ("0", "1"), 1, False, shape.get("chains"))
    assert e["1"].plan_info()["fwd_per_cu"] == 1
    for name in FWD_BUFFERS:
        np.testing.assert_array_equal(_n(e["1"].debug_read(name)), _n(e["0"].debug_read(name)), err_msg=name)
    monkeypatch.delenv("PAULE_HIP_DEBUG")
    e = _fused_pair(HipPlanner, monkeypatch, wl, B, T, ("0", "1"), 4, shape["graph"], shape.get("chains"))
    np.testing.assert_array_equal(e["1"].losses, e["0"].losses)
    np.testing.assert_array_equal(_n(e["1"].get_cp()), _n(e["0"].get_cp()))
    assert (e["1"].losses[-1, :, 0] < e["1"].losses[0, :, 0]).all()


@pytest.mark.parametrize("shape", [dict(B=144, T=61, H=720, gpp="2"), dict(B=200, T=33, H=96, gpp="4"), dict(B=300, T=20, H=96, gpp="2", set_b=True)])
def test_fused_forward_in_passes_is_bit_identical(HipPlanner, monkeypatch, shape):
    """Round 4: batches of more 32-row groups than the forward launch's roles hold at once (cfg4's 2048 rows on one GPU: 64 groups) run
    the launch in PASSES -- the role table of one chip-load, every role walking its sets pass after pass (lstm_fused.hip:
    fused_fwd_kernel, FusedArgs::gpp).  Forced here at small batches (PAULE_HIP_FUSED_GPP: 5 groups in passes of 2, 7 in passes of 4,
    ragged last groups, a last pass with fewer sets than roles have; the stacked two-width predictor): every forward stash, the pooled
    mel, losses and the plan equal the per-layer path's bit for bit."""
    B, T, H = shape["B"], shape["T"], shape["H"]
    if "FUSED_PASSES=1" not in os.environ.get("PAULE_HIP_BUILD_OPTIONS", ""):
        pytest.skip("the pass loop is a build option since round 5 (make EXTRA=-DFUSED_PASSES=1; set PAULE_HIP_BUILD_OPTIONS=FUSED_PASSES=1 to run this test)")
    if shape.get("set_b"):
        wl = synthetic.make_workload(B, T, "B")
        bufs = [f"pred.{k}{l}" for l in range(4) for k in "hcG"] + ["mel", "mel_tm", "emb.h0", "emb.c0", "emb.G0"]
    else:
        wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
        bufs = FWD_BUFFERS
    monkeypatch.setenv("PAULE_HIP_DEBUG", "stop_after_fwd")
    e = _fused_pair(HipPlanner, monkeypatch, wl, B, T, ("0", "1"), 1, False, dict(PAULE_HIP_FUSED_GPP=shape["gpp"]))
    for name in bufs:
        np.testing.assert_array_equal(_n(e["1"].debug_read(name)), _n(e["0"].debug_read(name)), err_msg=name)
    monkeypatch.delenv("PAULE_HIP_DEBUG")
    e = _fused_pair(HipPlanner, monkeypatch, wl, B, T, ("0", "1"), 3, True, dict(PAULE_HIP_FUSED_GPP=shape["gpp"]))
    np.testing.assert_array_equal(e["1"].losses, e["0"].losses)
    np.testing.assert_array_equal(_n(e["1"].get_cp()), _n(e["0"].get_cp()))
    monkeypatch.delenv("PAULE_HIP_FUSED_GPP")


@pytest.mark.parametrize("shape", [dict(B=256, T=300, H=720, graph=True), dict(B=144, T=61, H=720, graph=False),
                                   dict(B=80, T=33, H=720, graph=True, chains=dict(PAULE_HIP_FUSED_CP="2", PAULE_HIP_FUSED_CE="2")),
                                   dict(B=40, T=50, H=96, graph=False), dict(B=256, T=60, set_b=True, graph=True), dict(B=70, T=31, set_b=True, graph=False)])
def test_fused_forward_two_per_cu_is_bit_identical(HipPlanner, monkeypatch, shape):
    """Round 4: the forward launch written for TWO workgroups per CU (lstm_fused2.hip: 256 registers, 80 KB of LDS, the h tile straight
    into LDS by LDS-DMA in a layout permuted for conflict-free operand reads, weight fragments partly re-fetched per chain-step, role
    table planned for 2 x n_cu slots) computes what the per-layer sweeps and GEMMs compute, bit for bit: every forward stash, the pooled
    mel, losses and the plan.  cfg3's shape under a graph, a ragged batch (last group of 16 rows), two chains per workgroup with an odd
    number of groups, a narrow model, the stacked two-width predictor of model set B (full and ragged)."""
    B, T = shape["B"], shape["T"]
    if shape.get("set_b"):
        wl = synthetic.make_workload(B, T, "B")
        bufs = [f"pred.{k}{l}" for l in range(4) for k in "hcG"] + ["mel", "mel_tm", "emb.h0", "emb.c0", "emb.G0"]
    else:
        H = shape["H"]
        wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
        bufs = FWD_BUFFERS
    env = dict(PAULE_HIP_FUSED_OCC2="1", **(shape.get("chains") or {}))
    monkeypatch.setenv("PAULE_HIP_DEBUG", "stop_after_fwd")
    e = _fused_pair(HipPlanner, monkeypatch, wl, B, T, ("0", "1"), 1, False, env)
    assert e["1"].plan_info()["fwd_per_cu"] == 2, e["1"].plan_info()
    for name in bufs:
        np.testing.assert_array_equal(_n(e["1"].debug_read(name)), _n(e["0"].debug_read(name)), err_msg=name)
    monkeypatch.delenv("PAULE_HIP_DEBUG")
    e = _fused_pair(HipPlanner, monkeypatch, wl, B, T, ("0", "1"), 3, shape["graph"], env)
    np.testing.assert_array_equal(e["1"].losses, e["0"].losses)
    np.testing.assert_array_equal(_n(e["1"].get_cp()), _n(e["0"].get_cp()))
    monkeypatch.delenv("PAULE_HIP_FUSED_OCC2")


@pytest.mark.parametrize("shape", [dict(B=800, T=14, H=720), dict(B=403, T=15, H=720, force="1"), dict(B=130, T=20, H=96, force="1")])
def test_two_per_cu_forward_sweep_is_bit_identical(HipPlanner, monkeypatch, shape):
    """Round 4: per-layer forward sweeps of batches with more 32-row groups than one pass of the one-per-CU sweep holds run the two-per-CU
    recurrence role (lstm_fused2.hip: lstm_fwd2_sweep_kernel; 25 groups at H = 720 = two passes of 13 / 12 sets where the old sweep takes
    three of 11): every forward stash of every layer (fused input projection and precomputed one), the pooled mel, losses and the plan equal
    the one-per-CU sweeps' bit for bit; ragged last groups; forced at smaller batches and a narrow model."""
    B, T, H = shape["B"], shape["T"], shape["H"]
    wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
    monkeypatch.setenv("PAULE_HIP_FUSED", "0")
    monkeypatch.setenv("PAULE_HIP_SWEEP16", "0")
    out = {}
    for v in ("0", shape.get("force", "-1")):
        monkeypatch.setenv("PAULE_HIP_SWEEP2", v)
        monkeypatch.setenv("PAULE_HIP_DEBUG", "stop_after_fwd")
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", use_graph=False)
        assert eng.plan_info()["fused_fwd"] == 0
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        eng.step(1)
        eng.synchronize()
        bufs = {name: _n(eng.debug_read(name)) for name in FWD_BUFFERS}
        monkeypatch.delenv("PAULE_HIP_DEBUG")
        eng2 = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", use_graph=True)
        eng2.set_targets(wl.target_mel, wl.target_semvec)
        eng2.set_cp(wl.cp0)
        losses = _n(eng2.step(3))
        eng2.synchronize()
        out[v] = (bufs, losses, _n(eng2.get_cp()))
    a, b = out["0"], out[shape.get("force", "-1")]
    for name in FWD_BUFFERS:
        np.testing.assert_array_equal(b[0][name], a[0][name], err_msg=name)
    np.testing.assert_array_equal(b[1], a[1])
    np.testing.assert_array_equal(b[2], a[2])


def test_two_per_cu_forward_beside_the_fused_backward_launch(HipPlanner, monkeypatch):
    """Round 4: the two-per-CU forward launch has chain counts of its own, so it can sit in front of the (one-per-CU) fused BACKWARD launch:
    model set B at 256 rows -- the planner takes it by itself (predictor roles 2 -> 1 chains), the backward launch keeps its plan, and six
    iterations equal the one-per-CU schedule bit for bit (losses, dL/dCP, CP)."""
    B, T = 256, 40
    wl = synthetic.make_workload(B, T, "B")
    out = {}
    for occ2 in ("0", None):
        if occ2 is None:
            monkeypatch.delenv("PAULE_HIP_FUSED_OCC2", raising=False)
        else:
            monkeypatch.setenv("PAULE_HIP_FUSED_OCC2", occ2)
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", use_graph=True)
        plan = eng.plan_info()
        assert plan["fused_fwd"] == 1 and plan["fused_bwd"] == 1, plan
        assert plan["fwd_per_cu"] == (1 if occ2 == "0" else 2), plan
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        losses = _n(eng.step(6))
        eng.synchronize()
        out[occ2] = (plan, losses, _n(eng.get_cp()), _n(eng.get_grad()) if hasattr(eng, "get_grad") else None)
    assert out[None][0]["bwd_chains_pred"] == out["0"][0]["bwd_chains_pred"] and out[None][0]["bwd_chains_emb"] == out["0"][0]["bwd_chains_emb"]
    assert out[None][0]["fwd_chains_pred"] == 1 < out["0"][0]["fwd_chains_pred"], (out[None][0], out["0"][0])
    np.testing.assert_array_equal(out[None][1], out["0"][1])
    np.testing.assert_array_equal(out[None][2], out["0"][2])


@pytest.mark.parametrize("shape", [dict(B=256, T=60, graph=True), dict(B=70, T=31, graph=False)])
def test_fused_forward_stacked_predictor_is_bit_identical(HipPlanner, monkeypatch, shape):
    """Round 3: the fused forward launch takes the class-default STACKED predictor (4 x 180, paule/models.py:335-339) in front of
    a 720-wide embedder (model set B) -- its four recurrences and the three projections between them as further roles at the
    predictor's width, the mel head on the top layer, the embedder's roles at its own width.  Every forward stash of every layer,
    the pooled mel, the losses and the plan equal the per-layer path's (32-row kernels) bit for bit; full groups with a graph, a
    ragged batch eagerly."""
    B, T = shape["B"], shape["T"]
    wl = synthetic.make_workload(B, T, "B")
    bufs = [f"pred.{k}{l}" for l in range(4) for k in "hcG"] + ["mel", "mel_tm", "emb.h0", "emb.c0", "emb.G0"]
    monkeypatch.setenv("PAULE_HIP_FUSED_OCC2", "0")   # lstm_fused.hip's launch (since round 4 the planner's choice for set B is the two-per-CU one: its own test above)
    monkeypatch.setenv("PAULE_HIP_DEBUG", "stop_after_fwd")
    e = _fused_pair(HipPlanner, monkeypatch, wl, B, T, ("0", "1"), 1, False)
    assert e["1"].plan_info()["fwd_per_cu"] == 1
    for name in bufs:
        np.testing.assert_array_equal(_n(e["1"].debug_read(name)), _n(e["0"].debug_read(name)), err_msg=name)
    monkeypatch.delenv("PAULE_HIP_DEBUG")
    e = _fused_pair(HipPlanner, monkeypatch, wl, B, T, ("0", "1"), 3, shape["graph"])
    np.testing.assert_array_equal(e["1"].losses, e["0"].losses)
    np.testing.assert_array_equal(_n(e["1"].get_cp()), _n(e["0"].get_cp()))


@pytest.mark.parametrize("extra", [dict(objective="semvec"), dict(objective="acoustic_semvec", smiling=True),
                                   dict(objective="acoustic_semvec", classifier=True, past=True)])
def test_fused_launches_other_objectives_vs_oracle(HipPlanner, extra):
    """The fused forward + backward launches (the default at 64 rows) under the `semvec` objective (no mel term: the backward mel
    head starts from the embedder's gradient alone) and with the smiling projection: model gradient of the first iteration against
    the exact float64 oracle (<= 2 %), and the plan after 6 iterations against the torch oracle at the bf16 bars."""
    from oracle import manual as mo
    B, T, H = 64, 40, 96
    objective, smiling = extra["objective"], extra.get("smiling", False)
    wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
    ex = mo.ManualPlanner(wl.pred_sd, wl.emb_sd, objective=objective, smiling=smiling)
    ex.set_targets(wl.target_mel.numpy(), wl.target_semvec.numpy())
    ex.set_cp(wl.cp0.numpy())
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective=objective, dtype="bf16", smiling=smiling)
    plan = eng.plan_info()
    assert plan["fused_fwd"] == 1 and plan["fused_bwd"] == 1, plan
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    if extra.get("classifier"):   # the speech-classifier term (its gradient joins dL/dY in front of the backward mel head) and a fixed past
        gen = torch.Generator().manual_seed(3)
        clf = {"linear.weight": torch.randn(1, 60, generator=gen, dtype=torch.float64) * 0.2, "linear.bias": torch.tensor([0.1], dtype=torch.float64)}
        past = wl.cp0[0, :6].clone()
        for pl in (ex, eng):
            pl.set_speech_classifier(clf)
            pl.set_past_cp(past if pl is eng else past.numpy())
    _, _, px = mo.loss_and_grad(ex.models, objective, ex.x, ex.target_mel, ex.target_semvec, ex.classifier)
    l1 = _n(eng.step(1))
    eng.synchronize()
    dX = _n(eng.debug_read("dX")).reshape(T, B, 32)[:, :, :30].transpose(1, 0, 2)
    err = np.linalg.norm(dX - px["grad_model"]) / np.linalg.norm(px["grad_model"])
    assert err <= 2e-2, err
    lh = np.concatenate([l1, _n(eng.step(5))])
    eng.synchronize()
    lo = ex.step(6)
    np.testing.assert_allclose(lh[:, :, :7], lo[:, :, :7], rtol=LOSS_RTOL_BF16, atol=5e-3)
    d = np.abs(_n(eng.get_cp()) - ex.get_cp())
    assert d.mean() <= 0.05 * 0.01 * 6 and d.max() <= 0.5 * 0.01 * 6, (d.mean(), d.max())


def test_fused_backward_after_pipelined_forward(HipPlanner, monkeypatch):
    """PAULE_HIP_FUSED=2: the fused backward launch behind a chunk-pipelined forward pass (small batch, 16-row forward kernels): the
    backward launch does not care how the stashes were made; the plan stays with the per-layer path's to the bf16 exchange rounding."""
    B, T, H = 40, 50, 96
    wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
    out = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("PAULE_HIP_FUSED", mode)
        monkeypatch.setenv("PAULE_HIP_FUSED_MIN_B", "1")
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
        plan = eng.plan_info()
        assert plan["fused_fwd"] == 0 and plan["fused_bwd"] == (1 if mode == "2" else 0), (mode, plan)
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        loss = _n(eng.step(6))
        eng.synchronize()
        out[mode] = (loss, _n(eng.get_cp()), _n(eng.debug_read("dX")))
    assert _cos(out["2"][2], out["0"][2]) >= 0.99999
    np.testing.assert_allclose(out["2"][0], out["0"][0], rtol=1e-4, atol=1e-6)
    assert np.abs(out["2"][1] - out["0"][1]).max() <= 2e-4


@pytest.mark.parametrize("shape", [dict(B=256, T=300, H=720), dict(B=48, T=41, H=96)])
def test_fused_backward_matches_per_layer_path(HipPlanner, monkeypatch, shape):
    """The fused backward launch (PAULE_HIP_FUSED=3: embedder recurrences, their dL/dh product and the backward mel head in
    reduce-scatter form, the predictor's recurrence) against the per-layer path.  The embedder's top layer is the same
    arithmetic (its dA stash is bit-identical); below it dL/dh crosses the exchange as 23 bf16 partial tiles instead of one f32
    sum rounded once, so the model gradient agrees to bf16 noise: cosine >= 0.99999, and the plans stay together."""
    B, T, H = shape["B"], shape["T"], shape["H"]
    wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
    e = _fused_pair(HipPlanner, monkeypatch, wl, B, T, ("0", "3"), 1, False)
    np.testing.assert_array_equal(_n(e["3"].debug_read("emb.G1")), _n(e["0"].debug_read("emb.G1")))
    assert _cos(e["3"].debug_read("dX"), e["0"].debug_read("dX")) >= 0.99999
    assert _cos(e["3"].debug_read("emb.G0"), e["0"].debug_read("emb.G0")) >= 0.9999
    e = _fused_pair(HipPlanner, monkeypatch, wl, B, T, ("0", "3"), 6, True)
    np.testing.assert_allclose(e["3"].losses, e["0"].losses, rtol=1e-4, atol=1e-6)
    assert np.abs(_n(e["3"].get_cp()) - _n(e["0"].get_cp())).max() <= 2e-4   # lr = 0.01: 2 % of one step


@pytest.mark.parametrize("shape", [dict(B=64, T=41, set="A"), dict(B=100, T=33, set="A"), dict(B=128, T=300, set="A"), dict(B=64, T=30, set="B")])
def test_fused_backward_same_xcd_exchange_is_bit_identical(HipPlanner, monkeypatch, shape):
    """Round 4: a 32-row fused backward role whose workgroups find themselves on one XCD hands its OWN partial tiles over through that
    XCD's L2 (plain stores, nt LDS-DMA, a second plain flag set; lstm_fused.hip: fused_lstm_bwd) -- same tiles, same order of every
    sum.  Against the write-through form (PAULE_HIP_FUSED_XCD=0): every layer's dA, dL/dCP, the losses and the plan after four
    iterations are bit-identical; full and ragged groups, one and two chains, the two-width launch of model set B."""
    B, T = shape["B"], shape["T"]
    wl = synthetic.make_workload(B, T, shape["set"])
    names = ["emb.G0", "pred.G0", "dX"] + (["emb.G1"] if shape["set"] == "A" else ["pred.G3"])
    out = {}
    for xcd in ("0", "1"):
        monkeypatch.setenv("PAULE_HIP_FUSED_XCD", xcd)
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
        plan = eng.plan_info()
        assert plan["fused_bwd"] == 1 and plan["fused_rows"] == 32, plan
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        l1 = _n(eng.step(1))
        eng.synchronize()
        bufs = {k: _n(eng.debug_read(k)) for k in names}
        l4 = _n(eng.step(3))
        eng.synchronize()
        out[xcd] = (l1, bufs, l4, _n(eng.get_cp()))
        eng.close()
    monkeypatch.delenv("PAULE_HIP_FUSED_XCD")
    np.testing.assert_array_equal(out["1"][0], out["0"][0])
    for k in names:
        np.testing.assert_array_equal(out["1"][1][k], out["0"][1][k], err_msg=k)
    np.testing.assert_array_equal(out["1"][2], out["0"][2])
    np.testing.assert_array_equal(out["1"][3], out["0"][3])


@pytest.mark.parametrize("shape", [dict(B=256, T=60, graph=True), dict(B=70, T=31, graph=False)])
def test_fused_backward_stacked_predictor_tracks_the_per_layer_path(HipPlanner, monkeypatch, shape):
    """Round 3: the fused BACKWARD launch takes the class-default stacked predictor too (model set B: four recurrences and three dL/dh
    product roles at the predictor's width, the backward mel head at both widths, the embedder's recurrence at its own) and is the
    default for such models at every batch the roles fit (cfg3_setB: 5.47 -> 4.52 ms per iteration).  Against the per-layer backward
    (PAULE_HIP_FUSED_BWD_STACKED=0, same fused forward launch): the embedder's dA stash bit-identical (the top layer: same arithmetic),
    every predictor layer's dA and dL/dCP to bf16 noise (cosine >= 0.99999), six iterations of the plan within 2 % of one step."""
    B, T = shape["B"], shape["T"]
    wl = synthetic.make_workload(B, T, "B")
    out = {}
    for stacked in ("1", "0"):
        monkeypatch.setenv("PAULE_HIP_FUSED_BWD_STACKED", stacked)
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", use_graph=shape["graph"])
        plan = eng.plan_info()
        assert plan["fused_fwd"] == 1 and plan["fused_bwd"] == int(stacked) and plan["fused_rows"] == 32, plan
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        l1 = _n(eng.step(1))
        eng.synchronize()
        bufs = {k: _n(eng.debug_read(k)) for k in ("emb.G0", "pred.G3", "pred.G2", "pred.G1", "pred.G0", "dX")}
        l6 = _n(eng.step(5))
        eng.synchronize()
        out[stacked] = (l1, bufs, l6, _n(eng.get_cp()))
        eng.close()
    monkeypatch.delenv("PAULE_HIP_FUSED_BWD_STACKED")
    np.testing.assert_array_equal(out["1"][0], out["0"][0])
    np.testing.assert_array_equal(out["1"][1]["emb.G0"], out["0"][1]["emb.G0"])
    for k in ("pred.G3", "pred.G2", "pred.G1", "pred.G0", "dX"):
        assert _cos(out["1"][1][k], out["0"][1][k]) >= 0.99999, k
    np.testing.assert_allclose(out["1"][2], out["0"][2], rtol=1e-4, atol=1e-6)
    assert np.abs(out["1"][3] - out["0"][3]).max() <= 2e-4


# ---- fused launches on 16-row tiles (lstm_fused16.h): batches of up to 16 rows -- the reference's own B = 1 ---------------------
def _pair16(HipPlanner, monkeypatch, wl, B, T, iters, use_graph, other, stop_after_fwd=False):
    """The same plan with the 16-row fused launches (the default for up to 16 rows) and on `other`: "pipelines" (PAULE_HIP_FUSED16=0:
    the chunk pipelines of the 16-row per-layer kernels, lstm_persist16.hip) or "fused32" (the 32-row roles of lstm_fused.hip,
    forced down to this batch with PAULE_HIP_FUSED_MIN_B=1); the plan the library made is asserted."""
    out = {}
    for which in ("fused16", other):
        for k in ("PAULE_HIP_FUSED16", "PAULE_HIP_FUSED_MIN_B", "PAULE_HIP_FUSED"):
            monkeypatch.delenv(k, raising=False)
        if which == "pipelines":
            monkeypatch.setenv("PAULE_HIP_FUSED16", "0")
        elif which == "fused32":
            monkeypatch.setenv("PAULE_HIP_FUSED_MIN_B", "1")
            monkeypatch.setenv("PAULE_HIP_FUSED", "3")
        if stop_after_fwd:
            monkeypatch.setenv("PAULE_HIP_DEBUG", "stop_after_fwd")
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", use_graph=use_graph)
        plan = eng.plan_info()
        want = {"fused16": (16, 1, 1), "fused32": (32, 1, 1), "pipelines": (0, 0, 0)}[which]
        assert (plan["fused_rows"], plan["fused_fwd"], plan["fused_bwd"]) == want, (which, plan)
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        eng.losses = _n(eng.step(iters))
        eng.synchronize()
        out[which] = eng
    for k in ("PAULE_HIP_FUSED16", "PAULE_HIP_FUSED_MIN_B", "PAULE_HIP_FUSED", "PAULE_HIP_DEBUG"):
        monkeypatch.delenv(k, raising=False)
    return out


@pytest.mark.parametrize("shape", [dict(B=16, T=61, H=720, graph=True), dict(B=1, T=300, H=720, graph=True), dict(B=5, T=41, H=96, graph=False),
                                   dict(B=9, T=14, H=720, graph=False), dict(B=32, T=61, H=720, graph=True), dict(B=48, T=40, H=720, graph=False),
                                   dict(B=20, T=33, H=720, graph=True), dict(B=37, T=30, H=96, graph=False)])
def test_fused16_launches_equal_pipelines_forward_and_fused32_throughout(HipPlanner, monkeypatch, shape):
    """Batches of up to 48 rows (one to three 16-row groups, each a set of workgroups of its own in every LSTM role: the reference's
    B = 1, cfg5's 16 per GPU, ragged batches of 20 and 37) run both fused launches with the LSTM
    roles on 16-row tiles, each role's own exchange in the verified same-XCD form (plain stores / nt loads, a second flag set; the
    write-through flags other roles wait for are raised a step late).
      * Forward: the arithmetic of the 16-row per-layer kernels, instruction for instruction -- every stash, the pooled mel and the
        first losses carry the same bits as the chunk pipelines'; the first model gradient agrees with theirs to bf16 noise
        (cosine >= 0.99999: below the embedder's top layer dL/dh crosses the product roles' bf16 exchanges).
      * Whole plan: bit-identical to the 32-row fused launches over six iterations (losses, dL/dCP, CP).  The recurrences hand over
        the same bf16 partial tiles and sum them in f32 in another order (wave-wise instead of source by source) -- sums of 23 bf16
        values are exact in f32, so the order does not show.
    Odd T, the shortest plannable T, B = 1 at the reference's length, graph and eager."""
    B, T, H = shape["B"], shape["T"], shape["H"]
    wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
    e = _pair16(HipPlanner, monkeypatch, wl, B, T, 1, False, "pipelines", stop_after_fwd=True)
    for name in FWD_BUFFERS:
        np.testing.assert_array_equal(_n(e["fused16"].debug_read(name)), _n(e["pipelines"].debug_read(name)), err_msg=name)
    e = _pair16(HipPlanner, monkeypatch, wl, B, T, 1, False, "pipelines")
    np.testing.assert_array_equal(e["fused16"].losses, e["pipelines"].losses)
    assert _cos(e["fused16"].debug_read("emb.G1"), e["pipelines"].debug_read("emb.G1")) >= 0.999999
    assert _cos(e["fused16"].debug_read("emb.G0"), e["pipelines"].debug_read("emb.G0")) >= 0.9999
    assert _cos(e["fused16"].debug_read("dX"), e["pipelines"].debug_read("dX")) >= 0.99999
    e = _pair16(HipPlanner, monkeypatch, wl, B, T, 6, shape["graph"], "fused32")
    np.testing.assert_array_equal(e["fused16"].losses, e["fused32"].losses)
    np.testing.assert_array_equal(_n(e["fused16"].debug_read("dX")), _n(e["fused32"].debug_read("dX")))
    np.testing.assert_array_equal(_n(e["fused16"].get_cp()), _n(e["fused32"].get_cp()))
    assert (e["fused16"].losses[-1, :, 0] < e["fused16"].losses[0, :, 0]).all()


@pytest.mark.parametrize("shape", [dict(B=16, T=60, graph=True), dict(B=1, T=300, graph=True), dict(B=7, T=31, graph=False),
                                   dict(B=30, T=40, graph=True), dict(B=48, T=26, graph=False)])
def test_fused16_stacked_predictor_of_another_width(HipPlanner, monkeypatch, shape):
    """Model set B (the class-default stacked 4 x 180 predictor in front of a 720-wide embedder) at up to 16 rows: BOTH launches
    fused on 16-row tiles -- backward: four predictor recurrences and the three dL/dh product roles between them at the predictor's
    width, the backward mel head at both widths, the embedder's recurrence at its own (round 3; the 32-row backward launch still
    takes one predictor layer).  Forward: every stash of every layer and the pooled mel bit-identical to the chunk pipelines; first
    iteration: identical losses, dL/dCP cosine >= 0.99999; six iterations: losses and CP stay with the pipelines' (bf16 exchange
    roundings only)."""
    B, T = shape["B"], shape["T"]
    wl = synthetic.make_workload(B, T, "B")
    bufs = [f"pred.{k}{l}" for l in range(4) for k in "hcG"] + ["mel", "mel_tm", "emb.h0", "emb.c0", "emb.G0"]
    e = _pair16(HipPlanner, monkeypatch, wl, B, T, 1, False, "pipelines", stop_after_fwd=True)
    for name in bufs:
        np.testing.assert_array_equal(_n(e["fused16"].debug_read(name)), _n(e["pipelines"].debug_read(name)), err_msg=name)
    e = _pair16(HipPlanner, monkeypatch, wl, B, T, 1, False, "pipelines")
    np.testing.assert_array_equal(e["fused16"].losses, e["pipelines"].losses)
    np.testing.assert_array_equal(_n(e["fused16"].debug_read("emb.G0")), _n(e["pipelines"].debug_read("emb.G0")))   # the top layer: same arithmetic
    for name in ("pred.G3", "pred.G0", "dX"):
        assert _cos(e["fused16"].debug_read(name), e["pipelines"].debug_read(name)) >= 0.99999, name
    e = _pair16(HipPlanner, monkeypatch, wl, B, T, 6, shape["graph"], "pipelines")
    np.testing.assert_allclose(e["fused16"].losses, e["pipelines"].losses, rtol=1e-4, atol=1e-6)
    assert np.abs(_n(e["fused16"].get_cp()) - _n(e["pipelines"].get_cp())).max() <= 2e-4   # lr = 0.01: 2 % of one step
    assert (e["fused16"].losses[-1, :, 0] < e["fused16"].losses[0, :, 0]).all()


def test_fused16_launches_are_reproducible_run_to_run(HipPlanner):
    """Ten fresh engines, one iteration each of cfg5's shape (16 x 2000 frames, Paule's models): every layer's dA and dL/dCP carry
    the same bits in all ten.  Round 3 found the first version of the 16-row backward role NOT reproducible (one run in four
    differed in the low bits of dL/dCP): waves 2 and 3 skipped their surplus partial tile (46 tiles over 4 waves) with a run-time
    branch inside the MFMA sequence, and the tile before it then came out with a stale accumulator register.  The tile is now
    computed and only its store dropped (lstm_fused16.h); the tolerance-based parity tests had not seen it."""
    import hashlib
    B, T = 16, 2000
    for mset, names in (("A", ("emb.G1", "emb.G0", "pred.G0", "dX")), ("B", ("emb.G0", "pred.G3", "pred.G2", "pred.G1", "pred.G0", "dX"))):
        wl = synthetic.make_workload(B, T, mset)   # set B: the stacked 4 x 180 predictor in front of the 720-wide embedder (two-width launches)
        seen = {}
        for k in range(10 if mset == "A" else 6):
            eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
            assert eng.plan_info()["fused_rows"] == 16
            eng.set_targets(wl.target_mel, wl.target_semvec)
            eng.set_cp(wl.cp0)
            eng.step(1, return_loss=False)
            eng.synchronize()
            for name in names:
                seen.setdefault(name, set()).add(hashlib.md5(_n(eng.debug_read(name).float()).tobytes()).hexdigest())
            eng.close()
        assert all(len(v) == 1 for v in seen.values()), (mset, {k: len(v) for k, v in seen.items()})


@pytest.mark.parametrize("shape", [dict(B=256, T=150, set="A", what="fused forward launch + streamed 32-row backward sweeps (cfg3's schedule)"),
                                   dict(B=100, T=300, set="A", what="32-row fused forward + backward launches"),
                                   dict(B=40, T=300, set="A", what="16-row fused launches, three groups"),
                                   dict(B=40, T=300, set="C", what="chunk pipelines of the 16-row sweeps (an embedder variant keeps them)"),
                                   dict(B=256, T=100, set="B", what="fused forward + backward launches of two widths (stacked predictor)")])
def test_every_bf16_schedule_is_reproducible_run_to_run(HipPlanner, monkeypatch, shape):
    """The check that found the 16-row fused role's stale accumulator register, applied to the other bf16 schedules the planner picks:
    six fresh engines, two iterations each -- every layer's dA, dL/dCP and the updated CP carry the same bits in all six.  (The predictor's
    dA is asked for: where dL/dCP rides along in the sweep a planning iteration does not keep it by default, round 5.)"""
    import hashlib
    monkeypatch.setenv("PAULE_HIP_BWD_XT", "1")
    B, T = shape["B"], shape["T"]
    wl = synthetic.make_workload(B, T, shape["set"])
    n_pred = 4 if shape["set"] == "B" else 1
    names = [f"pred.G{l}" for l in range(n_pred)] + ["emb.G0", "dX"]
    seen = {}
    for k in range(6):
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        eng.step(2, return_loss=False)
        eng.synchronize()
        for name in names:
            seen.setdefault(name, set()).add(hashlib.md5(_n(eng.debug_read(name).float()).tobytes()).hexdigest())
        seen.setdefault("cp", set()).add(hashlib.md5(_n(eng.get_cp()).tobytes()).hexdigest())
        eng.close()
    assert all(len(v) == 1 for v in seen.values()), (shape["what"], {k: len(v) for k, v in seen.items()})


@pytest.mark.parametrize("shape", [dict(B=16, T=61), dict(B=40, T=30), dict(B=1, T=24)])
def test_fused16_stash_prefetchers_change_nothing(HipPlanner, monkeypatch, shape):
    """Round 5: the 16-row fused backward launch gets stash prefetcher workgroups behind every recurrence set (block-table entries with a
    prefetcher index, lstm_fused16.h: fused_pf_bwd16) -- they only read, paced on the role's flags.  With PAULE_HIP_FUSED16_PF=0, 2 and 8 the
    loss log, every layer's dA, dL/dCP and CP are the same bits; the census of the launch counts them (a plan with them still runs)."""
    B, T = shape["B"], shape["T"]
    wl = synthetic.make_workload(B, T, "A")
    out = {}
    for pf in ("0", "2", "8"):
        monkeypatch.setenv("PAULE_HIP_FUSED16_PF", pf)
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
        assert eng.plan_info()["fused_rows"] == 16
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        l1 = _n(eng.step(1))
        eng.synchronize()
        bufs = {k: _n(eng.debug_read(k)) for k in ("emb.G1", "emb.G0", "pred.G0", "dX")}
        l3 = _n(eng.step(3))
        eng.synchronize()
        out[pf] = (l1, bufs, l3, _n(eng.get_cp()))
        eng.close()
    for pf in ("2", "8"):
        np.testing.assert_array_equal(out[pf][0], out["0"][0])
        for k in out["0"][1]:
            np.testing.assert_array_equal(out[pf][1][k], out["0"][1][k], err_msg=f"{k} pf={pf}")
        np.testing.assert_array_equal(out[pf][2], out["0"][2])
        np.testing.assert_array_equal(out[pf][3], out["0"][3])


@pytest.mark.parametrize("shape", [dict(B=3, T=24, H=720), dict(B=16, T=40, H=96), dict(B=1, T=64, H=720)])
def test_fused16_vs_oracle_and_rounding_emulation(HipPlanner, shape):
    """The 16-row fused launches against the float64 oracle (model gradient <= 2 %) and against the rounding emulation
    (oracle/bf16_emul.py: <= 3e-3 -- the path differs from the reference arithmetic by the declared roundings only), then five
    iterations of the plan against both (the bars of test_bf16_path_equals_rounding_emulation)."""
    from oracle import bf16_emul as be
    from oracle import manual as mo
    B, T, H = shape["B"], shape["T"], shape["H"]
    wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
    em = be.EmulPlanner(wl.pred_sd, wl.emb_sd, objective="acoustic_semvec")
    ex = mo.ManualPlanner(wl.pred_sd, wl.emb_sd, objective="acoustic_semvec")
    for o in (em, ex):
        o.set_targets(wl.target_mel.numpy(), wl.target_semvec.numpy())
        o.set_cp(wl.cp0.numpy())
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", use_graph=False)
    assert eng.plan_info()["fused_rows"] == 16, eng.plan_info()
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    eng.step(1, return_loss=False)
    eng.synchronize()
    dX = _tm(eng.debug_read("dX"), T, 16, 32, B, 30)
    _, _, pe = be.loss_and_grad(em.models, "acoustic_semvec", em.x, em.target_mel, em.target_semvec)
    _, _, px = mo.loss_and_grad(ex.models, "acoustic_semvec", ex.x, ex.target_mel, ex.target_semvec)
    nrm = np.linalg.norm(px["grad_model"])
    err_emul, err_exact = np.linalg.norm(dX - pe["dX"]) / nrm, np.linalg.norm(dX - px["grad_model"]) / nrm
    print(f"[fused16] dL/dCP relative error vs emulation {err_emul:.2e}, vs exact oracle {err_exact:.2e}")
    assert err_emul <= 3e-3 and err_exact <= 2e-2, (err_emul, err_exact)
    eng.step(4, return_loss=False)
    eng.synchronize()
    em.step(5)
    ex.step(5)
    cp = _n(eng.get_cp())
    d_emul, d_exact = np.abs(cp - em.get_cp()), np.abs(cp - ex.get_cp())
    print(f"[fused16] CP after 5 iterations: mean |diff| vs emulation {d_emul.mean():.2e} (max {d_emul.max():.2e}), vs exact oracle {d_exact.mean():.2e}")
    assert d_emul.mean() <= 1e-5 and d_emul.max() <= 1e-3 and d_exact.max() <= 0.5 * 0.01 * 5, (d_emul.mean(), d_emul.max(), d_exact.max())


# ---- continued learning of the predictive model (paule/paule.py:1353-1379) --------------------------
def _unpad_grad(flat, nblk, R, C, Rp, Cp):
    a = _n(flat).reshape(nblk, Rp, Cp)[:, :R, :C]
    return a.reshape(nblk * R, C)


def _train_engine(HipPlanner, g, dtype, batch=None):
    sd = state_dict_from(g, "pred")
    return HipPlanner(sd, None, batch=int(batch or g["N"]), n_frames=int(g["T"]), objective="acoustic", dtype=dtype), sd


@pytest.mark.parametrize("batch", [None, 21])
def test_train_pred_f32_vs_reference_fixture(HipPlanner, golden_train, batch):
    """pl_train_pred_step against the reference's own run of the mini-batch loop body (ForwardModel, RMSELoss(eps=0),
    torch.optim.Adam(lr=0.001); tests/golden/train_small.npz): loss of every step rtol 1e-5; every parameter gradient of
    step 0 (padded compute layout, unpadded here) within 2e-5 of the largest entry; parameters after 1 and 6 steps.
    Adam divides by sqrt(v): where |g| is of the order of eps = 1e-8 the f32 noise in g changes the update by a
    fraction of lr, so parameters are held to 2 % of the lr * steps budget at the worst element and 1e-6 on average.
    batch = 21: an engine built for more utterances than the mini-batch holds (rows beyond n_rows must not contribute)."""
    g = golden_train
    eng, sd = _train_engine(HipPlanner, g, "f32", batch)
    L, H, I, M = 2, 24, 30, 60
    Hp, Ip, Mp = 32, 32, 64
    losses = []
    for k in range(int(g["n_steps"])):
        j = g[f"batch_{k}"]
        losses.append(float(eng.train_pred_step(g["cps"][j], g["prod_mel"][j])))
        if k == 0:
            for l in range(L):
                in_l, in_p = (I, Ip) if l == 0 else (H, Hp)
                for tag, key, shape in (("i", "weight_ih", (4, H, in_l, Hp, in_p)), ("r", "weight_hh", (4, H, H, Hp, Hp))):
                    ref = g[f"grad_step0/lstm.{key}_l{l}"]
                    got = _unpad_grad(eng.debug_read(f"pred.{tag}{l}"), *shape)
                    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5 * np.abs(ref).max(), err_msg=f"{key}_l{l}")
                    assert _cos(got, ref) > 1 - 1e-9
                gb = _n(eng.debug_read(f"pred.d{l}")).reshape(4, Hp)[:, :H].reshape(-1)
                for key in ("bias_ih", "bias_hh"):
                    ref = g[f"grad_step0/lstm.{key}_l{l}"]
                    np.testing.assert_allclose(gb, ref, rtol=0, atol=2e-5 * np.abs(ref).max(), err_msg=f"{key}_l{l}")
            ref = g["grad_step0/post_linear.weight"]
            got = _n(eng.debug_read("pred.gWlin")).reshape(Mp, Hp)[:M, :H]
            np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5 * np.abs(ref).max())
            ref = g["grad_step0/post_linear.bias"]
            np.testing.assert_allclose(_n(eng.debug_read("pred.gblin"))[:M], ref, rtol=0, atol=2e-5 * np.abs(ref).max())
        if k in (0, int(g["n_steps"]) - 1):
            got_sd = eng.get_weights("pred")
            assert set(got_sd) == set(sd)
            for name, w in got_sd.items():
                d = np.abs(_n(w) - g[f"after_{k + 1}/{name}"])
                assert d.max() <= 0.02 * 1e-3 * (k + 1) and d.mean() <= 1e-6, (name, k, d.max(), d.mean())
    np.testing.assert_allclose(losses, g["losses"], rtol=LOSS_RTOL_F32)
    # the planner reads the refreshed compute copies: forward at the training inputs with the trained weights
    n = int(g["N"])
    cp = np.zeros((eng.B, int(g["T"]), I), dtype=np.float32)
    cp[:n] = g["cps"]
    eng.set_cp(cp)
    mel, _ = eng.get_pred(with_semvec=False)
    np.testing.assert_allclose(_n(mel)[:n], g["final_pred_mel"], atol=1e-4, rtol=0)


def test_train_pred_shorter_samples_vs_oracle(HipPlanner, golden_train):
    """Mini-batches padded to fewer frames than the engine was built for (pad_batch_online pads to the longest sample of a
    batch, paule/paule.py:1362-1368): 26 and 33 (odd: last frame has no mel partner) of 40 frames, against the oracle."""
    g = golden_train
    eng, sd = _train_engine(HipPlanner, g, "f32", batch=9)
    tr = op.OracleTrainer(op.forward_model_from_state_dict(sd))
    lh, lo = [], []
    for t in (26, 33, 40, 26):
        cp, mel = g["cps"][:4, :t], g["prod_mel"][:4, :t // 2]
        lo.append(float(tr.train_pred_step(cp, mel)))
        lh.append(float(eng.train_pred_step(cp.copy(), mel.copy())))
    np.testing.assert_allclose(lh, lo, rtol=LOSS_RTOL_F32)
    ref = tr.state_dict()
    for name, w in eng.get_weights("pred").items():
        d = np.abs(_n(w) - ref[name].numpy())
        assert d.max() <= 0.02 * 1e-3 * 4 and d.mean() <= 1e-6, (name, d.max(), d.mean())


def test_train_pred_bf16_vs_reference_fixture(HipPlanner, golden_train):
    """bf16 activations / weights with f32-accumulated gradients and f64 masters: loss curve rtol 2e-2, gradient cosine
    >= 0.999 for the big matrices, parameters within 50 % of the lr * steps budget at the worst element (Adam is sign-like
    where a gradient is bf16 noise) and 5 % on average."""
    g = golden_train
    eng, _ = _train_engine(HipPlanner, g, "bf16")
    losses = []
    for k in range(int(g["n_steps"])):
        j = g[f"batch_{k}"]
        losses.append(float(eng.train_pred_step(g["cps"][j], g["prod_mel"][j])))
        if k == 0:
            for l, shape in ((0, (4, 24, 30, 32, 32)), (1, (4, 24, 24, 32, 32))):
                got = _unpad_grad(eng.debug_read(f"pred.i{l}"), *shape)
                assert _cos(got, g[f"grad_step0/lstm.weight_ih_l{l}"]) >= COS_BF16
                got = _unpad_grad(eng.debug_read(f"pred.r{l}"), 4, 24, 24, 32, 32)
                assert _cos(got, g[f"grad_step0/lstm.weight_hh_l{l}"]) >= COS_BF16
    np.testing.assert_allclose(losses, g["losses"], rtol=LOSS_RTOL_BF16)
    n_steps = int(g["n_steps"])
    for name, w in eng.get_weights("pred").items():
        d = np.abs(_n(w) - g[f"after_{n_steps}/{name}"])
        assert d.max() <= 0.5 * 1e-3 * n_steps and d.mean() <= 0.05 * 1e-3 * n_steps, (name, d.max(), d.mean())


@pytest.mark.parametrize("spec", [dict(L=1, H=96), dict(L=2, H=192), dict(L=1, H=256), dict(L=1, H=720)])
def test_train_pred_bf16_weight_gradients_across_widths(HipPlanner, spec):
    """W_hh / W_ih gradients of one bf16 continued-learning step against the oracle trainer (torch autograd, float64) for narrow
    and wide predictive models: relative error <= 2 % per tensor (the backward-data recurrence that feeds these products is the
    planner's own; see test_bf16_model_gradient_across_hidden_sizes for why widths below Hp = 288 get their own case)."""
    L, H, B, T, n = spec["L"], spec["H"], 16, 40, 8
    wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=L, hidden_size=H), emb=dict(num_lstm_layers=1, hidden_size=24))
    tr = op.OracleTrainer(op.forward_model_from_state_dict(wl.pred_sd))
    tr.train_pred_step(wl.cp0[:n], wl.target_mel[:n])
    ref = {k: _n(v) for k, v in tr.gradients().items()}
    eng = HipPlanner(wl.pred_sd, None, batch=B, n_frames=T, objective="acoustic", dtype="bf16")
    eng.train_pred_step(wl.cp0[:n], wl.target_mel[:n])
    eng.synchronize()
    Hp = -(-H // 32) * 32
    for l in range(L):
        in_dim = 30 if l == 0 else H
        in_p = 32 if l == 0 else Hp
        for kind, name, cols, cols_p in (("r", f"lstm.weight_hh_l{l}", H, Hp), ("i", f"lstm.weight_ih_l{l}", in_dim, in_p)):
            got = _unpad_grad(eng.debug_read(f"pred.{kind}{l}"), 4, H, cols, Hp, cols_p)
            err = np.linalg.norm(got - ref[name]) / np.linalg.norm(ref[name])
            assert err <= 2e-2, (name, err)


def test_weight_gradients_bf16_mfma_equals_f32_mfma(HipPlanner, monkeypatch):
    """The weight-gradient products of a bf16 training step on the bf16 MFMA (gemm_tn_bf16_kernel: transposed LDS reads of the
    k-major operands) against the same products on the exact f32 MFMA (PAULE_HIP_TN_BF16=0): products of bf16 values are exact
    in f32 either way and both accumulate in f32, so the gradients agree to summation order -- 1e-5 of each tensor's largest
    entry; H = 720 (split-K, the 128-wide tiles) and a ragged small model (64-wide tiles, partial tiles in m and n)."""
    for kw, B, T, n in ((dict(set="A"), 24, 60, 8), (dict(pred=dict(num_lstm_layers=2, hidden_size=40)), 5, 31, 5)):
        wl = synthetic.make_workload(B, T, kw.get("set"), pred=kw.get("pred"), emb=dict(num_lstm_layers=1, hidden_size=24)) if "pred" in kw \
            else synthetic.make_workload(B, T, "A")
        grads = {}
        for form in ("1", "0"):
            monkeypatch.setenv("PAULE_HIP_TN_BF16", form)
            eng = HipPlanner(wl.pred_sd, None, batch=B, n_frames=T, objective="acoustic", dtype="bf16")
            eng.train_pred_step(wl.cp0[:n], wl.target_mel[:n])
            eng.synchronize()
            L = sum(1 for k in wl.pred_sd if k.startswith("lstm.weight_hh_l"))
            grads[form] = {f"{kind}{l}": _n(eng.debug_read(f"pred.{kind}{l}")) for l in range(L) for kind in ("i", "r")}
        for name, g1 in grads["1"].items():
            g0 = grads["0"][name]
            assert np.abs(g0).max() > 0, name
            assert np.abs(g1 - g0).max() <= 1e-5 * np.abs(g0).max(), (name, np.abs(g1 - g0).max(), np.abs(g0).max())


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_train_pred_set_a_vs_oracle(HipPlanner, dtype):
    """Paule's default ForwardModel (L = 1, H = 720: persistent sweeps, 46 / 23 workgroups per group) trained for 3
    mini-batch steps of 8 samples inside an engine built for 24 utterances, against the oracle trainer; then planning
    continues on the trained weights (pred_mel of the planner == oracle model's output)."""
    wl = synthetic.make_workload(24, 60, "A")
    tr = op.OracleTrainer(op.forward_model_from_state_dict(wl.pred_sd))
    eng = HipPlanner(wl.pred_sd, None, batch=24, n_frames=60, objective="acoustic", dtype=dtype)
    lo, lh = [], []
    for k in range(3):
        j = np.arange(8 * k, 8 * k + 8)
        lo.append(float(tr.train_pred_step(wl.cp0[j], wl.target_mel[j])))
        lh.append(float(eng.train_pred_step(wl.cp0[j], wl.target_mel[j])))
    eng.synchronize()
    f32 = dtype == "f32"
    np.testing.assert_allclose(lh, lo, rtol=LOSS_RTOL_F32 if f32 else LOSS_RTOL_BF16)
    ref = tr.state_dict()
    for name, w in eng.get_weights("pred").items():
        d = np.abs(_n(w) - ref[name].numpy())
        if f32:
            assert d.max() <= 0.02 * 1e-3 * 3 and d.mean() <= 1e-6, (name, d.max(), d.mean())
        else:
            assert d.max() <= 2.0 * 1e-3 * 3 and d.mean() <= 0.1 * 1e-3 * 3, (name, d.max(), d.mean())
    eng.set_cp(wl.cp0)
    mel, _ = eng.get_pred(with_semvec=False)
    with torch.no_grad():
        ref_mel = tr.pred_model(wl.cp0).numpy()
    np.testing.assert_allclose(_n(mel), ref_mel, atol=1e-4 if f32 else 3e-2, rtol=0)


def test_paule_continue_learning_hip_equals_oracle_engine():
    """Paule.plan_resynth with continue_learning=True end to end: the HIP planner (planning + pl_train_pred_step between the
    outer iterations) against the same host code driving the CPU oracle: epoch losses of the continued learning, the trained
    parameters and the plan that was made on them."""
    from oracle_engine import OracleEngine
    from paule_amd import paule as pp
    small = synthetic.make_workload(2, 24, None, pred=dict(num_lstm_layers=1, hidden_size=12),
                                    emb=dict(num_lstm_layers=1, hidden_size=10))

    def synth(cp):
        return np.zeros(100), 44100

    def melx(sig, sr):
        return np.full((12, 60), 0.25)

    out = []
    for factory in (None, lambda pm, em, **kw: OracleEngine(pm, em, **kw)):
        model = pp.Paule(pred_model={k: v.clone() for k, v in small.pred_sd.items()}, embedder=small.emb_sd,
                         planner_factory=factory, synthesizer=synth, mel_extractor=melx,
                         device=torch.device("cuda" if factory is None else "cpu"))
        res = model.plan_resynth(target_acoustic=small.target_mel.numpy(), initial_cp=small.cp0.numpy(), initialize_from=None,
                                 objective="acoustic", n_outer=2, n_inner=4, log_ii=2, continue_learning=True, n_batches=2,
                                 batch_size=2, n_epochs=3, seed=7, verbose=False)
        out.append((np.asarray(res.pred_model_loss), {k: _n(v) for k, v in model.pred_model.items()}, _n(res.planned_cp)))
    (lh, wh, ch), (lo, wo, co) = out
    np.testing.assert_allclose(lh, lo, rtol=1e-4)
    for k in wo:
        d = np.abs(wh[k] - wo[k])
        assert d.max() <= 0.02 * 1e-3 * 12 and d.mean() <= 2e-6, (k, d.max(), d.mean())
    np.testing.assert_allclose(ch, co, atol=1e-4, rtol=0)


# ---- inverse model: initial CP from the target mel (paule/paule.py:550-556) -------------------------
def _stub_pred(mel_dim=60, cp_dim=30):
    z = torch.zeros
    return {"lstm.weight_ih_l0": z(4, cp_dim), "lstm.weight_hh_l0": z(4, 1), "lstm.bias_ih_l0": z(4), "lstm.bias_hh_l0": z(4),
            "post_linear.weight": z(mel_dim, 1), "post_linear.bias": z(mel_dim)}


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_inverse_forward_vs_reference_fixture(HipPlanner, golden_inverse, dtype):
    """pl_inverse_forward against the reference's InverseModelMelTimeSmoothResidual run (tests/golden/inverse_small.npz;
    outputs up to |4|): f32 atol 5e-5 raw and clipped, and a 13-frame sequence in a handle built for 20; bf16 (LSTM and
    post_linear in bf16, convolutions f32): atol 0.08."""
    g = golden_inverse
    B, Tm = g["mel"].shape[:2]
    eng = HipPlanner(_stub_pred(), None, batch=B, n_frames=2 * Tm, dtype=dtype, inv_model=state_dict_from(g, "inv"))
    atol = 5e-5 if dtype == "f32" else 0.08
    np.testing.assert_allclose(_n(eng.inverse_forward(g["mel"], clip=False)), g["cp_raw"], atol=atol, rtol=0)
    np.testing.assert_allclose(_n(eng.inverse_forward(g["mel"], clip=True)), g["cp_clipped"], atol=atol, rtol=0)
    np.testing.assert_allclose(_n(eng.inverse_forward(g["mel"][:, :13].copy(), clip=False)), g["cp_raw_13"], atol=atol, rtol=0)


def test_inverse_model_module_default_shape_vs_oracle():
    """paule_amd.models.InverseModelMelTimeSmoothResidual(num_lstm_layers=1, hidden_size=720) -- what Paule instantiates
    (paule/paule.py:146) -- random init, forward on the GPU against the oracle model with the same state dict; B = 37 runs
    the LSTM through the f32 persistent sweeps (3 groups of 16 rows)."""
    from paule_amd import models
    torch.manual_seed(5)
    m = models.InverseModelMelTimeSmoothResidual(num_lstm_layers=1, hidden_size=720)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    mel = synthetic.make_workload(37, 64, "B").target_mel                    # (37, 32, 60)
    ref = op.inverse_model_from_state_dict(sd)
    with torch.no_grad():
        want = ref(mel).numpy()
    got = m(mel.float().cuda())
    assert got.shape == (37, 64, 30)
    np.testing.assert_allclose(_n(got), want, atol=2e-5, rtol=0)


def test_paule_with_embedder_variant_hip_equals_oracle_engine(golden_embvar):
    """Paule(embedder=MelEmbeddingModelMelSmoothResidualUpsampling(...)).plan_resynth on the device against the same host code
    driving the CPU oracle: the planned CP, the logged semantic losses and the final predicted semantic vector."""
    from oracle_engine import OracleEngine
    from paule_amd import models
    from paule_amd import paule as pp
    g = golden_embvar
    emb = models.MelEmbeddingModelMelSmoothResidualUpsampling(hidden_size=20, num_lstm_layers=2, post_upsampling_size=96).double()
    emb.load_state_dict(state_dict_from(g, "melsmooth/emb"))
    out = []
    for factory in (None, lambda pm, em, **kw: OracleEngine(pm, em, **kw)):
        model = pp.Paule(pred_model=state_dict_from(g, "pred"), embedder=emb, planner_factory=factory,
                         device=torch.device("cuda" if factory is None else "cpu"))
        res = model.plan_resynth(target_acoustic=g["target_mel"][:2], target_semvec=g["melsmooth/target_semvec"][:2],
                                 initial_cp=g["cp0"][:2], initialize_from=None, objective="acoustic_semvec", n_outer=1, n_inner=6,
                                 log_ii=2, continue_learning=False, verbose=False)
        out.append((_n(res.planned_cp), np.asarray(res.pred_semvec_loss_steps, dtype=np.float64), _n(res.pred_semvec)))
    (ch, lh, sh), (co, lo, so) = out
    np.testing.assert_allclose(ch, co, atol=1e-4, rtol=0)
    np.testing.assert_allclose(lh, lo, rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(sh, so, atol=2e-4, rtol=0)


def test_paule_somatosensory_hip_equals_oracle_engine(golden_soma):
    """Paule(use_somatosensory_feedback=True, ...).plan_resynth on the device against the same host code on the CPU oracle:
    planned CP, the logged tube losses and the final tube predictions."""
    from oracle_engine import OracleEngine
    from paule_amd import paule as pp
    g = golden_soma
    sdf = state_dict_from
    out = []
    for factory in (None, lambda pm, em, **kw: OracleEngine(pm, em, **kw)):
        model = pp.Paule(pred_model=sdf(g, "pred"), embedder=sdf(g, "emb"), use_somatosensory_feedback=True,
                         cp_tube_model=sdf(g, "cp_tube"), tube_mel_model=sdf(g, "tube_mel"), tube_embedder=sdf(g, "tube_emb"),
                         tube_extractor=lambda cp: np.full(cp.shape[:2] + (10,), 0.1), planner_factory=factory,
                         device=torch.device("cuda" if factory is None else "cpu"))
        res = model.plan_resynth(target_acoustic=g["target_mel"][:2], target_semvec=g["target_semvec"][:2], initial_cp=g["cp0"][:2],
                                 initialize_from=None, objective="semvec", n_outer=1, n_inner=6, log_ii=3, continue_learning=False,
                                 verbose=False)
        out.append([_n(res.planned_cp), np.asarray(res.pred_tube_mel_loss_steps, dtype=np.float64),
                    np.asarray(res.pred_tube_semvec_loss_steps, dtype=np.float64), _n(res.pred_tube), _n(res.pred_tube_mel),
                    _n(res.pred_tube_semvec), _n(res.prod_tube_mel), _n(res.prod_tube_semvec),
                    np.asarray(res.prod_tube_mel_loss_steps, dtype=np.float64)])
    for a, b in zip(*out):
        np.testing.assert_allclose(a, b, atol=2e-4, rtol=2e-4)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_tube_models_training_steps_vs_oracle(HipPlanner, golden_soma, dtype):
    """continue_learning_tube (paule/paule.py:1381-1404): Adam steps of the cp -> tube model (no half sequence) and of the
    tube -> mel model on the device (pl_train_model_step) against torch autograd + torch.optim.Adam on the CPU: step losses,
    the trained parameters, and the tube predictions the planner then makes with them."""
    from oracle import planner as op
    g = golden_soma
    eng = _soma_engine(HipPlanner, g, "acoustic_semvec", dtype=dtype)
    rng = np.random.default_rng(3)
    cps = g["cp0"].astype(np.float32)
    tubes = (g["fwd/pred_tube"] + 0.2 * rng.standard_normal(g["fwd/pred_tube"].shape)).astype(np.float32)
    mels = g["target_mel"].astype(np.float32)
    tr_u = op.OracleTrainer(op.forward_model_from_state_dict(state_dict_from(g, "cp_tube"), apply_half_sequence=False))
    tr_m = op.OracleTrainer(op.forward_model_from_state_dict(state_dict_from(g, "tube_mel")))
    sched = [[0, 1, 2], [1], [2, 0], [0, 1, 2]]
    lh, lo = [], []
    for j in sched:
        lh.append((float(eng.train_model_step("cp_tube", cps[j], tubes[j])), float(eng.train_model_step("tube_mel", tubes[j], mels[j]))))
        lo.append((float(tr_u.train_pred_step(cps[j], tubes[j])), float(tr_m.train_pred_step(tubes[j], mels[j]))))
    f32 = dtype == "f32"
    np.testing.assert_allclose(np.array(lh), np.array(lo), rtol=1e-5 if f32 else 2e-2)
    for name, tr in (("cp_tube", tr_u), ("tube_mel", tr_m)):
        got, want = eng.get_weights(name), tr.state_dict()
        for k in want:
            d = np.abs(_n(got[k]) - _n(want[k]))
            # Adam moves a parameter by at most ~lr per step: f32 has to agree far inside that, bf16 within the budget
            assert d.max() <= (2e-5 if f32 else 1.0 * 1e-3 * len(sched)), (name, k, d.max())
    if f32:
        with torch.no_grad():
            want_tube = tr_u.pred_model(torch.from_numpy(g["cp0"]))
        np.testing.assert_allclose(_n(eng.get_tube_pred()[0]), _n(want_tube), atol=5e-5, rtol=0)
    with pytest.raises(ValueError, match="model_id"):
        eng._call(eng.lib.pl_train_model_step, 5, 1, 4, 1, 1, 0.001, 0.9, 0.999, 1e-8, None)


def test_paule_continue_learning_tube_hip(golden_soma):
    """Paule(..., use_somatosensory_feedback=True).plan_resynth(continue_learning_tube=True) on the device: epoch losses of all three
    trained models, the tube optimisers' state outliving the plan (their step counts add up over two calls), models written back."""
    from paule_amd import paule as pp
    g = golden_soma
    sdf = state_dict_from
    rng = np.random.default_rng(1)
    model = pp.Paule(pred_model=sdf(g, "pred"), embedder=sdf(g, "emb"), use_somatosensory_feedback=True, cp_tube_model=sdf(g, "cp_tube"),
                     tube_mel_model=sdf(g, "tube_mel"), tube_embedder=sdf(g, "tube_emb"),
                     tube_extractor=lambda cp: 0.3 * rng.standard_normal(cp.shape[:2] + (10,)),
                     synthesizer=lambda cp: (np.zeros(100), 44100), mel_extractor=lambda sig, sr: np.full((20, 60), 0.25),
                     device=torch.device("cuda"))
    before = {k: v.clone() for k, v in model.tube_mel_model.items()}
    steps = []
    for _ in range(2):
        res = model.plan_resynth(target_acoustic=g["target_mel"][:2], target_semvec=g["target_semvec"][:2], initial_cp=g["cp0"][:2],
                                 initialize_from=None, objective="acoustic_semvec", n_outer=1, n_inner=4, log_ii=2, continue_learning=True,
                                 continue_learning_tube=True, n_batches=2, batch_size=2, n_epochs=2, seed=3, verbose=False)
        assert len(res.tube_model_loss) == 2 and len(res.tube_mel_model_loss) == 2 and np.isfinite(res.tube_model_loss).all()
        steps.append(float(model.tube_optimizer.state_dict()["state"][0]["step"]))
    assert steps[0] > 0 and steps[1] == 2 * steps[0]
    assert any(not torch.equal(torch.as_tensor(model.tube_mel_model[k]).cpu(), before[k].cpu()) for k in before)


def test_paule_initialize_from_acoustic_hip(golden_inverse):
    """Paule.plan_resynth(initialize_from='acoustic') with the inverse model on the device: initial_cp equals the reference's
    clipped inverse output, and the plan starts from it."""
    from paule_amd import paule as pp
    small = synthetic.make_workload(2, 40, None, pred=dict(num_lstm_layers=1, hidden_size=12),
                                    emb=dict(num_lstm_layers=1, hidden_size=10))
    model = pp.Paule(pred_model=small.pred_sd, embedder=small.emb_sd, inv_model=state_dict_from(golden_inverse, "inv"),
                     device=torch.device("cuda"))
    res = model.plan_resynth(target_acoustic=golden_inverse["mel"][:2], target_semvec=small.target_semvec.numpy(),
                             initialize_from="acoustic", objective="acoustic_semvec", n_outer=1, n_inner=3, log_ii=3,
                             continue_learning=False, log_cps=True, verbose=False)
    np.testing.assert_allclose(res.initial_cp, golden_inverse["cp_clipped"][:2], atol=5e-5, rtol=0)
    assert res.planned_cp.shape == (2, 40, 30) and np.isfinite(res.planned_cp).all()


def test_two_handles_on_two_streams_do_not_starve_each_other(HipPlanner):
    """Persistent sweeps of different handles must not run concurrently (each needs its groups' workgroups co-resident): two
    engines on two streams, driven from two threads at B = 256 (184 workgroups each), finish without a bounded-wait timeout
    and give the result of a solo run (the library chains sweep launches per device)."""
    import threading
    wl = synthetic.make_workload(256, 40, "A")
    def make(stream):
        with torch.cuda.stream(stream):
            e = HipPlanner(wl.pred_sd, wl.emb_sd, batch=256, n_frames=40, objective="acoustic_semvec", dtype="bf16")
            e.set_targets(wl.target_mel, wl.target_semvec)
            e.set_cp(wl.cp0)
        return e
    solo = make(torch.cuda.current_stream())
    solo.step(12, return_loss=False)
    want = solo.get_cp()
    solo.synchronize()
    want = _n(want)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    engines = [make(st) for st in streams]
    torch.cuda.synchronize()
    errors = []
    def drive(e):
        try:
            for _ in range(4):
                e.step(3, return_loss=False)
            e.synchronize()
        except Exception as ex:   # noqa: BLE001
            errors.append(ex)
    threads = [threading.Thread(target=drive, args=(e,)) for e in engines]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for e in engines:
        got = e.get_cp()          # enqueued on the engine's own stream
        e.synchronize()
        np.testing.assert_array_equal(_n(got), want)


def test_long_sequences_f32_vs_oracle(HipPlanner):
    """T = 2000 CP frames (cfg5's length; T' = 1000 embedder steps): flag / stash indexing over long sweeps, f32 against the
    oracle on small stacked models (2 x 64 predictor, 1 x 96 embedder), 2 iterations.  Oracle side: tests/golden/oracle_long_small.npz
    (tests/oracle_golden.py)."""
    wl = og.workload("long_small")
    ref = og.get("long_small")
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=2, n_frames=2000, objective="acoustic_semvec")
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    lh = _n(eng.step(2))
    eng.synchronize()
    np.testing.assert_allclose(lh, ref["loss"], rtol=LOSS_RTOL_F32, atol=1e-7)
    np.testing.assert_allclose(_n(eng.get_cp()), ref["cp_after"][1], atol=CP_ATOL_F32, rtol=0)


def test_long_sequences_bf16_sweeps(HipPlanner):
    """The same length on Paule's default models in bf16 (cfg5: 16 utterances x 2000 frames; 16-row sweeps of 2000 / 1000 steps):
    finite, falling loss, and bit-identical on replay."""
    wl = synthetic.make_workload(16, 2000, "A")
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=16, n_frames=2000, objective="acoustic_semvec", dtype="bf16")
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    loss = _n(eng.step(4))
    eng.synchronize()
    assert np.isfinite(loss).all() and (loss[-1, :, 0] < loss[0, :, 0]).all()
    eng.set_cp(wl.cp0)
    eng.reset_optimizer()
    np.testing.assert_array_equal(_n(eng.step(4)), loss)
    eng.synchronize()


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_long_sequences_set_a_vs_oracle(HipPlanner, dtype):
    """BASELINE configs[4]'s shape on Paule's default models (set A, H = 720, T = 2000, T' = 1000) against the float64 oracle on
    the same two utterances, one iteration (losses, the model gradient through the 2000 / 1000-step recurrences, the updated CP): the
    long-form path of cfg5 (one 16-row group; bf16: the 16-row fused launches, f32: pipelined sweeps) compared with the reference
    arithmetic, not only checked for its properties.  f32: the f32 bars; bf16: the bf16 bars.  The oracle's iteration (~2 minutes of float64 torch)
    is the committed fixture tests/golden/oracle_long_set_a.npz."""
    B, T = 2, 2000
    wl = og.workload("long_set_a")
    ref = og.get("long_set_a")
    lo, cpo, g_model = ref["loss"], ref["cp_after"][0], ref["g_model"]
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype=dtype)
    if dtype == "bf16":   # round 3: one group of 16 rows runs the role-fused launches on 16-row tiles (lstm_fused16.h), 2000 / 1000 steps per role
        assert eng.plan_info()["fused_rows"] == 16, eng.plan_info()
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    l1 = _n(eng.step(1))
    eng.synchronize()
    dX = _n(eng.debug_read("dX")).reshape(T, 16, 32)[:, :B, :30].transpose(1, 0, 2)
    g_err = np.linalg.norm(dX - g_model) / np.linalg.norm(g_model)   # through 2000 + 1000 + 1000 recurrent steps each way
    assert g_err <= (1e-4 if dtype == "f32" else 2e-2), g_err
    lh = l1
    dcp = np.abs(_n(eng.get_cp()) - cpo)
    if dtype == "f32":
        np.testing.assert_allclose(lh, lo, rtol=LOSS_RTOL_F32, atol=1e-7)
        assert dcp.max() <= CP_ATOL_F32
    else:
        np.testing.assert_allclose(lh, lo, rtol=LOSS_RTOL_BF16, atol=5e-3)
        assert dcp.max() <= 0.5 * 0.01 * 2 and dcp.mean() <= 1e-4, (dcp.max(), dcp.mean())   # one Adam step of lr = 0.01


def test_minimum_length_and_single_utterance(HipPlanner):
    """B = 1 (the reference's own batch) at the shortest plannable length T = 14 (jerk needs T - 12 >= 1 frames), both dtypes'
    sweeps with a single group of 8 / 16 rows."""
    wl = synthetic.make_workload(1, 14, "A")
    orc = op.OraclePlanner(op.forward_model_from_state_dict(wl.pred_sd), op.embedding_model_from_state_dict(wl.emb_sd),
                           objective="acoustic_semvec")
    orc.set_targets(wl.target_mel, wl.target_semvec)
    orc.set_cp(wl.cp0)
    lo, co = _n(orc.step(5)), _n(orc.get_cp())
    for dtype in ("f32", "bf16"):
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=1, n_frames=14, objective="acoustic_semvec", dtype=dtype)
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        lh = _n(eng.step(5))
        eng.synchronize()
        if dtype == "f32":
            np.testing.assert_allclose(lh, lo, rtol=LOSS_RTOL_F32, atol=1e-7)
            np.testing.assert_allclose(_n(eng.get_cp()), co, atol=CP_ATOL_F32, rtol=0)
        else:
            np.testing.assert_allclose(lh, lo, rtol=LOSS_RTOL_BF16, atol=5e-3)
            assert np.abs(_n(eng.get_cp()) - co).max() <= 0.5 * 0.01 * 5


def test_bf16_long_run_tracks_f32(HipPlanner):
    """A whole plan's worth of inner iterations (the reference's default: 5 outer x 24 inner = 120, paule/paule.py:399-400) on
    Paule's default models: the bf16 path must not drift away from the f32 path -- loss curve within 2 % over all 120
    iterations (total loss falls by orders of magnitude meanwhile), final CP within 20 % of the distance travelled."""
    wl = synthetic.make_workload(6, 60, "A")
    runs = {}
    for dtype in ("f32", "bf16"):
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=6, n_frames=60, objective="acoustic_semvec", dtype=dtype)
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        loss = _n(eng.step(120))
        eng.synchronize()
        runs[dtype] = (loss[:, :, 0], _n(eng.get_cp()))
    lf, cf = runs["f32"]
    lb, cb = runs["bf16"]
    assert np.isfinite(lb).all()
    np.testing.assert_allclose(lb, lf, rtol=2e-2)
    assert lf[-1].mean() < 0.5 * lf[0].mean()
    travelled = np.abs(cf - _n(wl.cp0)).mean()
    assert np.abs(cb - cf).mean() <= 0.2 * travelled, (np.abs(cb - cf).mean(), travelled)


@pytest.mark.parametrize("shape", [dict(B=3, T=40, set=None), dict(B=20, T=30, set="A"), dict(B=100, T=24, set="B")])
def test_bf16_sweep16_equals_sweep32(HipPlanner, golden_small, shape, monkeypatch):
    """Batches of up to 128 rows use groups of 16 rows on the 16x16x32 MFMA (lstm_persist16.hip); same arithmetic as the
    32-row kernels up to the f32 summation order of the MFMA k-steps and of the partial dh tiles: loss curves agree to
    2e-3, CP to 2e-3 (bf16 noise through Adam), and both sit inside the bf16 bars against the oracle elsewhere."""
    if shape["set"] is None:
        g = golden_small
        mk = lambda: _engine(HipPlanner, g, "acoustic_semvec", dtype="bf16")
    else:
        wl = synthetic.make_workload(shape["B"], shape["T"], shape["set"])
        def mk():
            e = HipPlanner(wl.pred_sd, wl.emb_sd, batch=shape["B"], n_frames=shape["T"], objective="acoustic_semvec", dtype="bf16")
            e.set_targets(wl.target_mel, wl.target_semvec)
            e.set_cp(wl.cp0)
            return e
    outs = []
    monkeypatch.setenv("PAULE_HIP_FUSED", "0")   # this test compares the per-layer kernels (B = 100 would take the fused launches)
    for s16 in ("1", "0"):
        monkeypatch.setenv("PAULE_HIP_SWEEP16", s16)
        eng = mk()
        loss = _n(eng.step(5))
        eng.synchronize()
        outs.append((loss, _n(eng.get_cp())))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=2e-3, atol=2e-4)
    d = np.abs(outs[0][1] - outs[1][1])
    assert d.max() <= 0.5 * 0.01 * 5 and d.mean() <= 1e-4, (d.max(), d.mean())


@pytest.mark.parametrize("shape", [dict(B=2, T=64, set="A", dtype="f32", graph=True), dict(B=20, T=50, set="A", dtype="f32", graph=False),
                                   dict(B=5, T=61, set="B", dtype="bf16", graph=True), dict(B=40, T=36, set="B", dtype="bf16", graph=False),
                                   dict(B=140, T=24, set="B", dtype="bf16", graph=True), dict(B=3, T=37, set="B", dtype="f32", graph=True)])
def test_layer_wavefront_is_bit_identical(HipPlanner, shape, monkeypatch):
    """Stacked models whose sweeps leave CUs free run their layers as a wavefront over time chunks (planner.hip,
    model_forward_wavefront): the same kernels on step ranges, the f32 cell state handed over through a carry buffer, the
    projections between the layers per chunk.  Nothing about the arithmetic changes: losses and the planned CP are
    bit-identical to the layer-after-layer schedule, for any number of chunks (uneven chunk lengths included), with and
    without graph capture."""
    wl = synthetic.make_workload(shape["B"], shape["T"], shape["set"])
    monkeypatch.setenv("PAULE_HIP_FUSED", "0")   # (from 49 rows the stacked predictor's forward pass is a fused launch by default: round 3)
    outs = []
    for chunks in ("0", "4", "7"):
        monkeypatch.setenv("PAULE_HIP_WAVEFRONT", chunks)
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=shape["B"], n_frames=shape["T"], objective="acoustic_semvec", dtype=shape["dtype"],
                         use_graph=shape["graph"])
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        loss = _n(eng.step(4))
        eng.synchronize()
        outs.append((loss, _n(eng.get_cp())) + tuple(_n(x) for x in eng.get_pred()))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert np.array_equal(a, b)


def _variant_engine(HipPlanner, g, variant, objective, dtype="f32", **extra):
    eng = HipPlanner(state_dict_from(g, "pred"), state_dict_from(g, f"{variant}/emb"), batch=int(g["B"]), n_frames=int(g["T"]),
                     objective=objective, dtype=dtype, **extra)
    eng.set_targets(g["target_mel"], g[f"{variant}/target_semvec"])
    eng.set_cp(g["cp0"])
    return eng


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("objective", ["acoustic_semvec", "semvec"])
@pytest.mark.parametrize("variant", ["melsmooth", "upsampling"])
def test_embedder_variants_f32_vs_reference_fixture(HipPlanner, golden_embvar, variant, objective, use_graph):
    """MelEmbeddingModelMelSmoothResidualUpsampling (residual mel blocks -> LSTM -> post_linear -> LeakyReLU ->
    upsampling, paule/models.py:362-409) and EmbeddingModel(post_upsampling_size > 0) (:432-446) as the embedder of the planning
    loop: embedding of a mel (ragged lens), gradients, CP and losses of the reference loop run through the reference's classes,
    same bars as the default embedder."""
    g = golden_embvar
    eng = _variant_engine(HipPlanner, g, variant, objective, use_graph=use_graph)
    sem = eng.embed_mel(g["target_mel"], lens=g[f"{variant}/embed_lens"])
    np.testing.assert_allclose(_n(sem), g[f"{variant}/embed_semvec_lens"], atol=FWD_ATOL_F32, rtol=0)
    np.testing.assert_allclose(_n(eng.embed_mel(g["target_mel"])), g[f"{variant}/embed_semvec_full"], atol=FWD_ATOL_F32, rtol=0)
    logs, done = [], 0
    for k in (1, 5, 20):
        loss, grad = eng.step(k - done, return_grad=True)
        logs.append(_n(loss))
        done = k
        np.testing.assert_allclose(_n(eng.get_cp()), g[f"{variant}/{objective}/cp_after_{k}"], atol=CP_ATOL_F32, rtol=0,
                                   err_msg=f"cp after {k}")
        ref_g = g[f"{variant}/{objective}/grad_at_{k}"]
        np.testing.assert_allclose(_n(grad), ref_g, atol=1e-5 * max(1.0, np.abs(ref_g).max()), rtol=0, err_msg=f"grad at {k}")
    np.testing.assert_allclose(np.concatenate(logs), g[f"{variant}/{objective}/loss_log"], rtol=LOSS_RTOL_F32, atol=1e-7)
    np.testing.assert_allclose(_n(eng.get_pred()[1]), g[f"{variant}/{objective}/final_pred_semvec"], atol=FWD_ATOL_F32, rtol=0)


@pytest.mark.parametrize("variant", ["melsmooth", "upsampling"])
def test_embedder_variants_bf16_against_oracle(HipPlanner, golden_embvar, variant):
    g = golden_embvar
    name = f"{variant}/acoustic_semvec"
    eng = _variant_engine(HipPlanner, g, variant, "acoustic_semvec", dtype="bf16")
    loss, grad = eng.step(1, return_grad=True)
    assert _cos(grad, g[f"{name}/grad_at_1"]) >= COS_BF16
    more = eng.step(19)
    np.testing.assert_allclose(np.concatenate([_n(loss), _n(more)]), g[f"{name}/loss_log"], rtol=LOSS_RTOL_BF16, atol=1e-4)
    np.testing.assert_allclose(_n(eng.get_cp()), g[f"{name}/cp_after_20"], atol=0.05 * 0.01 * 20, rtol=0)


def test_embedder_variant_full_size_vs_oracle_rows(HipPlanner):
    """The class-default MelEmbeddingModelMelSmoothResidualUpsampling (3 mel blocks, LSTM 4 x 180, head 8192) as the embedder
    of a B = 40 x 60-frame plan, f32: five iterations against the CPU oracle on three of the utterances."""
    from oracle import planner as op
    from paule_amd import models
    torch.manual_seed(5)
    emb = models.MelEmbeddingModelMelSmoothResidualUpsampling()
    wl = synthetic.make_workload(40, 60, "A")
    emb_sd = {k: v.detach().clone() for k, v in emb.state_dict().items()}
    with torch.no_grad():
        for k in emb_sd:
            if k.startswith("MelBlocks."):
                emb_sd[k] = emb_sd[k] * 1.5
    ora_e = op.embedding_model_from_state_dict(emb_sd, dtype=torch.float32)
    with torch.no_grad():
        target_sem = ora_e(wl.target_mel.float(), [torch.tensor(30)] * 40) + 0.02
    eng = HipPlanner(wl.pred_sd, emb_sd, batch=40, n_frames=60, objective="acoustic_semvec")
    eng.set_targets(wl.target_mel, target_sem)
    eng.set_cp(wl.cp0)
    loss = _n(eng.step(5))
    rows = [0, 17, 39]
    P = op.OraclePlanner(op.forward_model_from_state_dict(wl.pred_sd, dtype=torch.float32), ora_e, objective="acoustic_semvec",
                         dtype=torch.float32)
    P.set_targets(wl.target_mel[rows], target_sem[rows])
    P.set_cp(wl.cp0[rows])
    want = P.step(5).numpy()
    np.testing.assert_allclose(loss[:, rows, :6], want[:, :, :6], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(_n(eng.get_cp())[rows], _n(P.get_cp()), atol=2e-4, rtol=0)


def test_embedder_variant_bf16_model_gradient_vs_oracle(HipPlanner):
    """The class-default MelEmbeddingModelMelSmoothResidualUpsampling embedder (mel blocks, LSTM 4 x 180 = the narrow width class,
    head 8192) in bf16: the model part of dL/dCP of one iteration against the torch float64 oracle, relative error <= 2 %."""
    from oracle import manual as mo
    from paule_amd import models
    torch.manual_seed(5)
    emb = models.MelEmbeddingModelMelSmoothResidualUpsampling()
    B, T = 12, 60
    wl = synthetic.make_workload(B, T, "A")
    emb_sd = {k: v.detach().clone() for k, v in emb.state_dict().items()}
    ora_e = op.embedding_model_from_state_dict(emb_sd)
    with torch.no_grad():
        target_sem = ora_e(wl.target_mel.double(), [torch.tensor(T // 2)] * B) + 0.02
    orc = op.OraclePlanner(op.forward_model_from_state_dict(wl.pred_sd), ora_e, objective="acoustic_semvec")
    orc.set_targets(wl.target_mel, target_sem)
    orc.set_cp(wl.cp0)
    orc.step(1)
    g_model = _n(orc.last_grad) - mo.smoothness_loss_grad(_n(wl.cp0))[3]
    eng = HipPlanner(wl.pred_sd, emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
    eng.set_targets(wl.target_mel, target_sem)
    eng.set_cp(wl.cp0)
    eng.step(1, return_loss=False)
    eng.synchronize()
    got = _n(eng.debug_read("dX")).reshape(T, 16, 32)[:, :B, :30].transpose(1, 0, 2)
    err = np.linalg.norm(got - g_model) / np.linalg.norm(g_model)
    assert err <= 2e-2, err


def _soma_engine(HipPlanner, g, objective, dtype="f32", **extra):
    eng = HipPlanner(state_dict_from(g, "pred"), state_dict_from(g, "emb"), batch=int(g["B"]), n_frames=int(g["T"]), objective=objective,
                     dtype=dtype, tube_models=(state_dict_from(g, "cp_tube"), state_dict_from(g, "tube_mel"), state_dict_from(g, "tube_emb")),
                     **extra)
    eng.set_targets(g["target_mel"], g["target_semvec"])
    eng.set_cp(g["cp0"])
    return eng


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("objective", ["acoustic_semvec", "semvec"])
def test_somatosensory_feedback_f32_vs_reference_fixture(HipPlanner, golden_soma, objective, use_graph):
    """the somatosensory path of the loop (cp -> tube -> mel and tube -> semantic vector, two more loss terms,
    paule/paule.py:916-929, :624-644) on the device against the reference's classes run through the reference loop: tube
    predictions, gradients, CP and all eight loss columns, same bars as the acoustic path."""
    g = golden_soma
    eng = _soma_engine(HipPlanner, g, objective, use_graph=use_graph)
    tube, tmel, tsem = eng.get_tube_pred()
    np.testing.assert_allclose(_n(tube), g["fwd/pred_tube"], atol=FWD_ATOL_F32, rtol=0)
    np.testing.assert_allclose(_n(tmel), g["fwd/pred_tube_mel"], atol=FWD_ATOL_F32, rtol=0)
    np.testing.assert_allclose(_n(tsem), g["fwd/pred_tube_semvec"], atol=FWD_ATOL_F32, rtol=0)
    logs, done = [], 0
    for k in (1, 5, 20):
        loss, grad = eng.step(k - done, return_grad=True)
        logs.append(_n(loss))
        done = k
        np.testing.assert_allclose(_n(eng.get_cp()), g[f"{objective}/cp_after_{k}"], atol=CP_ATOL_F32, rtol=0, err_msg=f"cp after {k}")
        ref_g = g[f"{objective}/grad_at_{k}"]
        np.testing.assert_allclose(_n(grad), ref_g, atol=1e-5 * max(1.0, np.abs(ref_g).max()), rtol=0, err_msg=f"grad at {k}")
    np.testing.assert_allclose(np.concatenate(logs), g[f"{objective}/loss_log"], rtol=LOSS_RTOL_F32, atol=1e-7)


def test_somatosensory_feedback_bf16_and_errors(HipPlanner, golden_soma):
    g = golden_soma
    name = "acoustic_semvec"
    eng = _soma_engine(HipPlanner, g, name, dtype="bf16")
    loss, grad = eng.step(1, return_grad=True)
    assert _cos(grad, g[f"{name}/grad_at_1"]) >= COS_BF16
    more = eng.step(19)
    np.testing.assert_allclose(np.concatenate([_n(loss), _n(more)]), g[f"{name}/loss_log"], rtol=LOSS_RTOL_BF16, atol=1e-4)
    np.testing.assert_allclose(_n(eng.get_cp()), g[f"{name}/cp_after_20"], atol=0.05 * 0.01 * 20, rtol=0)
    with pytest.raises(ValueError, match="acoustic_semvec"):
        _soma_engine(HipPlanner, g, "acoustic")


@pytest.mark.parametrize("shape", [dict(T=46, set="A"), dict(T=61, set="B"), dict(T=40, set=None), dict(T=52, set="A", B=2),
                                   dict(T=40, set=None, B=2)])
def test_f32_one_row_kernels_vs_oracle(HipPlanner, golden_small, shape, monkeypatch):
    """B = 1 (the reference's own operating point) and B = 2 in f32 run the recurrent products as FMA chains on the rows in use
    (lstm_persist_f32.hip, NV = 1 / 2) instead of 16x16x4 MFMAs: against the CPU oracle at the f32 bars, and against the MFMA kernels
    (PAULE_HIP_F32_VALU=0) to summation-order noise; with the layer wavefront on top (set B: 4 layers)."""
    from oracle import planner as op
    B = shape.get("B", 1)
    if shape["set"] is None:
        g = golden_small
        pred_sd, emb_sd = state_dict_from(g, "pred"), state_dict_from(g, "emb")
        tm, ts, cp0 = g["target_mel"][:B], g["target_semvec"][:B], g["cp0"][:B]
    else:
        wl = synthetic.make_workload(B, shape["T"], shape["set"])
        pred_sd, emb_sd, tm, ts, cp0 = wl.pred_sd, wl.emb_sd, wl.target_mel, wl.target_semvec, wl.cp0
    T = int(np.asarray(cp0).shape[1])
    outs = []
    for valu in ("1", "0"):
        monkeypatch.setenv("PAULE_HIP_F32_VALU", valu)
        eng = HipPlanner(pred_sd, emb_sd, batch=B, n_frames=T, objective="acoustic_semvec")
        eng.set_targets(tm, ts)
        eng.set_cp(cp0)
        loss, grad = eng.step(1, return_grad=True)
        more = eng.step(9)
        outs.append((np.concatenate([_n(loss), _n(more)]), _n(grad), _n(eng.get_cp())))
    P = op.OraclePlanner(op.forward_model_from_state_dict(pred_sd, dtype=torch.float32), op.embedding_model_from_state_dict(emb_sd, dtype=torch.float32),
                         objective="acoustic_semvec", dtype=torch.float32)
    P.set_targets(np.asarray(tm), np.asarray(ts))
    P.set_cp(np.asarray(cp0))
    want = P.step(10).numpy()
    for loss, grad, cp in outs:
        np.testing.assert_allclose(loss[:, :, :6], want[:, :, :6], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(cp, _n(P.get_cp()), atol=2e-4, rtol=0)
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(outs[0][1], outs[1][1], atol=1e-5 * max(1.0, np.abs(outs[1][1]).max()), rtol=0)
    np.testing.assert_allclose(outs[0][2], outs[1][2], atol=1e-5, rtol=0)


@pytest.mark.parametrize("combo", ["soma+smiling+past", "melsmooth+classifier+past", "upsampling+smiling", "soma+melsmooth"])
def test_feature_combinations_f32_vs_oracle(HipPlanner, golden_soma, golden_embvar, golden_small, combo):
    """The widened rows combined (somatosensory feedback, embedder variants, speech classifier, smiling, past_cp): eight
    iterations on the device against the CPU oracle given the same models, f32 bars."""
    from oracle import planner as op
    gs, ge, g0 = golden_soma, golden_embvar, golden_small
    pred_sd = state_dict_from(gs, "pred")
    emb_sd = state_dict_from(ge, "melsmooth/emb") if "melsmooth" in combo else \
        state_dict_from(ge, "upsampling/emb") if "upsampling" in combo else state_dict_from(gs, "emb")
    tube = ("soma" in combo)
    tube_sds = (state_dict_from(gs, "cp_tube"), state_dict_from(gs, "tube_mel"), state_dict_from(gs, "tube_emb")) if tube else None
    smiling, past, cls = "smiling" in combo, "past" in combo, "classifier" in combo
    B, T = 3, 40
    eng = HipPlanner(pred_sd, emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", smiling=smiling, tube_models=tube_sds)
    ora_tube = None
    if tube:
        ora_tube = (op.forward_model_from_state_dict(tube_sds[0], apply_half_sequence=False), op.forward_model_from_state_dict(tube_sds[1]),
                    op.embedding_model_from_state_dict(tube_sds[2]))
    P = op.OraclePlanner(op.forward_model_from_state_dict(pred_sd), op.embedding_model_from_state_dict(emb_sd), objective="acoustic_semvec",
                         smiling=smiling, tube_models=ora_tube)
    for pl in (eng, P):
        pl.set_targets(gs["target_mel"], gs["target_semvec"])
        pl.set_cp(gs["cp0"])
        if past:
            pl.set_past_cp(g0["past_cp"])
        if cls:
            pl.set_speech_classifier(state_dict_from(g0, "clf"))
    got = _n(eng.step(8))
    want = P.step(8).numpy()
    np.testing.assert_allclose(got, want, rtol=LOSS_RTOL_F32, atol=1e-6)
    np.testing.assert_allclose(_n(eng.get_cp()), _n(P.get_cp()), atol=CP_ATOL_F32, rtol=0)


@pytest.mark.parametrize("shape", [dict(B=1, T=64, objective="acoustic_semvec", graph=True), dict(B=5, T=61, objective="semvec", graph=True),
                                   dict(B=16, T=46, objective="acoustic_semvec", graph=False), dict(B=3, T=40, objective="acoustic_semvec", graph=True, variant=True),
                                   dict(B=2, T=51, objective="acoustic_semvec", graph=True, classifier=True),
                                   dict(B=4, T=57, objective="acoustic_semvec", graph=True, set="B"), dict(B=18, T=44, objective="semvec", graph=False, set="B"),
                                   dict(B=3, T=53, objective="acoustic_semvec", graph=True, dtype="bf16"), dict(B=30, T=40, objective="acoustic_semvec", graph=True, dtype="bf16"),
                                   dict(B=9, T=47, objective="semvec", graph=False, dtype="bf16", set="B")])
def test_acoustic_pipeline_is_bit_identical(HipPlanner, golden_small, golden_embvar, shape, monkeypatch):
    """Small batches (f32; bf16 with the sweeps spread over the XCDs) run predictor -> mel head + pooling -> embedder layers as
    ONE pipeline over time chunks, forward and backward (planner.hip, acoustic_forward_pipeline / acoustic_backward_pipeline): same kernels on frame ranges, so losses,
    gradients, CP and predictions are bit-identical to the model-after-model schedule (odd T, semvec objective, speech
    classifier, an embedder with a post_linear head, eager and graph)."""
    B, T = shape["B"], shape["T"]
    wl = synthetic.make_workload(B, T, shape.get("set", "A"))   # set B: a stacked predictor (4 x 180) in front of the embedder
    emb_sd = state_dict_from(golden_embvar, "upsampling/emb") if shape.get("variant") else wl.emb_sd
    monkeypatch.setenv("PAULE_HIP_FUSED16", "0")   # bf16 batches of up to 16 rows would take the 16-row fused launches: the pipelines are the subject here
    outs = []
    for pipe in ("1", "0"):
        monkeypatch.setenv("PAULE_HIP_WF_PIPELINE", pipe)
        eng = HipPlanner(wl.pred_sd, emb_sd, batch=B, n_frames=T, objective=shape["objective"], use_graph=shape["graph"],
                         dtype=shape.get("dtype", "f32"))
        eng.set_targets(wl.target_mel, wl.target_semvec)
        eng.set_cp(wl.cp0)
        if shape.get("classifier"):
            eng.set_speech_classifier(state_dict_from(golden_small, "clf"))
        loss, grad = eng.step(1, return_grad=True)
        more = eng.step(4)
        outs.append((_n(loss), _n(grad), _n(more), _n(eng.get_cp())) + tuple(_n(x) for x in eng.get_pred()))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_somatosensory_pipeline_is_bit_identical(HipPlanner, golden_soma, dtype, monkeypatch):
    """The somatosensory path as a pipeline over time chunks (tube_forward_pipeline / tube_backward_pipeline; with it the acoustic
    pipeline) against model after model: losses (all eight columns), gradients, CP and the tube predictions are bit-identical."""
    g = golden_soma
    outs = []
    for pipe in ("1", "0"):
        monkeypatch.setenv("PAULE_HIP_WF_PIPELINE", pipe)
        eng = _soma_engine(HipPlanner, g, "acoustic_semvec", dtype=dtype)
        loss, grad = eng.step(1, return_grad=True)
        more = eng.step(5)
        outs.append((_n(loss), _n(grad), _n(more), _n(eng.get_cp())) + tuple(_n(x) for x in eng.get_tube_pred()))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_error_paths_through_the_c_abi(HipPlanner, golden_small, golden_train):
    """Nonzero return code -> ValueError with the library's message (the reference's convention for its one C library,
    paule/util.py:33-34): call-sequence errors (PL_ERR_STATE) and bad arguments (PL_ERR_INVALID); nothing aborts, and the
    handle stays usable afterwards."""
    g = golden_small
    sd_p, sd_e = state_dict_from(g, "pred"), state_dict_from(g, "emb")
    eng = HipPlanner(sd_p, sd_e, batch=int(g["B"]), n_frames=int(g["T"]), objective="acoustic_semvec")
    with pytest.raises(ValueError, match="pl_set_cp has not been called"):
        eng.step(1)
    eng.set_cp(g["cp0"])
    with pytest.raises(ValueError, match="pl_set_targets has not been called"):
        eng.step(1)
    eng.set_targets(g["target_mel"], None)
    with pytest.raises(ValueError, match="target_semvec"):
        eng.step(1)                                    # acoustic_semvec without a semantic target
    with pytest.raises(ValueError, match="expected shape"):
        eng.set_cp(g["cp0"][:, :-1])                   # wrong length is caught on the host
    with pytest.raises(ValueError, match="without inv_model"):
        eng.inverse_forward(g["target_mel"])
    with pytest.raises(ValueError, match="does not fit an engine"):
        eng.train_pred_step(np.zeros((17, int(g["T"]), 30), np.float32), np.zeros((17, int(g["T"]) // 2, 60), np.float32))
    with pytest.raises(ValueError, match="n_frames"):
        eng._call(eng.lib.pl_train_pred_step, 1, 1, 1, 1, 0.001, 0.9, 0.999, 1e-8, None)   # n_frames = 1 (the non-NULL dummy pointers are never touched)
    eng.set_targets(g["target_mel"], g["target_semvec"])
    eng.set_cp(g["cp0"])
    loss = _n(eng.step(20))
    np.testing.assert_allclose(loss, g["acoustic_semvec/loss_log"], rtol=LOSS_RTOL_F32, atol=1e-7)   # still a working handle
    with pytest.raises(ValueError, match="need an embedder"):
        HipPlanner(sd_p, None, batch=2, n_frames=20, objective="semvec")
    with pytest.raises(ValueError, match="n_frames must be >= 14"):
        HipPlanner(sd_p, sd_e, batch=2, n_frames=12)


def test_full_size_cfg2_f32_against_oracle_rows(HipPlanner):
    """cfg2 at full size (B = 64 x 300 frames, `acoustic`, f32, Paule's default predictive model): the first two utterances of the
    batched HIP run against a float64 oracle run on those two alone (per-utterance rule a-0: the rows of a batch are B = 1
    problems) at the f32 bar, and rows 16..31 bit-equal to a 16-utterance engine (same 16-row sweeps, no cross-row coupling)."""
    B, T = 64, 300
    wl = og.workload("cfg2_rows")
    ref = og.get("cfg2_rows")   # the oracle on rows 0 and 1: tests/golden/oracle_cfg2_rows.npz
    eng = HipPlanner(wl.pred_sd, None, batch=B, n_frames=T, objective="acoustic")
    eng.set_targets(wl.target_mel, None)
    eng.set_cp(wl.cp0)
    loss = _n(eng.step(3))
    cp = _n(eng.get_cp())
    eng.synchronize()
    np.testing.assert_allclose(loss[:, :2], ref["loss"], rtol=LOSS_RTOL_F32, atol=1e-7)
    np.testing.assert_allclose(cp[:2], ref["cp_after"][2], atol=CP_ATOL_F32, rtol=0)
    sub = HipPlanner(wl.pred_sd, None, batch=16, n_frames=T, objective="acoustic")
    sub.set_targets(wl.target_mel[16:32], None)
    sub.set_cp(wl.cp0[16:32])
    np.testing.assert_array_equal(_n(sub.step(3)), loss[:, 16:32])
    np.testing.assert_array_equal(_n(sub.get_cp()), cp[16:32])


def test_full_size_cfg3_f32_against_oracle_rows(HipPlanner):
    """cfg3 in f32 at full size (B = 256 x 300 frames, `acoustic_semvec`, Paule's default models): 16 groups of 16 rows run as four
    chains per workgroup (lstm_chain_f32.hip) in all three LSTM layers; the first and the last utterance against a float64 oracle
    run on those two alone at the f32 bars, and rows 32..47 (one whole group, the third chain of its set) bit-equal to a
    16-utterance engine -- the same MFMA order per element, no coupling between the chains of a workgroup."""
    B, T, n = 256, 300, 2
    wl = og.workload("cfg3_rows")
    ref = og.get("cfg3_rows")   # the oracle on rows 0 and 255 (three iterations; the first two are used here)
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec")
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    loss = _n(eng.step(n))
    cp = _n(eng.get_cp())
    eng.synchronize()
    rows = [0, B - 1]
    np.testing.assert_allclose(loss[:, rows], ref["loss"][:n], rtol=LOSS_RTOL_F32, atol=1e-7)
    np.testing.assert_allclose(cp[rows], ref["cp_after"][n - 1], atol=CP_ATOL_F32, rtol=0)
    sub = HipPlanner(wl.pred_sd, wl.emb_sd, batch=16, n_frames=T, objective="acoustic_semvec")
    sub.set_targets(wl.target_mel[32:48], wl.target_semvec[32:48])
    sub.set_cp(wl.cp0[32:48])
    np.testing.assert_array_equal(_n(sub.step(n)), loss[:, 32:48])
    np.testing.assert_array_equal(_n(sub.get_cp()), cp[32:48])


@pytest.mark.parametrize("B", [256, 100])
def test_full_size_cfg3_bf16_against_oracle_rows(HipPlanner, B):
    """cfg3 at full size (B = 256 x 300 frames, `acoustic_semvec`, bf16, Paule's default models): the first and the last utterance
    of the batched HIP run against a float64 oracle run on those two alone, at the bf16 bars (loss rtol 2e-2 with the 5e-3 floor of
    the small weighted terms; CP within 5 % of the lr * iterations budget on average, one lr step at worst).  B = 256 takes the
    fused forward launch and the per-layer backward sweeps, B = 100 (a ragged last group) the fused forward AND backward launches
    -- the library's defaults for those sizes."""
    T, n = 300, 3
    case = "cfg3_rows" if B == 256 else "cfg3_100_rows"
    wl = og.workload(case)
    ref = og.get(case)   # the float64 oracle on the two rows: tests/golden/oracle_<case>.npz
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
    plan = eng.plan_info()   # the schedule this test says it covers is the one the library planned
    assert plan["fused_fwd"] == 1 and plan["fused_bwd"] == (0 if B == 256 else 1) and plan["bwd_waves"] == 8, plan
    # ... down to the forward launch's form (256 rows: two workgroups per CU, lstm_fused2.hip) and the prefetcher workgroups beside the
    # backward sweeps (round 5): a declined plan must not pass this test on the older kernels
    assert plan["fwd_per_cu"] == (2 if B == 256 else 1) and (plan["bwd_prefetchers"] > 0) == (B == 256), plan
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    loss = _n(eng.step(n))
    cp = _n(eng.get_cp())
    eng.synchronize()
    rows = [0, B - 1]
    np.testing.assert_allclose(loss[:, rows], ref["loss"], rtol=LOSS_RTOL_BF16, atol=5e-3)
    d = np.abs(cp[rows] - ref["cp_after"][n - 1])
    assert d.mean() <= 0.05 * 0.01 * n and d.max() <= 0.01 * n, (d.mean(), d.max())


@pytest.mark.parametrize("H", [64, 96, 128, 192, 256, 384])
def test_bf16_model_gradient_across_hidden_sizes(HipPlanner, H):
    """dL/dCP of one bf16 iteration against the exact float64 oracle for every width class of the 16-row kernels' instantiation list
    (P = Hp / 32 = 2 ... 12 workgroups per group): relative error <= 1.5 % (0.35 ... 0.5 % measured).  The loose Adam-level bars of the
    trajectory tests did not notice a 13 ... 18 % error that the backward kernel made for Hp < 288 (its wide ingest summed the four
    waves' partial sums in an LDS array sized for the partial image, which is smaller than those sums for narrow models) -- the
    rounding-emulation oracle did; this test keeps it from coming back."""
    from oracle import manual as mo
    B, T = 8, 40
    wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
    ex = mo.ManualPlanner(wl.pred_sd, wl.emb_sd, objective="acoustic_semvec")
    ex.set_targets(wl.target_mel.numpy(), wl.target_semvec.numpy())
    ex.set_cp(wl.cp0.numpy())
    _, _, px = mo.loss_and_grad(ex.models, "acoustic_semvec", ex.x, ex.target_mel, ex.target_semvec)
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    eng.step(1, return_loss=False)
    eng.synchronize()
    dX = _n(eng.debug_read("dX")).reshape(T, 16, 32)[:, :B, :30].transpose(1, 0, 2)
    err = np.linalg.norm(dX - px["grad_model"]) / np.linalg.norm(px["grad_model"])
    assert err <= 1.5e-2, err


_GRAD_FAMILIES = [
    ("fused forward + backward launches, H = 96", dict(B=64, T=40, pred=(1, 96), emb=(2, 96)), {}),
    ("fused forward + backward launches, H = 720, ragged", dict(B=100, T=24, pred=(1, 720), emb=(2, 720)), {}),
    ("fused launches, three embedder layers", dict(B=128, T=20, pred=(1, 720), emb=(3, 720)), {}),
    ("fused forward launch, 32-row backward sweeps", dict(B=160, T=20, pred=(1, 720), emb=(2, 720)), {}),
    ("per-layer 32-row sweeps", dict(B=160, T=20, pred=(1, 720), emb=(2, 720)), {"PAULE_HIP_FUSED": "0"}),
    ("stacked predictor (set B), 16-row sweeps + wavefront", dict(B=100, T=24, pred=(4, 180), emb=(1, 720)), {"PAULE_HIP_FUSED": "0"}),
    ("stacked predictor (set B), fused forward + backward launches of two widths, ragged", dict(B=100, T=24, pred=(4, 180), emb=(1, 720), fused_fwd=1, fused_bwd=1), {}),
    ("stacked predictor (set B), B = 256, fused forward + backward launches", dict(B=256, T=20, pred=(4, 180), emb=(1, 720), fused_fwd=1, fused_bwd=1), {}),
    ("stacked predictor (set B), B = 256, fused forward launch + per-layer backward", dict(B=256, T=20, pred=(4, 180), emb=(1, 720), fused_fwd=1, fused_bwd=0), {"PAULE_HIP_FUSED_BWD_STACKED": "0"}),
    ("2 x 180 predictor + 2 x 720 embedder, fused launches of two widths", dict(B=70, T=22, pred=(2, 180), emb=(2, 720), fused_fwd=1, fused_bwd=1), {}),
    ("stacked predictor (set B), B = 256, per-layer path", dict(B=256, T=20, pred=(4, 180), emb=(1, 720)), {"PAULE_HIP_FUSED": "0"}),
    ("16-row fused launches, one utterance", dict(B=1, T=40, pred=(1, 720), emb=(2, 720), fused_rows=16), {}),
    ("16-row fused launches, H = 96, 16 rows", dict(B=16, T=31, pred=(1, 96), emb=(2, 96), fused_rows=16), {}),
    ("16-row fused launches, stacked predictor of another width (set B), ragged", dict(B=5, T=40, pred=(4, 180), emb=(1, 720), fused_rows=16), {}),
    ("16-row fused launches, 2 x 180 predictor + 3 x 720 embedder", dict(B=12, T=26, pred=(2, 180), emb=(3, 720), fused_rows=16), {}),
    ("16-row fused launches, three groups (41 rows)", dict(B=41, T=24, pred=(1, 720), emb=(2, 720), fused_rows=16), {}),
    ("16-row fused launches, two groups, stacked predictor (set B)", dict(B=27, T=30, pred=(4, 180), emb=(1, 720), fused_rows=16), {}),
    ("2 x 360 / 2 x 360", dict(B=40, T=30, pred=(2, 360), emb=(2, 360)), {}),
    ("tiny ragged model, odd T", dict(B=5, T=31, pred=(1, 48), emb=(1, 40)), {}),
    ("launch-per-step kernels", dict(B=20, T=30, pred=(1, 96), emb=(2, 96)), {"PAULE_HIP_NO_SWEEP": "1"}),
]


@pytest.mark.parametrize("case", _GRAD_FAMILIES, ids=[c[0] for c in _GRAD_FAMILIES])
def test_bf16_model_gradient_across_kernel_families(HipPlanner, case, monkeypatch):
    """dL/dCP of one bf16 iteration against the exact float64 oracle, one case per kernel family the planner can pick: relative
    error <= 2 % (0.4 ... 0.8 % measured everywhere: the bf16 rounding level).  The trajectory tests compare plans through Adam's
    sign-like update and are blind to gradient errors of this size (see test_bf16_model_gradient_across_hidden_sizes)."""
    from oracle import manual as mo
    _, c, env = case
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    B, T = c["B"], c["T"]
    wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=c["pred"][0], hidden_size=c["pred"][1]),
                                 emb=dict(num_lstm_layers=c["emb"][0], hidden_size=c["emb"][1]))
    ex = mo.ManualPlanner(wl.pred_sd, wl.emb_sd, objective="acoustic_semvec")
    ex.set_targets(wl.target_mel.numpy(), wl.target_semvec.numpy())
    ex.set_cp(wl.cp0.numpy())
    _, _, px = mo.loss_and_grad(ex.models, "acoustic_semvec", ex.x, ex.target_mel, ex.target_semvec)
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
    if "fused_fwd" in c:   # the family this case names is the one the library planned
        assert eng.plan_info()["fused_fwd"] == c["fused_fwd"], eng.plan_info()
    if "fused_rows" in c:
        assert eng.plan_info()["fused_rows"] == c["fused_rows"] and eng.plan_info()["fused_bwd"] == 1, eng.plan_info()
    if "fused_bwd" in c:
        assert eng.plan_info()["fused_bwd"] == c["fused_bwd"], eng.plan_info()
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    eng.step(1, return_loss=False)
    eng.synchronize()
    Bp = -(-B // 16) * 16
    dX = _n(eng.debug_read("dX")).reshape(T, Bp, 32)[:, :B, :30].transpose(1, 0, 2)
    err = np.linalg.norm(dX - px["grad_model"]) / np.linalg.norm(px["grad_model"])
    assert err <= 2e-2, err


@pytest.mark.parametrize("tube_hidden", [(360, 360, 720), (96, 128, 192), (64, 48, 256)])
def test_bf16_somatosensory_model_gradient_vs_oracle(HipPlanner, tube_hidden):
    """The model part of dL/dCP with somatosensory feedback (acoustic path + cp -> tube -> mel / semantic vector) in bf16 against the
    torch float64 oracle (autograd gradient minus the smoothness terms' gradient): Paule's tube-model sizes and narrow ones (the width
    classes below Hp = 288 that the trajectory-level bars cannot tell apart from a wrong gradient): relative error <= 2 %."""
    from oracle import manual as mo
    B, T = 6, 40
    wl = synthetic.make_workload(B, T, "A")
    specs = {"cp_tube": dict(num_lstm_layers=1, hidden_size=tube_hidden[0]), "tube_mel": dict(num_lstm_layers=1, hidden_size=tube_hidden[1]),
             "tube_emb": dict(num_lstm_layers=2, hidden_size=tube_hidden[2])}
    tube = synthetic.make_tube_models(specs=specs)
    ora_tube = (op.forward_model_from_state_dict(tube[0], apply_half_sequence=False), op.forward_model_from_state_dict(tube[1]),
                op.embedding_model_from_state_dict(tube[2]))
    orc = op.OraclePlanner(op.forward_model_from_state_dict(wl.pred_sd), op.embedding_model_from_state_dict(wl.emb_sd),
                           objective="acoustic_semvec", tube_models=ora_tube)
    orc.set_targets(wl.target_mel, wl.target_semvec)
    orc.set_cp(wl.cp0)
    orc.step(1)
    g_model = _n(orc.last_grad) - mo.smoothness_loss_grad(_n(wl.cp0))[3]
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", tube_models=tube)
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    eng.step(1, return_loss=False)
    eng.synchronize()
    rd = lambda name: _n(eng.debug_read(name)).reshape(T, 16, 32)[:, :B, :30].transpose(1, 0, 2)
    got = rd("dX") + rd("dX2")
    err = np.linalg.norm(got - g_model) / np.linalg.norm(g_model)
    assert err <= 2e-2, err


def _tm(a, T, Bp, Fp, B, F):
    """time-major padded device buffer [T][Bp][Fp] -> (B, T, F)"""
    return _n(a).reshape(T, Bp, Fp)[:, :B, :F].transpose(1, 0, 2)


def _gates_tm(a, T, Bp, Hp, B, H):
    """[T][Bp][4 Hp] (gate blocks of Hp) -> (B, T, 4 H)"""
    g = _n(a).reshape(T, Bp, 4, Hp)[:, :B, :, :H].transpose(1, 0, 2, 3)
    return g.reshape(B, T, 4 * H)


@pytest.mark.parametrize("shape", [dict(B=40, T=41, H=96), dict(B=3, T=24, H=720), dict(B=130, T=20, H=720)])
def test_bf16_path_equals_rounding_emulation(HipPlanner, shape, monkeypatch):
    """The bf16 path deviates from the reference's arithmetic ONLY by where it rounds (bf16 weights and stored activations, bf16
    partial tiles in the backward exchange): oracle/bf16_emul.py is the float64 oracle with exactly those roundings written in.
    Against it the device agrees far more tightly than against the exact oracle -- what is left is f32 accumulation order and the
    fast activation forms, which flip a bf16 rounding in a small fraction of the entries.  Checked stage by stage: forward stashes
    (bit-equal in >= 88 % of the entries -- 90 % in the third layer at H = 720, 99.6 % in the first at H = 96 --, never more than 4 bf16 steps apart), pooled mel, model gradient (relative error
    ~0.15 %, a third of the error against the exact oracle), and the plan after 5 iterations (mean |difference| below 1e-6 at lr = 0.01).  Shapes: 16-row
    kernels + chunk pipelines (B = 40, 3), and the cfg3 path (B = 130: fused forward launch, 32-row backward sweeps)."""
    from oracle import bf16_emul as be
    from oracle import manual as mo
    B, T, H = shape["B"], shape["T"], shape["H"]
    wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
    monkeypatch.setenv("PAULE_HIP_FUSED", "1" if B >= 129 else "0")   # the fused BACKWARD rounds its partial products elsewhere
    Bp, Hp, Tp = -(-B // 16) * 16, -(-H // 32) * 32, T // 2
    em = be.EmulPlanner(wl.pred_sd, wl.emb_sd, objective="acoustic_semvec")
    em.set_targets(wl.target_mel.numpy(), wl.target_semvec.numpy())
    em.set_cp(wl.cp0.numpy())
    ex = mo.ManualPlanner(wl.pred_sd, wl.emb_sd, objective="acoustic_semvec")
    ex.set_targets(wl.target_mel.numpy(), wl.target_semvec.numpy())
    ex.set_cp(wl.cp0.numpy())

    # forward stashes of the first iteration
    monkeypatch.setenv("PAULE_HIP_DEBUG", "stop_after_fwd")
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", use_graph=False)
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    eng.step(1, return_loss=False)
    eng.synchronize()
    _, _, parts = be.loss_and_grad(em.models, "acoustic_semvec", em.x, em.target_mel, em.target_semvec)
    ulp = lambda ref: np.maximum(np.abs(ref), 0.25) * 2.0 ** -8   # one bf16 step at the value's magnitude (floor: that of 0.25 -- small values inherit absolute differences of their inputs)
    for name, got, ref in (("pred.h0", _tm(eng.debug_read("pred.h0"), T, Bp, Hp, B, H), parts["pred_h"][0]),
                           ("emb.h0", _tm(eng.debug_read("emb.h0"), Tp, Bp, Hp, B, H), parts["emb_h"][0]),
                           ("emb.h1", _tm(eng.debug_read("emb.h1"), Tp, Bp, Hp, B, H), parts["emb_h"][1]),
                           ("pred.c0", _tm(eng.debug_read("pred.c0"), T, Bp, Hp, B, H), parts["pred_stash"][0]["c"]),
                           ("emb.G1", _gates_tm(eng.debug_read("emb.G1"), Tp, Bp, Hp, B, H),
                            np.concatenate([parts["emb_stash"][1][k] for k in "ifgo"], axis=2))):
        same = np.mean(got == ref)
        steps = np.abs(got - ref) / ulp(ref)
        print(f"[emul] {name}: bit-equal {same:.4f}, max bf16 steps {steps.max():.2f}")
        assert same >= 0.88 and steps.max() <= 4, (name, same, steps.max())
    mel = _n(eng.debug_read("mel")).reshape(B, Tp, -1)
    np.testing.assert_allclose(mel, parts["mel"], atol=2e-3 * np.abs(parts["mel"]).max(), rtol=0)
    monkeypatch.delenv("PAULE_HIP_DEBUG")

    # one full iteration: the model gradient dL/dCP; then 5 iterations: the plan
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", use_graph=False)
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    eng.step(1, return_loss=False)
    eng.synchronize()
    Cp = 32
    dX = _tm(eng.debug_read("dX"), T, Bp, Cp, B, 30)
    _, _, pe = be.loss_and_grad(em.models, "acoustic_semvec", em.x, em.target_mel, em.target_semvec)
    _, _, px = mo.loss_and_grad(ex.models, "acoustic_semvec", ex.x, ex.target_mel, ex.target_semvec)
    nrm = np.linalg.norm(px["grad_model"])
    err_emul = np.linalg.norm(dX - pe["dX"]) / nrm
    err_exact = np.linalg.norm(dX - px["grad_model"]) / nrm
    print(f"[emul] dL/dCP relative error vs emulation {err_emul:.2e}, vs exact oracle {err_exact:.2e}")
    assert err_emul <= 0.6 * err_exact and err_emul <= 3e-3, (err_emul, err_exact)
    lh = _n(eng.step(4))
    eng.synchronize()
    em.step(5)
    ex.step(5)
    cp = _n(eng.get_cp())
    d_emul, d_exact = np.abs(cp - em.get_cp()), np.abs(cp - ex.get_cp())
    print(f"[emul] CP after 5 iterations: mean |diff| vs emulation {d_emul.mean():.2e} (max {d_emul.max():.2e}), vs exact oracle {d_exact.mean():.2e}")
    assert d_emul.mean() <= 1e-6 and d_emul.max() <= 1e-3, (d_emul.mean(), d_exact.mean(), d_emul.max())   # lr = 0.01: 4e-9 / 3e-7 measured


def test_full_size_cfg3_bf16_vs_rounding_emulation(HipPlanner):
    """The headline configuration itself (cfg3: B = 256 x 300 frames, Paule's models, bf16; fused forward launch + 32-row backward
    sweeps) against the rounding emulation on its first and last utterance (rows of a batch are independent problems, a-0): the
    predictor's h stash bit-equal in >= 90 % of the entries, the model gradient dL/dCP within 3e-3 relative -- i.e. at full size,
    too, the bf16 plan differs from the reference's arithmetic by the declared roundings and nothing else."""
    B, T, H, Hp, rows = 256, 300, 720, 736, [0, 255]
    wl = og.workload("cfg3_rows")
    ref = og.get("cfg3_rows")   # oracle/bf16_emul.py on rows 0 and 255: the predictor's h stash (bf16 patterns) and dL/dCP
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
    plan = eng.plan_info()   # the headline schedule and nothing older: two-per-CU forward launch, streamed backward sweeps with their prefetchers
    assert plan["fused_fwd"] == 1 and plan["fwd_per_cu"] == 2 and plan["fused_bwd"] == 0 and plan["bwd_waves"] == 8 and plan["bwd_prefetchers"] > 0, plan
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    eng.step(1, return_loss=False)
    eng.synchronize()
    h0 = _n(eng.debug_read("pred.h0")).reshape(T, B, Hp)[:, rows, :H].transpose(1, 0, 2)
    same = np.mean(h0 == og.bf16_from_bits(ref["emul_pred_h0_bits"]))
    dX = _n(eng.debug_read("dX")).reshape(T, B, 32)[:, rows, :30].transpose(1, 0, 2)
    err = np.linalg.norm(dX - ref["emul_dX"]) / np.linalg.norm(ref["emul_dX"])
    assert same >= 0.90 and err <= 3e-3, (same, err)


def test_full_size_cfg5_128_fused_launches_vs_oracle_and_emulation(HipPlanner):
    """BASELINE configs[4] on ONE GPU (cfg5_128: all 128 utterances x 2000 frames, bf16, Paule's models): the fused forward AND
    backward launches at T = 2000 -- 2000 / 1000 chain-steps per role, four 32-row groups -- which round 2 only ever tested up to
    T = 300 (VERDICT r2 missing #2a).  First and last utterance of the batched run after one iteration against the float64 oracle
    (loss, model gradient <= 2 %, updated CP at the bf16 bars) and against the rounding emulation (gradient <= 3e-3: the long
    recurrences differ from the reference arithmetic by the declared roundings only).  Oracle side: tests/golden/oracle_cfg5_128_rows.npz."""
    B, T, rows = 128, 2000, [0, 127]
    wl = og.workload("cfg5_128_rows")
    ref = og.get("cfg5_128_rows")
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
    plan = eng.plan_info()
    assert plan["fused_fwd"] == 1 and plan["fused_bwd"] == 1, plan
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    l1 = _n(eng.step(1))
    eng.synchronize()
    dX = _n(eng.debug_read("dX")).reshape(T, B, 32)[:, rows, :30].transpose(1, 0, 2)
    g_err = np.linalg.norm(dX - ref["g_model"]) / np.linalg.norm(ref["g_model"])
    e_err = np.linalg.norm(dX - ref["emul_dX"]) / np.linalg.norm(ref["emul_dX"])
    assert g_err <= 2e-2 and e_err <= 3e-3, (g_err, e_err)
    np.testing.assert_allclose(l1[:, rows], ref["loss"], rtol=LOSS_RTOL_BF16, atol=5e-3)
    dcp = np.abs(_n(eng.get_cp())[rows] - ref["cp_after"][0])
    assert dcp.max() <= 0.5 * 0.01 * 2 and dcp.mean() <= 1e-4, (dcp.max(), dcp.mean())


def test_full_size_cfg4_one_gpu_vs_oracle_and_row_independence(HipPlanner):
    """BASELINE configs[3]'s whole batch on ONE GPU (cfg4_1gpu: 2048 utterances x 300 frames, bf16; 64 groups of 32 rows swept 8
    at a time by the same workgroups) -- round 2 measured it and tested the multi-pass loop only at B = 270 x 17 frames (VERDICT
    r2 missing #2b).  First and last utterance after two iterations against the float64 oracle (bf16 bars), and one whole group in
    the middle of the batch (rows 1024 .. 1055, swept in the fifth pass) bit-equal to the same rows planned by a 256-utterance
    engine: the passes do not leak into each other.  Oracle side: tests/golden/oracle_cfg4_rows.npz."""
    B, T, n, rows = 2048, 300, 2, [0, 2047]
    wl = og.workload("cfg4_rows")
    ref = og.get("cfg4_rows")
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16")
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    loss = _n(eng.step(n))
    cp = _n(eng.get_cp())
    eng.synchronize()
    np.testing.assert_allclose(loss[:, rows], ref["loss"], rtol=LOSS_RTOL_BF16, atol=5e-3)
    d = np.abs(cp[rows] - ref["cp_after"][n - 1])
    assert d.mean() <= 0.05 * 0.01 * n and d.max() <= 0.01 * n, (d.mean(), d.max())
    eng.close()
    # rows 896 .. 1151 as a batch of their own (the default plan at 256 rows: fused forward launch + per-layer backward sweeps, the
    # same kernels the 2048-row handle runs group by group)
    lo_, hi_ = 896, 1152
    sub = HipPlanner(wl.pred_sd, wl.emb_sd, batch=hi_ - lo_, n_frames=T, objective="acoustic_semvec", dtype="bf16")
    sub.set_targets(wl.target_mel[lo_:hi_], wl.target_semvec[lo_:hi_])
    sub.set_cp(wl.cp0[lo_:hi_])
    ls = _n(sub.step(n))
    sub.synchronize()
    np.testing.assert_array_equal(ls[:, 128:160], loss[:, 1024:1056])
    np.testing.assert_array_equal(_n(sub.get_cp())[128:160], cp[1024:1056])


def test_plan_resynth_like_the_reference_test_on_the_gpu(golden_inverse):
    """The reference's own smoke test (tests/test_paule.py:65-70, same arguments) through the HIP planner: initialisation by
    the inverse model, planning, a synthesised log step every iteration, continued learning after every outer iteration -- and
    the result equals the same host code driving the CPU oracle."""
    from test_host import _reference_smoke_model, _factory
    small = synthetic.make_workload(2, 24, None, pred=dict(num_lstm_layers=1, hidden_size=12),
                                    emb=dict(num_lstm_layers=1, hidden_size=10))
    outs = []
    for factory, dev in ((None, "cuda"), (_factory, "cpu")):
        model = _reference_smoke_model(small, golden_inverse, factory, torch.device(dev))
        res = model.plan_resynth(target_acoustic=golden_inverse["mel"][0], objective='acoustic_semvec',
                                 initialize_from='acoustic', n_outer=2, n_inner=2, n_batches=1, batch_size=2,
                                 n_epochs=2, seed=11, verbose=False)
        outs.append(res)
    hip, orc = outs
    np.testing.assert_allclose(hip.initial_cp, orc.initial_cp, atol=5e-5, rtol=0)
    np.testing.assert_allclose(hip.planned_cp, orc.planned_cp, atol=1e-4, rtol=0)
    np.testing.assert_allclose(hip.planned_loss_steps, orc.planned_loss_steps, rtol=1e-4)
    np.testing.assert_allclose(hip.pred_model_loss, orc.pred_model_loss, rtol=1e-4)
    np.testing.assert_allclose(hip.pred_semvec, orc.pred_semvec, atol=1e-4, rtol=0)


def test_pred_optimizer_state_round_trip(HipPlanner, golden_train):
    """pl_get / pl_set_pred_optimizer_state: two training steps, export of parameters + Adam state (torch.optim.Adam.state_dict()
    layout), a NEW handle loaded with them, two more steps == four steps on one handle (the export is float32: 1e-6)."""
    g = golden_train
    sd = state_dict_from(g, "pred")
    j = g["batch_2"]
    a, _ = _train_engine(HipPlanner, g, "f32")
    for _ in range(4):
        a.train_pred_step(g["cps"][j], g["prod_mel"][j])
    b, _ = _train_engine(HipPlanner, g, "f32")
    for _ in range(2):
        b.train_pred_step(g["cps"][j], g["prod_mel"][j])
    state = b.get_pred_optimizer_state()
    assert float(state["state"][0]["step"]) == 2 and len(state["state"]) == len(sd)
    c = HipPlanner(b.get_weights("pred"), None, batch=int(g["N"]), n_frames=int(g["T"]), objective="acoustic")
    c.set_pred_optimizer_state(state)
    for _ in range(2):
        c.train_pred_step(g["cps"][j], g["prod_mel"][j])
    wa, wc = a.get_weights("pred"), c.get_weights("pred")
    for name in wa:
        np.testing.assert_allclose(_n(wc[name]), _n(wa[name]), atol=1e-6, rtol=0, err_msg=name)
    c.set_pred_optimizer_state({"state": {}})
    assert c.get_pred_optimizer_state()["state"] == {}


def test_tube_optimizer_state_round_trip(HipPlanner, golden_soma):
    """pl_{get,set}_model_optimizer_state / _step for the somatosensory models: two training steps of the cp -> tube model, export of
    parameters + Adam state, a NEW handle loaded with them, two more steps == four steps on one handle."""
    g = golden_soma
    rng = np.random.default_rng(5)
    cps = g["cp0"].astype(np.float32)
    tubes = (g["fwd/pred_tube"] + 0.2 * rng.standard_normal(g["fwd/pred_tube"].shape)).astype(np.float32)
    a = _soma_engine(HipPlanner, g, "acoustic_semvec")
    for _ in range(4):
        a.train_model_step("cp_tube", cps, tubes)
    b = _soma_engine(HipPlanner, g, "acoustic_semvec")
    for _ in range(2):
        b.train_model_step("cp_tube", cps, tubes)
    state = b.get_optimizer_state("cp_tube")
    assert float(state["state"][0]["step"]) == 2 and b.get_optimizer_state("tube_mel")["state"] == {}
    c = HipPlanner(state_dict_from(g, "pred"), state_dict_from(g, "emb"), batch=int(g["B"]), n_frames=int(g["T"]), objective="acoustic_semvec",
                   tube_models=(b.get_weights("cp_tube"), state_dict_from(g, "tube_mel"), state_dict_from(g, "tube_emb")))
    c.set_optimizer_state("cp_tube", state)
    for _ in range(2):
        c.train_model_step("cp_tube", cps, tubes)
    wa, wc = a.get_weights("cp_tube"), c.get_weights("cp_tube")
    for name in wa:
        np.testing.assert_allclose(_n(wc[name]), _n(wa[name]), atol=1e-6, rtol=0, err_msg=name)
    c.set_optimizer_state("cp_tube", {"state": {}})
    assert c.get_optimizer_state("cp_tube")["state"] == {}
    with pytest.raises(ValueError):
        c.get_optimizer_state("tube_embedder")


def test_random_shapes_f32_vs_oracle(HipPlanner):
    """Property test (hypothesis, derandomised): batch, length, layer counts, hidden sizes (padded to 32 inside the library), objective
    and projection flags drawn at random -- the f32 HIP path against the float64 oracle at the f32 bar, 3 iterations each."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    @settings(max_examples=10, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
    @given(B=st.integers(1, 40), T=st.integers(14, 70), Lp=st.integers(1, 3), Hp=st.sampled_from([5, 24, 33, 64, 100]),
           Le=st.integers(1, 2), He=st.sampled_from([7, 32, 48]), objective=st.sampled_from(["acoustic", "acoustic_semvec", "semvec"]),
           smiling=st.booleans(), with_past=st.booleans())
    def check(B, T, Lp, Hp, Le, He, objective, smiling, with_past):
        wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=Lp, hidden_size=Hp), emb=dict(num_lstm_layers=Le, hidden_size=He))
        orc = op.OraclePlanner(op.forward_model_from_state_dict(wl.pred_sd), op.embedding_model_from_state_dict(wl.emb_sd),
                               objective=objective, smiling=smiling)
        eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective=objective, smiling=smiling)
        past = (wl.cp0[0, :4] * 0.5).numpy() if with_past else None
        for pl in (orc, eng):
            pl.set_targets(wl.target_mel, wl.target_semvec)
            pl.set_cp(wl.cp0)
            if past is not None:
                pl.set_past_cp(past)
        lo, lh = _n(orc.step(3)), _n(eng.step(3))
        eng.synchronize()
        np.testing.assert_allclose(lh, lo, rtol=LOSS_RTOL_F32, atol=1e-7)
        np.testing.assert_allclose(_n(eng.get_cp()), _n(orc.get_cp()), atol=CP_ATOL_F32, rtol=0)
        eng.close()

    check()


def test_bench_two_ranks_on_one_gpu():
    """bench.py --gpus 2 started without a launcher: it spawns its two ranks itself; here they share the one GPU of the box (gloo
    collectives on host copies; the RCCL path needs a GPU per rank), plan the small `rehearsal` workload and rank 0 prints the line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "rehearsal", "--dist-backend", "gloo",
                        "--device-index", "0", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["finite"] and out["scaling"] == "weak" and out["final_cp_all_gather_ms"] > 0
    assert out["config"]["global_batch"] == 2 * out["config"]["batch_per_gpu"]


def test_fused_launch_residency_census(HipPlanner, monkeypatch):
    """A fused launch whose workgroups are not all resident gives up within the census bound and pl_synchronize says so
    (PL_ERR_STATE -> ValueError), instead of spinning in its waits: here one expected workgroup never signs in (test hook)."""
    B, T, H = 40, 30, 96
    wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=1, hidden_size=H), emb=dict(num_lstm_layers=2, hidden_size=H))
    monkeypatch.setenv("PAULE_HIP_FUSED", "1")
    monkeypatch.setenv("PAULE_HIP_FUSED_MIN_B", "1")
    monkeypatch.setenv("PAULE_HIP_DEBUG", "census_ms=20,census_expect_extra=1")
    eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=T, objective="acoustic_semvec", dtype="bf16", use_graph=False)
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    eng.step(1, return_loss=False)
    with pytest.raises(ValueError, match="not all resident"):
        eng.synchronize()
    monkeypatch.setenv("PAULE_HIP_DEBUG", "census_ms=20")
    eng.set_cp(wl.cp0)            # the handle is usable again once the GPU is its own
    eng.reset_optimizer()
    loss = _n(eng.step(2))
    eng.synchronize()
    assert np.isfinite(loss).all()
