"""The CPU oracle against the fixtures generated from the REFERENCE's own code (tests/golden/make_golden.py).
Bar: float64 vs float64, <= 1e-12 absolute (relative for the large loss values)."""
import os

import numpy as np
import pytest
import torch

from conftest import state_dict_from
from oracle import manual as om
from oracle import planner as op
from paule_amd import synthetic

TOL = 1e-12
CASES = {"acoustic": dict(objective="acoustic"), "acoustic_semvec": dict(objective="acoustic_semvec"),
         "semvec": dict(objective="semvec"), "smiling": dict(objective="acoustic_semvec", smiling=True),
         "past_cp": dict(objective="acoustic"), "classifier": dict(objective="acoustic_semvec")}


def _close(a, b, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    err = np.abs(a - b).max()
    assert err <= tol * max(1.0, np.abs(b).max()), err


def _planners(g, name):
    pred_sd, emb_sd = state_dict_from(g, "pred"), state_dict_from(g, "emb")
    kw = dict(CASES[name])
    past = g["past_cp"] if name == "past_cp" else None
    P = op.OraclePlanner(op.forward_model_from_state_dict(pred_sd), op.embedding_model_from_state_dict(emb_sd), **kw)
    M = om.ManualPlanner(pred_sd, emb_sd, **kw)
    for pl in (P, M):
        pl.set_targets(g["target_mel"], g["target_semvec"])
        pl.set_cp(g["cp0"])
        pl.set_past_cp(past)
        if name == "classifier":
            pl.set_speech_classifier(state_dict_from(g, "clf"))
    return P, M


def test_forward_outputs(golden_small):
    g = golden_small
    pm = op.forward_model_from_state_dict(state_dict_from(g, "pred"))
    em = op.embedding_model_from_state_dict(state_dict_from(g, "emb"))
    with torch.no_grad():
        cp0 = torch.from_numpy(g["cp0"])
        mel = pm(cp0)
        _close(mel, g["fwd/pred_mel"])
        _close(em(mel, [torch.tensor(mel.shape[1])] * mel.shape[0]), g["fwd/pred_semvec"])
        _close(pm(cp0[:, :39]), g["fwd/pred_mel_oddT"])      # odd T: last frame dropped
        lens = [torch.tensor(int(l)) for l in g["fwd/embed_lens"]]
        _close(em(torch.from_numpy(g["target_mel"]), lens), g["fwd/embed_semvec_lens"])
    mm = om.ManualModels(state_dict_from(g, "pred"), state_dict_from(g, "emb"))
    mel_m, _ = mm.pred_forward(g["cp0"])
    _close(mel_m, g["fwd/pred_mel"])
    _close(mm.emb_forward(mel_m)[0], g["fwd/pred_semvec"])


@pytest.mark.parametrize("name", sorted(CASES))
def test_trajectories_small(golden_small, name):
    g = golden_small
    P, M = _planners(g, name)
    logs_p, logs_m, done = [], [], 0
    for k in (1, 5, 20):
        logs_p.append(P.step(k - done).numpy())
        logs_m.append(M.step(k - done))
        done = k
        _close(P.get_cp(), g[f"{name}/cp_after_{k}"])
        _close(M.get_cp(), g[f"{name}/cp_after_{k}"])
        _close(P.last_grad, g[f"{name}/grad_at_{k}"], 1e-11)
        _close(M.last_grad, g[f"{name}/grad_at_{k}"], 1e-11)
    _close(np.concatenate(logs_p), g[f"{name}/loss_log"])
    _close(np.concatenate(logs_m), g[f"{name}/loss_log"])
    mel, sem = P.get_pred()
    _close(mel, g[f"{name}/final_pred_mel"])
    _close(sem, g[f"{name}/final_pred_semvec"])


def test_terms_in_isolation(golden_small):
    g = golden_small
    x = g["cp0"][0:1]
    vel, jerk, ll, _ = om.smoothness_loss_grad(x)
    _close(vel / op.VELOCITY_WEIGHT, g["terms/vel"].reshape(1))
    _close(jerk / op.JERK_WEIGHT, g["terms/jerk"].reshape(1))
    _close(ll / op.LOCAL_LINEAR_WEIGHT, g["terms/ll"].reshape(1))
    for taps, nm in ((om.VEL_TAPS, "vel"), (om.JERK_TAPS, "jerk"), (om.LL_TAPS, "ll")):
        _, gr = om._corr_loss_grad(x, taps, 1.0)
        _close(gr[0], g[f"terms/{nm}_grad"])
    mm = om.ManualModels(state_dict_from(g, "pred"), state_dict_from(g, "emb"))
    mel, st = mm.pred_forward(x)
    l, dmel = om.rmse_loss_grad(mel, g["target_mel"][0:1], 1.0)
    _close(l, g["terms/mel"].reshape(1))
    _close(mm.pred_backward(dmel, st, x.shape[1])[0], g["terms/mel_grad"], 1e-11)
    sem, st_e = mm.emb_forward(mel)
    l, dsem = om.rmse_loss_grad(sem, g["target_semvec"][0:1], 1.0)
    _close(l, g["terms/sem"].reshape(1))
    _close(mm.pred_backward(mm.emb_backward(dsem, st_e, mel.shape[1]), st, x.shape[1])[0], g["terms/sem_grad"], 1e-11)


def test_adam_clamp(golden_small):
    g = golden_small
    x = g["cp0"][0:1].copy()
    m, v = np.zeros_like(x), np.zeros_like(x)
    for k in range(3):
        x, m, v = om.adam_step(x, g["adam/gfield"][None] * (k + 1), m, v, k + 1)
        x = om.project(x)
    _close(x[0], g["adam/x_after_3"])


@pytest.mark.parametrize("name", ["acoustic", "acoustic_semvec"])
def test_trajectories_set_a(golden_set_a, name):
    """Paule's default architecture (H = 720).  The 8.8 M weights are regenerated, from numpy's frozen RandomState stream
    (synthetic.make_models_frozen): no torch upgrade changes them, so this test cannot be skipped; the checksum pins them."""
    g = golden_set_a
    pred_sd, emb_sd = synthetic.make_models_frozen("A")
    chk = np.array([float(sum(v.double().abs().sum() for v in pred_sd.values())), float(sum(v.double().abs().sum() for v in emb_sd.values()))])
    np.testing.assert_allclose(chk, g["weights_checksum"], rtol=1e-13)
    _close(synthetic.make_inputs_frozen(int(g["B"]), int(g["T"]))[2], g["cp0"])
    P = op.OraclePlanner(op.forward_model_from_state_dict(pred_sd), op.embedding_model_from_state_dict(emb_sd),
                         objective=name)
    P.set_targets(g["target_mel"], g["target_semvec"])
    P.set_cp(g["cp0"])
    log = P.step(20).numpy()
    _close(log, g[f"{name}/loss_log"])
    _close(P.get_cp(), g[f"{name}/cp_after_20"])


def test_per_utterance_rule(golden_small):
    """Row b of a batched run equals a B = 1 run on utterance b."""
    g = golden_small
    pred_sd, emb_sd = state_dict_from(g, "pred"), state_dict_from(g, "emb")
    full = om.ManualPlanner(pred_sd, emb_sd, objective="acoustic_semvec")
    full.set_targets(g["target_mel"], g["target_semvec"])
    full.set_cp(g["cp0"])
    lf = full.step(3)
    one = om.ManualPlanner(pred_sd, emb_sd, objective="acoustic_semvec")
    one.set_targets(g["target_mel"][1:2], g["target_semvec"][1:2])
    one.set_cp(g["cp0"][1:2])
    lo = one.step(3)
    _close(lf[:, 1:2], lo)
    _close(full.get_cp()[1:2], one.get_cp())


def test_continued_learning_step_vs_reference_fixture(golden_train):
    """OracleTrainer (paule/paule.py:287-288, :1372-1377 restated) against the reference's own ForwardModel / RMSELoss /
    Adam run: losses of 6 mini-batch steps, every parameter gradient of step 0, the parameters after 1 and 6 steps."""
    g = golden_train
    tr = op.OracleTrainer(op.forward_model_from_state_dict(state_dict_from(g, "pred")))
    cps, mel = torch.from_numpy(g["cps"]), torch.from_numpy(g["prod_mel"])
    losses = []
    for k in range(int(g["n_steps"])):
        j = g[f"batch_{k}"]
        losses.append(float(tr.train_pred_step(cps[j], mel[j])))
        if k == 0:
            for name, gr in tr.gradients().items():
                np.testing.assert_allclose(gr.numpy(), g[f"grad_step0/{name}"], rtol=0, atol=1e-14, err_msg=name)
        if k in (0, int(g["n_steps"]) - 1):
            for name, w in tr.state_dict().items():
                np.testing.assert_allclose(w.numpy(), g[f"after_{k + 1}/{name}"], rtol=0, atol=1e-13, err_msg=name)
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-13)
    with torch.no_grad():
        np.testing.assert_allclose(tr.pred_model(cps).numpy(), g["final_pred_mel"], rtol=0, atol=1e-12)


def test_inverse_model_vs_reference_fixture(golden_inverse):
    """OracleInverseModel against the reference's InverseModelMelTimeSmoothResidual (paule/models.py:177-247): raw output,
    the clipped initial CP of paule/paule.py:555, and a shorter sequence through the same parameters."""
    g = golden_inverse
    m = op.inverse_model_from_state_dict(state_dict_from(g, "inv"))
    with torch.no_grad():
        y = m(torch.from_numpy(g["mel"]))
        y13 = m(torch.from_numpy(g["mel"][:, :13].copy()))
    np.testing.assert_allclose(y.numpy(), g["cp_raw"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(y.clamp(-1, 1).numpy(), g["cp_clipped"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(y13.numpy(), g["cp_raw_13"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("variant", ["melsmooth", "upsampling"])
def test_embedder_variants_vs_reference_fixture(golden_embvar, variant):
    """OracleEmbeddingModel with a post_linear -> LeakyReLU -> mapping head, with and without residual mel blocks, against the
    reference's MelEmbeddingModelMelSmoothResidualUpsampling (paule/models.py:362-409) and EmbeddingModel(post_upsampling_size
    > 0) (:432-446): model outputs (ragged lens), and the planning trajectories of the reference loop through them."""
    g = golden_embvar
    pred_sd, emb_sd = state_dict_from(g, "pred"), state_dict_from(g, f"{variant}/emb")
    em = op.embedding_model_from_state_dict(emb_sd)
    mel = torch.from_numpy(g["target_mel"])
    with torch.no_grad():
        _close(em(mel, [torch.tensor(int(l)) for l in g[f"{variant}/embed_lens"]]), g[f"{variant}/embed_semvec_lens"])
        _close(em(mel, [torch.tensor(20)] * 3), g[f"{variant}/embed_semvec_full"])
    for objective in ("acoustic_semvec", "semvec"):
        P = op.OraclePlanner(op.forward_model_from_state_dict(pred_sd), op.embedding_model_from_state_dict(emb_sd), objective=objective)
        P.set_targets(g["target_mel"], g[f"{variant}/target_semvec"])
        P.set_cp(g["cp0"])
        logs, done = [], 0
        for k in (1, 5, 20):
            logs.append(P.step(k - done).numpy())
            done = k
            _close(P.get_cp(), g[f"{variant}/{objective}/cp_after_{k}"])
            _close(P.last_grad, g[f"{variant}/{objective}/grad_at_{k}"], 1e-11)
        _close(np.concatenate(logs), g[f"{variant}/{objective}/loss_log"])
        _close(P.get_pred()[1], g[f"{variant}/{objective}/final_pred_semvec"])


def _soma_models(g, dtype=torch.float64):
    return (op.forward_model_from_state_dict(state_dict_from(g, "cp_tube"), dtype, apply_half_sequence=False),
            op.forward_model_from_state_dict(state_dict_from(g, "tube_mel"), dtype),
            op.embedding_model_from_state_dict(state_dict_from(g, "tube_emb"), dtype))


@pytest.mark.parametrize("objective", ["acoustic_semvec", "semvec"])
def test_somatosensory_feedback_vs_reference_fixture(golden_soma, objective):
    """The somatosensory path of the loop (paule/paule.py:916-929, criterion :624-644 / :739-757) through the reference's
    ForwardModel / EmbeddingModel classes as cp_tube_model, tube_mel_model and tube_embedder (dropout 0): model outputs,
    gradients, CP and the eight loss columns (6, 7 = tube mel, tube semvec)."""
    g = golden_soma
    cp_tube, tube_mel, tube_emb = _soma_models(g)
    with torch.no_grad():
        pt = cp_tube(torch.from_numpy(g["cp0"]))
        _close(pt, g["fwd/pred_tube"])
        _close(tube_mel(pt), g["fwd/pred_tube_mel"])
        _close(tube_emb(pt, [torch.tensor(pt.shape[1])] * pt.shape[0]), g["fwd/pred_tube_semvec"])
    P = op.OraclePlanner(op.forward_model_from_state_dict(state_dict_from(g, "pred")),
                         op.embedding_model_from_state_dict(state_dict_from(g, "emb")), objective=objective,
                         tube_models=_soma_models(g))
    P.set_targets(g["target_mel"], g["target_semvec"])
    P.set_cp(g["cp0"])
    logs, done = [], 0
    for k in (1, 5, 20):
        logs.append(P.step(k - done).numpy())
        done = k
        _close(P.get_cp(), g[f"{objective}/cp_after_{k}"])
        _close(P.last_grad, g[f"{objective}/grad_at_{k}"], 1e-11)
    log = np.concatenate(logs)
    _close(log, g[f"{objective}/loss_log"])
    assert (log[:, :, 6] > 0).all() and (log[:, :, 7] > 0).all()
    with pytest.raises(ValueError):
        P.objective = "acoustic"
        P.step(1)


def test_torch_and_manual_oracles_agree_on_random_shapes():
    """The two independent restatements -- torch autograd (oracle/planner.py) and numpy with explicit BPTT (oracle/manual.py:
    the arithmetic the kernels implement) -- against each other on shapes no fixture holds (property test, hypothesis):
    batch 1..4, T 14..31 (odd lengths drop the last frame), 1..3 layers, hidden 3..17, every objective, smiling on / off."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    @settings(max_examples=12, deadline=None, suppress_health_check=[HealthCheck.too_slow])
    @given(B=st.integers(1, 4), T=st.integers(14, 31), Lp=st.integers(1, 3), Hp=st.integers(3, 17), Le=st.integers(1, 2),
           He=st.integers(3, 13), objective=st.sampled_from(["acoustic", "acoustic_semvec", "semvec"]), smiling=st.booleans())
    def check(B, T, Lp, Hp, Le, He, objective, smiling):
        wl = synthetic.make_workload(B, T, None, pred=dict(num_lstm_layers=Lp, hidden_size=Hp), emb=dict(num_lstm_layers=Le, hidden_size=He))
        P = op.OraclePlanner(op.forward_model_from_state_dict(wl.pred_sd), op.embedding_model_from_state_dict(wl.emb_sd),
                             objective=objective, smiling=smiling)
        M = om.ManualPlanner(wl.pred_sd, wl.emb_sd, objective=objective, smiling=smiling)
        for pl in (P, M):
            pl.set_targets(wl.target_mel.numpy(), wl.target_semvec.numpy())
            pl.set_cp(wl.cp0.numpy())
        lp, lm = P.step(3), M.step(3)
        _close(np.asarray(lm), lp.numpy() if hasattr(lp, "numpy") else np.asarray(lp), tol=1e-9)
        _close(np.asarray(M.get_cp()), P.get_cp().numpy(), tol=1e-9)

    check()


# ---- committed outputs of the oracle on the full-size configurations (tests/oracle_golden.py) ------------------------------------
def test_oracle_golden_fixtures_match_their_workloads():
    """Every tests/golden/oracle_<case>.npz carries the digest of the inputs it was computed for; it must be the digest of the
    workload the GPU tests will build today (generator, seeds, weights) -- a stale fixture fails here, on the CPU."""
    import oracle_golden as og
    for name in og.CASES:
        assert os.path.exists(og.path(name)), f"{og.path(name)} is missing: python tests/golden/make_oracle_golden.py {name}"
        assert og.check_digest(name), name


@pytest.mark.parametrize("name", ["cfg2_rows", "cfg4_rows"])
def test_oracle_golden_is_what_the_oracle_computes(name):
    """The fixtures are outputs of oracle/ (the repo's CPU restatement, pinned to the reference above), nothing else: two of them
    are recomputed here and compared with the committed files (loss log to 1e-12; CP trajectories are stored in float32)."""
    import oracle_golden as og
    want = dict(np.load(og.path(name)))
    got = og.compute(name)
    assert str(got["digest"]) == str(want["digest"])
    np.testing.assert_allclose(got["loss"], want["loss"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(got["cp_after"], want["cp_after"], rtol=0, atol=2e-7)


def test_oracle_golden_emulation_is_what_the_emulation_computes():
    """The emul_* arrays of the cfg3 fixture (the rounding emulation's predictor h stash and dL/dCP for utterances 0 and 255 of the
    headline batch) recomputed here from oracle/bf16_emul.py: a change of the emulation cannot leave them stale."""
    import oracle_golden as og
    want = dict(np.load(og.path("cfg3_rows")))
    got = og.compute_emul("cfg3_rows")
    np.testing.assert_array_equal(got["emul_pred_h0_bits"], want["emul_pred_h0_bits"])
    np.testing.assert_allclose(got["emul_dX"], want["emul_dX"], rtol=1e-6, atol=1e-12)


def test_oracle_sources_are_the_ones_the_fixtures_were_computed_with():
    """tests/golden/oracle_sources.sha1 = hash of oracle/{planner,manual,bf16_emul}.py when the oracle_*.npz fixtures were computed.
    The fixtures' own digests cover their INPUTS only; this covers the code: after a change of the oracle regenerate the fixtures
    (python tests/golden/make_oracle_golden.py rewrites the record) -- only cfg2 / cfg4 / the cfg3 emulation are recomputed per test run."""
    import oracle_golden as og
    assert os.path.exists(og.SOURCES_RECORD), "python tests/golden/make_oracle_golden.py --record-sources"
    assert open(og.SOURCES_RECORD).read().split()[0] == og.oracle_sources_sha1(), (
        "oracle/*.py changed since tests/golden/oracle_*.npz were computed: regenerate them with python tests/golden/make_oracle_golden.py")
