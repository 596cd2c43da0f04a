"""Host logic on the CPU: Paule.plan_resynth validation / result contract (with the oracle engine injected as
test infrastructure) and the multi-process batch sharding over gloo (world size 2)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import state_dict_from
from oracle_engine import OracleEngine
from paule_amd import paule as pp
from paule_amd import synthetic
from paule_amd.distributed import gather_final_cp, plan_sharded, shard_bounds


def _factory(pred_model, embedder, **kw):
    return OracleEngine(pred_model, embedder, **kw)


@pytest.fixture(scope="module")
def small():
    return synthetic.make_workload(2, 24, None, pred=dict(num_lstm_layers=1, hidden_size=12),
                                   emb=dict(num_lstm_layers=1, hidden_size=10))


@pytest.fixture()
def model(small):
    return pp.Paule(pred_model=small.pred_sd, embedder=small.emb_sd, planner_factory=_factory,
                    device=torch.device("cpu"))


def test_exceptions(model, small):
    """The ValueErrors pinned by the reference's tests/test_paule.py:31-62, same messages."""
    mel = small.target_mel[0].numpy()
    cp_11 = np.zeros((11, 30))
    with pytest.raises(ValueError, match="Either target_acoustic or target_semvec"):
        model.plan_resynth(target_acoustic=None, target_semvec=None)
    with pytest.raises(ValueError, match="results can only be logged"):
        model.plan_resynth(target_acoustic=mel, target_semvec=None, n_inner=5, log_ii=10)
    with pytest.raises(ValueError, match="if target_acoustic is None you need"):
        model.plan_resynth(target_acoustic=None, target_semvec=np.zeros(300))
    with pytest.raises(ValueError, match="initialize_from has to be either"):
        model.plan_resynth(target_acoustic=mel, initialize_from="ERROR")
    with pytest.raises(ValueError, match="one of initial_cp and initialize_from has to be None"):
        model.plan_resynth(target_acoustic=mel, initial_cp=cp_11, initialize_from="ERROR")
    with pytest.raises(ValueError, match="one of initial_cp and initialize_from has to be None"):
        model.plan_resynth(target_acoustic=mel, initial_cp=cp_11)
    with pytest.raises(ValueError, match="past_cp have to be None or the sequence length"):
        model.plan_resynth(target_acoustic=mel, past_cp=cp_11)
    with pytest.raises(ValueError, match="objective has to be one of"):
        model.plan_resynth(target_acoustic=mel, initial_cp=small.cp0[0].numpy(), initialize_from=None, objective="ERROR")
    with pytest.raises(ValueError, match="initial_cp 11, target_mel 24"):
        model.plan_resynth(target_acoustic=mel, initial_cp=cp_11, initialize_from=None)
    with pytest.raises(NotImplementedError):
        pp.Paule(pred_model=small.pred_sd, embedder=small.emb_sd, use_somatosensory_feedback=True,
                 use_speech_classifier=True)


def test_plan_resynth_contract(model, small):
    """33-field PlanningResults, log-step bookkeeping (pre-step CP at log steps, post-step planned_cp), and the
    values of the loop equal a direct oracle run."""
    n_outer, n_inner, log_ii = 2, 6, 3
    res = model.plan_resynth(target_acoustic=small.target_mel[0].numpy(), target_semvec=small.target_semvec[0].numpy(),
                             initial_cp=small.cp0[0].numpy(), initialize_from=None, objective="acoustic_semvec",
                             n_outer=n_outer, n_inner=n_inner, log_ii=log_ii, continue_learning=False, log_cps=True,
                             log_gradients=True, verbose=False)
    assert len(res) == 33 and res._fields[0] == "planned_cp" and res._fields[-1] == "inv_model_loss"
    assert res.planned_cp.shape == (24, 30) and res.pred_mel.shape == (12, 60) and res.pred_semvec.shape == (300,)
    n_logs = n_outer * (n_inner // log_ii)
    assert len(res.planned_loss_steps) == n_logs == len(res.vel_loss_steps) == len(res.pred_semvec_loss_steps)
    assert len(res.cp_steps) == n_outer and len(res.cp_steps[0]) == n_inner // log_ii
    assert len(res.grad_steps) == n_outer * n_inner
    assert res.prod_mel is None and res.prod_loss_steps == []          # no synthesizer injected
    eng = OracleEngine(small.pred_sd, small.emb_sd, batch=1, n_frames=24, objective="acoustic_semvec")
    eng.set_targets(small.target_mel[0:1], small.target_semvec[0:1])
    eng.set_cp(small.cp0[0:1])
    log = eng.step(n_outer * n_inner).numpy()
    np.testing.assert_allclose(res.planned_cp, eng.get_cp()[0].numpy(), atol=1e-12)
    np.testing.assert_allclose(res.planned_loss_steps, log[log_ii - 1::log_ii, 0, 0], rtol=1e-12)
    pre = OracleEngine(small.pred_sd, small.emb_sd, batch=1, n_frames=24, objective="acoustic_semvec")
    pre.set_targets(small.target_mel[0:1], small.target_semvec[0:1])
    pre.set_cp(small.cp0[0:1])
    pre.step(log_ii - 1)                                                # CP handed out at a log step = PRE-step CP
    np.testing.assert_allclose(res.cp_steps[0][0], pre.get_cp()[0].numpy(), atol=1e-12)


def test_batched_targets_and_hooks(small):
    calls = []

    def synth(cp):
        calls.append(cp.shape)
        return np.zeros(100), 44100

    def melx(sig, sr):
        return np.full((12, 60), 0.25)

    model = pp.Paule(pred_model=small.pred_sd, embedder=small.emb_sd, planner_factory=_factory,
                     device=torch.device("cpu"), synthesizer=synth, mel_extractor=melx, smiling=True)
    res = model.plan_resynth(target_acoustic=small.target_mel.numpy(), initial_cp=small.cp0.numpy(), initialize_from=None,
                             objective="acoustic", n_outer=1, n_inner=4, log_ii=2, continue_learning=False, verbose=False)
    assert res.planned_cp.shape == (2, 24, 30)
    assert np.all(res.planned_cp[:, :, 4] == -1.0) and np.all(res.planned_cp[:, :, 1] == 1.0)   # smiling
    assert res.prod_mel.shape == (2, 12, 60) and len(res.prod_loss_steps) == 2
    assert len(calls) == 2 * (1 + 2)                                    # initial + two log steps, per utterance
    assert model.best_synthesis_acoustic.mel_loss < np.inf


def _reference_smoke_model(small, golden_inverse, factory, device):
    """Paule as tests/test_paule.py:21-24 builds it, with what the offline box lacks handed in: small random-init models
    instead of the downloaded weights, a stub synthesiser + mel extractor instead of VocalTractLab + librosa."""
    return pp.Paule(pred_model={k: v.clone() for k, v in small.pred_sd.items()}, embedder=small.emb_sd,
                    inv_model=state_dict_from(golden_inverse, "inv"), planner_factory=factory, device=device,
                    synthesizer=lambda cp: (np.zeros(100), 44100),
                    mel_extractor=lambda sig, sr: np.full((20, 60), 0.3))


def test_plan_resynth_like_the_reference_test(small, golden_inverse):
    """tests/test_paule.py:65-70 with its own arguments (objective='acoustic_semvec', initialize_from='acoustic', n_outer=2,
    n_inner=2, n_batches=1, batch_size=2, n_epochs=2; log_ii and continue_learning at their defaults 1 / True): inverse-model
    initialisation, a log step with synthesis every iteration, continued learning after each outer iteration.  The reference
    asserts nothing about the result; here its shape contract is."""
    model = _reference_smoke_model(small, golden_inverse, _factory, torch.device("cpu"))
    results = model.plan_resynth(target_acoustic=golden_inverse["mel"][0], objective='acoustic_semvec',
                                 initialize_from='acoustic', n_outer=2, n_inner=2, n_batches=1, batch_size=2,
                                 n_epochs=2, verbose=False)
    assert results.planned_cp.shape == (40, 30) and results.initial_cp.shape == (40, 30)
    assert len(results.planned_loss_steps) == 4 and len(results.prod_loss_steps) == 4      # log_ii = 1: every iteration
    assert len(results.pred_model_loss) == 2 * 2                                           # n_outer x n_epochs
    assert results.prod_mel.shape == (20, 60) and results.pred_semvec.shape == (300,)


def test_create_epoch_batches():
    """Same-size batching (paule/paule.py:349-371): every sample exactly once, full batches hold one length, only the
    left-over batches mix lengths, at most one batch is smaller; plain mode wraps around (:373-381)."""
    lens = np.array([40] * 7 + [52] * 5 + [60] * 3)
    by_len = {int(l): np.where(lens == l)[0] for l in np.unique(lens)}
    epoch = pp.Paule.create_epoch_batches(len(lens), 4, same_size_batching=True, training_length_dict=by_len)
    flat = np.concatenate([np.asarray(b) for b in epoch])
    assert sorted(flat.tolist()) == list(range(len(lens)))
    assert sorted(len(b) for b in epoch) == [3, 4, 4, 4]
    assert sum(len(set(lens[np.asarray(b)])) == 1 for b in epoch) >= 2
    plain = pp.Paule.create_epoch_batches(10, 4, shuffle=False)
    assert plain == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 0, 1]]
    with pytest.raises(ValueError, match="Dictionary containing indices"):
        pp.Paule.create_epoch_batches(10, 4, same_size_batching=True)


def test_continue_learning_on_the_planner(small):
    """continue_learning=True (paule/paule.py:1243-1407): after every outer iteration the produced (cp, mel) pairs train
    the predictive model through the planner's train_pred_step; pred_model_loss gets one mean per epoch, the module's
    parameters follow the planner's, and the next outer iteration plans on the trained model."""
    def synth(cp):
        return np.zeros(100), 44100

    def melx(sig, sr):
        return np.full((12, 60), 0.25)

    model = pp.Paule(pred_model=small.pred_sd, embedder=small.emb_sd, planner_factory=_factory,
                     device=torch.device("cpu"), synthesizer=synth, mel_extractor=melx)
    before = {k: v.clone() for k, v in model.pred_model.items()}
    res = model.plan_resynth(target_acoustic=small.target_mel.numpy(), initial_cp=small.cp0.numpy(), initialize_from=None,
                             objective="acoustic", n_outer=2, n_inner=4, log_ii=2, continue_learning=True, n_batches=2,
                             batch_size=2, n_epochs=3, seed=7, verbose=False)
    assert len(res.pred_model_loss) == 2 * 3 and all(np.isfinite(res.pred_model_loss))
    assert res.pred_model_loss[-1] < res.pred_model_loss[0]            # the constant produced mel is easy to fit
    after = model.pred_model
    assert any(not torch.equal(before[k].double(), after[k].double()) for k in before)
    # 2 log steps x 2 utterances = 4 produced samples = batch_size * n_batches: 2 steps per epoch, 3 epochs, 2 outer its
    changed = max(float((before[k].double() - after[k].double()).abs().max()) for k in before)
    assert 0 < changed <= 0.001 * 2 * 3 * 2 * 1.1                       # 12 Adam steps move a parameter by ~lr each


def test_initialize_from_acoustic_on_the_planner(small, golden_inverse):
    """initialize_from='acoustic' (paule/paule.py:550-556) with the inverse model handed in as a state dict: the planner is
    built first and runs inv_model(target_mel).clip(-1, 1) itself; with past_cp the plan is past + 2 x mel frames long."""
    inv_sd = state_dict_from(golden_inverse, "inv")
    model = pp.Paule(pred_model=small.pred_sd, embedder=small.emb_sd, inv_model=inv_sd, planner_factory=_factory,
                     device=torch.device("cpu"))
    mel = golden_inverse["mel"][:2]                                   # (2, 20, 60)
    res = model.plan_resynth(target_acoustic=mel, initialize_from="acoustic", objective="acoustic", n_outer=1, n_inner=2,
                             log_ii=2, continue_learning=False, verbose=False)
    np.testing.assert_allclose(res.initial_cp, golden_inverse["cp_clipped"][:2], atol=1e-12)
    assert res.planned_cp.shape == (2, 40, 30)
    past = np.zeros((6, 30))
    res = model.plan_resynth(target_acoustic=mel, target_semvec=small.target_semvec.numpy(), initialize_from="acoustic",
                             past_cp=past, objective="acoustic", n_outer=1, n_inner=2, log_ii=2, continue_learning=False,
                             verbose=False)
    assert res.planned_cp.shape == (2, 46, 30)
    np.testing.assert_allclose(res.initial_cp[:, 6:], golden_inverse["cp_clipped"][:2], atol=1e-12)


def test_pred_optimizer_outlives_a_plan(small):
    """The reference's pred_optimizer belongs to the Paule instance (paule/paule.py:284-287): its Adam state carries over from
    one plan_resynth call to the next, `param_groups[0]['lr']` follows learning_rate_learning (:473-474), and state_dict() is
    what docs/examples/minimal_example.py:51 saves."""
    model = pp.Paule(pred_model={k: v.clone() for k, v in small.pred_sd.items()}, embedder=small.emb_sd, planner_factory=_factory,
                     device=torch.device("cpu"), synthesizer=lambda cp: (np.zeros(100), 44100),
                     mel_extractor=lambda sig, sr: np.full((12, 60), 0.25))
    kw = dict(target_acoustic=small.target_mel.numpy(), initial_cp=small.cp0.numpy(), initialize_from=None, objective="acoustic",
              n_outer=1, n_inner=2, log_ii=2, continue_learning=True, n_batches=1, batch_size=2, n_epochs=3, verbose=False)
    assert model.pred_optimizer.state_dict()["state"] == {}
    model.plan_resynth(learning_rate_learning=0.002, **kw)
    sd = model.pred_optimizer.state_dict()
    assert model.pred_optimizer.param_groups[0]["lr"] == 0.002
    assert float(sd["state"][0]["step"]) == 3 and set(sd["state"][0]) >= {"step", "exp_avg", "exp_avg_sq"}
    assert len(sd["state"]) == len(small.pred_sd)
    model.plan_resynth(learning_rate_learning=0.002, **kw)
    assert float(model.pred_optimizer.state_dict()["state"][0]["step"]) == 6     # continued, not restarted
    import pickle
    pickle.loads(pickle.dumps(model.pred_optimizer.state_dict()))


def test_continue_learning_with_stored_training_data(small):
    """add_training_data_pred=True (paule/paule.py:1250-1287): half of every training batch comes from continue_data, whose
    samples may be shorter than the plan (padded to the batch's longest sample by repeating the last frame); the produced
    samples are appended to continue_data afterwards (:1439-1443), here given as a pandas DataFrame like the reference's."""
    import pandas as pd
    rng = np.random.default_rng(3)
    stored = pd.DataFrame({"vector": [np.zeros(300)] * 6,
                           "cp_norm": [rng.uniform(-1, 1, (t, 30)) for t in (24, 24, 20, 20, 16, 24)],
                           "melspec_norm_synthesized": [rng.uniform(0, 1, (t // 2, 60)) for t in (24, 24, 20, 20, 16, 24)],
                           "segment_data": [False] * 6})
    model = pp.Paule(pred_model=small.pred_sd, embedder=small.emb_sd, planner_factory=_factory, device=torch.device("cpu"),
                     synthesizer=lambda cp: (np.zeros(100), 44100), mel_extractor=lambda sig, sr: np.full((12, 60), 0.25),
                     continue_data=stored)
    res = model.plan_resynth(target_acoustic=small.target_mel.numpy(), target_semvec=small.target_semvec.numpy(),
                             initial_cp=small.cp0.numpy(), initialize_from=None, objective="acoustic", n_outer=1, n_inner=4,
                             log_ii=2, continue_learning=True, add_training_data_pred=True, n_batches=2, batch_size=2, n_epochs=2,
                             seed=3, verbose=False)
    assert len(res.pred_model_loss) == 2 and all(np.isfinite(res.pred_model_loss))
    assert len(model.continue_data) == 6 + 4 and list(model.continue_data.columns) == list(stored.columns)
    assert model.continue_data["cp_norm"].iloc[-1].shape == (24, 30)
    with pytest.raises(ValueError, match="needs continue_data"):
        pp.Paule(pred_model=small.pred_sd, embedder=small.emb_sd, planner_factory=_factory, device=torch.device("cpu"),
                 synthesizer=lambda cp: (np.zeros(100), 44100), mel_extractor=lambda sig, sr: np.full((12, 60), 0.25)
                 ).plan_resynth(target_acoustic=small.target_mel.numpy(), initial_cp=small.cp0.numpy(), initialize_from=None,
                                objective="acoustic", n_outer=1, n_inner=2, log_ii=2, continue_learning=True,
                                add_training_data_pred=True, verbose=False)


def test_default_models_from_pretrained_dir(small, golden_inverse, tmp_path, monkeypatch):
    """Paule() without models loads the reference's pretrained files (paule/paule.py:121-127, :146-150, :167-171) from
    $PAULE_PRETRAINED_DIR when they are there, and says which file is missing when they are not."""
    monkeypatch.setenv("PAULE_PRETRAINED_DIR", str(tmp_path))
    with pytest.raises(FileNotFoundError, match="pred_model_common_voice_1_720"):
        pp.Paule(planner_factory=_factory, device=torch.device("cpu"))
    files = {"predictive/pred_model_common_voice_1_720_lr_0001_50_00001_50_000001_50_0000001_200.pt": small.pred_sd,
             "embedder/embed_model_common_voice_syn_rec_2_720_0_dropout_07_noise_6e05_rmse_lr_00001_200.pt": small.emb_sd,
             "inverse/inv_model_common_voice_3_1_720_5_lr_0001_50_00001_50_000001_50_0000001_200.pt": state_dict_from(golden_inverse, "inv")}
    for rel, sd in files.items():
        (tmp_path / rel).parent.mkdir(parents=True, exist_ok=True)
        torch.save({k: v.clone() for k, v in sd.items()}, tmp_path / rel)
    model = pp.Paule(planner_factory=_factory, device=torch.device("cpu"))
    assert isinstance(model.pred_model, torch.nn.Module) and set(model.pred_model.state_dict()) == set(small.pred_sd)
    assert set(model.inv_model.state_dict()) == set(files[list(files)[2]]) and set(model.embedder.state_dict()) == set(small.emb_sd)
    res = model.plan_resynth(target_acoustic=golden_inverse["mel"][:1], initialize_from="acoustic", objective="acoustic", n_outer=1,
                             n_inner=2, log_ii=2, continue_learning=False, verbose=False)
    np.testing.assert_allclose(res.initial_cp, golden_inverse["cp_clipped"][0], atol=1e-12)


def test_embedder_variant_containers_keep_the_reference_key_layout(golden_embvar):
    """paule_amd.models.MelEmbeddingModelMelSmoothResidualUpsampling / EmbeddingModel(post_upsampling_size > 0) take the state
    dicts of the reference's classes as they are (keys and shapes of the fixture come from the reference's modules); unsupported
    constructor choices are refused instead of being computed differently."""
    import torch
    from paule_amd import models
    g = golden_embvar
    for variant, mod in (("melsmooth", models.MelEmbeddingModelMelSmoothResidualUpsampling(hidden_size=20, num_lstm_layers=2,
                                                                                          post_upsampling_size=96)),
                         ("upsampling", models.EmbeddingModel(hidden_size=20, num_lstm_layers=1, post_upsampling_size=64))):
        sd = state_dict_from(g, f"{variant}/emb")
        own = mod.state_dict()
        assert list(own) == list(sd)
        assert all(tuple(own[k].shape) == tuple(sd[k].shape) for k in sd)
        mod.double().load_state_dict(sd)
    d = models.MelEmbeddingModelMelSmoothResidualUpsampling()
    assert d.post_linear.out_features == 8192 and d.lstm.num_layers == 4 and d.lstm.hidden_size == 180 and len(d.MelBlocks) == 3
    with pytest.raises(NotImplementedError):
        models.MelEmbeddingModelMelSmoothResidualUpsampling(post_activation=torch.nn.ReLU())
    with pytest.raises(NotImplementedError):
        models.MelEmbeddingModelMelSmoothResidualUpsampling(mel_resid_activation=torch.nn.ReLU())
    with pytest.raises(NotImplementedError):
        models.EmbeddingModel(post_upsampling_size=16, post_activation=torch.nn.LeakyReLU(0.2))


def _soma_paule(g, factory, device, **kw):
    from conftest import state_dict_from as sdf
    return pp.Paule(pred_model=sdf(g, "pred"), embedder=sdf(g, "emb"), use_somatosensory_feedback=True,
                    cp_tube_model=sdf(g, "cp_tube"), tube_mel_model=sdf(g, "tube_mel"), tube_embedder=sdf(g, "tube_emb"),
                    planner_factory=factory, device=device, **kw)


def test_somatosensory_feedback_on_the_planner(golden_soma):
    """Paule(use_somatosensory_feedback=True, cp_tube_model=..., tube_mel_model=..., tube_embedder=...): the host code around
    the planner (here the CPU oracle behind the planner interface) -- result type and fields of the reference
    (paule/paule.py:59, :1534-1541), logged tube losses = columns 6 / 7 of the reference loop's loss log (fixture), the
    production side through the tube_extractor hook, and the refusals."""
    g = golden_soma
    factory = lambda pm, em, **kw: OracleEngine(pm, em, **kw)
    calls = []

    def tube_extractor(cp):
        calls.append(cp.shape)
        return np.full(cp.shape[:2] + (10,), 0.1)

    model = _soma_paule(g, factory, torch.device("cpu"), tube_extractor=tube_extractor)
    res = model.plan_resynth(target_acoustic=g["target_mel"], target_semvec=g["target_semvec"], initial_cp=g["cp0"],
                             initialize_from=None, objective="acoustic_semvec", n_outer=1, n_inner=20, log_ii=5,
                             continue_learning=False, log_cps=True, verbose=False)
    assert type(res).__name__ == "PlanningResultsWithSomatosensory" and len(res._fields) == 58
    np.testing.assert_allclose(res.planned_cp, g["acoustic_semvec/cp_after_20"], atol=1e-12, rtol=0)
    log = g["acoustic_semvec/loss_log"]                     # (20, B, 8); log steps are iterations 5, 10, 15, 20 (1-based)
    np.testing.assert_allclose(np.asarray(res.pred_tube_mel_loss_steps), log[[4, 9, 14, 19], :, 6], rtol=1e-12)
    np.testing.assert_allclose(np.asarray(res.pred_tube_semvec_loss_steps), log[[4, 9, 14, 19], :, 7], rtol=1e-12)
    np.testing.assert_allclose(np.asarray(res.planned_loss_steps), log[[4, 9, 14, 19], :, 0], rtol=1e-12)
    np.testing.assert_allclose(res.initial_pred_tube, g["fwd/pred_tube"], atol=1e-12, rtol=0)
    np.testing.assert_allclose(res.initial_pred_tube_mel, g["fwd/pred_tube_mel"], atol=1e-12, rtol=0)
    np.testing.assert_allclose(res.initial_pred_tube_semvec, g["fwd/pred_tube_semvec"], atol=1e-12, rtol=0)
    assert res.pred_tube.shape == (3, 40, 10) and res.pred_tube_mel.shape == (3, 20, 60) and res.pred_tube_semvec.shape == (3, 300)
    assert len(calls) == 1 + 4 and len(res.prod_tube_loss_steps) == 4 and len(res.prod_tube_steps[0]) == 4
    assert res.prod_tube.shape == (3, 40, 10) and res.prod_tube_mel.shape == (3, 20, 60)
    assert model.best_synthesis_somatosensory.tube_loss < np.inf
    with pytest.raises(NotImplementedError, match="acoustic"):
        model.plan_resynth(target_acoustic=g["target_mel"], initial_cp=g["cp0"], initialize_from=None, objective="acoustic",
                           n_outer=1, n_inner=1, continue_learning=False, verbose=False)
    with pytest.raises(NotImplementedError, match="tube_embedder"):
        pp.Paule(pred_model=g, embedder=g, use_somatosensory_feedback=True, planner_factory=factory, device=torch.device("cpu"))
    drop = torch.nn.Module()
    drop.lstm = torch.nn.LSTM(10, 4, num_layers=2, dropout=0.7)
    with pytest.raises(NotImplementedError, match="dropout"):
        pp.Paule(pred_model=g, embedder=g, use_somatosensory_feedback=True, cp_tube_model=g, tube_mel_model=g, tube_embedder=drop,
                 planner_factory=factory, device=torch.device("cpu"))
    with pytest.raises(NotImplementedError):
        pp.Paule(pred_model=g, embedder=g, use_somatosensory_feedback=True, use_speech_classifier=True, planner_factory=factory)


def test_continue_learning_tube_on_the_planner(golden_soma):
    """continue_learning_tube=True (paule/paule.py:1381-1409): every mini-batch of the continued learning also trains the
    cp -> tube and the tube -> mel model on the produced tubes; epoch means land in tube_model_loss / tube_mel_model_loss, the
    instance's models change, produced samples carry 'tube_norm'."""
    g = golden_soma
    factory = lambda pm, em, **kw: OracleEngine(pm, em, **kw)
    rng = np.random.default_rng(0)
    model = _soma_paule(g, factory, torch.device("cpu"), tube_extractor=lambda cp: 0.3 * rng.standard_normal(cp.shape[:2] + (10,)),
                        synthesizer=lambda cp: (np.zeros(100), 44100), mel_extractor=lambda sig, sr: np.full((20, 60), 0.25),
                        continue_data=[])
    before = {k: v.clone() for k, v in model.cp_tube_model.items()}
    res = model.plan_resynth(target_acoustic=g["target_mel"][:2], target_semvec=g["target_semvec"][:2], initial_cp=g["cp0"][:2],
                             initialize_from=None, objective="acoustic_semvec", n_outer=2, n_inner=4, log_ii=2, continue_learning=True,
                             continue_learning_tube=True, n_batches=2, batch_size=2, n_epochs=3, seed=3, verbose=False)
    assert len(res.tube_model_loss) == 2 * 3 and len(res.tube_mel_model_loss) == 2 * 3 and len(res.pred_model_loss) == 2 * 3
    assert all(np.isfinite(res.tube_model_loss)) and res.tube_model_loss[-1] < res.tube_model_loss[0]
    assert any(not torch.equal(torch.as_tensor(model.cp_tube_model[k]), before[k]) for k in before)
    assert all("tube_norm" in rec and rec["tube_norm"].shape == (40, 10) for rec in model.continue_data)
    with pytest.raises(NotImplementedError, match="tube_extractor"):
        _soma_paule(g, factory, torch.device("cpu")).plan_resynth(
            target_acoustic=g["target_mel"][:2], target_semvec=g["target_semvec"][:2], initial_cp=g["cp0"][:2], initialize_from=None,
            objective="semvec", n_outer=1, n_inner=2, continue_learning=True, continue_learning_tube=True, verbose=False)


def test_speech_classifier_config(small):
    """minimal_example.py's configuration (use_speech_classifier=True, acoustic_semvec; docs/examples/minimal_example.py:13-47)."""
    clf = {"linear.weight": torch.full((1, 60), 0.05, dtype=torch.float64), "linear.bias": torch.tensor([0.3], dtype=torch.float64)}
    with pytest.raises(FileNotFoundError, match="speech_classifier"):
        pp.Paule(pred_model=small.pred_sd, embedder=small.emb_sd, use_speech_classifier=True)
    model = pp.Paule(pred_model=small.pred_sd, embedder=small.emb_sd, planner_factory=_factory, device=torch.device("cpu"),
                     use_speech_classifier=True, speech_classifier=clf)
    res = model.plan_resynth(target_acoustic=small.target_mel[0].numpy(), initial_cp=small.cp0[0].numpy(),
                             initialize_from=None, objective="acoustic_semvec", n_outer=1, n_inner=4, log_ii=2,
                             continue_learning=False, verbose=False)
    assert type(res).__name__ == "PlanningResultsWithSpeechClassifier" and len(res) == 35
    assert len(res.pred_speech_classifier_loss_steps) == 2 and all(v > 0 for v in res.pred_speech_classifier_loss_steps)
    ref = model.plan_resynth   # the logged total contains the classifier term
    assert res.planned_loss_steps[0] > res.planned_mel_loss_steps[0] + res.pred_speech_classifier_loss_steps[0]


def test_shard_bounds():
    assert [shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [shard_bounds(2048, r, 8) for r in range(8)][-1] == (1792, 2048)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    wl = synthetic.make_workload(5, 24, None, pred=dict(num_lstm_layers=1, hidden_size=12),
                                 emb=dict(num_lstm_layers=1, hidden_size=10))
    mk = lambda batch: OracleEngine(wl.pred_sd, wl.emb_sd, batch=batch, n_frames=24, objective="acoustic_semvec")
    cp_all, loss = plan_sharded(mk, wl.cp0, wl.target_mel, wl.target_semvec, 4)
    if rank == 0:
        torch.save(cp_all, out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_planning_world2_gloo(tmp_path):
    """N > 1 path: two ranks plan 3 + 2 utterances, one all_gather; equals the single-process batch of 5."""
    out = str(tmp_path / "cp.pt")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    cp_all = torch.load(out)
    wl = synthetic.make_workload(5, 24, None, pred=dict(num_lstm_layers=1, hidden_size=12),
                                 emb=dict(num_lstm_layers=1, hidden_size=10))
    eng = OracleEngine(wl.pred_sd, wl.emb_sd, batch=5, n_frames=24, objective="acoustic_semvec")
    eng.set_targets(wl.target_mel, wl.target_semvec)
    eng.set_cp(wl.cp0)
    eng.step(4)
    np.testing.assert_allclose(cp_all.numpy(), eng.get_cp().numpy(), atol=1e-13)


class _FailingEngine(OracleEngine):
    """An engine whose device status check fails after stepping (what HipPlanner.check() does after a timed-out in-kernel wait)."""

    def check(self):
        raise ValueError("persistent LSTM sweep: a bounded in-kernel wait timed out (injected)")


def _worker_one_rank_fails(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from paule_amd.distributed import ShardFailed
    wl = synthetic.make_workload(5, 24, None, pred=dict(num_lstm_layers=1, hidden_size=12),
                                 emb=dict(num_lstm_layers=1, hidden_size=10))
    cls = _FailingEngine if rank == 1 else OracleEngine
    mk = lambda batch: cls(wl.pred_sd, wl.emb_sd, batch=batch, n_frames=24, objective="acoustic_semvec")
    try:
        plan_sharded(mk, wl.cp0, wl.target_mel, wl.target_semvec, 2)
        verdict = "returned"
    except ShardFailed as e:
        verdict = "own" if e.__cause__ is not None else "peer"
    with open(f"{out}.{rank}", "w") as f:
        f.write(verdict)
    dist.barrier()   # both ranks are still able to meet: nobody is stuck in the gather
    dist.destroy_process_group()


def test_sharded_planning_one_rank_fails_everybody_raises(tmp_path):
    """a rank whose planner.check() raises must not leave its peers blocked in the all_gather.  Rank 1's
    engine fails its status check; BOTH ranks raise ShardFailed (rank 1 with its own error as the cause) and meet again at a
    barrier afterwards -- the processes are joined with a timeout, a hang fails the test."""
    out = str(tmp_path / "verdict")
    port = 29500 + (os.getpid() + 7) % 2000
    ctx = mp.spawn(_worker_one_rank_fails, args=(2, port, out), nprocs=2, join=False)
    import time
    deadline = time.time() + 120
    while not ctx.join(timeout=1):
        assert time.time() < deadline, "plan_sharded left a rank blocked in a collective"
    assert open(out + ".0").read() == "peer" and open(out + ".1").read() == "own"


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` outside a torch.distributed launcher starts N ranks itself (child processes), they meet, take the
    max over ranks of the timed region, gather CP-shaped tensors and rank 0 prints ONE JSON line.  Here without an engine
    (--plumbing-only: no GPU in this container); tests/test_hip_parity.py::test_bench_two_ranks_on_one_gpu runs the real thing."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "rehearsal", "--plumbing-only", "--steps", "2"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["plumbing_only"] and out["n_gpus"] == 2 and out["gathered_rows"] == 16 and out["gathered_ranks_ok"]
    assert out["ms_per_step"] >= 0.02 * 1e3 / 2   # the slowest rank's time, not rank 0's


def test_planner_cache_is_keyed_by_architecture_and_lru(small):
    """the handle cache of Paule._get_planner -- keyed by the models' architecture (not by id(): a freed model's id can
    come back), least recently used out, bounded by bytes, and the handle published as `paule.planner` is never closed under the
    caller."""
    closed = []

    class Eng(OracleEngine):
        device_bytes = 1 << 30

        def close(self):
            closed.append(self)

    model = pp.Paule(pred_model=small.pred_sd, embedder=small.emb_sd, planner_factory=lambda p, e, **kw: Eng(p, e, **kw),
                     device=torch.device("cpu"))
    kw = dict(objective="acoustic", dtype="f32", lr=0.01, smiling=False, device="cpu")
    a = model._get_planner(batch=1, n_frames=24, **kw)
    assert model._get_planner(batch=1, n_frames=24, **kw) is a                    # same shape, same architecture: reused
    model.pred_model = {k: v.clone() for k, v in model.pred_model.items()} if isinstance(model.pred_model, dict) else model.pred_model
    assert model._get_planner(batch=1, n_frames=24, **kw) is a                    # another OBJECT of the same architecture: still reused
    model.planner = a                                                             # published to the caller
    model.PLANNER_CACHE_ENTRIES = 3
    others = [model._get_planner(batch=1, n_frames=24 + 2 * i, **kw) for i in range(1, 5)]
    assert a not in closed                                                        # never the published one
    assert closed == others[:2]                                                   # the least recently used of the rest went first
    assert len(model._planners) == 3
    model.PLANNER_CACHE_BYTES = 2 << 30                                           # a byte bound below what is cached: trims further
    b = model._get_planner(batch=2, n_frames=24, **kw)
    assert b not in closed and a not in closed and len(model._planners) == 2
