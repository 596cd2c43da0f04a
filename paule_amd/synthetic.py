"""Synthetic planning workloads: random-init models of the named architecture,
random smooth targets and initial CP trajectories.  There is no network for the pretrained weights
(paule/util.py:936-955) or data, so benchmarks and parity tests use these; timing is data independent.

Model sets:
  A -- what ``Paule`` instantiates: ForwardModel(L=1, H=720) (paule/paule.py:124) and
       EmbeddingModel(L=2, H=720) (paule/paule.py:167)                      [primary]
  B -- the class defaults: ForwardModel(L=4, H=180) (paule/models.py:335-339) and
       EmbeddingModel(L=1, H=720) (paule/models.py:421-427)                 [secondary, "stacked"]
"""
from __future__ import annotations

from collections import namedtuple

import torch

SEED = 20200905   # the reference's own module-level seed (paule/paule.py:38)

MODEL_SETS = {
    "A": dict(pred=dict(num_lstm_layers=1, hidden_size=720), emb=dict(num_lstm_layers=2, hidden_size=720)),
    "B": dict(pred=dict(num_lstm_layers=4, hidden_size=180), emb=dict(num_lstm_layers=1, hidden_size=720)),
    # Paule's predictive model with the older embedder, class defaults (MelEmbeddingModelMelSmoothResidualUpsampling,
    # paule/models.py:370-378): 3 residual mel blocks -> LSTM 4 x 180 -> post_linear 8192 -> LeakyReLU -> upsampling
    "C": dict(pred=dict(num_lstm_layers=1, hidden_size=720),
              emb=dict(num_lstm_layers=4, hidden_size=180, mel_smooth_layers=3, post_upsampling_size=8192)),
}

Workload = namedtuple("Workload", "pred_sd emb_sd target_mel target_semvec cp0 batch n_frames")

# the somatosensory models as Paule builds them (paule/paule.py:233-237, :248-252, :263-267; the tube embedder without dropout)
TUBE_SPECS = dict(cp_tube=dict(num_lstm_layers=1, hidden_size=360), tube_mel=dict(num_lstm_layers=1, hidden_size=360),
                  tube_emb=dict(num_lstm_layers=2, hidden_size=720))


def make_tube_models(cp_dim=30, tube_dim=10, mel_dim=60, sem_dim=300, dtype=torch.float64, seed=SEED, specs=None):
    """State dicts of (cp_tube_model, tube_mel_model, tube_embedder) with default torch init under ``seed + 2``."""
    sp = specs or TUBE_SPECS
    with torch.random.fork_rng():
        torch.manual_seed(seed + 2)
        return (_lstm_linear_state_dict(cp_dim, sp["cp_tube"]["hidden_size"], sp["cp_tube"]["num_lstm_layers"], tube_dim, "post_linear", dtype),
                _lstm_linear_state_dict(tube_dim, sp["tube_mel"]["hidden_size"], sp["tube_mel"]["num_lstm_layers"], mel_dim, "post_linear", dtype),
                _lstm_linear_state_dict(tube_dim, sp["tube_emb"]["hidden_size"], sp["tube_emb"]["num_lstm_layers"], sem_dim, "linear_mapping", dtype))


def _lstm_linear_state_dict(in_size, hidden, layers, out_size, lin_name, dtype):
    """torch default init (U(-1/sqrt(H), 1/sqrt(H))) in the reference's parameter-creation order."""
    lstm = torch.nn.LSTM(in_size, hidden, num_layers=layers, batch_first=True)
    lin = torch.nn.Linear(hidden, out_size)
    sd = {f"lstm.{k}": v.detach().to(dtype) for k, v in lstm.state_dict().items()}
    sd.update({f"{lin_name}.{k}": v.detach().to(dtype) for k, v in lin.state_dict().items()})
    return sd


def _melsmooth_state_dict(in_size, hidden, layers, out_size, n_blocks, post, dtype):
    """MelEmbeddingModelMelSmoothResidualUpsampling parameters, torch default init in the reference's creation order
    (paule/models.py:384-389)."""
    sd = {}
    for i in range(n_blocks):
        for j in range(3):
            conv = torch.nn.Conv1d(in_size, in_size // 3, 5, padding=2, groups=in_size // 3)
            sd.update({f"MelBlocks.{i}.ConvLayers.{j}.{k}": v.detach().to(dtype) for k, v in conv.state_dict().items()})
    lstm = torch.nn.LSTM(in_size, hidden, num_layers=layers, batch_first=True)
    sd.update({f"lstm.{k}": v.detach().to(dtype) for k, v in lstm.state_dict().items()})
    for name, lin in (("post_linear", torch.nn.Linear(hidden, post)), ("upsampling", torch.nn.Linear(post, out_size))):
        sd.update({f"{name}.{k}": v.detach().to(dtype) for k, v in lin.state_dict().items()})
    return sd


def smooth_time(x, k):
    """Moving average over time (dim 1) with window k, same length, borders averaged over valid samples."""
    return torch.nn.functional.avg_pool1d(x.permute(0, 2, 1), kernel_size=k, stride=1, padding=k // 2,
                                          count_include_pad=False).permute(0, 2, 1).contiguous()


def make_models(model_set="A", *, pred=None, emb=None, cp_dim=30, mel_dim=60, sem_dim=300, dtype=torch.float64,
                seed=SEED, with_embedder=True):
    """State dicts of (ForwardModel, EmbeddingModel) with default torch init under ``seed``."""
    spec = MODEL_SETS[model_set] if model_set else dict(pred=pred, emb=emb)
    pspec, espec = dict(spec["pred"]), dict(spec["emb"]) if spec.get("emb") else None
    with torch.random.fork_rng():
        torch.manual_seed(seed)
        pred_sd = _lstm_linear_state_dict(cp_dim, pspec["hidden_size"], pspec["num_lstm_layers"], mel_dim,
                                          "post_linear", dtype)
        emb_sd = None
        if with_embedder and espec and espec.get("post_upsampling_size"):
            emb_sd = _melsmooth_state_dict(mel_dim, espec["hidden_size"], espec["num_lstm_layers"], sem_dim,
                                           espec.get("mel_smooth_layers", 3), espec["post_upsampling_size"], dtype)
        elif with_embedder and espec:
            emb_sd = _lstm_linear_state_dict(mel_dim, espec["hidden_size"], espec["num_lstm_layers"], sem_dim,
                                             "linear_mapping", dtype)
    return pred_sd, emb_sd


def make_workload(batch, n_frames, model_set="A", *, pred=None, emb=None, cp_dim=30, mel_dim=60, sem_dim=300,
                  dtype=torch.float64, seed=SEED, with_embedder=True):
    """Models + targets + initial CP, generated on the CPU in this fixed order."""
    pred_sd, emb_sd = make_models(model_set, pred=pred, emb=emb, cp_dim=cp_dim, mel_dim=mel_dim, sem_dim=sem_dim,
                                  dtype=dtype, seed=seed, with_embedder=with_embedder)
    with torch.random.fork_rng():
        torch.manual_seed(seed + 1)
        tp = n_frames // 2
        target_mel = smooth_time(torch.rand(batch, tp, mel_dim, dtype=dtype), 5)
        target_semvec = 0.1 * torch.randn(batch, sem_dim, dtype=dtype)
        # smooth start, like the inverse model's output clipped to [-1, 1] (paule/paule.py:555)
        cp0 = smooth_time(2.0 * torch.rand(batch, n_frames, cp_dim, dtype=dtype) - 1.0, 25)
    return Workload(pred_sd, emb_sd, target_mel, target_semvec, cp0, batch, n_frames)


def make_models_frozen(model_set="A", *, cp_dim=30, mel_dim=60, sem_dim=300, dtype=torch.float64, seed=SEED):
    """State dicts of (ForwardModel, EmbeddingModel) drawn from numpy's FROZEN legacy generator (``RandomState``: its stream is
    guaranteed never to change), U(-1/sqrt(H), 1/sqrt(H)) like torch's default init, parameters in the reference's creation
    order.  For fixtures that must reproduce their weights on any torch version (tests/golden/set_a_h720.npz)."""
    import numpy as np
    spec = MODEL_SETS[model_set]
    rs = np.random.RandomState(seed)

    def lstm_linear(in_size, hidden, layers, out_size, lin_name):
        k = 1.0 / np.sqrt(hidden)
        sd = {}
        for l in range(layers):
            i = in_size if l == 0 else hidden
            for name, shape in ((f"lstm.weight_ih_l{l}", (4 * hidden, i)), (f"lstm.weight_hh_l{l}", (4 * hidden, hidden)),
                                (f"lstm.bias_ih_l{l}", (4 * hidden,)), (f"lstm.bias_hh_l{l}", (4 * hidden,))):
                sd[name] = torch.from_numpy(rs.uniform(-k, k, size=shape)).to(dtype)
        sd[f"{lin_name}.weight"] = torch.from_numpy(rs.uniform(-k, k, size=(out_size, hidden))).to(dtype)
        sd[f"{lin_name}.bias"] = torch.from_numpy(rs.uniform(-k, k, size=(out_size,))).to(dtype)
        return sd

    p, e = spec["pred"], spec["emb"]
    return (lstm_linear(cp_dim, p["hidden_size"], p["num_lstm_layers"], mel_dim, "post_linear"),
            lstm_linear(mel_dim, e["hidden_size"], e["num_lstm_layers"], sem_dim, "linear_mapping"))


def make_inputs_frozen(batch, n_frames, *, cp_dim=30, mel_dim=60, sem_dim=300, dtype=torch.float64, seed=SEED + 1):
    """(target_mel, target_semvec, cp0) from the frozen generator: smooth like make_workload's."""
    import numpy as np
    rs = np.random.RandomState(seed)
    tp = n_frames // 2
    target_mel = smooth_time(torch.from_numpy(rs.uniform(0, 1, size=(batch, tp, mel_dim))).to(dtype), 5)
    target_semvec = torch.from_numpy(0.1 * rs.standard_normal(size=(batch, sem_dim))).to(dtype)
    cp0 = smooth_time(torch.from_numpy(rs.uniform(-1, 1, size=(batch, n_frames, cp_dim))).to(dtype), 25)
    return target_mel, target_semvec, cp0
