"""``paule.models`` API surface for the planning path: ForwardModel, EmbeddingModel and the inverse model of the
initialisation (InverseModelMelTimeSmoothResidual).

Same constructor signatures, defaults and ``state_dict`` key layout as the reference
(paule/models.py:335-346, :421-437), so ``load_state_dict`` of the reference's pretrained
weights works unchanged.  The ``torch.nn.LSTM`` / ``torch.nn.Linear`` sub-modules are parameter
containers only: ``forward`` runs the hand-written HIP kernels through the C-ABI.  There is no CPU
fallback -- a CPU input raises, as does a missing libpaule_hip.so.
"""
from __future__ import annotations

import torch

from .engine import HipPlanner
from ._capi import HipLibraryError


def _require_gpu(x, who):
    if not isinstance(x, torch.Tensor) or x.device.type != "cuda":
        raise HipLibraryError(f"{who}.forward needs a tensor on an MI355X (cuda/HIP) device; "
                              "paule_amd has no CPU execution path")


def _check_leaky(act, what):
    if not isinstance(act, torch.nn.LeakyReLU) or abs(act.negative_slope - 0.01) > 1e-12:
        raise NotImplementedError(f"{what}: only torch.nn.LeakyReLU() (slope 0.01, the reference's default) runs on the HIP path")


class _HipModule(torch.nn.Module):
    """Caches one forward-only engine handle per (batch, frames, device, dtype) and re-uploads the weights
    when any parameter changed (version counters), e.g. after continued learning."""

    def __init__(self):
        super().__init__()
        self._engines = {}
        self.compute_dtype = "f32"

    def _versions(self):
        return tuple(p._version for p in self.parameters())

    def _engine(self, key, make):
        ent = self._engines.get(key)
        if ent is None:
            ent = [make(), self._versions()]
            self._engines[key] = ent
        elif ent[1] != self._versions():
            self._refresh(ent[0])
            ent[1] = self._versions()
        return ent[0]

    def release_engines(self):
        for eng, _ in self._engines.values():
            eng.close()
        self._engines = {}


class ForwardModel(_HipModule):
    """CP -> mel predictive model: stacked LSTM -> Linear -> AvgPool1d(2, 2) over time
    (paule/models.py:326-356)."""

    def __init__(self, input_size=30, output_size=60, hidden_size=180, num_lstm_layers=4,
                 apply_half_sequence=True):
        super().__init__()
        self.apply_half_sequence = apply_half_sequence   # False: every frame's output (how Paule builds cp_tube_model, paule/paule.py:232-237)
        self.lstm = torch.nn.LSTM(input_size, hidden_size, num_layers=num_lstm_layers, batch_first=True)
        self.post_linear = torch.nn.Linear(hidden_size, output_size)

    def _refresh(self, eng):
        eng.set_weights(pred_model=self.state_dict())

    def forward(self, x, *args):
        _require_gpu(x, "ForwardModel")
        B, T, _ = x.shape
        key = (B, T, x.device.index, self.compute_dtype)
        eng = self._engine(key, lambda: HipPlanner(self.state_dict(), None, batch=B, n_frames=T,
                                                   dtype=self.compute_dtype, device=x.device))
        eng.set_cp(x)
        if not self.apply_half_sequence:
            return eng.get_pred_frames().to(x.dtype)
        mel, _ = eng.get_pred(with_semvec=False)
        return mel.to(x.dtype)


class EmbeddingModel(_HipModule):
    """mel -> semantic vector: stacked LSTM, output at lens-1, Linear (paule/models.py:413-448;
    post_upsampling_size = 0 path, which is what Paule instantiates, paule/paule.py:167)."""

    def __init__(self, input_size=60, output_size=300, hidden_size=720, num_lstm_layers=1,
                 post_activation=torch.nn.LeakyReLU(), post_upsampling_size=0, dropout=0):
        super().__init__()
        if dropout:
            raise NotImplementedError("dropout > 0 is not on the planning path (Paule uses dropout 0)")
        _check_leaky(post_activation, "post_activation")
        self.post_upsampling_size = post_upsampling_size
        self.lstm = torch.nn.LSTM(input_size, hidden_size, num_layers=num_lstm_layers, batch_first=True, dropout=dropout)
        if post_upsampling_size > 0:      # paule/models.py:432-435: post_linear -> post_activation -> linear_mapping
            self.post_linear = torch.nn.Linear(hidden_size, post_upsampling_size)
            self.linear_mapping = torch.nn.Linear(post_upsampling_size, output_size)
        else:
            self.linear_mapping = torch.nn.Linear(hidden_size, output_size)
        self._pred_stub = None

    def _refresh(self, eng):
        eng.set_weights(embedder=self.state_dict())

    def _stub_pred_sd(self, mel_dim):
        # the engine handle always carries a predictive model; a 1-unit placeholder feeds nothing here
        if self._pred_stub is None or self._pred_stub["post_linear.weight"].shape[0] != mel_dim:
            z = torch.zeros
            self._pred_stub = {"lstm.weight_ih_l0": z(4, 30), "lstm.weight_hh_l0": z(4, 1), "lstm.bias_ih_l0": z(4),
                               "lstm.bias_hh_l0": z(4), "post_linear.weight": z(mel_dim, 1), "post_linear.bias": z(mel_dim)}
        return self._pred_stub

    def forward(self, x, lens, *args):
        _require_gpu(x, "EmbeddingModel")
        B, Tp, M = x.shape
        key = (B, Tp, x.device.index, self.compute_dtype)
        eng = self._engine(key, lambda: HipPlanner(self._stub_pred_sd(M), self.state_dict(), batch=B,
                                                   n_frames=max(2 * Tp, 14), dtype=self.compute_dtype, device=x.device))
        if eng.Tp != Tp:
            raise ValueError("mel sequences shorter than 7 frames are not supported by the HIP engine")
        lens = [int(l) for l in lens]
        if len(lens) == 1 and B > 1:
            lens = lens * B
        return eng.embed_mel(x, lens).to(x.dtype)


class MelEmbeddingModelMelSmoothResidualUpsampling(EmbeddingModel):
    """The older embedder (paule/models.py:362-409): residual MelChannelConv1D blocks -> stacked LSTM, output at lens-1 ->
    post_linear -> LeakyReLU -> upsampling.  Same state_dict keys as the reference (``MelBlocks.i.ConvLayers.j.*``, ``lstm.*``,
    ``post_linear.*``, ``upsampling.*``); usable as ``Paule(embedder=...)``: the planning loop differentiates through it on
    the device like through the default embedder."""

    def __init__(self, input_size=60, output_size=300, hidden_size=180, num_lstm_layers=4, mel_smooth_layers=3,
                 mel_smooth_filter_size=3, mel_resid_activation=torch.nn.Identity(), post_activation=torch.nn.LeakyReLU(),
                 post_upsampling_size=8192):
        _HipModule.__init__(self)
        if mel_smooth_filter_size != 3:
            raise NotImplementedError("only the default mel_smooth_filter_size = 3 is supported")
        if not isinstance(mel_resid_activation, torch.nn.Identity):
            raise NotImplementedError("only the default Identity mel_resid_activation is supported")
        if post_upsampling_size <= 0:
            raise ValueError("post_upsampling_size has to be positive")
        _check_leaky(post_activation, "post_activation")
        self.post_upsampling_size = post_upsampling_size
        self.MelBlocks = torch.nn.ModuleList([_MelChannelConv1D(input_size, mel_smooth_filter_size) for _ in range(mel_smooth_layers)])
        self.lstm = torch.nn.LSTM(input_size, hidden_size, num_layers=num_lstm_layers, batch_first=True)
        self.post_linear = torch.nn.Linear(hidden_size, post_upsampling_size)
        self.upsampling = torch.nn.Linear(post_upsampling_size, output_size)
        self._pred_stub = None


def _stub_pred_sd(mel_dim, cp_dim=30):
    z = torch.zeros
    return {"lstm.weight_ih_l0": z(4, cp_dim), "lstm.weight_hh_l0": z(4, 1), "lstm.bias_ih_l0": z(4), "lstm.bias_hh_l0": z(4),
            "post_linear.weight": z(mel_dim, 1), "post_linear.bias": z(mel_dim)}


class _MelChannelConv1D(torch.nn.Module):          # parameter container, paule/models.py:142-150
    def __init__(self, input_units, filter_size_channel):
        super().__init__()
        assert input_units % filter_size_channel == 0, 'output_size has to devisible by %d' % filter_size_channel
        out_units = input_units // filter_size_channel
        self.ConvLayers = torch.nn.ModuleList([torch.nn.Conv1d(input_units, out_units, 5, padding=2, groups=out_units)
                                               for _ in range(filter_size_channel)])


class _TimeConvResBlock(torch.nn.Module):          # parameter container, paule/models.py:114-128 (filter size 5, channelwise)
    def __init__(self, input_units):
        super().__init__()
        self.band_conv1d_1 = torch.nn.Conv1d(input_units, input_units, kernel_size=5, padding=2, groups=input_units)
        self.band_conv1d_2 = torch.nn.Conv1d(input_units, input_units, kernel_size=5, padding=2, groups=input_units)


class InverseModelMelTimeSmoothResidual(_HipModule):
    """mel -> CP inverse model (paule/models.py:177-247): mel-channel smoothing convolutions with residual connections,
    velocity / acceleration features, stacked LSTM, Linear, double_sequence, time-smoothing residual blocks and the
    channelwise weighting of smoothed and raw LSTM output.  Same constructor and state-dict keys as the reference; the
    default filter sizes (3 / 5) and Identity activations -- what Paule instantiates (paule/paule.py:146) -- are
    supported.  ``forward`` runs on the MI355X (pl_inverse_forward)."""

    def __init__(self, input_size=60, output_size=30, hidden_size=180, num_lstm_layers=4, mel_smooth_layers=3,
                 mel_smooth_filter_size=3, mel_resid_activation=torch.nn.Identity(), resid_blocks=5, time_filter_size=5,
                 pre_resid_activation=torch.nn.Identity(), post_resid_activation=torch.nn.Identity(),
                 output_activation=torch.nn.Identity(), lstm_resid=True):
        super().__init__()
        for act in (mel_resid_activation, pre_resid_activation, post_resid_activation, output_activation):
            if not isinstance(act, torch.nn.Identity):
                raise NotImplementedError("only Identity activations (the reference's defaults) run on the HIP path")
        if mel_smooth_filter_size != 3 or time_filter_size != 5:
            raise NotImplementedError("only mel_smooth_filter_size=3 / time_filter_size=5 (the defaults) run on the HIP path")
        if resid_blocks > 0 and not lstm_resid:
            raise NotImplementedError("lstm_resid=False is not supported")
        self.lstm_resid = lstm_resid
        self.MelBlocks = torch.nn.ModuleList([_MelChannelConv1D(input_size, mel_smooth_filter_size) for _ in range(mel_smooth_layers)])
        self.lstm = torch.nn.LSTM(3 * input_size, hidden_size, num_layers=num_lstm_layers, batch_first=True)
        self.post_linear = torch.nn.Linear(hidden_size, output_size)
        self.ResidualConvBlocks = torch.nn.ModuleList([_TimeConvResBlock(output_size) for _ in range(resid_blocks)])
        if self.lstm_resid and len(self.ResidualConvBlocks) > 0:
            self.resid_weighting = torch.nn.Conv1d(2 * output_size, output_size, time_filter_size, padding=2, groups=output_size)

    def _refresh(self, eng):
        eng.set_inverse_weights(self.state_dict())

    def forward(self, x, *args):
        _require_gpu(x, "InverseModelMelTimeSmoothResidual")
        B, Tp, M = x.shape
        key = (B, Tp, x.device.index, self.compute_dtype)
        eng = self._engine(key, lambda: HipPlanner(_stub_pred_sd(M, self.post_linear.out_features), None, batch=B,
                                                   n_frames=max(2 * Tp, 14), dtype=self.compute_dtype, device=x.device,
                                                   inv_model=self.state_dict()))
        return eng.inverse_forward(x, clip=False).to(x.dtype)
