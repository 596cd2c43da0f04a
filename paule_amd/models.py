"""``paule.models`` API surface for the planning path: ForwardModel and EmbeddingModel.

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
        if not apply_half_sequence:
            raise NotImplementedError("apply_half_sequence=False is not used on the planning path")
        self.apply_half_sequence = apply_half_sequence
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
        mel, _ = eng.get_pred(with_semvec=False)
        return mel.to(x.dtype)


class EmbeddingModel(_HipModule):
    """mel -> semantic vector: stacked LSTM, output at lens-1, Linear (paule/models.py:413-448;
    post_upsampling_size = 0 path, which is what Paule instantiates, paule/paule.py:167)."""

    def __init__(self, input_size=60, output_size=300, hidden_size=720, num_lstm_layers=1,
                 post_activation=torch.nn.LeakyReLU(), post_upsampling_size=0, dropout=0):
        super().__init__()
        if post_upsampling_size > 0:
            raise NotImplementedError("post_upsampling_size > 0 is not on the planning path (SURVEY 8 a-2)")
        if dropout:
            raise NotImplementedError("dropout > 0 is not on the planning path (Paule uses dropout 0)")
        self.post_upsampling_size = post_upsampling_size
        self.lstm = torch.nn.LSTM(input_size, hidden_size, num_layers=num_lstm_layers, batch_first=True, dropout=dropout)
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
