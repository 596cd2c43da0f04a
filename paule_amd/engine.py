"""Host-side driver of the MI355X planning engine (one handle <-> one GPU <-> one stream).

``HipPlanner`` owns a ``pl_handle`` of libpaule_hip.so and exposes the inner loop of
``Paule.plan_resynth`` (paule/paule.py:910-1211) as ``step(n_iters)``; PyTorch-ROCm is used
only to allocate device tensors and hand their addresses across the C-ABI.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi

DTYPES = {"f32": _capi.PL_F32, "fp32": _capi.PL_F32, "float32": _capi.PL_F32,
          "bf16": _capi.PL_BF16, "bfloat16": _capi.PL_BF16}


def _lstm_dims(sd):
    n_layers = len([k for k in sd if k.startswith("lstm.weight_hh_l")])
    return int(sd["lstm.weight_ih_l0"].shape[1]), int(sd["lstm.weight_hh_l0"].shape[1]), n_layers


def _state_dict(model_or_sd):
    if model_or_sd is None:
        return None
    if hasattr(model_or_sd, "state_dict"):
        return model_or_sd.state_dict()
    return model_or_sd


class HipPlanner:
    """Batched gradient planning of CP trajectories on one MI355X.

    pred_model / embedder: ``torch.nn.Module`` or state dict with the reference's key layout
    (``lstm.weight_ih_l{k}``, ``lstm.weight_hh_l{k}``, ``lstm.bias_ih_l{k}``, ``lstm.bias_hh_l{k}``,
    ``post_linear.*`` / ``linear_mapping.*``; paule/models.py:344-346, :431-437).
    """

    def __init__(self, pred_model, embedder=None, *, batch, n_frames, objective="acoustic", dtype="f32",
                 lr=0.01, betas=(0.9, 0.999), eps=1e-8, clamp=(-1.05, 1.05), smiling=False,
                 weights=None, device=None, use_graph=True, inv_model=None, tube_models=None):
        self.lib = _capi.load_library()          # raises HipLibraryError when the extension is missing
        if not torch.cuda.is_available():
            raise _capi.HipLibraryError("no HIP device visible: paule_amd runs on MI355X only (no CPU fallback)")
        if objective not in _capi.PL_OBJ:
            raise ValueError("objective has to be one of 'acoustic_semvec', 'acoustic' or 'semvec'")
        if dtype not in DTYPES:
            raise ValueError(f"dtype has to be one of {sorted(DTYPES)}")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise ValueError("HipPlanner needs a cuda (HIP) device")
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)
        pred_sd, emb_sd = _state_dict(pred_model), _state_dict(embedder)
        self.B, self.T, self.Tp = int(batch), int(n_frames), int(n_frames) // 2
        self.train_capacity = (self.B + 15) // 16 * 16   # rows of the handle: what a training mini-batch may use
        self.objective, self.dtype = objective, dtype
        in_p, hid_p, lay_p = _lstm_dims(pred_sd)
        self.C = in_p
        self.M = int(pred_sd["post_linear.weight"].shape[0])
        cfg = _capi.PlConfig()
        _capi.check(self.lib, self.lib.pl_default_config(C.byref(cfg)), "pl_default_config")
        cfg.batch, cfg.n_frames, cfg.cp_dim, cfg.mel_dim = self.B, self.T, self.C, self.M
        cfg.pred_layers, cfg.pred_hidden = lay_p, hid_p
        self.pred_hidden = int(hid_p)
        if emb_sd is not None:
            in_e, hid_e, lay_e = _lstm_dims(emb_sd)
            if in_e != self.M:
                raise ValueError("embedder input size does not match the predictive model's output size")
            # embedder variants (paule/models.py:362-409, :432-446): post_linear -> LeakyReLU -> output mapping, mel blocks in front
            self._emb_out = "upsampling" if "upsampling.weight" in emb_sd else "linear_mapping"
            self._emb_post = int(emb_sd["post_linear.weight"].shape[0]) if "post_linear.weight" in emb_sd else 0
            self._emb_blocks = len({k.split(".")[1] for k in emb_sd if k.startswith("MelBlocks.")})
            if self._emb_post == 0 and self._emb_out != "linear_mapping":
                raise ValueError("embedder state dict has an upsampling layer but no post_linear")
            for k, v in emb_sd.items():
                if k.startswith("MelBlocks.") and k.endswith("weight") and tuple(v.shape[1:]) != (3, 5):
                    raise NotImplementedError("only the default mel smoothing (filter size 3 over channels, 5 over time) is supported")
            self.S = int(emb_sd[self._emb_out + ".weight"].shape[0])
            cfg.emb_layers, cfg.emb_hidden, cfg.sem_dim = lay_e, hid_e, self.S
            cfg.emb_post_size, cfg.emb_mel_blocks = self._emb_post, self._emb_blocks
        else:
            self.S = 0
            cfg.emb_layers, cfg.emb_hidden = 0, 0
        inv_sd = _state_dict(inv_model)
        self.has_inverse = inv_sd is not None
        if inv_sd is not None:
            in_i, hid_i, lay_i = _lstm_dims(inv_sd)
            if in_i != 3 * self.M or int(inv_sd["post_linear.weight"].shape[0]) != self.C:
                raise ValueError("inverse model has to map mel_dim (x 3 with velocity / acceleration) to cp_dim")
            n_mel = len({k.split(".")[1] for k in inv_sd if k.startswith("MelBlocks.")})
            n_res = len({k.split(".")[1] for k in inv_sd if k.startswith("ResidualConvBlocks.")})
            if n_res > 0 and "resid_weighting.weight" not in inv_sd:
                raise NotImplementedError("lstm_resid=False with residual blocks is not supported (Paule uses the default True)")
            for k, v in inv_sd.items():
                if k.startswith(("MelBlocks.", "ResidualConvBlocks.", "resid_weighting.")) and k.endswith("weight") and v.shape[-1] != 5:
                    raise NotImplementedError("only the default filter sizes (mel 3 / time 5) are supported")
            cfg.inv_layers, cfg.inv_hidden, cfg.inv_mel_blocks, cfg.inv_res_blocks = lay_i, hid_i, n_mel, n_res
            self._inv_blocks = (n_mel, n_res)
        # somatosensory feedback (paule/paule.py:227-273): (cp_tube_model, tube_mel_model, tube_embedder)
        self.has_tube = tube_models is not None
        if self.has_tube:
            if emb_sd is None or objective == "acoustic":
                raise ValueError("somatosensory feedback needs an embedder and the objective 'acoustic_semvec' or 'semvec' (the "
                                 "reference's acoustic criterion fails there, paule/paule.py:692)")
            tube_sds = [_state_dict(m) for m in tube_models]
            if len(tube_sds) != 3:
                raise ValueError("tube_models has to be (cp_tube_model, tube_mel_model, tube_embedder)")
            for m in tube_models:
                if getattr(getattr(m, "lstm", None), "dropout", 0):
                    raise NotImplementedError("a tube model with dropout > 0 makes the planning loss random in the reference (it runs in "
                                              ".train() mode inside the loop, paule/paule.py:927); only dropout 0 is supported")
            (in_u, hid_u, lay_u), (in_m, hid_m, lay_m), (in_e2, hid_e2, lay_e2) = [_lstm_dims(sd) for sd in tube_sds]
            self.U = int(tube_sds[0]["post_linear.weight"].shape[0])
            if in_u != self.C or in_m != self.U or in_e2 != self.U or int(tube_sds[1]["post_linear.weight"].shape[0]) != self.M:
                raise ValueError("tube models have to map cp_dim -> tube_dim, tube_dim -> mel_dim and tube_dim -> sem_dim")
            if "linear_mapping.weight" not in tube_sds[2] or "post_linear.weight" in tube_sds[2] or \
                    int(tube_sds[2]["linear_mapping.weight"].shape[0]) != self.S:
                raise ValueError("the tube embedder has to be an EmbeddingModel(post_upsampling_size=0) with sem_dim outputs")
            cfg.tube_dim = self.U
            cfg.cp_tube_layers, cfg.cp_tube_hidden = lay_u, hid_u
            cfg.tube_mel_layers, cfg.tube_mel_hidden = lay_m, hid_m
            cfg.tube_emb_layers, cfg.tube_emb_hidden = lay_e2, hid_e2
        cfg.dtype, cfg.objective = DTYPES[dtype], _capi.PL_OBJ[objective]
        if weights:
            for k in ("w_mel", "w_sem", "w_vel", "w_jerk", "w_ll"):
                if k in weights:
                    setattr(cfg, k, float(weights[k]))
        cfg.lr, cfg.beta1, cfg.beta2, cfg.eps = float(lr), float(betas[0]), float(betas[1]), float(eps)
        cfg.clamp_lo, cfg.clamp_hi = float(clamp[0]), float(clamp[1])
        cfg.smiling, cfg.device, cfg.use_graph = int(bool(smiling)), dev_index, int(bool(use_graph))
        self._stream = torch.cuda.current_stream(self.device)
        cfg.stream = self._stream.cuda_stream
        self._h = C.c_void_p()
        _capi.check(self.lib, self.lib.pl_create(C.byref(cfg), C.byref(self._h)), "pl_create")
        self.has_embedder = emb_sd is not None
        self._dims = {"pred": (lay_p, hid_p, "post_linear"), "pred_in": in_p}
        if emb_sd is not None:
            self._dims.update({"embedder": (lay_e, hid_e, "post_linear" if self._emb_post else "linear_mapping"), "embedder_in": self.M})
        self.set_weights(pred_sd, emb_sd)
        if self.has_tube:
            self._dims.update({"cp_tube": (lay_u, hid_u, "post_linear"), "cp_tube_in": self.C,
                               "tube_mel": (lay_m, hid_m, "post_linear"), "tube_mel_in": self.U,
                               "tube_embedder": (lay_e2, hid_e2, "linear_mapping"), "tube_embedder_in": self.U})
            self.set_tube_weights(*tube_sds)
        if inv_sd is not None:
            self._dims.update({"inverse": (lay_i, hid_i, "post_linear"), "inverse_in": 3 * self.M})
            self.set_inverse_weights(inv_sd)

    # ---- plumbing ---------------------------------------------------------------------------
    def _dev(self, a, shape=None):
        t = torch.as_tensor(a) if not isinstance(a, torch.Tensor) else a
        t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    def _call(self, fn, *args):
        _capi.check(self.lib, fn(self._h, *args), fn.__name__ if hasattr(fn, "__name__") else "pl call")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.pl_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights (re-upload after continued learning, paule/paule.py:1372-1377) ----------------
    def set_weights(self, pred_model=None, embedder=None):
        emb_lin = "post_linear" if getattr(self, "_emb_post", 0) else "linear_mapping"
        for model_id, sd, lin in ((_capi.PL_MODEL_PRED, _state_dict(pred_model), "post_linear"),
                                  (_capi.PL_MODEL_EMBED, _state_dict(embedder), emb_lin)):
            if sd is None:
                continue
            _, _, n_layers = _lstm_dims(sd)
            for l in range(n_layers):
                ts = [self._dev(sd[f"lstm.{k}_l{l}"]) for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
                self._call(self.lib.pl_set_lstm_weights, model_id, l, *[t.data_ptr() for t in ts])
            w, b = self._dev(sd[f"{lin}.weight"]), self._dev(sd[f"{lin}.bias"])
            self._call(self.lib.pl_set_linear, model_id, w.data_ptr(), b.data_ptr())
            if model_id == _capi.PL_MODEL_EMBED:
                if self._emb_post:
                    w, b = self._dev(sd[self._emb_out + ".weight"], (self.S, self._emb_post)), self._dev(sd[self._emb_out + ".bias"])
                    self._call(self.lib.pl_set_embedder_output, w.data_ptr(), b.data_ptr())
                for i in range(self._emb_blocks):
                    for j in range(3):
                        key = f"MelBlocks.{i}.ConvLayers.{j}"
                        w, b = self._dev(sd[key + ".weight"]), self._dev(sd[key + ".bias"])
                        self._call(self.lib.pl_set_embedder_conv, i, j, w.data_ptr(), b.data_ptr())

    def set_tube_weights(self, cp_tube_model=None, tube_mel_model=None, tube_embedder=None):
        for model_id, sd, lin in ((_capi.PL_MODEL_CP_TUBE, _state_dict(cp_tube_model), "post_linear"),
                                  (_capi.PL_MODEL_TUBE_MEL, _state_dict(tube_mel_model), "post_linear"),
                                  (_capi.PL_MODEL_TUBE_EMBED, _state_dict(tube_embedder), "linear_mapping")):
            if sd is None:
                continue
            _, _, n_layers = _lstm_dims(sd)
            for l in range(n_layers):
                ts = [self._dev(sd[f"lstm.{k}_l{l}"]) for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
                self._call(self.lib.pl_set_lstm_weights, model_id, l, *[t.data_ptr() for t in ts])
            w, b = self._dev(sd[f"{lin}.weight"]), self._dev(sd[f"{lin}.bias"])
            self._call(self.lib.pl_set_linear, model_id, w.data_ptr(), b.data_ptr())

    def get_tube_pred(self):
        """pred_tube (B, T, tube_dim), pred_tube_mel (B, T/2, mel_dim), pred_tube_semvec (B, sem_dim) at the current CP
        (paule/paule.py:916-919, :926-929)."""
        if not self.has_tube:
            raise ValueError("this engine was built without tube_models=")
        tube = torch.empty((self.B, self.T, self.U), dtype=torch.float32, device=self.device)
        mel = torch.empty((self.B, self.Tp, self.M), dtype=torch.float32, device=self.device)
        sem = torch.empty((self.B, self.S), dtype=torch.float32, device=self.device)
        self._call(self.lib.pl_get_tube_pred, tube.data_ptr(), mel.data_ptr(), sem.data_ptr())
        return tube, mel, sem

    def embed_tube(self, tube):
        """``tube_mel_model(tube)``, ``tube_embedder(tube, T)`` for a tube (B, T, tube_dim), e.g. the one extracted from the
        synthesis (paule/paule.py:1084, :1147-1150)."""
        if not self.has_tube:
            raise ValueError("this engine was built without tube_models=")
        t = self._dev(tube, (self.B, self.T, self.U))
        mel = torch.empty((self.B, self.Tp, self.M), dtype=torch.float32, device=self.device)
        sem = torch.empty((self.B, self.S), dtype=torch.float32, device=self.device)
        self._call(self.lib.pl_embed_tube, t.data_ptr(), mel.data_ptr(), sem.data_ptr())
        return mel, sem

    # ---- inverse model: initial CP from the target mel (paule/paule.py:550-556) ------------------
    def set_inverse_weights(self, inv_model):
        sd = _state_dict(inv_model)
        _, _, n_layers = _lstm_dims(sd)
        for l in range(n_layers):
            ts = [self._dev(sd[f"lstm.{k}_l{l}"]) for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
            self._call(self.lib.pl_set_lstm_weights, _capi.PL_MODEL_INVERSE, l, *[t.data_ptr() for t in ts])
        w, b = self._dev(sd["post_linear.weight"]), self._dev(sd["post_linear.bias"])
        self._call(self.lib.pl_set_linear, _capi.PL_MODEL_INVERSE, w.data_ptr(), b.data_ptr())
        n_mel, n_res = self._inv_blocks
        convs = [(_capi.PL_CONV_MEL, i, j, f"MelBlocks.{i}.ConvLayers.{j}") for i in range(n_mel) for j in range(3)]
        convs += [(_capi.PL_CONV_RES, i, j, f"ResidualConvBlocks.{i}.band_conv1d_{j + 1}") for i in range(n_res) for j in range(2)]
        if n_res > 0:
            convs.append((_capi.PL_CONV_RW, 0, 0, "resid_weighting"))
        for kind, blk, idx, key in convs:
            w, b = self._dev(sd[key + ".weight"]), self._dev(sd[key + ".bias"])
            self._call(self.lib.pl_set_inverse_conv, kind, blk, idx, w.data_ptr(), b.data_ptr())

    def inverse_forward(self, mel, clip=True):
        """``inv_model(target_mel)`` (+ ``.clip(-1, 1)``, paule/paule.py:552-555): mel (B, n, mel_dim), n <= T/2 -> (B, 2n, cp_dim)."""
        if not self.has_inverse:
            raise ValueError("this engine was built without inv_model=")
        mel = self._dev(mel)
        if mel.dim() == 2:
            mel = mel.unsqueeze(0)
        n = int(mel.shape[1])
        mel = self._dev(mel, (self.B, n, self.M))
        out = torch.empty((self.B, 2 * n, self.C), dtype=torch.float32, device=self.device)
        self._call(self.lib.pl_inverse_forward, mel.data_ptr(), n, out.data_ptr(), int(bool(clip)))
        self._keep_inv = mel
        return out

    def get_weights(self, model="pred"):
        """Current parameters of ``"pred"`` / ``"embedder"`` as a torch state dict (float32, on the engine's device):
        after ``train_pred_step`` this is what ``pred_model.load_state_dict`` needs (the reference mutates the module in
        place, paule/paule.py:1372-1377)."""
        pred = model == "pred"
        if model == "embedder" and not self.has_embedder:
            raise ValueError("this engine has no embedder")
        ids = {"pred": _capi.PL_MODEL_PRED, "embedder": _capi.PL_MODEL_EMBED, "cp_tube": _capi.PL_MODEL_CP_TUBE,
               "tube_mel": _capi.PL_MODEL_TUBE_MEL, "tube_embedder": _capi.PL_MODEL_TUBE_EMBED}
        if model not in ids or model not in self._dims:
            raise ValueError("model has to be 'pred', 'embedder' or (with tube_models=) 'cp_tube', 'tube_mel', 'tube_embedder'")
        model_id = ids[model]
        n_layers, H, lin = self._dims[model]
        sd = {}
        for l in range(n_layers):
            in_l = self._dims[model + "_in"] if l == 0 else H
            ts = [torch.empty(shape, dtype=torch.float32, device=self.device)
                  for shape in ((4 * H, in_l), (4 * H, H), (4 * H,), (4 * H,))]
            self._call(self.lib.pl_get_lstm_weights, model_id, l, *[t.data_ptr() for t in ts])
            for k, t in zip(("weight_ih", "weight_hh", "bias_ih", "bias_hh"), ts):
                sd[f"lstm.{k}_l{l}"] = t
        out = {"pred": self.M, "cp_tube": getattr(self, "U", 0), "tube_mel": self.M, "tube_embedder": self.S}.get(
            model, getattr(self, "_emb_post", 0) or self.S)   # the linear that reads the LSTM output
        w = torch.empty((out, H), dtype=torch.float32, device=self.device)
        b = torch.empty((out,), dtype=torch.float32, device=self.device)
        self._call(self.lib.pl_get_linear, model_id, w.data_ptr(), b.data_ptr())
        sd[f"{lin}.weight"], sd[f"{lin}.bias"] = w, b
        return sd

    # ---- continued learning of the predictive model (paule/paule.py:1353-1379) ------------------
    def train_pred_step(self, cp, mel_target, lr=0.001, betas=(0.9, 0.999), eps=1e-8):
        """One ``pred_optimizer`` step on a mini-batch: ``cp`` (n, t, cp_dim) -> ``mel_target`` (n, t/2, mel_dim), n <= batch,
        t <= n_frames of the engine;
        RMSE over the whole batch, torch.optim.Adam on the parameters (paule/paule.py:287-288, :1372-1377).  Returns the
        loss of this step as a 0-d device tensor (no host sync)."""
        cp = self._dev(cp)
        if cp.dim() == 2:
            cp = cp.unsqueeze(0)
        n = int(cp.shape[0])
        if not 1 <= n <= self.train_capacity:
            raise ValueError(f"mini-batch of {n} samples does not fit an engine built for batch {self.B} "
                             f"({self.train_capacity} rows with padding)")
        t = int(cp.shape[1])
        if not 2 <= t <= self.T:
            raise ValueError(f"samples of {t} frames do not fit an engine built for {self.T} frames")
        cp = self._dev(cp, (n, t, self.C))
        mel = self._dev(mel_target)
        if mel.dim() == 2:
            mel = mel.unsqueeze(0)
        mel = self._dev(mel, (n, t // 2, self.M))
        loss = torch.empty((), dtype=torch.float32, device=self.device)
        self._call(self.lib.pl_train_pred_step, n, t, cp.data_ptr(), mel.data_ptr(), C.c_float(lr), C.c_float(betas[0]),
                   C.c_float(betas[1]), C.c_float(eps), loss.data_ptr())
        self._keep = (cp, mel)   # the launches are asynchronous: keep the operands alive until the next call
        return loss

    _TRAINABLE = {"pred": _capi.PL_MODEL_PRED, "cp_tube": _capi.PL_MODEL_CP_TUBE, "tube_mel": _capi.PL_MODEL_TUBE_MEL}

    def train_model_step(self, model, inputs, target, lr=0.001, betas=(0.9, 0.999), eps=1e-8):
        """One Adam step of ``model`` in ("pred", "cp_tube", "tube_mel") on a mini-batch (paule/paule.py:1372-1377, :1386-1404):
        pred: cp (n, t, cp_dim) -> mel (n, t/2, mel_dim); cp_tube: cp -> tube (n, t, tube_dim); tube_mel: tube -> mel (n, t/2,
        mel_dim).  RMSE over the whole batch; returns the loss as a 0-d device tensor."""
        if model not in self._TRAINABLE or (model != "pred" and not self.has_tube):
            raise ValueError("model has to be 'pred' or (with tube_models=) 'cp_tube' / 'tube_mel'")
        x = self._dev(inputs)
        if x.dim() == 2:
            x = x.unsqueeze(0)
        n, t = int(x.shape[0]), int(x.shape[1])
        if not 1 <= n <= self.train_capacity:
            raise ValueError(f"mini-batch of {n} samples does not fit an engine built for batch {self.B} "
                             f"({self.train_capacity} rows with padding)")
        if not 2 <= t <= self.T:
            raise ValueError(f"samples of {t} frames do not fit an engine built for {self.T} frames")
        in_dim = self.U if model == "tube_mel" else self.C
        out_shape = (n, t, self.U) if model == "cp_tube" else (n, t // 2, self.M)
        x = self._dev(x, (n, t, in_dim))
        y = self._dev(target)
        if y.dim() == 2:
            y = y.unsqueeze(0)
        y = self._dev(y, out_shape)
        loss = torch.empty((), dtype=torch.float32, device=self.device)
        self._call(self.lib.pl_train_model_step, self._TRAINABLE[model], n, t, x.data_ptr(), y.data_ptr(), C.c_float(lr),
                   C.c_float(betas[0]), C.c_float(betas[1]), C.c_float(eps), loss.data_ptr())
        self._keep = (x, y)   # the launches are asynchronous: keep the operands alive until the next call
        return loss

    def reset_pred_optimizer(self):
        self._call(self.lib.pl_reset_pred_optimizer)

    def _pred_param_shapes(self, model="pred"):
        """(name, shape) in the order of ``ForwardModel.parameters()`` (= the parameter indices of torch.optim.Adam's state)."""
        n_layers, H, _ = self._dims[model]
        out = []
        for l in range(n_layers):
            in_l = self._dims[model + "_in"] if l == 0 else H
            out += [(f"lstm.weight_ih_l{l}", (4 * H, in_l)), (f"lstm.weight_hh_l{l}", (4 * H, H)),
                    (f"lstm.bias_ih_l{l}", (4 * H,)), (f"lstm.bias_hh_l{l}", (4 * H,))]
        n_out = self.U if model == "cp_tube" else self.M
        return out + [("post_linear.weight", (n_out, H)), ("post_linear.bias", (n_out,))]

    def get_pred_optimizer_state(self, lr=0.001, betas=(0.9, 0.999), eps=1e-8):
        """The parameter optimiser's state in the layout of ``torch.optim.Adam.state_dict()`` (what the reference's users save,
        docs/examples/minimal_example.py:51): step / exp_avg / exp_avg_sq per parameter, in ``parameters()`` order."""
        return self.get_optimizer_state("pred", lr=lr, betas=betas, eps=eps)

    def set_pred_optimizer_state(self, state_dict):
        """Loads a ``torch.optim.Adam.state_dict()`` of the predictive model's optimiser (an empty state = a fresh optimiser)."""
        self.set_optimizer_state("pred", state_dict)

    def _trainable_id(self, model):
        if model not in self._TRAINABLE or (model != "pred" and not self.has_tube):
            raise ValueError("model has to be 'pred' or (with tube_models=) 'cp_tube' / 'tube_mel'")
        return self._TRAINABLE[model]

    def get_optimizer_state(self, model="pred", lr=0.001, betas=(0.9, 0.999), eps=1e-8):
        """Adam state of ``model`` in ("pred", "cp_tube", "tube_mel") as a ``torch.optim.Adam.state_dict()`` (pred_optimizer,
        tube_optimizer, tube_mel_optimizer of the reference, paule/paule.py:284-306)."""
        mid = self._trainable_id(model)
        step = int(self.lib.pl_get_model_optimizer_step(self._h, mid))
        shapes = self._pred_param_shapes(model)
        bufs = {w: [torch.empty(shape, dtype=torch.float32, device=self.device) for _, shape in shapes] for w in (1, 2)}
        n_layers = self._dims[model][0]
        for which in (1, 2):
            for l in range(n_layers):
                self._call(self.lib.pl_get_model_optimizer_state, mid, l, which, *[t.data_ptr() for t in bufs[which][4 * l:4 * l + 4]])
            w, b = bufs[which][-2:]
            self._call(self.lib.pl_get_model_optimizer_state, mid, -1, which, w.data_ptr(), None, b.data_ptr(), None)
        state = {} if step == 0 else {i: {"step": torch.tensor(float(step)), "exp_avg": bufs[1][i], "exp_avg_sq": bufs[2][i]}
                                      for i in range(len(shapes))}
        group = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False,
                     differentiable=False, fused=None, params=list(range(len(shapes))))
        return {"state": state, "param_groups": [group]}

    def set_optimizer_state(self, model, state_dict):
        mid = self._trainable_id(model)
        state = state_dict.get("state", {})
        shapes = self._pred_param_shapes(model)
        if not state:
            self._call(self.lib.pl_reset_model_optimizer, mid)
            return
        if sorted(state) != list(range(len(shapes))):
            raise ValueError(f"optimizer state does not match the parameters of '{model}'")
        n_layers = self._dims[model][0]
        for which, key in ((1, "exp_avg"), (2, "exp_avg_sq")):
            ts = [self._dev(state[i][key], shape) for i, (_, shape) in enumerate(shapes)]
            for l in range(n_layers):
                self._call(self.lib.pl_set_model_optimizer_state, mid, l, which, *[t.data_ptr() for t in ts[4 * l:4 * l + 4]])
            self._call(self.lib.pl_set_model_optimizer_state, mid, -1, which, ts[-2].data_ptr(), None, ts[-1].data_ptr(), None)
        self._call(self.lib.pl_set_model_optimizer_step, mid, int(round(float(state[0]["step"]))))

    def set_speech_classifier(self, classifier=None, weight=0.1):
        """LinearClassifier(mel_dim -> 1) term (paule/models.py:887-910; ``use_speech_classifier=True``): module or state
        dict with ``linear.weight`` [1, mel_dim] and ``linear.bias`` [1]; ``None`` switches the term off."""
        sd = _state_dict(classifier)
        if sd is None:
            self._call(self.lib.pl_set_speech_classifier, None, None, C.c_float(weight))
            return
        w = self._dev(sd["linear.weight"]).reshape(-1)
        b = self._dev(sd["linear.bias"]).reshape(-1)
        if w.numel() != self.M or b.numel() != 1:
            raise ValueError("speech classifier has to be Linear(mel_dim -> 1)")
        self._call(self.lib.pl_set_speech_classifier, w.data_ptr(), b.data_ptr(), C.c_float(weight))

    # ---- state --------------------------------------------------------------------------------
    def set_targets(self, target_mel, target_semvec=None):
        tm = self._dev(target_mel, (self.B, self.Tp, self.M))
        ts = self._dev(target_semvec, (self.B, self.S)) if target_semvec is not None else None
        self._call(self.lib.pl_set_targets, tm.data_ptr(), ts.data_ptr() if ts is not None else None)

    def set_cp(self, cp):
        t = self._dev(cp, (self.B, self.T, self.C))
        self._call(self.lib.pl_set_cp, t.data_ptr())

    def set_past_cp(self, past_cp):
        if past_cp is None:
            self._call(self.lib.pl_set_past_cp, None, 0, 0)
            return
        t = self._dev(past_cp)
        per_utt = 1 if t.dim() == 3 else 0
        if per_utt and t.shape[0] != self.B:
            raise ValueError("per-utterance past_cp needs shape (B, P, cp_dim)")
        self._call(self.lib.pl_set_past_cp, t.data_ptr(), int(t.shape[-2]), per_utt)

    def reset_optimizer(self):
        self._call(self.lib.pl_reset_optimizer)

    # ---- the hot path -------------------------------------------------------------------------
    def step(self, n_iters=1, *, return_loss=True, return_grad=False):
        """n inner iterations.  Returns loss_log (n_iters, B, 8) [total, mel, semvec, vel, jerk, local_linear,
        speech_classifier, 0] at the pre-step CP (device tensor, stream-asynchronous) and optionally the last gradient."""
        loss = torch.empty((n_iters, self.B, _capi.PL_LOSS_COLS), dtype=torch.float32, device=self.device) \
            if return_loss else None
        grad = torch.empty((self.B, self.T, self.C), dtype=torch.float32, device=self.device) if return_grad else None
        self._call(self.lib.pl_step, int(n_iters), loss.data_ptr() if loss is not None else None,
                   grad.data_ptr() if grad is not None else None)
        if return_grad:
            return loss, grad
        return loss

    def check(self):
        """Raises if a device-side bounded wait of a persistent sweep timed out (the results would be garbage); synchronises."""
        self.synchronize()

    def get_cp(self):
        out = torch.empty((self.B, self.T, self.C), dtype=torch.float32, device=self.device)
        self._call(self.lib.pl_get_cp, out.data_ptr())
        return out

    def get_pred(self, with_semvec=None):
        if with_semvec is None:
            with_semvec = self.has_embedder
        mel = torch.empty((self.B, self.Tp, self.M), dtype=torch.float32, device=self.device)
        sem = torch.empty((self.B, self.S), dtype=torch.float32, device=self.device) if with_semvec else None
        self._call(self.lib.pl_get_pred, mel.data_ptr(), sem.data_ptr() if sem is not None else None)
        return mel, sem

    def get_pred_frames(self):
        """post_linear(lstm(cp)) of every frame, (B, T, mel_dim): the predictive model without the half sequence
        (``ForwardModel(apply_half_sequence=False)``, paule/models.py:348-356)."""
        y = torch.empty((self.B, self.T, self.M), dtype=torch.float32, device=self.device)
        self._call(self.lib.pl_get_pred_frames, y.data_ptr())
        return y

    def embed_mel(self, mel, lens=None):
        m = self._dev(mel, (self.B, self.Tp, self.M))
        ln = None
        if lens is not None:
            ln = torch.as_tensor([int(x) for x in lens], dtype=torch.int32, device=self.device)
        out = torch.empty((self.B, self.S), dtype=torch.float32, device=self.device)
        self._call(self.lib.pl_embed_mel, m.data_ptr(), ln.data_ptr() if ln is not None else None, out.data_ptr())
        return out

    # ---- introspection ------------------------------------------------------------------------
    def debug_read(self, name):
        n = C.c_int64(0)
        self._call(self.lib.pl_debug_read, name.encode(), None, 0, C.byref(n))
        out = torch.empty((n.value,), dtype=torch.float32, device=self.device)
        self._call(self.lib.pl_debug_read, name.encode(), out.data_ptr(), n.value, C.byref(n))
        return out

    def bench_kernel(self, kernel="bwd", model="pred", reps=300):
        """(avg ms per launch, algorithmic FLOPs per launch) of one LSTM kernel -- "fwd" / "bwd" = launch-per-step
        kernels, "fwd_sweep" / "bwd_sweep" = persistent sweeps, "fused_fwd" / "fused_bwd" = the fused acoustic launches -- timed with hipEvents
        on the engine's stream."""
        ms, fl = C.c_float(0), C.c_double(0)
        kid = {"fwd": 0, "bwd": 1, "fwd_sweep": 2, "bwd_sweep": 3, "fused_fwd": 4, "fused_bwd": 5}[kernel]
        self._call(self.lib.pl_bench_kernel, kid,
                   _capi.PL_MODEL_PRED if model == "pred" else _capi.PL_MODEL_EMBED, int(reps), C.byref(ms), C.byref(fl))
        return ms.value, fl.value

    PLAN_FIELDS = ("fused_fwd", "fused_bwd", "fwd_chains_pred", "fwd_chains_emb", "bwd_chains_pred", "bwd_chains_emb",
                   "fwd_workgroups", "bwd_workgroups", "bwd_waves", "n_cu", "retained_execs", "fused_rows", "fwd_per_cu", "bwd_prefetchers")

    def plan_info(self):
        """The launch schedule the library planned for this handle (include/paule_hip.h: pl_plan_info) as a dict:
        fused_fwd / fused_bwd 0 or 1 (the role-fused launches of lstm_fused.hip), their chain counts and workgroups, ...,
        fused_rows 32 / 16 (batch rows of their LSTM roles: 16 for batches of up to 16 rows, lstm_fused16.h) or 0, fwd_per_cu 1 / 2
        (workgroups per CU the forward launch is written for: 2 = lstm_fused2.hip) or 0, bwd_prefetchers = prefetcher workgroups beside the
        predictor's streamed backward sweep (round 5) or 0."""
        buf = (C.c_int32 * len(self.PLAN_FIELDS))()
        self._call(self.lib.pl_plan_info, buf, len(self.PLAN_FIELDS))
        return dict(zip(self.PLAN_FIELDS, (int(v) for v in buf)))

    @property
    def device_bytes(self):
        return int(self.lib.pl_device_bytes(self._h))

    @property
    def flops_per_iteration(self):
        return float(self.lib.pl_flops_per_iteration(self._h))

    def synchronize(self):
        """Waits for the engine's stream; raises if a device-side bounded wait timed out."""
        self._call(self.lib.pl_synchronize)
