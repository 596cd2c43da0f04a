"""numpy float64 restatement with EXPLICIT backward (no autograd) -- CPU oracle, test infrastructure.

This is the arithmetic the HIP kernels implement, written out step by step so that
every intermediate buffer of the device pipeline (gate stash, dA, dH_ext, per-term
gradients) has a CPU twin to be compared with:

* LSTM cell, gate order i, f, g, o, both biases added (torch.nn.LSTM as used at
  paule/models.py:344, :431; semantics listed in SURVEY.md 8 a-1)
* backward-DATA only: dL/dx through the cells, never dL/dW (paule/paule.py:1052 fills
  parameter grads that planning never reads)
* Linear + AvgPool1d(2,2) (paule/models.py:349-354), last-step gather + Linear (:441-446)
* RMSE (paule/util.py:564-572, eps = 0), 5-point stencil applied 1x/3x (:577-637),
  local_linear (:608-614), weights paule/paule.py:592-597
* Adam (torch.optim.Adam defaults, paule/paule.py:797) + clamp / smiling / past_cp
  (paule/paule.py:1201-1211)

It is checked against ``oracle.planner`` (autograd) in tests/test_oracle.py.
"""
from __future__ import annotations

import numpy as np

from .planner import (MEL_WEIGHT, VELOCITY_WEIGHT, JERK_WEIGHT, SEMANTIC_WEIGHT,
                      LOCAL_LINEAR_WEIGHT, OBJECTIVES)

# correlation taps: d[t] = sum_k taps[k] * x[t + k]
VEL_TAPS = np.array([1.0, -8.0, 0.0, 8.0, -1.0]) / 12.0               # paule/util.py:600
_ACC = np.convolve(VEL_TAPS, VEL_TAPS)
JERK_TAPS = np.convolve(_ACC, VEL_TAPS)                                # 13 taps (stencil applied 3x)
LL_TAPS = np.array([-1.0, 2.0, -1.0]) / 2.0                            # paule/util.py:614


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _np(sd):
    return {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float64)
            for k, v in sd.items()}


def lstm_layer_forward(x, w_ih, w_hh, b_ih, b_hh):
    """x (B,T,in) -> h (B,T,H) and the stash (i,f,g,o,c) each (B,T,H); h0 = c0 = 0."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = np.zeros((B, H))
    c = np.zeros((B, H))
    hs = np.zeros((B, T, H))
    st = {k: np.zeros((B, T, H)) for k in "ifgoc"}
    gx = x @ w_ih.T + (b_ih + b_hh)                    # batched input projection
    for t in range(T):
        a = gx[:, t, :] + h @ w_hh.T
        i, f, g, o = (sigmoid(a[:, 0:H]), sigmoid(a[:, H:2 * H]),
                      np.tanh(a[:, 2 * H:3 * H]), sigmoid(a[:, 3 * H:4 * H]))
        c = f * c + i * g
        h = o * np.tanh(c)
        hs[:, t] = h
        st["i"][:, t], st["f"][:, t], st["g"][:, t], st["o"][:, t], st["c"][:, t] = i, f, g, o, c
    return hs, st


def lstm_layer_backward(dh_ext, st, w_ih, w_hh):
    """dh_ext (B,T,H) = dL/dh_t from above -> dx (B,T,in), dA (B,T,4H)."""
    B, T, H = dh_ext.shape
    dA = np.zeros((B, T, 4 * H))
    dh_rec = np.zeros((B, H))
    dc_next = np.zeros((B, H))
    for t in range(T - 1, -1, -1):
        i, f, g, o, c = (st[k][:, t] for k in "ifgoc")
        c_prev = st["c"][:, t - 1] if t > 0 else np.zeros((B, H))
        tc = np.tanh(c)
        dh = dh_ext[:, t] + dh_rec
        dc = dc_next + dh * o * (1.0 - tc * tc)
        da_i = dc * g * i * (1.0 - i)
        da_f = dc * c_prev * f * (1.0 - f)
        da_g = dc * i * (1.0 - g * g)
        da_o = dh * tc * o * (1.0 - o)
        da = np.concatenate([da_i, da_f, da_g, da_o], axis=1)
        dA[:, t] = da
        dc_next = dc * f
        dh_rec = da @ w_hh
    dx = dA @ w_ih
    return dx, dA


def _corr_loss_grad(x, taps, weight):
    """loss_b = weight * mean_b(d^2), d[t] = sum_k taps[k] x[t+k]; returns (loss (B,), grad (B,T,C))."""
    B, T, C = x.shape
    n = T - len(taps) + 1
    d = np.zeros((B, n, C))
    for k, w in enumerate(taps):
        d += w * x[:, k:k + n, :]
    loss = weight * (d * d).reshape(B, -1).mean(axis=1)
    coef = weight * 2.0 / (n * C)
    grad = np.zeros_like(x)
    for k, w in enumerate(taps):
        grad[:, k:k + n, :] += coef * w * d
    return loss, grad


def smoothness_loss_grad(x):
    """velocity, jerk, local-linear terms (weighted) and their summed gradient."""
    vel_l, g_v = _corr_loss_grad(x, VEL_TAPS, VELOCITY_WEIGHT)
    jerk_l, g_j = _corr_loss_grad(x, JERK_TAPS, JERK_WEIGHT)
    ll_l, g_l = _corr_loss_grad(x, LL_TAPS, LOCAL_LINEAR_WEIGHT)
    return vel_l, jerk_l, ll_l, g_v + g_j + g_l


def rmse_loss_grad(yhat, y, weight):
    """loss_b = weight * sqrt(mean_b((yhat-y)^2)); grad = weight * (yhat-y) / (N * rmse)."""
    B = yhat.shape[0]
    d = (yhat - y).reshape(B, -1)
    n = d.shape[1]
    rmse = np.sqrt((d * d).mean(axis=1))
    grad = weight * d / (n * rmse[:, None])
    return weight * rmse, grad.reshape(yhat.shape)


class ManualModels:
    """Weights of ForwardModel / EmbeddingModel as numpy arrays (torch state-dict layout)."""

    def __init__(self, pred_sd, emb_sd=None):
        self.p = _np(pred_sd)
        self.e = _np(emb_sd) if emb_sd is not None else None
        self.Lp = len([k for k in self.p if k.startswith("lstm.weight_hh_l")])
        self.Le = len([k for k in self.e if k.startswith("lstm.weight_hh_l")]) if self.e else 0

    @staticmethod
    def _layer(sd, l):
        return (sd[f"lstm.weight_ih_l{l}"], sd[f"lstm.weight_hh_l{l}"],
                sd[f"lstm.bias_ih_l{l}"], sd[f"lstm.bias_hh_l{l}"])

    # ---- forward model ----
    def pred_forward(self, x):
        stashes, inp = [], x
        for l in range(self.Lp):
            inp, st = lstm_layer_forward(inp, *self._layer(self.p, l))
            stashes.append(st)
        y = inp @ self.p["post_linear.weight"].T + self.p["post_linear.bias"]
        Tp = y.shape[1] // 2
        mel = 0.5 * (y[:, 0:2 * Tp:2] + y[:, 1:2 * Tp:2])
        return mel, stashes

    def pred_backward(self, dmel, stashes, T):
        B, Tp, M = dmel.shape
        dy = np.zeros((B, T, M))
        dy[:, 0:2 * Tp:2] = 0.5 * dmel
        dy[:, 1:2 * Tp:2] = 0.5 * dmel
        d = dy @ self.p["post_linear.weight"]
        for l in range(self.Lp - 1, -1, -1):
            w_ih, w_hh, _, _ = self._layer(self.p, l)
            d, _ = lstm_layer_backward(d, stashes[l], w_ih, w_hh)
        return d

    # ---- embedder ----
    def emb_forward(self, mel):
        stashes, inp = [], mel
        for l in range(self.Le):
            inp, st = lstm_layer_forward(inp, *self._layer(self.e, l))
            stashes.append(st)
        v = inp[:, -1, :]                                # lens = T' (paule/paule.py:922-924)
        sem = v @ self.e["linear_mapping.weight"].T + self.e["linear_mapping.bias"]
        return sem, stashes

    def emb_backward(self, dsem, stashes, Tp):
        B = dsem.shape[0]
        H = self.e["lstm.weight_hh_l0"].shape[1]
        d = np.zeros((B, Tp, H))
        d[:, -1, :] = dsem @ self.e["linear_mapping.weight"]
        for l in range(self.Le - 1, -1, -1):
            w_ih, w_hh, _, _ = self._layer(self.e, l)
            d, _ = lstm_layer_backward(d, stashes[l], w_ih, w_hh)
        return d


def loss_and_grad(models, objective, x, target_mel, target_semvec=None, classifier=None):
    """One evaluation of the criterion and dL/dCP.  Returns (sub (B,8), grad (B,T,30), parts dict).
    classifier = (w (M,), b, weight) or None: + weight * softplus(mean_t(w . mel_t + b)) (paule/paule.py:604-622)."""
    assert objective in OBJECTIVES
    B, T, _ = x.shape
    mel, st_p = models.pred_forward(x)
    Tp = mel.shape[1]
    mel_l, dmel_rmse = rmse_loss_grad(mel, target_mel, MEL_WEIGHT)
    vel_l, jerk_l, ll_l, g_smooth = smoothness_loss_grad(x)
    dmel = np.zeros_like(mel)
    sem_l = np.zeros(B)
    parts = {"pred_mel": mel}
    if objective in ("acoustic", "acoustic_semvec"):
        dmel += dmel_rmse
    if objective in ("acoustic_semvec", "semvec"):
        sem, st_e = models.emb_forward(mel)
        sem_l, dsem = rmse_loss_grad(sem, target_semvec, SEMANTIC_WEIGHT)
        dmel_e = models.emb_backward(dsem, st_e, Tp)
        dmel += dmel_e
        parts.update(pred_semvec=sem, dmel_from_embedder=dmel_e)
    cls_l = np.zeros(B)
    if classifier is not None:
        w, b, weight = classifier
        z = (mel @ w + b).mean(axis=1)
        cls_l = weight * (np.maximum(z, 0.0) + np.log1p(np.exp(-np.abs(z))))
        dmel += (weight * sigmoid(z))[:, None, None] * w[None, None, :] / Tp
    g_model = models.pred_backward(dmel, st_p, T)
    grad = g_model + g_smooth
    if objective == "acoustic":
        total = mel_l + vel_l + jerk_l + ll_l
    elif objective == "acoustic_semvec":
        total = mel_l + vel_l + jerk_l + sem_l + ll_l
    else:
        total = vel_l + jerk_l + sem_l + ll_l
    total = total + cls_l
    sub = np.stack([total, mel_l, sem_l, vel_l, jerk_l, ll_l, cls_l, np.zeros(B)], axis=1)
    parts.update(grad_model=g_model, grad_smooth=g_smooth, dmel=dmel)
    return sub, grad, parts


def adam_step(x, g, m, v, k, lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam (no amsgrad / weight decay), step count k = 1, 2, ...  Returns (x, m, v)."""
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** k
    bc2 = 1.0 - beta2 ** k
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    x = x - (lr / bc1) * (m / denom)
    return x, m, v


def project(x, clamp=(-1.05, 1.05), smiling=False, past_cp=None):
    """paule/paule.py:1201-1211."""
    x = np.clip(x, clamp[0], clamp[1])
    if smiling:
        x[:, :, 4] = -1.0
        x[:, :, 1] = 1.0
    if past_cp is not None:
        x[:, 0:past_cp.shape[-2], :] = past_cp
    return x


class ManualPlanner:
    def __init__(self, pred_sd, emb_sd=None, *, objective="acoustic", lr=0.01, betas=(0.9, 0.999),
                 eps=1e-8, clamp=(-1.05, 1.05), smiling=False):
        self.models = ManualModels(pred_sd, emb_sd)
        self.objective, self.lr, self.betas, self.eps = objective, lr, betas, eps
        self.clamp, self.smiling = clamp, smiling
        self.x = self.m = self.v = None
        self.k = 0
        self.past_cp = None
        self.last_grad = None
        self.classifier = None

    def set_speech_classifier(self, classifier=None, weight=0.1):
        if classifier is None:
            self.classifier = None
            return
        sd = _np(classifier.state_dict() if hasattr(classifier, "state_dict") else classifier)
        self.classifier = (sd["linear.weight"].reshape(-1), float(sd["linear.bias"].reshape(-1)[0]), weight)

    def set_targets(self, target_mel, target_semvec=None):
        self.target_mel = np.asarray(target_mel, dtype=np.float64)
        self.target_semvec = None if target_semvec is None else np.asarray(target_semvec, dtype=np.float64)

    def set_cp(self, cp):
        self.x = np.array(cp, dtype=np.float64)
        if self.m is None:
            self.reset_optimizer()

    def set_past_cp(self, past_cp):
        self.past_cp = None if past_cp is None else np.asarray(past_cp, dtype=np.float64)

    def reset_optimizer(self):
        self.m = np.zeros_like(self.x)
        self.v = np.zeros_like(self.x)
        self.k = 0

    def step(self, n_iters=1):
        log = []
        for _ in range(n_iters):
            sub, grad, _ = loss_and_grad(self.models, self.objective, self.x, self.target_mel, self.target_semvec, self.classifier)
            self.last_grad = grad
            log.append(sub)
            self.k += 1
            self.x, self.m, self.v = adam_step(self.x, grad, self.m, self.v, self.k, self.lr,
                                               self.betas[0], self.betas[1], self.eps)
            self.x = project(self.x, self.clamp, self.smiling, self.past_cp)
        return np.stack(log)

    def get_cp(self):
        return self.x.copy()

    def get_pred(self):
        mel, _ = self.models.pred_forward(self.x)
        sem = None
        if self.models.e is not None:
            sem, _ = self.models.emb_forward(mel)
        return mel, sem
