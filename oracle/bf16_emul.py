"""bf16 rounding emulation of the planning iteration -- CPU oracle, test infrastructure (never imported by paule_amd/).

``oracle.manual`` restates the reference's arithmetic exactly (float64).  The HIP path in bf16 deviates from it ONLY by where it
rounds: weights and stored activations are bf16, everything is accumulated in f32 (or f64).  This module is ``oracle.manual``
with those roundings written in, at the places DESIGN.md section 3 lists, for the per-layer 32-row kernels (and the fused forward
launch, which is bit-identical to them):

  forward   CP master (f64) -> bf16 input; weights bf16, biases f32; gate pre-activations accumulated unrounded (first layer of a
            model: bias + W_hh h + W_ih x in one accumulator; further layers: the batched projection W_ih h_below + b is ROUNDED
            to bf16 first); running cell state unrounded; h_t handed on (and out) as bf16; gates and c stashed as bf16;
            mel head and pooling unrounded (f32 on the device), the embedder reads the pooled mel as bf16
  backward  dL/dsemvec, dL/dY as bf16; every batched product that feeds a recurrence rounded to bf16 (dL/dh of the layer below,
            dL/dh_last); the cell backward reads the bf16 STASH (gates, c_t, c_{t-1}), keeps dL/dc unrounded, rounds dA to bf16;
            the recurrent product is summed per slice of 32 hidden units (x 4 gates) and each partial is rounded to bf16 before the
            slices are added (reduce-scatter exchange, lstm_persist_rs.hip); input gradients (dL/dmel, dL/dCP) unrounded
  update    as oracle.manual (f64 smoothness terms and Adam)

What remains between this emulation and the device is f32-vs-f64 accumulation order and the fast activation forms (~1e-7
relative), which flip a bf16 rounding now and then -- so stashes agree bit for bit in almost every entry and gradients far more
tightly than against the exact oracle.  tests/test_hip_parity.py::test_bf16_path_equals_rounding_emulation.
"""
from __future__ import annotations

import numpy as np

from . import manual as mo
from .planner import MEL_WEIGHT, SEMANTIC_WEIGHT, OBJECTIVES


def q(x):
    """round to nearest-even bf16, returned as float64"""
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64).astype(np.float32))
    u = a.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    return r.astype(np.float64)


def f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32).astype(np.float64)


def lstm_layer_forward(x, w_ih, w_hh, bias, fused_input):
    """x (B,T,in) bf16 values; returns h (bf16 values), stash (bf16 values; 'c' rounded, the running state is not)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = np.zeros((B, H))
    c = np.zeros((B, H))
    hs = np.zeros((B, T, H))
    st = {k: np.zeros((B, T, H)) for k in "ifgoc"}
    gx = x @ w_ih.T + bias
    if not fused_input:
        gx = q(gx)                                         # the batched projection is stored as bf16
    for t in range(T):
        a = gx[:, t, :] + h @ w_hh.T
        i, f, g, o = (mo.sigmoid(a[:, 0:H]), mo.sigmoid(a[:, H:2 * H]), np.tanh(a[:, 2 * H:3 * H]), mo.sigmoid(a[:, 3 * H:4 * H]))
        c = f * c + i * g
        h = q(o * np.tanh(c))
        hs[:, t] = h
        st["i"][:, t], st["f"][:, t], st["g"][:, t], st["o"][:, t], st["c"][:, t] = q(i), q(f), q(g), q(o), q(c)
    return hs, st


def lstm_layer_backward(dh_ext, st, w_hh, slice_units=32):
    """dh_ext (B,T,H) bf16 values -> dA (B,T,4H) bf16 values."""
    B, T, H = dh_ext.shape
    dA = np.zeros((B, T, 4 * H))
    dh_rec = np.zeros((B, H))
    dc_next = np.zeros((B, H))
    n_slices = (H + slice_units - 1) // slice_units
    for t in range(T - 1, -1, -1):
        i, f, g, o, c = (st[k][:, t] for k in "ifgoc")
        c_prev = st["c"][:, t - 1] if t > 0 else np.zeros((B, H))
        tc = np.tanh(c)
        dh = dh_ext[:, t] + dh_rec
        dc = dc_next + dh * o * (1.0 - tc * tc)
        da = np.concatenate([dc * g * i * (1.0 - i), dc * c_prev * f * (1.0 - f), dc * i * (1.0 - g * g), dh * tc * o * (1.0 - o)], axis=1)
        da = q(da)
        dA[:, t] = da
        dc_next = dc * f
        dh_rec = np.zeros((B, H))
        for p in range(n_slices):                           # one bf16 partial tile per slice of the group's workgroups
            u0, u1 = p * slice_units, min(H, (p + 1) * slice_units)
            rows = np.concatenate([np.arange(gate * H + u0, gate * H + u1) for gate in range(4)])
            dh_rec = dh_rec + q(da[:, rows] @ w_hh[rows, :])
    return dA


class EmulModels:
    def __init__(self, pred_sd, emb_sd=None):
        self.p, self.e = mo._np(pred_sd), (mo._np(emb_sd) if emb_sd is not None else None)
        self.Lp = len([k for k in self.p if k.startswith("lstm.weight_hh_l")])
        self.Le = len([k for k in self.e if k.startswith("lstm.weight_hh_l")]) if self.e else 0

    @staticmethod
    def _layer(sd, l):
        bias = f32(f32(sd[f"lstm.bias_ih_l{l}"]) + f32(sd[f"lstm.bias_hh_l{l}"]))
        return q(sd[f"lstm.weight_ih_l{l}"]), q(sd[f"lstm.weight_hh_l{l}"]), bias

    def _stack_forward(self, sd, L, x):
        stashes, hs_all, inp = [], [], x
        for l in range(L):
            w_ih, w_hh, bias = self._layer(sd, l)
            inp, st = lstm_layer_forward(inp, w_ih, w_hh, bias, fused_input=(l == 0 and w_ih.shape[1] <= 64))
            stashes.append(st)
            hs_all.append(inp)
        return inp, stashes, hs_all

    def _stack_backward(self, sd, L, d, stashes):
        """d = dL/dh of the top layer (bf16 values); returns dL/d(input) unrounded and the dA of every layer."""
        dAs = [None] * L
        for l in range(L - 1, -1, -1):
            w_ih, w_hh, _ = self._layer(sd, l)
            dAs[l] = lstm_layer_backward(d, stashes[l], w_hh)
            d = dAs[l] @ w_ih
            if l > 0:
                d = q(d)                                    # dL/dh of the layer below feeds a recurrence: stored as bf16
        return d, dAs

    def pred_forward(self, x):
        h, stashes, hs = self._stack_forward(self.p, self.Lp, q(x))
        y = h @ q(self.p["post_linear.weight"]).T + f32(self.p["post_linear.bias"])
        Tp = y.shape[1] // 2
        mel = f32(0.5 * (f32(y[:, 0:2 * Tp:2]) + f32(y[:, 1:2 * Tp:2])))
        return mel, stashes, hs

    def pred_backward(self, dmel_half, stashes, T):
        """dmel_half = 0.5 * dL/dmel (f64): every pooled frame's gradient goes half to each of its two frames, as bf16."""
        B, Tp, M = dmel_half.shape
        dy = np.zeros((B, T, M))
        dy[:, 0:2 * Tp:2] = dmel_half
        dy[:, 1:2 * Tp:2] = dmel_half
        dy = q(f32(dy))
        d = q(dy @ q(self.p["post_linear.weight"]))
        dx, dAs = self._stack_backward(self.p, self.Lp, d, stashes)
        return f32(dx), dAs, dy

    def emb_forward(self, mel):
        h, stashes, hs = self._stack_forward(self.e, self.Le, q(mel))
        sem = f32(h[:, -1, :] @ q(self.e["linear_mapping.weight"]).T + f32(self.e["linear_mapping.bias"]))
        return sem, stashes, hs

    def emb_backward(self, dsem, stashes, Tp):
        B = dsem.shape[0]
        H = self.e["lstm.weight_hh_l0"].shape[1]
        d = np.zeros((B, Tp, H))
        d[:, -1, :] = q(q(f32(dsem)) @ q(self.e["linear_mapping.weight"]))
        dmel, dAs = self._stack_backward(self.e, self.Le, d, stashes)
        return f32(dmel), dAs


def loss_and_grad(models, objective, x, target_mel, target_semvec=None):
    """as oracle.manual.loss_and_grad (without the classifier term) on the rounding emulation; parts holds the device's buffers"""
    assert objective in OBJECTIVES
    B, T, _ = x.shape
    mel, st_p, hs_p = models.pred_forward(x)
    Tp = mel.shape[1]
    mel_l, dmel_rmse = mo.rmse_loss_grad(mel, target_mel, MEL_WEIGHT)
    vel_l, jerk_l, ll_l, g_smooth = mo.smoothness_loss_grad(x)
    dmel = np.zeros_like(mel)
    sem_l = np.zeros(B)
    parts = {"mel": mel, "pred_h": hs_p, "pred_stash": st_p}
    if objective in ("acoustic", "acoustic_semvec"):
        dmel += dmel_rmse
    if objective in ("acoustic_semvec", "semvec"):
        sem, st_e, hs_e = models.emb_forward(mel)
        sem_l, dsem = mo.rmse_loss_grad(sem, target_semvec, SEMANTIC_WEIGHT)
        dmel_e, dA_e = models.emb_backward(dsem, st_e, Tp)
        dmel += dmel_e
        parts.update(sem=sem, emb_h=hs_e, emb_stash=st_e, dmel_e=dmel_e, emb_dA=dA_e)
    g_model, dA_p, dy = models.pred_backward(0.5 * dmel, st_p, T)
    grad = g_model + g_smooth
    if objective == "acoustic":
        total = mel_l + vel_l + jerk_l + ll_l
    elif objective == "acoustic_semvec":
        total = mel_l + vel_l + jerk_l + sem_l + ll_l
    else:
        total = vel_l + jerk_l + sem_l + ll_l
    sub = np.stack([total, mel_l, sem_l, vel_l, jerk_l, ll_l, np.zeros(B), np.zeros(B)], axis=1)
    parts.update(dX=g_model, pred_dA=dA_p, dY=dy)
    return sub, grad, parts


class EmulPlanner:
    """ManualPlanner on the rounding emulation (acoustic path without the optional terms)."""

    def __init__(self, pred_sd, emb_sd=None, *, objective="acoustic_semvec", lr=0.01, betas=(0.9, 0.999), eps=1e-8, clamp=(-1.05, 1.05)):
        self.models = EmulModels(pred_sd, emb_sd)
        self.objective, self.lr, self.betas, self.eps, self.clamp = objective, lr, betas, eps, clamp
        self.x = self.m = self.v = None
        self.k = 0
        self.last_parts = None

    def set_targets(self, target_mel, target_semvec=None):
        self.target_mel = f32(target_mel)
        self.target_semvec = None if target_semvec is None else f32(target_semvec)

    def set_cp(self, cp):
        self.x = np.array(cp, dtype=np.float64)
        self.m, self.v, self.k = np.zeros_like(self.x), np.zeros_like(self.x), 0

    def step(self, n_iters=1):
        log = []
        for _ in range(n_iters):
            sub, grad, self.last_parts = loss_and_grad(self.models, self.objective, self.x, self.target_mel, self.target_semvec)
            log.append(sub)
            self.k += 1
            self.x, self.m, self.v = mo.adam_step(self.x, grad, self.m, self.v, self.k, self.lr, self.betas[0], self.betas[1], self.eps)
            self.x = mo.project(self.x, self.clamp)
        return np.stack(log)

    def get_cp(self):
        return self.x.copy()
