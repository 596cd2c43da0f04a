"""Adapter giving the CPU oracle the HipPlanner interface -- TEST INFRASTRUCTURE (host-logic tests on CPU,
gloo multi-process tests).  The product never imports this."""
import numpy as np
import torch

from oracle import planner as op


class OracleEngine:
    def __init__(self, pred_model, embedder=None, *, batch, n_frames, objective="acoustic", dtype="f32", lr=0.01,
                 betas=(0.9, 0.999), eps=1e-8, clamp=(-1.05, 1.05), smiling=False, weights=None, device=None,
                 use_graph=True, inv_model=None, tube_models=None):
        sd = lambda m: m.state_dict() if hasattr(m, "state_dict") else m
        self.pred_sd, self.emb_sd = sd(pred_model), sd(embedder) if embedder is not None else None
        self.B, self.T, self.Tp = batch, n_frames, n_frames // 2
        self.train_capacity = (batch + 15) // 16 * 16
        self.kw = dict(objective=objective, lr=lr, betas=betas, eps=eps, clamp=clamp, smiling=smiling)
        self.has_embedder = self.emb_sd is not None
        self.inv = op.inverse_model_from_state_dict(sd(inv_model)) if inv_model is not None else None
        self.has_inverse = self.inv is not None
        self.tube_sds = None if tube_models is None else [sd(m) for m in tube_models]
        self.has_tube = tube_models is not None
        self._build()

    def _build(self):
        pm = op.forward_model_from_state_dict(self.pred_sd)
        em = op.embedding_model_from_state_dict(self.emb_sd) if self.emb_sd is not None else None
        old = getattr(self, "p", None)
        tube = None
        if self.tube_sds is not None:
            tube = (op.forward_model_from_state_dict(self.tube_sds[0], apply_half_sequence=False),
                    op.forward_model_from_state_dict(self.tube_sds[1]), op.embedding_model_from_state_dict(self.tube_sds[2]))
        self.p = op.OraclePlanner(pm, em, tube_models=tube, **self.kw)
        if old is not None:
            self.p.xx, self.p.optimizer = old.xx, old.optimizer
            self.p.target_mel, self.p.target_semvec, self.p.past_cp = old.target_mel, old.target_semvec, old.past_cp
            self.p.classifier = old.classifier

    def get_tube_pred(self):
        return self.p.get_tube_pred()

    def embed_tube(self, tube):
        _, tube_mel, tube_emb = self.p.tube_models
        with torch.no_grad():
            t = torch.as_tensor(np.asarray(tube), dtype=torch.float64)
            return tube_mel(t), tube_emb(t, [torch.tensor(t.shape[1])] * t.shape[0])

    def inverse_forward(self, mel, clip=True):
        with torch.no_grad():
            cp = self.inv(torch.as_tensor(np.asarray(mel), dtype=torch.float64))
        return cp.clamp(-1.0, 1.0) if clip else cp

    # continued learning (OracleTrainer keeps its own model + optimizer; the planner is rebuilt on the new weights)
    def train_pred_step(self, cp, mel_target, lr=0.001, betas=(0.9, 0.999), eps=1e-8):
        self._ensure_trainer(lr, betas, eps)
        for grp in self.trainer.optimizer.param_groups:
            grp["lr"] = lr
        loss = self.trainer.train_pred_step(np.asarray(cp), np.asarray(mel_target))
        self.pred_sd = self.trainer.state_dict()
        self._build()
        return loss

    def train_model_step(self, model, inputs, target, lr=0.001, betas=(0.9, 0.999), eps=1e-8):
        if model == "pred":
            return self.train_pred_step(inputs, target, lr=lr, betas=betas, eps=eps)
        idx = {"cp_tube": 0, "tube_mel": 1}[model]
        trainers = self.__dict__.setdefault("tube_trainers", {})
        if model not in trainers:
            trainers[model] = op.OracleTrainer(op.forward_model_from_state_dict(self.tube_sds[idx], apply_half_sequence=(idx == 1)),
                                               lr=lr, betas=betas, eps=eps)
        for grp in trainers[model].optimizer.param_groups:
            grp["lr"] = lr
        loss = trainers[model].train_pred_step(np.asarray(inputs), np.asarray(target))
        self.tube_sds[idx] = trainers[model].state_dict()
        self._build()
        return loss

    def _ensure_trainer(self, lr=0.001, betas=(0.9, 0.999), eps=1e-8):
        if getattr(self, "trainer", None) is None:
            self.trainer = op.OracleTrainer(op.forward_model_from_state_dict(self.pred_sd), lr=lr, betas=betas, eps=eps)

    def get_pred_optimizer_state(self, lr=0.001, betas=(0.9, 0.999), eps=1e-8):
        self._ensure_trainer(lr, betas, eps)
        return self.trainer.optimizer.state_dict()

    def set_pred_optimizer_state(self, state_dict):
        self._ensure_trainer()
        if state_dict.get("state"):
            sd = self.trainer.optimizer.state_dict()
            sd["state"] = {i: {k: torch.as_tensor(v).double() if k != "step" else torch.as_tensor(float(v)) for k, v in st.items()}
                           for i, st in state_dict["state"].items()}
            self.trainer.optimizer.load_state_dict(sd)

    def get_weights(self, model="pred"):
        if model in ("cp_tube", "tube_mel", "tube_embedder"):
            return dict(self.tube_sds[("cp_tube", "tube_mel", "tube_embedder").index(model)])
        return dict(self.pred_sd if model == "pred" else self.emb_sd)

    def set_weights(self, pred_model=None, embedder=None):
        sd = lambda m: m.state_dict() if hasattr(m, "state_dict") else m
        if pred_model is not None:
            self.pred_sd = sd(pred_model)
        if embedder is not None:
            self.emb_sd = sd(embedder)
        self._build()

    def set_targets(self, target_mel, target_semvec=None):
        self.p.set_targets(np.asarray(target_mel), None if target_semvec is None else np.asarray(target_semvec))

    def set_cp(self, cp):
        self.p.set_cp(np.asarray(cp))

    def set_speech_classifier(self, classifier=None, weight=0.1):
        self.p.set_speech_classifier(classifier, weight)

    def set_past_cp(self, past):
        self.p.set_past_cp(past)

    def reset_optimizer(self):
        self.p.reset_optimizer()

    def step(self, n_iters=1, *, return_loss=True, return_grad=False):
        log = self.p.step(n_iters)
        return (log, self.p.last_grad) if return_grad else log

    def get_cp(self):
        return self.p.get_cp()

    def get_pred(self, with_semvec=None):
        mel, sem = self.p.get_pred()
        return mel, (sem if with_semvec in (None, True) else None)

    def embed_mel(self, mel, lens=None):
        mel = torch.as_tensor(np.asarray(mel)).to(self.p.dtype)
        lens = [torch.tensor(mel.shape[1])] * mel.shape[0] if lens is None else [torch.tensor(int(l)) for l in lens]
        with torch.no_grad():
            return self.p.embedder(mel, lens)
