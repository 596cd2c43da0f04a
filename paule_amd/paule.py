"""``paule.Paule`` API surface for the gradient-planning path, driving the MI355X engine.

Keeps the reference's keyword-only signatures (paule/paule.py:101-107, :391-414), argument
validation (the ``ValueError``s pinned by tests/test_paule.py:31-62) and the 33-field
``PlanningResults`` (paule/paule.py:57), and routes the inner loop (paule/paule.py:910-1211) to
``HipPlanner.step``.  Everything the reference does around the loop with VocalTractLab / librosa
(synthesis, mel extraction; host code, out of scope here) is injectable:

* ``synthesizer(cp_normalised (T, 30) ndarray) -> (sig, sr)``  -- stands in for
  ``speak(inv_normalize_cp(cp))`` (paule/paule.py:859, :1097);
* ``mel_extractor(sig, sr) -> (T', 60) ndarray``               -- stands in for
  ``normalize_mel_librosa(librosa_melspec(sig, sr))`` (paule/paule.py:861-862, :1102-1103);

when they are ``None`` the produced-* result fields stay ``None`` / empty.  Extensions that the
reference does not have (it always plans one utterance, paule/paule.py:585-588): ``target_acoustic``
may be a ``(B, T', 60)`` mel array and ``initial_cp`` a ``(B, T, 30)`` array; B independent
utterances are then planned together (per-utterance losses).
"""
from __future__ import annotations

import random
import time
import warnings
from collections import namedtuple

import numpy as np
import torch

# Set seed (paule/paule.py:37-39)
torch.manual_seed(20200905)
random.seed(20200905)

PlanningResults = namedtuple(
    "PlanningResults",
    "planned_cp, initial_cp, initial_sig, initial_sr, initial_prod_mel,initial_pred_mel, target_sig, target_sr, "
    "target_mel, prod_sig, prod_sr, prod_mel, pred_mel, initial_prod_semvec, initial_pred_semvec, prod_semvec, "
    "pred_semvec, prod_loss_steps, planned_loss_steps, planned_mel_loss_steps, vel_loss_steps, jerk_loss_steps, "
    "pred_semvec_loss_steps, prod_semvec_loss_steps, cp_steps, pred_semvec_steps, prod_semvec_steps, grad_steps, "
    "sig_steps, prod_mel_steps, pred_mel_steps, pred_model_loss, inv_model_loss")

PlanningResultsWithSpeechClassifier = namedtuple(
    "PlanningResultsWithSpeechClassifier",
    "planned_cp, initial_cp, initial_sig, initial_sr, initial_prod_mel, initial_pred_mel, target_sig, target_sr, "
    "target_mel, prod_sig, prod_sr, prod_mel, pred_mel, initial_prod_semvec, initial_pred_semvec, prod_semvec, "
    "pred_semvec, prod_loss_steps, planned_loss_steps, planned_mel_loss_steps, vel_loss_steps, jerk_loss_steps, "
    "pred_semvec_loss_steps, prod_semvec_loss_steps, pred_speech_classifier_loss_steps, "
    "prod_speech_classifier_loss_steps, cp_steps, pred_semvec_steps, prod_semvec_steps, grad_steps, sig_steps, "
    "prod_mel_steps, pred_mel_steps, pred_model_loss, inv_model_loss")   # paule/paule.py:58

PlanningResultsWithSomatosensory = namedtuple(
    "PlanningResultsWithSomatosensory",
    "planned_cp, initial_cp, initial_sig, initial_sr, initial_prod_mel,initial_pred_mel, initial_prod_tube, initial_pred_tube, "
    "initial_prod_tube_mel, initial_pred_tube_mel, target_sig, target_sr, target_mel, prod_sig, prod_sr, prod_mel, pred_mel, "
    "prod_tube, pred_tube, prod_tube_mel, pred_tube_mel, initial_prod_semvec, initial_pred_semvec, initial_prod_tube_semvec, "
    "initial_pred_tube_semvec, prod_semvec, pred_semvec, prod_tube_semvec, pred_tube_semvec, prod_loss_steps, planned_loss_steps, "
    "planned_mel_loss_steps, vel_loss_steps, jerk_loss_steps, pred_semvec_loss_steps, prod_semvec_loss_steps, prod_tube_loss_steps, "
    "pred_tube_mel_loss_steps,prod_tube_mel_loss_steps, pred_tube_semvec_loss_steps, prod_tube_semvec_loss_steps, cp_steps, "
    "pred_semvec_steps, prod_semvec_steps, grad_steps, sig_steps, prod_mel_steps, pred_mel_steps, prod_tube_steps, pred_tube_steps, "
    "prod_tube_mel_steps, pred_tube_mel_steps, prod_tube_semvec_steps, pred_tube_semvec_steps, pred_model_loss, inv_model_loss, "
    "tube_model_loss, tube_mel_model_loss")   # paule/paule.py:59

BestSynthesisSomatosensory = namedtuple(
    "BestSynthesisSomatosensory", "tube_loss, tube_mel_loss, tube_semvec_loss, planned_cp, prod_sig, prod_tube, pred_tube, "
    "prod_tube_mel, pred_tube_mel, prod_tube_semvec, pred_tube_semvec")   # paule/paule.py:64

BestSynthesisAcoustic = namedtuple("BestSynthesisAcoustic", "mel_loss, planned_cp, prod_sig, prod_mel, pred_mel")
BestSynthesisSemantic = namedtuple("BestSynthesisSemantic", "semvec_loss, planned_cp, prod_sig, prod_semvec, pred_semvec")

# paule/paule.py:592-597
MEL_WEIGHT = 5.0
VELOCITY_WEIGHT = 80.0
JERK_WEIGHT = 400.0
SEMANTIC_WEIGHT = 10.0
SPEECH_CLASSIFIER_WEIGHT = 0.1
LOCAL_LINEAR_WEIGHT = 100_000

_COL = dict(total=0, mel=1, semvec=2, vel=3, jerk=4, ll=5, cls=6, tube_mel=6, tube_semvec=7)


def _default_planner_factory(pred_model, embedder, **kw):
    from .engine import HipPlanner   # imported lazily: raises HipLibraryError if libpaule_hip.so is missing
    return HipPlanner(pred_model, embedder, **kw)


def _np(t):
    return t.detach().cpu().numpy().copy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _scalar_or_vec(v):
    v = np.asarray(v, dtype=np.float64)
    return float(v[0]) if v.shape[0] == 1 else v.copy()


def _rmse_rows(a, b):
    d = (np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).reshape(a.shape[0], -1)
    return np.sqrt((d * d).mean(axis=1))


class PlannerAdam:
    """Stand-in for the reference's ``pred_optimizer = torch.optim.Adam(pred_model.parameters(), lr=0.001)`` (paule/paule.py:287):
    the moments live in the planner's handle on the device; this object carries them between plans and offers the two things
    the reference's users touch -- ``param_groups[0]['lr']`` (:473-474) and ``state_dict()`` / ``load_state_dict()``
    (docs/examples/minimal_example.py:51)."""

    def __init__(self, lr=0.001, betas=(0.9, 0.999), eps=1e-8):
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False)]
        self._state = {}

    def state_dict(self):
        n = len(self._state)
        return {"state": self._state, "param_groups": [dict(self.param_groups[0], params=list(range(n)))]}

    def load_state_dict(self, state_dict):
        self._state = dict(state_dict.get("state", {}))
        groups = state_dict.get("param_groups") or [{}]
        for k in ("lr", "betas", "eps"):
            if k in groups[0]:
                self.param_groups[0][k] = groups[0][k]


class Paule():
    """State of the Predictive Articulatory speech synthesis Using Lexical Embeddings planner
    (paule/paule.py:92-318), restricted to what the planning path needs."""

    def __init__(self, *, pred_model=None, pred_optimizer=None, inv_model=None, inv_optimizer=None,
                 embedder=None, cp_gen_model=None, mel_gen_model=None,
                 use_somatosensory_feedback=False, cp_tube_model=None, tube_optimizer=None,
                 tube_mel_model=None, tube_mel_optimizer=None, tube_embedder=None,
                 continue_data=None, device=torch.device('cuda'), smiling=False,
                 use_speech_classifier=False, speech_classifier=None,
                 speech_classifier_optimizer=None,
                 compute_dtype="f32", synthesizer=None, mel_extractor=None, planner_factory=None,
                 continue_learning_hook=None, tube_extractor=None):
        self.device = device
        self.smiling = smiling
        if use_somatosensory_feedback and use_speech_classifier:   # paule/paule.py:117-118
            raise NotImplementedError("at the moment you have to choose either to use `use_somatosenrosry_feedback=True` OR to use `use_speech_classifier=True` or none")
        # somatosensory feedback (paule/paule.py:227-273, :916-929): cp -> tube, tube -> mel, tube -> semantic vector beside the
        # acoustic path.  The three models have to be given: the reference's default tube embedder is built with dropout 0.7 and
        # switched to .train() inside the loop (:266, :927), which makes its planning loss random; the device path runs a tube
        # embedder without dropout (EmbeddingModel(input_size=10, ..., dropout=0) with the same weights is deterministic).
        self.use_somatosensory_feedback = bool(use_somatosensory_feedback)
        self.cp_tube_model, self.tube_mel_model, self.tube_embedder = cp_tube_model, tube_mel_model, tube_embedder
        # torch.optim.Adam(lr=0.001) each in the reference (paule/paule.py:296-306): stand-ins that carry the moments between plans
        self.tube_optimizer = tube_optimizer if tube_optimizer is not None else (PlannerAdam() if use_somatosensory_feedback else None)
        self.tube_mel_optimizer = tube_mel_optimizer if tube_mel_optimizer is not None else (PlannerAdam() if use_somatosensory_feedback else None)
        # tube_extractor(cp (B, T, 30) normalised) -> normalised tube (B, T, 10): the reference's
        # speak_and_extract_tube_information + get_area_info_within_oral_cavity + normalize_tube (paule/paule.py:1070-1078, VTL)
        self.tube_extractor = tube_extractor
        self.best_synthesis_somatosensory = None
        if self.use_somatosensory_feedback:
            if cp_tube_model is None or tube_mel_model is None or tube_embedder is None:
                raise NotImplementedError("use_somatosensory_feedback=True needs cp_tube_model=, tube_mel_model= and tube_embedder= (the "
                                          "reference's default tube embedder has dropout 0.7 and runs in .train() mode inside the "
                                          "loop: its loss is random; pass one built with dropout=0)")
            for m in (cp_tube_model, tube_mel_model, tube_embedder):
                if getattr(getattr(m, "lstm", None), "dropout", 0):
                    raise NotImplementedError("tube models with dropout > 0 are not supported (random planning loss in the reference)")
        self.use_speech_classifier = use_speech_classifier
        # default models: the reference loads its pretrained weights from paule/pretrained_models/ next to the module
        # (paule/paule.py:121-127, :146-150, :167-171, :215-222; a 200 MB download, paule/util.py:936-955).  Same file names, looked
        # up under $PAULE_PRETRAINED_DIR or <this package>/pretrained_models; the architecture is read off the state dict.
        def pretrained(rel, what, needed=True):
            import os
            root = os.environ.get("PAULE_PRETRAINED_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "pretrained_models")
            path = os.path.join(root, rel)
            if not os.path.isfile(path):
                if not needed:
                    return None
                raise FileNotFoundError(f"{what}: pretrained weights are not bundled (looked for {path}); pass {what}= (a module or "
                                        "state dict) or point PAULE_PRETRAINED_DIR at the reference's pretrained_models directory")
            sd = torch.load(path, map_location="cpu", weights_only=True)
            # the reference keeps torch modules (its users call paule_model.pred_model.state_dict(), docs/examples/minimal_example.py:50)
            from . import models as _m
            n_layers = len([k for k in sd if k.startswith("lstm.weight_hh_l")])
            if n_layers == 0:
                return sd                                                           # the speech classifier: a plain state dict
            hidden = int(sd["lstm.weight_hh_l0"].shape[1])
            if what == "pred_model":
                mod = _m.ForwardModel(input_size=int(sd["lstm.weight_ih_l0"].shape[1]), output_size=int(sd["post_linear.weight"].shape[0]),
                                      hidden_size=hidden, num_lstm_layers=n_layers)
            elif what == "embedder":
                mod = _m.EmbeddingModel(input_size=int(sd["lstm.weight_ih_l0"].shape[1]), output_size=int(sd["linear_mapping.weight"].shape[0]),
                                        hidden_size=hidden, num_lstm_layers=n_layers)
            else:
                mod = _m.InverseModelMelTimeSmoothResidual(
                    input_size=int(sd["lstm.weight_ih_l0"].shape[1]) // 3, output_size=int(sd["post_linear.weight"].shape[0]),
                    hidden_size=hidden, num_lstm_layers=n_layers,
                    mel_smooth_layers=len({k.split(".")[1] for k in sd if k.startswith("MelBlocks.")}),
                    resid_blocks=len({k.split(".")[1] for k in sd if k.startswith("ResidualConvBlocks.")}))
            mod = mod.double()            # the reference's models are float64 (paule/paule.py:124, :146, :167)
            mod.load_state_dict(sd)
            return mod

        if use_speech_classifier and speech_classifier is None:
            speech_classifier = pretrained("speech_classifier/linear_model_rec_as_nonspeech.pt", "speech_classifier")
        self.speech_classifier = speech_classifier
        if pred_model is None:
            pred_model = pretrained("predictive/pred_model_common_voice_1_720_lr_0001_50_00001_50_000001_50_0000001_200.pt", "pred_model")
        if embedder is None:
            embedder = pretrained("embedder/embed_model_common_voice_syn_rec_2_720_0_dropout_07_noise_6e05_rmse_lr_00001_200.pt", "embedder")
        if inv_model is None:   # only needed by initialize_from="acoustic": optional here
            inv_model = pretrained("inverse/inv_model_common_voice_3_1_720_5_lr_0001_50_00001_50_000001_50_0000001_200.pt", "inv_model",
                                   needed=False)
        self.pred_model = pred_model
        self.embedder = embedder
        self.inv_model = inv_model
        self.cp_gen_model = cp_gen_model
        self.mel_gen_model = mel_gen_model
        self.pred_optimizer = pred_optimizer if pred_optimizer is not None else PlannerAdam(lr=0.001)   # paule/paule.py:284-287
        self.inv_optimizer = inv_optimizer
        self.continue_data = continue_data
        self.continue_data_limit = 1000   # max amount of training data stored in the instance (paule/paule.py:277)
        if self.continue_data is not None and len(self.continue_data) > self.continue_data_limit:   # :279-282
            keep = random.sample(range(len(self.continue_data)), self.continue_data_limit)
            self.continue_data = (self.continue_data.iloc[keep].reset_index(drop=True) if hasattr(self.continue_data, "iloc")
                                  else [self.continue_data[i] for i in keep])
        self.compute_dtype = compute_dtype
        self.synthesizer = synthesizer
        self.mel_extractor = mel_extractor
        self.continue_learning_hook = continue_learning_hook
        self._planner_factory = planner_factory or _default_planner_factory
        self.best_synthesis_acoustic = None
        self.best_synthesis_semantic = None
        self.planner = None
        self._planners = {}   # handles of earlier plan_resynth calls, reused when batch, length, objective, dtype, lr and models agree

    def _get_planner(self, inv_sd=None, **kw):
        """The engine handle for this plan: a new one, or the one an earlier call with the same shape built (the reference's
        notebook plans utterance after utterance, paule/gradient_planning.ipynb cell 28: a handle per call would allocate, capture
        and retire one each time).  A reused handle gets the models' CURRENT weights (continued learning changes them between calls)
        and a clean slate for everything a call sets."""
        tube = kw.get("tube_models")

        def arch(m):   # what decides whether a handle fits a model: its tensors' names and shapes (NOT id(): a freed model's id can come back)
            if m is None:
                return None
            sd = m.state_dict() if hasattr(m, "state_dict") else m
            return tuple((k, tuple(v.shape)) for k, v in sd.items())

        key = (kw["batch"], kw["n_frames"], kw["objective"], kw["dtype"], float(kw["lr"]), bool(kw["smiling"]), str(kw["device"]),
               arch(self.pred_model), arch(self.embedder), arch(inv_sd), None if tube is None else tuple(arch(m) for m in tube))
        planner = self._planners.get(key)
        if planner is not None and hasattr(planner, "set_weights"):
            self._planners[key] = self._planners.pop(key)   # most recently used last
            planner.set_weights(pred_model=self.pred_model, embedder=self.embedder)
            if tube is not None and hasattr(planner, "set_tube_weights"):
                planner.set_tube_weights(*tube)
            if inv_sd is not None and hasattr(planner, "set_inverse_weights"):
                planner.set_inverse_weights(inv_sd)
            planner.set_past_cp(None)
            if hasattr(planner, "set_speech_classifier"):
                planner.set_speech_classifier(None)
            return planner
        planner = self._planner_factory(self.pred_model, self.embedder, **kw, **({"inv_model": inv_sd} if inv_sd is not None else {}))
        if hasattr(planner, "set_weights"):
            self._planners[key] = planner
            self._trim_planners(keep=planner)
        return planner

    # a few shapes stay resident; a handle holds its activations (1.9 GB at B = 256 x 300), so the cache is bounded by bytes as well
    PLANNER_CACHE_ENTRIES = 8
    PLANNER_CACHE_BYTES = 24 << 30

    def _trim_planners(self, keep=None):
        """Least recently used handles out until the cache fits its bounds; never the handle just built nor the one published as
        ``self.planner`` (user code may hold it)."""
        def total():
            return sum(int(getattr(pl_, "device_bytes", 0) or 0) for pl_ in self._planners.values())
        for key in list(self._planners):
            if len(self._planners) <= self.PLANNER_CACHE_ENTRIES and total() <= self.PLANNER_CACHE_BYTES:
                break
            pl_ = self._planners[key]
            if pl_ is keep or pl_ is self.planner:
                continue
            del self._planners[key]
            if hasattr(pl_, "close"):
                pl_.close()

    def release_planners(self):
        """Frees the cached engine handles (device memory)."""
        for pl_ in self._planners.values():
            if hasattr(pl_, "close"):
                pl_.close()
        self._planners = {}
        self.planner = None

    def plan_iterative(self, *, target_acoustic=None, target_semvecs=None, target_seq_lengths=None, overlap=8, **kwargs):
        pass   # a stub in the reference too (paule/paule.py:383-388)

    # ------------------------------------------------------------------------------------------
    def _produce(self, cp_bt):
        """synthesis checkpoint (host, outside the timed path): CP (B,T,30) -> (sigs, sr, prod_mel (B,T',60)) or None"""
        if self.synthesizer is None or self.mel_extractor is None:
            return None
        sigs, mels, sr = [], [], None
        for b in range(cp_bt.shape[0]):
            sig, sr = self.synthesizer(cp_bt[b])
            sigs.append(sig)
            mels.append(np.asarray(self.mel_extractor(sig, sr), dtype=np.float64))
        return sigs, sr, np.stack(mels)

    @staticmethod
    def create_epoch_batches(df_length, batch_size, shuffle=True, same_size_batching=False,
                             sorted_training_length_keys=None, training_length_dict=None):
        """Index batches of one epoch (behaviour of paule/paule.py:320-382).  ``same_size_batching``: samples are grouped
        by sequence length (``training_length_dict``: length -> indices), every length is shuffled and cut into full
        batches, the left-overs of all lengths are batched together in length order (last batch may be smaller), and the
        batch order is shuffled.  Otherwise: (shuffled) indices, wrapped around to fill the last batch."""
        if same_size_batching and training_length_dict is None:
            raise ValueError("Dictionary containing indices of samples with corresponding length needed for same_size_batching!")
        if same_size_batching:
            epoch, left_over = [], []
            for length in np.sort(list(training_length_dict.keys())):
                idxs = training_length_dict[length]
                random.shuffle(idxs)                       # in place, like the reference (:356)
                n_full = len(idxs) // batch_size
                epoch += [idxs[i * batch_size:(i + 1) * batch_size] for i in range(n_full)]
                if len(idxs) % batch_size:
                    left_over += list(idxs[n_full * batch_size:])
            left_over = np.asarray(left_over)
            n_full = len(left_over) // batch_size
            epoch += [left_over[i * batch_size:(i + 1) * batch_size] for i in range(n_full)]
            if len(left_over) % batch_size:
                epoch += [left_over[n_full * batch_size:]]
            random.shuffle(epoch)
            return epoch
        idxs = list(range(df_length))
        if shuffle:
            random.shuffle(idxs)
        if df_length % batch_size:
            idxs += idxs[:batch_size - df_length % batch_size]
        return [idxs[i * batch_size:(i + 1) * batch_size] for i in range(len(idxs) // batch_size)]

    def _continue_learning_pred(self, planner, cp_steps_ii, prod_mel_steps_ii, *, n_batches, batch_size, n_epochs, lr,
                                add_training_data_pred=False, target_semvec=None, prod_tube_steps_ii=None, tube_losses=None):
        """Continued learning of ``pred_model`` after an outer iteration (paule/paule.py:1244-1320, :1353-1379, :1406,
        :1439-1443): the (cp, mel) pairs produced in this iteration -- with ``add_training_data_pred`` half of every batch comes
        from ``self.continue_data`` instead (:1250-1287) -- are sampled, sorted by length and cut into same-size mini-batches for
        ``n_epochs`` epochs; a batch is padded to its longest sample by repeating the last frame (``pad_batch_online``,
        paule/util.py:674-726) and is one ``pred_optimizer`` step on the device.  Returns the mean loss of every epoch
        (``pred_model_loss``).  With B > 1 every utterance of a logged step is one produced sample.  Afterwards
        ``self.pred_model`` holds the trained parameters and ``self.continue_data`` has grown by the produced samples.
        ``prod_tube_steps_ii`` (continue_learning_tube, paule/paule.py:1381-1404): the produced tubes of the same samples; every
        mini-batch then also is one step of the cp -> tube model (cp_norm -> tube_norm) and one of the tube -> mel model
        (tube_norm -> melspec_norm_synthesized); their epoch means are appended to ``tube_losses`` (two lists)."""
        prod_cps = [np.asarray(x, dtype=np.float32) for c in cp_steps_ii for x in np.asarray(c).reshape(-1, *np.shape(c)[-2:])]
        prod_mels = [np.asarray(x, dtype=np.float32) for m in prod_mel_steps_ii for x in np.asarray(m).reshape(-1, *np.shape(m)[-2:])]
        learn_tube = prod_tube_steps_ii is not None
        prod_tubes = [np.asarray(x, dtype=np.float32) for u in (prod_tube_steps_ii or []) for x in np.asarray(u).reshape(-1, *np.shape(u)[-2:])]
        if learn_tube and len(prod_tubes) != len(prod_cps):
            raise ValueError("continue_learning_tube needs one produced tube per produced sample (tube_extractor)")
        train_tubes = None
        n = len(prod_cps)
        records = None
        if self.continue_data is not None:
            records = self.continue_data.to_dict("records") if hasattr(self.continue_data, "to_dict") else list(self.continue_data)
        train_cps, train_mels = None, None
        if add_training_data_pred:
            if not records:
                raise ValueError("add_training_data_pred=True needs continue_data (samples with 'cp_norm' and 'melspec_norm_synthesized')")
            half = int(0.5 * batch_size) * n_batches
            if n < half:                                                         # :1252-1263 (order of the two draws as there)
                pick_prod = random.sample(range(n), k=n)
                pick_data = random.sample(range(len(records)), k=n)
            else:                                                                # :1264-1267
                pick_data = random.sample(range(len(records)), k=half)
                pick_prod = random.sample(range(n), k=half)
            train_cps = [np.asarray(records[i]["cp_norm"], dtype=np.float32) for i in pick_data] + [prod_cps[i] for i in pick_prod]
            train_mels = [np.asarray(records[i]["melspec_norm_synthesized"], dtype=np.float32) for i in pick_data] + \
                         [prod_mels[i] for i in pick_prod]
            if learn_tube:                                                       # 'tube_norm' column, :1273
                train_tubes = [np.asarray(records[i]["tube_norm"], dtype=np.float32) for i in pick_data] + [prod_tubes[i] for i in pick_prod]
        k = n if n < batch_size * n_batches else batch_size * n_batches           # :1289-1303 (drawn in either case)
        picked = random.sample(range(n), k=k)
        if train_cps is None:                                                     # :1310-1313
            train_cps, train_mels = [prod_cps[i] for i in picked], [prod_mels[i] for i in picked]
            if learn_tube:
                train_tubes = [prod_tubes[i] for i in picked]
        order = np.argsort([len(c) for c in train_cps], kind="stable")            # sort_values(by="lens_input"), :1283, :1308
        train_cps, train_mels = [train_cps[i] for i in order], [train_mels[i] for i in order]
        if learn_tube:
            train_tubes = [train_tubes[i] for i in order]
        lens = np.array([len(c) for c in train_cps])
        if batch_size > getattr(planner, "train_capacity", planner.B):
            raise ValueError(f"batch_size={batch_size} of continued learning exceeds the planner's rows ({planner.train_capacity})")
        if lens.max() > planner.T:
            raise ValueError(f"a training sample of {lens.max()} frames does not fit the planner built for {planner.T} frames")

        def padded(seqs, max_len):                                               # add_and_pad: repeat the last frame
            return np.stack([np.concatenate((x, np.tile(x[-1:], (max_len - len(x), 1))), axis=0) for x in seqs])

        grp = self.pred_optimizer.param_groups[0] if self.pred_optimizer is not None else {}
        lr = grp.get("lr", lr)
        losses = []
        tube_grp = getattr(self.tube_optimizer, "param_groups", [{}])[0] if self.tube_optimizer is not None else {}
        tmel_grp = getattr(self.tube_mel_optimizer, "param_groups", [{}])[0] if self.tube_mel_optimizer is not None else {}
        for _ in range(n_epochs):
            by_len = {int(l): np.where(lens == l)[0] for l in np.unique(lens)}    # :1313-1319 (rebuilt: shuffled in place)
            epoch = self.create_epoch_batches(len(lens), batch_size, shuffle=True, same_size_batching=True, training_length_dict=by_len)
            step_losses, tube_step, tmel_step = [], [], []
            for j in epoch:
                cp_b = padded([train_cps[i] for i in j], max(len(train_cps[i]) for i in j))
                mel_b = padded([train_mels[i] for i in j], max(len(train_mels[i]) for i in j))
                if mel_b.shape[1] != cp_b.shape[1] // 2:
                    raise ValueError("a training batch's mel length has to be half its cp length (ForwardModel halves the sequence)")
                step_losses.append(planner.train_pred_step(cp_b, mel_b, lr=lr, betas=grp.get("betas", (0.9, 0.999)), eps=grp.get("eps", 1e-8)))
                if learn_tube:                                                   # :1381-1404, Adam(lr 0.001) each (:300, :305)
                    tube_b = padded([train_tubes[i] for i in j], cp_b.shape[1])
                    tube_step.append(planner.train_model_step("cp_tube", cp_b, tube_b, lr=tube_grp.get("lr", 0.001),
                                                              betas=tube_grp.get("betas", (0.9, 0.999)), eps=tube_grp.get("eps", 1e-8)))
                    tmel_step.append(planner.train_model_step("tube_mel", tube_b, mel_b, lr=tmel_grp.get("lr", 0.001),
                                                              betas=tmel_grp.get("betas", (0.9, 0.999)), eps=tmel_grp.get("eps", 1e-8)))
            losses.append(float(np.mean([float(x) for x in step_losses])))
            if learn_tube and tube_losses is not None:                           # :1407-1409
                tube_losses[0].append(float(np.mean([float(x) for x in tube_step])))
                tube_losses[1].append(float(np.mean([float(x) for x in tmel_step])))
        if self.continue_data is not None:                                        # :1439-1443
            vec = None if target_semvec is None else np.asarray(target_semvec)
            new = [{"vector": None if vec is None else vec[min(i % max(len(vec), 1), len(vec) - 1)].copy(), "cp_norm": c,
                    "melspec_norm_synthesized": m, "segment_data": False} for i, (c, m) in enumerate(zip(prod_cps, prod_mels))]
            if prod_tubes:                                                       # :1251
                for rec, u in zip(new, prod_tubes):
                    rec["tube_norm"] = u
            if hasattr(self.continue_data, "to_dict"):
                import pandas as pd
                data = pd.concat([self.continue_data, pd.DataFrame(new)]).reset_index(drop=True)
                if len(data) > self.continue_data_limit:
                    data = data.iloc[random.sample(range(len(data)), k=self.continue_data_limit)].reset_index(drop=True)
            else:
                data = list(self.continue_data) + new
                if len(data) > self.continue_data_limit:
                    data = [data[i] for i in random.sample(range(len(data)), k=self.continue_data_limit)]
            self.continue_data = data
        if self.pred_optimizer is not None and hasattr(planner, "get_pred_optimizer_state"):
            sd = planner.get_pred_optimizer_state(lr=lr, betas=grp.get("betas", (0.9, 0.999)), eps=grp.get("eps", 1e-8))
            try:
                self.pred_optimizer.load_state_dict(sd)
            except (ValueError, KeyError):   # a torch optimizer over different parameters: keep ours alongside
                self.pred_optimizer = PlannerAdam(lr=lr)
                self.pred_optimizer.load_state_dict(sd)
        if self.pred_model is not None:      # the reference trains self.pred_model in place: keep module / state dict in sync
            sd = planner.get_weights("pred")
            ref = self.pred_model.state_dict() if hasattr(self.pred_model, "state_dict") else self.pred_model
            new = {k_: torch.as_tensor(v).to(device=ref[k_].device, dtype=ref[k_].dtype) for k_, v in sd.items()}
            if hasattr(self.pred_model, "load_state_dict"):
                self.pred_model.load_state_dict(new)
            else:
                self.pred_model.update(new)
        if learn_tube and hasattr(planner, "get_optimizer_state"):   # tube_optimizer / tube_mel_optimizer outlive the plan
            for attr, name, g_ in (("tube_optimizer", "cp_tube", tube_grp), ("tube_mel_optimizer", "tube_mel", tmel_grp)):
                sd = planner.get_optimizer_state(name, lr=g_.get("lr", 0.001), betas=g_.get("betas", (0.9, 0.999)), eps=g_.get("eps", 1e-8))
                try:
                    getattr(self, attr).load_state_dict(sd)
                except (ValueError, KeyError, AttributeError):
                    opt = PlannerAdam(lr=g_.get("lr", 0.001))
                    opt.load_state_dict(sd)
                    setattr(self, attr, opt)
        if learn_tube:                       # the trained tube models back into the instance, like pred_model
            for attr, name in (("cp_tube_model", "cp_tube"), ("tube_mel_model", "tube_mel")):
                cur = getattr(self, attr)
                sd = planner.get_weights(name)
                ref = cur.state_dict() if hasattr(cur, "state_dict") else cur
                new = {k_: torch.as_tensor(v).to(device=ref[k_].device, dtype=ref[k_].dtype) for k_, v in sd.items()}
                if hasattr(cur, "load_state_dict"):
                    cur.load_state_dict(new)
                else:
                    cur.update(new)
        return losses

    def plan_resynth(self, *, learning_rate_planning=0.01, learning_rate_learning=0.001,
                     learning_rate_learning_inv=None,
                     target_acoustic=None,
                     target_semvec=None,
                     target_seq_length=None,
                     initial_cp=None,
                     past_cp=None,
                     initialize_from="acoustic",
                     objective="acoustic",
                     n_outer=5, n_inner=24,
                     continue_learning=True,
                     continue_learning_inv=False,
                     continue_learning_tube=False,
                     add_training_data_pred=False,
                     add_training_data_inv=False,
                     n_batches=3, batch_size=8, n_epochs=10,
                     log_ii=1,
                     log_semantics=True,
                     log_gradients=False,
                     log_signals=False,
                     log_cps=False,
                     plot=False,
                     seed=None,
                     verbose=True):
        """plans resynthesis cp trajectories (paule/paule.py:391-1550); see the module docstring for the
        differences (injectable synthesis, batched targets)."""
        if seed:
            torch.manual_seed(seed)
            random.seed(seed)

        if target_acoustic is None and target_semvec is None:
            raise ValueError("Either target_acoustic or target_semvec has to be not None.")

        if learning_rate_learning and self.pred_optimizer is not None:
            for param_group in self.pred_optimizer.param_groups:
                param_group['lr'] = learning_rate_learning
        if learning_rate_learning_inv and self.inv_optimizer is not None:
            for param_group in self.inv_optimizer.param_groups:
                param_group['lr'] = learning_rate_learning_inv

        if log_ii is None:
            log_ii = n_inner
        if log_ii > n_inner:
            raise ValueError('results can only be logged between first and last planning step')

        # ---- target (paule/paule.py:486-531) ----
        target_sig = target_sr = None
        target_mel = None
        if isinstance(target_acoustic, str) or (target_acoustic is not None and not hasattr(target_acoustic, "shape")
                                                and len(target_acoustic) == 2):
            if self.mel_extractor is None:
                raise NotImplementedError("audio targets need mel_extractor= (librosa/soundfile are host code outside "
                                          "the planning path); pass a (T', 60) mel array instead")
            if isinstance(target_acoustic, str):
                import soundfile as sf   # noqa: deferred, optional
                target_sig, target_sr = sf.read(target_acoustic)
                if len(target_sig.shape) == 2:
                    target_sig = target_sig.mean(axis=1)
            else:
                target_sig, target_sr = target_acoustic
            target_mel = np.asarray(self.mel_extractor(target_sig, target_sr), dtype=np.float64)
            target_mel = target_mel - target_mel.min()
            target_mel = target_mel[None]
            target_seq_length = target_mel.shape[1]
        elif target_acoustic is not None:
            target_mel = _np(target_acoustic).astype(np.float64)
            if target_mel.ndim == 2:
                target_mel = target_mel[None]
            if target_mel.ndim != 3:
                raise ValueError("target_acoustic has to be torch.Tensor at this point")
            target_seq_length = target_mel.shape[1]

        if target_acoustic is None and (target_seq_length is None or target_semvec is None):
            raise ValueError("if target_acoustic is None you need to give a target_seq_length and a target_semvec")
        elif target_acoustic is None:
            if self.mel_gen_model is None:
                raise NotImplementedError("semvec-only targets need mel_gen_model= (paule/paule.py:515-521)")
            noise = torch.randn(1, 1, 100)
            sv = torch.as_tensor(_np(target_semvec)).view(1, -1)
            target_mel = _np(self.mel_gen_model(noise, target_seq_length, sv)).astype(np.float64)

        # argument checks the reference reaches only after running the inverse model / building the criterion
        # (paule/paule.py:566-576, :775-776); they are pure validation, so they are done before any model is needed
        if initial_cp is None and initialize_from not in ("acoustic", "semvec"):
            raise ValueError("initialize_from has to be either 'acoustic' or 'semvec'")
        if initial_cp is not None and initialize_from is not None:
            raise ValueError('one of initial_cp and initialize_from has to be None')
        if past_cp is not None and np.asarray(_np(past_cp)).shape[0] % 2 != 0:
            raise ValueError("past_cp have to be None or the sequence length has to be an even number")
        if objective not in ("acoustic", "acoustic_semvec", "semvec"):
            raise ValueError("objective has to be one of 'acoustic_semvec', 'acoustic' or 'semvec'")

        B = target_mel.shape[0]

        # ---- initial cp (paule/paule.py:550-573) ----
        planner = None
        tube_kw = {}
        if self.use_somatosensory_feedback:
            if objective == "acoustic":
                raise NotImplementedError("use_somatosensory_feedback with objective='acoustic': the reference's criterion fails there "
                                          "(unassigned pred_tube_semvec, paule/paule.py:692); use 'acoustic_semvec' or 'semvec'")
            if continue_learning_tube and self.tube_extractor is None:
                raise NotImplementedError("continue_learning_tube needs tube_extractor= (the produced tubes are the training targets)")
            tube_kw = dict(tube_models=(self.cp_tube_model, self.tube_mel_model, self.tube_embedder))
        if initial_cp is None:
            if initialize_from == "acoustic":
                if self.inv_model is None:
                    raise NotImplementedError("initialize_from='acoustic' needs inv_model= (an InverseModelMelTimeSmoothResidual "
                                              "module / state dict, or any callable); or pass initial_cp= with initialize_from=None")
                inv_sd = self.inv_model if isinstance(self.inv_model, dict) else (
                    self.inv_model.state_dict() if hasattr(self.inv_model, "MelBlocks") else None)
                if inv_sd is not None:
                    # the inverse model runs on the device, inside the planner's handle (pl_inverse_forward): the planner is
                    # built first; its length is known: past_cp + 2 x target mel frames (paule/paule.py:553, :582-583)
                    n_past = 0 if past_cp is None else np.asarray(_np(past_cp)).shape[0]
                    planner = self._get_planner(inv_sd=inv_sd, batch=B, n_frames=n_past + 2 * target_mel.shape[1],
                                                objective=objective, dtype=self.compute_dtype, lr=learning_rate_planning,
                                                smiling=self.smiling, device=self.device, **tube_kw)
                    initial_cp = _np(planner.inverse_forward(target_mel, clip=True))
                else:
                    with torch.no_grad():
                        initial_cp = _np(self.inv_model(torch.as_tensor(target_mel))).clip(min=-1, max=1)
            elif initialize_from == "semvec":
                if self.cp_gen_model is None:
                    raise NotImplementedError("initialize_from='semvec' needs cp_gen_model=")
                noise = torch.randn(1, 1, 100)
                sv = torch.as_tensor(_np(target_semvec)).view(1, -1)
                initial_cp = _np(self.cp_gen_model(noise, 2 * target_seq_length, sv))
            else:
                raise ValueError("initialize_from has to be either 'acoustic' or 'semvec'")
            initial_cp = np.asarray(initial_cp, dtype=np.float64)
            if initial_cp.ndim == 2:
                initial_cp = initial_cp[None]
        else:
            if initialize_from is not None:
                raise ValueError('one of initial_cp and initialize_from has to be None')
            initial_cp = _np(initial_cp).astype(np.float64)
            if initial_cp.ndim == 2:
                initial_cp = initial_cp[None]
            if not initial_cp.shape[1] == (target_mel.shape[1] * 2):
                raise ValueError(f"initial_cp {initial_cp.shape[1]}, target_mel {target_mel.shape[1] * 2}")
        if initial_cp.shape[0] != B:
            raise ValueError(f"initial_cp holds {initial_cp.shape[0]} utterances, target {B}")

        if past_cp is not None:
            past_cp = _np(past_cp).astype(np.float64)
            if past_cp.shape[0] % 2 != 0:
                raise ValueError("past_cp have to be None or the sequence length has to be an even number")
            initial_cp = np.concatenate((np.broadcast_to(past_cp, (B,) + past_cp.shape), initial_cp), axis=1)

        T = initial_cp.shape[1]
        Tp = T // 2

        # ---- target semvec (paule/paule.py:533-540) ----
        emb_for_target = None
        if target_semvec is None:
            emb_for_target = target_mel   # embedded below, once the engine exists
        else:
            target_semvec = _np(target_semvec).astype(np.float64)
            target_semvec = target_semvec.reshape(-1, target_semvec.shape[-1])   # .view(1, 300), paule/paule.py:539
            if target_semvec.shape[0] == 1 and B > 1:
                target_semvec = np.repeat(target_semvec, B, axis=0)

        # ---- engine ----
        if planner is None:
            planner = self._get_planner(batch=B, n_frames=T, objective=objective, dtype=self.compute_dtype, lr=learning_rate_planning,
                                        smiling=self.smiling, device=self.device, **tube_kw)
        self.planner = planner
        if continue_learning and hasattr(planner, "set_pred_optimizer_state"):   # the optimiser outlives a plan (paule/paule.py:284-287)
            planner.set_pred_optimizer_state(self.pred_optimizer.state_dict())
        if self.use_somatosensory_feedback and continue_learning and continue_learning_tube and hasattr(planner, "set_optimizer_state"):
            for name, opt in (("cp_tube", self.tube_optimizer), ("tube_mel", self.tube_mel_optimizer)):
                try:
                    planner.set_optimizer_state(name, opt.state_dict())
                except (ValueError, KeyError):   # a torch optimizer over other parameter objects: start fresh on the device
                    planner.set_optimizer_state(name, {"state": {}})
        planner.set_cp(initial_cp)
        planner.reset_optimizer()                       # a fresh Adam per call (paule/paule.py:797)
        cls_w = cls_b = None
        if self.use_speech_classifier:                  # paule/paule.py:604-622, :914-915
            planner.set_speech_classifier(self.speech_classifier, SPEECH_CLASSIFIER_WEIGHT)
            sd = self.speech_classifier.state_dict() if hasattr(self.speech_classifier, "state_dict") else self.speech_classifier
            cls_w, cls_b = _np(sd["linear.weight"]).reshape(-1).astype(np.float64), float(_np(sd["linear.bias"]).reshape(-1)[0])
        if past_cp is not None:
            planner.set_past_cp(past_cp)

        # initial results (paule/paule.py:821-878)
        initial_pred_mel, initial_pred_semvec = planner.get_pred()
        initial_pred_mel, initial_pred_semvec = _np(initial_pred_mel), _np(initial_pred_semvec)
        produced = self._produce(initial_cp)
        initial_sig = initial_sr = initial_prod_mel = initial_prod_semvec = None
        if produced is not None:
            initial_sig, initial_sr, initial_prod_mel = produced

        if past_cp is not None:
            # the reference prepends the initial PRODUCTION's first P/2 frames (paule/paule.py:868-870); without a
            # synthesizer the initial prediction stands in for it
            head = initial_prod_mel if initial_prod_mel is not None else initial_pred_mel
            target_mel = np.concatenate((head[:, 0:(past_cp.shape[0] // 2), :], target_mel), axis=1)
        if target_mel.shape[1] != Tp:
            raise ValueError(f"initial_cp {T}, target_mel {target_mel.shape[1] * 2}")

        if emb_for_target is not None:
            if emb_for_target.shape[1] == Tp:
                target_semvec = _np(planner.embed_mel(emb_for_target))
            else:   # past_cp changed the length: embed the un-prefixed target with its own handle
                target_semvec = _np(self.embedder(torch.as_tensor(emb_for_target, device=self.device),
                                                  (torch.tensor(emb_for_target.shape[1]),)))
        planner.set_targets(target_mel, target_semvec)
        if initial_prod_mel is not None:
            initial_prod_semvec = _np(planner.embed_mel(initial_prod_mel))

        soma = self.use_somatosensory_feedback
        initial_pred_tube = initial_pred_tube_mel = initial_pred_tube_semvec = None
        initial_prod_tube = initial_prod_tube_mel = initial_prod_tube_semvec = None
        if soma:                                        # paule/paule.py:826-862
            initial_pred_tube, initial_pred_tube_mel, initial_pred_tube_semvec = [_np(a) for a in planner.get_tube_pred()]
            if self.tube_extractor is not None:
                initial_prod_tube = np.asarray(self.tube_extractor(initial_cp), dtype=np.float64).reshape(initial_pred_tube.shape)
                initial_prod_tube_mel, initial_prod_tube_semvec = [_np(a) for a in planner.embed_tube(initial_prod_tube)]
            self.best_synthesis_somatosensory = BestSynthesisSomatosensory(
                np.inf, np.inf, np.inf, initial_cp, initial_sig, initial_prod_tube, initial_pred_tube, initial_prod_tube_mel,
                initial_pred_tube_mel, initial_prod_tube_semvec, initial_pred_tube_semvec)
        prod_tube_loss_steps, pred_tube_mel_loss_steps, prod_tube_mel_loss_steps = [], [], []
        pred_tube_semvec_loss_steps, prod_tube_semvec_loss_steps = [], []
        prod_tube_steps, pred_tube_steps, prod_tube_mel_steps, pred_tube_mel_steps = [], [], [], []
        prod_tube_semvec_steps, pred_tube_semvec_steps, tube_model_loss, tube_mel_model_loss = [], [], [], []
        prod_tube = prod_tube_mel = prod_tube_semvec = None
        self.best_synthesis_acoustic = BestSynthesisAcoustic(np.inf, initial_cp, initial_sig, initial_prod_mel, initial_pred_mel)
        self.best_synthesis_semantic = BestSynthesisSemantic(np.inf, initial_cp, initial_sig, initial_prod_semvec, initial_pred_semvec)

        # ---- logging variables (paule/paule.py:778-795) ----
        prod_loss_steps, planned_loss_steps, planned_mel_loss_steps = [], [], []
        vel_loss_steps, jerk_loss_steps, pred_semvec_loss_steps, prod_semvec_loss_steps = [], [], [], []
        cp_steps, pred_semvec_steps, prod_semvec_steps, grad_steps, sig_steps = [], [], [], [], []
        pred_mel_steps, prod_mel_steps, pred_model_loss, inv_model_loss = [], [], [], []
        pred_speech_classifier_loss_steps, prod_speech_classifier_loss_steps = [], []
        sig = sr = prod_mel = None
        pred_mel = initial_pred_mel
        squeeze = (lambda a: a[-1]) if B == 1 else (lambda a: a)

        def log_rows(rows, first_ii):
            if not verbose:
                return
            for k, row in enumerate(rows):   # paule/paule.py:964-978
                print("Iteration %d" % (first_ii + k))
                print("Planned Loss: ", _scalar_or_vec(row[:, _COL["total"]]))
                print("Mel Loss: ", _scalar_or_vec(row[:, _COL["mel"]]))
                print("Vel Loss: ", _scalar_or_vec(row[:, _COL["vel"]]))
                print("Jerk Loss: ", _scalar_or_vec(row[:, _COL["jerk"]]))
                print("Local Linear Loss: ", _scalar_or_vec(row[:, _COL["ll"]]))
                if objective != "acoustic":
                    print("Semvec Loss: ", _scalar_or_vec(row[:, _COL["semvec"]]))
                if self.use_speech_classifier:
                    print("Speech Classifier Loss: ", _scalar_or_vec(row[:, _COL["cls"]]))
                if self.use_somatosensory_feedback:
                    print("Tube Mel Loss: ", _scalar_or_vec(row[:, _COL["tube_mel"]]))
                    print("Tube Semvec Loss: ", _scalar_or_vec(row[:, _COL["tube_semvec"]]))

        def run(n, first_ii):
            """n plain iterations (no log step inside)"""
            if n <= 0:
                return
            if log_gradients:
                for k in range(n):
                    loss, grad = planner.step(1, return_grad=True)
                    grad_steps.append(grad.detach().clone())
                    log_rows(_np(loss), first_ii + k)
            else:
                log_rows(_np(planner.step(n)), first_ii)

        start_time = time.time()
        for ii_outer in range(n_outer):                                        # paule/paule.py:894
            pred_mel_steps_ii, prod_mel_steps_ii, cp_steps_ii = [], [], []
            pred_semvec_steps_ii, prod_semvec_steps_ii = [], []
            pred_tube_steps_ii, prod_tube_steps_ii, pred_tube_mel_steps_ii, prod_tube_mel_steps_ii = [], [], [], []
            pred_tube_semvec_steps_ii, prod_tube_semvec_steps_ii = [], []
            ii = 0
            while ii < n_inner:
                to_log = log_ii - 1 - (ii % log_ii)            # plain iterations before the next log step
                if ii + to_log >= n_inner:                     # no further log step in this outer iteration
                    run(n_inner - ii, ii)
                    break
                run(to_log, ii)
                ii += to_log
                # ---- log step (paule/paule.py:941-962, :1065-1197): losses, CP and predictions at the PRE-step CP ----
                xx_pre = _np(planner.get_cp())
                pm, ps = planner.get_pred(with_semvec=(objective != "acoustic" or log_semantics))
                pred_mel = _np(pm)
                pred_semvec = _np(ps) if ps is not None else None
                if soma:                                   # predictions of the tube path at the pre-step CP (paule/paule.py:1081-1086)
                    pred_tube, pred_tube_mel, pred_tube_semvec = [_np(a) for a in planner.get_tube_pred()]
                if log_gradients:
                    loss, grad = planner.step(1, return_grad=True)
                    grad_steps.append(grad.detach().clone())
                else:
                    loss = planner.step(1)
                row = _np(loss)[0]
                log_rows([row], ii)
                planned_loss_steps.append(_scalar_or_vec(row[:, _COL["total"]]))
                planned_mel_loss_steps.append(_scalar_or_vec(row[:, _COL["mel"]]))
                vel_loss_steps.append(_scalar_or_vec(row[:, _COL["vel"]]))
                jerk_loss_steps.append(_scalar_or_vec(row[:, _COL["jerk"]]))
                if self.use_speech_classifier:
                    pred_speech_classifier_loss_steps.append(_scalar_or_vec(row[:, _COL["cls"]]))
                if soma:                                   # paule/paule.py:947-949, :995-997
                    pred_tube_mel_loss_steps.append(_scalar_or_vec(row[:, _COL["tube_mel"]]))
                    pred_tube_semvec_loss_steps.append(_scalar_or_vec(row[:, _COL["tube_semvec"]]))
                    pred_tube_steps_ii.append(squeeze(pred_tube))
                    pred_tube_mel_steps_ii.append(squeeze(pred_tube_mel))
                    pred_tube_semvec_steps_ii.append(squeeze(pred_tube_semvec))
                    if self.tube_extractor is not None:    # production side (paule/paule.py:1069-1095, :1147-1160)
                        prod_tube = np.asarray(self.tube_extractor(xx_pre), dtype=np.float64).reshape(pred_tube.shape)
                        prod_tube_mel, prod_tube_semvec = [_np(a) for a in planner.embed_tube(prod_tube)]
                        prod_tube_steps_ii.append(squeeze(prod_tube))
                        prod_tube_mel_steps_ii.append(squeeze(prod_tube_mel))
                        prod_tube_semvec_steps_ii.append(squeeze(prod_tube_semvec))
                        prod_tube_loss = _rmse_rows(pred_tube, prod_tube)
                        prod_tube_mel_loss = MEL_WEIGHT * _rmse_rows(prod_tube_mel, target_mel)
                        prod_tube_semvec_loss = SEMANTIC_WEIGHT * _rmse_rows(prod_tube_semvec, target_semvec)
                        prod_tube_loss_steps.append(_scalar_or_vec(prod_tube_loss))
                        prod_tube_mel_loss_steps.append(_scalar_or_vec(prod_tube_mel_loss))
                        prod_tube_semvec_loss_steps.append(_scalar_or_vec(prod_tube_semvec_loss))
                        if verbose:
                            print("Produced Tube Loss: ", _scalar_or_vec(prod_tube_loss))
                        new_so = BestSynthesisSomatosensory(float(prod_tube_loss.mean()), float(prod_tube_mel_loss.mean()),
                                                            float(prod_tube_semvec_loss.mean()), xx_pre, None, prod_tube, pred_tube,
                                                            prod_tube_mel, pred_tube_mel, prod_tube_semvec, pred_tube_semvec)
                        if self.best_synthesis_somatosensory.tube_loss > new_so.tube_loss:
                            self.best_synthesis_somatosensory = new_so
                if objective != "acoustic":
                    pred_semvec_loss_steps.append(_scalar_or_vec(row[:, _COL["semvec"]]))
                elif log_semantics and pred_semvec is not None:
                    pred_semvec_loss_steps.append(_scalar_or_vec(SEMANTIC_WEIGHT * _rmse_rows(pred_semvec, target_semvec)))
                if pred_semvec is not None:
                    pred_semvec_steps_ii.append(squeeze(pred_semvec))
                cp_steps_ii.append(squeeze(xx_pre))
                pred_mel_steps_ii.append(squeeze(pred_mel))
                produced = self._produce(xx_pre)
                if produced is not None:
                    sig, sr, prod_mel = produced
                    if log_signals:
                        sig_steps.append(sig if B > 1 else sig[0])
                    prod_mel_steps_ii.append(squeeze(prod_mel))
                    prod_loss = MEL_WEIGHT * _rmse_rows(prod_mel, target_mel)
                    prod_loss_steps.append(_scalar_or_vec(prod_loss))
                    if self.use_speech_classifier:      # paule/paule.py:1114-1122, on the host (log steps only)
                        z = (prod_mel @ cls_w + cls_b).mean(axis=1)
                        prod_speech_classifier_loss_steps.append(_scalar_or_vec(
                            SPEECH_CLASSIFIER_WEIGHT * (np.maximum(z, 0.0) + np.log1p(np.exp(-np.abs(z))))))
                    if verbose:
                        print("Produced Mel Loss: ", _scalar_or_vec(prod_loss))
                    new_ac = BestSynthesisAcoustic(float(prod_loss.mean()), xx_pre, sig, prod_mel, pred_mel)
                    if self.best_synthesis_acoustic.mel_loss > new_ac.mel_loss:
                        self.best_synthesis_acoustic = new_ac
                    if objective in ('semvec', 'acoustic_semvec') or log_semantics:
                        prod_semvec = _np(planner.embed_mel(prod_mel))
                        prod_semvec_steps_ii.append(squeeze(prod_semvec))
                        prod_semvec_loss = SEMANTIC_WEIGHT * _rmse_rows(prod_semvec, target_semvec)
                        prod_semvec_loss_steps.append(_scalar_or_vec(prod_semvec_loss))
                        if verbose:
                            print("Produced Semvec Loss: ", _scalar_or_vec(prod_semvec_loss))
                        new_se = BestSynthesisSemantic(float(prod_semvec_loss.mean()), xx_pre, sig, prod_semvec, pred_semvec)
                        if self.best_synthesis_semantic.semvec_loss > new_se.semvec_loss:
                            self.best_synthesis_semantic = new_se
                ii += 1

            prod_mel_steps.append(prod_mel_steps_ii)
            if log_cps:
                cp_steps.append(cp_steps_ii)
            pred_mel_steps.append(pred_mel_steps_ii)
            pred_semvec_steps.append(pred_semvec_steps_ii)
            prod_semvec_steps.append(prod_semvec_steps_ii)
            if soma:
                prod_tube_steps.append(prod_tube_steps_ii)
                pred_tube_steps.append(pred_tube_steps_ii)
                prod_tube_mel_steps.append(prod_tube_mel_steps_ii)
                pred_tube_mel_steps.append(pred_tube_mel_steps_ii)
                prod_tube_semvec_steps.append(prod_tube_semvec_steps_ii)
                pred_tube_semvec_steps.append(pred_tube_semvec_steps_ii)

            # execute and continue learning (paule/paule.py:1243-1454).  The predictive model's mini-batch steps
            # (:1353-1379) run on the device through the planner (pl_train_pred_step); a hook may replace them.
            if continue_learning:
                if self.continue_learning_hook is not None:
                    losses = self.continue_learning_hook(self, cp_steps_ii, prod_mel_steps_ii, n_batches=n_batches,
                                                         batch_size=batch_size, n_epochs=n_epochs)
                    if losses:
                        pred_model_loss.extend(losses)
                    planner.set_weights(self.pred_model, None)
                elif prod_mel_steps_ii and hasattr(planner, "train_pred_step"):
                    pred_model_loss.extend(self._continue_learning_pred(
                        planner, cp_steps_ii, prod_mel_steps_ii, n_batches=n_batches, batch_size=batch_size,
                        n_epochs=n_epochs, lr=learning_rate_learning or 0.001, add_training_data_pred=add_training_data_pred,
                        target_semvec=target_semvec,
                        prod_tube_steps_ii=prod_tube_steps_ii if (soma and continue_learning_tube) else None,
                        tube_losses=(tube_model_loss, tube_mel_model_loss)))
                elif ii_outer == 0:
                    warnings.warn("continue_learning=True but nothing was synthesised (no synthesizer / mel_extractor): the "
                                  "predictive model is kept fixed", stacklevel=2)

        if verbose:
            print("--- %.2f min ---" % ((time.time() - start_time) / 60))

        # ---- results (paule/paule.py:1456-1550): CP AFTER the last step, predictions recomputed from it ----
        if hasattr(planner, "check"):
            planner.check()   # a device-side wait that timed out must surface here, not as a plan of garbage
        planned_cp = _np(planner.get_cp())
        pm, ps = planner.get_pred()
        pred_mel, pred_semvec = _np(pm), _np(ps)
        prod_semvec = _np(planner.embed_mel(prod_mel)) if prod_mel is not None else None
        sq = (lambda a: None if a is None else a[-1]) if B == 1 else (lambda a: a)
        sqs = (lambda s: None if s is None else s[0]) if B == 1 else (lambda s: s)
        if soma:                                        # paule/paule.py:1466-1470, :1534-1541
            pred_tube, pred_tube_mel, pred_tube_semvec = [_np(a) for a in planner.get_tube_pred()]
            return PlanningResultsWithSomatosensory(
                sq(planned_cp), sq(initial_cp), sqs(initial_sig), initial_sr, sq(initial_prod_mel), sq(initial_pred_mel),
                sq(initial_prod_tube), sq(initial_pred_tube), sq(initial_prod_tube_mel), sq(initial_pred_tube_mel),
                target_sig, target_sr, sq(target_mel), sqs(sig), sr, sq(prod_mel), sq(pred_mel),
                sq(prod_tube), sq(pred_tube), sq(prod_tube_mel), sq(pred_tube_mel),
                sq(initial_prod_semvec), sq(initial_pred_semvec), sq(initial_prod_tube_semvec), sq(initial_pred_tube_semvec),
                sq(prod_semvec), sq(pred_semvec), sq(prod_tube_semvec), sq(pred_tube_semvec),
                prod_loss_steps, planned_loss_steps, planned_mel_loss_steps, vel_loss_steps, jerk_loss_steps,
                pred_semvec_loss_steps, prod_semvec_loss_steps, prod_tube_loss_steps, pred_tube_mel_loss_steps,
                prod_tube_mel_loss_steps, pred_tube_semvec_loss_steps, prod_tube_semvec_loss_steps, cp_steps,
                pred_semvec_steps, prod_semvec_steps, grad_steps, sig_steps, prod_mel_steps, pred_mel_steps,
                prod_tube_steps, pred_tube_steps, prod_tube_mel_steps, pred_tube_mel_steps, prod_tube_semvec_steps,
                pred_tube_semvec_steps, pred_model_loss, inv_model_loss, tube_model_loss, tube_mel_model_loss)
        if self.use_speech_classifier:
            return PlanningResultsWithSpeechClassifier(
                sq(planned_cp), sq(initial_cp), sqs(initial_sig), initial_sr, sq(initial_prod_mel), sq(initial_pred_mel),
                target_sig, target_sr, sq(target_mel), sqs(sig), sr, sq(prod_mel), sq(pred_mel),
                sq(initial_prod_semvec), sq(initial_pred_semvec), sq(prod_semvec), sq(pred_semvec),
                prod_loss_steps, planned_loss_steps, planned_mel_loss_steps, vel_loss_steps, jerk_loss_steps,
                pred_semvec_loss_steps, prod_semvec_loss_steps, pred_speech_classifier_loss_steps,
                prod_speech_classifier_loss_steps, cp_steps, pred_semvec_steps, prod_semvec_steps,
                grad_steps, sig_steps, prod_mel_steps, pred_mel_steps, pred_model_loss, inv_model_loss)
        return PlanningResults(
            sq(planned_cp), sq(initial_cp), sqs(initial_sig), initial_sr, sq(initial_prod_mel), sq(initial_pred_mel),
            target_sig, target_sr, sq(target_mel), sqs(sig), sr, sq(prod_mel), sq(pred_mel),
            sq(initial_prod_semvec), sq(initial_pred_semvec), sq(prod_semvec), sq(pred_semvec),
            prod_loss_steps, planned_loss_steps, planned_mel_loss_steps, vel_loss_steps, jerk_loss_steps,
            pred_semvec_loss_steps, prod_semvec_loss_steps, cp_steps, pred_semvec_steps, prod_semvec_steps,
            grad_steps, sig_steps, prod_mel_steps, pred_mel_steps, pred_model_loss, inv_model_loss)
