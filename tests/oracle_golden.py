"""Outputs of this repository's OWN CPU oracle (oracle/: test infrastructure, pinned to the reference by tests/test_oracle.py) on the
full-size configurations, kept as committed fixtures tests/golden/oracle_<case>.npz.

Why.  The GPU box gives the test step 900 s.  The float64 oracle on Paule's models costs seconds to minutes per case there (T = 2000
through 720-wide recurrences: 100 s), and the full-size parity tests spent more than half of the GPU suite on the CPU.  The oracle's
side of those tests is deterministic in (workload generator, oracle code), so it is computed ONCE in the build container by
``python tests/golden/make_oracle_golden.py`` and the GPU tests load it; the reference never travels (these are outputs of oracle/,
the repo's code) and nothing here is on the product path.

Safety against stale fixtures.  A fixture stores the SHA-1 of its inputs (the rows of the synthetic workload it was computed for and
a fingerprint of the weights): ``get()`` regenerates the workload, compares the digest and refuses a fixture that does not belong to
today's generator.  ``tests/test_oracle.py::test_oracle_golden_is_what_the_oracle_computes`` recomputes two cases on the CPU and
compares them with the committed files, ``::test_oracle_golden_emulation_is_what_the_emulation_computes`` does so for the rounding
emulation's arrays of cfg3; ``::test_oracle_golden_fixtures_match_their_workloads`` checks every digest; and
``::test_oracle_sources_are_the_ones_the_fixtures_were_computed_with`` compares a hash of oracle/*.py with the one recorded when the
fixtures were made (the input digest does not see a change of the oracle's CODE).  A missing fixture is computed on the
spot (slow, never wrong).
"""
from __future__ import annotations

import hashlib
import os

import numpy as np
import torch

from oracle import planner as op
from paule_amd import synthetic

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# case -> workload (batch, frames, model set or explicit models), the utterances the oracle runs on, what it records
CASES = {
    # BASELINE configs[4]'s length on Paule's models, two utterances (tests: test_long_sequences_set_a_vs_oracle)
    "long_set_a": dict(B=2, T=2000, set="A", rows=[0, 1], objective="acoustic_semvec", iters=1, grad=True),
    # T = 2000 on small stacked models, f32 bars (test_long_sequences_f32_vs_oracle)
    "long_small": dict(B=2, T=2000, set=None, pred=dict(num_lstm_layers=2, hidden_size=64), emb=dict(num_lstm_layers=1, hidden_size=96),
                       rows=[0, 1], objective="acoustic_semvec", iters=2),
    # cfg3 (B = 256 x 300, set A): first and last utterance, three iterations (bf16 test uses 3, the f32 test the first 2)
    "cfg3_rows": dict(B=256, T=300, set="A", rows=[0, 255], objective="acoustic_semvec", iters=3, emul=True),
    "cfg3_100_rows": dict(B=100, T=300, set="A", rows=[0, 99], objective="acoustic_semvec", iters=3),
    # cfg2 (B = 64 x 300, `acoustic`, predictor only)
    "cfg2_rows": dict(B=64, T=300, set="A", rows=[0, 1], objective="acoustic", iters=3, with_emb=False),
    # BASELINE configs[4] on ONE GPU (cfg5_128: B = 128 x 2000): first and last utterance, one iteration + the rounding emulation
    "cfg5_128_rows": dict(B=128, T=2000, set="A", rows=[0, 127], objective="acoustic_semvec", iters=1, grad=True, emul="dX"),
    # BASELINE configs[3]'s whole batch on ONE GPU (cfg4_1gpu: B = 2048 x 300)
    "cfg4_rows": dict(B=2048, T=300, set="A", rows=[0, 2047], objective="acoustic_semvec", iters=2),
}

_CACHE: dict = {}


def workload(name):
    c = CASES[name]
    if c["set"]:
        return synthetic.make_workload(c["B"], c["T"], c["set"])
    return synthetic.make_workload(c["B"], c["T"], None, pred=c["pred"], emb=c["emb"])


def _digest(c, wl):
    h = hashlib.sha1()
    rows = c["rows"]
    for t in (wl.cp0[rows], wl.target_mel[rows], wl.target_semvec[rows]):
        h.update(np.ascontiguousarray(t.double().numpy()).tobytes())
    for sd in (wl.pred_sd, wl.emb_sd):
        for k in sorted(sd):   # a fingerprint of every weight tensor: a strided sample of its entries (no reductions: a sum's last
            v = sd[k].double().reshape(-1)   # bits depend on the thread count of the machine that adds it up)
            h.update(k.encode())
            h.update(np.ascontiguousarray(v[:: max(1, v.numel() // 61)].numpy()).tobytes())
            h.update(np.array([float(v[0]), float(v[-1]), float(v.numel())]).tobytes())
    h.update(repr((c["objective"], c["iters"], c["B"], c["T"])).encode())
    return h.hexdigest()


def path(name):
    return os.path.join(GOLDEN_DIR, f"oracle_{name}.npz")


def bf16_bits(x):
    """bf16-valued float array -> its 16-bit patterns (the emulation's stashes are bf16 values held in float64)."""
    return (np.asarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def bf16_from_bits(b):
    return (np.asarray(b, dtype=np.uint32) << 16).view(np.float32)


def compute(name, wl=None):
    """Runs the oracle (and, where the case asks for it, the rounding emulation) on the case's utterances."""
    from oracle import manual as mo
    c = CASES[name]
    wl = wl or workload(name)
    rows = c["rows"]
    with_emb = c.get("with_emb", True)
    orc = op.OraclePlanner(op.forward_model_from_state_dict(wl.pred_sd), op.embedding_model_from_state_dict(wl.emb_sd) if with_emb else None,
                           objective=c["objective"])
    orc.set_targets(wl.target_mel[rows], wl.target_semvec[rows] if with_emb else None)
    orc.set_cp(wl.cp0[rows])
    losses, cps = [], []
    out = {}
    for k in range(c["iters"]):
        losses.append(orc.step(1).numpy())
        cps.append(orc.get_cp().numpy().astype(np.float32))   # compared at >= 1e-5: float32 storage is 6e-8
        if k == 0 and c.get("grad"):   # model part of dL/dCP at the first iteration
            out["g_model"] = (orc.last_grad.numpy() - mo.smoothness_loss_grad(wl.cp0[rows].numpy())[3]).astype(np.float32)
    out["loss"] = np.concatenate(losses, axis=0)          # [iters, rows, columns]
    out["cp_after"] = np.stack(cps)                        # [iters, rows, T, 30]
    out.update(compute_emul(name, wl))
    out["digest"] = np.array(_digest(c, wl))
    out["rows"] = np.array(rows)
    return out


def compute_emul(name, wl=None):
    """The rounding emulation's part of a case (oracle/bf16_emul.py: the emul_* arrays), on its own: cheap enough to be recomputed on
    the CPU in tests/test_oracle.py, so that a change of the emulation cannot leave a stale fixture behind."""
    c = CASES[name]
    out = {}
    if not c.get("emul"):
        return out
    from oracle import bf16_emul as be
    wl = wl or workload(name)
    rows = c["rows"]
    em = be.EmulPlanner(wl.pred_sd, wl.emb_sd, objective=c["objective"])
    em.set_targets(wl.target_mel[rows].numpy(), wl.target_semvec[rows].numpy())
    em.set_cp(wl.cp0[rows].numpy())
    _, _, pe = be.loss_and_grad(em.models, c["objective"], em.x, em.target_mel, em.target_semvec)
    if c["emul"] is True:
        out["emul_pred_h0_bits"] = bf16_bits(pe["pred_h"][0])   # [rows, T, H] bf16 patterns of the predictor's h stash
    out["emul_dX"] = np.asarray(pe["dX"], dtype=np.float32)
    return out


ORACLE_FILES = ("planner.py", "manual.py", "bf16_emul.py")


def oracle_sources_sha1():
    """SHA-1 over the oracle's source files (line endings and trailing blanks normalised).  tests/golden/oracle_sources.sha1 records the
    value the committed fixtures were computed with: a change of the oracle's code fails tests/test_oracle.py until the fixtures are
    regenerated (python tests/golden/make_oracle_golden.py, which rewrites the record)."""
    h = hashlib.sha1()
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    for f in ORACLE_FILES:
        with open(os.path.join(root, f), "rb") as fh:
            for line in fh.read().decode().splitlines():
                h.update(line.rstrip().encode() + b"\n")
    return h.hexdigest()


SOURCES_RECORD = os.path.join(GOLDEN_DIR, "oracle_sources.sha1")


def get(name):
    """The oracle's outputs for a case: the committed fixture if it belongs to today's workload, else computed now."""
    if name in _CACHE:
        return _CACHE[name]
    wl = workload(name)
    p = path(name)
    if os.path.exists(p):
        z = dict(np.load(p))
        want = _digest(CASES[name], wl)
        if str(z["digest"]) != want:
            raise AssertionError(f"{p} was computed for other inputs (digest {z['digest']} != {want}): the workload generator or the case "
                                 "changed -- regenerate with `python tests/golden/make_oracle_golden.py`")
        _CACHE[name] = z
        return z
    _CACHE[name] = compute(name, wl)
    return _CACHE[name]


def rows_of(name):
    return list(CASES[name]["rows"])


def check_digest(name):
    z = dict(np.load(path(name)))
    return str(z["digest"]) == _digest(CASES[name], workload(name))

