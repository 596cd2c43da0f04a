#!/usr/bin/env python3
"""Diagnostic (-DPL_STAMPS build): per-phase time of one step of the persistent LSTM sweeps, averaged over the
steps of the last pred-model backward sweep and the last forward sweep (= embedder layer 1, T' steps)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PAULE_HIP_LIB"] = os.path.join(ROOT, "paule_amd", "csrc", os.environ.get("PL_STAMP_LIB", "libpaule_hip_stamps.so"))
os.environ.setdefault("PL_STAMP_FILE", os.path.join(ROOT, "gpurun_out", "stamps"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
import numpy as np  # noqa: E402
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
DTYPE = sys.argv[3] if len(sys.argv) > 3 else "bf16"
OBJ = sys.argv[2] if len(sys.argv) > 2 else "acoustic_semvec"   # "acoustic": the last forward sweep is the predictive model's (fused input)
wl = synthetic.make_workload(B, 300, "A")
eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=300, objective=OBJ, dtype=DTYPE)
eng.set_targets(wl.target_mel, wl.target_semvec)
eng.set_cp(wl.cp0)
eng.step(3)
eng.synchronize()
raw = np.fromfile(os.environ["PL_STAMP_FILE"] + ".sweep", dtype=np.uint64).reshape(2, 256, 8).astype(np.float64) * 0.01
for d, name, steps, labels in (
        (0, "forward sweep (embedder layer 1, 150 steps)" if OBJ != "acoustic" else "forward sweep (pred model, fused input, 300 steps)",
         150 if OBJ != "acoustic" else 300,
         ["step top/prefetch", "wait arrivals", "h tile sc1 loads+LDS", "MFMA chain", "cell+store issue", "store drain", "barrier+add"]),
        (1, "backward sweep (pred model, 300 steps), mode " + os.environ.get("PAULE_HIP_BWD_MODE", "1"), 300,
         (["stash loads issued", "tile loads issued + landed", "sums", "cell+stash+dA image+barrier", "tiles", "token checks", "-", "stash loads landed (diagnostic wait)"]
          if "stamps2" in os.environ.get("PL_STAMP_LIB", "") else
          ["step top/stash loads", "tile loads + token checks", "sums", "cell+stash+dA image+barrier", "tiles", "SWEEPS per step (count)",
           "TILE LOADS per step (count)"] if os.environ.get("PAULE_HIP_BWD_STREAM", "1") == "2" else
          ["step top/stash loads issued", "polls + tile loads issued", "tiles landed + sums", "cell+stash+dA image+barrier", "tiles + flags"]
          if os.environ.get("PAULE_HIP_BWD_STREAM", "1") == "1" and os.environ.get("PAULE_HIP_BWD_WAVES", "8") != "4" else
          ["step top/prefetch", "wait arrivals", "partial ingest", "cell+stash+dA image", "MFMA+partial image", "hand-off store issue",
           "drain+barrier+add"]) if os.environ.get("PAULE_HIP_BWD_MODE", "1") == "1" else
         ["step top/prefetch", "wait arrivals", "dA loads+LDS+MFMA", "partial reduce", "cell+store issue", "store drain", "barrier+add"])):
    blk = raw[d]
    used = blk[blk.sum(axis=1) > 0]
    print(f"--- {name}: {len(used)} workgroups; per-step phase time (us), median / max over workgroups")
    tot = 0.0
    chains = max(1, int(os.environ.get("PAULE_HIP_BWD_CHAINS", "0"))) if d == 1 else 1   # chained form: a workgroup's stamps cover steps x chains chain-steps
    for i, lab in enumerate(labels):
        v = used[:, i] / steps / chains
        tot += np.median(v)
        print(f"  {lab:24s} {np.median(v):6.2f} {v.max():6.2f}")
    print(f"  {'sum':24s} {tot:6.2f}")
