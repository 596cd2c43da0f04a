#!/usr/bin/env python3
"""Diagnostic: where does one LSTM step launch spend its time?  Uses the -DPL_STAMPS build
(make -C paule_amd/csrc stamps), whose step kernels record s_memrealtime (100 MHz) at: entry, DMA prologue
issued, first stage landed, product done, epilogue done.  Never used for the shipped numbers."""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PAULE_HIP_LIB"] = os.path.join(ROOT, "paule_amd", "csrc", "libpaule_hip_stamps.so")
os.environ.setdefault("PL_STAMP_FILE", os.path.join(ROOT, "gpurun_out", "stamps"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)

import numpy as np  # noqa: E402
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
wl = synthetic.make_workload(B, 300, "A")
eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B, n_frames=300, objective="acoustic_semvec", dtype=dtype)
eng.set_targets(wl.target_mel, wl.target_semvec)
eng.set_cp(wl.cp0)
eng.step(2)
for name in ("fwd", "bwd"):
    ms, fl = eng.bench_kernel(name, "pred", reps=200)
    raw = open(os.environ["PL_STAMP_FILE"] + "." + name, "rb").read()
    nblk, reps = struct.unpack("ii", raw[:8])
    st = np.frombuffer(raw[8:], dtype=np.uint64).reshape(reps, nblk, 8).astype(np.float64) * 0.01  # -> us
    st = st[20:]                                           # steady state
    t0 = st[:, :, 0]
    first = t0.min(axis=1, keepdims=True)
    print(f"--- {name} step kernel ({dtype}, B={B}): {ms * 1e3:.2f} us per launch (events), {nblk} blocks")
    print("  block start skew after first block (us): med %.2f  p90 %.2f  max %.2f" % tuple(
        np.percentile(t0 - first, [50, 90, 100])))
    names = ["entry->DMA issued", "DMA issued->stage 0 landed", "stage 0 landed->product done", "product done->end"]
    for i, nm in enumerate(names):
        d = st[:, :, i + 1] - st[:, :, i]
        print("  %-32s med %.2f  p90 %.2f  max %.2f" % (nm, *np.percentile(d, [50, 90, 100])))
    tot = st[:, :, 4] - st[:, :, 0]
    print("  block lifetime                   med %.2f  p90 %.2f  max %.2f" % tuple(np.percentile(tot, [50, 90, 100])))
    span = st[:, :, 4].max(axis=1) - first[:, 0]
    print("  launch span (first entry -> last end) med %.2f us" % np.median(span))
    gap = first[1:, 0] - st[:-1, :, 4].max(axis=1)
    print("  gap last end -> next first entry  med %.2f us" % np.median(gap))
    xcd = st[0, :, 5] / 0.01
    print("  blocks per XCD:", np.bincount(xcd.astype(int) & 15, minlength=8).tolist())
