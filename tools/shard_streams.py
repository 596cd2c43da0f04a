#!/usr/bin/env python3
"""Experiment: does splitting the batch into S independent engines on S streams (their GEMM phases then overlap other
shards' latency-bound LSTM sweeps) beat one engine?  usage: shard_streams.py [S ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

B, T, ITERS, ROUNDS = 256, 300, 10, 5
wl = synthetic.make_workload(B, T, "A")
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    streams = [torch.cuda.Stream() for _ in range(S)]
    engs = []
    for k, st in enumerate(streams):
        sl = slice(k * B // S, (k + 1) * B // S)
        with torch.cuda.stream(st):
            e = HipPlanner(wl.pred_sd, wl.emb_sd, batch=B // S, n_frames=T, objective="acoustic_semvec", dtype="bf16")
            e.set_targets(wl.target_mel[sl], wl.target_semvec[sl])
            e.set_cp(wl.cp0[sl])
            e.step(2, return_loss=False)
        engs.append(e)
    torch.cuda.synchronize()
    ts = []
    for r in range(ROUNDS):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(ITERS):          # interleave the launches so that every stream always has work queued
            for e, st in zip(engs, streams):
                with torch.cuda.stream(st):
                    e.step(1, return_loss=False)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / ITERS * 1e3)
    for e in engs:
        e.synchronize()
    print(f"S={S}: median {np.median(ts):.3f} ms per iteration of all {B} utterances (min {min(ts):.3f})", flush=True)
    del engs
