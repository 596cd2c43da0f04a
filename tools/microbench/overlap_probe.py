#!/usr/bin/env python3
"""Feasibility probe: does a big bf16 GEMM running on a second stream slow a persistent LSTM sweep down (and vice versa)?
Sweep = the predictive model's forward / backward sweep (pl_bench_kernel, 184 of 256 CUs); GEMM = torch.matmul of the
embedder's layer-2 input projection shape (38400 x 736 @ 736 x 2944, bf16) on another stream."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

wl = synthetic.make_workload(256, 300, "A")
eng = HipPlanner(wl.pred_sd, wl.emb_sd, batch=256, n_frames=300, objective="acoustic_semvec", dtype="bf16")
eng.set_targets(wl.target_mel, wl.target_semvec)
eng.set_cp(wl.cp0)
eng.step(2, return_loss=False)
eng.synchronize()
side = torch.cuda.Stream()
a = torch.randn(38400, 736, device="cuda", dtype=torch.bfloat16)
b = torch.randn(736, 2944, device="cuda", dtype=torch.bfloat16)


def gemms(n):
    with torch.cuda.stream(side):
        for _ in range(n):
            torch.matmul(a, b)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, out


gemms(3)
for kind in ("fwd_sweep", "bwd_sweep"):
    t_sweep, (ms, _) = timed(lambda: eng.bench_kernel(kind, "pred", reps=10))
    t_gemm, _ = timed(lambda: gemms(40))
    def both():
        gemms(40)
        return eng.bench_kernel(kind, "pred", reps=10)
    t_both, (ms_b, _) = timed(both)
    print(f"{kind}: alone {ms:.3f} ms/launch (10 launches {t_sweep:.1f} ms); 40 GEMMs alone {t_gemm:.1f} ms ({t_gemm / 40 * 1e3:.0f} us each); "
          f"together: sweep {ms_b:.3f} ms/launch, wall {t_both:.1f} ms (serial sum {t_sweep + t_gemm:.1f} ms)")
