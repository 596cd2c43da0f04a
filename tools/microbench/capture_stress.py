#!/usr/bin/env python3
"""Stress of the multi-stream graph capture of the layer wavefront: many handles, each capturing once (usage: capture_stress.py [reps] [off])."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from paule_amd import synthetic  # noqa: E402
from paule_amd.engine import HipPlanner  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
off = len(sys.argv) > 2 and sys.argv[2] == "off"   # wavefront off: single-stream captures only
shapes = [(2, 64, "A", "f32"), (5, 61, "B", "bf16"), (140, 24, "B", "bf16"), (3, 37, "B", "f32"), (1, 40, "A", "f32"), (20, 30, "C", "bf16")]
wls = {s: synthetic.make_workload(s[0], s[1], s[2]) for s in shapes}
n = 0
hold = []
for r in range(reps):
    for s in shapes:
        for chunks in ("4", "7", "2"):
            os.environ["PAULE_HIP_WAVEFRONT"] = "0" if off else chunks
            wl = wls[s]
            e = HipPlanner(wl.pred_sd, wl.emb_sd, batch=s[0], n_frames=s[1], objective="acoustic_semvec", dtype=s[3])
            e.set_targets(wl.target_mel, wl.target_semvec)
            e.set_cp(wl.cp0)
            e.step(2)
            e.synchronize()
            n += 1
            if n % 7 == 0:
                hold.append(e)      # some handles stay alive, as after a failed test
            if len(hold) > 6:
                hold.pop(0)
    print(f"rep {r}: {n} captures ok", flush=True)
print("all ok")
