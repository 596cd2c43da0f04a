#!/usr/bin/env python3
"""Writes tests/golden/oracle_<case>.npz: what this repository's CPU oracle (oracle/planner.py, oracle/manual.py,
oracle/bf16_emul.py) computes on the utterances the full-size GPU parity tests compare (tests/oracle_golden.py: CASES).

usage: python tests/golden/make_oracle_golden.py [case ...]        (no argument: every case; takes ~15 minutes of CPU)

Runs anywhere the repo runs (it does not touch /root/reference: the oracle is the repo's own restatement, pinned to the
reference's outputs by tests/test_oracle.py).  The GPU tests load the files instead of spending the GPU box's time on float64
CPU arithmetic; tests/test_oracle.py recomputes one case and compares it with the committed file."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_golden as og  # noqa: E402

if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    names = args or ([] if "--record-sources" in sys.argv else list(og.CASES))
    if not args:   # all cases (or --record-sources alone, when the oracle's text changed but not its arithmetic): the fixtures belong to today's oracle code
        with open(og.SOURCES_RECORD, "w") as fh:
            fh.write(og.oracle_sources_sha1() + "  oracle/" + ",".join(og.ORACLE_FILES) + "\n")
        print("recorded", og.SOURCES_RECORD)
    for name in names:
        t0 = time.time()
        out = og.compute(name)
        np.savez_compressed(og.path(name), **out)
        kb = os.path.getsize(og.path(name)) / 1024
        print(f"{name}: {time.time() - t0:.1f} s, {kb:.0f} KB  ({', '.join(f'{k}{list(v.shape)}' for k, v in out.items() if k not in ('digest', 'rows'))})", flush=True)
