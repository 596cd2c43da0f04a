#!/usr/bin/env python3
"""Summarises the rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs as the MI355X
guide prescribes: FETCH_SIZE takes 3 of the 4 TCC slots) into HBM bytes per launch for the dominant kernels.
gfx950 corrections (MI355X_MICROARCH.md, HBM): counters are in KiB; FETCH_SIZE reads exactly 1/2 of the bytes of a
wide (16 B / lane) coalesced stream -> doubled; WRITE_SIZE is exact.
usage: pmc_summary.py <dir_fetch> <dir_write> <config> [out.json]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def is_kernel(recorded, kname):
    """EXACT function name (a substring match made fused_bwd_kernel also collect fused_bwd_kernel2): the name as a whole
    identifier in a demangled signature, or as a length-prefixed component of a mangled one."""
    if recorded.startswith("_Z"):
        return re.search(r"%d%s(?=[IE])" % (len(kname), re.escape(kname)), recorded) is not None
    return re.search(r"(?<![A-Za-z0-9_])%s(?![A-Za-z0-9_])" % re.escape(kname), recorded) is not None


def per_kernel(dirname, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def instantiation(recorded, kname):
    """The template arguments of a recorded kernel name as text ("<46, 0, 1>", "" for none): launches of different instantiations of one
    kernel are different kernels (the predictor's backward sweep with its ride-along tile, the embedder layers' without) and are
    reported separately (no more "upper third by value" of a mixed population)."""
    if recorded.startswith("_Z"):   # rocprofv3 leaves some names mangled: builtin types (f, d, DF16b = __bf16) and integer literals (Li46E)
        m = re.search(r"%d%sI((?:DF16b|f|d|L[a-z]\d+E)+)E" % (len(kname), re.escape(kname)), recorded)
        if not m:
            return ""
        names = {"DF16b": "bf16", "f": "float", "d": "double"}
        toks = re.findall(r"DF16b|f|d|L[a-z]\d+E", m.group(1))
        return "<" + ", ".join(names.get(t) or re.sub(r"L[a-z](\d+)E", r"\1", t) for t in toks) + ">"
    m = re.search(r"(?<![A-Za-z0-9_])%s(<[^>]*>)" % re.escape(kname), recorded)
    return m.group(1).replace("bool _Accum", "bf16") if m else ""   # (the profiler's demangler prints __bf16 as "bool _Accum")


def main():
    dfetch, dwrite, config = sys.argv[1], sys.argv[2], sys.argv[3]
    out_path = sys.argv[4] if len(sys.argv) > 4 else None
    fetch, write = per_kernel(dfetch, "FETCH_SIZE"), per_kernel(dwrite, "WRITE_SIZE")
    res = {}
    for kname, key in (("fused_fwd_kernel", "fused_fwd_kernel"), ("fused_fwd2_kernel", "fused_fwd2_kernel"), ("fused_bwd_kernel", "fused_bwd_kernel"),
                       ("fused_fwd16_kernel", "fused_fwd16_kernel"), ("fused_bwd16_kernel", "fused_bwd16_kernel"),
                       ("fused_bwd_kernel2", "fused_bwd_kernel2"), ("fused_bwd16_kernel2", "fused_bwd16_kernel2"),
                       ("lstm_bwd16_rs_sweep_kernel", "lstm_bwd16_rs_sweep_kernel"), ("lstm_fwd16_sweep_kernel", "lstm_fwd16_sweep_kernel"),
                       ("lstm_bwd_sweep_f32_kernel", "lstm_bwd_sweep_f32_kernel"), ("lstm_fwd_sweep_f32_kernel", "lstm_fwd_sweep_f32_kernel"),
                       ("lstm_bwd_rs_stream_kernel", "lstm_bwd_rs_stream_kernel"),
                       ("lstm_bwd_rs_sweep_kernel", "lstm_bwd_rs_sweep_kernel"), ("lstm_bwd_sweep_kernel", "lstm_bwd_sweep_kernel"),
                       ("lstm_fwd_sweep_kernel", "lstm_fwd_sweep_kernel"),
                       ("lstm_bwd_step_kernel", "lstm_bwd_step_kernel"), ("lstm_fwd_step_kernel", "lstm_fwd_step_kernel"),
                       ("lstm_fwd_chain_f32_kernel", "lstm_fwd_chain_f32_kernel"), ("lstm_bwd_chain_f32_kernel", "lstm_bwd_chain_f32_kernel"),
                       ("gemm_nt_big_kernel", "gemm_nt_big_kernel"), ("dx_reduce_kernel", "dx_reduce_kernel"), ("gemm_nt_kernel", "gemm_nt_kernel")):
        insts = sorted({instantiation(k, kname) for k in list(fetch) + list(write) if is_kernel(k, kname)})
        if not insts:
            continue
        per_inst = {}
        for inst in insts:
            fv = [v for k, vs in fetch.items() if is_kernel(k, kname) and instantiation(k, kname) == inst for v in vs]
            wv = [v for k, vs in write.items() if is_kernel(k, kname) and instantiation(k, kname) == inst for v in vs]
            # launches of ONE instantiation still differ in length (the embedder's layers run T' = T / 2 steps, bench_kernel repeats the predictor's
            # sweep): the launches of the most frequent size class (within 10 % of the median) stand for it
            def typical(vals):
                if not vals:
                    return 0.0, 0
                vals = sorted(vals)
                med = vals[len(vals) // 2]
                cls = [v for v in vals if abs(v - med) <= 0.1 * med] or vals
                return sum(cls) / len(cls), len(vals)
            f_avg, nf = typical(fv)
            w_avg, nw = typical(wv)
            fb, wb = 2.0 * 1024.0 * f_avg, 1024.0 * w_avg
            per_inst[inst] = (fb, wb, max(nf, nw))
            tag = key + inst
            res[tag + "_bytes_per_launch"] = fb + wb
            res[tag + "_fetch_bytes"] = fb
            res[tag + "_write_bytes"] = wb
            res[tag + "_launches"] = max(nf, nw)
            print(f"{kname + inst:40s} launches {max(nf, nw):5d}  fetch {fb / 1e6:10.2f} MB  write {wb / 1e6:10.2f} MB per launch (median size class)")
        # the un-suffixed key bench.py reads: the instantiation that moves the most bytes per launch (cfg3: the predictor's sweep)
        top = max(per_inst, key=lambda i: per_inst[i][0] + per_inst[i][1])
        fb, wb, n = per_inst[top]
        res[key + "_bytes_per_launch"] = fb + wb
        res[key + "_fetch_bytes"] = fb
        res[key + "_write_bytes"] = wb
        res[key + "_launches"] = n
        res[key + "_instantiation"] = top
    if out_path:
        data = {}
        if os.path.exists(out_path):
            data = json.load(open(out_path))
        data[config] = res
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench   # noqa: E402  (source_digest: the kernel sources these counters were taken on)
        data["source_digest"] = bench.source_digest()
        json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
