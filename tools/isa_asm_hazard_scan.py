#!/usr/bin/env python3
"""Scan gfx950 device assembly for the software-managed hazards that inline asm can break (round 5).

hipcc pads its OWN instruction stream for the hazards the hardware does not interlock, but it does not look inside an `asm volatile` block,
and this library issues its LDS-DMA from such blocks (s_mov_b32 m0, <lds base> ; s_nop 0 ; global_load_lds_dwordx4 v, s[a:b] ; s_mov_b32 m0, <saved>):
whether the instructions AROUND the block leave the block's reads enough distance is nobody's job but ours.  Checked on every control-flow path
(the CFG walk of tools/isa_mfma_hazard_scan.py, wait states counted the compiler's way: every instruction 1, s_nop N = N + 1):

  H1  SALU writes M0                 -> an instruction that reads M0 (LDS-DMA loads, GWS, s_movrel / v_movrel, s_sendmsg, lane selects by m0)   1 wait state
  H2  VALU writes an SGPR / VCC      -> a VMEM instruction that reads that SGPR (resource, scalar offset, scalar base address)                   5 wait states
  H3  VALU writes an SGPR / VCC      -> v_readlane / v_writelane with that SGPR as the lane select                                              4 wait states
  H4  VALU writes VCC                -> v_div_fmas                                                                                              4 wait states

(figures: the GCN3 / CDNA ISA guides' "manually inserted wait states" table, the same LLVM's GCNHazardRecognizer pads for.)  A SALU write of
the register in between ends the VALU write's claim on it.

usage: isa_asm_hazard_scan.py file.s [file.s ...]      exit code 1 if anything is reported
       isa_asm_hazard_scan.py --build                  compile every paule_amd/csrc/*.hip to /tmp/pl_isa/*.s first, then scan them"""
import importlib.util
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("isa_mfma_scan", os.path.join(ROOT, "tools", "isa_mfma_hazard_scan.py"))
_mfma = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mfma)
functions = _mfma.functions

SREG = re.compile(r"\b(?:s\[(\d+):(\d+)\]|s(\d+)\b|(vcc_lo|vcc_hi|vcc|m0))")
VMEM = ("buffer_", "global_", "flat_", "scratch_", "tbuffer_", "image_")
M0_READERS = ("global_load_lds", "ds_gws", "s_movrel", "v_movrel", "s_sendmsg", "v_interp", "ds_param_load", "ds_direct_load")


def sregs(tok):
    out = set()
    for m in SREG.finditer(tok):
        if m.group(3) is not None:
            out.add("s%d" % int(m.group(3)))
        elif m.group(1) is not None:
            out.update("s%d" % r for r in range(int(m.group(1)), int(m.group(2)) + 1))
        elif m.group(4) == "vcc":
            out.update(("vcc_lo", "vcc_hi"))
        else:
            out.add(m.group(4))
    return out


def split_ops(ops):
    return [x.strip() for x in ops.split(",")] if ops else []


def valu_scalar_dst(mn, ops):
    """SGPRs / VCC halves a VALU instruction writes."""
    if not mn.startswith("v_"):
        return set()
    o = split_ops(ops)
    out = set()
    if o and not o[0].startswith(("v", "a")) and SREG.fullmatch(o[0]):
        out |= sregs(o[0])                                   # v_readfirstlane / v_readlane / v_cmp_*_e64 sdst, ...
    if len(o) > 1 and ("_co_" in mn or mn.startswith("v_div_scale") or mn.startswith("v_mad_u64_u32") or mn.startswith("v_mad_i64_i32")) and SREG.fullmatch(o[1]):
        out |= sregs(o[1])                                   # carry-out / scale flag
    if (mn.startswith("v_cmp") and not mn.endswith("_e64")) or mn.startswith("v_cmpx"):
        out |= {"vcc_lo", "vcc_hi"}                          # e32 compares write VCC implicitly
    return out


def reads_m0(mn, ops):
    if mn.startswith(M0_READERS) or ("_lds" in mn and mn.startswith(VMEM)) or " lds" in (" " + ops):
        return True
    return mn in ("v_readlane_b32", "v_writelane_b32") and split_ops(ops)[-1:] == ["m0"]


def build_cfg(lines):
    ins, label_at = [], {}
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = len(ins)
            continue
        t = l.split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        parts = t.split(None, 1)
        ins.append((t, parts[0], parts[1] if len(parts) > 1 else ""))
    n = len(ins)

    def succ(k):
        t, mn, ops = ins[k]
        if mn == "s_endpgm":
            return []
        if mn == "s_branch":
            return [label_at[ops.strip()]] if ops.strip() in label_at else []
        if mn.startswith("s_cbranch"):
            tgt = ops.split(",")[-1].strip()
            out = [k + 1] if k + 1 < n else []
            if tgt in label_at:
                out.append(label_at[tgt])
            return out
        if mn in ("s_setpc_b64", "s_swappc_b64"):
            return []
        return [k + 1] if k + 1 < n else []
    return ins, succ


def scan_function(name, lines):
    ins, succ = build_cfg(lines)
    n = len(ins)
    findings = []

    def walk(k, regs_live, need, hit):
        """From the instruction behind k: every path until `need` wait states have passed; hit(q, live) -> text of a finding or None."""
        best = {}
        stack = [(s, 0, frozenset(regs_live)) for s in succ(k)]
        while stack:
            q, ws, live = stack.pop()
            if ws >= need or q >= n or not live:
                continue
            key = (q, live)
            if key in best and best[key] <= ws:
                continue
            best[key] = ws
            qt, qmn, qops = ins[q]
            why = hit(qmn, qops, live)
            if why:
                findings.append((name, k, ins[k][0], q, qt, f"{why}: {ws} wait state(s) on the shortest path, {need} required"))
                continue
            if qmn.startswith("s_") and not qmn.startswith(("s_cbranch", "s_branch", "s_nop", "s_waitcnt", "s_barrier", "s_cmp", "s_bitcmp")):
                o = split_ops(qops)
                if o:
                    live = live - sregs(o[0])               # a SALU write takes the register over
            step = int(qops.strip(), 0) + 1 if qmn == "s_nop" else 1
            for s in succ(q):
                stack.append((s, ws + step, live))

    for k, (t, mn, ops) in enumerate(ins):
        o = split_ops(ops)
        if mn.startswith("s_") and o and o[0] == "m0" and not mn.startswith(("s_cmp", "s_bitcmp")):
            walk(k, {"m0"}, 1, lambda qmn, qops, live: "H1 M0 read behind a SALU write of M0" if reads_m0(qmn, qops) else None)
        w = valu_scalar_dst(mn, ops)
        if w:
            walk(k, w, 5, lambda qmn, qops, live: ("H2 VMEM reads an SGPR a VALU instruction wrote (" + ", ".join(sorted(sregs(qops) & live)) + ")")
                 if qmn.startswith(VMEM) and (sregs(qops) & live) else None)
            walk(k, w, 4, lambda qmn, qops, live: "H3 lane select written by a VALU instruction"
                 if qmn in ("v_readlane_b32", "v_writelane_b32") and (sregs(split_ops(qops)[-1]) & live) else None)
            if w & {"vcc_lo", "vcc_hi"}:
                walk(k, w & {"vcc_lo", "vcc_hi"}, 4, lambda qmn, qops, live: "H4 v_div_fmas behind a VALU write of VCC" if qmn.startswith("v_div_fmas") else None)
    seen, out = set(), []
    for f in findings:
        key = (f[1], f[3], f[5][:2])
        if key not in seen:
            seen.add(key)
            out.append(f)
    n_dma = sum(1 for i in ins if i[1].startswith("global_load_lds"))
    n_m0 = sum(1 for i in ins if i[1].startswith("s_") and split_ops(i[2])[:1] == ["m0"])
    return out, n_dma, n_m0


def main():
    args = sys.argv[1:]
    files = [a for a in args if not a.startswith("--")]
    if "--build" in args:
        os.makedirs("/tmp/pl_isa", exist_ok=True)
        src = os.path.join(ROOT, "paule_amd", "csrc")
        for f in sorted(os.listdir(src)):
            if f.endswith(".hip"):
                out = f"/tmp/pl_isa/{f[:-4]}.s"
                subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-w",
                                       "-S", "--cuda-device-only", "-o", out, os.path.join(src, f)])
                files.append(out)
    bad = 0
    for path in files:
        n_fn = n_dma = n_m0 = 0
        for name, lines in functions(path):
            res, dma, m0 = scan_function(name, lines)
            n_fn += 1
            n_dma += dma
            n_m0 += m0
            for (fn, k, t, q, qt, why) in res:
                bad += 1
                print(f"{os.path.basename(path)}: {fn}\n    writer #{k}: {t}\n    reader #{q}: {qt}\n    -> {why}")
        print(f"{os.path.basename(path)}: {n_fn} functions, {n_dma} LDS-DMA loads, {n_m0} writes of M0 scanned")
    print("FINDINGS:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
