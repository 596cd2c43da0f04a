#!/usr/bin/env python3
"""Scan gfx950 device assembly for MFMA results that are read too early on SOME control-flow path.

Background (profiles/r04_isa_stale_accumulator.txt): on CDNA the matrix pipe does not interlock -- between a
v_mfma and the first non-accumulate read of its destination registers the instruction stream itself must provide a minimum number
of wait states (gfx950: passes + 4 for XDL ops, i.e. 8 for v_mfma_f32_16x16x32_bf16 / 16x16x4_f32, 12 for 32x32x16_bf16; the
figures of LLVM's GCNHazardRecognizer).  hipcc pads straight-line code with s_nop, but round 3 found a kernel (fused_bwd16_kernel
at 81e779f^) in which a conditional branch directly behind an MFMA led, when TAKEN, straight to the v_accvgpr_read of that MFMA's
result: one wait state instead of eight, and one run in four read a stale accumulator register.

This tool rebuilds the control-flow graph of every kernel in a `.s` file (hipcc -S --cuda-device-only), and for every MFMA walks all
paths forward until a non-MFMA-accumulate instruction touches a destination register, counting wait states the way the compiler
does (every instruction 1, s_nop N = N + 1; a branch counts 1 whether taken or not -- conservative).  It reports every (MFMA, reader)
pair whose shortest path is below the requirement.

usage: isa_mfma_hazard_scan.py file.s [file.s ...]        exit code 1 if anything is reported
       isa_mfma_hazard_scan.py --build                    compile every paule_amd/csrc/*.hip to /tmp/pl_isa/*.s first, then scan them"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQ = {  # wait states between an MFMA and a VALU / memory / LDS read (or a non-MFMA write) of its result on gfx950
    "v_mfma_f32_16x16x32_bf16": 8, "v_mfma_f32_16x16x32_f16": 8, "v_mfma_f32_16x16x4_f32": 8, "v_mfma_f32_16x16x16_bf16": 8,
    "v_mfma_f32_32x32x16_bf16": 12, "v_mfma_f32_32x32x16_f16": 12, "v_mfma_f32_32x32x2_f32": 20, "v_mfma_f32_32x32x8_bf16": 12,
}
REG = re.compile(r"\b([av])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(4) is not None:
            out.add((m.group(1), int(m.group(4))))
        else:
            out.update((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def functions(path):
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z[\w$.]+):", lines[i])
        if m:
            j = i + 1
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            yield m.group(1), lines[i + 1:j]
            i = j
        i += 1


def scan_function(name, lines):
    # instruction list with block structure
    ins = []          # (text, mnemonic, operands)
    label_at = {}
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

    findings = []
    for k, (t, mn, ops) in enumerate(ins):
        base = mn.replace("_e64", "")
        if not base.startswith("v_mfma") and not base.startswith("v_smfmac"):
            continue
        need = REQ.get(base)
        if need is None:
            findings.append((name, k, t, None, None, "unknown MFMA opcode: add it to REQ"))
            continue
        dst = regs(ops.split(",")[0])
        # depth-first over paths; state = (instruction index, wait states so far, what the path knows about vcc, SGPR pairs the path has seen
        # set to 0 / -1).  The last two follow ONE idiom of the compiler (round 5): it merges the tails of an if / else by setting a flag pair
        # (s_mov_b64 s[a:b], 0 | -1) in each arm and branching on `s_and_b64 vcc, exec, s[a:b]` behind the merge -- a path that has seen the
        # s_mov takes only the edge that flag allows; without this the scan walks from the MFMAs of one arm into the other arm's code
        best = {}
        stack = [(s, 0, None, ()) for s in succ(k)]
        while stack:
            q, ws, vcc, pairs = stack.pop()
            if ws >= need or q >= n:
                continue
            key = (q, vcc, pairs)
            if key in best and best[key] <= ws:
                continue
            best[key] = ws
            qt, qmn, qops = ins[q]
            touched = regs(qops) & dst
            if touched:
                qb = qmn.replace("_e64", "")
                if qb.startswith("v_mfma") or qb.startswith("v_smfmac"):
                    o = [x.strip() for x in qops.split(",")]
                    # accumulate chain: the registers appear only as vDst / srcC (operands 0 and 3) -> no software wait needed here
                    if not (regs(o[1]) & dst or regs(o[2]) & dst):
                        continue   # a new MFMA owns the registers from here on; its own hazards are checked from its own start
                findings.append((name, k, t, q, qt, f"{ws} wait state(s) on the shortest path, {need} required"))
                continue
            step = 1
            if qmn == "s_nop":
                step = int(qops.strip(), 0) + 1
            o = [x.strip() for x in qops.split(",")] if qops else []
            nxt = succ(q)
            if qmn in ("s_cbranch_vccz", "s_cbranch_vccnz") and vcc is not None and len(nxt) == 2:
                taken = (vcc == 0) == (qmn == "s_cbranch_vccz")
                nxt = [nxt[1]] if taken else [nxt[0]]
            pd = dict(pairs)
            if qmn == "s_mov_b64" and len(o) == 2 and o[0].startswith("s[") and o[1] in ("0", "-1"):
                pd[o[0]] = 0 if o[1] == "0" else -1
            elif o and o[0].startswith("s") and o[0] in pd:
                del pd[o[0]]   # (any other write to a tracked pair; partial writes of its halves are not produced for these flags)
            if qmn == "s_and_b64" and len(o) == 3 and o[0] == "vcc" and "exec" in o[1:] and any(x in pd for x in o[1:]):
                vcc = 0 if pd[[x for x in o[1:] if x in pd][0]] == 0 else 1   # exec is not zero on a path that is being executed
            elif (o and o[0] in ("vcc", "vcc_lo", "vcc_hi")) or (qmn.startswith("v_cmp") and not qmn.endswith("_e64")) or qmn.startswith("v_div_scale") or qmn in ("v_add_co_u32_e32", "v_sub_co_u32_e32", "v_addc_co_u32_e32", "v_subb_co_u32_e32", "v_subrev_co_u32_e32"):
                vcc = None
            npairs = tuple(sorted(pd.items()))
            for s in nxt:
                stack.append((s, ws + step, vcc, npairs))
    # one line per (mfma, reader) pair
    seen, out = set(), []
    for f in findings:
        key = (f[1], f[3])
        if key not in seen:
            seen.add(key)
            out.append(f)
    return out, sum(1 for i in ins if i[1].startswith("v_mfma"))


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
        n_fn = n_mfma = 0
        for name, lines in functions(path):
            res, cnt = scan_function(name, lines)
            n_fn += 1
            n_mfma += cnt
            for (fn, k, t, q, qt, why) in res:
                bad += 1
                print(f"{os.path.basename(path)}: {fn}\n    MFMA   #{k}: {t}\n    reader #{q}: {qt}\n    -> {why}")
        print(f"{os.path.basename(path)}: {n_fn} functions, {n_mfma} MFMA instructions scanned")
    print("FINDINGS:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
