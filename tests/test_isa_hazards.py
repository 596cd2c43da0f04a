"""CPU: the gfx950 device code of the kernels that branch near MFMA sequences holds no MFMA result that is read too early on any
control-flow path (tools/isa_mfma_hazard_scan.py; the cause of round 3's stale-accumulator bug, profiles/r04_isa_stale_accumulator.txt)."""
import importlib.util
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_scan", os.path.join(ROOT, "tools", "isa_mfma_hazard_scan.py"))
isa_scan = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa_scan)

spec2 = importlib.util.spec_from_file_location("isa_asm_scan", os.path.join(ROOT, "tools", "isa_asm_hazard_scan.py"))
isa_asm = importlib.util.module_from_spec(spec2)
spec2.loader.exec_module(isa_asm)

# the shape of the bug as hipcc compiled it at 81e779f^: an MFMA directly in front of a conditional branch whose target reads its result
BUGGY = """
_Z5buggyv:
	v_mfma_f32_16x16x32_bf16 a[8:11], v[158:161], v[210:213], a[8:11]
	s_and_b64 vcc, exec, s[20:21]
	v_mfma_f32_16x16x32_bf16 a[12:15], v[174:177], v[210:213], a[12:15]
	s_cbranch_vccnz .LBB0_2
; %bb.1:
	v_accvgpr_read_b32 v2, a56
	v_accvgpr_read_b32 v3, a57
	v_accvgpr_read_b32 v4, a58
	v_accvgpr_read_b32 v5, a59
	s_nop 1
	v_mfma_f32_16x16x32_bf16 a[0:3], v[2:5], v[210:213], a[4:7]
.LBB0_2:
	v_accvgpr_read_b32 v2, a15
	v_accvgpr_read_b32 v3, a14
	s_endpgm
.Lfunc_end0:
"""
FIXED = BUGGY.replace(".LBB0_2:\n", ".LBB0_2:\n\ts_nop 7\n")


def _scan_text(text, tmp_path, name):
    p = tmp_path / name
    p.write_text(text)
    out = []
    for fn, lines in isa_scan.functions(str(p)):
        res, _ = isa_scan.scan_function(fn, lines)
        out += res
    return out


def test_scanner_finds_the_branch_hazard_and_accepts_the_padded_form(tmp_path):
    bad = _scan_text(BUGGY, tmp_path, "buggy.s")
    assert any("a15" in f[4] and f[5].startswith("1 wait") for f in bad), bad
    assert _scan_text(FIXED, tmp_path, "fixed.s") == []


# the compiler's merged if / else tails (round 5): each arm sets a flag pair, the code behind the merge branches on it.  From the MFMA's arm the
# branch is always taken; the fall-through block (the OTHER arm's continuation, which reuses the accumulator's registers) is not reachable
FLAG_IDIOM = """
_Z4flagv:
	v_mfma_f32_32x32x16_bf16 v[2:17], v[182:185], v[202:205], v[2:17]
	s_mov_b64 s[86:87], 0
.LBB0_1:
	v_lshlrev_b32_e32 v195, 4, v215
	s_and_b64 vcc, exec, s[86:87]
	s_cbranch_vccz .LBB0_3
; %bb.2:
	v_mul_lo_u32 v2, v218, s45
.LBB0_3:
	s_nop 7
	s_nop 7
	v_add_f32_e32 v20, v2, v3
	s_endpgm
.Lfunc_end0:
"""


def test_scanner_follows_the_compilers_flag_idiom(tmp_path):
    assert _scan_text(FLAG_IDIOM, tmp_path, "flag.s") == []
    # without the flag's value on the path both edges are possible again: the write of v2 four wait states behind the MFMA is reported
    bad = _scan_text(FLAG_IDIOM.replace("\ts_mov_b64 s[86:87], 0\n", "\ts_nop 0\n"), tmp_path, "noflag.s")
    assert any("v_mul_lo_u32 v2" in f[4] for f in bad), bad


# ---- hazards inline asm can break (tools/isa_asm_hazard_scan.py): the LDS-DMA blocks read M0 and an SGPR base the surrounding code wrote ----
# the library's block as it is written (s_nop 0 behind the M0 write) with its SGPR operands produced far enough ahead ...
DMA_OK = """
_Z6dma_okv:
	v_readfirstlane_b32 s4, v10
	v_readfirstlane_b32 s5, v11
	v_readfirstlane_b32 s6, v12
	v_add_u32_e32 v1, v2, v3
	v_add_u32_e32 v4, v2, v3
	s_mov_b32 s20, m0
	s_mov_b32 m0, s6
	s_nop 0
	global_load_lds_dwordx4 v9, s[4:5] sc1
	s_mov_b32 m0, s20
	s_endpgm
.Lfunc_end0:
"""
# ... the M0 write directly in front of the load (H1), and the base address read two wait states behind its v_readfirstlane on the TAKEN edge (H2)
DMA_NO_NOP = DMA_OK.replace("\ts_nop 0\n", "")
DMA_SGPR_LATE = """
_Z8dma_latev:
	v_readfirstlane_b32 s5, v11
	s_cbranch_scc1 .LBB0_2
; %bb.1:
	s_nop 7
.LBB0_2:
	s_mov_b32 m0, s6
	s_nop 0
	global_load_lds_dwordx4 v9, s[4:5] sc1
	s_endpgm
.Lfunc_end0:
"""
LANE_SELECT = """
_Z4lanev:
	v_readfirstlane_b32 s7, v3
	s_nop 1
	v_readlane_b32 s8, v5, s7
	v_cmp_lt_f32_e32 vcc, v1, v2
	s_nop 2
	v_div_fmas_f32 v6, v7, v8, v9
	s_endpgm
.Lfunc_end0:
"""


def _asm_scan_text(text, tmp_path, name):
    p = tmp_path / name
    p.write_text(text)
    out = []
    for fn, lines in isa_asm.functions(str(p)):
        out += isa_asm.scan_function(fn, lines)[0]
    return out


def test_asm_hazard_scanner_on_the_lds_dma_block(tmp_path):
    assert _asm_scan_text(DMA_OK, tmp_path, "ok.s") == []
    bad = _asm_scan_text(DMA_NO_NOP, tmp_path, "nonop.s")
    assert len(bad) == 1 and bad[0][5].startswith("H1") and "0 wait" in bad[0][5], bad
    bad = _asm_scan_text(DMA_SGPR_LATE, tmp_path, "late.s")
    assert len(bad) == 1 and bad[0][5].startswith("H2") and "(s5)" in bad[0][5] and "3 wait" in bad[0][5], bad   # branch, s_mov, s_nop 0 on the taken edge
    bad = _asm_scan_text(LANE_SELECT, tmp_path, "lane.s")
    assert sorted(f[5][:2] for f in bad) == ["H3", "H4"], bad
    assert _asm_scan_text(LANE_SELECT.replace("s_nop 1", "s_nop 3").replace("s_nop 2", "s_nop 3"), tmp_path, "lane_ok.s") == []


def _makefile_sources():
    """The SRCS of paule_amd/csrc/Makefile: what the shipped library is built from -- a new kernel file is scanned without anybody listing it here."""
    mk = open(os.path.join(ROOT, "paule_amd", "csrc", "Makefile")).read()
    return re.search(r"^SRCS\s*:=\s*(.*)$", mk, re.M).group(1).split()


# MFMA instructions per source file as compiled in round 5 (the scan must really have seen the kernels: a floor of ~90 % each)
MFMA_FLOOR = {"gemm.hip": 430, "gemm_big.hip": 120, "lstm.hip": 36, "lstm_chain_f32.hip": 800, "lstm_fused.hip": 2050, "lstm_fused2.hip": 800, "lstm_persist.hip": 760,
              "lstm_persist16.hip": 770, "lstm_persist_f32.hip": 3080, "lstm_persist_rs.hip": 500, "train.hip": 190}


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_shipped_kernels_have_no_mfma_read_hazard_and_no_asm_hazard(tmp_path):
    files = _makefile_sources()
    assert set(MFMA_FLOOR) <= set(files), (sorted(MFMA_FLOOR), files)

    def build(f):
        out = str(tmp_path / (f[:-4] + ".s"))
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-w", "-S", "--cuda-device-only", "-o", out,
                               os.path.join(ROOT, "paule_amd", "csrc", f)])
        return out

    with ThreadPoolExecutor(8) as ex:
        outs = list(ex.map(build, files))
    findings, n_mfma, n_dma = [], {}, 0
    for f, o in zip(files, outs):
        n_mfma[f] = 0
        for fn, lines in isa_scan.functions(o):
            res, cnt = isa_scan.scan_function(fn, lines)
            findings += res
            n_mfma[f] += cnt
            # ... and none of the hazards inline asm can break: M0 written directly in front of an LDS-DMA load, a VMEM instruction reading an SGPR
            # a VALU instruction wrote less than five wait states ago, VALU-written lane selects, v_div_fmas behind a VALU write of VCC
            res2, dma, _ = isa_asm.scan_function(fn, lines)
            findings += res2
            n_dma += dma
    assert n_dma >= 600, n_dma   # the LDS-DMA loads of lstm_fused.hip, lstm_fused2.hip, gemm_big.hip were really seen
    for f, floor in MFMA_FLOOR.items():
        assert n_mfma[f] >= floor, (f, n_mfma[f], floor)
    for f in files:   # a file that grew matrix code gets a floor of its own
        assert f in MFMA_FLOOR or n_mfma[f] == 0, (f, n_mfma[f])
    assert findings == [], findings[:3]
