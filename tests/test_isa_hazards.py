"""CPU check of the SHIPPED gfx950 ISA (no GPU needed): the library's code objects are disassembled and every VMEM
instruction with an SGPR operand is checked for the gfx9 hazard "VALU writes an SGPR -> VMEM reads it: 5 wait states".

Round 4's four GPU memory faults were this hazard: the scalar-base stores of the fused kernels are inline
assembly (csrc/mfma_common.h: store_uniform_base), which LLVM's hazard recogniser does not look into, and an inline-asm
`v_readfirstlane_b32 sN` sat one to three instructions in front of `global_store_dword v, v, s[N:N+1]`.  The committed
tree has no such pair; this test keeps it that way for every later tile shape / compiler release."""
import os

import pytest

import tools.codeobj as codeobj
from uda_amd import capi


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIB_PATH):
        capi.build()
    return capi.LIB_PATH


def test_no_vmem_instruction_reads_a_valu_written_sgpr_too_early(lib):
    n, bad = codeobj.lint_lib(lib)
    assert n >= 2000, "the scan looks vacuous: %d VMEM instructions with SGPR operands" % n
    assert not bad, "\n".join("%s: %s <- %s (%d wait states)" % (f, s, w, g) for f, s, w, g, _ in bad[:20])


# the pattern of the faulting round-4 build, as `hipcc -S` printed it for mbxp_kernel<5, 13, true, 4> when the hack was
# re-applied to a scratch copy (tools/codeobj.py lint --asm: 4334 findings in kernels_pwb.hip)
HACK = """
_ZN3uda4testEv:
\ts_addc_u32 s79, s53, s38
\tv_mov_b32_e32 v160, s78
\tv_mov_b32_e32 v161, s79
\t;;#ASMSTART
\tv_readfirstlane_b32 s78, v160
\t;;#ASMEND
\tv_mul_f32_e32 v160, v158, v159
\t;;#ASMSTART
\tv_readfirstlane_b32 s79, v161
\t;;#ASMEND
\tv_fmac_f32_e32 v157, v158, v159
%s\t;;#ASMSTART
\tglobal_store_dword v184, v160, s[78:79]
\t;;#ASMEND
\ts_endpgm
"""


def test_the_lint_finds_the_round4_pattern_and_accepts_the_padded_one():
    n, bad = codeobj.lint_text(HACK % "")
    assert n == 1 and len(bad) == 2
    assert sorted(g for _, _, _, g, _ in bad) == [1, 3]          # s79: one instruction in between, s78: three
    assert codeobj.lint_text(HACK % "\ts_nop 3\n")[1] == []      # 1 + 4 wait states for s79: enough
    assert len(codeobj.lint_text(HACK % "\ts_nop 2\n")[1]) == 1  # 1 + 3: still short for s79


def test_the_lint_follows_branches_into_the_window():
    asm = """
_ZN3uda5test2Ev:
\tv_readfirstlane_b32 s10, v1
\ts_branch .LBB0_2
.LBB0_1:
\ts_mov_b32 s10, 0
\ts_nop 4
.LBB0_2:
\tglobal_load_dword v2, v3, s[10:11]
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
"""
    n, bad = codeobj.lint_text(asm)
    assert n == 1 and len(bad) == 1 and bad[0][3] == 1      # reached through the s_branch, not through the fall-through
    # a VOP3 compare and a carry-out write SGPR pairs as well
    asm2 = "_ZN3uda5test3Ev:\n\tv_cmp_gt_u32_e64 s[4:5], v0, v1\n\tbuffer_load_dword v2, v3, s[4:7], 0 offen\n\ts_endpgm\n"
    assert len(codeobj.lint_text(asm2)[1]) == 1
    asm3 = "_ZN3uda5test4Ev:\n\tv_add_co_u32_e64 v1, s[8:9], v0, v1\n\ts_nop 4\n\tglobal_store_dword v2, v3, s[8:9]\n\ts_endpgm\n"
    assert codeobj.lint_text(asm3)[1] == []


def test_streaming_1x1_kernels_do_not_spill(lib):
    """pwb_kernel runs three (two for the widest tiles) blocks per CU by its register count: a spill there is a silent
    occupancy / scratch-traffic regression (tools/codeobj.py resources prints the whole table)."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        rows = [r for elf in codeobj.extract(lib, td) for r in codeobj.resources(elf)]
    names = codeobj.demangle([r["name"] for r in rows])
    pwb = [(n, r) for n, r in zip(names, rows) if "pwb_kernel<" in n]
    assert len(pwb) >= 12
    for n, r in pwb:
        if n.endswith(", true>(uda::PwArgs)"):
            # (the two-chunks-in-flight variant of the widest tile holds two operand sets at the 256-register ceiling of its
            # two blocks per CU: one spilled register, outside the chunk loop, is tolerated - not a trend)
            assert r["vgpr_spill"] <= 1 and r["scratch"] <= 8, (n, r)
        else:
            assert r["vgpr_spill"] == 0 and r["scratch"] == 0, (n, r)
