#!/usr/bin/env python3
"""gfx950 code objects of libuda_hip.so on the CPU: per-kernel resources and an ISA hazard lint.

    python tools/codeobj.py resources [--lib PATH] [--match SUBSTR]     registers / spills / LDS / scratch per kernel
    python tools/codeobj.py lint [--lib PATH | --asm FILE.s]            VALU-written SGPR read by a VMEM instruction too early

Why the lint exists: on gfx9 a VMEM instruction that reads an SGPR (the `s[n:n+1]` base of
`global_store_dword v, v, s[..]`, a buffer resource, a scalar offset) needs FIVE wait states after a VALU instruction
wrote that SGPR (`v_readfirstlane_b32`, `v_readlane_b32`, a VOP3 compare or a carry-out with an SGPR destination).  LLVM's
hazard recogniser inserts the `s_nop`s for instructions it selected itself, but an `asm volatile` body is opaque to it
(GCNHazardRecognizer::checkVMEMHazards looks at SIInstrInfo::isVMEM instructions; INLINEASM is none): the scalar-base
stores of the fused kernels are written as inline assembly (csrc/mfma_common.h: store_uniform_base), so whenever the
compiler - or another inline-asm statement - forms such a base with a VALU instruction right in front of the store, the
store goes out with the register's OLD contents: a wild address.  That is a timing-dependent memory fault, which is what
round 4 saw.  Nothing but a scan of the shipped ISA keeps a build honest; tests/test_isa_hazards.py runs this one.

Wait states are counted conservatively: one per instruction issued in between (`s_nop k` = k + 1), no credit for
multi-cycle instructions.  The scan walks backwards from every VMEM instruction with an SGPR operand through the straight
line code AND through every branch that targets a label inside the window.
"""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "uncertainty-detection-autolabeling_amd", "csrc", "libuda_hip.so")
VMEM_SGPR_WAIT_STATES = 5
_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def extract(lib, outdir):
    """The gfx950 code objects embedded in a HIP shared library / object (.hip_fatbin: one uncompressed
    clang-offload-bundle per translation unit) -> list of ELF paths."""
    fat = os.path.join(outdir, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib,
                           os.path.join(outdir, "stripped.tmp")])
    data = open(fat, "rb").read()
    out, i = [], data.find(_MAGIC)
    while i >= 0:
        n = struct.unpack_from("<Q", data, i + 24)[0]
        q = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            q += 24
            triple = data[q:q + tl].decode()
            q += tl
            if "gfx950" in triple and size:
                path = os.path.join(outdir, "co%d.elf" % len(out))
                with open(path, "wb") as f:
                    f.write(data[i + off:i + off + size])
                out.append(path)
        i = data.find(_MAGIC, i + 1)
    if not out:
        raise RuntimeError("no gfx950 code object in %s" % lib)
    return out


def resources(elf):
    """[{name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, lds, scratch, max_wg}] from the code object's metadata note."""
    import yaml
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], text=True)
    doc = txt[txt.index("---"):]
    doc = doc[:doc.index("\n...")] if "\n..." in doc else doc
    meta = yaml.safe_load(doc)
    keys = {".vgpr_count": "vgpr", ".agpr_count": "agpr", ".sgpr_count": "sgpr", ".vgpr_spill_count": "vgpr_spill",
            ".sgpr_spill_count": "sgpr_spill", ".group_segment_fixed_size": "lds", ".private_segment_fixed_size": "scratch",
            ".max_flat_workgroup_size": "max_wg"}
    out = []
    for k in meta.get("amdhsa.kernels", []):
        row = {"name": k[".name"]}
        row.update({v: int(k.get(src, 0)) for src, v in keys.items()})
        out.append(row)
    return out


def demangle(names):
    import shutil
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt", path=LLVM)
    if not tool:
        return list(names)
    p = subprocess.run([tool], input="\n".join(names), text=True, capture_output=True)
    return p.stdout.splitlines() if p.returncode == 0 else list(names)


def disassemble(elf):
    return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--symbolize-operands", elf], text=True)


# ------------------------------------------------------------------------------------------------ hazard lint
_SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]|\b(vcc)\b")
_VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_|image_|tbuffer_)")
_TWO_DEST = re.compile(r"^v_(add|sub|subrev|addc|subb|subbrev)_co_|^v_div_scale_|^v_mad_[ui]64_")


def _sregs(operand):
    out = set()
    for m in _SREG.finditer(operand):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        elif m.group(2) is not None:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.update((106, 107))      # vcc
    return out


def parse_functions(asm):
    """objdump -d --symbolize-operands (or hipcc -S) text -> {function: [(labels_here, mnemonic, operands, raw)]}."""
    funcs, cur, pending = {}, None, []
    for raw in asm.splitlines():
        line = raw.split("//")[0].split(";")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line.strip())
        if m:                                    # objdump: function symbol or <L12> label
            if re.match(r"^L\d+$", m.group(1)):
                pending.append(m.group(1))
            else:
                cur = funcs.setdefault(m.group(1), [])
                pending = []
            continue
        m = re.match(r"^([.\w$]+):$", line.strip())
        if m:                                    # hipcc -S: "name:" / ".LBB0_3:"
            if m.group(1).startswith(".L"):
                pending.append(m.group(1))
            elif m.group(1).startswith("_Z") or cur is None:
                cur = funcs.setdefault(m.group(1), [])
                pending = []
            continue
        if cur is None or not raw[:1].isspace():
            continue
        body = line.strip()
        if body.startswith("."):
            continue                             # assembler directive
        parts = body.split(None, 1)
        cur.append((tuple(pending), parts[0], parts[1] if len(parts) > 1 else "", body))
        pending = []
    return funcs


def _valu_sgpr_defs(mn, ops):
    """SGPRs a VALU instruction writes (readfirstlane / readlane, VOP3 compares, carry-outs)."""
    if not mn.startswith("v_"):
        return set()
    fields = [o.strip() for o in ops.split(",")]
    dests = fields[:2] if _TWO_DEST.match(mn) else fields[:1]
    if mn.startswith("v_cmpx"):
        return set()
    out = set()
    for d in dests:
        if d.startswith("s") or d.startswith("vcc"):
            out |= _sregs(d)
    return out


def lint_function(insts, need=VMEM_SGPR_WAIT_STATES):
    """[(index of the VMEM instruction, index of the VALU writer, wait states in between, registers)]."""
    label_at, branches = {}, {}
    for i, (labels, mn, ops, _) in enumerate(insts):
        for lb in labels:
            label_at[lb] = i
    for i, (_, mn, ops, _) in enumerate(insts):
        if mn.startswith("s_branch") or mn.startswith("s_cbranch"):
            tgt = ops.split(",")[-1].strip()
            if tgt in label_at:
                branches.setdefault(label_at[tgt], []).append(i)
    findings = []

    def walk(i, regs, budget, origin, seen):
        """instructions that can execute right before position i, with `budget` wait states still missing."""
        while budget > 0 and i >= 0:
            if (i, budget) in seen:
                return
            seen.add((i, budget))
            labels, mn, ops, _ = insts[i]
            hit = _valu_sgpr_defs(mn, ops) & regs
            if hit:
                findings.append((origin, i, need - budget, sorted(hit)))
            budget -= (int(ops.strip() or "0", 0) + 1) if mn == "s_nop" else 1
            # fall through to the linear predecessor unless it cannot fall through
            preds = []
            if i > 0 and not (insts[i - 1][1] in ("s_branch", "s_endpgm", "s_setpc_b64")):
                preds.append(i - 1)
            preds += branches.get(i, [])
            if not preds:
                return
            for p in preds[1:]:
                walk(p, regs, budget, origin, seen)
            i = preds[0]

    for i, (labels, mn, ops, _) in enumerate(insts):
        if not _VMEM.match(mn):
            continue
        regs = _sregs(ops)
        regs.discard(106), regs.discard(107)
        if not regs:
            continue
        preds = []
        if i > 0 and insts[i - 1][1] not in ("s_branch", "s_endpgm", "s_setpc_b64"):
            preds.append(i - 1)
        preds += branches.get(i, [])
        seen = set()
        for p in preds:
            walk(p, regs, need, i, seen)
    return findings


def lint_text(asm, need=VMEM_SGPR_WAIT_STATES):
    """-> (number of VMEM instructions with an SGPR operand, [(function, store text, writer text, wait states, regs)])."""
    n_sites, out = 0, []
    for name, insts in parse_functions(asm).items():
        n_sites += sum(1 for _, mn, ops, _ in insts if _VMEM.match(mn) and (_sregs(ops) - {106, 107}))
        for at, w, gap, regs in lint_function(insts, need):
            out.append((name, insts[at][3], insts[w][3], gap, regs))
    return n_sites, out


def lint_lib(lib=DEFAULT_LIB):
    with tempfile.TemporaryDirectory() as td:
        n, bad = 0, []
        for elf in extract(lib, td):
            a, b = lint_text(disassemble(elf))
            n += a
            bad += b
    return n, bad


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("cmd", choices=["resources", "lint"])
    ap.add_argument("--lib", default=DEFAULT_LIB)
    ap.add_argument("--asm", help="lint an assembly listing (hipcc -S output) instead of the library")
    ap.add_argument("--match", default="", help="resources: only kernels whose demangled name contains this")
    a = ap.parse_args()
    if a.cmd == "lint":
        n, bad = lint_text(open(a.asm).read()) if a.asm else lint_lib(a.lib)
        for fn, st, wr, gap, regs in bad:
            print("HAZARD %s\n   %s\n   <- %s   (%d wait states, s%s)" % (fn, st, wr, gap, regs))
        print("%d VMEM instructions with SGPR operands scanned, %d within %d wait states of a VALU write of their SGPR"
              % (n, len(bad), VMEM_SGPR_WAIT_STATES))
        sys.exit(1 if bad else 0)
    with tempfile.TemporaryDirectory() as td:
        rows = []
        for elf in extract(a.lib, td):
            rows += resources(elf)
    names = demangle([r["name"] for r in rows])
    print("%-100s %5s %5s %5s %6s %6s %7s %7s" % ("kernel", "vgpr", "agpr", "sgpr", "vspill", "sspill", "lds", "scratch"))
    for r, nm in sorted(zip(rows, names), key=lambda t: t[1]):
        if a.match in nm:
            print("%-100s %5d %5d %5d %6d %6d %7d %7d" % (nm[:100], r.get("vgpr", -1), r.get("agpr", 0), r.get("sgpr", -1),
                                                       r.get("vgpr_spill", 0), r.get("sgpr_spill", 0), r.get("lds", 0), r.get("scratch", 0)))


if __name__ == "__main__":
    main()
