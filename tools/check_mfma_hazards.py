#!/usr/bin/env python3
"""Scan the device code of a built library for AccVGPR accesses that race an MFMA still in flight.

Why: the fp32 GEMM and GRU kernels issue `v_mfma_f32_32x32x2_f32` as INLINE ASM (the only way to keep the accumulators in
AccVGPRs).  Inline asm is opaque to the compiler's hazard recognizer, so nothing stops the register
allocator from placing a `v_accvgpr_mov / read / write` of an accumulator register straight behind such an MFMA -- e.g. a copy
on the exit edge of a K loop when two regions of a kernel disagree about where a tile lives.  The copy then reads registers
the matrix core is still writing.  This happened once (round 3, gru_step_fwd_kernel with a two-entry K loop: rows 0 and 27 of
every 32-row tile wrong); the kernels' own epilogues are protected by `laed_mfma_drain`, compiler-made copies are not.

What is checked, per kernel, on the disassembly of every gfx950 code object in the .so:
  a forward data-flow over the control-flow graph carries, per AccVGPR, the number of wait states since the last MFMA that wrote
  it (every instruction counts 1, `s_nop N` counts N + 1; at joins the minimum wins; loops iterate to a fixed point); any
  non-MFMA instruction that names an AccVGPR fewer than NEED = 19 wait states after such a write is reported (the gfx940/950
  requirement for "XDL write VGPR -> VALU / v_accvgpr read or write of the same register" after a 16-pass operation; a
  4- or 8-pass MFMA needs fewer, so the rule is conservative for them).  MFMA -> MFMA on the same tile is interlocked
  by the hardware and not reported.  MFMAs the compiler issued itself (builtins) come with their s_nops and pass.

usage:  python tools/check_mfma_hazards.py [lib.so ...]      (default: both built libraries)   exit status 1 = findings
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
NEED = 19
HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(HERE, "..", "neuraldialog-laed_amd", "laed_amd")

_AREG = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")
_LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_FUNC = re.compile(r"^[0-9a-f]+ <(.+)>:$")
_TARGET = re.compile(r"<[^>]*\+0x([0-9a-fA-F]+)>")


def aregs(text):
    out = set()
    for m in _AREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def disassemble(so_path, tmp):
    """-> {kernel name: [(addr, mnemonic, operands)]} over every gfx950 code object of the library."""
    local = os.path.join(tmp, os.path.basename(so_path))
    shutil.copy(so_path, local)                      # llvm-objdump --offloading writes the bundles next to its input
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    funcs = {}
    for name in sorted(os.listdir(tmp)):
        if not (name.startswith(os.path.basename(so_path) + ".") and name.endswith("gfx950")):
            continue
        txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, name)], check=True,
                             stdout=subprocess.PIPE, universal_newlines=True).stdout
        cur, base = None, 0
        for line in txt.splitlines():
            f = _FUNC.match(line)
            if f:
                cur = funcs.setdefault(f.group(1), [])
                base = None
                continue
            m = _LINE.match(line)
            if m and cur is not None:
                addr = int(m.group(3), 16)
                if base is None:
                    base = addr
                    cur.append(("base", base, "", None))
                t = _TARGET.search(line[m.end():])
                cur.append((addr, m.group(1), m.group(2), int(t.group(1), 16) if t else None))
    return funcs


def check_kernel(name, ins):
    base = ins[0][1]
    ins = ins[1:]
    index = {t[0]: i for i, t in enumerate(ins)}
    n = len(ins)
    succ = [[] for _ in range(n)]
    for i, (addr, op, args, off) in enumerate(ins):
        if op == "s_endpgm":
            continue
        if op.startswith("s_cbranch") or op == "s_branch":
            if off is None or (base + off) not in index:
                raise RuntimeError("%s: branch at %#x without a resolvable target" % (name, addr))
            succ[i].append(index[base + off])
        if op != "s_branch" and i + 1 < n:
            succ[i].append(i + 1)
    # state before instruction i: {agpr: (wait states since the MFMA write, address of that MFMA)}, entries < NEED only
    state = [None] * n
    state[0] = {}
    work = [0]
    findings = {}
    while work:
        i = work.pop()
        st = state[i]
        addr, op, args, _ = ins[i]
        is_mfma = op.startswith("v_mfma") or op.startswith("v_smfmac")
        if st and not is_mfma:
            used = aregs(args)
            if used:
                hot = [st[r] for r in used if r in st]
                if hot:
                    findings[addr] = (op, args) + min(hot)
        step = 1
        if op == "s_nop":
            try:
                step = int(args.strip(), 0) + 1
            except ValueError:
                step = 1
        out = {r: (d + step, src) for r, (d, src) in st.items() if d + step < NEED}
        if is_mfma:
            dst = args.split(",")[0]
            for r in aregs(dst):
                out[r] = (0, addr)
        for j in succ[i]:
            old = state[j]
            if old is None:
                state[j] = dict(out)
                work.append(j)
            else:
                changed = False
                for r, d in out.items():
                    if r not in old or old[r][0] > d[0]:
                        old[r] = d
                        changed = True
                if changed:
                    work.append(j)
    return findings


def check_library(path):
    tmp = tempfile.mkdtemp(prefix="laed_hazard_")
    try:
        funcs = disassemble(path, tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    bad = 0
    n_mfma = 0
    for name, ins in sorted(funcs.items()):
        if len(ins) < 2:
            continue
        n_mfma += sum(1 for t in ins[1:] if t[1].startswith("v_mfma"))
        for addr, (op, args, d, src) in sorted(check_kernel(name, ins).items()):
            bad += 1
            print("%s: %s\n    %#x  %s %s   <- %d wait state(s) after the MFMA at %#x wrote it (need %d)" %
                  (os.path.basename(path), name[:150], addr, op, args, d, src, NEED))
    print("%s: %d kernels, %d MFMA instructions, %d finding(s)" % (os.path.basename(path), len(funcs), n_mfma, bad))
    return bad


def main(argv):
    libs = argv or [p for p in (os.path.join(PKG, "liblaed_hip.so"), os.path.join(PKG, "liblaed_hip_hooks.so")) if os.path.exists(p)]
    total = sum(check_library(p) for p in libs)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
