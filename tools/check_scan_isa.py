#!/usr/bin/env python3
"""Static check of the hand-scheduled load pipeline in k_scan_probe (phasm_amd/csrc/kernels.hip.h).

The kernel issues its steady-state loads with inline asm and waits for them with counted
`s_waitcnt vmcnt(N)` asm statements.  That is only correct if no instruction reads a landing register
between its asm load and the asm wait that retires it (hipcc does not know the register is pending:
a stray v_mov would read stale data).  This script compiles the library to assembly, walks every
k_scan_probe kernel in layout order (main loop walked twice, for the back edge) and reports any read
of a pending register.  Rule used for retirement (conservative): an asm `s_waitcnt vmcnt(N)` retires
the waits with N = 4 / 8 retire the record/word loads, N = 3 both probe rounds (see the kernel).

    python tools/check_scan_isa.py [--strict]

Limits: the walk follows every edge of the structurised control-flow graph, including edges that
cannot be taken at run time (s_cbranch_execz with a non-empty EXEC, correlated wave-uniform
conditions such as "this pass issued probes" / "the stream ended"), so it reports CANDIDATES for
review, not proofs of a bug; registers that hipcc legitimately reuses as temporaries on such paths
show up too.  It is a diagnostic to run after touching the kernel (look for v_mov / v_readfirstlane
of a landing register right before an asm wait); the run-to-run determinism tests in
tests/test_gpu_parity.py (few waves per workgroup = long pipelines) are the gate.  Exit status is 0
unless --strict is given.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "phasm_amd", "csrc", "c_api.hip")


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def parse(lines):
    """-> list of (lineno, text, in_asm)"""
    out, in_asm = [], False
    for i, ln in enumerate(lines):
        t = ln.strip()
        if ";" in t and not t.startswith(";;#ASM"):
            t = t.split(";", 1)[0].strip()   # drop trailing comments (labels carry them)
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") and not t.endswith(":"):
            continue
        out.append((i + 1, t, in_asm))
    return out


def sources(text):
    ops = text.split(None, 1)
    if len(ops) < 2:
        return set()
    mnem, rest = ops
    parts = [p.strip() for p in rest.split(",")]
    if mnem.startswith(("global_store", "global_atomic", "ds_write", "ds_add", "ds_or", "v_cmp", "v_readfirstlane",
                        "v_readlane", "s_", "buffer_store", "scratch_store")):
        src = parts if not mnem.startswith(("v_readfirstlane", "v_readlane")) else parts[1:]
    else:
        src = parts[1:]
    used = set()
    for p in src:
        used |= regs(p)
    return used


def check_kernel(name, body):
    """Forward dataflow over the kernel's control-flow graph: pending = asm loads whose landing
    registers may still be in flight; any read of such a register is an error."""
    ins = parse(body)
    n = len(ins)
    label_at = {t[:-1]: k for k, (_, t, _) in enumerate(ins) if t.endswith(":")}
    # classify asm loads by their place in the layout: the 3 loads after a record/word wait (N = 4, or 8 when
    # a second probe round is in flight) are RW; the asm loads that follow, up to the next such wait, are the
    # first probe round (P1, 4 loads) and then the optional second round (P2, 4 loads).  Before the first wait:
    # 3 RW loads and the 4 dummy probes.
    cls, since = {}, 0
    for k, (_, t, a) in enumerate(ins):
        if a and t.startswith("s_waitcnt vmcnt"):
            if int(re.search(r"vmcnt\((\d+)\)", t).group(1)) in (4, 8):
                since = 0
        elif a and t.startswith("global_load"):
            cls[k] = "RW" if since < 3 else ("P1" if since < 7 else "P2")
            since += 1
    if not cls:
        return ["%s: no asm loads found (pipeline removed?)" % name]
    dest = {k: frozenset(regs(ins[k][1].split(None, 1)[1].split(",")[0])) for k in cls}

    def succs(k):
        t = ins[k][1]
        if t.startswith("s_endpgm"):
            return []
        m = re.match(r"(s_c?branch\w*)\s+(\.LBB\w+)", t)
        if m and m.group(2) in label_at:
            return [label_at[m.group(2)]] if m.group(1) == "s_branch" else [label_at[m.group(2)], k + 1]
        return [k + 1] if k + 1 < n else []

    def transfer(k, pend):
        _, t, a = ins[k]
        if a and t.startswith("s_waitcnt vmcnt"):
            nn = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
            if nn == 0:
                return frozenset()
            # 4 / 8: record + words done; 3: both probe rounds done
            done = {4: ("RW",), 8: ("RW",), 3: ("P1", "P2")}.get(nn, ())
            return frozenset(x for x in pend if cls[x] not in done)
        if k in cls:
            return pend | {k}
        return pend

    state = {0: frozenset()}
    work = [0]
    while work:
        k = work.pop()
        out = transfer(k, state[k])
        for j in succs(k):
            new = state.get(j, frozenset()) | out
            if j not in state or new != state[j]:
                state[j] = new
                work.append(j)
    errors = []
    for k, pend in sorted(state.items()):
        lineno, t, a = ins[k]
        if t.endswith(":") or (a and t.startswith("s_waitcnt")):
            continue
        used = sources(t)
        bad = sorted({r for x in pend for r in dest[x] if r in used})
        if bad:
            errors.append("%s:%d reads in-flight v%s: %s" % (name, lineno, bad, t))
    return errors


def main():
    with tempfile.TemporaryDirectory() as d:
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC",
               "-shared", "-save-temps", "-o", os.path.join(d, "x.so"), SRC]
        subprocess.run(cmd, cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(d, "c_api-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    errors, found = [], 0
    i = 0
    while i < len(asm):
        m = re.match(r"^(_ZN2po12k_scan_probe\w+):", asm[i].split(";")[0].strip())
        if m:
            j = i
            while "s_endpgm" not in asm[j]:
                j += 1
            errors += check_kernel(m.group(1), asm[i:j + 1])
            found += 1
            i = j
        i += 1
    print("checked %d k_scan_probe kernels: %d problem(s)" % (found, len(errors)))
    for e in errors[:40]:
        print("  " + e)
    return 1 if (errors and "--strict" in sys.argv) or not found else 0


if __name__ == "__main__":
    sys.exit(main())
