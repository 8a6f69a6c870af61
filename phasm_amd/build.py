"""Build libphasm_overlap.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "c_api.hip")
DEPS = ([SRC, os.path.join(ROOT, "include", "phasm_overlap.h")] +
        sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith(".h")))
LIB = os.path.join(HERE, "libphasm_overlap.so")
ARCH = "gfx950"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


# k_scan_probe waits for its inline-asm loads with hand-counted s_waitcnt; tools/check_scan_isa.py walks the generated
# code for reads of a register whose load is still in flight.  The walk was validated against the code these hipcc
# versions generate; another compiler may lay the kernel out differently, so the build says so loudly and the CPU
# test (tests/test_kernel_resources.py) fails until the checker has been looked at again.
VALIDATED_HIPCC = ("7.2.26015",)


def hipcc_version(hipcc: str = None) -> str:
    hipcc = hipcc or os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    for line in out.splitlines():
        if line.startswith("HIP version:"):
            return line.split(":", 1)[1].strip().split("-")[0]
    return "unknown"


def check_scan_isa() -> None:
    """Fails (CalledProcessError) when the generated k_scan_probe reads a landing register before its wait."""
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "check_scan_isa.py"), "--strict"],
                          stdout=subprocess.DEVNULL)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-shared",
           "-Wall", "-Wno-unused-result", "-o", LIB, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    if not os.environ.get("PHASM_SKIP_ISA_CHECK"):
        try:
            check_scan_isa()   # a library whose scan pipeline reads in-flight registers must not ship
        except subprocess.CalledProcessError:
            os.remove(LIB)
            raise RuntimeError("k_scan_probe: tools/check_scan_isa.py --strict found a read of an in-flight register; library removed")
        v = hipcc_version(hipcc)
        if v not in VALIDATED_HIPCC and not os.environ.get("PHASM_ALLOW_UNVALIDATED_HIPCC"):
            os.remove(LIB)
            raise RuntimeError("hipcc %s is not one of %s: k_scan_probe is hand-scheduled around this compiler's code -- re-validate "
                               "tools/check_scan_isa.py against the new compiler's output, then add the version to VALIDATED_HIPCC "
                               "(PHASM_ALLOW_UNVALIDATED_HIPCC=1 builds anyway); library removed" % (v, VALIDATED_HIPCC))
    return LIB


PYT_SRC = os.path.join(HERE, "csrc", "pytuples.c")
PYT_LIB = os.path.join(HERE, "_pytuples.so")


def build_pytuples(force: bool = False):
    """The native list-of-tuples builder behind ExactOverlapper.overlaps() (plain C against Python.h; optional: without it
    the shim builds the list in Python).  Returns the path, or None when it cannot be built here."""
    import sysconfig
    if not force and os.path.exists(PYT_LIB) and os.path.getmtime(PYT_LIB) >= os.path.getmtime(PYT_SRC):
        return PYT_LIB
    inc = sysconfig.get_paths().get("include")
    if not inc or not os.path.exists(os.path.join(inc, "Python.h")):
        return None
    cmd = [os.environ.get("CC", "gcc"), "-O2", "-fPIC", "-shared", "-I" + inc, "-o", PYT_LIB, PYT_SRC]
    if subprocess.call(cmd) != 0:
        return None
    return PYT_LIB


if __name__ == "__main__":
    build_pytuples(force=True)
    print(build_library(force=True, verbose="-v" in sys.argv))
