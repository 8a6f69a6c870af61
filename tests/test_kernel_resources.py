"""Compile-time guard on the hot kernels' register budget (hipcc cross-compiles gfx950 without a GPU).

A harmless-looking edit to a helper that the scan kernels inline (a different canonical-pair rule) once
took k_scan_fill from 54 to 121 VGPRs and k_scan_probe into scratch, costing 8 % of the step before any
test noticed: parity tests cannot see that, this one can."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel (mangled-name fragment) -> (max VGPRs, why)
BUDGET = {
    "k_scan_probeILi2ELb1": (128, "16 waves per CU: one persistent 1024-thread workgroup"),
    "k_scan_probeILi2ELb0": (128, "same"),
    "k_scan_fillILi2": (64, "8 waves per SIMD"),
    "k_verify_aILi2ELb": (64, "8 waves per SIMD; LDS-limited beyond that"),
    "k_scan_fixupILi2": (64, ""),
    "k_wide_scanILi2ELb0": (64, ""),
    "k_wide_scanILi2ELb1": (64, ""),
    "k_emit": (64, ""),
    "k_layout_classify": (64, ""),
    "k_layout_insert": (64, ""),
    "k_layout_winner": (64, ""),
}


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_hot_kernels_stay_in_registers(tmp_path):
    src = os.path.join(ROOT, "phasm_amd", "csrc", "c_api.hip")
    out = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-c", src, "-o",
                          str(tmp_path / "c_api.o"), "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    usage = {}
    name = None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and name:
            usage[name][m.group(1).split(" ")[0]] = int(m.group(2))
    assert len(usage) > 20
    for frag, (max_vgpr, why) in BUDGET.items():
        hits = {k: v for k, v in usage.items() if frag in k}
        assert hits, "kernel %s not found in the compiler remarks" % frag
        for k, v in hits.items():
            assert v["ScratchSize"] == 0, "%s spills to scratch (%d bytes/lane)" % (k, v["ScratchSize"])
            assert v["VGPRs"] <= max_vgpr, "%s uses %d VGPRs (budget %d: %s)" % (k, v["VGPRs"], max_vgpr, why)
    # nothing in the library may spill
    spills = {k: v["ScratchSize"] for k, v in usage.items() if v.get("ScratchSize")}
    assert not spills, spills


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_scan_pipeline_never_reads_a_register_whose_load_is_in_flight():
    """k_scan_probe issues its steady-state loads with inline asm and waits for them with counted s_waitcnt
    statements; hipcc does not know those registers are pending.  Two asm waits in the arms of an `if` once made
    it copy the landing registers in FRONT of the wait (stale data, and the parity tests still passed on that
    build): tools/check_scan_isa.py walks the generated code for exactly that."""
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_scan_isa.py"), "--strict"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert "0 problem(s)" in out.stdout


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_the_isa_walk_was_validated_for_this_compiler():
    """The walk above is only as good as its reading of the compiler's output.  A hipcc whose code for k_scan_probe nobody
    has looked at is a FAILURE of its own, by name and with instructions (the hand-scheduled scan relies on the compiler
    leaving the landing registers of in-flight loads alone: a new compiler must not ship a library unseen)."""
    from phasm_amd import build
    v = build.hipcc_version(shutil.which("hipcc"))
    assert v in build.VALIDATED_HIPCC, (
        "hipcc %s: look at k_scan_probe's generated code again (tools/check_scan_isa.py), then add the version to "
        "phasm_amd/build.py:VALIDATED_HIPCC" % v)
