"""Host side of the library under AddressSanitizer + UBSan (CPU build; sanitizers do not run on the GPU box): the
host-only tests -- packers and the two read stores, FASTA / GFA2 ingest threads, writers, result lifetimes, shard
arithmetic -- must run clean against the instrumented build (tools/asan_cpu_suite.sh runs the whole CPU suite)."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
@pytest.mark.skipif(not glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"), reason="no ASan runtime")
@pytest.mark.skipif(os.environ.get("PHASM_SKIP_ASAN_TEST") == "1", reason="already inside the sanitizer run")
@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "tools", "asan_cpu_suite.sh")),
                    reason="tools/asan_cpu_suite.sh does not travel to the GPU box (.gpurunignore): sanitizers run on the CPU build only")
def test_host_paths_are_clean_under_asan_and_ubsan(tmp_path):
    env = dict(os.environ, ASAN_OUT=str(tmp_path))
    out = subprocess.run([os.path.join(ROOT, "tools", "asan_cpu_suite.sh"), "tests/test_host_io.py", "tests/test_abi.py",
                          "-q", "-m", "not gpu", "-x"], env=env, capture_output=True, text=True, timeout=900)
    tail = out.stdout[-3000:] + out.stderr[-3000:]
    assert out.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in out.stdout
