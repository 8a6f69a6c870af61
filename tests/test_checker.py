"""The checker side of the GPU parity tests, checked on the CPU: the checker process answers like the in-process
oracle, and the plain-Python contract evaluator that assert_same_rows uses to say which side of a mismatch is
wrong agrees with the reference's own outputs (every small golden)."""
import os
from collections import Counter

import numpy as np
import pytest

import checker as ck
import golden_utils as gu
from oracle import overlap_oracle as oo


@pytest.fixture(scope="module", autouse=True)
def _checker_process():
    mine = ck._sidecar is None   # (a session that runs GPU tests has started it already and keeps it)
    ck.start()
    yield
    if mine:
        ck.stop()


def test_checker_process_is_another_process_and_agrees_with_the_oracle():
    import os
    assert ck.sidecar().pid != os.getpid()
    for name, seqs, m, want in gu.all_small_cases()[:60]:
        assert np.array_equal(ck.oracle_overlaps(seqs, m), want), name
    rc, out, _ = ck.run(["/bin/echo", "hello"], capture_output=True, text=True)
    assert rc == 0 and out == "hello\n"


def test_contract_evaluator_matches_the_reference_goldens():
    """Every golden row has the multiplicity the evaluator computes; rows the reference does not emit get 0."""
    for name, seqs, m, want in gu.all_small_cases():
        seqs = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]
        cnt = Counter(map(tuple, want.tolist()))
        for row, k in cnt.items():
            assert ck.expected_multiplicity(seqs, m, row) == k, (name, row)
        for row in list(cnt)[:5]:   # neighbours of true rows that are not rows
            a, b, s, e, bs, be = row
            if e - s > max(m, 1) and (a, b, s + 1, e, 0, be - 1) not in cnt:
                assert ck.expected_multiplicity(seqs, m, (a, b, s + 1, e, 0, be - 1)) == 0, (name, row)


def test_mismatch_is_a_failure_with_a_verdict(tmp_path, monkeypatch):
    # (this mismatch is deliberate: its dump goes to the test's own directory -- gpurun_out/mismatch/ holds real evidence only)
    monkeypatch.setenv("PHASM_MISMATCH_DIR", str(tmp_path))
    name, seqs, m, want = next(c for c in gu.all_small_cases() if len(c[3]) >= 2)
    short = want[1:]
    with pytest.raises(AssertionError) as ei:
        ck.assert_same_rows(want, short, seqs, m, "ctx")
    assert "CHECKER WRONG" in str(ei.value) and "HIP wrong on 0" in str(ei.value)
    with pytest.raises(AssertionError) as ei:
        ck.assert_same_rows(short, want, seqs, m, "ctx")
    assert "HIP WRONG" in str(ei.value) and "checker wrong on 0" in str(ei.value)


# ---- GuardedReads: the instruments of DESIGN.md section 6.1 (b) and (c), checked on the CPU -----------------------------

_STORE_INTO_GUARDED = r"""
import ctypes, faulthandler, sys
sys.path.insert(0, %r)
sys.path.insert(0, %r)
faulthandler.enable(all_threads=True)
from phasm_amd import _lib
assert _lib.load().po_debug_fault_backtrace(2) == 0    # (installed AFTER faulthandler here: it runs first, then chains to it)
import checker as ck
g = ck.GuardedReads([b"ACGTACGTAC" * 50, b"TTTTGGGGCC" * 30])
assert bytes((ctypes.c_char * 10).from_address(g.address(1))) == b"TTTTGGGGCC"   # reading is fine
print("reading ok", flush=True)
ctypes.memset(g.address(1) + 5, 0x41, 1)     # a CPU store into a read handed to the library
print("NOT REACHED", flush=True)
"""


def test_a_cpu_store_into_guarded_reads_faults_at_the_store():
    import os
    import signal
    import subprocess
    import sys
    p = subprocess.run([sys.executable, "-c", _STORE_INTO_GUARDED % (os.path.dirname(os.path.abspath(__file__)),
                                                                 os.path.dirname(os.path.dirname(os.path.abspath(__file__))))],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == -signal.SIGSEGV, (p.returncode, p.stdout, p.stderr)
    assert "reading ok" in p.stdout and "NOT REACHED" not in p.stdout
    assert "Segmentation fault" in p.stderr and "in <module>" in p.stderr   # faulthandler: the Python frame of the store
    assert "[phasm] signal 11 at address 0x" in p.stderr and "native stack" in p.stderr   # ... and the native one, with the address


def test_guarded_reads_name_which_copy_changed(tmp_path, monkeypatch):
    import ctypes
    import mmap
    monkeypatch.setenv("PHASM_MISMATCH_DIR", str(tmp_path))
    seqs = [b"ACGT" * 100, "GGCC" * 77, b"N" * 5]
    g = ck.GuardedReads(seqs)
    assert len(g) == 3 and g.length(1) == 308
    assert bytes((ctypes.c_char * 5).from_address(g.address(2))) == b"NNNNN"
    g.verify()                                     # nothing changed (and no page is known to a GPU runtime here)
    # what a DMA would do: the mapping changes although no CPU store can reach it (here: protection lifted for the test)
    g._libc.mprotect(g.base, g._size, mmap.PROT_READ | mmap.PROT_WRITE)
    ctypes.memset(g.address(1) + 7, ord("T"), 2)
    with pytest.raises(AssertionError) as ei:
        g.verify()
    assert "read 1" in str(ei.value) and "read-only mapping" in str(ei.value) and "[7, 8]" in str(ei.value)
    assert "heap copy" not in str(ei.value).split("read 1")[1].split("read-only mapping")[0]
    assert any(f.startswith("input_changed_") for f in os.listdir(tmp_path))
    g.close()
    # ... and a stray store into the heap: the heap copy differs, the mapping does not
    g = ck.GuardedReads(seqs)
    g._copy[0] = g._copy[0][:10] + b"X" + g._copy[0][11:]
    with pytest.raises(AssertionError) as ei:
        g.verify()
    assert "read 0" in str(ei.value) and "heap copy" in str(ei.value) and "read-only mapping" not in str(ei.value)
    g.close()


def test_pin_registry_and_pointer_info_without_a_gpu():
    """po_debug_host_ranges / po_debug_pointer_info answer without a device: nothing registered, nothing known."""
    import ctypes
    from phasm_amd import _lib
    lib = _lib.load()
    buf = (ctypes.c_uint64 * 30)()
    n = lib.po_debug_host_ranges(buf, 10)
    assert n >= 0
    x = ctypes.create_string_buffer(4096)
    ht, st = ctypes.c_int32(7), ctypes.c_int32(7)
    b, nn = ctypes.c_uint64(1), ctypes.c_uint64(1)
    known = lib.po_debug_pointer_info(ctypes.addressof(x), ctypes.byref(ht), ctypes.byref(st), ctypes.byref(b), ctypes.byref(nn))
    assert known == 0 and b.value == 0 and nn.value == 0
