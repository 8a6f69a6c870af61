"""What the GPU tests are checked against, reached through the checker process (oracle/sidecar.py).

``conftest.py`` starts that process before anything in pytest touches the GPU.  Everything here runs THERE:
the C oracle, the numpy layout restatement, the compiled reference, and every child command the tests need
(hipcc, python children) -- the process under test shares no heap with its checker and never forks.

``assert_same_rows`` is the comparison of the parity tests.  A mismatch is always a failure; the message
says which side is wrong, row by row, from the contract itself evaluated on the
read strings in plain Python -- independent of the HIP path AND of the oracle.
"""
from __future__ import annotations

import os
from collections import Counter
from typing import List, Optional, Sequence

import numpy as np

_sidecar = None


def start() -> None:
    global _sidecar
    if _sidecar is None:
        from oracle.sidecar import Sidecar
        _sidecar = Sidecar()


def stop() -> None:
    global _sidecar
    if _sidecar is not None:
        _sidecar.close()
        _sidecar = None


def sidecar():
    if _sidecar is None:  # (a test module run outside the normal session: late start, still a separate process)
        start()
    return _sidecar


def _plain(seqs: Sequence) -> List[bytes]:
    return [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]


def oracle_overlaps(seqs: Sequence, min_length: int) -> np.ndarray:
    """``oracle.overlap_oracle.oracle_overlaps`` in the checker process: sorted (n, 6) int64 rows."""
    return sidecar().call("oracle.overlap_oracle", "oracle_overlaps", _plain(seqs), int(min_length))


def reference_overlaps(seqs: Sequence, min_length: int, quiet: bool = False):
    return sidecar().call("oracle.overlap_oracle", "reference_overlaps", _plain(seqs), int(min_length), quiet)


def have_reference() -> bool:
    return sidecar().call("oracle.overlap_oracle", "have_reference")


def layout_vectorised(rows, lengths, **kwargs) -> dict:
    return sidecar().call("oracle.layout_oracle", "layout_vectorised", np.asarray(rows), np.asarray(lengths), **kwargs)


def layout_sequential(rows, lengths, **kwargs) -> dict:
    """``oracle.layout_oracle.layout_sequential`` in the checker process, plus ``edges_array`` = its edge dict as the
    sorted (n, 4) array the tests compare."""
    rows = [tuple(int(x) for x in r) for r in rows]
    want = sidecar().call("oracle.layout_oracle", "layout_sequential", rows, [int(x) for x in lengths], **kwargs)
    want["edges_array"] = sidecar().call("oracle.layout_oracle", "edges_dict_to_array", want["edges"])
    return want


def run(argv, **kwargs):
    """``subprocess.run`` in the checker process -> (returncode, stdout, stderr)."""
    return sidecar().run(list(argv), **kwargs)


def expected_multiplicity(seqs: Sequence[bytes], m: int, row) -> int:
    """How often the contract puts ``row`` into overlaps(m): 0, 1 or 2 (A and B family are not de-duplicated).
    Plain Python on the read strings (overlapper.cpp:64-116 as SURVEY.md section 8a-2 states it)."""
    a, b, s, e, bs, be = (int(x) for x in row)
    m = max(int(m), 1)
    if a == b or not (0 <= a < len(seqs)) or not (0 <= b < len(seqs)) or bs != 0:
        return 0
    A, B = seqs[a], seqs[b]
    la, lb = len(A), len(B)
    l = be
    if not (0 <= s < e <= la) or e - s != l or l < m or l > lb or A[s:e] != B[:l]:
        return 0
    n = 0
    if e == la:  # A family: the LONGEST suffix of a that is a prefix of b
        longest = next((k for k in range(min(la, lb), m - 1, -1) if A[la - k:] == B[:k]), None)
        n += 1 if longest == l else 0
    if l == lb:  # B family: every occurrence of the whole of b
        n += 1
    return n


def explain_difference(got: np.ndarray, want: np.ndarray, seqs: Optional[Sequence], m: int, limit: int = 12) -> str:
    cg, cw = Counter(map(tuple, got.tolist())), Counter(map(tuple, want.tolist()))
    diff = sorted(set((cg - cw).keys()) | set((cw - cg).keys()))
    lines = ["HIP %d rows, checker %d rows, %d distinct rows differ" % (len(got), len(want), len(diff))]
    hip_wrong = chk_wrong = 0
    plain = _plain(seqs) if seqs is not None else None
    for row in diff[:limit]:
        line = "  row %s: HIP x%d, checker x%d" % (row, cg[row], cw[row])
        if plain is not None:
            exp = expected_multiplicity(plain, m, row)
            hip_wrong += cg[row] != exp
            chk_wrong += cw[row] != exp
            line += ", contract x%d -> %s" % (exp, "HIP WRONG" if cg[row] != exp else "CHECKER WRONG")
        lines.append(line)
    if plain is not None:
        lines.append("verdict over the rows shown: HIP wrong on %d, checker wrong on %d" % (hip_wrong, chk_wrong))
    return "\n".join(lines)


# what the guards of a session saw (conftest.py writes it to gpurun_out/guard_summary.json at the end of a GPU session)
GUARD_TALLY = {"guards": 0, "pages_checked": 0, "reused_addresses_of_freed_pinned_memory": 0, "verified_unchanged": 0}


class GuardedReads:
    """The reads of one GPU parity call, held so that any writer into them is either caught in the act or classified.

    * The bytes live in an anonymous mapping of their own, made READ-ONLY (``mprotect(PROT_READ)``) before the library
      sees a pointer into it: a CPU store by anything in this process -- the library, the HIP runtime's threads, Python,
      numpy -- faults AT the store (``faulthandler`` prints every thread's stack, conftest.py).  A change that still
      appears can only have come from outside the CPU's page protection: a DMA or a device-side store.
    * ``check_not_gpu_visible``: a DMA or device store needs the page mapped for the GPU.  Every page of the mapping is
      looked up in the HIP runtime (``hipPointerGetAttributes``) and in ROCr underneath it (``hsa_amd_pointer_info``, which
      also sees the pins HIP takes by itself for pageable copies) through ``po_debug_pointer_info``; and against the list
      of every host range the library has EVER pinned or registered in this process (``po_debug_host_ranges``).
    * ``verify``: the mapping, a heap copy and a digest taken when the reads were made are compared after the call; the
      message names which of the three changed (mapping changed = DMA / device store; heap copy changed = a stray store
      into this process's heap).
    """

    def __init__(self, seqs: Sequence):
        import ctypes
        import hashlib
        import mmap
        self._plain = _plain(seqs)
        self._digest = [hashlib.blake2b(s, digest_size=16).digest() for s in self._plain]
        self._copy = [bytes(bytearray(s)) for s in self._plain]   # (separate heap objects)
        self._off = []
        off = 0
        for s in self._plain:
            self._off.append(off)
            off += (len(s) + 7) & ~7   # (8-byte alignment only: reads start anywhere inside their pages, like heap objects)
        self._size = max(mmap.PAGESIZE, (off + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE)
        self._mm = mmap.mmap(-1, self._size, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, prot=mmap.PROT_READ | mmap.PROT_WRITE)
        for o, s in zip(self._off, self._plain):
            self._mm[o:o + len(s)] = s
        self._anchor = ctypes.c_char.from_buffer(self._mm)
        self.base = ctypes.addressof(self._anchor)
        self._libc = ctypes.CDLL(None, use_errno=True)
        self._libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        if self._libc.mprotect(self.base, self._size, mmap.PROT_READ) != 0:
            raise OSError(ctypes.get_errno(), "mprotect(PROT_READ)")
        self._visible_log = []

    def __len__(self) -> int:
        return len(self._plain)

    def address(self, i: int) -> int:
        return self.base + self._off[i]

    def length(self, i: int) -> int:
        return len(self._plain[i])

    def add_all(self, ov, fmt: str = "r%d") -> None:
        """``po_add_sequence`` for every read, with pointers into the read-only mapping."""
        for i in range(len(self)):
            ov.add_sequence_ptr(fmt % i, self.address(i), self.length(i))
        self.check_not_gpu_visible("after po_add_sequence")

    def check_not_gpu_visible(self, when: str) -> None:
        import ctypes
        import mmap
        from phasm_amd import _lib
        lib = _lib.load()
        ht, st = ctypes.c_int32(), ctypes.c_int32()
        b, n = ctypes.c_uint64(), ctypes.c_uint64()
        for page in range(self.base, self.base + self._size, mmap.PAGESIZE):
            if lib.po_debug_pointer_info(ctypes.c_void_p(page), ctypes.byref(ht), ctypes.byref(st), ctypes.byref(b), ctypes.byref(n)):
                self._visible_log.append((when, page, ht.value, st.value, b.value, n.value))
        cap = 1 << 16
        buf = (ctypes.c_uint64 * (3 * cap))()
        k = min(int(lib.po_debug_host_ranges(buf, cap)), cap)
        lo, hi = self.base, self.base + self._size
        reused = False
        for i in range(k):
            rb, rn, kind = buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]
            if rb < hi and lo < rb + rn:
                if kind & 0x100:     # a LIVE pinned / registered range of the library covers an input page
                    self._visible_log.append((when, rb, -2, int(kind), rb, rn))
                else:                # a range the library has given back: this mapping reuses its addresses (tallied, not an error)
                    reused = True
        if when == "after po_add_sequence":
            GUARD_TALLY["guards"] += 1
            GUARD_TALLY["pages_checked"] += self._size // mmap.PAGESIZE
            GUARD_TALLY["reused_addresses_of_freed_pinned_memory"] += int(reused)
        if self._visible_log:
            _evidence("gpu_visible_inputs_%d.txt" % os.getpid(),
                      "\n".join("%s: page 0x%x hip_type %d hsa_type %d range 0x%x + %d" % e for e in self._visible_log) + "\n")
            raise AssertionError("pages of the reads handed to po_add_sequence are known to the GPU runtime %s: %r"
                                 % (when, self._visible_log[:4]))

    def verify(self) -> None:
        import hashlib
        self.check_not_gpu_visible("after the call")
        for i, (o, s) in enumerate(zip(self._off, self._plain)):
            now = bytes(self._mm[o:o + len(s)])
            d = self._digest[i]
            sides = {"read-only mapping (only a DMA / device store gets past PROT_READ)": now,
                     "heap copy (a stray store into this process's heap)": self._copy[i],
                     "the test's own object": s}
            bad = [name for name, v in sides.items() if hashlib.blake2b(v, digest_size=16).digest() != d]
            if bad:
                ref = next((v for v in sides.values() if hashlib.blake2b(v, digest_size=16).digest() == d), None)
                detail = []
                for name in bad:
                    v = sides[name]
                    where = [k for k in range(len(v)) if ref is not None and v[k] != ref[k]]
                    detail.append("%s: %d byte offsets differ, first %s" % (name, len(where), where[:8]))
                dump = os.environ.get("PHASM_MISMATCH_DIR")
                if dump:
                    os.makedirs(dump, exist_ok=True)
                    np.savez_compressed(os.path.join(dump, "input_changed_%d.npz" % os.getpid()), index=i,
                                        mapping=np.frombuffer(now, dtype=np.uint8), heap_copy=np.frombuffer(self._copy[i], dtype=np.uint8),
                                        original=np.frombuffer(s, dtype=np.uint8))
                raise AssertionError("host memory of the test process changed under the call: read %d (%d bytes): %s"
                                     % (i, len(s), "; ".join(detail)))
        GUARD_TALLY["verified_unchanged"] += 1

    def close(self) -> None:
        if self._mm is not None:
            self._anchor = None
            try:
                self._mm.close()
            except BufferError:
                pass
            self._mm = None

    def verify_and_close(self) -> None:
        try:
            self.verify()
        finally:
            self.close()


def _evidence(name: str, text: str) -> None:
    dump = os.environ.get("PHASM_MISMATCH_DIR")
    if dump:
        os.makedirs(dump, exist_ok=True)
        with open(os.path.join(dump, name), "a") as fh:
            fh.write(text)


def host_ranges() -> List[tuple]:
    """Every host range the library has made visible to the GPU in this process: (base, bytes, kind, live)."""
    import ctypes
    from phasm_amd import _lib
    lib = _lib.load()
    cap = 1 << 16
    buf = (ctypes.c_uint64 * (3 * cap))()
    k = min(int(lib.po_debug_host_ranges(buf, cap)), cap)
    return [(int(buf[3 * i]), int(buf[3 * i + 1]), int(buf[3 * i + 2]) & 0xFF, bool(buf[3 * i + 2] & 0x100)) for i in range(k)]


def snapshot(seqs: Sequence) -> List[bytes]:
    """Real copies (separate memory) of the reads a test is about to hand to the library."""
    return [bytes(bytearray(s)) for s in _plain(seqs)]


def assert_inputs_unchanged(seqs, snap) -> None:
    """The reads a test hands to the library are immutable Python objects; the rows are checked against the SAME objects
    afterwards (checker process + the contract in plain Python).  If one of them reads differently after the call than
    before it, the comparison that follows would blame whichever side saw the other version -- say so instead, with the
    bytes that changed (seen once in round 3: the library's rows equalled the oracle's on the regenerated reads, while the
    test process's own copy of ONE read no longer did)."""
    for i, (s, c) in enumerate(zip(_plain(seqs), snap)):
        if s != c:
            sb, cb = bytes(s), bytes(c)
            where = [k for k in range(min(len(sb), len(cb))) if sb[k] != cb[k]]
            dump = os.environ.get("PHASM_MISMATCH_DIR")
            if dump:
                os.makedirs(dump, exist_ok=True)
                np.savez_compressed(os.path.join(dump, "input_changed_%d.npz" % os.getpid()), index=i,
                                    before=np.frombuffer(cb, dtype=np.uint8), after=np.frombuffer(sb, dtype=np.uint8))
            raise AssertionError("host memory of the test process changed under the call: read %d (%d bytes) differs from its "
                                 "copy taken before the call at %d byte offsets, first %s: before %r after %r"
                                 % (i, len(cb), len(where), where[:8], cb[where[0]:where[0] + 16] if where else b"",
                                    sb[where[0]:where[0] + 16] if where else b""))


def assert_same_rows(got: np.ndarray, want: np.ndarray, seqs: Optional[Sequence] = None, m: int = 1, ctx="") -> None:
    """Sorted-multiset equality of (n, 6) row arrays; a mismatch fails with a per-row verdict."""
    if got.shape == want.shape and np.array_equal(got, want):
        return
    dump = os.environ.get("PHASM_MISMATCH_DIR")
    if dump:
        os.makedirs(dump, exist_ok=True)
        np.savez_compressed(os.path.join(dump, "mismatch_%d.npz" % os.getpid()), got=got, want=want, m=m,
                            seqs=np.array(_plain(seqs), dtype=object) if seqs is not None else np.array([], dtype=object))
    raise AssertionError("%s\n%s" % (ctx, explain_difference(got, want, seqs, m)))


def oracle_overlaps_ex(seqs: Sequence, min_length: int, max_diff: int, band: int, anchor: int = 32) -> np.ndarray:
    """``oracle.extend_oracle.oracle_overlaps_ex`` (CPU restatement of po_overlaps_ex) in the checker process."""
    return sidecar().call("oracle.extend_oracle", "oracle_overlaps_ex", _plain(seqs), int(min_length), int(max_diff),
                          int(band), int(anchor))
