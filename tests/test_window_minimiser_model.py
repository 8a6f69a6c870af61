"""A lane-by-lane Python model of the window-minimiser selection of the wide index (phasm_amd/csrc/kernels.hip.h:
WideEnc, window_key, window_selected, k_wide_insert, k_wide_scan) -- the arithmetic of the HIP code restated with
numpy, checked against the DEFINITION it implements:

* a word of a read is selected iff it is the (leftmost) minimum of some window of WW consecutive complete words;
* every suffix-prefix overlap and every containment of >= m bases (m >= W WW + W - 1) is found, exactly once, by probing
  only the selected words of a against an index that holds, per read b and phase j, the minimiser of the first WW
  word-spaced K-mers of that phase.

The HIP kernels themselves are held against the reference goldens and the full-size truth on the GPU
(tests/test_gpu_parity.py, tests/test_gpu_fullsize.py); this file guards the reasoning they were written from."""
import numpy as np
import pytest

W = 32
INF = 0xFFFFFFFF


def kmer_hash(k):
    lo, hi = k & 0xFFFFFFFF, k >> 32
    x = lo ^ (((hi << 13) | (hi >> 19)) & 0xFFFFFFFF)
    h1 = (x * 0x9E3779B1) & 0xFFFFFFFF
    h2 = ((x ^ (x >> 15) ^ ((hi * 5) & 0xFFFFFFFF)) * 0x85EBCA77) & 0xFFFFFFFF
    return h1, h2


def window_key(h2, pos):
    return (h2 & ~127 & 0xFFFFFFFF) | pos


def shfl(v, idx):
    return v[idx & 63]


def window_selected(own, side, WW):
    """own[64], side[64] as the kernel holds them -> selected[64]; the kernel's statements, lane by lane."""
    PAD = WW - 1
    lane = np.arange(64)
    up = shfl(own, lane - PAD)
    dn = shfl(own, lane + 64 - PAD)
    r0 = np.where(lane < PAD, side, up)
    r1 = np.where(lane < PAD, dn, np.where(lane < 2 * PAD, side, INF))
    k = 1
    while k < WW:
        x0, x1 = shfl(r0, lane + k), shfl(r1, lane + k)
        n0 = np.where(lane + k < 64, x0, x1)
        n1 = np.where(lane + k < 2 * PAD, x1, INF)
        r0, r1 = np.minimum(r0, n0), np.minimum(r1, n1)
        k <<= 1
    k = 1
    while k < WW:
        y0, y1 = shfl(r0, lane - k), shfl(r1, lane - k)
        m0 = np.where(lane >= k, y0, 0)
        m1 = np.where(lane >= k, y1, y0)
        r0, r1 = np.maximum(r0, m0), np.maximum(r1, m1)
        k <<= 1
    a0, a1 = shfl(r0, lane + PAD), shfl(r1, lane + PAD)
    y = np.where(lane + PAD < 64, a0, a1)
    return (own != INF) & (y == own)


def pack(seq):
    """2-bit words of a read (base i at bits [2i, 2i+2) of word i // 32), plus two zero guard words."""
    codes = np.frombuffer(seq.translate(bytes.maketrans(b"ACGT", b"\0\1\2\3")), dtype=np.uint8).astype(np.uint64)
    nw = (len(seq) + 31) // 32
    words = [0] * (nw + 2)
    for i, c in enumerate(codes.tolist()):
        words[i // 32] |= c << (2 * (i % 32))
    return words


def kmer_at(words, off):
    w, sh = off // 32, 2 * (off % 32)
    lo, hi = words[w], words[w + 1]
    return ((lo >> sh) | (hi << (64 - sh))) & 0xFFFFFFFFFFFFFFFF if sh else lo


def selected_words(words, la, WW):
    """The kernel's selection over a whole read, tile by tile (64 words per tile)."""
    n_words = la // W
    PAD = WW - 1
    keys = [kmer_hash(words[i])[1] for i in range(n_words)]
    out = np.zeros(n_words, dtype=bool)
    for word0 in range(0, max(n_words, 1), 64):
        own = np.full(64, INF, dtype=np.int64)
        side = np.full(64, INF, dtype=np.int64)
        for lane in range(64):
            wi = word0 + lane
            if wi < n_words:
                own[lane] = window_key(keys[wi], PAD + lane)
            if lane < 2 * PAD:
                sw = word0 - PAD + lane if lane < PAD else word0 + 64 + lane - PAD
                if 0 <= sw < n_words:
                    side[lane] = window_key(keys[sw], lane if lane < PAD else 64 + lane)
        sel = window_selected(own, side, WW)
        for lane in range(64):
            if word0 + lane < n_words:
                out[word0 + lane] = sel[lane]
    return out, keys


def minimisers_by_definition(keys, WW):
    """words that are the leftmost minimum of (hash with its low 7 bits cleared, position) over some full window"""
    n = len(keys)
    out = np.zeros(n, dtype=bool)
    t = [k & ~127 for k in keys]
    for s in range(0, n - WW + 1):
        win = t[s:s + WW]
        out[s + win.index(min(win))] = True
    return out


@pytest.mark.parametrize("WW", [4, 16])
def test_lane_arithmetic_selects_every_window_minimiser(WW):
    rng = np.random.default_rng(WW)
    n_extra = 0
    for trial in range(30):
        la = int(rng.choice([W * WW + W - 1, 700, 2048, 2049, 4096 + 31, 15000, 64 * 32, 65 * 32, 129 * 32 + 5]))
        seq = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=la))
        if trial % 5 == 0:           # low complexity: equal hashes inside windows, ties decided by position
            unit = seq[:int(rng.integers(1, 40))]
            seq = (unit * (la // len(unit) + 1))[:la]
        words = pack(seq)
        sel, keys = selected_words(words, la, WW)
        want = minimisers_by_definition(keys, WW)
        assert not (want & ~sel).any(), (trial, la)          # every true minimiser is selected (no overlap can be lost)
        n_extra += int((sel & ~want).sum())                   # (over-selection near the read's ends only costs probes)
        assert (sel & ~want).sum() <= 2 * (WW - 1)
        if la // W >= 4 * WW and trial % 5:
            assert sel.mean() < 0.5                           # ... and most words are not probed
    assert n_extra < 30 * 2 * (WW - 1)


@pytest.mark.parametrize("WW", [4, 16])
def test_every_overlap_is_found_exactly_once(WW):
    """Index: per read b and phase j the minimiser of the K-mers at j + 32 t, t < WW.  Scan: selected words of a.  Every
    true (a, p, b) with overlap >= m must be produced by exactly one probe."""
    rng = np.random.default_rng(100 + WW)
    m = W * WW + W - 1
    for trial in range(6):
        glen = int(rng.integers(3 * m, 6 * m))
        genome = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=glen))
        reads = []
        for _ in range(14):
            ln = int(rng.integers(m, min(glen, 3 * m)))
            st = int(rng.integers(0, glen - ln + 1))
            reads.append(genome[st:st + ln])
        reads.append(reads[0])                                # an identical read
        packed = [pack(r) for r in reads]
        index = {}
        for b, (r, wb) in enumerate(zip(reads, packed)):
            for j in range(W):
                best = None
                for t in range(WW):
                    k = kmer_at(wb, j + W * t)
                    c = window_key(kmer_hash(k)[1], t)
                    if best is None or c < best[0]:
                        best = (c, k, j + W * t)
                index.setdefault(best[1], []).append((b, best[2]))
        found = {}
        for a, (r, wa) in enumerate(zip(reads, packed)):
            sel, _ = selected_words(wa, len(r), WW)
            for wi in np.flatnonzero(sel).tolist():
                for b, o in index.get(wa[wi], []):
                    q = wi * W
                    if b != a and o <= q:
                        found[(a, q - o, b)] = found.get((a, q - o, b), 0) + 1
        assert all(v == 1 for v in found.values())
        for a, ra in enumerate(reads):
            for b, rb in enumerate(reads):
                if a == b:
                    continue
                for p in range(0, len(ra) - m + 1):
                    l = min(len(ra) - p, len(rb))
                    if l >= m and ra[p:p + l] == rb[:l]:      # suffix of a = prefix of b, or b inside a
                        assert (a, p, b) in found, (trial, a, p, b)
