"""Seeded synthetic read sets for tests and ``bench.py`` (new code).

Generator contract (every report states the parameters):

* haploid base genome: i.i.d. uniform ``ACGT`` of length ``G``;
* haplotypes 2..p: copies of the base with independent substitutions at rate ``snp``
  (default 0.005), each by a uniformly chosen *different* base;
* reads: for i in 0..N-1 pick a haplotype uniformly, a start uniformly in [0, G-L], length
  ``L`` (fixed) or N(mean, sd) clipped (config 1), strand by fair coin (reverse complement
  if tails); name ``read{i}``;
* optional substitution noise at rate ``noise`` after strand choice (config 4).

``numpy.random.default_rng(seed)`` (PCG64) drives everything, so a (config, seed) pair names
one exact FASTA on every machine with the same numpy major version.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import List, Optional, Tuple

import numpy as np

_ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)


@dataclass(frozen=True)
class SynthConfig:
    n_reads: int
    read_len: int            # fixed length, or mean when len_sd > 0
    genome_len: int
    ploidy: int = 2
    snp: float = 0.005
    seed: int = 1
    noise: float = 0.0
    len_sd: float = 0.0
    len_min: int = 2000
    len_max: int = 20000

    def describe(self) -> dict:
        return asdict(self)


# BASELINE.json configs (SURVEY.md section 8d "Concrete configs"); min_length 1000 everywhere.
CONFIGS = {
    "cfg1": SynthConfig(n_reads=1000, read_len=10000, genome_len=500_000, ploidy=2, seed=1,
                        len_sd=1500.0),
    "cfg2": SynthConfig(n_reads=50_000, read_len=15000, genome_len=5_000_000, ploidy=2, seed=2),
    "cfg3": SynthConfig(n_reads=200_000, read_len=15000, genome_len=20_000_000, ploidy=3, seed=3),
    "cfg4": SynthConfig(n_reads=50_000, read_len=15000, genome_len=5_000_000, ploidy=2, seed=4,
                        noise=0.01),
    "cfg5": SynthConfig(n_reads=1_000_000, read_len=12000, genome_len=100_000_000, ploidy=4, seed=5),
}


def scaled(cfg: SynthConfig, n_reads: int) -> SynthConfig:
    """Same coverage per haplotype (hence the same overlaps per read) at a smaller read count."""
    g = max(int(round(cfg.genome_len * (n_reads / cfg.n_reads))), cfg.read_len * 2)
    return SynthConfig(**{**asdict(cfg), "n_reads": n_reads, "genome_len": g})


def _substitute(codes: np.ndarray, rate: float, rng: np.random.Generator) -> np.ndarray:
    if rate <= 0.0:
        return codes
    hit = rng.random(codes.shape[0]) < rate
    k = int(hit.sum())
    out = codes.copy()
    out[hit] = (codes[hit] + rng.integers(1, 4, size=k, dtype=np.uint8)) & 3
    return out


@dataclass
class Truth:
    """Where every read of a (config, seed) pair comes from: what ``generate_codes`` draws before it cuts the reads."""
    haps: List[np.ndarray]       # haplotype code arrays (A=0,C=1,G=2,T=3), haps[0] = base genome
    hap_of: np.ndarray           # haplotype index per read
    starts: np.ndarray           # genome start per read (forward coordinates)
    lens: np.ndarray             # read length
    tails: np.ndarray            # True: the read as sequenced is the reverse complement of the genome interval


def _draw_truth(cfg: SynthConfig, rng: np.random.Generator) -> Truth:
    base = rng.integers(0, 4, size=cfg.genome_len, dtype=np.uint8)
    haps = [base] + [_substitute(base, cfg.snp, rng) for _ in range(cfg.ploidy - 1)]
    n = cfg.n_reads
    hap_of = rng.integers(0, cfg.ploidy, size=n)
    if cfg.len_sd > 0:
        lens = np.clip(np.rint(rng.normal(cfg.read_len, cfg.len_sd, size=n)), cfg.len_min,
                       min(cfg.len_max, cfg.genome_len)).astype(np.int64)
    else:
        lens = np.full(n, min(cfg.read_len, cfg.genome_len), dtype=np.int64)
    starts = (rng.random(n) * (cfg.genome_len - lens + 1)).astype(np.int64)
    tails = rng.random(n) < 0.5
    return Truth(haps, hap_of, starts, lens, tails)


def generate_truth(cfg: SynthConfig) -> Truth:
    """Haplotypes, and the haplotype / start / length / strand of every read ``generate_codes(cfg)`` returns."""
    return _draw_truth(cfg, np.random.default_rng(cfg.seed))


def generate_codes(cfg: SynthConfig) -> Tuple[List[np.ndarray], np.ndarray]:
    """Reads as uint8 code arrays (A=0,C=1,G=2,T=3) in sequencing orientation, plus strand flags."""
    rng = np.random.default_rng(cfg.seed)
    t = _draw_truth(cfg, rng)
    haps, hap_of, starts, lens, tails = t.haps, t.hap_of, t.starts, t.lens, t.tails
    reads: List[np.ndarray] = []
    for i in range(cfg.n_reads):
        r = haps[hap_of[i]][starts[i]:starts[i] + lens[i]]
        if tails[i]:
            r = (3 - r[::-1])
        if cfg.noise > 0:
            r = _substitute(np.ascontiguousarray(r), cfg.noise, rng)
        reads.append(np.ascontiguousarray(r))
    return reads, tails


def codes_to_ascii(codes: np.ndarray) -> bytes:
    return _ASCII[codes].tobytes()


def revcomp_codes(codes: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(3 - codes[::-1])


def generate_reads(cfg: SynthConfig) -> List[Tuple[str, bytes]]:
    """``[(name, sequence)]`` as a FASTA reader would yield them (one strand per read)."""
    reads, _ = generate_codes(cfg)
    return [("read%d" % i, codes_to_ascii(r)) for i, r in enumerate(reads)]


def oriented(reads: List[Tuple[str, bytes]]) -> List[Tuple[str, bytes]]:
    """What ``phasm overlap`` adds to the overlapper: name+ / fwd, name- / revcomp
    (/root/reference/phasm/cli/assembler.py:38-40)."""
    from .io.fasta import reverse_complement
    out = []
    for name, seq in reads:
        out.append((name + "+", seq))
        out.append((name + "-", reverse_complement(seq)))
    return out


def write_fasta(path: str, reads: List[Tuple[str, bytes]], width: Optional[int] = None) -> None:
    with open(path, "wb") as f:
        for name, seq in reads:
            f.write(b">" + name.encode() + b"\n")
            if width:
                for i in range(0, len(seq), width):
                    f.write(seq[i:i + width] + b"\n")
            else:
                f.write(seq + b"\n")


# ----------------------------------------------------------------------------------------------------------------
# The expected output of the exact overlapper, from the generator's truth alone (no overlapper involved).
#
# Contract restated (/root/reference/src/overlapper.cpp:64-116): for oriented reads a != b
#   A  the single longest l >= max(m,1) with a[la-l:] == b[:l]            -> (a, b, la-l, la, 0, l)
#   B  every occurrence p of the whole of b (lb >= max(m,1)) inside a     -> (a, b, p, p+lb, 0, lb)
# A and B are not de-duplicated against each other.  In an i.i.d. uniform genome two reads share >= m >= ~30 equal
# bases only where their genome intervals intersect in the same orientation (a chance match of l bases has
# probability 4^-l per position pair), so with x = [sx, ex) and y = [sy, ey) in FORWARD coordinates the forward
# copies Fx, Fy give
#   A(Fx, Fy)  iff  sx <= sy, ex <= ey, l = ex - sy >= m, and the two reads agree on [sy, ex)
#   B(Fx, Fy)  iff  sx <= sy, ey <= ex, ly >= m,           and the two reads agree on [sy, ey)
# and the reverse copies give exactly the strand mirrors:
#   A (a, b, la-l, la, 0, l)  <->  (b^1, a^1, lb-l, lb, 0, l)       B (a, b, p, p+lb, 0, lb)  <->  (a^1, b^1, la-p-lb, la-p, 0, lb)
# "agree" = no differing position between the two haplotypes on the interval (prefix sums over the haplotype
# difference masks) and, for noisy configs, the noisy reads themselves compared base by base.
# Pinned against the reference's own outputs (tests/golden/: all nine ladder goldens, tests/test_synth_truth.py).
# ----------------------------------------------------------------------------------------------------------------

def _pairs_by_start(starts: np.ndarray, ends: np.ndarray, m: int, max_pairs: int):
    """Yield (x, y) index arrays of all ordered pairs x != y with sx <= sy <= ex - m, in chunks of <= ~max_pairs."""
    order = np.argsort(starts, kind="stable")
    ss = starts[order]
    lo = np.searchsorted(ss, ss, side="left")                       # equal starts: both orders are pairs
    hi = np.searchsorted(ss, ends[order] - m, side="right")
    hi = np.maximum(hi, lo)
    cnt = hi - lo
    cum = np.concatenate([[0], np.cumsum(cnt)])
    n = len(ss)
    i = 0
    while i < n:
        j = int(np.searchsorted(cum, cum[i] + max_pairs, side="right")) - 1
        j = min(max(j, i + 1), n)
        c = cnt[i:j]
        tot = int(c.sum())
        if tot:
            xs = np.repeat(np.arange(i, j), c)
            ys = np.arange(tot) - np.repeat(cum[i:j] - cum[i], c) + np.repeat(lo[i:j], c)
            keep = xs != ys
            yield order[xs[keep]], order[ys[keep]]
        i = j


def expected_rows(cfg: SynthConfig, min_length: int, max_pairs: int = 8_000_000,
                  reads: Optional[List[np.ndarray]] = None) -> np.ndarray:
    """The complete multiset A u B the exact overlapper must return for ``oriented(generate_reads(cfg))``
    (read 2i = read i as sequenced, 2i+1 = its reverse complement), as an (n, 6) int64 array in no particular
    order.  ``reads`` (from ``generate_codes``) is needed only when ``cfg.noise > 0``."""
    m = max(int(min_length), 1)
    assert m >= 24, "below ~24 bases chance matches in a random genome are no longer negligible"
    t = generate_truth(cfg)
    s, ln = t.starts, t.lens
    e = s + ln
    fwd = np.where(t.tails, 2 * np.arange(cfg.n_reads) + 1, 2 * np.arange(cfg.n_reads)).astype(np.int64)   # index of the forward copy
    cums = {}

    def hap_cum(i, j):
        if (i, j) not in cums:
            c = np.zeros(cfg.genome_len + 1, dtype=np.int32)
            np.cumsum(t.haps[i] != t.haps[j], out=c[1:])
            cums[(i, j)] = c
        return cums[(i, j)]

    noisy = None
    if cfg.noise > 0:
        if reads is None:
            reads, _ = generate_codes(cfg)
        noisy = [revcomp_codes(r) if t.tails[i] else r for i, r in enumerate(reads)]     # forward frame
        # one array of all forward reads IN START ORDER: the windows of a chunk of pairs (x ascending by start, y its
        # neighbours) then come from one stretch of it instead of from all over 750 MB
        by_start = np.argsort(s, kind="stable")
        off = np.empty(cfg.n_reads, dtype=np.int64)
        off[by_start] = np.concatenate([[0], np.cumsum(ln[by_start])[:-1]])
        cat = np.concatenate([noisy[i] for i in by_start.tolist()])
        cat8 = np.ndarray(shape=(len(cat) - 7,), dtype=np.uint64, buffer=cat, strides=(1,))    # 8 bases at any offset

    def noisy_equal(x, y, lo, hi, ok):
        """ok[k] &= the noisy reads x[k], y[k] agree on genome interval [lo[k], hi[k]): two vectorised window
        passes (the first 32 and 512 bases) reject nearly everything at 1 % noise, the survivors are compared whole."""
        idx = np.nonzero(ok)[0]
        for w in (4, 64):                                   # windows of 8 bases (one unaligned 64-bit load each)
            for c0 in range(0, len(idx), 1 << 20):
                k = idx[c0:c0 + (1 << 20)]
                j = np.minimum(8 * np.arange(w)[None, :], (hi[k] - lo[k] - 8)[:, None])
                ax = (off[x[k]] + lo[k] - s[x[k]])[:, None] + j
                ay = (off[y[k]] + lo[k] - s[y[k]])[:, None] + j
                ok[k] = (cat8[ax] == cat8[ay]).all(1)
            idx = idx[ok[idx]]
        for k in idx.tolist():
            xi, yi = int(x[k]), int(y[k])
            ok[k] = np.array_equal(noisy[xi][lo[k] - s[xi]:hi[k] - s[xi]], noisy[yi][lo[k] - s[yi]:hi[k] - s[yi]])

    def agree(x, y, lo, hi):
        ok = np.ones(len(x), dtype=bool)
        hx, hy = t.hap_of[x], t.hap_of[y]
        a, b = np.minimum(hx, hy), np.maximum(hx, hy)
        for i in range(cfg.ploidy):
            for j in range(i + 1, cfg.ploidy):
                sel = np.nonzero((a == i) & (b == j))[0]
                if len(sel):
                    c = hap_cum(i, j)
                    ok[sel] = c[hi[sel]] == c[lo[sel]]
        if noisy is not None:
            # (a noise hit can undo a haplotype difference only by hitting that very position with that very base;
            # the reads are compared themselves, so such a pair is at worst missed as a candidate here -- the
            # haplotype test is skipped for noisy configs to be exact)
            ok[:] = True
            noisy_equal(x, y, lo, hi, ok)
        return ok

    out = []
    for x, y in _pairs_by_start(s, e, m, max_pairs):
        lo = s[y]
        hi = np.minimum(e[x], e[y])
        ok = (hi - lo >= m)
        x, y, lo, hi = x[ok], y[ok], lo[ok], hi[ok]
        ok = agree(x, y, lo, hi)
        x, y, lo, hi = x[ok], y[ok], lo[ok], hi[ok]
        lx, ly = ln[x], ln[y]
        fx, fy = fwd[x], fwd[y]
        zero = np.zeros(len(x), dtype=np.int64)
        is_a = e[x] <= e[y]
        is_b = (e[y] <= e[x]) & (ly >= m)
        l = hi - lo
        k = is_a
        out.append(np.stack([fx[k], fy[k], lx[k] - l[k], lx[k], zero[k], l[k]], 1))
        out.append(np.stack([fy[k] ^ 1, fx[k] ^ 1, ly[k] - l[k], ly[k], zero[k], l[k]], 1))
        k = is_b
        p = lo - s[x]
        out.append(np.stack([fx[k], fy[k], p[k], p[k] + ly[k], zero[k], ly[k]], 1))
        out.append(np.stack([fx[k] ^ 1, fy[k] ^ 1, lx[k] - p[k] - ly[k], lx[k] - p[k], zero[k], ly[k]], 1))
    if not out:
        return np.zeros((0, 6), dtype=np.int64)
    return np.concatenate(out).astype(np.int64)


def expected_candidates(cfg: SynthConfig, min_length: int, anchor: int = 32,
                        reads: Optional[List[np.ndarray]] = None, max_pairs: int = 8_000_000) -> dict:
    """The anchors of the seed-extension mode (``po_overlaps_ex``, an extension beyond the reference) from the generator's
    truth: every (a, p, b) over the oriented reads with ``p <= la - m``, ``lb >= m`` and b's first ``anchor`` bases
    equal to ``a[p:p+anchor]`` -- in a random genome that happens only where the two reads' genome intervals say so
    (a chance match of 32 bases: 4^-32 per position pair).  Both strands: the reverse copies are a read set of their own
    on the reverse genome.  Returns oriented indices ``a``, ``b``, positions ``p``, and ``cat`` / ``off`` / ``lens``
    (all oriented reads back to back as code arrays) so that a checker can evaluate the listed pairs."""
    m = max(int(min_length), 1)
    K = min(int(anchor), m)
    assert K >= 24 and K % 8 == 0
    t = generate_truth(cfg)
    if reads is None:
        reads, _ = generate_codes(cfg)
    n = cfg.n_reads
    ori = []
    for r in reads:
        ori.append(r)
        ori.append(revcomp_codes(r))
    lens = np.repeat(t.lens, 2)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    cat = np.concatenate(ori + [np.zeros(8, dtype=np.uint8)])
    del ori
    cat8 = np.ndarray(shape=(len(cat) - 7,), dtype=np.uint64, buffer=cat, strides=(1,))
    fwd = np.where(t.tails, 2 * np.arange(n) + 1, 2 * np.arange(n)).astype(np.int64)
    out_a, out_b, out_p = [], [], []
    for idx, start in ((fwd, t.starts), (fwd ^ 1, cfg.genome_len - (t.starts + t.lens))):
        end = start + t.lens
        for x, y in _pairs_by_start(start, end, m, max_pairs):
            ok = t.lens[y] >= m
            x, y = x[ok], y[ok]
            p = start[y] - start[x]
            ax = off[idx[x]] + p
            ay = off[idx[y]]
            ok = np.ones(len(x), dtype=bool)
            for w in range(0, K, 8):
                ok &= cat8[ax + w] == cat8[ay + w]
            out_a.append(idx[x[ok]])
            out_b.append(idx[y[ok]])
            out_p.append(p[ok])
    return {"a": np.concatenate(out_a), "b": np.concatenate(out_b), "p": np.concatenate(out_p),
            "cat": cat, "off": off, "lens": lens}
