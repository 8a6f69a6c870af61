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


def generate_codes(cfg: SynthConfig) -> Tuple[List[np.ndarray], np.ndarray]:
    """Reads as uint8 code arrays (A=0,C=1,G=2,T=3) in sequencing orientation, plus strand flags."""
    rng = np.random.default_rng(cfg.seed)
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
    reads: List[np.ndarray] = []
    for i in range(n):
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
