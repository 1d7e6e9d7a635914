"""Deterministic synthetic workloads for the spliced-alignment DP path.

Implements the input recipe of SURVEY.md §8(d) / BASELINE.md §3: an i.i.d.
40 %-GC genome with planted multi-exon genes (log-normal exon lengths, intron
lengths from a heavy-tailed mixture clipped to [60, 20000], ``GTAAGT .. (Y)10 N
CAG`` intron boundaries) and cDNA / EST queries derived from the spliced
transcript with substitutions and short indels.  Nothing here touches the GPU;
the same generator feeds the golden-fixture script (reference run in the build
container), the parity tests and ``bench.py``.
"""
from __future__ import annotations

import dataclasses
import numpy as np

SEED = 20250523
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_dna(rng: np.random.Generator, n: int, gc: float = 0.40) -> np.ndarray:
    """i.i.d. nucleotides as ASCII bytes; P(G)=P(C)=gc/2."""
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    return _ACGT[rng.choice(4, size=n, p=p)]


def _intron_len(rng, lo=60, hi=20000):
    # two-component Frechet-like mixture (short mode ~90 nt, long tail ~1-10 kb)
    if rng.random() < 0.45:
        x = 70 + 25 * (-np.log(rng.random())) ** (-1 / 3.0)
    else:
        x = 200 + 900 * (-np.log(rng.random())) ** (-1 / 1.6)
    return int(min(max(x, lo), hi))


def _intron(rng, length: int) -> np.ndarray:
    s = random_dna(rng, length)
    s[:6] = np.frombuffer(b"GTAAGT", dtype=np.uint8)
    py = np.frombuffer(b"CT", dtype=np.uint8)[rng.integers(0, 2, size=10)]
    s[-14:-4] = py
    s[-3:] = np.frombuffer(b"CAG", dtype=np.uint8)
    return s


@dataclasses.dataclass
class Gene:
    window: np.ndarray          # genomic window, ASCII
    transcript: np.ndarray      # spliced exons, ASCII (error-free)
    exons: list                 # [(start, end)] 0-based half-open in window coordinates
    query: np.ndarray           # mutated transcript (the cDNA / EST), ASCII


def mutate(rng, seq: np.ndarray, sub: float, indel: float) -> np.ndarray:
    out = []
    i = 0
    n = len(seq)
    r = rng.random(n)
    pick = rng.integers(0, 3, size=n)
    while i < n:
        if r[i] < indel / 2:                       # deletion of 1-3 nt
            i += int(rng.integers(1, 4))
            continue
        if r[i] < indel:                           # insertion of 1-3 nt
            out.extend(random_dna(rng, int(rng.integers(1, 4))).tolist())
        c = seq[i]
        if r[i] > 1 - sub:
            others = [x for x in b"ACGT" if x != c]
            c = others[pick[i]]
        out.append(int(c))
        i += 1
    return np.array(out, dtype=np.uint8)


def make_gene(rng, n_exons: int = 8, mrna_len: int = 2000, flank: int = 1000,
              sub: float = 0.02, indel: float = 0.002,
              intron_lo: int = 60, intron_hi: int = 20000,
              exon_min: int = 30) -> Gene:
    """One planted gene with its +-flank window and a mutated cDNA."""
    w = rng.lognormal(mean=0.0, sigma=0.6, size=n_exons)
    lens = np.maximum(exon_min, (w / w.sum() * mrna_len).astype(int))
    lens[-1] = max(exon_min, mrna_len - int(lens[:-1].sum()))
    parts = [random_dna(rng, flank)]
    exons = []
    pos = flank
    tr = []
    for k, L in enumerate(lens):
        ex = random_dna(rng, int(L))
        tr.append(ex)
        parts.append(ex)
        exons.append((pos, pos + int(L)))
        pos += int(L)
        if k + 1 < n_exons:
            il = _intron_len(rng, intron_lo, intron_hi)
            parts.append(_intron(rng, il))
            pos += il
    parts.append(random_dna(rng, flank))
    window = np.concatenate(parts)
    transcript = np.concatenate(tr)
    return Gene(window, transcript, exons, mutate(rng, transcript, sub, indel))


def write_fasta(path: str, name: str, seq: np.ndarray, width: int = 60) -> None:
    with open(path, "w") as fh:
        fh.write(f">{name}\n")
        s = seq.tobytes().decode()
        for i in range(0, len(s), width):
            fh.write(s[i:i + width] + "\n")
