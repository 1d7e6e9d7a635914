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
    """Substitutions at rate `sub`, 1-3 nt insertions / deletions at rate `indel` (vectorised)."""
    n = len(seq)
    out = seq.copy()
    r = rng.random(n)
    hit = r > 1 - sub
    if hit.any():
        idx = np.nonzero(hit)[0]
        cur = np.searchsorted(_ACGT, out[idx])          # A C G T -> 0..3 (ACGT is sorted)
        out[idx] = _ACGT[(cur + rng.integers(1, 4, size=idx.size)) % 4]
    ev = np.nonzero(r < indel)[0]
    if ev.size == 0:
        return out
    pieces, pos = [], 0
    for i in ev:
        if i < pos:
            continue
        pieces.append(out[pos:i])
        k = int(rng.integers(1, 4))
        if r[i] < indel / 2:
            pos = min(n, i + k)                          # deletion
        else:
            pieces.append(random_dna(rng, k))            # insertion
            pos = i
    pieces.append(out[pos:])
    return np.concatenate(pieces)


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


# ---------------------------------------------------------------------------
# Synthetic splice-signal tables.  In the reference these are Exinon::data_n
# (sig5 / sig3 per genomic position, src/codepot.h:27-54), computed by a PSSM
# scan on the host (SURVEY.md §8f row 1 -- a "next" row, not built yet).  For
# benchmarks without the reference we synthesise tables of the same shape and
# value range (read off the golden fixtures: canonical sites +15..+80, all other
# positions -460..-65, median -350): donor GT at b[n], b[n+1] scores sig5[n],
# acceptor AG at b[n-2], b[n-1] scores sig3[n].
def splice_signals(window_ascii: np.ndarray):
    w = np.asarray(window_ascii, dtype=np.uint8)
    n = w.size
    G, T, A, C = ord("G"), ord("T"), ord("A"), ord("C")
    pad = np.concatenate([np.full(16, ord("N"), np.uint8), w, np.full(16, ord("N"), np.uint8)])

    def at(off):                                   # base at string index i + off, i = 0 .. n
        return pad[16 + off: 16 + off + n + 1]
    h = (np.arange(n + 1, dtype=np.uint32) * np.uint32(2654435761)) ^ (at(0).astype(np.uint32) * np.uint32(40503))
    h ^= at(-1).astype(np.uint32) * np.uint32(97) + at(1).astype(np.uint32) * np.uint32(193)
    noise = ((h >> 7) % 381).astype(np.int32)      # 0 .. 380
    base = -455 + np.minimum(noise, 380 - noise) * 2 + (noise % 7)
    sig5 = base.copy()
    is_gt = (at(0) == G) & (at(1) == T)
    cons5 = (at(2) == A).astype(np.int32) + (at(3) == A) + (at(4) == G) + (at(5) == T) + (at(-1) == G)
    sig5[is_gt] = 15 + 13 * cons5[is_gt]
    sig3 = np.roll(base, 3).copy()
    is_ag = (at(-2) == A) & (at(-1) == G)
    py = np.zeros(n + 1, dtype=np.int32)
    for off in range(-13, -3):
        b = at(off)
        py += ((b == C) | (b == T))
    sig3[is_ag] = 10 + 5 * py[is_ag] + 6 * (at(-3)[is_ag] == C)
    return sig5.astype(np.int16), sig3.astype(np.int16)


def make_batch(n_queries: int, seed: int = SEED, *, n_exons: int = 8, mrna_len: int = 2000,
               flank: int = 1000, sub: float = 0.02, indel: float = 0.002,
               intron_lo: int = 60, intron_hi: int = 20000):
    """C2-style batch: list of (window_codes, query_codes, sig5, sig3, exons)."""
    from . import defaults
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_queries):
        g = make_gene(rng, n_exons=n_exons, mrna_len=mrna_len, flank=flank, sub=sub, indel=indel,
                      intron_lo=intron_lo, intron_hi=intron_hi)
        s5, s3 = splice_signals(g.window)
        out.append((defaults.encode(g.window), defaults.encode(g.query), s5, s3, g.exons))
    return out


def make_est_batch(n_queries: int, seed: int = SEED, *, frag_len: int = 500, margin: int = 1000):
    """C4-style batch: `frag_len`-nt fragments of C2-style transcripts with 1 % error, each against the
    genomic span of its fragment +- `margin`.  Same tuple layout as make_batch (exons = None)."""
    rng = np.random.default_rng(seed + 404)
    out = []
    for w, q, s5, s3, exons in make_batch(n_queries, seed=seed, sub=0.01, indel=0.001):
        a0 = int(rng.integers(0, max(1, len(q) - frag_len)))
        frag = q[a0:a0 + frag_len]
        pos, lo, hi = 0, None, None
        for e0, e1 in exons:
            L = e1 - e0
            if lo is None and a0 < pos + L:
                lo = e0 + (a0 - pos)
            if a0 + frag_len <= pos + L:
                hi = e0 + (a0 + frag_len - pos)
                break
            pos += L
        lo = exons[0][0] if lo is None else lo
        hi = exons[-1][1] if hi is None else hi
        b0, b1 = max(0, lo - margin), min(len(w), hi + margin)
        out.append((w[b0:b1], frag, s5[b0:b1 + 1], s3[b0:b1 + 1], None))
    return out


# ---------------------------------------------------------------------------
# protein x genome (BASELINE config C3): an ORF split into coding exons
_CODON_AA = {}
_BASES = "TCAG"
_AAS = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
for _i, _a in enumerate(_BASES):
    for _j, _b in enumerate(_BASES):
        for _k, _c in enumerate(_BASES):
            _CODON_AA[_a + _b + _c] = _AAS[16 * _i + 4 * _j + _k]
_AA_LETTERS = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)


def translate(dna_ascii: np.ndarray) -> np.ndarray:
    s = dna_ascii.tobytes().decode()
    return np.frombuffer("".join(_CODON_AA[s[i:i + 3]] for i in range(0, len(s) - 2, 3)).encode(), dtype=np.uint8)


@dataclasses.dataclass
class ProteinGene:
    window: np.ndarray          # genomic window, ASCII
    protein: np.ndarray         # error-free translation, ASCII amino acids
    query: np.ndarray           # mutated protein (the query)
    exons: list                 # [(start, end)] of the coding exons in window coordinates


def make_protein_gene(rng, n_exons: int = 4, aa_len: int = 400, flank: int = 500,
                      sub: float = 0.10, intron_lo: int = 60, intron_hi: int = 3000) -> ProteinGene:
    """ATG ... stop ORF of `aa_len` codons (no internal stop), cut into exons at arbitrary phases."""
    orf = random_dna(rng, 3 * aa_len)
    s = bytearray(orf.tobytes())
    for i in range(0, len(s), 3):                       # remove stop codons
        while _CODON_AA[s[i:i + 3].decode()] == "*":
            s[i:i + 3] = random_dna(rng, 3).tobytes()
    s[0:3] = b"ATG"
    orf = np.frombuffer(bytes(s), dtype=np.uint8)
    cds = np.concatenate([orf, np.frombuffer(b"TAA", dtype=np.uint8)])
    w = rng.lognormal(mean=0.0, sigma=0.5, size=n_exons)
    lens = np.maximum(20, (w / w.sum() * len(cds)).astype(int))
    lens[-1] = max(20, len(cds) - int(lens[:-1].sum()))
    parts, exons, pos, off = [random_dna(rng, flank)], [], flank, 0
    for k, L in enumerate(lens):
        L = int(min(L, len(cds) - off))
        if L <= 0:
            break
        parts.append(cds[off:off + L])
        exons.append((pos, pos + L))
        pos += L
        off += L
        if off < len(cds) and k + 1 < n_exons:
            il = _intron_len(rng, intron_lo, intron_hi)
            parts.append(_intron(rng, il))
            pos += il
    parts.append(random_dna(rng, flank))
    prot = translate(orf)
    q = prot.copy()
    hit = rng.random(q.size) < sub
    q[hit] = _AA_LETTERS[rng.integers(0, 20, size=int(hit.sum()))]
    return ProteinGene(np.concatenate(parts), prot, q, exons)


# ---- aa x genome inputs (synthetic stand-ins for Seq::nuc2tron + Exinon::intron53_p) ----------
# tron / amino-acid alphabet of the reference (A = 3 ... V = 22, the AGY serines 23, TGA 24, TAA / TAG 25)
_AA_CODE = {"A": 3, "R": 4, "N": 5, "D": 6, "C": 7, "Q": 8, "E": 9, "G": 10, "H": 11, "I": 12, "L": 13,
            "K": 14, "M": 15, "F": 16, "P": 17, "S": 18, "T": 19, "W": 20, "Y": 21, "V": 22}


def encode_protein(aa_ascii: np.ndarray) -> np.ndarray:
    lut = np.zeros(256, dtype=np.uint8)
    for k, v in _AA_CODE.items():
        lut[ord(k)] = v
    return lut[aa_ascii]


def protein_signals(window: np.ndarray, rng) -> dict:
    """Per-position inputs of the aa x genome DP for a genomic window (ASCII): tron codes
    (b_len + 1 entries) and the SGPT6 fields, index 0 .. b_len + 2.  Position conventions follow
    the reference's (a codon's fields sit at the 0-based index of its middle base; a GT donor at
    the count of bases before it, an AG acceptor at the count of bases through it); the values are
    synthetic: canonical GT / AG sites only, flat coding potential."""
    w = np.asarray(window, dtype=np.uint8)
    L = w.size
    N = L + 3
    base = np.zeros(256, dtype=np.int64)
    for i, c in enumerate(b"TCAG"):
        base[c] = i
    idx = base[w]
    cod = 16 * idx[:-2] + 4 * idx[1:-1] + idx[2:]                 # codon starting at i, order TCAG
    tl = np.zeros(64, dtype=np.uint8)
    bases = "TCAG"
    for c in range(64):
        s3 = bases[c >> 4] + bases[(c >> 2) & 3] + bases[c & 3]
        aa = _CODON_AA[s3]
        tl[c] = (24 if s3 == "TGA" else 25) if aa == "*" else (23 if (aa == "S" and s3[0] == "A") else _AA_CODE[aa])
    tron = np.zeros(L + 1, dtype=np.uint8)
    tron[1:L - 1] = tl[cod]
    sig5 = (-700 + rng.integers(-150, 150, size=N)).astype(np.int16)
    sig3 = (-700 + rng.integers(-150, 150, size=N)).astype(np.int16)
    sigS = np.full(N, -700, dtype=np.int16)
    sigT = np.full(N, -1360, dtype=np.int16)
    sigE = rng.integers(-8, 8, size=N).astype(np.int16)
    stop = np.zeros(N, dtype=bool)
    stop[1:L - 1] = tron[1:L - 1] >= 24
    sigT[stop] = 336
    sigE[stop] = -475
    atg = np.zeros(N, dtype=bool)
    atg[1:L - 1] = cod == (16 * 2 + 4 * 0 + 3)
    sigS[atg] = 650
    gt = np.zeros(N, dtype=bool)
    ag = np.zeros(N, dtype=bool)
    gt[1:L - 1] = (w[1:L - 1] == ord("G")) & (w[2:L] == ord("T"))
    ag[2:L - 1] = (w[0:L - 3] == ord("A")) & (w[1:L - 2] == ord("G"))
    nd, na = int(gt.sum()), int(ag.sum())
    sig5[gt] = rng.integers(-200, 120, size=nd).astype(np.int16)
    sig3[ag] = rng.integers(-200, 150, size=na).astype(np.int16)

    def phases(site):
        # the reference's left-to-right rule (src/codepot.cc:599-606) for canonical sites: a site at n
        # sets phs[n] = 0 (unless already set), phs[n+1] = 1, phs[n-1] = 2 if that was 1 else -1.
        # canonical dinucleotides cannot sit at adjacent positions, so the rule has a closed form
        ph = np.full(N, -2, dtype=np.int8)
        pos = np.nonzero(site)[0]
        ph[pos + 1] = 1
        prev1 = ph[pos - 1] == 1
        ph[pos - 1] = np.where(prev1, 2, -1).astype(np.int8)
        ph[pos] = 0
        return ph

    return dict(b=tron, sig5=sig5, sig3=sig3, sigS=sigS, sigT=sigT, sigE=sigE, phs5=phases(gt), phs3=phases(ag))


def make_protein_batch(n: int, seed: int = SEED, aa_len: int = 400, n_exons: int = 6, flank: int = 1000,
                       intron_hi: int = 5000, sub: float = 0.10):
    """C3-style work: `n` (window, protein) pairs, each a planted `aa_len`-codon ORF +- `flank` nt."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        g = make_protein_gene(rng, n_exons=n_exons, aa_len=aa_len, flank=flank, sub=sub, intron_hi=intron_hi)
        out.append((g, protein_signals(g.window, rng)))
    return out


# ---- exact-model inputs (what the -A0 / -A1 engines read besides the signals) ----------------------------
def exact_inputs(window_codes: np.ndarray) -> dict:
    """cano5 / cano3 / dinc of a genomic window as Exinon::intron53_c assigns them with the default
    `algmode.any = 0` (src/codepot.cc:435-476: donor after AT / GC / GT, acceptor after AC / AG; dinucleotide class
    = the two reduced codes, 'N' counted as C) -- checked against the reference's own arrays in the fixtures
    (tests/test_synth_exact.py).  Returned arrays have len(window) + 1 entries, indexed by position n."""
    b = np.asarray(window_codes, dtype=np.uint8)
    n = b.size
    red = np.full(256, 1, dtype=np.int64)
    for code, c in ((2, 0), (3, 1), (5, 2), (9, 3)):
        red[code] = c
    c = red[b]
    prev = np.concatenate([[1], c[:-1]])                     # the reference starts from the reduced code of 'C'
    nc = ((prev << 2) + c) & 0xf                             # class of the dinucleotide ending at base i
    d5 = np.zeros(n + 2, dtype=np.uint8); d3 = np.zeros(n + 2, dtype=np.uint8)
    d5[:n - 1] = nc[1:]                                      # dinc5[i - 1] = class at base i
    d3[1:n + 1] = nc                                         # dinc3[i + 1] = class at base i
    k5 = np.where(nc == 3, 2, np.where((nc == 9) | (nc == 11), 3, 0)).astype(np.uint8)     # AT, GC, GT
    k3 = np.where(nc == 1, 2, np.where(nc == 2, 3, 0)).astype(np.uint8)                    # AC, AG
    c5 = np.zeros(n + 2, dtype=np.uint8); c3 = np.zeros(n + 2, dtype=np.uint8)
    c5[:n - 1] = k5[1:]
    c3[1:n + 1] = k3
    return dict(cano5=(c5[:n + 1] > 0).astype(np.uint8), cano3=(c3[:n + 1] > 0).astype(np.uint8),
                dinc=((d5[:n + 1] << 4) | d3[:n + 1]).astype(np.uint8))


# ---- chunked, multi-process generation for the large bench batches ---------------------------------------
def _chunk_job(job):
    kind, n, seed, kw = job
    return {"c2": make_batch, "c4": make_est_batch, "c3": make_protein_batch}[kind](n, seed=seed, **kw)


def make_chunked(kind: str, n: int, seed: int = SEED, procs: int = 1, chunk: int = 500, **kw):
    """`n` items of make_batch ("c2") / make_est_batch ("c4") / make_protein_batch ("c3") as chunks of `chunk`, chunk j
    drawn from seed + 7919 j: the batch is a function of (n, seed, chunk) only, however many processes make it."""
    jobs = [(kind, min(chunk, n - j), seed + 7919 * (j // chunk), kw) for j in range(0, n, chunk)]
    if procs <= 1 or len(jobs) == 1:
        parts = [_chunk_job(j) for j in jobs]
    else:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(procs, len(jobs))) as pool:
            parts = pool.map(_chunk_job, jobs, chunksize=1)
    return [x for part in parts for x in part]
