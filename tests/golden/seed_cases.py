"""Synthetic (window, query) pairs for the seeded-path fixtures (`ref_dump -Q`): a seeded family of genes with
the kinds of damage that send Aln2s1::interpolateS (src/fwd2s1.cc:2405-2539) down its different joins, and a few
hand-made pairs for joins the random family rarely reaches.  Used by make_goldens.py (the committed q_* fixtures)
and by tools/seed_fuzz.py (the same comparison over thousands of seeds, build container only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spaln_amd import synth  # noqa: E402

# seeds of the random family that are committed as fixtures: between them they reach every join at least three times
FIXTURE_SEEDS = [65, 92, 170, 262, 302, 320, 389, 513, 545, 687, 745, 763, 819, 820, 832, 902, 1057, 1287, 1309, 1332,
                 1362, 1386, 1454, 1572, 1576]


def make_case(seed):
    """(window, query, harness options, description)"""
    rng = np.random.default_rng(synth.SEED + 40000 + seed)
    kind = seed % 10
    n_exons = int(rng.integers(2, 9))
    mrna = int(rng.integers(200, 1400))
    sub = float(rng.choice([0.0, 0.02, 0.05, 0.1, 0.15, 0.2]))
    indel = float(rng.choice([0.0, 0.002, 0.01, 0.03]))
    exon_min = int(rng.choice([5, 12, 30]))
    g = synth.make_gene(rng, n_exons=n_exons, mrna_len=mrna, flank=int(rng.integers(100, 1200)),
                        intron_hi=int(rng.choice([300, 1500, 6000])), sub=sub, indel=indel, exon_min=exon_min)
    w, q = g.window, g.query
    opts = ["-Q", str(int(rng.integers(1, 4)))]
    if rng.random() < 0.4:
        opts += ["-X", "0"]
    if rng.random() < 0.1:
        opts += ["-C"]
    desc = f"ex{n_exons} m{mrna} sub{sub} indel{indel} emin{exon_min}"
    if kind == 1:                                   # poly-A tail and junk head on the query
        q = np.concatenate([synth.random_dna(rng, int(rng.integers(3, 40))), q, np.frombuffer(b"A" * int(rng.integers(5, 40)), dtype=np.uint8)])
        desc += " junk+polyA"
    elif kind == 2:                                 # window cut inside the gene: the query overhangs
        e = g.exons
        lo = e[0][0] + int(rng.integers(5, max(6, e[0][1] - e[0][0] - 5))) if rng.random() < 0.7 else 0
        hi = e[-1][1] - int(rng.integers(5, max(6, e[-1][1] - e[-1][0] - 5))) if rng.random() < 0.7 else len(w)
        w = w[lo:hi]
        desc += " cut"
    elif kind == 3:
        opts.append("-L")
        desc += " local"
    elif kind == 4:                                 # a block of the query replaced by noise (HSP desert)
        a0 = int(rng.integers(0, max(1, len(q) - 60)))
        ln = int(rng.integers(20, min(400, len(q) - a0)))
        q = q.copy()
        q[a0:a0 + ln] = synth.random_dna(rng, ln)
        desc += f" noise{ln}"
    elif kind == 5:                                 # an exon missing from the query / duplicated piece
        k = int(rng.integers(0, n_exons))
        lens = [b - a for a, b in g.exons]
        off = sum(lens[:k])
        q = np.concatenate([g.transcript[:off], g.transcript[off + lens[k]:]])
        q = synth.mutate(rng, q, sub, indel)
        desc += f" skip_exon{k}"
    elif kind == 6:                                 # genomic insertion / deletion inside an exon of the window
        k = int(rng.integers(0, n_exons))
        a, b = g.exons[k]
        at = a + (b - a) // 2
        if rng.random() < 0.5:
            w = np.concatenate([w[:at], synth.random_dna(rng, int(rng.integers(1, 30))), w[at:]])
        else:
            w = np.concatenate([w[:at], w[at + int(rng.integers(1, min(30, b - at))):]])
        desc += " exon_indel"
    elif kind == 7:                                 # small MaxVmfSpace: DP calls take the linear-space branch
        opts += ["-V", str(int(rng.choice([20000, 100000, 400000])))]
        desc += " smallV"
    elif kind == 8:                                 # unrelated pair
        q = synth.random_dna(rng, int(rng.integers(60, 400)))
        desc += " random"
    elif kind == 9:                                 # a tandem copy of the locus: several HSP units
        w = np.concatenate([w, w[len(w) // 3:]])
        desc += " tandem"
    return w, q, opts, desc




def make_case_h(seed):
    """protein twin of make_case: (window, protein query, harness options, description)"""
    rng = np.random.default_rng(synth.SEED + 60000 + seed)
    kind = seed % 8
    n_exons = int(rng.integers(2, 7))
    aa = int(rng.integers(80, 420))
    sub = float(rng.choice([0.0, 0.05, 0.15, 0.3, 0.45]))
    g = synth.make_protein_gene(rng, n_exons=n_exons, aa_len=aa, flank=int(rng.integers(100, 900)),
                                sub=sub, intron_hi=int(rng.choice([300, 1200, 4000])))
    w, q = g.window, g.query
    opts = ["-Q", str(int(rng.integers(1, 4)))]
    if rng.random() < 0.4:
        opts += ["-X", "0"]
    desc = f"ex{n_exons} aa{aa} sub{sub}"
    if kind == 1:                                   # junk residues at both ends of the protein
        aas = synth._AA_LETTERS
        q = np.concatenate([aas[rng.integers(0, 20, size=int(rng.integers(2, 25)))], q,
                            aas[rng.integers(0, 20, size=int(rng.integers(2, 25)))]])
        desc += " junk_ends"
    elif kind == 2:                                 # window cut inside the gene
        e = g.exons
        lo = e[0][0] + int(rng.integers(5, max(6, e[0][1] - e[0][0] - 5))) if rng.random() < 0.7 else 0
        hi = e[-1][1] - int(rng.integers(5, max(6, e[-1][1] - e[-1][0] - 5))) if rng.random() < 0.7 else len(w)
        w = w[lo:hi]
        desc += " cut"
    elif kind == 3:                                 # a stretch of the protein replaced by noise
        a0 = int(rng.integers(0, max(1, len(q) - 30)))
        ln = int(rng.integers(8, min(120, len(q) - a0)))
        q = q.copy()
        q[a0:a0 + ln] = synth._AA_LETTERS[rng.integers(0, 20, size=ln)]
        desc += f" noise{ln}"
    elif kind == 4:                                 # residues missing from / inserted into the query
        a0 = int(rng.integers(5, max(6, len(q) - 40)))
        if rng.random() < 0.5:
            q = np.concatenate([q[:a0], q[a0 + int(rng.integers(1, 30)):]])
        else:
            q = np.concatenate([q[:a0], synth._AA_LETTERS[rng.integers(0, 20, size=int(rng.integers(1, 20)))], q[a0:]])
        desc += " query_indel"
    elif kind == 5:                                 # a frame shift in an exon of the window
        k = int(rng.integers(0, n_exons))
        a_, b_ = g.exons[k]
        at = a_ + (b_ - a_) // 2
        w = np.concatenate([w[:at], w[at + int(rng.integers(1, 3)):]]) if rng.random() < 0.5 else \
            np.concatenate([w[:at], synth.random_dna(rng, int(rng.integers(1, 3))), w[at:]])
        desc += " frameshift"
    elif kind == 6:
        opts += ["-V", str(int(rng.choice([60000, 200000, 600000])))]
        desc += " smallV"
    elif kind == 7:                                 # unrelated pair
        q = synth._AA_LETTERS[rng.integers(0, 20, size=int(rng.integers(30, 200)))]
        desc += " random"
    return w, q, opts, desc


def special_cases():
    """name -> (window, query, options)"""
    c = {}
    # the window is the gene exactly, error-free query: the walk closes on an HSP that ends where both sequences end
    rng = np.random.default_rng(synth.SEED + 49001)
    g = synth.make_gene(rng, n_exons=4, mrna_len=500, flank=0, intron_hi=400, sub=0.0, indel=0.0)
    c["q_exact_ends"] = (g.window, g.query, ["-Q", "1"])
    # short terminal exons, error-free: too short for an HSP, found by first_exon / last_exon's exact search
    for k, (qck, crs) in enumerate(((1, 1), (3, 0), (2, 1))):
        rng = np.random.default_rng(synth.SEED + 49010 + k)
        g = synth.make_gene(rng, n_exons=5, mrna_len=600, flank=400, intron_hi=500, sub=0.0, indel=0.0, exon_min=30)
        e = g.exons
        # shrink the first and the last exon to 9-16 nt: window and transcript both lose the outer part
        cut5 = (e[0][1] - e[0][0]) - int(rng.integers(9, 17))
        cut3 = (e[-1][1] - e[-1][0]) - int(rng.integers(9, 17))
        w = np.concatenate([g.window[:e[0][0]], g.window[e[0][0] + cut5:e[-1][1] - cut3], g.window[e[-1][1]:]])
        q = g.transcript[cut5:len(g.transcript) - cut3]
        c[f"q_short_ends{k}"] = (w, q, ["-Q", str(qck), "-X", str(crs)])
    # alignS_ng's default orientation handling with seeding on (ori = 3: both strands walked, the better one kept); "minus"
    # = query and window both reverse-complemented, i.e. the pair matches as given but its introns read CT..AC
    from tests.golden.make_goldens import revcomp
    for k, (qck, sub) in enumerate(((1, 0.03), (3, 0.1), (2, 0.06))):
        rng = np.random.default_rng(synth.SEED + 49030 + k)
        g = synth.make_gene(rng, n_exons=5, mrna_len=700, flank=500, intron_hi=1200, sub=sub, indel=0.01)
        c[f"q_o3_plus{k}"] = (g.window, g.query, ["-Q", str(qck), "-O"])
        c[f"q_o3_minus{k}"] = (revcomp(g.window), revcomp(g.query), ["-Q", str(qck), "-O"])
    # annotated intron positions on the query (ref_dump -I / -J; use_spb()): indelfreespjS adds the position's bonus at a
    # canonical junction (src/fwd2s1.cc:2030-2037), the DP engines add it through Cip_score; positions at the true
    # junctions and displaced by one / two residues, weights from modest to one that outbids the splice signals
    for k, (qck, shift, wgt) in enumerate(((1, 0, 10), (3, 1, 60), (2, -2, 120), (1, 2, 300))):
        rng = np.random.default_rng(synth.SEED + 49050 + k)
        g = synth.make_gene(rng, n_exons=6, mrna_len=900, flank=400, intron_hi=900, sub=0.02, indel=0.0)
        cum = np.cumsum([b - a for a, b in g.exons])[:-1]
        pos = sorted(set(int(x) + shift for x in cum))
        c[f"q_cip{k}"] = (g.window, g.query, ["-Q", str(qck), "-I", ",".join(map(str, pos)), "-J", str(wgt)])
    # BASELINE's headline size under -Q7: the DP calls between HSPs stay far below the 1472 rows at which the
    # reference's int16 engines start re-basing, so its -A2 output IS a witness at 2 kb here
    for k in range(2):
        g = synth.make_gene(np.random.default_rng(synth.SEED + 7000 + k), sub=0.06, indel=0.006)
        c[f"q_c2_seed{k}"] = (g.window, g.query, ["-Q", "3"])
    return c


# the protein walk (seededH_ng / interpolateH): seeds of make_case_h picked with tools/seed_fuzz_h.py so that every join
# the walk serves is reached (diagonal, CDS ends, first / last exon, indel-free junction, micro exon, shortcut with its
# cut range, back-and-forth, small DP without introns, recursion, DP, the three give-ups), all three -Q levels, both
# -X settings; 65 has no HSP at any level and an optimum that hangs on the code behind the query (SpdpProblemH::a_pad)
FIXTURE_SEEDS_H = [116, 45, 13, 82, 85, 89, 123, 0, 3, 5, 6, 14, 24, 32, 38, 65, 98]


def special_cases_h():
    """short terminal coding exons, error-free, same-species mode (-X 0): too short for an HSP, found by the exact
    three-frame search of first_exon / last_exon (BoyerMoore::nexthit3, src/fwd2h1.cc:2803-2834, 2962-3000)"""
    c = {}
    for k, qck in ((3, 3), (5, 2), (12, 1), (13, 3)):          # (picked: both searches succeed, at all three -Q levels)
        rng = np.random.default_rng(synth.SEED + 61010 + k)
        g = synth.make_protein_gene(rng, n_exons=5, aa_len=260, flank=400, sub=0.0, intron_hi=500)
        e = g.exons
        # shrink the first and the last coding exon to 3 - 6 codons (+ up to two bases of a split codon): window and
        # protein both lose the outer part; the stop codon stays where it is
        cut5 = max(0, (e[0][1] - e[0][0]) - int(rng.integers(9, 19))) // 3 * 3
        cut3 = max(0, (e[-1][1] - 3 - e[-1][0]) - int(rng.integers(9, 19))) // 3 * 3
        w = np.concatenate([g.window[:e[0][0]], g.window[e[0][0] + cut5:e[-1][1] - 3 - cut3], g.window[e[-1][1] - 3:]])
        q = g.query[cut5 // 3:len(g.query) - cut3 // 3]
        c[f"qh_short_ends{k}"] = (w, q, ["-Q", str(qck), "-X", "0"])
    return c


def cases_h():
    c = {}
    for s in FIXTURE_SEEDS_H:
        w, q, opts, _ = make_case_h(s)
        c[f"qh_{s:04d}"] = (w, q, opts)
    c.update(special_cases_h())
    c.update({k: v for k, v in a1_cases().items() if k.startswith("qh_")})
    # the same seeds under double affine gaps (-yl3, Noll = 3; -A0: forwardH_ng with and without a cut range, hirschbergH_ng
    # behind the walk; round 5)
    for s in FIXTURE_SEEDS_H:
        w, q, opts, _ = make_case_h(s)
        c[f"qhl3_{s:04d}"] = (w, q, opts + ["-l", "3", "-A", "0"])
    return c


def a1_cases():
    """a few of the random cases once more with the -A1 engines (forwardS1 / hirschbergS1, forwardH1 / hirschbergH1)
    behind the walk as well: ref_dump -A 0,1,2"""
    c = {}
    for s in FIXTURE_SEEDS[:5]:
        w, q, opts, _ = make_case(s)
        c[f"q_a1_{s:04d}"] = (w, q, opts + ["-A", "0,1,2"])
    for s in FIXTURE_SEEDS_H[:5]:
        w, q, opts, _ = make_case_h(s)
        c[f"qh_a1_{s:04d}"] = (w, q, opts + ["-A", "0,1,2"])
    return c


# the same family under double affine gaps (-yl3, Noll = 3; -A0 only: the reference's SIMD engines are not consistent with
# themselves there): seeds that between them make every kind of DP call (lspS_ng, trcbkalignS_ng with and without a cut
# range, the linear-space ladder) with long gaps in the pieces between HSPs
FIXTURE_SEEDS_L3 = [36, 170, 302, 320, 389, 687, 745, 763, 902, 1454]


def cases():
    c = {}
    for s in FIXTURE_SEEDS:
        w, q, opts, _ = make_case(s)
        c[f"q_{s:04d}"] = (w, q, opts)
    for s in FIXTURE_SEEDS_L3:
        w, q, opts, _ = make_case(s)
        c[f"ql3_{s:04d}"] = (w, q, opts + ["-l", "3", "-A", "0"])
    c.update(special_cases())
    c.update({k: v for k, v in a1_cases().items() if k.startswith("q_")})
    return c
