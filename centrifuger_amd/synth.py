"""Seeded synthetic inputs for the parity tests and bench.py.

Nothing here is on the product path: it only manufactures genomes, an NCBI-style
taxonomy (nodes.dmp / names.dmp / seqid map in the formats `centrifuger-build`
accepts, reference Taxonomy.hpp:146-180) and reads.

Genome model (the one the survey probe used): `n_species` species x `n_strains`
strains, strain k = species base with k % i.i.d. substitutions, so that the BWT has
runs and the run-block compressor picks a real block size.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.full(256, ord("N"), dtype=np.uint8)
for _a, _b in zip(b"ACGT", b"TGCA"):
    _COMP[_a] = _b


@dataclass
class Genomes:
    names: list            # sequence names, in index order
    taxids: list           # original tax id of every sequence
    seqs: list             # list of np.uint8 arrays (ASCII ACGT)
    nodes: list            # (taxid, parent, rank)
    tax_names: list        # (taxid, name)

    @property
    def total_len(self) -> int:
        return int(sum(len(s) for s in self.seqs))


def make_genomes(n_species: int, n_strains: int, genome_len: int, seed: int,
                 divergence_step: float = 0.01, species_per_genus: int = 2, model: str = "star") -> Genomes:
    """model "star": strain k = the species base with k x divergence_step i.i.d. substitutions (cfg2's text).  model "tree": strain k
    descends from strain (k - 1) // 2 with divergence_step NEW substitutions per edge - a binary phylogeny, so that close relatives
    are one or two steps apart and the far ends of a 200-strain species 2 log2(200) steps (what a redundant database looks like)."""
    rng = np.random.default_rng(seed)
    names, taxids, seqs = [], [], []
    nodes = [(1, 1, "no rank"), (2, 1, "superkingdom")]
    tax_names = [(1, "root"), (2, "Bacteria")]
    next_tid = 1000
    genus_tid = None
    for sp in range(n_species):
        if sp % species_per_genus == 0:
            genus_tid = next_tid
            next_tid += 1
            nodes.append((genus_tid, 2, "genus"))
            tax_names.append((genus_tid, f"Genus{sp // species_per_genus}"))
        sp_tid = next_tid
        next_tid += 1
        nodes.append((sp_tid, genus_tid, "species"))
        tax_names.append((sp_tid, f"Genus{sp // species_per_genus} species{sp}"))
        base = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
        for k in range(n_strains):
            st_tid = next_tid
            next_tid += 1
            nodes.append((st_tid, sp_tid, "strain"))
            tax_names.append((st_tid, f"species{sp} strain{k}"))
            if model == "tree":
                g = base.copy() if k == 0 else codes[(k - 1) // 2].copy()
                if k > 0:
                    nmut = int(genome_len * divergence_step)
                    pos = rng.integers(0, genome_len, size=nmut)
                    g[pos] = (g[pos] + rng.integers(1, 4, size=nmut, dtype=np.uint8)) & 3
                if k == 0:
                    codes = []
                codes.append(g)
            else:
                g = base.copy()
                if k > 0:
                    nmut = int(genome_len * divergence_step * k)
                    pos = rng.integers(0, genome_len, size=nmut)
                    g[pos] = (g[pos] + rng.integers(1, 4, size=nmut, dtype=np.uint8)) & 3
            names.append(f"SEQ_{sp:04d}_{k}.1")
            taxids.append(st_tid)
            seqs.append(ACGT[g])
    return Genomes(names, taxids, seqs, nodes, tax_names)


def make_genomes_fast(n_species: int, n_strains: int, genome_len: int, seed: int, divergence_step: float = 0.01,
                       species_per_genus: int = 2, threads: int = 32):
    """The same genome model as make_genomes for multi-Gbp texts: one generator per species (seed + species number) so that
    species can be produced by a thread pool, and the sequences land in ONE preallocated buffer.  Returns (Genomes with
    `seqs` = views into that buffer, the buffer itself).  Not bit-compatible with make_genomes (different random streams)."""
    from concurrent.futures import ThreadPoolExecutor
    names, taxids = [], []
    nodes = [(1, 1, "no rank"), (2, 1, "superkingdom")]
    tax_names = [(1, "root"), (2, "Bacteria")]
    next_tid = 1000
    genus_tid = None
    for sp in range(n_species):
        if sp % species_per_genus == 0:
            genus_tid = next_tid
            next_tid += 1
            nodes.append((genus_tid, 2, "genus"))
            tax_names.append((genus_tid, f"Genus{sp // species_per_genus}"))
        sp_tid = next_tid
        next_tid += 1
        nodes.append((sp_tid, genus_tid, "species"))
        tax_names.append((sp_tid, f"Genus{sp // species_per_genus} species{sp}"))
        for k in range(n_strains):
            nodes.append((next_tid, sp_tid, "strain"))
            tax_names.append((next_tid, f"species{sp} strain{k}"))
            names.append(f"SEQ_{sp:04d}_{k}.1")
            taxids.append(next_tid)
            next_tid += 1
    text = np.empty(n_species * n_strains * genome_len, dtype=np.uint8)

    def one(sp):
        rng = np.random.default_rng([seed, sp])
        base = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
        for k in range(n_strains):
            g = base.copy() if k else base
            if k > 0:
                nmut = int(genome_len * divergence_step * k)
                pos = rng.integers(0, genome_len, size=nmut)
                g[pos] = (g[pos] + rng.integers(1, 4, size=nmut, dtype=np.uint8)) & 3
            lo = (sp * n_strains + k) * genome_len
            np.take(ACGT, g, out=text[lo:lo + genome_len])
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(one, range(n_species)))
    seqs = [text[i * genome_len:(i + 1) * genome_len] for i in range(n_species * n_strains)]
    return Genomes(names, taxids, seqs, nodes, tax_names), text


def write_reference_inputs(g: Genomes, outdir: str, line_width: int = 80):
    """ref.fa + nodes.dmp + names.dmp + seqid.map (inputs of centrifuger-build)."""
    os.makedirs(outdir, exist_ok=True)
    fa = os.path.join(outdir, "ref.fa")
    with open(fa, "wb") as f:
        for name, s in zip(g.names, g.seqs):
            f.write(b">" + name.encode() + b"\n")
            n = len(s)
            full = n // line_width * line_width
            if full:
                body = np.empty((full // line_width, line_width + 1), dtype=np.uint8)
                body[:, :line_width] = s[:full].reshape(-1, line_width)
                body[:, line_width] = 10
                f.write(body.tobytes())
            if n > full:
                f.write(s[full:].tobytes() + b"\n")
    with open(os.path.join(outdir, "nodes.dmp"), "w") as f:
        for tid, par, rank in g.nodes:
            f.write(f"{tid}\t|\t{par}\t|\t{rank}\t|\n")
    with open(os.path.join(outdir, "names.dmp"), "w") as f:
        for tid, name in g.tax_names:
            f.write(f"{tid}\t|\t{name}\t|\t\t|\tscientific name\t|\n")
    with open(os.path.join(outdir, "seqid.map"), "w") as f:
        for name, tid in zip(g.names, g.taxids):
            f.write(f"{name}\t{tid}\n")
    return fa


def revcomp(a: np.ndarray) -> np.ndarray:
    return _COMP[a[..., ::-1]]


@dataclass
class ReadSet:
    """Reads as one flat ASCII buffer + offsets (what the C-ABI batch takes)."""
    bases: np.ndarray      # uint8, concatenated reads (no terminators)
    offsets: np.ndarray    # uint64, len n+1

    @property
    def n(self) -> int:
        return len(self.offsets) - 1

    def get(self, i: int) -> bytes:
        return self.bases[int(self.offsets[i]):int(self.offsets[i + 1])].tobytes()

    def slice(self, lo: int, hi: int) -> "ReadSet":
        o = self.offsets[lo:hi + 1]
        return ReadSet(self.bases[int(o[0]):int(o[-1])].copy(), (o - o[0]).astype(np.uint64))


def _mutate(reads: np.ndarray, rng, sub_rate: float, n_rate: float) -> np.ndarray:
    """reads: (n, L) ASCII.  i.i.d. substitutions, then N's."""
    n, L = reads.shape
    if sub_rate > 0:
        m = rng.random((n, L)) < sub_rate
        code = np.zeros(256, dtype=np.uint8)
        code[ACGT] = np.arange(4, dtype=np.uint8)
        c = code[reads[m]]
        reads[m] = ACGT[(c + rng.integers(1, 4, size=c.shape, dtype=np.uint8)) & 3]
    if n_rate > 0:
        reads[rng.random((n, L)) < n_rate] = ord("N")
    return reads


def _concat(g: Genomes):
    cat = np.concatenate(g.seqs)
    starts = np.zeros(len(g.seqs) + 1, dtype=np.int64)
    starts[1:] = np.cumsum([len(s) for s in g.seqs])
    return cat, starts


def make_reads(g: Genomes, n_reads: int, read_len: int, seed: int,
               sub_rate: float = 0.01, n_rate: float = 0.001, chunk: int = 1 << 20, cat=None) -> ReadSet:
    """Fixed-length single-end reads, uniform over genomes / positions / strands.  cat: the genomes back to back, if the
    caller already holds them in one buffer (make_genomes_fast)."""
    rng = np.random.default_rng(seed)
    if cat is None:
        cat, starts = _concat(g)
    else:
        starts = np.zeros(len(g.seqs) + 1, dtype=np.int64)
        starts[1:] = np.cumsum([len(s) for s in g.seqs])
    lens = np.diff(starts)
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    ar = np.arange(read_len, dtype=np.int64)
    for lo in range(0, n_reads, chunk):
        hi = min(n_reads, lo + chunk)
        m = hi - lo
        gi = rng.integers(0, len(lens), size=m)
        pos = (rng.random(m) * (lens[gi] - read_len)).astype(np.int64)
        r = cat[(starts[gi] + pos)[:, None] + ar[None, :]]
        rc = rng.random(m) < 0.5
        r[rc] = revcomp(r[rc])
        out[lo:hi] = _mutate(r, rng, sub_rate, n_rate)
    offs = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len))
    return ReadSet(out.reshape(-1), offs)


def make_pairs(g: Genomes, n_pairs: int, read_len: int, seed: int, ins_lo: int = 250,
               ins_hi: int = 500, sub_rate: float = 0.01, n_rate: float = 0.001):
    """FR-oriented pairs, insert size U[ins_lo, ins_hi]."""
    rng = np.random.default_rng(seed)
    cat, starts = _concat(g)
    lens = np.diff(starts)
    gi = rng.integers(0, len(lens), size=n_pairs)
    ins = rng.integers(ins_lo, ins_hi + 1, size=n_pairs)
    pos = (rng.random(n_pairs) * (lens[gi] - ins)).astype(np.int64)
    ar = np.arange(read_len, dtype=np.int64)
    left = cat[(starts[gi] + pos)[:, None] + ar[None, :]]
    right = revcomp(cat[(starts[gi] + pos + ins - read_len)[:, None] + ar[None, :]])
    flip = rng.random(n_pairs) < 0.5
    r1 = np.where(flip[:, None], right, left)
    r2 = np.where(flip[:, None], left, right)
    r1 = _mutate(np.ascontiguousarray(r1), rng, sub_rate, n_rate)
    r2 = _mutate(np.ascontiguousarray(r2), rng, sub_rate, n_rate)
    offs = (np.arange(n_pairs + 1, dtype=np.uint64) * np.uint64(read_len))
    return ReadSet(r1.reshape(-1), offs), ReadSet(r2.reshape(-1), offs.copy())


def make_long_reads(g: Genomes, n_reads: int, len_lo: int, len_hi: int, seed: int,
                    sub_rate=0.04, ins_rate=0.03, del_rate=0.03) -> ReadSet:
    """ONT-like long reads (variable length, indels)."""
    rng = np.random.default_rng(seed)
    cat, starts = _concat(g)
    lens = np.diff(starts)
    chunks, offs = [], [0]
    for _ in range(n_reads):
        gi = int(rng.integers(0, len(lens)))
        L = int(rng.integers(len_lo, len_hi + 1))
        L = min(L, int(lens[gi]) - 1)
        p = int(rng.integers(0, lens[gi] - L))
        r = cat[starts[gi] + p: starts[gi] + p + L].copy()
        if rng.random() < 0.5:
            r = revcomp(r)
        keep = rng.random(L) >= del_rate
        r = r[keep]
        r = _mutate(r[None, :], rng, sub_rate, 0.0)[0]
        nins = int(len(r) * ins_rate)
        if nins:
            ip = np.sort(rng.integers(0, len(r), size=nins))
            r = np.insert(r, ip, ACGT[rng.integers(0, 4, size=nins)])
        chunks.append(r)
        offs.append(offs[-1] + len(r))
    return ReadSet(np.concatenate(chunks), np.asarray(offs, dtype=np.uint64))


def write_fastq(rs: ReadSet, path: str, prefix: str = "r", suffix: str = ""):
    with open(path, "wb") as f:
        for i in range(rs.n):
            s = rs.get(i)
            f.write(b"@%s%d%s\n%s\n+\n%s\n" % (prefix.encode(), i, suffix.encode(), s, b"I" * len(s)))


def write_fasta(rs: ReadSet, path: str, prefix: str = "r"):
    """Fast path for big fixed-length read sets (vectorised)."""
    n = rs.n
    lens = np.diff(rs.offsets.astype(np.int64))
    if n and (lens == lens[0]).all():
        L = int(lens[0])
        ids = np.char.add(f">{prefix}", np.arange(n).astype(str)).astype("S")
        with open(path, "wb") as f:
            body = rs.bases.reshape(n, L)
            for lo in range(0, n, 1 << 16):
                hi = min(n, lo + (1 << 16))
                f.write(b"".join(i + b"\n" + r.tobytes() + b"\n" for i, r in zip(ids[lo:hi], body[lo:hi])))
    else:
        with open(path, "wb") as f:
            for i in range(n):
                f.write(b">%s%d\n%s\n" % (prefix.encode(), i, rs.get(i)))
