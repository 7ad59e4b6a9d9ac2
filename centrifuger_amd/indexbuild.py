"""A from-scratch writer of Centrifuger `*.cfr` indexes.

Not on the classification path: it exists so that bench.py / tests can manufacture indexes on the GPU box
without the reference's `centrifuger-build`, and so that the on-disk contract is
exercised from the writing side too.  What it must reproduce (all verified in tests/test_indexbuild.py
against indexes written by the real reference):

  * text = concatenation of the genomes (ACGT only, SequenceCompactor.hpp:59-84), no terminator;
  * suffix order: plain lexicographic order of the suffixes, a proper prefix sorts first;
  * BWT[i] = T[SA[i]-1], and T[n-1] at the row with SA == 0 (FMBuilder.hpp:244-250); firstISA = that row;
  * sampled SA every `2^offrate`-th ROW, values turned into sequence ids with the `ftab+1` fuzzy genome
    boundary (Builder.hpp:27-71); selectedSA rows at genome boundaries (Builder.hpp:224-234);
  * ftab: (first row, count) of every w-mer over suffixes of length >= w (FMBuilder.hpp:256-283);
  * run-block compression with the reference's automatic block size (Sequence_RunBlock.hpp:135-177, 231-358),
    wavelet trees, rank9 counters (DS_Rank.hpp:206-247);
  * taxonomy file: compact ids in ascending original tax-id order (Taxonomy.hpp:146-232).

The suffix array is built by prefix doubling with torch sorts (radix sort in HBM when a GPU is present;
the same code runs on CPU for the small test inputs).  The `_space` bookkeeping fields of the reference's
classes are written as 0: every loader ignores them.
"""
from __future__ import annotations

import math
import struct
import time

import numpy as np
import torch

ACGT = b"ACGT"


# ----------------------------------------------------------------------------------------------- suffix array
def suffix_array(codes: torch.Tensor, log=None) -> torch.Tensor:
    """codes: uint8 tensor of 0..3, length n.  Returns int64 SA (prefix doubling, O(n log maxLCP) sorts)."""
    n = codes.numel()
    dev = codes.device
    k = 27                                   # 5^27 < 2^63: base-5 digits, 0 = past the end, 1..4 = A,C,G,T
    key = torch.zeros(n, dtype=torch.int64, device=dev)
    ext = torch.zeros(n + k, dtype=torch.int64, device=dev)
    ext[:n] = codes.to(torch.int64) + 1
    for j in range(k):
        key.mul_(5).add_(ext[j:j + n])
    del ext
    rank = _dense_rank(key)
    del key
    h = k
    rounds = 0
    while int(rank.max().item()) < n - 1:
        nxt = torch.zeros(n, dtype=torch.int64, device=dev)          # rank of suffix i+h, 0 when past the end
        if h < n:
            nxt[:n - h] = rank[h:] + 1
        key = rank * (n + 1) + nxt
        del nxt
        rank = _dense_rank(key)
        del key
        h *= 2
        rounds += 1
        if log:
            log(f"suffix array: doubling round {rounds}, h = {h}")
    sa = torch.empty(n, dtype=torch.int64, device=dev)
    sa[rank] = torch.arange(n, dtype=torch.int64, device=dev)
    return sa


def _dense_rank(key: torch.Tensor) -> torch.Tensor:
    """rank[i] = number of distinct keys smaller than key[i]"""
    sk, order = torch.sort(key)
    new = torch.ones(key.numel(), dtype=torch.int64, device=key.device)
    new[0] = 0
    new[1:] = (sk[1:] != sk[:-1]).to(torch.int64)
    del sk
    dense = torch.cumsum(new, 0)
    del new
    rank = torch.empty_like(dense)
    rank[order] = dense
    return rank


# ----------------------------------------------------------------------------------------------- bit helpers
def _pack_bits(bits: np.ndarray) -> np.ndarray:
    """bool/0-1 array -> little-endian uint64 words (bit i of the vector = bit i%64 of word i//64)"""
    n = len(bits)
    nw = (n + 63) // 64
    b = np.zeros(nw * 64, dtype=np.uint8)
    b[:n] = bits
    return np.packbits(b, bitorder="little").view("<u8")


def _rank9(words: np.ndarray) -> np.ndarray:
    """DS_Rank9::Init (DS_Rank.hpp:206-247): per 8 words an absolute count and 7 nine-bit relative counts."""
    nw = len(words)
    nblk = (nw + 7) // 8
    pc = np.bitwise_count(words).astype(np.uint64)
    padded = np.zeros(nblk * 8, dtype=np.uint64)
    padded[:nw] = pc
    blocks = padded.reshape(nblk, 8)
    tot = blocks.sum(axis=1)
    absc = np.zeros(nblk, dtype=np.uint64)
    absc[1:] = np.cumsum(tot)[:-1]
    within = np.cumsum(blocks, axis=1)                  # within[:, j] = ones in words 0..j of the block
    rel = np.zeros(nblk, dtype=np.uint64)
    for br in range(1, 8):                               # entry br-1 = ones in words 0..br-1
        rel |= within[:, br - 1] << np.uint64(9 * (br - 1))
    # the last block: entries of non-existing words are filled with the block total only when the block has >= 2 words
    last_words = nw - (nblk - 1) * 8
    if last_words < 8:
        v = np.uint64(0)
        for br in range(1, 8):
            if br < last_words:
                v |= np.uint64(within[-1, br - 1]) << np.uint64(9 * (br - 1))
            elif last_words >= 2:
                v |= np.uint64(tot[-1]) << np.uint64(9 * (br - 1))
        rel[-1] = v
    out = np.empty(nblk * 2, dtype="<u8")
    out[0::2] = absc
    out[1::2] = rel
    return out


class _Out:
    def __init__(self, path):
        self.f = open(path, "wb")

    def u64(self, *v):
        self.f.write(struct.pack("<%dQ" % len(v), *v))

    def i32(self, *v):
        self.f.write(struct.pack("<%di" % len(v), *v))

    def raw(self, b):
        self.f.write(b if isinstance(b, (bytes, bytearray)) else np.ascontiguousarray(b).tobytes())

    def close(self):
        self.f.close()


def _write_alphabet(o: _Out, empty=False):
    """Alphabet::Save (Alphabet.hpp:194-205), plain coding of ACGT"""
    if empty:
        o.u64(0); o.i32(0); o.u64(0)
        return
    o.u64(4); o.i32(1); o.u64(4)
    o.raw(ACGT)
    code = np.zeros(256, dtype="<i4")
    clen = np.zeros(256, dtype="<i2")
    for i, ch in enumerate(ACGT):
        code[ch] = i
        clen[ch] = 2
    o.raw(code); o.raw(clen)


def _write_bitvector(o: _Out, bits: np.ndarray):
    """Bitvector_Plain::Save (Bitvector_Plain.hpp:182-196) with select speed 0"""
    n = len(bits)
    o.u64(0); o.u64(n); o.i32(0, 0, 0, 3)
    if n == 0:
        return
    words = _pack_bits(bits)
    o.raw(words)
    r = _rank9(words)
    o.u64(0); o.u64(len(words)); o.raw(r)
    o.u64(0); o.u64(n); o.i32(0)


def _write_wavelet(o: _Out, sym: np.ndarray | None):
    """Sequence_WaveletTree::Save (Sequence_WaveletTree.hpp:303-310); sym = codes 0..3, None = never initialised"""
    if sym is None:
        o.u64(0); o.u64(0); _write_alphabet(o, empty=True); o.i32(0, 3)
        return
    n = len(sym)
    o.u64(0); o.u64(n); _write_alphabet(o); o.i32(3, 0)
    hi = (sym >> 1) & 1
    lo = sym & 1
    nodes = [(0, 0, (1, 2), hi), (0, 1, (-1, -1), lo[hi == 0]), (1, 1, (-1, -1), lo[hi == 1])]
    for prefix, plen, children, bits in nodes:
        o.u64(prefix); o.i32(plen, children[0], children[1])
        _write_bitvector(o, bits)


# ----------------------------------------------------------------------------------------------- run blocks
def _run_block_len(S: np.ndarray, n: int, s: int, e: int, b: int) -> int:
    """Sequence_RunBlock::GetRunBlockLength (Sequence_RunBlock.hpp:26-49)"""
    e = min(e, n - 1)
    if s > e:
        return 0
    starts = np.arange(s, e + 1, b)
    total = 0
    full = starts[starts + b <= n]
    if len(full):
        blk = S[(full[:, None] + np.arange(b)[None, :])]
        total += int((blk == blk[:, :1]).all(axis=1).sum()) * b
    for st in starts[starts + b > n]:                   # a block cut by the end of the sequence
        seg = S[st:n]
        if (seg == seg[0]).all():
            total += n - st
    return total


def _estimate_space(S, n, b, abits, infer_len=1024, cases=1024):
    """EstimateSpace (Sequence_RunBlock.hpp:51-81)"""
    if infer_len * cases >= n:
        rbl = _run_block_len(S, n, 0, n - 1, b)
        m = n
    else:
        rbl = 0
        m = 0
        step = -(-n // cases)
        for i in range(0, n, step):
            e = min(i + infer_len - 1, n - 1)
            rbl += _run_block_len(S, n, i, i + infer_len - 1, b)
            m += e - i + 1
    rbc = -(-rbl // b)
    if b > 1:
        return -(-m // b) + abits * (rbc + m - rbl)
    return abits * m


def _avg_run_length(S, n, infer_len=1024, cases=1024):
    """EstimateAverageRunLength (Sequence_RunBlock.hpp:84-132)"""
    if infer_len * cases >= n:
        r = int((S[1:] != S[:-1]).sum()) + 1
        return n / r
    r = 0
    m = 0
    step = -(-n // cases)
    for i in range(0, n, step):
        e = min(i + infer_len - 1, n - 1)
        seg = S[i:e + 1]
        r += int((seg[1:] != seg[:-1]).sum()) + 1
        m += e - i + 1
    return m / r


def compute_block_size(S: np.ndarray, n: int) -> int:
    """ComputeBlockSize (Sequence_RunBlock.hpp:135-177) for sigma = 4"""
    abits = 2
    best_space, best = 0, 0
    i = 1
    while i <= 1024:
        sp = _estimate_space(S, n, i, abits)
        if best_space == 0 or sp < best_space:
            best_space, best = sp, i
        i *= 2
    if best >= 2:
        sp = _estimate_space(S, n, best // 2 * 3, abits)
        if sp < best_space:
            best_space, best = sp, best // 2 * 3
    x = math.sqrt(_avg_run_length(S, n))
    test = int(x) if int(x) == x else int(x) + 1
    if test > 2:
        sp = _estimate_space(S, n, test, abits)
        if sp < best_space:
            best_space, best = sp, test
    return best


# ----------------------------------------------------------------------------------------------- the builder
def build_index(names, taxids, seqs, nodes, tax_names, out_prefix: str, ftab_chars: int = 10, offrate: int = 4,
                rbbwt_b: int = 0, device=None, log=None):
    """names/taxids/seqs: the sequences in conversion-table order (seqs: np.uint8 ASCII ACGT arrays);
    nodes: (taxid, parent, rank string); tax_names: (taxid, scientific name).  Writes <out_prefix>.{1,2,3,4}.cfr."""
    t_start = time.time()
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    w = ftab_chars
    rate = 1 << offrate
    lut = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(ACGT):
        lut[ch] = i
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    T = lut[np.concatenate(seqs)]
    if (T > 3).any():
        raise ValueError("genomes must be upper-case ACGT only")
    n = len(T)
    psum = np.concatenate([[0], np.cumsum(lens)])
    G = len(seqs)
    Td = torch.from_numpy(T).to(device)
    sa = suffix_array(Td, log)
    if log:
        log(f"suffix array of {n} symbols done in {time.time() - t_start:.1f}s")
    # ---- BWT, firstISA
    prev = torch.where(sa == 0, torch.tensor(n - 1, device=device), sa - 1)
    Bd = Td[prev]
    first_isa = int(torch.nonzero(sa == 0)[0].item())
    last_chr = ACGT[int(T[n - 1])]
    # ---- sampled SA -> sequence ids with the fuzzy boundary (Builder.hpp:27-51)
    psum_d = torch.from_numpy(psum).to(device)
    samp_pos = sa[::rate]
    adj = torch.where(samp_pos + w + 1 < n, samp_pos + w + 1, samp_pos)
    samp_ids = (torch.searchsorted(psum_d, adj, right=True) - 1).cpu().numpy().astype(np.uint64)
    # ---- selectedSA: row of text position psum - w - 1 for every genome boundary (Builder.hpp:224-234)
    sel = {}
    isa_needed = [int(psum[g + 1]) - w - 1 for g in range(G - 1) if int(psum[g + 1]) >= w + 1]
    if isa_needed:
        need = torch.tensor(isa_needed, dtype=torch.int64, device=device)
        # rows of those text positions: invert SA only where needed
        isa = torch.empty(n, dtype=torch.int64, device=device)
        isa[sa] = torch.arange(n, dtype=torch.int64, device=device)
        rows = isa[need].cpu().numpy()
        del isa
        for pos, row in zip(isa_needed, rows):
            sel[int(row)] = int(np.searchsorted(psum, pos + w + 1, side="right") - 1)
    # ---- ftab over suffixes with >= w characters (FMBuilder.hpp:256-283)
    nk = 1 << (2 * w)
    ftab = np.zeros((nk, 2), dtype="<u8")
    if n >= w:
        key = torch.zeros(n - w + 1, dtype=torch.int64, device=device)
        Tl = Td.to(torch.int64)
        for j in range(w):                                  # PackRead: symbol j of the w-mer sits at bits 2j (first symbol least significant)
            key.add_(Tl[j:n - w + 1 + j] << (2 * j))
        valid = sa + w <= n
        rows_valid = torch.nonzero(valid).squeeze(1)
        keys_by_row = key[sa[rows_valid]]
        cnt = torch.bincount(keys_by_row, minlength=nk)
        # rows are in suffix order, so equal w-mers are contiguous among the valid rows (not sorted by this key packing)
        flag = torch.ones(keys_by_row.numel(), dtype=torch.bool, device=device)
        flag[1:] = keys_by_row[1:] != keys_by_row[:-1]
        first_idx = torch.nonzero(flag).squeeze(1)
        first = torch.zeros(nk, dtype=torch.int64, device=device)
        first[keys_by_row[first_idx]] = rows_valid[first_idx]
        ftab[:, 0] = first.cpu().numpy()
        ftab[:, 1] = cnt.cpu().numpy()
        del key, Tl, valid, rows_valid, keys_by_row
    B = Bd.cpu().numpy()
    del sa, prev, Bd
    # ---- C[] (exclusive prefix sums of the symbol counts)
    counts = np.bincount(B, minlength=4).astype(np.uint64)
    C = np.zeros(5, dtype="<u8")
    C[1:] = np.cumsum(counts)
    # ---- run blocks (Sequence_RunBlock::Init, Sequence_RunBlock.hpp:231-358)
    b = rbbwt_b if rbbwt_b else compute_block_size(B, n)
    if b == 1:
        b = n
    nblk = -(-n // b)
    pad = nblk * b - n
    Bp = np.concatenate([B, np.full(pad, B[-1], dtype=np.uint8)]) if pad else B
    blocks = Bp.reshape(nblk, b)
    is_run = (blocks == blocks[:, :1]).all(axis=1)
    if pad:                                                # the last block is judged on its real symbols only
        lastseg = B[(nblk - 1) * b:]
        is_run[-1] = bool((lastseg == lastseg[0]).all())
    keep = np.repeat(~is_run, b)[:n]
    plain = B[keep]
    runs = blocks[is_run, 0]
    if log:
        log(f"run-block: b = {b}, {int(is_run.sum())} run blocks of {nblk}; wavelet part {len(plain)}, run part {len(runs)}")

    # ---- .1.cfr
    o = _Out(out_prefix + ".1.cfr")
    o.u64(n, 2, first_isa); o.raw(bytes([last_chr]))
    o.u64(0, n); _write_alphabet(o)
    o.u64(b, nblk)
    _write_bitvector(o, is_run.astype(np.uint8))
    _write_wavelet(o, plain if len(plain) else None)
    _write_wavelet(o, runs if len(runs) else None)
    _write_alphabet(o); _write_alphabet(o)
    o.raw(C)
    nsamp = -(-n // rate)
    o.u64(n); o.i32(0, rate); o.u64(nsamp, w, nk, 0)      # n, sampleStrategy, sampleRate, sampleSize, width, size, adjustedSA0 (= seq id 0)
    bits = max(1, int(samp_ids.max()).bit_length()) if len(samp_ids) else 1
    words = _pack_fixed(samp_ids, bits)
    o.u64(len(words)); o.i32(bits); o.u64(len(samp_ids)); o.raw(words)
    o.raw(ftab)
    o.u64(0)                                               # maxLcp
    o.u64(len(sel)); o.i32(1024)
    for row in sorted(sel):
        o.u64(row, sel[row])
    o.raw(b"\x00")                                         # hasEndMarker = false
    o.close()
    # ---- .2.cfr / .3.cfr / .4.cfr
    write_taxonomy(out_prefix + ".2.cfr", names, taxids, nodes, tax_names)
    with open(out_prefix + ".3.cfr", "wb") as f:
        for i, L in enumerate(lens):
            f.write(struct.pack("<2Q", i, int(L)))
    with open(out_prefix + ".4.cfr", "w") as f:
        f.write(f"version\t1.1.3-r347\nSA_sample_rate\t{rate}\nsequence_type\tnucleotide\nbuild_date\t{time.strftime('%c')}")
    if log:
        log(f"index written to {out_prefix}.*.cfr in {time.time() - t_start:.1f}s")
    return {"n": n, "b": b, "first_isa": first_isa}


def _pack_fixed(vals: np.ndarray, bits: int) -> np.ndarray:
    """FixedSizeElemArray layout: element i at bits [i*l, (i+1)*l), LSB first"""
    n = len(vals)
    nw = (n * bits + 63) // 64
    out = np.zeros(nw + 1, dtype=np.uint64)
    pos = np.arange(n, dtype=np.uint64) * np.uint64(bits)
    wi = (pos >> np.uint64(6)).astype(np.int64)
    sh = pos & np.uint64(63)
    v = vals.astype(np.uint64)
    np.bitwise_or.at(out, wi, v << sh)
    spill = (sh + np.uint64(bits)) > np.uint64(64)
    if spill.any():
        np.bitwise_or.at(out, wi[spill] + 1, v[spill] >> (np.uint64(64) - sh[spill]))
    return out[:nw].astype("<u8")


_RANKS = ["no rank", "strain", "species", "genus", "family", "order", "class", "phylum", "kingdom", "domain", "forma",
          "infraclass", "infraorder", "parvorder", "subclass", "subfamily", "subgenus", "subkingdom", "suborder",
          "subphylum", "subspecies", "subtribe", "superclass", "superfamily", "superkingdom", "superorder", "superphylum",
          "tribe", "varietas", "life", "acellular root"]


def write_taxonomy(path, names, taxids, nodes, tax_names):
    """Taxonomy::Init + Save (Taxonomy.hpp:146-232, 1238-1257)"""
    tree = {}
    for tid, parent, rank in nodes:
        if tid not in tree:
            tree[tid] = (parent, _RANKS.index(rank) if rank in _RANKS else 0)
    selected = set()
    for tid in sorted(set(taxids)):
        p = tid
        if p not in tree:
            continue
        while p not in selected:
            selected.add(p)
            p = tree[p][0]
    order = sorted(selected)                              # compact ids follow ascending original id (std::map order)
    cid = {tid: i for i, tid in enumerate(order)}
    leaf = {tid: 1 for tid in order}
    parent_c = {}
    for tid in order:
        par = tree[tid][0]
        if par in cid:
            parent_c[tid] = cid[par]
            leaf[par] = 0
        else:
            parent_c[tid] = cid[tid]
    sci = {}
    for tid, name in tax_names:
        if tid in cid:
            sci[tid] = "_".join(name.split())
    with open(path, "wb") as f:
        f.write(struct.pack("<3Q", len(order), len(names), 0))
        for tid in order:
            f.write(struct.pack("<QBB6x", parent_c[tid], tree[tid][1], leaf[tid]))
        f.write(struct.pack("<Q", len(order)))
        f.write(struct.pack("<%dQ" % len(order), *order))
        for tid in order:
            s = sci.get(tid, "").encode()
            f.write(struct.pack("<Q", len(s)) + s)
        f.write(struct.pack("<%dQ" % len(names), *[cid[t] for t in taxids]))
        for nm in names:
            s = nm.encode()
            f.write(struct.pack("<Q", len(s)) + s)


# ----------------------------------------------------------------------------------------------- command line
def read_fasta(paths):
    """[(name, np.uint8 ACGT-only sequence)]: everything that is not an upper-case A,C,G,T is dropped, like
    SequenceCompactor::Compact (SequenceCompactor.hpp:59-84; no capitalisation, no replacement)."""
    import gzip
    keep = np.zeros(256, dtype=bool)
    for ch in ACGT:
        keep[ch] = True
    out = []
    for p in paths:
        opener = gzip.open if p.endswith(".gz") else open
        with opener(p, "rb") as f:
            data = f.read()
        for rec in data.split(b">")[1:]:
            head, _, body = rec.partition(b"\n")
            arr = np.frombuffer(body, dtype=np.uint8)
            out.append((head.split()[0].decode(), arr[keep[arr]].copy()))
    return out


def read_taxonomy_files(nodes_path, names_path, map_path):
    nodes, names, seqmap = [], [], []
    for line in open(nodes_path):
        if not line.strip() or line.startswith("#"):
            continue
        f = [x.strip() for x in line.split("|")]
        nodes.append((int(f[0]), int(f[1]), f[2]))
    for line in open(names_path):
        if "scientific name" not in line:
            continue
        f = [x.strip() for x in line.split("|")]
        names.append((int(f[0]), f[1]))
    for line in open(map_path):
        if not line.strip() or line.startswith("#"):
            continue
        a, b = line.split()[:2]
        seqmap.append((a, int(b)))
    return nodes, names, seqmap


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog="python -m centrifuger_amd.indexbuild",
                                 description="write a Centrifuger .cfr index (subset of centrifuger-build's options)")
    ap.add_argument("-r", action="append", required=True, help="reference FASTA (repeatable)")
    ap.add_argument("--taxonomy-tree", required=True)
    ap.add_argument("--name-table", required=True)
    ap.add_argument("--conversion-table", required=True)
    ap.add_argument("-o", default="centrifuger")
    ap.add_argument("--ftabchars", type=int, default=10)
    ap.add_argument("--offrate", type=int, default=4)
    ap.add_argument("--rbbwt-b", type=int, default=0)
    a = ap.parse_args(argv)
    nodes, names, seqmap = read_taxonomy_files(a.taxonomy_tree, a.name_table, a.conversion_table)
    seqs = dict(read_fasta(a.r))
    order = [(nm, tid) for nm, tid in seqmap if nm in seqs and len(seqs[nm]) >= a.ftabchars + 1]
    if not order:
        raise SystemExit("no sequence of the conversion table found in the FASTA input")
    if [nm for nm, _ in order] != [nm for nm, _ in seqmap]:
        raise SystemExit("every sequence of the conversion table must be present (and longer than ftabchars) in this writer")
    build_index([nm for nm, _ in order], [t for _, t in order], [seqs[nm] for nm, _ in order], nodes, names, a.o,
                ftab_chars=a.ftabchars, offrate=a.offrate, rbbwt_b=a.rbbwt_b, log=lambda m: print("[indexbuild]", m))


if __name__ == "__main__":
    main()
