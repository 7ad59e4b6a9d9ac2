"""Size-independent properties of the HIP path on a mid-size workload (100 Mbp index written on the box by
the product's writer cfr_build_index, 1 M x 150 bp reads + pairs + long reads): what must hold whatever the batch looks like -
permutation equivariance, independence of how a batch is cut (host batches, shards, device sub-batches), determinism -
plus agreement with the C oracle on random subsamples.  bench.py covers BASELINE's full size (1 Gbp, 10 M reads) with a
byte comparison against the reference binary on 2 M reads.  Bit-exact everywhere.  -m gpu only."""
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as ora
from centrifuger_amd import capi, shard, synth

pytestmark = pytest.mark.gpu
N_READS = 1_000_000
READ_LEN = 150


def canon(results, matches, k):
    """(n, 5) scalar fields and (n, k, 3) match slots with match_begin resolved (order inside a read is part of the result)."""
    n = len(results)
    sc = np.stack([results["score"].astype(np.int64), results["secondary_score"].astype(np.int64), results["hit_length"].astype(np.int64),
                   results["query_length"].astype(np.int64), results["n_match"].astype(np.int64)], axis=1)
    slots = np.zeros((n, k, 3), dtype=np.int64)
    nm = results["n_match"].astype(np.int64)
    mb = results["match_begin"].astype(np.int64)
    for q in range(k):
        live = nm > q
        src = matches[np.where(live, mb + q, 0)]
        slots[:, q, 0] = np.where(live, src["id"].astype(np.int64), 0)
        slots[:, q, 1] = np.where(live, src["taxid"].astype(np.int64), 0)
        slots[:, q, 2] = np.where(live, src["kind"].astype(np.int64), 0)
    return sc, slots


def digest(sc, slots):
    return hashlib.md5(sc.tobytes() + slots.tobytes()).hexdigest()


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("scale"))
    g = synth.make_genomes(25, 4, 1_000_000, seed=4242)
    prefix = os.path.join(d, "idx")
    capi.build_index(g.names, g.taxids, g.seqs, g.nodes, g.tax_names, prefix)
    reads = synth.make_reads(g, N_READS, READ_LEN, seed=77, sub_rate=0.01, n_rate=0.001)
    r1, r2 = synth.make_pairs(g, 200_000, 125, seed=78)
    longs = synth.make_long_reads(g, 3000, 2000, 12000, seed=79)
    return {"prefix": prefix, "reads": reads, "pairs": (r1, r2), "long": longs, "exact70k": np.ascontiguousarray(g.seqs[0][1000:71000]),
            "exact2m": np.concatenate([np.frombuffer(bytes(g.seqs[0]), dtype=np.uint8), np.frombuffer(bytes(g.seqs[1]), dtype=np.uint8)])}


def _open(prefix, k, env=None):
    old = {}
    for key, val in (env or {}).items():
        old[key] = os.environ.get(key)
        os.environ[key] = val
    try:
        idx = capi.Index(prefix, capi.default_params(max_result=k))
        dev = capi.DeviceIndex(idx)
    finally:
        for key, val in old.items():
            if val is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = val
    return idx, dev


def test_permutation_cut_and_determinism_single_end(world):
    k = 3
    rs = world["reads"]
    idx, dev = _open(world["prefix"], k)
    chk = dev.selfcheck()                   # SA / ISA / text / locate memo of the 100 Mbp image against its BWT, every row
    assert chk["text_tables"] and chk["memo"] == 1 and (chk["bad_sa_isa"], chk["bad_text"], chk["bad_lf"], chk["bad_memo"]) == (0, 0, 0, 0), chk
    res, mat = dev.classify(rs.bases, rs.offsets)
    sc, slots = canon(res, mat, k)
    assert (res["n_match"] > 0).mean() > 0.99
    # determinism: the same call again
    res2, mat2 = dev.classify(rs.bases, rs.offsets)
    assert digest(*canon(res2, mat2, k)) == digest(sc, slots)
    # permutation equivariance
    rng = np.random.default_rng(5)
    perm = rng.permutation(N_READS)
    pb = rs.bases.reshape(N_READS, READ_LEN)[perm].reshape(-1)
    resp, matp = dev.classify(pb, rs.offsets)
    scp, slotsp = canon(resp, matp, k)
    assert np.array_equal(scp, sc[perm]) and np.array_equal(slotsp, slots[perm])
    # how the batch is cut does not matter: 3 unequal host batches, 5 shards (centrifuger_amd.shard), tiny device sub-batches
    cuts = [0, 123_457, 700_001, N_READS]
    parts = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        o = rs.offsets[lo:hi + 1]
        parts.append(dev.classify(rs.bases[int(o[0]):int(o[-1])], (o - o[0]).astype(np.uint64)))
    resm, matm = shard.merge_results(parts)
    assert digest(*canon(resm, matm, k)) == digest(sc, slots)
    parts = [dev.classify(*shard.shard_reads(rs.bases, rs.offsets, 5, r)) for r in range(5)]
    resm, matm = shard.merge_results(parts)
    assert digest(*canon(resm, matm, k)) == digest(sc, slots)
    dev.close()
    idx2, dev2 = _open(world["prefix"], k, {"CFR_SUBBATCH": "33333", "CFR_TAPER_FLOOR": "1000"})
    res3, mat3 = dev2.classify(rs.bases, rs.offsets)
    assert digest(*canon(res3, mat3, k)) == digest(sc, slots)
    dev2.close()
    # without any derived table (plain FM-index walk on the flat image) the same answers again
    idx3, dev3 = _open(world["prefix"], k, {"CFR_FTABX_WIDTH": "0", "CFR_TEXT_MODE": "0", "CFR_LOC_MEMO_GB": "0"})
    res4, mat4 = dev3.classify(rs.bases, rs.offsets)
    assert digest(*canon(res4, mat4, k)) == digest(sc, slots)
    dev3.close()
    # oracle on a random subsample
    o = ora.OracleIndex(world["prefix"], max_result=k)
    pick = np.sort(rng.choice(N_READS, size=4000, replace=False))
    sb = rs.bases.reshape(N_READS, READ_LEN)[pick].reshape(-1)
    so = (np.arange(len(pick) + 1, dtype=np.uint64) * np.uint64(READ_LEN))
    ores = o.classify(sb, so, threads=16)
    for j, i in enumerate(pick):
        assert idx.format_tsv("r", res[i], mat) == o.format("r", ores[j]), int(i)
    o.close()


def test_device_options_entry_gives_the_same_answers(world):
    """cfr_device_index_create_ex: profiles and table switches passed explicitly (no environment)."""
    k = 3
    rs = world["reads"]
    n = 200_000
    b, o = rs.bases[:n * READ_LEN], rs.offsets[:n + 1]
    idx = capi.Index(world["prefix"], capi.default_params(max_result=k))
    dev = capi.DeviceIndex(idx)
    ref = digest(*canon(*dev.classify(b, o), k))
    full_bytes = dev.info().device_bytes
    dev.close()
    for kw in (dict(profile=capi.PROFILE_FAST_LOAD), dict(ftabx_width=0, text_mode=0, loc_memo_gb=0.0), dict(ftabx_width=12, sub_batch=7777),
               dict(run_block_layout=1)):
        d2 = capi.DeviceIndex(idx, 0, capi.default_device_options(**kw))
        assert digest(*canon(*d2.classify(b, o), k)) == ref, kw
        assert d2.info().device_bytes < full_bytes
        d2.close()


def test_pairs_and_long_reads_against_oracle_and_cuts(world):
    k = 5
    idx, dev = _open(world["prefix"], k)
    r1, r2 = world["pairs"]
    n = len(r1.offsets) - 1
    res, mat = dev.classify(r1.bases, r1.offsets, r2.bases, r2.offsets)
    half = n // 2 + 17
    parts = []
    for lo, hi in ((0, half), (half, n)):
        o1, o2 = r1.offsets[lo:hi + 1], r2.offsets[lo:hi + 1]
        parts.append(dev.classify(r1.bases[int(o1[0]):int(o1[-1])], (o1 - o1[0]).astype(np.uint64),
                                  r2.bases[int(o2[0]):int(o2[-1])], (o2 - o2[0]).astype(np.uint64)))
    resm, matm = shard.merge_results(parts)
    assert digest(*canon(resm, matm, k)) == digest(*canon(res, mat, k))
    o = ora.OracleIndex(world["prefix"], max_result=k)
    m = 3000
    ores = o.classify(r1.bases[:int(r1.offsets[m])], r1.offsets[:m + 1], r2.bases[:int(r2.offsets[m])], r2.offsets[:m + 1], threads=16)
    for i in range(m):
        assert idx.format_tsv("p", res[i], mat) == o.format("p", ores[i]), i
    lg = world["long"]
    nl = len(lg.offsets) - 1
    resl, matl = dev.classify(lg.bases, lg.offsets)
    m = 150
    oresl = o.classify(lg.bases[:int(lg.offsets[m])], lg.offsets[:m + 1], threads=16)
    for i in range(m):
        assert idx.format_tsv("l", resl[i], matl) == o.format("l", oresl[i]), i
    # reversing the order of the long reads reverses the results
    order = np.arange(nl)[::-1]
    lens = np.diff(lg.offsets.astype(np.int64))
    rb = np.concatenate([lg.bases[int(lg.offsets[i]):int(lg.offsets[i + 1])] for i in order])
    ro = np.concatenate([[0], np.cumsum(lens[order])]).astype(np.uint64)
    resr, matr = dev.classify(rb, ro)
    scl, sll = canon(resl, matl, k)
    scr, slr = canon(resr, matr, k)
    assert np.array_equal(scr, scl[order]) and np.array_equal(slr, sll[order])
    o.close()
    dev.close()


def test_many_near_identical_strains_general_fold(tmp_path):
    """20 strains per species at 0.1 % steps: ranges of 10-40 rows per hit, so nearly every read takes the general form of
    the per-read fold (hash table in pool scratch, pool growth on the way).  Cross-checked against the sort-based fold of the
    two-kernel path on every read, against the plain FM-index walk, and against the C oracle on a subsample."""
    k = int(os.environ.get("CFR_TEST_K", "2"))     # (tests/test_gpu_variants.py runs this with -k 1 and -k 5 too: LCA by the team, listings)
    g = synth.make_genomes(8, 20, 60_000, seed=991, divergence_step=0.001)
    prefix = str(tmp_path / "idx")
    capi.build_index(g.names, g.taxids, g.seqs, g.nodes, g.tax_names, prefix)
    n = 150_000
    rs = synth.make_reads(g, n, 150, seed=992, sub_rate=0.005, n_rate=0.0005)
    idx, dev = _open(prefix, k, {"CFR_POOL_INIT": "1000"})
    res, mat = dev.classify(rs.bases, rs.offsets)
    ref = digest(*canon(res, mat, k))
    res_b, mat_b = dev.classify(rs.bases, rs.offsets)            # the pool has grown by now: same answers
    assert digest(*canon(res_b, mat_b, k)) == ref
    dev.close()
    for env in ({"CFR_FUSED_POST": "0"}, {"CFR_FTABX_WIDTH": "0", "CFR_TEXT_MODE": "0", "CFR_LOC_MEMO_GB": "0"}):
        idx2, dev2 = _open(prefix, k, env)
        r2, m2 = dev2.classify(rs.bases, rs.offsets)
        assert digest(*canon(r2, m2, k)) == ref, env
        dev2.close()
    o = ora.OracleIndex(prefix, max_result=k)
    m = 2500
    ores = o.classify(rs.bases[:m * 150], rs.offsets[:m + 1], threads=16)
    for i in range(m):
        assert idx.format_tsv("r", res[i], mat) == o.format("r", ores[i]), i
    o.close()
    assert float((res["n_match"] > 0).mean()) > 0.99


@pytest.mark.parametrize("k", [1, 3])
def test_families_of_70_strains_team_fold_limits(tmp_path, k):
    """70 near-identical strains per species: ranges of up to 70 rows.  With -k 1 a hit locates 40 of them by the strided
    enumeration of Classifier.hpp:640-666 (the team kernel's two-pass row sequence); with -k 3 all 70 (more entries than the
    team's table takes: the single-lane form from inside k_tail_heavy, also with the pool too small at first).  Single-end and
    pairs, against the sort-based fold of the two-kernel path on every read and against the C oracle on a subsample."""
    g = synth.make_genomes(3, 70, 30_000, seed=1203, divergence_step=0.0003)
    prefix = str(tmp_path / "idx")
    capi.build_index(g.names, g.taxids, g.seqs, g.nodes, g.tax_names, prefix)
    n = 60_000
    rs = synth.make_reads(g, n, 150, seed=1204, sub_rate=0.004, n_rate=0.0005)
    r1, r2 = synth.make_pairs(g, 20_000, 125, seed=1205)
    idx, dev = _open(prefix, k, {"CFR_POOL_INIT": "1000"})
    res, mat = dev.classify(rs.bases, rs.offsets)
    ref = digest(*canon(res, mat, k))
    res_b, mat_b = dev.classify(rs.bases, rs.offsets)            # the pool has grown by now: same answers
    assert digest(*canon(res_b, mat_b, k)) == ref
    resp, matp = dev.classify(r1.bases, r1.offsets, r2.bases, r2.offsets)
    refp = digest(*canon(resp, matp, k))
    dev.close()
    for env in ({"CFR_FUSED_POST": "0"}, {"CFR_TEAM_TAIL": "0"}):
        idx2, dev2 = _open(prefix, k, env)
        r2_, m2_ = dev2.classify(rs.bases, rs.offsets)
        assert digest(*canon(r2_, m2_, k)) == ref, env
        r3_, m3_ = dev2.classify(r1.bases, r1.offsets, r2.bases, r2.offsets)
        assert digest(*canon(r3_, m3_, k)) == refp, env
        dev2.close()
    o = ora.OracleIndex(prefix, max_result=k)
    m = 1500
    ores = o.classify(rs.bases[:m * 150], rs.offsets[:m + 1], threads=16)
    for i in range(m):
        assert idx.format_tsv("r", res[i], mat) == o.format("r", ores[i]), i
    mp = 500
    oresp = o.classify(r1.bases[:int(r1.offsets[mp])], r1.offsets[:mp + 1], r2.bases[:int(r2.offsets[mp])], r2.offsets[:mp + 1], threads=16)
    for i in range(mp):
        assert idx.format_tsv("r", resp[i], matp) == o.format("r", oresp[i]), i
    o.close()


@pytest.mark.parametrize("k", [1, 5])
def test_resident_entries_wide_and_compact(world, k):
    """cfr_classify_batch_resident (the entry the bench times) and cfr_classify_batch_resident_compact (the same results in the
    20 + 12 byte layout) against the host-buffer entry: single-end and pairs, reads of mixed lengths; a long read whose score
    does not fit 32 bits comes back flagged CFR_COMPACT_WIDE."""
    import torch                               # (device memory for the resident entries: the C-ABI takes device pointers, it does not allocate them)
    idx, dev = _open(world["prefix"], k)
    dv = torch.device("cuda")

    def up(a, dt):
        return torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dv)
    rs = world["reads"]
    n = 300_000
    b = rs.bases[:int(rs.offsets[n])]
    o = rs.offsets[:n + 1]
    want = digest(*canon(*dev.classify(b, o), k))
    db, do = up(b, np.uint8), up(o.astype(np.int64), np.int64)
    torch.cuda.synchronize()
    res, mat = dev.classify_resident(db.data_ptr(), do.data_ptr(), n, int(o[-1]))
    assert digest(*canon(res, mat, k)) == want
    cres, cmat = dev.classify_resident_compact(db.data_ptr(), do.data_ptr(), n, int(o[-1]))
    assert cres.dtype.itemsize == 20 and cmat.dtype.itemsize == 12
    assert digest(*canon(*capi.expand_compact(cres, cmat, k), k)) == want
    r1, r2 = world["pairs"]
    m = len(r1.offsets) - 1
    wantp = digest(*canon(*dev.classify(r1.bases, r1.offsets, r2.bases, r2.offsets), k))
    d1, o1, d2, o2 = up(r1.bases, np.uint8), up(r1.offsets.astype(np.int64), np.int64), up(r2.bases, np.uint8), up(r2.offsets.astype(np.int64), np.int64)
    torch.cuda.synchronize()
    cres, cmat = dev.classify_resident_compact(d1.data_ptr(), o1.data_ptr(), m, int(r1.offsets[-1]), d2.data_ptr(), o2.data_ptr(), int(r2.offsets[-1]))
    assert digest(*canon(*capi.expand_compact(cres, cmat, k), k)) == wantp
    # a 70 kbp exact copy of the text scores (70000 - 15)^2 > 2^32: flagged, and the wide entry has the value
    ex = world["exact70k"]
    L = len(ex)
    dl, ol = up(ex, np.uint8), up(np.array([0, L], dtype=np.int64), np.int64)
    torch.cuda.synchronize()
    cres, cmat = dev.classify_resident_compact(dl.data_ptr(), ol.data_ptr(), 1, L)
    wres, wmat = dev.classify_resident(dl.data_ptr(), ol.data_ptr(), 1, L)
    assert int(wres["score"][0]) == (L - 15) ** 2 and int(wres["score"][0]) >> 32
    assert cres["flags"][0] & capi.COMPACT_WIDE
    with pytest.raises(ValueError):
        capi.expand_compact(cres, cmat, k)
    # ... and the flagged read alone is handed out in the wide layout (cfr_compact_wide_reads): no second pass over a batch.
    # A batch of ordinary reads with the 70 kbp read in the middle: everything equals the wide entry, the one read patched in
    mix_b = np.concatenate([b[:int(o[1000])], ex, b[int(o[1000]):int(o[2000])]])
    mix_o = np.concatenate([o[:1001], [o[1000] + L], o[1001:2001] + L]).astype(np.uint64)
    dm, om = up(mix_b, np.uint8), up(mix_o.astype(np.int64), np.int64)
    torch.cuda.synchronize()
    nm = len(mix_o) - 1
    cres, cmat = dev.classify_resident_compact(dm.data_ptr(), om.data_ptr(), nm, int(mix_o[-1]))
    widx, wres2, wmat2 = dev.compact_wide()
    assert list(widx) == [1000] and int(wres2["score"][0]) == (L - 15) ** 2
    assert int((cres["flags"] & capi.COMPACT_WIDE).sum()) == 1
    full_r, full_m = dev.classify_resident(dm.data_ptr(), om.data_ptr(), nm, int(mix_o[-1]))
    assert digest(*canon(*capi.expand_compact(cres, cmat, k, wide=(widx, wres2, wmat2)), k)) == digest(*canon(full_r, full_m, k))
    # unused match slots of the compact layout are zero
    used = np.arange(k)[None, :] < cres["n_match"][:, None]
    assert not cmat.view(np.uint8).reshape(nm, k, 12)[~used].any()
    dev.close()


def test_ragged_reads_fuzz_against_oracle(world, tmp_path):
    """Reads of every length from 1 to 300 at every alignment of the flat buffer (the 16-byte packed blocks and the
    64-character register queue of the search kernel see every phase), with substitutions, N, lower case, pure noise, mates
    of different lengths; all fields against the C oracle."""
    rng = np.random.default_rng(2024)
    g = synth.make_genomes(6, 3, 40_000, seed=515)
    prefix = str(tmp_path / "fz")
    capi.build_index(g.names, g.taxids, g.seqs, g.nodes, g.tax_names, prefix, ftab_chars=8)
    cat = np.concatenate(g.seqs)
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b

    def make(n):
        chunks, offs = [], [0]
        for _ in range(n):
            L = int(rng.integers(1, 301))
            kind = rng.random()
            if kind < 0.08:
                r = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=L)].copy()      # noise
            else:
                p = int(rng.integers(0, len(cat) - L))
                r = cat[p:p + L].copy()
                if rng.random() < 0.5:
                    r = comp[r[::-1]]
                nsub = rng.binomial(L, 0.02)
                pos = rng.integers(0, L, size=nsub)
                r[pos] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=nsub)]
            if kind > 0.9:
                r[rng.integers(0, L, size=max(1, L // 20))] = ord("N")
            if 0.5 < kind < 0.55:
                r = np.frombuffer(bytes(r).lower(), dtype=np.uint8).copy()                          # lower case is not a symbol
            chunks.append(r)
            offs.append(offs[-1] + L)
        return np.concatenate(chunks), np.array(offs, dtype=np.uint64)

    n = 12_000
    b1, o1 = make(n)
    b2, o2 = make(n)
    for k in (1, 3):
        idx, dev = _open(prefix, k)
        o = ora.OracleIndex(prefix, max_result=k)
        for pe in (False, True):
            res, mat = dev.classify(b1, o1, b2 if pe else None, o2 if pe else None)
            ores = o.classify(b1, o1, b2 if pe else None, o2 if pe else None, threads=16)
            for i in range(n):
                assert idx.format_tsv("r", res[i], mat) == o.format("r", ores[i]), (k, pe, i, int(o1[i + 1] - o1[i]))
            assert 0.5 < float((res["n_match"] > 0).mean()) < 1.0
        o.close()
        dev.close()


def test_very_long_read_among_short_ones(world):
    """One 400 kbp read (two genomes glued, 5 % errors) between ordinary reads: per-chain capacities, the pool of the general
    fold and the 32-bit offsets inside a read all get a large value; fields against the oracle."""
    rng = np.random.default_rng(9)
    rs = world["reads"]
    short_b, short_o = rs.bases[:150 * 50], rs.offsets[:51]
    lg = world["long"]
    big = np.concatenate([lg.bases[int(lg.offsets[i]):int(lg.offsets[i + 1])] for i in range(60)])[:400_000].copy()
    pos = rng.integers(0, len(big), size=len(big) // 20)
    big[pos] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=len(pos))]
    b = np.concatenate([short_b[:150 * 25], big, short_b[150 * 25:]])
    o = np.concatenate([short_o[:26], [short_o[25] + len(big)], short_o[26:] + len(big)]).astype(np.uint64)
    assert int(o[-1]) == len(b) and len(o) == 52
    for k in (1, 5):
        idx, dev = _open(world["prefix"], k)
        res, mat = dev.classify(b, o)
        orc = ora.OracleIndex(world["prefix"], max_result=k)
        ores = orc.classify(b, o, threads=4)
        for i in range(51):
            assert idx.format_tsv("r", res[i], mat) == orc.format("r", ores[i]), (k, i)
        assert res[25]["query_length"] == len(big) and res[25]["n_match"] > 0
        orc.close()
        dev.close()


def test_exact_match_of_two_million_characters(world):
    """A read that IS two neighbouring genomes of the text (2 Mbp, exact; forward and reverse-complemented) among short reads: the search
    finishes on the text with more than 2^20 characters matched since it entered it - what a text-space hit stores in its row field
    (the field had 20 bits; it has 27 now and a search only enters the text when what is left of it fits) - and the located
    positions decide the sequence ids.  TSV lines against the oracle."""
    rs = world["reads"]
    short_b, short_o = rs.bases[:150 * 20], rs.offsets[:21]
    big = world["exact2m"]
    assert len(big) == 2_000_000
    comp = np.zeros(256, dtype=np.uint8)
    for a, c in zip(b"ACGT", b"TGCA"):
        comp[a] = c
    rc = comp[big[::-1]]
    b = np.concatenate([short_b[:150 * 10], big, short_b[150 * 10:], rc])
    o = np.concatenate([short_o[:11], [short_o[10] + len(big)], short_o[11:] + len(big), [short_o[20] + 2 * len(big)]]).astype(np.uint64)
    assert int(o[-1]) == len(b) and len(o) == 23
    for k in (1, 5):
        idx, dev = _open(world["prefix"], k)
        res, mat = dev.classify(b, o)
        orc = ora.OracleIndex(world["prefix"], max_result=k)
        ores = orc.classify(b, o, threads=2)
        for i in range(22):
            assert idx.format_tsv("r", res[i], mat) == orc.format("r", ores[i]), (k, i)
        assert res[10]["hit_length"] >= 1_999_000 and res[21]["hit_length"] >= 1_999_000
        orc.close()
        dev.close()


def test_submitted_batches_equal_the_blocking_call_and_one_call_at_a_time(world):
    """cfr_classify_batch_submit / _wait (the reference overlaps input, classification and output of consecutive batches,
    CentrifugerClass.cpp:776-800): two batches submitted back to back give what the blocking entry gives, a ticket is waited for
    once, and a synchronous entry that arrives while a queued batch is inside the device image gets CFR_ERR_BUSY."""
    import time
    k = 2
    rs = world["reads"]
    idx, dev = _open(world["prefix"], k)
    half = N_READS // 2
    oa = rs.offsets[:half + 1]
    ob = rs.offsets[half:] - rs.offsets[half]
    ba, bb = rs.bases[:int(oa[-1])], rs.bases[int(oa[-1]):]
    want_a, want_b = dev.classify(ba, oa), dev.classify(bb, ob.astype(np.uint64))
    ja = dev.submit(ba, oa)
    jb = dev.submit(bb, ob.astype(np.uint64))
    # the image is busy while the worker runs the batches: a synchronous probe must be refused, not raced
    saw_busy = False
    t0 = time.time()
    while time.time() - t0 < 5.0 and not saw_busy:
        try:
            dev.locate(np.zeros(1, dtype=np.uint64))
        except capi.CfrError as e:
            assert e.status == capi.CFR_ERR_BUSY, e
            saw_busy = True
    got_b, got_a = dev.wait(jb), dev.wait(ja)            # (any order)
    assert saw_busy
    assert digest(*canon(*got_a, k)) == digest(*canon(*want_a, k))
    assert digest(*canon(*got_b, k)) == digest(*canon(*want_b, k))
    with pytest.raises(capi.CfrError) as ei:
        dev.wait(ja)                                     # a ticket is waited for exactly once
    assert ei.value.status == capi.CFR_ERR_ARG
    # more than CFR_MAX_PENDING outstanding batches are refused
    small_o = rs.offsets[:50001]
    small_b = rs.bases[:int(small_o[-1])]
    jobs = []
    with pytest.raises(capi.CfrError) as ei:
        for _ in range(64):
            jobs.append(dev.submit(small_b, small_o))
    assert ei.value.status == capi.CFR_ERR_BUSY and len(jobs) >= 8
    ref_small = None
    for j in jobs:
        r = dev.wait(j)
        ref_small = ref_small or digest(*canon(*r, k))
        assert digest(*canon(*r, k)) == ref_small
    dev.close()


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "centrifuger")),
                    reason="oracle/_ref (compiled reference) not present")
def test_k_above_64_lists_every_strain_like_the_reference(tmp_path):
    """-k 100 over families of 90 near-identical strains: a read lists up to 90 sequences (more slots than any register or team
    table holds: the single-lane form with pool scratch), and the reference has no cap on -k (Classifier.hpp:17-38).  TSV of 3000
    reads and 1000 pairs == the reference binary's; -k 4097 is refused before any device work."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = synth.make_genomes(2, 90, 20_000, seed=3301, divergence_step=0.00002)
    prefix = str(tmp_path / "idx")
    capi.build_index(g.names, g.taxids, g.seqs, g.nodes, g.tax_names, prefix)
    rs = synth.make_reads(g, 3000, 150, seed=3302, sub_rate=0.003, n_rate=0.0005)
    r1, r2 = synth.make_pairs(g, 1000, 125, seed=3303)
    synth.write_fastq(rs, str(tmp_path / "r.fq"))
    synth.write_fastq(r1, str(tmp_path / "p_1.fq"), suffix="/1"); synth.write_fastq(r2, str(tmp_path / "p_2.fq"), suffix="/2")
    ref = os.path.join(root, "oracle", "_ref", "centrifuger")
    cli = os.path.join(root, "centrifuger_amd", "bin", "centrifuger")
    most = 0
    for reads in (["-u", str(tmp_path / "r.fq")], ["-1", str(tmp_path / "p_1.fq"), "-2", str(tmp_path / "p_2.fq")]):
        want = subprocess.run([ref, "-x", prefix, "-t", "8", "-k", "100"] + reads, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        for extra in ([], ["--gpu-throughput"]):
            got = subprocess.run([cli, "-x", prefix, "-t", "4", "-k", "100"] + reads + extra, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
            assert got == want, (reads[0], extra)
        most = max(most, max(int(ln.split(b"\t")[7]) for ln in want.split(b"\n")[1:-1]))
    assert most > 64, most
    with pytest.raises(capi.CfrError):
        capi.DeviceIndex(capi.Index(prefix, capi.default_params(max_result=4097)))
