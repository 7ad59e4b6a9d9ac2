"""The N>1 layout (reads shard, index replicated, no data-path collective) under torch.distributed/gloo with
world_size 2 on CPU.  The per-rank classify function is the C oracle here (the product has no CPU path):
what is under test is the sharding, the ordered merge and the timing reduction."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as ora
from centrifuger_amd import capi, shard
from conftest import GOLDEN, ROOT


def _oracle_classify_fn(prefix, k):
    o = ora.OracleIndex(prefix, max_result=k)

    def f(b1, o1, b2, o2):
        res = o.classify(b1, o1, b2, o2)
        n = len(o1) - 1
        results = np.zeros(n, dtype=capi.RESULT_DTYPE)
        matches = []
        for i in range(n):
            r = res[i]
            results[i] = (r.score, r.secondaryScore, r.hitLength, r.queryLength, r.nmatch, 0, len(matches))
            for q in range(r.nmatch):
                matches.append((r.id[q], r.taxid[q], r.kind[q], 0))
        return results, np.array(matches, dtype=capi.MATCH_DTYPE).reshape(-1)
    return f


def _worker(rank, world, port, golden_dir, out_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prefix = os.path.join(golden_dir, "f6")
    ids, b1, o1 = ora.read_fastx(os.path.join(golden_dir, "pe_1.fq"))
    _, b2, o2 = ora.read_fastx(os.path.join(golden_dir, "pe_2.fq"))
    capi.dust_mask(b1, o1); capi.dust_mask(b2, o2)
    results, matches = shard.classify_sharded(_oracle_classify_fn(prefix, 5), b1, o1, b2, o2, dist=dist)
    slow = shard.max_over_ranks(1.0 + rank, dist=dist)
    assert slow == float(world)
    if rank == 0:
        idx = capi.Index(prefix, capi.default_params(max_result=5))
        tsv = capi.tsv_header() + b"".join(idx.format_tsv(ids[i], results[i], matches) for i in range(len(ids)))
        open(out_path, "wb").write(tsv)
    else:
        assert results is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_classification_matches_reference_tsv(golden_dir, tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "merged.tsv")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, golden_dir, out), nprocs=2, join=True)
    assert open(out, "rb").read() == open(os.path.join(GOLDEN, "tsv", "f6.pe_k5.tsv"), "rb").read()


@pytest.mark.parametrize("n,world", [(0, 1), (1, 2), (7, 3), (10, 4), (1000, 8)])
def test_shard_bounds_partition(n, world):
    cuts = [shard.shard_bounds(n, world, r) for r in range(world)]
    assert cuts[0][0] == 0 and cuts[-1][1] == n
    for a, b in zip(cuts, cuts[1:]):
        assert a[1] == b[0]
    sizes = [hi - lo for lo, hi in cuts]
    assert max(sizes) - min(sizes) <= 1


def test_shard_reads_roundtrip():
    rng = np.random.default_rng(3)
    lens = rng.integers(0, 50, size=23)
    offs = np.zeros(24, dtype=np.uint64); offs[1:] = np.cumsum(lens)
    bases = rng.integers(65, 90, size=int(offs[-1])).astype(np.uint8)
    parts = [shard.shard_reads(bases, offs, 4, r) for r in range(4)]
    assert np.array_equal(np.concatenate([p[0] for p in parts]), bases)
    assert sum(len(p[1]) - 1 for p in parts) == 23
    assert all(p[1][0] == 0 for p in parts)
