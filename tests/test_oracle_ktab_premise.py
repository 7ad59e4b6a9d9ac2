"""The premise of the K-mer COUNT table designed for 40 Gbp (profiles/HISTORY.md section 10), checked with the pinned
oracle on the golden indexes: the BWT ranges of all patterns of one length K, taken in TEXT order, are consecutive - sp of the next
occurring K-mer = ep of the previous one + 1 - except where one of the text's last K - 1 suffixes (shorter than K) sorts in between.
So `sp` of a K-mer = a per-line base + the counts of the K-mers before it in its line, and at most K - 1 lines need the escape."""
import itertools
import os

import pytest

import oracle_lib as ora
from centrifuger_amd import capi


@pytest.mark.parametrize("iname,K", [("f6", 6), ("f6", 7), ("f6_b8", 6), ("f6_off3", 6)])      # (K >= the on-disk ftab width: below it BackwardSearch returns 0, FMIndex.hpp:489-490)
def test_ranges_of_equal_length_patterns_are_consecutive_in_text_order(iname, K, golden_dir):
    o = ora.OracleIndex(os.path.join(golden_dir, iname))
    idx = capi.Index(os.path.join(golden_dir, iname))
    n = idx.info().n
    idx.close()
    prev_end, gaps, rows, first = None, [], 0, None
    for t in itertools.product(b"ACGT", repeat=K):
        l, sp, ep = o.backward_search(bytes(t), K)
        if l == K and sp <= ep:
            rows += ep - sp + 1
            if prev_end is not None and sp != prev_end + 1:
                gaps.append(sp - prev_end - 1)
            first = sp if first is None else first
            prev_end = ep
    # every gap is a handful of short suffixes; together with what stands before the first and behind the last range they are the K - 1 of them
    # (the text of these indexes holds A, C, G, T only, so nothing else can sort between two K-mers)
    assert all(g > 0 for g in gaps)
    assert len(gaps) <= K - 1 and sum(gaps) <= K - 1, gaps
    assert rows + sum(gaps) + first + (n - 1 - prev_end) == n
    assert first + (n - 1 - prev_end) + sum(gaps) == K - 1
    o.close()
