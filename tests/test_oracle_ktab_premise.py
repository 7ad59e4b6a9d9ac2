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


def _build_lines(o, KT, min_hit_len, half=24, miss=200, poison=255):
    """k_build_ktab restated (cfr_kernels.hip.inc): per half of 24 consecutive text-order KT-mers the sp of the first occurring one and a byte
    each - count, `miss` + (K - stop) for one that does not occur and stops after `stop` characters (< min_hit_len), `poison` from the first
    K-mer on that cannot be described"""
    K = KT - 1
    keys = [bytes(t) for t in itertools.product(b"ACGT", repeat=KT)]
    halves = []
    for h0 in range(0, len(keys), half):
        base, expect, any_, poisoned, ent = 0, None, False, False, []
        for P in keys[h0:h0 + half]:
            if poisoned:
                ent.append(poison)
                continue
            l, sp, ep = o.backward_search(P, KT)
            if l == KT and sp <= ep:
                cnt = ep - sp + 1
                if cnt >= miss or (any_ and sp != expect):
                    poisoned = True
                    ent.append(poison)
                    continue
                if not any_:
                    base, any_ = sp, True
                expect = ep + 1
                ent.append(cnt)
            else:
                ent.append(miss + (K - l) if l < min_hit_len and 0 <= K - l < 32 else 0)
        halves.append((base, ent))
    return keys, halves


@pytest.mark.parametrize("iname,KT", [("f6", 7), ("f6_b8", 8)])
def test_count_table_lines_answer_like_backward_search(iname, KT, golden_dir):
    """the table's lookup - base + the counts before the key in its half, or the stop length - equals BackwardSearch for every KT-mer it
    answers, and it answers nearly all of them (what it gives up: the halves behind one of the text's last KT - 1 suffixes)"""
    o = ora.OracleIndex(os.path.join(golden_dir, iname))
    keys, halves = _build_lines(o, KT, min_hit_len=23)
    answered = 0
    for i, P in enumerate(keys):
        base, ent = halves[i // 24]
        q = i % 24
        e = ent[q]
        l, sp, ep = o.backward_search(P, KT)
        if 1 <= e < 200:
            below = sum(x for x in ent[:q] if x < 200)
            assert (l, sp, ep) == (KT, base + below, base + below + e - 1), P
            answered += 1
        elif 200 <= e < 232:
            assert l == KT - 1 - (e - 200) and not (l == KT and sp <= ep), P
            answered += 1
    assert answered >= len(keys) - 24 * (KT - 1)
    o.close()
