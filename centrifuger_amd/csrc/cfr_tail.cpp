// cfr_tail.cpp — host tail of Query: from located hits to a classification.
//
// Restates Classifier::GetClassificationFromHits (Classifier.hpp:585-843) from the point where the
// sequence ids of every hit are known (the device has already done the FM-index work), plus the
// taxonomy side-table logic it calls: Taxonomy::SeqIdToTaxId / GetOrigTaxId / LCA / ReduceTaxIds
// (Taxonomy.hpp:718-724, 633-639, 733-836, 839-973).  Pure host integer work, threaded over reads.
#include "cfr_tail.hpp"

#include <algorithm>
#include <thread>

namespace cfr {

namespace {

struct SeqRecord { uint64_t seq_id, score; int32_t hit_length; };

// ascending-key small map (the reference iterates std::map in key order, which fixes the
// output order of equal-score sequences: Classifier.hpp:745-757)
struct RecordMap {
  std::vector<SeqRecord> v;
  SeqRecord &at(uint64_t key, bool *created) {
    auto it = std::lower_bound(v.begin(), v.end(), key, [](const SeqRecord &r, uint64_t k) { return r.seq_id < k; });
    if (it != v.end() && it->seq_id == key) { if (created) *created = false; return *it; }
    if (created) *created = true;
    return *v.insert(it, SeqRecord{key, 0, 0});
  }
};

inline uint64_t score_of(const HostIndex &h, int32_t l) {
  if (l < h.params.min_hit_len) return 0;
  const uint64_t d = (uint64_t)(int64_t)(l - h.score_hit_len_adjust);
  return d * d;
}

inline uint64_t seq_to_tax(const Taxonomy &t, uint64_t seq_id) { return seq_id < t.seq_cnt ? t.seq_to_tax[seq_id] : t.node_cnt; }
inline uint64_t orig_taxid(const Taxonomy &t, uint64_t ctid) { return t.orig_taxid[ctid >= t.node_cnt ? t.root : ctid]; }

// path from a node to (and including) the root
void lineage(const Taxonomy &t, uint64_t x, std::vector<uint64_t> &path) {
  path.clear();
  do { path.push_back(x); x = t.parent[x]; } while (x != t.parent[x]);
  path.push_back(t.root);
}

}  // namespace

// Taxonomy::LCA (Taxonomy.hpp:733-836); the child bookkeeping is tax_lca_children below
uint64_t tax_lca(const Taxonomy &t, const std::vector<uint64_t> &ids) {
  const int cnt = (int)ids.size();
  int k = 0;
  while (k < cnt && ids[k] == t.root) ++k;
  if (k == cnt) return t.root;
  std::vector<uint64_t> backbone, other;
  lineage(t, ids[k], backbone);
  std::vector<int> shared(backbone.size(), 1);
  int root_count = 0;
  for (int i = 0; i < cnt; ++i) {
    if (i == k) continue;
    if (ids[i] == t.parent[ids[i]]) { ++root_count; continue; }
    lineage(t, ids[i], other);
    int ib = (int)backbone.size() - 1, io = (int)other.size() - 1;
    for (; ib >= 0 && io >= 0; --ib, --io) {
      if (other[io] != backbone[ib]) break;
      shared[ib] += 1;
    }
  }
  for (size_t j = 0; j < backbone.size(); ++j)
    if (shared[j] == cnt - root_count) return backbone[j];
  return t.root;
}

// lcaChildTaxIds of Taxonomy::LCA (Taxonomy.hpp:771-777, 806-813, 822-830) for its result `lca`: the reference keeps one ordered set per
// node of the first id's lineage - the lineage's own node below it, and for every other id the node at which its path leaves the
// lineage - and hands back the set of the LCA's place.  An id whose path leaves further down shares the lineage's node below the
// LCA, the LCA itself and root ids leave nothing: so the set is { child of lca on the way down to x : x in ids, x not a root, x != lca }.
void tax_lca_children(const Taxonomy &t, const std::vector<uint64_t> &ids, uint64_t lca, std::vector<uint64_t> &children) {
  children.clear();
  for (uint64_t x : ids) {
    if (x == t.parent[x] || x == lca) continue;
    for (;;) {
      const uint64_t p = t.parent[x];
      if (p == lca || p == t.parent[p]) break;
      x = p;
    }
    children.push_back(x);
  }
  std::sort(children.begin(), children.end());              // std::map order
  children.erase(std::unique(children.begin(), children.end()), children.end());
}

// Taxonomy::ReduceTaxIds (Taxonomy.hpp:839-973).  children (promotedChildTaxIds) may be null; otherwise it comes back with one list
// per entry of `out` - or with none, where the reference pushes none (Classifier.hpp:823 prints empty strings then).
void tax_reduce(const Taxonomy &t, const std::vector<uint64_t> &ids, int k, std::vector<uint64_t> &out,
                std::vector<std::vector<uint64_t>> *children) {
  out.clear();
  if (children) children->clear();
  if ((int)ids.size() <= k) { out = ids; return; }
  for (uint64_t x : ids)
    if (x >= t.node_cnt) {
      out.push_back(t.node_cnt);
      if (children) children->push_back(ids);              // :866-872: every input id, as it came
      return;
    }
  if (k == 1) {
    out.push_back(tax_lca(t, ids));
    if (children) { children->emplace_back(); tax_lca_children(t, ids, out[0], children->back()); }
    return;
  }
  const uint8_t unknown_level = t.rank_num[0];
  std::vector<std::vector<uint64_t>> level(32);      // sorted-unique id sets per rank level
  auto insert = [](std::vector<uint64_t> &s, uint64_t x) {
    auto it = std::lower_bound(s.begin(), s.end(), x);
    if (it != s.end() && *it == x) return false;
    s.insert(it, x);
    return true;
  };
  for (uint64_t start : ids) {
    uint64_t x = start;
    uint8_t prev = 0;
    insert(level[prev], x);
    do {
      const uint8_t lv = t.rank_num[t.rank[x]];
      if (lv != unknown_level && lv > prev) {
        for (uint8_t r = lv - 1; r > prev; --r) insert(level[r], x);
        if (!insert(level[lv], x)) break;
        prev = lv;
      }
      x = t.parent[x];
    } while (x != t.parent[x]);
  }
  uint8_t r = 0;
  for (; r < unknown_level; ++r) if ((int)level[r].size() <= k) break;
  out = level[r];
  if (out.empty()) out.push_back(t.root);
  else if (children && r > 0) {
    // :939-971: a member of the set one level down belongs to the first node above it that is filed at exactly this level - if a
    // higher-ranked (or unranked: the largest number) node comes first, to nobody
    children->assign(out.size(), {});
    for (uint64_t member : level[r - 1]) {
      uint64_t x = member;
      while (x != t.parent[x]) {
        x = t.parent[x];
        const uint8_t lv = t.rank_num[t.rank[x]];
        if (lv > r) break;
        if (lv == r) {
          auto it = std::lower_bound(out.begin(), out.end(), x);
          if (it != out.end() && *it == x) (*children)[(size_t)(it - out.begin())].push_back(member);
          break;
        }
      }
    }
  }
}

void classify_read(const HostIndex &h, const cfr_hit *hits, size_t nhits, const uint64_t *row_begin, const uint64_t *row_vals,
                   int32_t query_len, cfr_result &res, std::vector<cfr_match> &matches, ExpandedLists *expanded) {
  const cfr_params &P = h.params;
  RecordMap rec[2];
  SeqRecord prev_uniq{0, 0, 0};
  bool mix_strand = false;
  for (size_t i = 1; i < nhits; ++i) if (hits[i].strand != hits[i - 1].strand) { mix_strand = true; break; }

  std::vector<uint64_t> local;
  for (size_t i = 0; i < nhits; ++i) {
    if (hits[i].l < P.min_hit_len) continue;
    const uint64_t score = score_of(h, hits[i].l);
    const int k = (hits[i].strand + 1) / 2;
    local.assign(row_vals + row_begin[i], row_vals + row_begin[i + 1]);
    std::sort(local.begin(), local.end());
    local.erase(std::unique(local.begin(), local.end()), local.end());
    for (uint64_t seq_id : local) {
      const bool merge = !mix_strand && i > 0 && hits[i].ep == hits[i].sp && hits[i - 1].ep == hits[i - 1].sp &&
                         hits[i - 1].strand == hits[i].strand && hits[i - 1].offset + hits[i - 1].l + 1 == hits[i].offset &&
                         seq_id == prev_uniq.seq_id;
      if (merge) {   // adjacent unique hits separated by one base (Classifier.hpp:673-685)
        SeqRecord &r = rec[k].at(seq_id, nullptr);
        r.score -= prev_uniq.score;
        prev_uniq.hit_length += hits[i].l;
        prev_uniq.score = score_of(h, prev_uniq.hit_length);
        r.score += prev_uniq.score;
        r.hit_length += hits[i].l;
      } else {
        bool created;
        SeqRecord &r = rec[k].at(seq_id, &created);
        if (created) { r.score = score; r.hit_length = hits[i].l; }
        else { r.score += score; r.hit_length += hits[i].l; }
        if (hits[i].ep == hits[i].sp) prev_uniq = SeqRecord{seq_id, score, hits[i].l};
      }
    }
  }

  uint64_t best = 0, second = 0, best_len = 0, second_len = 0;
  for (int k = 0; k <= 1; ++k)
    for (const SeqRecord &r : rec[k].v) {
      if (r.score > best) { second = best; second_len = best_len; best = r.score; best_len = (uint64_t)(int64_t)r.hit_length; }
      else if (r.score > second) { second = r.score; second_len = (uint64_t)(int64_t)r.hit_length; }
    }
  res.score = best;
  res.secondary_score = second;
  res.hit_length = (int32_t)best_len;
  res.query_length = query_len;

  std::vector<uint64_t> best_ids;
  auto used = [&](uint64_t id) { return std::find(best_ids.begin(), best_ids.end(), id) != best_ids.end(); };
  for (int k = 0; k <= 1; ++k)
    for (const SeqRecord &r : rec[k].v)
      if (r.score == best && !used(r.seq_id)) best_ids.push_back(r.seq_id);
  if (best_ids.size() > 1) res.secondary_score = best;
  if (second_len >= P.consider_secondary_hit_len && second < best &&
      second >= (uint64_t)(P.consider_secondary_score_factor * (double)best)) {
    for (int k = 0; k <= 1; ++k)
      for (const SeqRecord &r : rec[k].v)
        if (r.score == second && !used(r.seq_id)) best_ids.push_back(r.seq_id);
    res.secondary_score = second;
  }

  res.match_begin = matches.size();
  if ((int)best_ids.size() <= P.max_result || P.max_result <= 0) {
    for (uint64_t id : best_ids) matches.push_back(cfr_match{id, orig_taxid(h.tax, seq_to_tax(h.tax, id)), 0, 0});
  } else {
    std::vector<uint64_t> tids, promoted;
    tids.reserve(best_ids.size());
    for (uint64_t id : best_ids) tids.push_back(seq_to_tax(h.tax, id));
    std::vector<std::vector<uint64_t>> children;
    tax_reduce(h.tax, tids, P.max_result, promoted, expanded ? &children : nullptr);
    for (uint64_t ctid : promoted) matches.push_back(cfr_match{ctid, orig_taxid(h.tax, ctid), 1, 0});
    if (expanded && children.size() == promoted.size()) {      // Classifier.hpp:821-839
      expanded->spans.resize(matches.size(), cfr_span{0, 0});
      for (size_t q = 0; q < promoted.size(); ++q) {
        expanded->spans[res.match_begin + q] = cfr_span{expanded->ids.size(), children[q].size()};
        for (uint64_t c : children[q]) expanded->ids.push_back(orig_taxid(h.tax, c));
      }
    }
  }
  if (expanded) expanded->spans.resize(matches.size(), cfr_span{0, 0});
  res.n_match = (int32_t)(matches.size() - res.match_begin);
  res.pad = 0;
}

void classify_batch_tail(const HostIndex &h, const DeviceIndex::BatchOut &b, size_t n, int threads, cfr_result *results,
                         std::vector<cfr_match> &matches, ExpandedLists *expanded) {
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = n ? (int)n : 1;
  std::vector<std::vector<cfr_match>> part((size_t)threads);
  std::vector<ExpandedLists> xpart(expanded ? (size_t)threads : 0);
  auto work = [&](int tid) {
    const size_t lo = n * (size_t)tid / (size_t)threads, hi = n * (size_t)(tid + 1) / (size_t)threads;
    for (size_t i = lo; i < hi; ++i) {
      const uint64_t hb = b.hit_begin[i], he = b.hit_begin[i + 1];
      classify_read(h, b.hits.data() + hb, he - hb, b.row_begin.data() + hb, b.row_vals.data(), b.read_len[i], results[i], part[tid],
                    expanded ? &xpart[tid] : nullptr);
    }
  };
  if (threads == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  // stitch per-thread match arrays; reads were assigned in contiguous blocks so order is preserved
  matches.clear();
  if (expanded) { expanded->spans.clear(); expanded->ids.clear(); }
  for (int t = 0; t < threads; ++t) {
    const size_t lo = n * (size_t)t / (size_t)threads, hi = n * (size_t)(t + 1) / (size_t)threads;
    const uint64_t shift = matches.size();
    for (size_t i = lo; i < hi; ++i) results[i].match_begin += shift;
    matches.insert(matches.end(), part[t].begin(), part[t].end());
    if (expanded) {
      const uint64_t idshift = expanded->ids.size();
      for (cfr_span s : xpart[t].spans) { s.begin += idshift; expanded->spans.push_back(s); }
      expanded->ids.insert(expanded->ids.end(), xpart[t].ids.begin(), xpart[t].ids.end());
    }
  }
}

}  // namespace cfr
