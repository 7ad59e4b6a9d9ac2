// cfr_dust.cpp — SDUST low-complexity masking of a read, the pre-step the reference applies before
// Query (ClassifyReads_Thread, CentrifugerClass.cpp:276-316): window 64, threshold 20 (x10 scale),
// linker 1 (Dustmasker.hpp:247-249), alphabet "ACGT" + one code for everything else.
// Masked positions are overwritten with 'N'.  Host-side, per read; results must equal
// Dustmasker::MaskWithBuffer (Dustmasker.hpp:357-421) interval for interval.
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include "cfr_tail.hpp"

namespace cfr {

namespace {

constexpr int kWindow = 64, kThreshold = 20, kCodeBits = 3, kOther = 4;

struct Interval { size_t start, end; int score; };

inline int code_of(uint8_t ch) {
  switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return kOther; }
}

// State of one SDUST scan (Morgulis et al. 2006).  Triplet ring buffer of capacity 128 like
// Dustmasker_Queue(64) (Dustmasker.hpp:46-55).
// One Scanner per thread, reused from read to read (reset() instead of construction: no allocation in the steady state,
// 1 KB of counters to clear instead of 4 KB).
class Scanner {
 public:
  explicit Scanner(std::vector<Interval> &result) : result_(result) { reset(); }
  void reset() {
    memset(cw_, 0, sizeof(cw_));
    memset(cv_, 0, sizeof(cv_));
    perfect_.clear();
    head_ = tail_ = 0;
    rw_ = rv_ = lv_ = 0;
  }

  void run(const uint8_t *s, size_t n) {
    if (n < 3) return;
    int triplet = (code_of(s[0]) << kCodeBits) + code_of(s[1]);
    size_t wfinish;
    for (wfinish = 2; wfinish < n; ++wfinish) {
      const size_t wstart = wfinish + 1 > (size_t)kWindow ? wfinish + 1 - kWindow : 0;
      flush_before(wstart);
      triplet = ((triplet << kCodeBits) & 511) + code_of(s[wfinish]);
      shift(triplet);
      if (rw_ * 10 > lv_ * kThreshold) find_perfect(wstart);
    }
    size_t wstart = wfinish + 1 > (size_t)kWindow ? wfinish + 1 - kWindow : 0;
    while (!perfect_.empty()) { flush_before(wstart); ++wstart; }
  }

 private:
  int size() const { return (tail_ - head_) & 127; }
  int at(int i) const { return ring_[(head_ + i) & 127]; }
  static void add(int t, uint8_t *cnt, int &r) { r += cnt[t]; ++cnt[t]; }       // counts stay below the window (64)
  static void remove(int t, uint8_t *cnt, int &r) { --cnt[t]; r -= cnt[t]; }

  // Dustmasker::ShiftWindow (:106-138)
  void shift(int t) {
    if (size() >= kWindow - 2) {
      const int old = ring_[head_];
      remove(old, cw_, rw_);
      head_ = (head_ + 1) & 127;
      if (lv_ > size()) { remove(old, cv_, rv_); --lv_; }
    }
    ring_[tail_] = t;
    tail_ = (tail_ + 1) & 127;
    ++lv_;
    add(t, cw_, rw_);
    add(t, cv_, rv_);
    if (cv_[t] * 10 > 2 * kThreshold) {
      for (;;) {
        const int s = at(size() - lv_);
        remove(s, cv_, rv_);
        --lv_;
        if (s == t) break;
      }
    }
  }

  // Dustmasker::SaveMaskedRegions (:141-167)
  void flush_before(size_t window_start) {
    if (perfect_.empty() || perfect_.back().start >= window_start) return;
    const Interval last = perfect_.back();
    if (!result_.empty() && last.start <= result_.back().end + 1) {
      if (last.end > result_.back().end) result_.back().end = last.end;
    } else {
      result_.push_back(last);
    }
    while (!perfect_.empty() && perfect_.back().start < window_start) perfect_.pop_back();
  }

  // Dustmasker::FindPerfect (:172-243); perfect_ is kept sorted by descending start
  void find_perfect(size_t window_start) {
    int rv = rv_;
    int max_score = 0, max_cnt = 1;
    const int first = size() - lv_ - 1;
    for (int i = first; i >= 0; --i) {
      add(at(i), cv_, rv);
      const int triplets = size() - i - 1;
      if (rv * 10 > kThreshold * triplets) {
        size_t it = 0;
        while (it != perfect_.size() && perfect_[it].start >= (size_t)i + window_start) {
          const Interval &p = perfect_[it];
          if ((uint64_t)(int64_t)p.score * (uint64_t)(int64_t)max_cnt > (uint64_t)(int64_t)max_score * (uint64_t)(p.end - p.start - 2)) {
            max_score = p.score;
            max_cnt = (int)(p.end - p.start - 2);
          }
          ++it;
        }
        if (rv * max_cnt >= max_score * triplets) {
          max_score = rv;
          max_cnt = triplets;
          perfect_.insert(perfect_.begin() + (long)it, Interval{(size_t)i + window_start, window_start + (size_t)size() + 1, rv});
        }
      }
    }
    for (int i = first; i >= 0; --i) remove(at(i), cv_, rv);
  }

  std::vector<Interval> &result_;
  std::vector<Interval> perfect_;
  int16_t ring_[128];
  int head_ = 0, tail_ = 0;
  uint8_t cw_[512], cv_[512];
  int rw_ = 0, rv_ = 0, lv_ = 0;
};

}  // namespace

// The segments of a read the scan runs on: leading non-symbols are skipped, a run of more than 64 non-symbols ends a
// segment (CentrifugerClass.cpp:276-316 through Dustmasker::MaskWithBuffer).  f(start, length) per segment.
template <class F>
static void for_each_segment(const uint8_t *s, size_t n, F f) {
  size_t i = 0;
  while (i < n && code_of(s[i]) == kOther) ++i;
  while (i < n) {
    size_t run = 0, last_valid = i, j = i;
    for (; j < n; ++j) {
      if (code_of(s[j]) == kOther) ++run;
      else {
        if (run > (size_t)kWindow) break;   // leave a very long run of Ns
        last_valid = j;
        run = 0;
      }
    }
    if (last_valid > i) f(i, last_valid - i + 1);
    i = j;
  }
}

// Literal form: the reference's list of perfect intervals as it is (grows to 1711 entries on a homopolymer and is rescanned
// for every window suffix: milliseconds per poly-A read).  Kept as the anchor the bounded form below and the device kernel
// are compared with (tests/test_host_cpu.py, tests/test_gpu_dust.py); cfr_dust_mask_batch runs the bounded form.
void dust_mask_literal(uint8_t *s, size_t n) {
  if (n < 3) return;
  thread_local std::vector<Interval> all, part;
  thread_local Scanner sc(part);
  all.clear();
  for_each_segment(s, n, [&](size_t i, size_t m) {
    part.clear();
    sc.reset();
    sc.run(s + i, m);
    for (const Interval &iv : part) all.push_back(Interval{iv.start + i, iv.end + i, iv.score});
  });
  for (const Interval &iv : all)
    for (size_t p = iv.start; p <= iv.end; ++p) s[p] = 'N';
}

// Bounded form (the same reformulation as the device kernel k_dust, cfr_kernels.hip.inc): what the scan takes from the list
// is the best score/length ratio among the entries with start >= a threshold that only moves down within one FindPerfect
// call, and the end of the entry inserted last for the smallest start; entries live only while their start is inside the
// window.  One slot per start (64 slots, a mask of the live ones), every live start folded once per call: same masks,
// linear work.
namespace {
inline uint64_t rotr64(uint64_t x, unsigned r) { r &= 63u; return r ? (x >> r) | (x << (64u - r)) : x; }
inline uint64_t rotl64(uint64_t x, unsigned r) { r &= 63u; return r ? (x << r) | (x >> (64u - r)) : x; }

void scan_bounded(uint8_t *seg, size_t m) {
  uint8_t cw[512] = {0}, cv[512] = {0};
  uint32_t pf[64];                     // slot (start & 63): score << 8 | (end - start)
  int head = 0, tail = 0, rw = 0, rv = 0, lv = 0;
  uint64_t live = 0;
  size_t base = 0;
  int16_t ring16[64];
  auto size = [&]() { return (tail - head) & 63; };
  auto at = [&](int q) -> int { return ring16[(head + q) & 63]; };
  auto add = [&](int t, uint8_t *cnt, int &r) { r += cnt[t]; ++cnt[t]; };
  auto rem = [&](int t, uint8_t *cnt, int &r) { --cnt[t]; r -= cnt[t]; };
  auto flush_before = [&](size_t ws) {
    if (live) {
      uint64_t rot = rotr64(live, (unsigned)base);
      const size_t smin = base + (size_t)__builtin_ctzll(rot);
      if (smin < ws) {
        const uint32_t e = pf[smin & 63];
        for (size_t p = smin; p <= smin + (e & 0xffu); ++p) seg[p] = 'N';
        const size_t drop = ws - base;
        rot = drop >= 64 ? 0ull : rot & ~((1ull << drop) - 1ull);
        live = rotl64(rot, (unsigned)base);
      }
    }
    base = ws;
  };
  int triplet = (code_of(seg[0]) << kCodeBits) + code_of(seg[1]);
  // (the masking stores land behind the scan position, so reading seg while writing it is safe)
  size_t wf;
  for (wf = 2; wf < m; ++wf) {
    const size_t ws = wf + 1 > (size_t)kWindow ? wf + 1 - kWindow : 0;
    flush_before(ws);
    triplet = ((triplet << kCodeBits) & 511) + code_of(seg[wf]);
    const int t = triplet;
    if (size() >= kWindow - 2) {
      const int old = ring16[head];
      rem(old, cw, rw);
      head = (head + 1) & 63;
      if (lv > size()) { rem(old, cv, rv); --lv; }
    }
    ring16[tail] = (int16_t)t;
    tail = (tail + 1) & 63;
    ++lv;
    add(t, cw, rw);
    add(t, cv, rv);
    if (cv[t] * 10 > 2 * kThreshold) {
      for (;;) {
        const int q = at(size() - lv);
        rem(q, cv, rv);
        --lv;
        if (q == t) break;
      }
    }
    if (rw * 10 > lv * kThreshold) {
      int rvl = rv, max_score = 0, max_cnt = 1;
      uint64_t unfolded = rotr64(live, (unsigned)base), fresh = 0;
      const int first = size() - lv - 1;
      int q_end = -1;       // a suffix's score never exceeds the window's (rw) and its triplet count only grows: stop where no start can qualify any more
      for (int q = first; q >= 0; --q) {
        const int triplets = size() - q - 1;
        if (kThreshold * triplets >= rw * 10) { q_end = q; break; }
        add(at(q), cv, rvl);
        if (rvl * 10 > kThreshold * triplets) {
          while (unfolded >> q) {
            const int p = 63 - __builtin_clzll(unfolded);
            unfolded &= ~(1ull << p);
            const uint32_t e = pf[(base + (size_t)p) & 63];
            const int sc = (int)(e >> 8), cn = (int)(e & 0xffu) - 2;
            if ((uint64_t)(int64_t)sc * (uint64_t)(int64_t)max_cnt > (uint64_t)(int64_t)max_score * (uint64_t)cn) { max_score = sc; max_cnt = cn; }
          }
          if (rvl * max_cnt >= max_score * triplets) {
            max_score = rvl;
            max_cnt = triplets;
            pf[(base + (size_t)q) & 63] = ((uint32_t)rvl << 8) | (uint32_t)(size() + 1 - q);
            fresh |= 1ull << q;
          }
        }
      }
      live |= rotl64(fresh, (unsigned)base);
      for (int q = first; q > q_end; --q) rem(at(q), cv, rvl);
    }
  }
  size_t ws = wf + 1 > (size_t)kWindow ? wf + 1 - kWindow : 0;
  while (live) { flush_before(ws); ++ws; }
}
}  // namespace

void dust_mask(uint8_t *s, size_t n) {
  if (n < 3) return;
  // segment boundaries are decided on the unmasked read (masking only turns symbols into non-symbols behind the scan,
  // and the literal form masks after all segments are done): collect them first
  thread_local std::vector<std::pair<size_t, size_t>> segs;
  segs.clear();
  for_each_segment(s, n, [&](size_t i, size_t m) { segs.emplace_back(i, m); });
  for (const auto &sg : segs) scan_bounded(s + sg.first, sg.second);
}

const char *tax_rank_string(uint8_t rank) {   // Taxonomy::GetTaxRankString (Taxonomy.hpp:497-533)
  static const char *names[] = {
      "no rank", "strain", "species", "genus", "family", "order", "class", "phylum", "kingdom", "domain", "forma",
      "infraclass", "infraorder", "parvorder", "subclass", "subfamily", "subgenus", "subkingdom", "suborder",
      "subphylum", "subspecies", "subtribe", "superclass", "superfamily", "superkingdom", "superorder", "superphylum",
      "tribe", "varietas", "life", "acellular root"};
  return rank < sizeof(names) / sizeof(names[0]) ? names[rank] : "no rank";
}

}  // namespace cfr
