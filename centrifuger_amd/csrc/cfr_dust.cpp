// cfr_dust.cpp — SDUST low-complexity masking of a read, the pre-step the reference applies before
// Query (ClassifyReads_Thread, CentrifugerClass.cpp:276-316): window 64, threshold 20 (x10 scale),
// linker 1 (Dustmasker.hpp:247-249), alphabet "ACGT" + one code for everything else.
// Masked positions are overwritten with 'N'.  Host-side, per read; results must equal
// Dustmasker::MaskWithBuffer (Dustmasker.hpp:357-421) interval for interval.
#include <cstdint>
#include <cstring>
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

void dust_mask(uint8_t *s, size_t n) {
  if (n < 3) return;
  thread_local std::vector<Interval> all, part;
  thread_local Scanner sc(part);
  all.clear();
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
    if (last_valid > i) {
      part.clear();
      sc.reset();
      sc.run(s + i, last_valid - i + 1);
      for (const Interval &iv : part) all.push_back(Interval{iv.start + i, iv.end + i, iv.score});
    }
    i = j;
  }
  for (const Interval &iv : all)
    for (size_t p = iv.start; p <= iv.end; ++p) s[p] = 'N';
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
