// cfr_stress.cpp — in-process stress of cfr_index_open / cfr_index_destroy beside live device images (a plain C++ client of the
// C-ABI; tools/stress_open.sh and tests/test_gpu_stress.py run it, `make SAN=...` builds it against the sanitizer variants).
//
// Why it exists: round 3 saw cfr_index_open fail its decode check on a golden protein index twice in ~110 GPU-side test runs, in
// processes that opened the same file before and after, and never in 24 000 CPU-side parses.  What differs on the GPU box is a process
// with live cfr_dev_index objects: worker threads, asynchronous copies into caller memory, pinned staging.  This program puts exactly
// that side by side: traffic threads keep batches in flight on long-lived images (submit / wait, result buffers allocated and freed
// per batch so that the heap churns) while opener threads open, digest and destroy every index of a list thousands of times and now
// and then build and destroy a device image of their own.  Every open must succeed, every digest must equal the first one, every
// batch's results must equal the first batch's.  Exit code 0 only then.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cfr_hip.h"

namespace {

struct Reads { std::vector<uint8_t> bases; std::vector<uint64_t> offs; size_t n() const { return offs.size() - 1; } };

Reads read_fastx(const std::string &path, size_t limit) {
  Reads r;
  r.offs.push_back(0);
  FILE *fp = fopen(path.c_str(), "r");
  if (!fp) { fprintf(stderr, "cfr_stress: cannot open %s\n", path.c_str()); exit(2); }
  char *line = nullptr;
  size_t cap = 0;
  ssize_t len;
  int state = 0;                  // 0: expect header, 1: FASTA sequence lines, 2: FASTQ sequence, 3: '+', 4: quality
  bool open_rec = false;
  while ((len = getline(&line, &cap, fp)) > 0) {
    while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) --len;
    if (state == 0 || state == 1) {
      if (len > 0 && (line[0] == '>' || (line[0] == '@' && state == 0))) {
        if (open_rec) { r.offs.push_back(r.bases.size()); if (r.n() >= limit) { open_rec = false; break; } }
        open_rec = true;
        state = line[0] == '>' ? 1 : 2;
        continue;
      }
      if (state == 1) r.bases.insert(r.bases.end(), line, line + len);
    } else if (state == 2) { r.bases.insert(r.bases.end(), line, line + len); state = 3; }
    else if (state == 3) state = 4;
    else { state = 0; }
  }
  if (open_rec) r.offs.push_back(r.bases.size());
  free(line);
  fclose(fp);
  return r;
}

Reads random_reads(size_t n, unsigned seed) {
  Reads r;
  r.offs.push_back(0);
  uint64_t x = 88172645463325252ull ^ seed;
  for (size_t i = 0; i < n; ++i) {
    for (int k = 0; k < 150; ++k) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; r.bases.push_back("ACGT"[x & 3]); }
    r.offs.push_back(r.bases.size());
  }
  return r;
}

std::vector<std::string> split(const std::string &s) {
  std::vector<std::string> out;
  size_t a = 0;
  while (a <= s.size()) { size_t b = s.find(',', a); if (b == std::string::npos) b = s.size(); if (b > a) out.push_back(s.substr(a, b - a)); a = b + 1; }
  return out;
}

std::mutex g_log;
std::atomic<uint64_t> g_fail{0};
void fail(const std::string &what) {
  std::lock_guard<std::mutex> lk(g_log);
  ++g_fail;
  fprintf(stderr, "cfr_stress: FAIL %s\n", what.c_str());
}

}  // namespace

int main(int argc, char **argv) {
  std::string load, opens_list, reads_path;
  long opens = 2000, devimg_every = 50, opener_threads = 2, batch_reads = 20000;
  int device = 0;
  for (int i = 1; i < argc; ++i) {
    auto arg = [&](const char *name) { return !strcmp(argv[i], name) && i + 1 < argc; };
    if (arg("--load")) load = argv[++i];
    else if (arg("--open")) opens_list = argv[++i];
    else if (arg("--reads")) reads_path = argv[++i];
    else if (arg("--opens")) opens = atol(argv[++i]);
    else if (arg("--devimg-every")) devimg_every = atol(argv[++i]);
    else if (arg("--opener-threads")) opener_threads = atol(argv[++i]);
    else if (arg("--batch-reads")) batch_reads = atol(argv[++i]);
    else if (arg("--device")) device = atoi(argv[++i]);
    else { fprintf(stderr, "usage: cfr_stress --load P1[,P2] --open P1,P2,... [--reads file] [--opens N] [--devimg-every M] [--opener-threads T] [--batch-reads R] [--device D]\n"); return 2; }
  }
  if (load == "none") { load.clear(); devimg_every = 0; }      // host only (no GPU in the process): the openers alone
  const auto load_prefixes = split(load), open_prefixes = split(opens_list);
  if (open_prefixes.empty()) { fprintf(stderr, "cfr_stress: --open is required (--load none: no device images)\n"); return 2; }
  Reads reads = reads_path.empty() ? random_reads((size_t)batch_reads, 1) : read_fastx(reads_path, (size_t)batch_reads);
  if (reads.n() == 0) { fprintf(stderr, "cfr_stress: no reads\n"); return 2; }
  for (const size_t n0 = reads.n(), b0 = reads.bases.size(); reads.n() + n0 <= (size_t)batch_reads;) {      // a small file: repeated up to the batch size
    const size_t at = reads.bases.size();
    reads.bases.insert(reads.bases.end(), reads.bases.begin(), reads.bases.begin() + (long)b0);
    for (size_t i = 1; i <= n0; ++i) reads.offs.push_back(at + reads.offs[i]);
  }
  const auto t0 = std::chrono::steady_clock::now();

  // ---- long-lived images with batches in flight
  struct Live { cfr_index *idx = nullptr; cfr_dev_index *dev = nullptr; int k = 1; std::vector<cfr_result> want_res; std::vector<cfr_match> want_match; };
  std::vector<Live> live(load_prefixes.size());
  for (size_t q = 0; q < live.size(); ++q) {
    cfr_params prm;
    cfr_params_default(&prm);
    prm.max_result = live[q].k = q % 2 ? 5 : 1;
    if (cfr_index_open(load_prefixes[q].c_str(), &prm, &live[q].idx) != CFR_OK) { fprintf(stderr, "cfr_stress: open %s: %s\n", load_prefixes[q].c_str(), cfr_last_error()); return 2; }
    if (cfr_device_index_create(live[q].idx, device, &live[q].dev) != CFR_OK) { fprintf(stderr, "cfr_stress: device image %s: %s\n", load_prefixes[q].c_str(), cfr_last_error()); return 2; }
    cfr_device_index_set_dust(live[q].dev, 1);
    live[q].want_res.resize(reads.n());
    live[q].want_match.resize(reads.n() * (size_t)live[q].k);
    size_t nm = 0;
    memset(live[q].want_match.data(), 0, live[q].want_match.size() * sizeof(cfr_match));
    if (cfr_classify_batch(live[q].dev, reads.bases.data(), reads.offs.data(), nullptr, nullptr, reads.n(), live[q].want_res.data(),
                           live[q].want_match.data(), live[q].want_match.size(), &nm) != CFR_OK) { fprintf(stderr, "cfr_stress: first batch: %s\n", cfr_last_error()); return 2; }
  }
  std::atomic<bool> stop{false};
  std::atomic<uint64_t> batches{0};
  auto same_results = [&](const Live &L, const cfr_result *res, const cfr_match *mat) {
    for (size_t i = 0; i < reads.n(); ++i) {
      const cfr_result &a = L.want_res[i], &b = res[i];
      if (a.score != b.score || a.secondary_score != b.secondary_score || a.hit_length != b.hit_length || a.query_length != b.query_length || a.n_match != b.n_match) return false;
      for (int m = 0; m < a.n_match; ++m) {
        const cfr_match &x = L.want_match[a.match_begin + (uint64_t)m], &y = mat[b.match_begin + (uint64_t)m];
        if (x.id != y.id || x.taxid != y.taxid || x.kind != y.kind) return false;
      }
    }
    return true;
  };
  std::vector<std::thread> traffic;
  for (size_t q = 0; q < live.size(); ++q) traffic.emplace_back([&, q]() {
    Live &L = live[q];
    const size_t mcap = reads.n() * (size_t)L.k;
    while (!stop.load()) {
      // two batches in flight; pageable result buffers allocated per batch and freed right after (heap churn next to the openers)
      cfr_result *r[2];
      cfr_match *m[2];
      cfr_ticket t[2];
      bool ok[2] = {false, false};
      for (int j = 0; j < 2; ++j) {
        r[j] = (cfr_result *)malloc(reads.n() * sizeof(cfr_result));
        m[j] = (cfr_match *)calloc(mcap, sizeof(cfr_match));
        ok[j] = cfr_classify_batch_submit(L.dev, reads.bases.data(), reads.offs.data(), nullptr, nullptr, reads.n(), r[j], m[j], mcap, &t[j]) == CFR_OK;
        if (!ok[j]) fail(std::string("submit: ") + cfr_last_error());
      }
      for (int j = 0; j < 2; ++j) {
        if (ok[j]) {
          size_t nm = 0;
          if (cfr_classify_batch_wait(L.dev, t[j], &nm) != CFR_OK) fail(std::string("wait: ") + cfr_last_error());
          else if (!same_results(L, r[j], m[j])) fail("a batch's results differ from the first batch's (" + load_prefixes[q] + ")");
          ++batches;
        }
        free(r[j]);
        free(m[j]);
      }
    }
  });

  // ---- openers
  std::vector<uint64_t> first_digest(open_prefixes.size(), 0);
  std::vector<std::atomic<int>> have(open_prefixes.size());
  for (auto &x : have) x = 0;
  std::mutex dig_mu;
  std::atomic<uint64_t> n_open{0}, n_img{0};
  std::vector<std::thread> openers;
  for (long t = 0; t < opener_threads; ++t) openers.emplace_back([&, t]() {
    for (long it = t; it < opens; it += opener_threads) {
      for (size_t q = 0; q < open_prefixes.size(); ++q) {
        cfr_index *idx = nullptr;
        cfr_params prm;
        cfr_params_default(&prm);
        if (cfr_index_open(open_prefixes[q].c_str(), &prm, &idx) != CFR_OK) { fail("open #" + std::to_string(it) + " of " + open_prefixes[q] + ": " + cfr_last_error()); continue; }
        ++n_open;
        uint64_t dg = 0;
        cfr_index_digest(idx, &dg);
        {
          std::lock_guard<std::mutex> lk(dig_mu);
          if (!have[q]) { first_digest[q] = dg; have[q] = 1; }
          else if (first_digest[q] != dg) fail("digest of " + open_prefixes[q] + " changed at open #" + std::to_string(it));
        }
        if (devimg_every > 0 && (it * (long)open_prefixes.size() + (long)q) % devimg_every == 0) {
          cfr_device_options o;
          cfr_device_options_default(&o);
          o.profile = CFR_PROFILE_FAST_LOAD;
          cfr_dev_index *d = nullptr;
          if (cfr_device_index_create_ex(idx, device, &o, &d) != CFR_OK) fail("device image of " + open_prefixes[q] + ": " + cfr_last_error());
          else {
            const size_t nr = std::min<size_t>(reads.n(), 500);
            std::vector<cfr_result> res(nr);
            std::vector<cfr_match> mat(nr);
            size_t nm = 0;
            if (cfr_classify_batch(d, reads.bases.data(), reads.offs.data(), nullptr, nullptr, nr, res.data(), mat.data(), mat.size(), &nm) != CFR_OK)
              fail("classify on a fresh image of " + open_prefixes[q] + ": " + cfr_last_error());
            cfr_device_index_destroy(d);
            ++n_img;
          }
        }
        cfr_index_destroy(idx);
      }
    }
  });
  for (auto &x : openers) x.join();
  stop = true;
  for (auto &x : traffic) x.join();
  for (auto &L : live) { cfr_device_index_destroy(L.dev); cfr_index_destroy(L.idx); }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("cfr_stress: %llu opens of %zu indexes, %llu device images built and destroyed, %llu batches of %zu reads on %zu live images, %llu failures, %.1f s\n",
         (unsigned long long)n_open.load(), open_prefixes.size(), (unsigned long long)n_img.load(), (unsigned long long)batches.load(), reads.n(), live.size(),
         (unsigned long long)g_fail.load(), secs);
  return g_fail.load() ? 1 : 0;
}
