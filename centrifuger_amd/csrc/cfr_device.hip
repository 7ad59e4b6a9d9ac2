// cfr_device.hip — builds the HBM image of an index and drives the batch pipeline (gfx950 only).
#include "cfr_device.hpp"

#include <hipcub/hipcub.hpp>
#include <chrono>

#include <algorithm>
#include <atomic>
#include <functional>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <type_traits>

#include "cfr_tail.hpp"      // dust_mask: the host twin the device SDUST falls back to
#include "cfr_kernels.hip.inc"

namespace cfr {

namespace {

inline void hip_check(hipError_t e, const char *what) {
  if (e != hipSuccess) throw HipError{std::string(what) + ": " + hipGetErrorString(e), (int)e};
}
#define HIP_CHECK(x) hip_check((x), #x)

// The CFR_* switches of profiles/HISTORY.md section 5 are A/B and test hooks, not part of the library's interface: they are only
// looked at when CFR_DEBUG_ENV=1 is set in the environment (one documented gate; a production caller never sets it and
// the library then has no hidden inputs - everything a caller chooses goes through cfr_device_options).
inline const char *dbg_env(const char *name) {
  static const bool on = [] { const char *g = ::getenv("CFR_DEBUG_ENV"); return g && atoi(g) != 0; }();
  return on ? ::getenv(name) : nullptr;
}

constexpr int kBlock = 256;
constexpr size_t kCtlWords = 4;      // words of a sub-batch's control block the host reads back: pool overflow, reads folded by small / large teams, reads left to k_adjust_tail
inline unsigned grid_for(size_t n, int block = kBlock) { return (unsigned)((n + block - 1) / block); }

enum Slot : size_t {
  S_CAP = 0, S_HITOFF, S_RAW, S_CHAINCNT, S_SCAN2, S_POOL_E, S_POOL_V, S_POOLCTL, S_POOLCTL1, S_FIN, S_FINCNT, S_FINROWS, S_FINOFF, S_HITS, S_ROWSPER, S_ROWOFF, S_ROWS,
  S_ROWVALS, S_PACK1, S_PACK2, S_READROWS, S_READROWOFF, S_ENTRIES, S_RESULTS, S_MATCHES, S_RESULTS1, S_MATCHES1, S_SCAN, S_IN_B1, S_IN_O1, S_IN_B2, S_IN_O2, S_DUSTPOOL, S_DUSTPOOL2, S_DUSTTMP, S_DUSTTMP2, S_DUSTFLAG, S_DUSTFLAG2, S_HEAVY, S_HEAVYB, S_HEAVYB1, S_CAP1, S_HITOFF1, S_RAW1, S_CHAINCNT1, S_SCAN1, S_HEAVY1, S_CRES, S_CRES1, S_CMATCH, S_CMATCH1, S_WIDEIDX, S_WIDERES, S_WIDEMATCH, S_WIDECNT, S_PCODES1, S_PCODES2, S_CAPALL, S_HITALL, S_SCANALL, S_P0, S_P1, S_P2, S_P3, S_P4, S_P5, S_EXPPOOL, S_EXPCUR, S_SLOW, S_SLOW1, S_LIST, S_LIST1, S_LISTCTR, S_COUNT
};

}  // namespace

template <class T> T *DeviceIndex::dev_alloc(size_t count) {
  void *p = nullptr;
  size_t bytes = std::max<size_t>(count * sizeof(T), 16);
  HIP_CHECK(hipMalloc(&p, bytes));
  owned_.push_back(p);
  device_bytes_ += bytes;
  return (T *)p;
}

template <class T> T *DeviceIndex::upload(const std::vector<T> &v) {
  T *d = dev_alloc<T>(v.size());
  if (!v.empty()) HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

void *DeviceIndex::scratch(size_t slot, size_t bytes) {
  if (slots_.size() < S_COUNT) slots_.resize(S_COUNT);
  Slot &s = slots_[slot];
  if (s.cap < bytes) {
    if (s.p) HIP_CHECK(hipFree(s.p));
    s.p = nullptr;
    size_t cap = std::max<size_t>(bytes + bytes / 4, 256);
    HIP_CHECK(hipMalloc(&s.p, cap));
    s.cap = cap;
  }
  return s.p;
}

// A constructor that throws never runs the destructor: everything the image owns is released here before the error leaves
// (a bad argument from a library caller must not strand tens of GB of HBM until the process exits).
DeviceIndex::DeviceIndex(const HostIndex &h, int device, const cfr_device_options &opt) : host_(&h), device_(device) {
  // arguments first, before anything is allocated
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw HipError{"no HIP device available (libcfr_hip has no CPU fallback)", -1};
  if (device < 0 || device >= count) throw HipError{"device ordinal out of range", -1};
  // (-k has no cap in the reference, Classifier.hpp:17-38, 738-781.  The device tail has none either since round 5 - its listings are
  // arrays in pool scratch or a team's slot list - but the match buffers are max_result slots per read: 4096 keeps a sub-batch's below 100 GB)
  if (h.params.max_result > 4096) throw HipError{"-k / max_result above 4096: the match buffers (max_result slots per read) would not fit beside the index", -2};
  try {
    init(h, opt);
  } catch (...) {
    release();
    throw;
  }
}

void DeviceIndex::init(const HostIndex &h, const cfr_device_options &opt) {
  const int device = device_;
  HIP_CHECK(hipSetDevice(device));
  HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  for (auto &set : evs_) for (auto &e : set) HIP_CHECK(hipEventCreate(&e));
  ev_ = evs_[0];
  HIP_CHECK(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
  for (auto &e : tail_done_) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto &e : copy_done_) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto &e : h2d_done_) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_CHECK(hipStreamCreateWithFlags(&h2d_stream_, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&search2_stream_, hipStreamNonBlocking));
  HIP_CHECK(hipEventCreateWithFlags(&prep_done_, hipEventDisableTiming));
  HIP_CHECK(hipStreamCreateWithFlags(&dust_stream_, hipStreamNonBlocking));
  for (auto &e : copied_) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  {
    // its own priority class: streams of one class share a few hardware queues round-robin, and a post-stage stream that
    // lands on the search stream's queue runs behind it instead of beside it (seen with the third image of a process)
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
    const bool high = dbg_env("CFR_TAIL_PRIO") && atoi(dbg_env("CFR_TAIL_PRIO")) > 0;      // A/B: the post stage in the HIGHEST class (profiles/r5_ab_post_fast.txt)
    if (least != greatest) HIP_CHECK(hipStreamCreateWithPriority(&tail_stream_, hipStreamNonBlocking, high ? greatest : least));
    else HIP_CHECK(hipStreamCreateWithFlags(&tail_stream_, hipStreamNonBlocking));
  }
  for (auto &e : search_done_) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  if (const char *e = dbg_env("CFR_TAIL_STREAM")) tail_overlap_mode_ = atoi(e) != 0 ? 1 : 0;
  if (const char *e = dbg_env("CFR_TAIL_BLOCKS")) tail_blocks_per_cu_ = atoi(e);
  if (const char *e = dbg_env("CFR_FUSED_POST")) fused_post_ = atoi(e) != 0;
  if (const char *e = dbg_env("CFR_TEAM_TAIL")) team_tail_ = atoi(e) != 0;
  if (const char *e = dbg_env("CFR_POOL_CAP")) pool_cap_ = strtoull(e, nullptr, 10);        // fixed size (no growth)
  else if (const char *e2 = dbg_env("CFR_POOL_INIT")) pool_cap_ = strtoull(e2, nullptr, 10);  // first size (grows on overflow)
  if (opt.sub_batch) sub_batch_ = (size_t)opt.sub_batch;
  if (const char *e = dbg_env("CFR_SUBBATCH")) sub_batch_ = std::max<size_t>(1, strtoull(e, nullptr, 10));
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, device));
  num_cus_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char *e = dbg_env("CFR_SEARCH_V1")) search_v1_ = atoi(e) != 0;
  if (const char *e = dbg_env("CFR_FUSED_TAIL")) fused_tail_ = atoi(e) != 0;
  if (const char *e = dbg_env("CFR_BLOCKS_PER_CU")) { blocks_per_cu_ = std::max(1, atoi(e)); blocks_forced_ = true; }
  if (const char *e = dbg_env("CFR_TAPER_FLOOR")) taper_floor_ = strtoull(e, nullptr, 10);
  wide_ = h.n >= 0xfffffff0ull;
  if (const char *e = dbg_env("CFR_FORCE_WIDE")) wide_ = wide_ || atoi(e) != 0;      // test hook: the n >= 2^32 code path on a small index

  // CFR_LOAD_TIMING=1: seconds per load stage on stderr
  const bool load_timing = dbg_env("CFR_LOAD_TIMING") && atoi(dbg_env("CFR_LOAD_TIMING"));
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!load_timing) return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[load] %-28s %7.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  lap("context, streams, events");
  // CFR_PROFILE=fast-load: skip the large derived tables (a command-line run is bound by FASTQ parsing, not by the device;
  // what it feels is the load time).  Default: throughput (all tables).  The specific switches below override either.
  bool fast_load = opt.profile == CFR_PROFILE_FAST_LOAD;
  bool balanced = opt.profile == CFR_PROFILE_BALANCED;
  if (const char *e = dbg_env("CFR_PROFILE")) { fast_load = std::string(e) == "fast-load"; balanced = std::string(e) == "balanced"; }
  bool layout_rb = opt.run_block_layout != 0;
  if (const char *e = dbg_env("CFR_LAYOUT")) layout_rb = std::string(e) == "rb";
  memset(&view_.rb, 0, sizeof(view_.rb));
  memset(&view_.prot, 0, sizeof(view_.prot));
  uint64_t *d_occ = nullptr;
  if (h.prot.enabled) {
    // ---- protein image: bit planes + per-block counts of the decoded BWT (cfr_device.hpp ProtView)
    const ProteinPart &P = h.prot;
    const uint64_t nblk = (h.n >> 6) + 2;
    std::vector<uint64_t> planes(nblk * 8, 0), counts(nblk * 32, 0);
    uint64_t run[32] = {0};
    for (uint64_t blk = 0; blk < nblk; ++blk) {
      for (int k = 0; k < 32; ++k) counts[blk * 32 + (uint64_t)k] = run[k];
      for (uint64_t j = 0; j < 64; ++j) {
        const uint64_t p = blk * 64 + j;
        if (p >= h.n) break;
        const uint32_t c = P.bwt[p];
        for (int k = 0; k < 5; ++k) planes[blk * 8 + (uint64_t)k] |= (uint64_t)((c >> k) & 1u) << j;
        ++run[c];
      }
    }
    view_.prot.enabled = 1;
    view_.prot.sigma = P.sigma;
    view_.prot.bits = P.bits;
    view_.prot.rec = nullptr; view_.prot.planes = nullptr; view_.prot.counts = nullptr;
    const bool one_line = h.n < 0xffffffffull && P.sigma <= 22 && !(dbg_env("CFR_PROT_TWO_ARRAYS") && atoi(dbg_env("CFR_PROT_TWO_ARRAYS")));
    if (one_line) {                      // counts and planes of a block in one 128-byte record (cfr_device.hpp)
      std::vector<uint64_t> rec(nblk * 16, 0);
      for (uint64_t blk = 0; blk < nblk; ++blk) {
        uint32_t *c32 = reinterpret_cast<uint32_t *>(&rec[blk * 16]);
        for (uint32_t k = 0; k < 22; ++k) c32[k] = (uint32_t)counts[blk * 32 + k];
        for (int k = 0; k < 5; ++k) rec[blk * 16 + 11 + (uint64_t)k] = planes[blk * 8 + (uint64_t)k];
      }
      view_.prot.rec = upload(rec);
    } else {
      view_.prot.planes = upload(planes);
      view_.prot.counts = upload(counts);
    }
    view_.prot.endmarker_bits = (uint32_t)P.end_marker_bits;
    view_.prot.endmarker_n = P.end_marker_n;
    view_.prot.endmarker = upload(P.end_marker_words.empty() ? std::vector<uint64_t>(2, 0) : P.end_marker_words);
    view_.prot.code_of = upload(std::vector<uint8_t>(P.code_of, P.code_of + 256));
    for (uint32_t k = 0; k <= P.sigma; ++k) view_.prot.C[k] = P.C[k];
    memcpy(view_.prot.list, P.list, 32);
    search_v1_ = true;                   // (the state-machine kernel is a nucleotide kernel)
    lap("protein image (host) + upload");
  } else {
    // ---- run-block image: the 7 bitvectors as rank lines (cfr_device.hpp).  CFR_LAYOUT=rb searches on it directly;
    // otherwise it is only the source the flat occ image is expanded from (on the device) and is freed afterwards.
    std::vector<void *> rb_allocs;
    // Small vectors: laid out in one host buffer and copied.  Large ones (a multi-Gbp index holds GBs of bitvector words) are
    // streamed: slices of 64 k lines are counted by host threads (ones before every slice), then laid out - with their absolute
    // counts - straight into two pinned staging buffers of 32 slices that travel to the device while the next ones are filled
    // (one pass of pageable 14 GB through hipMemcpy cost 2.8 s of the 40 Gbp load).
    struct Stage { uint64_t *p = nullptr; hipEvent_t done = nullptr; bool busy = false; };
    Stage stage[2];
    constexpr uint64_t kSliceLines = 1ull << 16, kChunkSlices = 32;                       // 4 MB slices, 128 MB per staging buffer
    auto release_stages = [&]() {
      for (auto &g : stage) { if (g.busy) (void)hipEventSynchronize(g.done); if (g.done) (void)hipEventDestroy(g.done); if (g.p) (void)hipHostFree(g.p); g = Stage{}; }
    };
    const unsigned host_threads = std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    auto run_parallel = [&](uint64_t count, const std::function<void(uint64_t)> &fn) {
      std::atomic<uint64_t> next{0};
      const unsigned nt = (unsigned)std::min<uint64_t>(host_threads, count);
      if (nt <= 1) { for (uint64_t k = 0; k < count; ++k) fn(k); return; }
      std::vector<std::thread> th;
      for (unsigned t = 0; t < nt; ++t) th.emplace_back([&]() { for (;;) { const uint64_t k = next.fetch_add(1); if (k >= count) return; fn(k); } });
      for (auto &x : th) x.join();
    };
    auto lines_of = [&](const RawBitvector &bv) -> RankLines {
      const uint64_t nl = bv.n / 448 + 2;
      // payload word wq of line k = bits [448k + 64wq, +64) of the vector (unaligned gather from the 64-bit words); the line's
      // first word = ones before the line
      auto fill_lines = [&](uint64_t lo, uint64_t hi, uint64_t ones, uint64_t *dst) -> uint64_t {      // lines [lo, hi) -> dst; returns the ones they hold
        const uint64_t ones0 = ones;
        for (uint64_t k = lo; k < hi; ++k) {
          uint64_t *L = dst + (k - lo) * 8;
          L[0] = ones;
          for (int wq = 0; wq < 7; ++wq) {
            const uint64_t bit = k * 448 + (uint64_t)wq * 64;
            uint64_t w = 0;
            if (bit < bv.n) {
              const uint64_t wi = bit >> 6, sh = bit & 63;
              w = bv.bits[wi] >> sh;
              if (sh && wi + 1 < bv.bits.size()) w |= bv.bits[wi + 1] << (64 - sh);
              if (bit + 64 > bv.n) w &= (1ull << (bv.n - bit)) - 1;      // nothing beyond the last bit
            }
            L[1 + wq] = w;
            ones += (uint64_t)__builtin_popcountll(w);
          }
        }
        return ones - ones0;
      };
      uint64_t *d = (uint64_t *)temp_alloc(nl * 64);
      rb_allocs.push_back(d);
      const uint64_t nslices = (nl + kSliceLines - 1) / kSliceLines;
      if (nslices <= 1) {                                  // tiny: one buffer, one copy
        std::vector<uint64_t> L(nl * 8);
        fill_lines(0, nl, 0, L.data());
        HIP_CHECK(hipMemcpy(d, L.data(), L.size() * 8, hipMemcpyHostToDevice));
        return RankLines{d, bv.n};
      }
      // ones before every slice: popcount of the slice's bit range
      std::vector<uint64_t> before(nslices + 1, 0);
      run_parallel(nslices, [&](uint64_t sidx) {
        const uint64_t from = std::min(bv.n, sidx * kSliceLines * 448), to = std::min(bv.n, (sidx + 1) * kSliceLines * 448);
        uint64_t c = 0;
        if (to > from) {
          const uint64_t w0 = from >> 6, w1 = (to - 1) >> 6;
          for (uint64_t w = w0; w <= w1; ++w) {
            uint64_t x = bv.bits[w];
            if (w == w0 && (from & 63)) x &= ~0ull << (from & 63);
            if (w == w1 && ((to & 63) != 0)) x &= (1ull << (to & 63)) - 1;
            c += (uint64_t)__builtin_popcountll(x);
          }
        }
        before[sidx + 1] = c;
      });
      for (uint64_t k = 0; k < nslices; ++k) before[k + 1] += before[k];
      if (nslices <= kChunkSlices) {                       // up to 128 MB: laid out by the threads, one copy
        std::vector<uint64_t> L(nl * 8);
        run_parallel(nslices, [&](uint64_t sidx) {
          const uint64_t lo = sidx * kSliceLines, hi = std::min(nl, lo + kSliceLines);
          fill_lines(lo, hi, before[sidx], L.data() + lo * 8);
        });
        HIP_CHECK(hipMemcpy(d, L.data(), L.size() * 8, hipMemcpyHostToDevice));
        return RankLines{d, bv.n};
      }
      for (auto &g : stage) if (!g.p) {
        HIP_CHECK(hipHostMalloc((void **)&g.p, kChunkSlices * kSliceLines * 64, hipHostMallocDefault));
        HIP_CHECK(hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
      }
      for (uint64_t c0 = 0, chunk = 0; c0 < nslices; c0 += kChunkSlices, ++chunk) {
        Stage &g = stage[chunk & 1];
        if (g.busy) { HIP_CHECK(hipEventSynchronize(g.done)); g.busy = false; }
        const uint64_t c1 = std::min(nslices, c0 + kChunkSlices);
        run_parallel(c1 - c0, [&](uint64_t j) {
          const uint64_t sidx = c0 + j, lo = sidx * kSliceLines, hi = std::min(nl, lo + kSliceLines);
          fill_lines(lo, hi, before[sidx], g.p + j * kSliceLines * 8);
        });
        const uint64_t lo = c0 * kSliceLines, hi = std::min(nl, c1 * kSliceLines);
        HIP_CHECK(hipMemcpyAsync(d + lo * 8, g.p, (hi - lo) * 64, hipMemcpyHostToDevice, stream_));
        HIP_CHECK(hipEventRecord(g.done, stream_));
        g.busy = true;
      }
      return RankLines{d, bv.n};
    };
    struct StageGuard { std::function<void()> f; ~StageGuard() { f(); } } stage_guard{release_stages};
    view_.rb.use = lines_of(h.use_run_block);
    for (int k = 0; k < 3; ++k) {
      view_.rb.plain[k] = h.wavelet_seq.node_cnt ? lines_of(h.wavelet_seq.node[k]) : RankLines{nullptr, 0};
      view_.rb.runs[k] = h.run_block_seq.node_cnt ? lines_of(h.run_block_seq.node[k]) : RankLines{nullptr, 0};
    }
    // children order: node 1 = codes 0x, node 2 = codes 1x (checked by the parser)
    if (h.wavelet_seq.node_cnt && h.wavelet_seq.children[0][0] == 2) std::swap(view_.rb.plain[1], view_.rb.plain[2]);
    if (h.run_block_seq.node_cnt && h.run_block_seq.children[0][0] == 2) std::swap(view_.rb.runs[1], view_.rb.runs[2]);
    view_.rb.b = h.b;
    view_.rb.block_cnt = h.block_cnt;
    view_.rb.filter_rate = (uint32_t)h.selected_filter_rate;
    HIP_CHECK(hipStreamSynchronize(stream_));
    release_stages();
    lap("rank lines (host) + upload");
    if (layout_rb) {
      for (void *q : rb_allocs) { temps_.erase(std::find(temps_.begin(), temps_.end(), q)); owned_.push_back(q); }
      if (!h.selected_rows.empty()) {
        std::vector<uint64_t> filt(((h.n + view_.rb.filter_rate - 1) / view_.rb.filter_rate + 63) / 64 + 1, 0);
        for (uint64_t r : h.selected_rows) { const uint64_t fb = r / view_.rb.filter_rate; filt[fb >> 6] |= 1ull << (fb & 63); }
        view_.rb.sel_filter = upload(filt);
      }
      view_.rb.enabled = 1;
      search_v1_ = true;                 // the state-machine kernel reads the flat occ records directly
    } else {
      // ---- occ records: 64 B per 128 symbols (layout in cfr_device.hpp), expanded from the image above
      const uint64_t nrec = (h.n >> 7) + 2, halves = nrec * 2;
      d_occ = dev_alloc<uint64_t>(nrec * 8);
      uint64_t *d_cnt = (uint64_t *)temp_alloc(3 * (halves + 1) * 8), *d_pre = (uint64_t *)temp_alloc(3 * (halves + 1) * 8);
      HIP_CHECK(hipMemsetAsync(d_cnt, 0, 3 * (halves + 1) * 8, stream_));
      const unsigned g = (unsigned)std::min<uint64_t>((halves + 255) / 256, 1u << 20);
      k_occ_expand<<<g, 256, 0, stream_>>>(view_.rb, h.n, halves, d_occ, d_cnt);
      HIP_CHECK(hipGetLastError());
      size_t tmp_bytes = 0;
      HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_cnt, d_pre, (int)(halves + 1), stream_));
      void *d_tmp = temp_alloc(tmp_bytes);
      for (int c = 0; c < 3; ++c)
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_cnt + c * (halves + 1), d_pre + c * (halves + 1), (int)(halves + 1), stream_));
      k_occ_fill_mid<<<g, 256, 0, stream_>>>(halves, d_pre, d_occ);
      HIP_CHECK(hipGetLastError());
      if (!h.selected_rows.empty()) {
        uint64_t *d_sel = (uint64_t *)temp_alloc(h.selected_rows.size() * 8);
        HIP_CHECK(hipMemcpyAsync(d_sel, h.selected_rows.data(), h.selected_rows.size() * 8, hipMemcpyHostToDevice, stream_));
        k_occ_flag_selected<<<(unsigned)((h.selected_rows.size() + 255) / 256), 256, 0, stream_>>>(d_sel, h.selected_rows.size(), d_occ, kSelFlag);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(stream_));
        temp_free(d_sel);
      }
      HIP_CHECK(hipStreamSynchronize(stream_));
      temp_free(d_tmp);
      temp_free(d_cnt);
      temp_free(d_pre);
      for (void *q : rb_allocs) temp_free(q);
      memset(&view_.rb, 0, sizeof(view_.rb));
      lap("occ expansion (device)");
    }
  }

  uint64_t *d_ftab = dev_alloc<uint64_t>(h.ftab.size());
  if (!h.ftab.empty()) HIP_CHECK(hipMemcpy(d_ftab, h.ftab.data(), h.ftab.size() * 8, hipMemcpyHostToDevice));
  uint64_t *d_sampled = dev_alloc<uint64_t>(h.sampled_words.size());
  HIP_CHECK(hipMemcpy(d_sampled, h.sampled_words.data(), h.sampled_words.size() * 8, hipMemcpyHostToDevice));
  uint64_t *d_rows = dev_alloc<uint64_t>(h.selected_rows.size());
  uint64_t *d_vals = dev_alloc<uint64_t>(h.selected_vals.size());
  if (!h.selected_rows.empty()) {
    HIP_CHECK(hipMemcpy(d_rows, h.selected_rows.data(), h.selected_rows.size() * 8, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_vals, h.selected_vals.data(), h.selected_vals.size() * 8, hipMemcpyHostToDevice));
  }

  view_.n = h.n;
  view_.first_isa = h.first_isa;
  view_.adjusted_sa0 = h.adjusted_sa0;
  memcpy(view_.C, h.C, sizeof(h.C));
  const bool protein = h.prot.enabled;
  view_.occ = d_occ;
  view_.ftab = d_ftab;
  view_.sampled = d_sampled;
  view_.sel_rows = d_rows;
  view_.sel_vals = d_vals;
  view_.sel_cnt = h.selected_rows.size();
  view_.last_code = h.last_code;
  view_.ftab_width = (uint32_t)h.precompute_width;
  view_.sampled_bits = (uint32_t)h.sampled_bits;
  view_.sample_rate = (uint32_t)h.sample_rate;
  view_.min_hit_len = h.params.min_hit_len;
  view_.score_adjust = h.score_hit_len_adjust;
  view_.max_result = h.params.max_result;
  view_.tax_parent = upload(h.tax.parent);
  view_.tax_orig = upload(h.tax.orig_taxid);
  view_.seq_to_tax = upload(h.tax.seq_to_tax);
  view_.tax_rank = upload(h.tax.rank);
  {
    const auto &par = h.tax.parent;
    std::vector<uint32_t> depth(par.size(), 0xffffffffu);
    std::vector<uint64_t> path;
    for (uint64_t x0 = 0; x0 < par.size(); ++x0) {
      path.clear();
      uint64_t x = x0;
      while (depth[x] == 0xffffffffu && par[x] != x && par[x] < par.size() && path.size() <= par.size()) { path.push_back(x); x = par[x]; }
      uint32_t d = depth[x] == 0xffffffffu ? 0u : depth[x];
      if (depth[x] == 0xffffffffu) depth[x] = 0;
      for (size_t k = path.size(); k-- > 0;) depth[path[k]] = ++d;
    }
    view_.tax_depth = upload(depth);
  }
  view_.node_cnt = h.tax.node_cnt;
  view_.seq_cnt = h.tax.seq_cnt;
  view_.tax_root = h.tax.root;
  view_.secondary_hit_len = h.params.consider_secondary_hit_len;
  view_.secondary_factor = h.params.consider_secondary_score_factor;
  memcpy(view_.rank_num, h.tax.rank_num, sizeof(view_.rank_num));
  // ---- derived tables (cfr_device.hpp).  What goes into HBM, in order of worth: the text-mode tables (suffix array + 2-bit
  // text: 4.25 bytes per row below 2^32 rows, 4.75 above), the derived K-mer table (16-byte entries, or 8-byte entries when
  // those do not fit beside the text-mode tables: 34 GB instead of 69 GB at K = 16), the locate memo (4 bytes per row; with
  // a suffix array every row is located through the step function anyway, so the memo only takes what is left over).
  view_.ftabx = nullptr;
  view_.ftabx_width = 0;
  view_.ftabx_e8 = 0;
  view_.sa32 = nullptr; view_.sa36 = nullptr; view_.text2 = nullptr; view_.text8 = nullptr; view_.text_min_l = 0;
  memset(&view_.steps, 0, sizeof(view_.steps));
  uint32_t log4n = 0;
  while (log4n < 31 && (1ull << (2 * (log4n + 1))) <= h.n) ++log4n;
  // (a protein index: u32 suffix array and one byte of text per symbol, below 2^32 symbols)
  // (test hook CFR_TEXT_LIMIT_LOG2: the size from which an image loads WITHOUT text mode - 36-bit suffix-array entries end at 2^36 rows -
  //  lowered, so that a test walks that branch on a small index)
  uint64_t text_limit = protein ? 0xfffffff0ull : (1ull << 36);
  if (const char *e = dbg_env("CFR_TEXT_LIMIT_LOG2")) text_limit = std::min<uint64_t>(text_limit, 1ull << std::min(36, std::max(6, atoi(e))));
  const bool text_possible = h.n >= 64 && h.n < text_limit && !layout_rb;
  bool text_want = text_possible && (opt.text_mode < 0 ? !fast_load : opt.text_mode != 0);
  if (const char *e = dbg_env("CFR_TEXT_MODE")) text_want = text_possible && atoi(e) != 0;
  const double sa_bytes = (double)h.n * (wide_ ? 4.5 : 4.0), text_tab_bytes = sa_bytes + (double)h.n * (protein ? 1.0 : 0.25);
  const double ruler_bytes = (double)(h.n >> kRulerShift) * 24.0;
  const double batch_reserve = 12e9;                       // buffers of a 10 M-read batch (raw hits, virtual rows, results)
  {
    size_t free_b = 0, total_b = 0;
    if (text_want && hipMemGetInfo(&free_b, &total_b) == hipSuccess && text_tab_bytes + ruler_bytes + batch_reserve > 0.97 * (double)free_b) text_want = false;   // no room
  }
  {
    // auto K: several K-mers per text position (an unmatched strand then ends inside the lookup; measured: 16 beats 15 on a
    // 1 Gbp index by 7 % of the search kernel), at least 2 characters wider than the on-disk ftab, at most 16 - or 17, below
    uint32_t K = std::min<uint32_t>(16, std::max<uint32_t>(view_.ftab_width + 2, log4n + 2));
    bool e8 = false;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const double rest = fast_load ? 0.0 : (text_want ? text_tab_bytes + ruler_bytes : (double)h.n * 4.0) + batch_reserve;
      auto fits = [&](uint32_t k, bool small) { return (double)((small ? 8ull : 16ull) << (2 * k)) + rest <= 0.97 * (double)free_b; };
      while (K > view_.ftab_width + 2 && !fits(K, false) && !fits(K, true)) --K;
      e8 = !fits(K, false);
      // K = 17 with 8-byte entries (137 GB; round 6) for indexes of 4^14 = 2.7e8 symbols and more, when it fits beside everything else: a
      // random 17-mer occurs a quarter as often as a 16-mer, and fewer chance matches are fewer suffix-array fetches, text steps, wide
      // ranges and (36-bit images) extends - it beats K = 16 WITH its 16-byte entries' text positions on every workload measured
      // (profiles/r6w_ab_k17.txt: 1 Gbp +4.5 %, pairs +3 %, long reads +10 %, 2.5 Gbp +12 %, 8 Gbp +15 % / long +24 %, strains +1 % / 0)
      static const bool k17_off = dbg_env("CFR_K17") && atoi(dbg_env("CFR_K17")) == 0;
      // (with room to spare: at 16 Gbp the 137 GB fit by the estimate above and left a 10 M-read batch no scratch - 287 GB of image, hipMalloc
      //  out of memory in the first call; the load's other tables are ~4.5 bytes per symbol more than `rest` counts.  K = 17 up to ~11 Gbp.)
      const bool room17 = (double)(8ull << 34) + rest + 4.5 * (double)h.n + 20e9 <= 0.97 * (double)free_b;
      if (!k17_off && !fast_load && K == 16 && log4n + 3 >= 17 && view_.ftab_width + 2 <= 17 && room17) { K = 17; e8 = true; }
    }
    if (fast_load || balanced) { K = std::min<uint32_t>(K, std::max<uint32_t>(view_.ftab_width + 2, 13)); e8 = false; }     // <= 1 GB
    if (opt.ftabx_width >= 0) K = (uint32_t)opt.ftabx_width;
    if (const char *e = dbg_env("CFR_FTABX_WIDTH")) K = (uint32_t)atoi(e);
    if (const char *e = dbg_env("CFR_FTABX_E8")) e8 = atoi(e) != 0;        // test hook: the 8-byte entries on a small index
    if (K > 17) K = 17;                  // (17 only by request - CFR_FTABX_WIDTH - and with 8-byte entries: 137 GB)
    if (K == 17) e8 = true;
    if (protein) K = 0;                  // derived K-mer table and text mode are nucleotide designs
    if (K > view_.ftab_width && view_.ftab_width > 0) try {
      const uint64_t entries = 1ull << (2 * K);
      uint64_t *d_tab = dev_alloc<uint64_t>(entries * (e8 ? 1 : 2) + 2);
      const unsigned gb = (unsigned)std::min<uint64_t>((entries + 255) / 256, 1u << 22);
      if (e8) k_build_ftabx<true><<<gb, 256, 0, stream_>>>(view_, K, d_tab);
      else k_build_ftabx<false><<<gb, 256, 0, stream_>>>(view_, K, d_tab);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(stream_));
      view_.ftabx = d_tab;
      view_.ftabx_width = K;
      view_.ftabx_e8 = e8 ? 1 : 0;
    } catch (const HipError &) { (void)hipGetLastError(); view_.ftabx = nullptr; view_.ftabx_width = 0; view_.ftabx_e8 = 0; }   // optional table: run without it
  }
  if (protein && view_.ftab_width > 0) {
    // derived K-mer table of a protein index: key in base sigma; K = log_sigma(n) + 1 (a random K-mer of a frame or strand that does
    // not match then ends inside the lookup), wider than the on-disk ftab, at most 7 (21^7 entries = 29 GB) and a quarter of the free HBM
    uint32_t K = 1;
    { double x = (double)h.n; while (x >= (double)h.prot.sigma && K < 12) { x /= (double)h.prot.sigma; ++K; } }
    K = std::min<uint32_t>(std::max<uint32_t>(K + 1, view_.ftab_width + 1), 7);
    auto entries_of = [&](uint32_t k) { uint64_t e = 1; for (uint32_t i = 0; i < k; ++i) e *= h.prot.sigma; return e; };
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) while (K > view_.ftab_width + 1 && entries_of(K) * 16 > free_b / 4) --K;
    if (fast_load) K = 0;
    if (opt.ftabx_width >= 0) K = std::min<uint32_t>((uint32_t)opt.ftabx_width, 7);
    if (const char *e = dbg_env("CFR_FTABX_WIDTH")) K = std::min<uint32_t>((uint32_t)atoi(e), 7);
    if (K > view_.ftab_width) try {
      const uint64_t entries = entries_of(K);
      uint64_t *d_tab = dev_alloc<uint64_t>(entries * 2 + 2);
      k_build_ftabx_prot<<<(unsigned)std::min<uint64_t>((entries + 255) / 256, 1u << 22), 256, 0, stream_>>>(view_, K, entries, d_tab);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(stream_));
      view_.ftabx = d_tab;
      view_.ftabx_width = K;
    } catch (const HipError &) { (void)hipGetLastError(); view_.ftabx = nullptr; view_.ftabx_width = 0; }
  }
  lap("side tables + ftabx");
  // derived text-mode tables: SA / 2-bit text by list ranking, then the step function of the locate values
  if (text_want) {
    uint32_t *d_sa = nullptr;
    uint64_t *d_text_alloc = nullptr;
    std::vector<void *> tmp_here;
    auto talloc = [&](size_t bytes) { void *q = temp_alloc(bytes); tmp_here.push_back(q); return q; };
    try {
      // list ranking by rulers (cfr_kernels.hip.inc): one ruler every 2^kRulerShift rows + the two ends of the list
      const uint64_t nrulers = ((h.n - 1) >> kRulerShift) + 1, cnt = nrulers + 2;      // + terminal + head of the list
      uint32_t *nx_a = (uint32_t *)talloc(cnt * 4), *nx_b = (uint32_t *)talloc(cnt * 4);
      uint64_t *ds_a = (uint64_t *)talloc(cnt * 8), *ds_b = (uint64_t *)talloc(cnt * 8);
      const unsigned gr = (unsigned)std::min<uint64_t>((cnt + 255) / 256, (uint64_t)num_cus_ * 32);
      k_ruler_walk<<<gr, 256, 0, stream_>>>(view_, nrulers, nx_a, ds_a);
      HIP_CHECK(hipGetLastError());
      for (uint64_t span = 1; span < cnt; span <<= 1) {
        k_ruler_jump<<<gr, 256, 0, stream_>>>(cnt, nx_a, ds_a, nx_b, ds_b);
        std::swap(nx_a, nx_b);
        std::swap(ds_a, ds_b);
      }
      HIP_CHECK(hipGetLastError());
      const size_t sa_alloc = (wide_ ? (size_t)((h.n * 36 + 7) / 8) : (size_t)h.n * 4) + 256;   // + pad: a wide range reads 16 entries past its first row
      {
        void *q = nullptr;
        HIP_CHECK(hipMalloc(&q, sa_alloc));
        d_sa = (uint32_t *)q;
      }
      const uint64_t twords = protein ? (h.n + 7) / 8 + 8 : (h.n + 31) / 32 + 6;       // protein: a byte per symbol
      {
        void *q = nullptr;
        HIP_CHECK(hipMalloc(&q, twords * 8));
        d_text_alloc = (uint64_t *)q;
      }
      uint64_t *d_text = d_text_alloc + 2;                                                     // 16 bytes of padding in front (see k_search_chains_v2: text_slot)
      HIP_CHECK(hipMemsetAsync(d_text_alloc, 0, twords * 8, stream_));
      if (wide_) HIP_CHECK(hipMemsetAsync(d_sa, 0, sa_alloc, stream_));                         // 36-bit entries are or-ed into place
      else HIP_CHECK(hipMemsetAsync((char *)d_sa + (size_t)h.n * 4, 0, 256, stream_));
      if (wide_) k_ruler_fill<true><<<gr, 256, 0, stream_>>>(view_, nrulers, ds_a, d_sa, (unsigned long long *)d_text);
      else k_ruler_fill<false><<<gr, 256, 0, stream_>>>(view_, nrulers, ds_a, d_sa, (unsigned long long *)d_text);      // (protein: bytes from d_text_alloc + 16 on)
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(stream_));
      for (void *q : tmp_here) temp_free(q);
      tmp_here.clear();
      if (wide_) view_.sa36 = d_sa; else view_.sa32 = d_sa;
      lap("SA / text (list ranking)");
      // ---- the step function: breakpoints = position 0 and the positions of the rows that store a value of their own - the
      // selectedSA rows (nucleotide index: one per genome boundary) or the end-marker rows (protein index: the row of every '$',
      // FMIndex.hpp:224-228) - with runs of equal values merged
      std::vector<std::pair<uint64_t, uint64_t>> bp;
      if (protein) {
        // every stop of the walk (k_collect_stops_prot), sorted by position on the device
        const uint64_t nsamp_p = (h.n + h.sample_rate - 1) / h.sample_rate, total = nsamp_p + h.prot.end_marker_n;
        if (total >= (1ull << 31)) throw HipError{"protein index too large for the stop sort", -5};       // (hipCUB takes an int count)
        uint64_t *d_pos = (uint64_t *)talloc(total * 8), *d_val = (uint64_t *)talloc(total * 8);
        uint64_t *d_pos2 = (uint64_t *)talloc(total * 8), *d_val2 = (uint64_t *)talloc(total * 8);
        k_collect_stops_prot<<<(unsigned)std::min<uint64_t>((total + 255) / 256, 1u << 20), 256, 0, stream_>>>(view_, nsamp_p, d_pos, d_val);
        HIP_CHECK(hipGetLastError());
        size_t sort_bytes = 0;
        HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, d_pos, d_pos2, d_val, d_val2, (int)total, 0, 64, stream_));
        void *d_sort = talloc(sort_bytes);
        HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(d_sort, sort_bytes, d_pos, d_pos2, d_val, d_val2, (int)total, 0, 64, stream_));
        std::vector<uint64_t> hp(total), hv(total);
        HIP_CHECK(hipMemcpyAsync(hp.data(), d_pos2, total * 8, hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipMemcpyAsync(hv.data(), d_val2, total * 8, hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
        if (hp.empty() || hp[0] != 0) bp.emplace_back(0, h.adjusted_sa0);       // (position 0 is the row firstISA: listed when that row is a sampled one)
        for (uint64_t k = 0; k < total && hp[k] != ~0ull; ++k) bp.emplace_back(hp[k], hv[k]);
      } else {
      std::vector<uint64_t> brk_rows = h.selected_rows, brk_vals = h.selected_vals;
      const uint64_t nsel = brk_rows.size();
      std::vector<uint64_t> bpos(nsel + 1, 0);
      if (nsel) {
        uint64_t *d_r = (uint64_t *)talloc(nsel * 8), *d_p = (uint64_t *)talloc(nsel * 8);
        HIP_CHECK(hipMemcpyAsync(d_r, brk_rows.data(), nsel * 8, hipMemcpyHostToDevice, stream_));
        k_gather_sa<<<grid_for(nsel), kBlock, 0, stream_>>>(view_, d_r, nsel, d_p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(bpos.data() + 1, d_p, nsel * 8, hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
      }
      bp.emplace_back(0, h.adjusted_sa0);
      for (uint64_t g = 0; g < nsel; ++g) if (brk_rows[g] != h.first_isa) bp.emplace_back(bpos[g + 1], brk_vals[g]);
      }
      std::sort(bp.begin(), bp.end());
      std::vector<uint64_t> spos, sval;
      for (const auto &e : bp) if (spos.empty() || e.second != sval.back()) { spos.push_back(e.first); sval.push_back(e.second); }
      uint32_t shift = 0;                                  // ~4 buckets per breakpoint, 2^10 .. 2^22 buckets
      {
        const uint64_t want_b = std::min<uint64_t>(std::max<uint64_t>(4 * spos.size(), 1ull << 10), 1ull << 22);
        while (shift < 40 && ((h.n - 1) >> shift) + 1 > want_b) ++shift;
      }
      const uint64_t nb = ((h.n - 1) >> shift) + 2;
      std::vector<uint32_t> bucket(nb + 1, 0);
      {
        uint64_t j = 0;
        for (uint64_t bkt = 0; bkt <= nb; ++bkt) {         // last breakpoint with pos <= bkt << shift
          const uint64_t start = bkt << shift;
          while (j + 1 < spos.size() && spos[j + 1] <= start) ++j;
          bucket[bkt] = (uint32_t)j;
        }
      }
      StepView S;
      S.pos = upload(spos); S.val = upload(sval); S.bucket = upload(bucket);
      S.cnt = spos.size(); S.n = h.n; S.shift = shift;
      view_.steps = S;
      unsigned long long *d_bad = (unsigned long long *)talloc(8), bad = 0;
      HIP_CHECK(hipMemsetAsync(d_bad, 0, 8, stream_));
      const uint64_t nsamp = (h.n + h.sample_rate - 1) / h.sample_rate;
      k_memo_check<<<(unsigned)std::min<uint64_t>((nsamp + 255) / 256, 1u << 20), 256, 0, stream_>>>(view_, nsamp, d_bad);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, stream_));
      HIP_CHECK(hipStreamSynchronize(stream_));
      for (void *q : tmp_here) temp_free(q);
      tmp_here.clear();
      if (bad || (dbg_env("CFR_STEPS_OFF") && atoi(dbg_env("CFR_STEPS_OFF")))) throw HipError{"the sampled rows do not follow the step function of the selectedSA positions", -5};
      owned_.push_back(d_sa); device_bytes_ += sa_alloc;
      owned_.push_back(d_text_alloc); device_bytes_ += twords * 8;
      if (protein) view_.text8 = reinterpret_cast<const uint8_t *>(d_text_alloc) + 16;
      else view_.text2 = d_text;
      view_.text_min_l = log4n + 2;                       // random matches rarely get past log4(n) characters
      if (protein) {                                      // ... log_sigma(n) for a protein index
        uint32_t ls = 0;
        for (double x = (double)h.n; x >= (double)h.prot.sigma; x /= (double)h.prot.sigma) ++ls;
        view_.text_min_l = ls + 2;
      }
      if (const char *e = dbg_env("CFR_TEXT_MIN_L")) view_.text_min_l = (uint32_t)atoi(e);
      lap("locate step function");
      // K-mer entries of one row carry the row's text position (k_ftabx_textpos): the search then skips its suffix-array fetch.
      // Only where such a search would move to the text right behind the table (K >= text_min_l) and the entries have the room (16 bytes).
      static const bool tp_off = dbg_env("CFR_FTABX_TEXTPOS") && atoi(dbg_env("CFR_FTABX_TEXTPOS")) == 0;
      if (!protein && !tp_off && view_.ftabx && !view_.ftabx_e8 && view_.ftabx_width >= view_.text_min_l && (view_.sa32 || view_.sa36)) {
        const uint64_t entries = 1ull << (2 * view_.ftabx_width);
        k_ftabx_textpos<<<(unsigned)std::min<uint64_t>((entries + 255) / 256, 1u << 22), 256, 0, stream_>>>(view_, view_.ftabx_width, const_cast<uint64_t *>(view_.ftabx));
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(stream_));
        lap("text positions into the K-mer table");
      }
    } catch (const HipError &) {                           // optional tables: run without them, and give back what they took
      (void)hipGetLastError();
      (void)hipStreamSynchronize(stream_);
      for (void *q : tmp_here) temp_free(q);
      if (d_sa) (void)hipFree(d_sa);
      if (d_text_alloc) (void)hipFree(d_text_alloc);
      view_.sa32 = nullptr; view_.sa36 = nullptr; view_.text2 = nullptr; view_.text8 = nullptr; view_.text_min_l = 0;
      memset(&view_.steps, 0, sizeof(view_.steps));       // (the three small step arrays stay owned until the image goes)
    }
  }
  // derived locate memo (cfr_device.hpp).  With a suffix array it is a convenience (one gather instead of a gather plus a short
  // search in cached tables) and only takes memory nobody else wants; without one (protein, fast-load, no room for the text-mode
  // tables) it is what keeps the locate walks short: densest power-of-two rate whose table fits the budget (0 = off).
  view_.loc_memo = nullptr;
  view_.memo_shift = 0;
  {
    const bool have_sa_now = have_sa();
    double budget_gb = opt.loc_memo_gb;
    if (budget_gb < 0) {
      budget_gb = fast_load ? 0.0 : 16.0;
      size_t free_b = 0, total_b = 0;
      // everything still free but the batch buffers
      if (!fast_load && hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget_gb = std::max(have_sa_now ? 0.0 : budget_gb, ((double)free_b - 16e9) / 1e9);
      if (have_sa_now && (double)h.n * 4.0 > budget_gb * 1e9) budget_gb = 0;      // all rows or nothing: the step function serves the rest
    }
    if (const char *e = dbg_env("CFR_LOC_MEMO_GB")) budget_gb = atof(e);
    uint64_t max_val = h.adjusted_sa0;
    for (uint64_t x : h.selected_vals) max_val = std::max(max_val, x);
    if (protein && h.prot.end_marker_bits > 32) max_val = ~0ull;
    const bool fits32 = h.sampled_bits <= 32 && max_val <= 0xffffffffull;
    uint32_t shift = 0;
    while (shift < 8 && (double)((h.n >> shift) + 1) * 4.0 > budget_gb * 1e9) ++shift;
    if (budget_gb > 0 && fits32 && (1u << shift) < (uint32_t)h.sample_rate) try {
      const uint64_t entries = ((h.n - 1) >> shift) + 1;
      uint32_t *d_memo = dev_alloc<uint32_t>(entries);
      const unsigned gm = (unsigned)std::min<uint64_t>((entries + 255) / 256, 1u << 20);
      if (have_sa_now && view_.steps.pos && !(dbg_env("CFR_MEMO_WALK") && atoi(dbg_env("CFR_MEMO_WALK"))))
        k_memo_fill<<<gm, 256, 0, stream_>>>(view_, shift, entries, d_memo);      // from the text order: one SA read and a short search per row
      else
        k_build_loc_memo<<<gm, 256, 0, stream_>>>(view_, shift, entries, d_memo); // the plain LF walk per row
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(stream_));
      view_.loc_memo = d_memo;
      view_.memo_shift = shift;
    } catch (const HipError &) { (void)hipGetLastError(); view_.loc_memo = nullptr; view_.memo_shift = 0; }   // optional table
  }
  lap("locate memo");
  {
    // what a sub-batch's buffers may take: 40 % of the HBM still free now that the image stands, at most 32 GB (cut_pieces)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const double budget = std::min(32e9, 0.4 * (double)free_b);
      piece_bases_max_ = (uint64_t)std::max(2.5e8, budget / 2.7);
    }
  }
  view_.max_entries = (uint64_t)(int64_t)(h.params.max_result * h.params.max_result_per_hit_factor);   // int*int -> size_t (Classifier.hpp:620)
  view_.locate_all = (h.params.max_result_per_hit_factor <= 0 || h.params.max_result <= 0) ? 1 : 0;
  // the finished description once more in device memory, for the kernels that take it by pointer (k_adjust_tail_p)
  d_view_ = dev_alloc<DevView>(1);
  HIP_CHECK(hipMemcpy(d_view_, &view_, sizeof(DevView), hipMemcpyHostToDevice));
}

DeviceIndex::~DeviceIndex() { release(); }

void DeviceIndex::release() {            // idempotent: also the clean-up of a constructor that failed half way
  (void)hipSetDevice(device_);
  (void)hipDeviceSynchronize();
  for (void *p : owned_) (void)hipFree(p);
  owned_.clear();
  for (void *p : temps_) (void)hipFree(p);
  temps_.clear();
  for (auto &s : slots_) if (s.p) (void)hipFree(s.p);
  slots_.clear();
  if (pinned_) (void)hipHostFree(pinned_);
  pinned_ = nullptr; pinned_cap_ = 0;
  auto drop_event = [](hipEvent_t &e) { if (e) (void)hipEventDestroy(e); e = nullptr; };
  for (auto &set : evs_) for (auto &e : set) drop_event(e);
  for (auto &e : tail_done_) drop_event(e);
  for (auto &e : copy_done_) drop_event(e);
  for (auto &e : h2d_done_) drop_event(e);
  for (auto &e : copied_) drop_event(e);
  for (auto &e : search_done_) drop_event(e);
  auto drop_stream = [](hipStream_t &s) { if (s) (void)hipStreamDestroy(s); s = nullptr; };
  drop_stream(tail_stream_);
  drop_stream(search2_stream_);
  if (prep_done_) { (void)hipEventDestroy(prep_done_); prep_done_ = nullptr; }
  drop_stream(h2d_stream_);
  drop_stream(dust_stream_);
  drop_stream(copy_stream_);
  drop_stream(stream_);
}

// load-time temporaries: registered in temps_ so that a throw anywhere in init() frees them (release()), freed early by temp_free
void *DeviceIndex::temp_alloc(size_t bytes) {
  void *p = nullptr;
  HIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 16)));
  temps_.push_back(p);
  return p;
}
void DeviceIndex::temp_free(void *p) {
  if (!p) return;
  auto it = std::find(temps_.begin(), temps_.end(), p);
  if (it != temps_.end()) temps_.erase(it);
  (void)hipFree(p);
}

// ------------------------------------------------------------------------------------ probes
void DeviceIndex::rank_batch(const char *chars, const uint64_t *pos, const uint8_t *incl, size_t n, uint64_t *out_rank, char *out_access) {
  HIP_CHECK(hipSetDevice(device_));
  if (n == 0) return;
  char *d_c = (char *)scratch(S_P0, n);
  uint64_t *d_p = (uint64_t *)scratch(S_P1, n * 8);
  uint8_t *d_i = (uint8_t *)scratch(S_P2, n);
  uint64_t *d_r = (uint64_t *)scratch(S_P3, n * 8);
  char *d_a = (char *)scratch(S_P4, n);
  HIP_CHECK(hipMemcpyAsync(d_c, chars, n, hipMemcpyHostToDevice, stream_));
  HIP_CHECK(hipMemcpyAsync(d_p, pos, n * 8, hipMemcpyHostToDevice, stream_));
  HIP_CHECK(hipMemcpyAsync(d_i, incl, n, hipMemcpyHostToDevice, stream_));
  k_rank_probe<<<grid_for(n), kBlock, 0, stream_>>>(view_, d_c, d_p, d_i, n, d_r, d_a);
  HIP_CHECK(hipGetLastError());
  if (out_rank) HIP_CHECK(hipMemcpyAsync(out_rank, d_r, n * 8, hipMemcpyDeviceToHost, stream_));
  if (out_access) HIP_CHECK(hipMemcpyAsync(out_access, d_a, n, hipMemcpyDeviceToHost, stream_));
  HIP_CHECK(hipStreamSynchronize(stream_));
}

void DeviceIndex::backward_search_batch(const uint8_t *bases, const uint64_t *offsets, const uint32_t *m, size_t n,
                                        uint64_t *out_l, uint64_t *out_sp, uint64_t *out_ep) {
  HIP_CHECK(hipSetDevice(device_));
  if (n == 0) return;
  const uint64_t total = offsets[n];
  uint8_t *d_b = (uint8_t *)scratch(S_IN_B1, total + 16);
  uint64_t *d_o = (uint64_t *)scratch(S_IN_O1, (n + 1) * 8);
  uint32_t *d_m = (uint32_t *)scratch(S_P0, n * 4);
  uint64_t *d_l = (uint64_t *)scratch(S_P1, n * 8), *d_sp = (uint64_t *)scratch(S_P2, n * 8), *d_ep = (uint64_t *)scratch(S_P3, n * 8);
  if (total) HIP_CHECK(hipMemcpyAsync(d_b, bases, total, hipMemcpyHostToDevice, stream_));
  HIP_CHECK(hipMemcpyAsync(d_o, offsets, (n + 1) * 8, hipMemcpyHostToDevice, stream_));
  HIP_CHECK(hipMemcpyAsync(d_m, m, n * 4, hipMemcpyHostToDevice, stream_));
  k_bs_probe<<<grid_for(n), kBlock, 0, stream_>>>(view_, d_b, d_o, d_m, n, d_l, d_sp, d_ep);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(out_l, d_l, n * 8, hipMemcpyDeviceToHost, stream_));
  HIP_CHECK(hipMemcpyAsync(out_sp, d_sp, n * 8, hipMemcpyDeviceToHost, stream_));
  HIP_CHECK(hipMemcpyAsync(out_ep, d_ep, n * 8, hipMemcpyDeviceToHost, stream_));
  HIP_CHECK(hipStreamSynchronize(stream_));
}

void DeviceIndex::locate_rows(const uint64_t *rows, size_t n, uint64_t *out_val, uint32_t *out_steps) {
  HIP_CHECK(hipSetDevice(device_));
  if (n == 0) return;
  uint64_t *d_r = (uint64_t *)scratch(S_P0, n * 8), *d_v = (uint64_t *)scratch(S_P1, n * 8);
  uint32_t *d_s = (uint32_t *)scratch(S_P2, n * 4);
  HIP_CHECK(hipMemcpyAsync(d_r, rows, n * 8, hipMemcpyHostToDevice, stream_));
  k_locate<<<grid_for(n), kBlock, 0, stream_>>>(view_, d_r, n, d_v, d_s);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(out_val, d_v, n * 8, hipMemcpyDeviceToHost, stream_));
  if (out_steps) HIP_CHECK(hipMemcpyAsync(out_steps, d_s, n * 4, hipMemcpyDeviceToHost, stream_));
  HIP_CHECK(hipStreamSynchronize(stream_));
}

void DeviceIndex::selfcheck(uint64_t out[6]) {
  HIP_CHECK(hipSetDevice(device_));
  unsigned long long *d_bad = (unsigned long long *)scratch(S_P0, 4 * 8), h_bad[4];
  HIP_CHECK(hipMemsetAsync(d_bad, 0, 4 * 8, stream_));
  const unsigned g = (unsigned)std::min<uint64_t>((view_.n + 255) / 256, (uint64_t)num_cus_ * 32);
  k_selfcheck<<<g, 256, 0, stream_>>>(view_, d_bad);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(h_bad, d_bad, 4 * 8, hipMemcpyDeviceToHost, stream_));
  HIP_CHECK(hipStreamSynchronize(stream_));
  for (int k = 0; k < 4; ++k) out[k] = h_bad[k];
  out[4] = have_sa() ? 1 : 0;                                                    // text-mode tables present
  out[5] = view_.loc_memo ? 1 + view_.memo_shift : 0;                            // locate memo present (1 + log2 of its row rate)
}

// ------------------------------------------------------------------------------------ the path
namespace {
void exclusive_scan(void *tmp, size_t tmp_bytes, const uint64_t *in, uint64_t *out, size_t count, hipStream_t st) {
  // out has count+1 entries; out[count] = total (input padded by the caller with one trailing zero)
  size_t need = tmp_bytes;
  HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, need, in, out, (int)(count + 1), st));
}
size_t scan_tmp_bytes(size_t count) {
  size_t need = 0;
  HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, (const uint64_t *)nullptr, (uint64_t *)nullptr, (int)(count + 1)));
  return need;
}
}  // namespace

DeviceIndex::Staged DeviceIndex::stage_inputs(const uint8_t *b1, const uint64_t *o1, const uint8_t *b2, const uint64_t *o2, size_t n) {
  Staged st{nullptr, nullptr, nullptr, nullptr, 0, 0};
  if (n == 0) return st;
  st.t1 = o1[n];
  st.t2 = b2 ? o2[n] : 0;
  uint8_t *d_b1 = (uint8_t *)scratch(S_IN_B1, st.t1 + 16);
  uint64_t *d_o1 = (uint64_t *)scratch(S_IN_O1, (n + 1) * 8);
  if (st.t1) HIP_CHECK(hipMemcpyAsync(d_b1, b1, st.t1, hipMemcpyHostToDevice, stream_));
  HIP_CHECK(hipMemcpyAsync(d_o1, o1, (n + 1) * 8, hipMemcpyHostToDevice, stream_));
  st.b1 = d_b1;
  st.o1 = d_o1;
  if (b2) {
    uint8_t *d_b2 = (uint8_t *)scratch(S_IN_B2, st.t2 + 16);
    uint64_t *d_o2 = (uint64_t *)scratch(S_IN_O2, (n + 1) * 8);
    if (st.t2) HIP_CHECK(hipMemcpyAsync(d_b2, b2, st.t2, hipMemcpyHostToDevice, stream_));
    HIP_CHECK(hipMemcpyAsync(d_o2, o2, (n + 1) * 8, hipMemcpyHostToDevice, stream_));
    st.b2 = d_b2;
    st.o2 = d_o2;
  }
  return st;
}

// search -> adjust/select -> compact -> (enumerate rows -> locate).  Two small D2H syncs (hit and row totals)
// size the dense arrays; everything else stays on the device.  Two halves: launch_search (caps, scan, the search kernel)
// is also what the one-launch post stage of classify_device follows; launch_post is everything behind the search kernel.
DeviceIndex::SearchBuf DeviceIndex::launch_search(const uint8_t *d_b1, const uint64_t *d_o1, const uint8_t *d_b2, const uint64_t *d_o2, size_t n,
                                                  uint64_t total1, uint64_t total2, int par, bool row_space_only) {
  const bool paired = d_b2 != nullptr;
  const int cpr = paired ? 4 : 2;
  hipStream_t sst = search_stream_ ? search_stream_ : stream_;      // (the stream this search is enqueued on: see two_search in classify_device)
  const size_t nchains = n * (size_t)cpr;
  // capacity bound without a host round trip: sum over chains of (len/(mhl+1) + 1)
  const uint64_t mhl1 = (uint64_t)view_.min_hit_len + 1;
  const uint64_t cap_total = 2 * (total1 / mhl1 + n) + (paired ? 2 * (total2 / mhl1 + n) : 0);

  if (view_.prot.enabled) return launch_search_protein(d_b1, d_o1, d_b2, d_o2, n, total1, total2, row_space_only);
  // two sets of output buffers: the post stage of sub-batch k (its own stream) reads one while the search of k + 1 fills the other
  uint64_t *cap = (uint64_t *)scratch(par ? S_CAP1 : S_CAP, (n + 1) * 8);
  uint64_t *hit_off = (uint64_t *)scratch(par ? S_HITOFF1 : S_HITOFF, (n + 1) * 8);
  cfr_hit *raw = (cfr_hit *)scratch(par ? S_RAW1 : S_RAW, cap_total * sizeof(cfr_hit));
  uint32_t *chain_cnt = (uint32_t *)scratch(par ? S_CHAINCNT1 : S_CHAINCNT, nchains * 4);
  size_t tmp_bytes = scan_tmp_bytes(n);
  void *tmp = scratch(par ? S_SCAN1 : S_SCAN, tmp_bytes);
  // text-space hits (k_search_chains_v2, virt_text_pos): searches that finish on the text keep virtual rows
  const bool text_hits = !search_v1_ && !row_space_only && have_sa() && view_.steps.pos && view_.text2;

  HIP_CHECK(hipEventRecord(ev_[0], sst));
  if (pre_hit_off_) {                    // the offsets exist (one pass over the whole batch): the sub-batch's lists start at pre_hit_base_
    hit_off = const_cast<uint64_t *>(pre_hit_off_);
    raw = reinterpret_cast<cfr_hit *>(reinterpret_cast<uintptr_t>(raw) - (uintptr_t)pre_hit_base_ * sizeof(cfr_hit));
  } else {
    HIP_CHECK(hipMemsetAsync(cap + n, 0, 8, sst));
    k_caps<<<grid_for(n), kBlock, 0, sst>>>(view_, d_o1, d_o2, n, cap);
    exclusive_scan(tmp, tmp_bytes, cap, hit_off, n, sst);
  }
  HIP_CHECK(hipEventRecord(ev_[1], sst));
  if (search_v1_) {
    if (paired) k_search_chains<4><<<grid_for(nchains), kBlock, 0, sst>>>(view_, d_b1, d_o1, d_b2, d_o2, n, hit_off, raw, chain_cnt);
    else k_search_chains<2><<<grid_for(nchains), kBlock, 0, sst>>>(view_, d_b1, d_o1, nullptr, nullptr, n, hit_off, raw, chain_cnt);
  } else {
    // persistent grid: every lane walks chains gid, gid + T, ... (T = resident lanes), see k_search_chains_v2
    // resident blocks only: a block that had to wait for a slot would start its share of the chains late
    int resident = blocks_per_cu_;
    if (overlap_now_ && !blocks_forced_) resident = std::min(resident, 4);      // leave the post stage of the previous sub-batch its wave slots
    {
      int occ = 0;
      const bool wide_k = wide_;
      const hipError_t e = wide_k ? (paired ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_search_chains_v2<4, false, true>, kBlock, 0)
                                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_search_chains_v2<2, false, true>, kBlock, 0))
                                  : (paired ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_search_chains_v2<4, false>, kBlock, 0)
                                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_search_chains_v2<2, false>, kBlock, 0));
      if (e == hipSuccess && occ > 0) resident = std::min(resident, occ);
      else (void)hipGetLastError();
    }
    unsigned blocks = std::min<unsigned>(grid_for(nchains), (unsigned)(num_cus_ * resident));
    SearchView sv;
    sv.n = view_.n; sv.first_isa = view_.first_isa;
    for (int c = 0; c < 4; ++c) sv.C[c] = view_.C[c];
    sv.occ = view_.occ; sv.ftab = view_.ftab; sv.ftabx = view_.ftabx; sv.text2 = view_.text2;
    sv.sa = text_hits ? (wide_ ? view_.sa36 : view_.sa32) : nullptr;
    sv.ftabx_e8 = view_.ftabx_e8;
    const bool wide = wide_;
    sv.last_code = view_.last_code; sv.ftab_width = view_.ftab_width; sv.ftabx_width = view_.ftabx_width;
    sv.text_min_l = view_.text_min_l; sv.min_hit_len = view_.min_hit_len;
    sv.wide_rows = wide_ ? kWideRowsWide : kWideRows;
    if (const char *e = dbg_env("CFR_WIDE_ROWS")) sv.wide_rows = std::min<uint32_t>(sv.wide_rows, (uint32_t)atoi(e));
    // the search reads the buffers through their packed form (k_pack_reads); callers of this function pack first
    // few chains per lane (long reads): chains are handed out dynamically (DYN), one atomic per chain
    bool dyn = nchains < 8ull * blocks * kBlock && (total1 + total2) / std::max<size_t>(1, nchains) >= 500;    // (per chain: half the mean read)
    // short reads by WAVE TILES (round 5: a wave draws 64 chains with one atomic and its lanes share them by ballot, k_search_chains_v2):
    // pairs 24.6 -> 23.9 ms per 10 M pairs (the four chains of a pair are of very different lengths: the static hand-out ends with the
    // lanes that drew the long ones), 20 strains 28.4 -> 27.6 ms, cfg2 unchanged (11.85 ms either way: it keeps the static hand-out);
    // tiles of 512 chains LOSE (cfg2 search 9.5 -> 12.5 ms: the launch ends with the waves that still hold a tile).
    // profiles/r5_ab_post_and_tiles.txt.  CFR_SEARCH_DYN=0 / 1 / 2: static / per-lane draws / wave tiles.
    int dyn_mode = -1;
    const bool short_reads = (total1 + total2) / std::max<size_t>(1, nchains) < 500;
    if (!dyn && short_reads && nchains > 4ull * blocks * kBlock && nchains < 0xfff00000ull && (paired || heavy_frac_ > 0.2 || two_search_now_)) { dyn = true; dyn_mode = 2; }
    if (const char *e = dbg_env("CFR_SEARCH_DYN")) { dyn_mode = atoi(e); dyn = dyn_mode != 0; }
    // short reads handed out dynamically (so that the post stage of the previous sub-batch can run beside this search: a block that
    // starts late then simply takes fewer chains): eight chains per draw; long reads: one
    // CFR_SEARCH_DYN=2 (round 5): wave tiles - a wave draws 512 chains (CFR_SEARCH_TILE) with one atomic and hands them to its lanes by ballot
    uint32_t dyn_chunk = dyn && (total1 + total2) / std::max<size_t>(1, nchains) < 500 ? 8u : 1u;
    if (dyn && dyn_mode == 2 && nchains < 0xfff00000ull) dyn_chunk = dbg_env("CFR_SEARCH_TILE") ? (uint32_t)std::max(64, atoi(dbg_env("CFR_SEARCH_TILE"))) : 64u;
    if (dyn) {
      // the instantiations that draw their chains need a few registers more (single-end narrow: 97 against 93 - four blocks per CU
      // instead of five): the grid is sized by the occupancy of the kernel that is launched
      int occ = 0;
      const hipError_t e = wide_ ? (paired ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_search_chains_v2<4, false, true, true>, kBlock, 0)
                                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_search_chains_v2<2, false, true, true>, kBlock, 0))
                                 : (paired ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_search_chains_v2<4, false, false, true>, kBlock, 0)
                                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_search_chains_v2<2, false, false, true>, kBlock, 0));
      if (e == hipSuccess && occ > 0) blocks = std::min<unsigned>(blocks, (unsigned)(num_cus_ * occ));
      else (void)hipGetLastError();
    }
    unsigned long long *d_ctr = nullptr;
    if (dyn) {
      d_ctr = (unsigned long long *)scratch(S_P5, 16 * 8) + (par ? 14 : 15);          // (a counter per set of outputs: two searches may be in flight)
      HIP_CHECK(hipMemsetAsync(d_ctr, 0, 8, sst));
    }
    const uint64_t *p2 = paired ? packed2_ : nullptr, *o2 = paired ? d_o2 : nullptr;
    const uint64_t nb2 = paired ? nblk2_ : 0;
#define CFR_LAUNCH_SEARCH(CPR_, PROF_, WIDE_, DYN_, PROFPTR_) \
    k_search_chains_v2<CPR_, PROF_, WIDE_, DYN_><<<blocks, kBlock, 0, sst>>>(sv, packed1_, d_o1, p2, o2, n, nblk1_, nb2, hit_off, raw, chain_cnt, PROFPTR_, d_ctr, dyn_chunk)
    // the two-launch form (round 6, k_search_chains_v2 STAGE 1 / 2): the state machine without wide text mode walks every chain and hands the
    // ones whose search reaches a range of 5 .. wide_rows rows over through a list; the full state machine runs over that list
    static const int split_env = dbg_env("CFR_SEARCH_SPLIT") ? atoi(dbg_env("CFR_SEARCH_SPLIT")) : -1;
    const bool prof_run = dbg_env("CFR_SEARCH_PROF") && atoi(dbg_env("CFR_SEARCH_PROF")) && !paired;
    bool split = sv.sa != nullptr && sv.wide_rows > 4 && nchains < 0xfff00000ull;
    if (split_env >= 0) split = split && split_env != 0; else split = split && search_split_default_;
    if (split) {
      uint64_t *list = (uint64_t *)scratch(par ? S_LIST1 : S_LIST, (nchains + ((size_t)num_cus_ * 8 * 4 + 1) * kListChunk) * 32);
      unsigned long long *lc = (unsigned long long *)scratch(S_LISTCTR, 64 * 8) + (par ? 8 : 0);       // [0] slots of the list, [1] / [2] the two launches' chain counters
      HIP_CHECK(hipMemsetAsync(lc, 0, 3 * 8, sst));
      unsigned long long *d_prof = nullptr;
      if (prof_run) { d_prof = (unsigned long long *)scratch(S_LISTCTR, 64 * 8) + 16; HIP_CHECK(hipMemsetAsync(d_prof, 0, 32 * 8, sst)); }
      auto both = [&](auto cpr_c, auto wide_c, auto dyn_c, auto prof_c) {
        constexpr int C = decltype(cpr_c)::value;
        constexpr bool W = decltype(wide_c)::value, D = decltype(dyn_c)::value, P = decltype(prof_c)::value;
        int occ = 0;
        unsigned first_blocks = blocks;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_search_chains_v2<C, P, W, D, 1>, kBlock, 0) == hipSuccess && occ > 0) {
          int want = blocks_forced_ ? blocks_per_cu_ : std::min(occ, 8);
          if (overlap_now_ && !blocks_forced_) want = std::min(want, 4);
          static const int first_cap = dbg_env("CFR_FIRST_BLOCKS") ? std::max(1, atoi(dbg_env("CFR_FIRST_BLOCKS"))) : 0;
          if (first_cap) want = std::min(occ, first_cap);
          first_blocks = std::min<unsigned>(grid_for(nchains), (unsigned)(num_cus_ * std::min(occ, want)));
        } else (void)hipGetLastError();
        k_search_chains_v2<C, P, W, D, 1><<<first_blocks, kBlock, 0, sst>>>(sv, packed1_, d_o1, p2, o2, n, nblk1_, nb2, hit_off, raw, chain_cnt, d_prof,
                                                                                 lc + 1, dyn_chunk, list, lc);
        unsigned list_blocks = blocks;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_search_chains_v2<C, P, W, D, 2>, kBlock, 0) == hipSuccess && occ > 0)
          list_blocks = std::min<unsigned>(list_blocks, (unsigned)(num_cus_ * occ));
        else (void)hipGetLastError();
        k_search_chains_v2<C, P, W, D, 2><<<list_blocks, kBlock, 0, sst>>>(sv, packed1_, d_o1, p2, o2, n, nblk1_, nb2, hit_off, raw, chain_cnt, d_prof ? d_prof + 16 : nullptr,
                                                                                lc + 2, dyn_chunk, list, lc);
      };
      auto pick_dyn = [&](auto cpr_c, auto wide_c) {
        if (prof_run) { if (dyn) both(cpr_c, wide_c, std::true_type{}, std::true_type{}); else both(cpr_c, wide_c, std::false_type{}, std::true_type{}); }
        else { if (dyn) both(cpr_c, wide_c, std::true_type{}, std::false_type{}); else both(cpr_c, wide_c, std::false_type{}, std::false_type{}); }
      };
      auto pick_wide = [&](auto cpr_c) { if (wide) pick_dyn(cpr_c, std::true_type{}); else pick_dyn(cpr_c, std::false_type{}); };
      if (paired) pick_wide(std::integral_constant<int, 4>{}); else pick_wide(std::integral_constant<int, 2>{});
      if (prof_run) {
        unsigned long long h_prof[32], h_lc[3];
        HIP_CHECK(hipMemcpyAsync(h_prof, d_prof, 32 * 8, hipMemcpyDeviceToHost, sst));
        HIP_CHECK(hipMemcpyAsync(h_lc, lc, 3 * 8, hipMemcpyDeviceToHost, sst));
        HIP_CHECK(hipStreamSynchronize(sst));
        static const char *names[] = {"idle", "table", "table10", "ext", "sa", "text", "text_hits", "lane_iterations", "ext_two_records", "text_rows", "block_loads", "saw", "textw"};
        for (int k = 0; k < 2; ++k) {
          fprintf(stderr, "[search prof %s] reads %zu list slots %llu:", k ? "stage 2" : "stage 1", n, h_lc[0]);
          for (int q = 0; q < 13; ++q) fprintf(stderr, " %s %.2f", names[q], (double)h_prof[16 * k + q] / (double)n);
          fprintf(stderr, " (per read)\n");
        }
      }
    } else
    if (dbg_env("CFR_SEARCH_PROF") && atoi(dbg_env("CFR_SEARCH_PROF")) && !paired) {
      // diagnostic: iteration mix of the state machine for this launch, on stderr
      unsigned long long *d_prof = (unsigned long long *)scratch(S_P5, 16 * 8), h_prof[16];
      HIP_CHECK(hipMemsetAsync(d_prof, 0, 15 * 8, sst));
      if (wide) { if (dyn) CFR_LAUNCH_SEARCH(2, true, true, true, d_prof); else CFR_LAUNCH_SEARCH(2, true, true, false, d_prof); }
      else { if (dyn) CFR_LAUNCH_SEARCH(2, true, false, true, d_prof); else CFR_LAUNCH_SEARCH(2, true, false, false, d_prof); }
      HIP_CHECK(hipMemcpyAsync(h_prof, d_prof, 15 * 8, hipMemcpyDeviceToHost, sst));
      HIP_CHECK(hipStreamSynchronize(sst));
      static const char *names[] = {"idle", "table", "table10", "ext", "sa", "text", "text_hits", "lane_iterations", "ext_two_records", "text_rows", "block_loads", "saw", "textw"};
      fprintf(stderr, "[search prof] reads %zu lanes %u:", n, blocks * kBlock);
      for (int q = 0; q < 13; ++q) fprintf(stderr, " %s %.2f", names[q], (double)h_prof[q] / (double)n);
      fprintf(stderr, " (per read)\n");
    } else if (wide) {
      if (paired) { if (dyn) CFR_LAUNCH_SEARCH(4, false, true, true, nullptr); else CFR_LAUNCH_SEARCH(4, false, true, false, nullptr); }
      else { if (dyn) CFR_LAUNCH_SEARCH(2, false, true, true, nullptr); else CFR_LAUNCH_SEARCH(2, false, true, false, nullptr); }
    } else {
      if (paired) { if (dyn) CFR_LAUNCH_SEARCH(4, false, false, true, nullptr); else CFR_LAUNCH_SEARCH(4, false, false, false, nullptr); }
      else { if (dyn) CFR_LAUNCH_SEARCH(2, false, false, true, nullptr); else CFR_LAUNCH_SEARCH(2, false, false, false, nullptr); }
    }
#undef CFR_LAUNCH_SEARCH
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipEventRecord(ev_[2], sst));
  return SearchBuf{hit_off, raw, chain_cnt, cap_total};
}

// Translated search (Classifier::TranslatedSearch): 6 chains per mate (strand x frame), one lane each
DeviceIndex::SearchBuf DeviceIndex::launch_search_protein(const uint8_t *d_b1, const uint64_t *d_o1, const uint8_t *d_b2, const uint64_t *d_o2, size_t n,
                                                          uint64_t total1, uint64_t total2, bool row_space_only) {
  DevView sview = view_;
  if (row_space_only) sview.text8 = nullptr;             // the hits leave with real BWT rows (no text mode)
  const bool paired = d_b2 != nullptr;
  const size_t nchains = n * (size_t)(paired ? 12 : 6);
  const uint64_t mhl1 = (uint64_t)view_.min_hit_len + 1;
  const uint64_t cap_total = 6 * (total1 / 3 / mhl1 + n) + (paired ? 6 * (total2 / 3 / mhl1 + n) : 0);
  uint64_t *cap = (uint64_t *)scratch(S_CAP, (n + 1) * 8);
  uint64_t *hit_off = (uint64_t *)scratch(S_HITOFF, (n + 1) * 8);
  cfr_hit *raw = (cfr_hit *)scratch(S_RAW, cap_total * sizeof(cfr_hit));
  uint32_t *chain_cnt = (uint32_t *)scratch(S_CHAINCNT, nchains * 4);
  size_t tmp_bytes = scan_tmp_bytes(n);
  void *tmp = scratch(S_SCAN, tmp_bytes);
  HIP_CHECK(hipEventRecord(ev_[0], stream_));
  HIP_CHECK(hipMemsetAsync(cap + n, 0, 8, stream_));
  k_caps_prot<<<grid_for(n), kBlock, 0, stream_>>>(view_, d_o1, d_o2, n, cap);
  exclusive_scan(tmp, tmp_bytes, cap, hit_off, n, stream_);
  HIP_CHECK(hipEventRecord(ev_[1], stream_));
  // the six translations of every read of the sub-batch, once (k_translate_prot), then the searches read plain codes
  // (regions are addressed by the read's byte offset and its number in the batch: prot_code_base)
  const uint64_t read0 = prot_o1_base_ ? (uint64_t)(d_o1 - prot_o1_base_) : 0;
  uint8_t *codes1 = (uint8_t *)scratch(S_PCODES1, 2 * prot_total1_ + 56 * (prot_reads_ + 1) + 128);
  uint8_t *codes2 = paired ? (uint8_t *)scratch(S_PCODES2, 2 * prot_total2_ + 56 * (prot_reads_ + 1) + 128) : nullptr;
  const unsigned tr_grid = std::min<unsigned>((unsigned)((n * (paired ? 2 : 1) + 3) / 4), (unsigned)(num_cus_ * 8));   // a wave per read and mate, grid-stride
  // the searches: lanes as state machines on a resident grid (k_search_prot_sm) when the image has the 128-byte records and the tables'
  // keys fit the twelve codes a lane looks at; CFR_PROT_SM=0: one chain per lane from start to end (k_search_prot), the same hits
  static const bool want_sm = !(dbg_env("CFR_PROT_SM") && atoi(dbg_env("CFR_PROT_SM")) == 0);
  const bool sm = want_sm && view_.prot.rec != nullptr && view_.ftabx_width <= 12 && view_.ftab_width <= 12 && nchains < 0xfc000000ull;
  unsigned sm_blocks = 0;
  static const int sm_minb = dbg_env("CFR_PROT_SM_MINB") ? atoi(dbg_env("CFR_PROT_SM_MINB")) : 1;
  static const int minb = dbg_env("CFR_PROT_MINB") ? atoi(dbg_env("CFR_PROT_MINB")) : 1;    // k_search_prot's register budget: 8 blocks per CU = 64 VGPRs (a few spills), 6 = 80
  if (sm) {
    int &occ = paired ? prot_occ_[1] : prot_occ_[0];        // (per image: two images on two devices may be inside this at once)
    if (occ == 0) {
      hipError_t e = paired ? (sm_minb == 6 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_search_prot_sm<2, 6>, kBlock, 0)
                                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_search_prot_sm<2, 1>, kBlock, 0))
                            : (sm_minb == 6 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_search_prot_sm<1, 6>, kBlock, 0)
                                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_search_prot_sm<1, 1>, kBlock, 0));
      if (e != hipSuccess || occ <= 0) occ = 4;
      if (dbg_env("CFR_PROT_BLOCKS")) occ = std::max(1, atoi(dbg_env("CFR_PROT_BLOCKS")));
    }
    sm_blocks = std::min<unsigned>(grid_for(nchains), (unsigned)(num_cus_ * occ));
  }
  // chains by wave tiles (round 5; CFR_PROT_DYN=0: chain + T)
  unsigned long long *prot_ctr = nullptr;
  static const bool prot_dyn = !(dbg_env("CFR_PROT_DYN") && atoi(dbg_env("CFR_PROT_DYN")) == 0);
  if (sm && prot_dyn) {
    prot_ctr = (unsigned long long *)scratch(S_P4, 16 * 8);
    HIP_CHECK(hipMemsetAsync(prot_ctr, 0, 8, stream_));
  }
  if (paired) {
    k_translate_prot<2><<<tr_grid, kBlock, 0, stream_>>>(view_, d_b1, d_o1, d_b2, d_o2, n, codes1, codes2, read0);
    if (sm && sm_minb == 6) k_search_prot_sm<2, 6><<<sm_blocks, kBlock, 0, stream_>>>(sview, d_o1, d_o2, n, hit_off, raw, chain_cnt, codes1, codes2, read0, prot_ctr, 64u);
    else if (sm) k_search_prot_sm<2, 1><<<sm_blocks, kBlock, 0, stream_>>>(sview, d_o1, d_o2, n, hit_off, raw, chain_cnt, codes1, codes2, read0, prot_ctr, 64u);
    else if (minb == 8) k_search_prot<2, 8><<<grid_for(nchains), kBlock, 0, stream_>>>(sview, d_b1, d_o1, d_b2, d_o2, n, hit_off, raw, chain_cnt, codes1, codes2, read0);
    else if (minb == 6) k_search_prot<2, 6><<<grid_for(nchains), kBlock, 0, stream_>>>(sview, d_b1, d_o1, d_b2, d_o2, n, hit_off, raw, chain_cnt, codes1, codes2, read0);
    else k_search_prot<2, 1><<<grid_for(nchains), kBlock, 0, stream_>>>(sview, d_b1, d_o1, d_b2, d_o2, n, hit_off, raw, chain_cnt, codes1, codes2, read0);
  } else {
    k_translate_prot<1><<<tr_grid, kBlock, 0, stream_>>>(view_, d_b1, d_o1, nullptr, nullptr, n, codes1, nullptr, read0);
    if (sm && sm_minb == 6) k_search_prot_sm<1, 6><<<sm_blocks, kBlock, 0, stream_>>>(sview, d_o1, nullptr, n, hit_off, raw, chain_cnt, codes1, nullptr, read0, prot_ctr, 64u);
    else if (sm) k_search_prot_sm<1, 1><<<sm_blocks, kBlock, 0, stream_>>>(sview, d_o1, nullptr, n, hit_off, raw, chain_cnt, codes1, nullptr, read0, prot_ctr, 64u);
    else if (minb == 8) k_search_prot<1, 8><<<grid_for(nchains), kBlock, 0, stream_>>>(sview, d_b1, d_o1, nullptr, nullptr, n, hit_off, raw, chain_cnt, codes1, nullptr, read0);
    else if (minb == 6) k_search_prot<1, 6><<<grid_for(nchains), kBlock, 0, stream_>>>(sview, d_b1, d_o1, nullptr, nullptr, n, hit_off, raw, chain_cnt, codes1, nullptr, read0);
    else k_search_prot<1, 1><<<grid_for(nchains), kBlock, 0, stream_>>>(sview, d_b1, d_o1, nullptr, nullptr, n, hit_off, raw, chain_cnt, codes1, nullptr, read0);
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipEventRecord(ev_[2], stream_));
  return SearchBuf{hit_off, raw, chain_cnt, cap_total};
}

void DeviceIndex::launch_post(const SearchBuf &sb, const uint8_t *d_b1, const uint64_t *d_o1, const uint8_t *d_b2, const uint64_t *d_o2, size_t n,
                              bool want_rows, Pipe &p, std::vector<uint64_t> *hit_begin_host, bool fused, hipStream_t st) {
  const bool paired = d_b2 != nullptr;
  const size_t nchains = n * (size_t)(paired ? 4 : 2);
  const uint64_t cap_total = sb.cap_total;
  uint64_t *hit_off = sb.hit_off;
  cfr_hit *raw = sb.raw;
  uint32_t *chain_cnt = sb.chain_cnt;
  cfr_hit *fin = (cfr_hit *)scratch(S_FIN, cap_total * sizeof(cfr_hit));
  uint64_t *fin_cnt = (uint64_t *)scratch(S_FINCNT, (n + 1) * 8);
  uint64_t *fin_rows = (uint64_t *)scratch(S_FINROWS, cap_total * 8);
  uint64_t *fin_off = (uint64_t *)scratch(S_FINOFF, (n + 1) * 8);
  size_t tmp_bytes = std::max(scan_tmp_bytes(n), scan_tmp_bytes(cap_total));
  void *tmp = scratch(S_SCAN2, tmp_bytes);
  uint64_t *totals = (uint64_t *)pinned(2 * 8);
  HIP_CHECK(hipEventRecord(ev_[8], st));
  HIP_CHECK(hipMemsetAsync(fin_cnt + n, 0, 8, st));
  uint64_t *read_rows = fused ? (uint64_t *)scratch(S_READROWS, (n + 1) * 8) : nullptr;
  if (view_.prot.enabled) {
    if (paired) k_select_prot<2><<<grid_for(n), kBlock, 0, st>>>(view_, d_o1, d_o2, n, hit_off, raw, chain_cnt, fin, fin_cnt, fin_rows, read_rows);
    else k_select_prot<1><<<grid_for(n), kBlock, 0, st>>>(view_, d_o1, nullptr, n, hit_off, raw, chain_cnt, fin, fin_cnt, fin_rows, read_rows);
  } else
  if (paired) k_adjust_select<4><<<grid_for(n), kBlock, 0, st>>>(view_, d_b1, d_o1, d_b2, d_o2, n, hit_off, raw, chain_cnt, fin, fin_cnt, fin_rows, read_rows);
  else k_adjust_select<2><<<grid_for(n), kBlock, 0, st>>>(view_, d_b1, d_o1, nullptr, nullptr, n, hit_off, raw, chain_cnt, fin, fin_cnt, fin_rows, read_rows);
  HIP_CHECK(hipGetLastError());
  if (fused) {
    // per-read row bases; one 8-byte sync sizes the tail's scratch
    uint64_t *read_row_off = (uint64_t *)scratch(S_READROWOFF, (n + 1) * 8);
    HIP_CHECK(hipMemsetAsync(read_rows + n, 0, 8, st));
    exclusive_scan(tmp, tmp_bytes, read_rows, read_row_off, n, st);
    HIP_CHECK(hipEventRecord(ev_[3], st));
    HIP_CHECK(hipMemcpyAsync(&totals[1], read_row_off + n, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const uint64_t nrows_f = totals[1];
    uint64_t *vals_f = (uint64_t *)scratch(S_ROWVALS, (nrows_f + 1) * 8);
    HIP_CHECK(hipEventRecord(ev_[4], st));
    HIP_CHECK(hipEventRecord(ev_[5], st));
    HIP_CHECK(hipEventRecord(ev_[6], st));
    p = Pipe{hit_off, fin_cnt, read_row_off, nullptr, vals_f, fin, 0, nrows_f};
    last_stats.n_chains += nchains;
    last_stats.n_rows += nrows_f;
    return;
  }
  exclusive_scan(tmp, tmp_bytes, fin_cnt, fin_off, n, st);
  HIP_CHECK(hipEventRecord(ev_[3], st));
  HIP_CHECK(hipMemcpyAsync(&totals[0], fin_off + n, 8, hipMemcpyDeviceToHost, st));
  if (hit_begin_host) {
    hit_begin_host->resize(n + 1);
    HIP_CHECK(hipMemcpyAsync(hit_begin_host->data(), fin_off, (n + 1) * 8, hipMemcpyDeviceToHost, st));
  }
  HIP_CHECK(hipStreamSynchronize(st));
  const uint64_t nhits = totals[0];

  cfr_hit *hits = (cfr_hit *)scratch(S_HITS, (nhits + 1) * sizeof(cfr_hit));
  uint64_t *rows_per = (uint64_t *)scratch(S_ROWSPER, (nhits + 1) * 8);
  uint64_t *row_off = (uint64_t *)scratch(S_ROWOFF, (nhits + 1) * 8);
  k_compact_hits<<<grid_for(n), kBlock, 0, st>>>(n, hit_off, fin_off, fin, fin_rows, hits, rows_per);
  HIP_CHECK(hipGetLastError());
  uint64_t nrows = 0;
  uint64_t *rows = nullptr, *vals = nullptr;
  HIP_CHECK(hipEventRecord(ev_[4], st));
  if (want_rows) {
    HIP_CHECK(hipMemsetAsync(rows_per + nhits, 0, 8, st));
    tmp_bytes = std::max(tmp_bytes, scan_tmp_bytes(nhits));
    tmp = scratch(S_SCAN2, tmp_bytes);
    exclusive_scan(tmp, tmp_bytes, rows_per, row_off, nhits, st);
    HIP_CHECK(hipMemcpyAsync(&totals[1], row_off + nhits, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    nrows = totals[1];
    rows = (uint64_t *)scratch(S_ROWS, (nrows + 1) * 8);
    vals = (uint64_t *)scratch(S_ROWVALS, (nrows + 1) * 8);
    if (nhits) k_enum_rows<<<grid_for(nhits), kBlock, 0, st>>>(view_, nhits, hits, row_off, rows);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(ev_[5], st));
    if (nrows) k_locate<<<grid_for(nrows), kBlock, 0, st>>>(view_, rows, nrows, vals, nullptr);
    HIP_CHECK(hipGetLastError());
  } else {
    HIP_CHECK(hipEventRecord(ev_[5], st));
  }
  HIP_CHECK(hipEventRecord(ev_[6], st));
  p = Pipe{hit_off, fin_off, row_off, rows, vals, hits, nhits, nrows};
  last_stats.n_chains += nchains;
  last_stats.n_hits += nhits;
  last_stats.n_rows += nrows;
}


void DeviceIndex::run_device_stages(const uint8_t *d_b1, const uint64_t *d_o1, const uint8_t *d_b2, const uint64_t *d_o2, size_t n,
                                    uint64_t total1, uint64_t total2, bool want_rows, Pipe &p, std::vector<uint64_t> *hit_begin_host,
                                    bool fused, bool row_space_only) {
  const SearchBuf sb = launch_search(d_b1, d_o1, d_b2, d_o2, n, total1, total2, 0, row_space_only);
  launch_post(sb, d_b1, d_o1, d_b2, d_o2, n, want_rows, p, hit_begin_host, fused, stream_);
}

// SDUST (the reference's pre-step of Query, CentrifugerClass.cpp:276-316) on reads that are already in HBM: masks d_bases in
// place, asynchronously on `st` (k_dust; bounded per-lane state, no host involvement).
void DeviceIndex::dust_on_device(uint8_t *d_bases, const uint64_t *d_offs, size_t n, hipStream_t st) {
  if (n == 0) return;
  // reads of A, C, G, T only go through the instantiation with 64-triplet tables (three waves per SIMD), the others
  // (flagged by one pass over the bases) through the one with 125; CFR_DUST_SPLIT=0: everything through the latter
  static const bool split = !(dbg_env("CFR_DUST_SPLIT") && atoi(dbg_env("CFR_DUST_SPLIT")) == 0);
  static const int pure_per_cu = dbg_env("CFR_DUST_BLOCKS") ? std::max(1, atoi(dbg_env("CFR_DUST_BLOCKS"))) : (CFR_DUST_RINGLESS ? 9 : 6) * (128 / kDustBlock) + (kDustBlock == 64 && !CFR_DUST_RINGLESS ? 1 : 0);   // resident blocks of k_dust<true>: 16.6 KB of LDS each without the ring
  const unsigned blocks_pure = std::min<unsigned>(grid_for(n, kDustBlock), (unsigned)(num_cus_ * pure_per_cu));
  const unsigned blocks_any = std::min<unsigned>(grid_for(n, kDustBlock), (unsigned)(num_cus_ * (kDustBlock == 64 ? 7 : 4)));
  uint32_t *pool = (uint32_t *)scratch(st == stream_ ? S_DUSTPOOL : S_DUSTPOOL2,
                                       ((size_t)(blocks_pure + blocks_any) * kDustBlock * 64 + kDustPoolHead) * sizeof(uint32_t));   // one table per stream
  HIP_CHECK(hipMemsetAsync(pool, 0, kDustPoolHead * sizeof(uint32_t), st));      // the counters the lanes draw reads from
  if (split) {
    // flags: a byte per read; behind them (8-byte aligned) the list of flagged reads; its length is the third counter of the pool head
    const size_t flag_bytes = (n + 16 + 7) & ~(size_t)7;
    uint8_t *flags = (uint8_t *)scratch(st == stream_ ? S_DUSTFLAG : S_DUSTFLAG2, flag_bytes + n * 4);
    uint32_t *list = (uint32_t *)(flags + flag_bytes);
    unsigned long long *list_cnt = (unsigned long long *)pool + 2;
    k_dust_flags<<<std::min<unsigned>((unsigned)((n + 255) / 256), (unsigned)(num_cus_ * 16)), 256, 0, st>>>(d_bases, d_offs, n, flags, list, list_cnt);
    // the screen (round 6): reads in which no triplet occurs six times inside 62 consecutive ones cannot hold a perfect interval; k_dust<true> skips them.
    // Not for long reads (nearly every one has such a window somewhere): CFR_DUST_SCREEN=0 / 1 forces it off / on
    static const int screen_env = dbg_env("CFR_DUST_SCREEN") ? atoi(dbg_env("CFR_DUST_SCREEN")) : -1;
    const bool screen = screen_env >= 0 ? screen_env != 0 : dust_mean_len_ < 600.0;
    if (screen) k_dust_screen<<<std::min<unsigned>((unsigned)((n + 255) / 256), (unsigned)(num_cus_ * 8)), 256, 0, st>>>(d_bases, d_offs, n, flags);
    // one after the other: side by side (second stream) the 125-triplet blocks take the LDS first and both get slower
    // (measured: 10.5 ms against 10.85 with 14 % flagged reads, 12.5 ms with its grid cut to the flagged count)
    k_dust<true><<<blocks_pure, kDustBlock, 0, st>>>(d_bases, d_offs, n, pool, 0, flags, nullptr, nullptr);
    k_dust<false><<<blocks_any, kDustBlock, 0, st>>>(d_bases, d_offs, n, pool, blocks_pure, flags, list, list_cnt);
  } else k_dust<false><<<blocks_any, kDustBlock, 0, st>>>(d_bases, d_offs, n, pool, 0, nullptr, nullptr, nullptr);
  HIP_CHECK(hipGetLastError());
#ifdef CFR_DUST_PROF
  {
    unsigned long long h[16];
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(dust_prof), sizeof(h)));
    static const char *nm[5] = {"step", "slow", "trim", "find", "next"};
    const double waves = (double)(blocks_pure + blocks_any) * kDustBlock / 64;      // (both launches add into the same counters)
    for (int k = 0; k < 5; ++k)
      fprintf(stderr, "[dust] %s rounds %llu lanes %llu (%.1f per round)  clocks per wave %.0f (%.0f per round)\n", nm[k], h[2 * k], h[2 * k + 1],
              h[2 * k] ? (double)h[2 * k + 1] / h[2 * k] : 0.0, h[10 + k] / waves, h[2 * k] ? (double)h[10 + k] / h[2 * k] : 0.0);
    unsigned long long z[16] = {0};
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(dust_prof), z, sizeof(z)));
  }
#endif
}

void DeviceIndex::dust_mask_host(uint8_t *bases, const uint64_t *offs, size_t n) {
  HIP_CHECK(hipSetDevice(device_));
  if (n == 0) return;
  const uint64_t total = offs[n];
  uint8_t *d_b = (uint8_t *)scratch(S_IN_B1, total + 16);
  uint64_t *d_o = (uint64_t *)scratch(S_IN_O1, (n + 1) * 8);
  if (total) HIP_CHECK(hipMemcpyAsync(d_b, bases, total, hipMemcpyHostToDevice, stream_));
  HIP_CHECK(hipMemcpyAsync(d_o, offs, (n + 1) * 8, hipMemcpyHostToDevice, stream_));
  dust_mean_len_ = (double)total / (double)n;
  dust_on_device(d_b, d_o, n, stream_);
  if (total) HIP_CHECK(hipMemcpyAsync(bases, d_b, total, hipMemcpyDeviceToHost, stream_));
  HIP_CHECK(hipStreamSynchronize(stream_));
}

// 2-bit packed form of the read buffers for k_search_chains_v2 (once per batch call, before the sub-batches)
void DeviceIndex::pack_inputs(const uint8_t *d_b1, uint64_t total1, const uint8_t *d_b2, uint64_t total2, bool pack_now) {
  // 4 zero blocks ("not a symbol") in front of and behind the packed form: the search kernel fetches block pairs
  auto pack_one = [&](size_t slot, const uint8_t *d_b, uint64_t total, uint64_t &nblk) -> uint64_t * {
    nblk = (total + 15) / 16;
    uint64_t *base = (uint64_t *)scratch(slot, (nblk + 8) * 8);
    HIP_CHECK(hipMemsetAsync(base, 0, 4 * 8, stream_));
    HIP_CHECK(hipMemsetAsync(base + 4 + nblk, 0, 4 * 8, stream_));
    if (nblk && pack_now) k_pack_reads<<<grid_for(nblk), kBlock, 0, stream_>>>(d_b, total, nblk, base + 4);
    return base + 4;
  };
  packed1_ = pack_one(S_PACK1, d_b1, total1, nblk1_);
  nblk2_ = 0;
  packed2_ = nullptr;
  if (d_b2) packed2_ = pack_one(S_PACK2, d_b2, total2, nblk2_);
  HIP_CHECK(hipGetLastError());
}

void DeviceIndex::finish_stats(bool want_rows) {
  auto ms = [&](int a, int b) { float t = 0; (void)hipEventElapsedTime(&t, ev_[a], ev_[b]); return t; };
  last_stats.pack_ms += ms(0, 1);
  last_stats.search_ms += ms(1, 2);
  last_stats.adjust_ms += ms(8, 3);
  last_stats.rows_ms += want_rows ? ms(4, 5) : 0.f;
  last_stats.locate_ms += want_rows ? ms(5, 6) : 0.f;
  last_stats.tail_ms += ms(6, 7);
  last_stats.total_ms += ms(0, 7);
}

void DeviceIndex::run_batch(const uint8_t *d_b1, const uint64_t *d_o1, const uint8_t *d_b2, const uint64_t *d_o2, size_t n,
                            uint64_t total1, uint64_t total2, bool want_rows, BatchOut &out) {
  HIP_CHECK(hipSetDevice(device_));
  out.hit_begin.assign(n + 1, 0);
  out.hits.clear();
  out.row_begin.clear();
  out.row_vals.clear();
  out.read_len.assign(n, 0);
  last_stats = cfr_batch_stats{};
  if (n == 0) return;
  prot_total1_ = total1; prot_total2_ = total2; prot_o1_base_ = d_o1; prot_reads_ = n;
  Pipe p;
  if (!search_v1_) pack_inputs(d_b1, total1, d_b2, total2);
  run_device_stages(d_b1, d_o1, d_b2, d_o2, n, total1, total2, want_rows, p, &out.hit_begin, false, /*row_space_only=*/true);   // the hits leave with real BWT rows
  out.hits.resize(p.nhits);
  if (p.nhits) HIP_CHECK(hipMemcpyAsync(out.hits.data(), p.hits, p.nhits * sizeof(cfr_hit), hipMemcpyDeviceToHost, stream_));
  if (want_rows) {
    out.row_begin.resize(p.nhits + 1);
    HIP_CHECK(hipMemcpyAsync(out.row_begin.data(), p.row_off, (p.nhits + 1) * 8, hipMemcpyDeviceToHost, stream_));
    out.row_vals.resize(p.nrows);
    if (p.nrows) HIP_CHECK(hipMemcpyAsync(out.row_vals.data(), p.vals, p.nrows * 8, hipMemcpyDeviceToHost, stream_));
  }
  // query lengths (Classifier.hpp:958-960) from the offsets
  std::vector<uint64_t> o1(n + 1), o2;
  HIP_CHECK(hipMemcpyAsync(o1.data(), d_o1, (n + 1) * 8, hipMemcpyDeviceToHost, stream_));
  if (d_b2) { o2.resize(n + 1); HIP_CHECK(hipMemcpyAsync(o2.data(), d_o2, (n + 1) * 8, hipMemcpyDeviceToHost, stream_)); }
  HIP_CHECK(hipEventRecord(ev_[7], stream_));
  HIP_CHECK(hipStreamSynchronize(stream_));
  for (size_t i = 0; i < n; ++i) {
    out.read_len[i] = (int32_t)(o1[i + 1] - o1[i]);
    if (d_b2) out.read_len[i] += (int32_t)(o2[i + 1] - o2[i]);
  }
  finish_stats(want_rows);
}

void DeviceIndex::run_batch_host(const uint8_t *b1, const uint64_t *o1, const uint8_t *b2, const uint64_t *o2, size_t n,
                                 bool want_rows, BatchOut &out) {
  HIP_CHECK(hipSetDevice(device_));
  Staged st = stage_inputs(b1, o1, b2, o2, n);
  run_batch(st.b1, st.o1, st.b2, st.o2, n, st.t1, st.t2, want_rows, out);
}

// Whole Query on the device.  matches: max_result > 0 -> read i owns [i*max_result, ...); otherwise the
// read's slice of the located-row space.  *match_extent = number of match slots the caller must provide.
// The batch is cut into sub-batches ("pieces"): the D2H copy of piece k (copy stream) overlaps the kernels of k+1.
std::vector<std::pair<size_t, size_t>> DeviceIndex::cut_pieces(size_t n, bool per_read_slots, size_t &sb, uint64_t total_bases) const {
  // full sub-batches, then the last one is halved down to taper_floor_ reads so that the copy left exposed after the
  // last kernel is small; at most kMaxSub pieces (one event set each).  A sub-batch of long reads is as large as the HBM left
  // beside the image allows (its raw hit lists take ~2.7 bytes per base): the search kernel walks one chain per lane, and a
  // launch with fewer chains than lanes runs at the pace of its longest reads (1 M long reads: 8.5e6 reads/s in one piece,
  // 5.7e6 in twelve)
  const size_t kTaperMax = 4;
  size_t want = sub_batch_;
  if (per_read_slots && view_.max_result > 16)            // two sets of max_result 24-byte match slots per read: at most ~8 GB of them
    want = std::max<size_t>(4096, std::min<size_t>(want, (size_t)(8e9 / (48.0 * (double)view_.max_result))));
  if (n && total_bases && (double)want * ((double)total_bases / (double)n) > (double)piece_bases_max_)
    want = std::max<size_t>(1, (size_t)((double)piece_bases_max_ / ((double)total_bases / (double)n)));
  sb = per_read_slots ? std::max(want, (n + (kMaxSub - kTaperMax) - 1) / (kMaxSub - kTaperMax)) : n;   // row-space matches: one piece
  std::vector<std::pair<size_t, size_t>> pieces;
  size_t lo = 0;
  while (n - lo > sb) { pieces.emplace_back(lo, sb); lo += sb; }
  size_t rem = n - lo;
  for (size_t t = 0; per_read_slots && taper_floor_ && t + 1 < kTaperMax && rem > 2 * taper_floor_; ++t) {
    const size_t c = rem / 2;
    pieces.emplace_back(lo, c);
    lo += c;
    rem -= c;
  }
  pieces.emplace_back(lo, rem);
  return pieces;
}

// --expand-taxid: the lists leave through a pool of the image (exp_append).  Its size is a guess that the workload corrects: a call
// whose records did not fit is run again with a pool that holds them (the cursor counts on past the end), and the size is kept.
void DeviceIndex::expand_begin() {
  expanded_raw_.clear();
  if (!host_->params.output_expanded) return;
  if (!exp_cap_) {
    exp_cap_ = 1ull << 20;
    if (const char *e = dbg_env("CFR_EXP_POOL_INIT")) exp_cap_ = std::max<uint64_t>(4, strtoull(e, nullptr, 10));    // test hook: a pool that has to grow
  }
  uint64_t *pool = (uint64_t *)scratch(S_EXPPOOL, exp_cap_ * 8);
  unsigned long long *cur = (unsigned long long *)scratch(S_EXPCUR, 8);
  HIP_CHECK(hipMemsetAsync(cur, 0, 8, stream_));
  if (view_.exp_pool != pool || view_.exp_cursor != cur || view_.exp_cap != exp_cap_) {
    view_.exp_pool = pool; view_.exp_cursor = cur; view_.exp_cap = exp_cap_;
    HIP_CHECK(hipMemcpyAsync(d_view_, &view_, sizeof(DevView), hipMemcpyHostToDevice, stream_));
  }
  HIP_CHECK(hipStreamSynchronize(stream_));                // (the post stage may run on another stream)
}
bool DeviceIndex::expand_end() {
  if (!host_->params.output_expanded) return true;
  unsigned long long used = 0;
  HIP_CHECK(hipMemcpy(&used, view_.exp_cursor, 8, hipMemcpyDeviceToHost));
  if (used > exp_cap_) { exp_cap_ = (uint64_t)used + (uint64_t)used / 4 + 1024; return false; }
  expanded_raw_.resize((size_t)used);
  if (used) HIP_CHECK(hipMemcpy(expanded_raw_.data(), view_.exp_pool, (size_t)used * 8, hipMemcpyDeviceToHost));
  return true;
}

void DeviceIndex::classify_device(const uint8_t *d_b1, const uint64_t *d_o1, const uint8_t *d_b2, const uint64_t *d_o2, size_t n,
                                  uint64_t total1, uint64_t total2, cfr_result *results, cfr_match *matches, size_t match_cap,
                                  size_t *match_extent, const HostSrc *src, bool compact) {
  HIP_CHECK(hipSetDevice(device_));
  last_stats = cfr_batch_stats{};
  pre_hit_off_ = nullptr;
  if (match_extent) *match_extent = 0;
  expanded_raw_.clear();
  if (n == 0) return;
  expand_begin();
  prot_total1_ = total1; prot_total2_ = total2; prot_o1_base_ = d_o1; prot_reads_ = n;
  const uint64_t stride = view_.max_result > 0 ? (uint64_t)view_.max_result : 0;
  if (stride && match_extent) *match_extent = stride * n;
  if (stride && stride * n > match_cap) throw CapacityError{"match buffer too small"};
  if (compact && !stride) throw HipError{"the compact result layout needs max_result > 0", -2};
  const size_t res_bytes = compact ? sizeof(cfr_result_compact) : sizeof(cfr_result), match_bytes = compact ? sizeof(cfr_match_compact) : sizeof(cfr_match);
  // SDUST of reads that are already on the device (the caller's buffer stays as it is: a private copy is masked): the whole
  // batch is copied and masked up front.  The other schedule - copy + mask kernel of sub-batch k + 1 on the dust stream beside the
  // search of sub-batch k, as the streamed host inputs do - is kept behind CFR_DUST_PIECES=1 because it LOSES here: 31.9 ms per
  // 10 M reads against 25.8 ms up front (the search kernel issues ~60 % of its VALU slots itself, so the mask kernel beside it
  // slows both; with host inputs the same overlap pays because the link, not the kernels, bounds the step).
  const bool dust_here = dust_ && !src && !view_.prot.enabled;      // (a protein index takes the reads as they are: CentrifugerClass.cpp:276)
  dust_mean_len_ = n ? (double)(total1 + total2) / (double)(n * (d_b2 ? 2 : 1)) : 0.0;
  bool dust_pieces = false;
  if (const char *e = dbg_env("CFR_DUST_PIECES")) dust_pieces = dust_here && !search_v1_ && stride > 0 && one_launch_ready() && atoi(e) != 0;
  const uint8_t *orig_b1 = d_b1, *orig_b2 = d_b2;
  if (dust_here && !dust_pieces) {
    auto masked_copy = [&](size_t slot, const uint8_t *d_b, const uint64_t *d_o, uint64_t total) -> const uint8_t * {
      uint8_t *c = (uint8_t *)scratch(slot, total + 16);
      if (total) HIP_CHECK(hipMemcpyAsync(c, d_b, total, hipMemcpyDeviceToDevice, stream_));
      dust_on_device(c, d_o, n, stream_);
      return c;
    };
    d_b1 = masked_copy(S_DUSTTMP, d_b1, d_o1, total1);
    if (d_b2) d_b2 = masked_copy(S_DUSTTMP2, d_b2, d_o2, total2);
  } else if (dust_pieces) {
    d_b1 = (const uint8_t *)scratch(S_DUSTTMP, total1 + 16);
    if (d_b2) d_b2 = (const uint8_t *)scratch(S_DUSTTMP2, total2 + 16);
  }
  const bool by_piece = src != nullptr || dust_pieces;             // the bases of a sub-batch arrive (and are packed) right before its search
  // resident reads in several sub-batches: each sub-batch is packed right in front of its own search (test hook CFR_PACK_PIECES=1) instead of
  // the whole batch up front - the packing of sub-batch k + 1 then runs while the post stage of k - 1 still has the other stream
  static const bool pack_pieces_on = dbg_env("CFR_PACK_PIECES") && atoi(dbg_env("CFR_PACK_PIECES")) != 0;
  // test hook CFR_PACK_SPLIT=1: the first sub-batch packed in front of its search, the rest in one launch on the (otherwise idle) upload
  // stream beside that search.  Measured (profiles/r4x_ab_pack_split.txt): 13.2-13.3 against 13.05 ms per step, pairs 29.0 against 28.2 -
  // the 0.4 ms of packing it takes out of the front come back as a slower first search; everything up front on the main stream stays
  static const bool pack_split_on = dbg_env("CFR_PACK_SPLIT") && atoi(dbg_env("CFR_PACK_SPLIT")) != 0;
  const bool pack_late = (pack_pieces_on || pack_split_on) && !by_piece && !search_v1_;
  if (!search_v1_) pack_inputs(d_b1, total1, d_b2, total2, /*pack_now=*/!by_piece && !pack_late);
  size_t sb = 0;
  const auto pieces = cut_pieces(n, stride > 0, sb, total1 + total2);
  const size_t nsub = pieces.size();
  const bool paired = d_b2 != nullptr;
  // bases of every piece (its buffers are sized by them): from the caller's offsets, or 8 bytes per boundary from the device
  std::vector<uint64_t> pt1(nsub, total1), pt2(nsub, total2);
  std::vector<uint64_t> b1(nsub + 1, 0), b2(nsub + 1, 0);          // byte offsets of the pieces' first reads (and of the end)
  // resident reads in several sub-batches: the hit-list offsets of all reads in one pass (k_caps + scan per sub-batch were 12 small
  // launches in front of 12 searches: 0.33 ms of a 14.4 ms step), the sub-batches' first offsets fetched with their boundaries
  static const bool caps_once_on = !(dbg_env("CFR_CAPS_ONCE") && atoi(dbg_env("CFR_CAPS_ONCE")) == 0);
  const bool caps_once = caps_once_on && !src && !dust_pieces && nsub > 1 && stride > 0 && one_launch_ready() && !search_v1_;
  uint64_t *hit_all = nullptr;
  std::vector<uint64_t> hbase(nsub + 1, 0);
  if (caps_once) {
    uint64_t *cap_all = (uint64_t *)scratch(S_CAPALL, (n + 1) * 8);
    hit_all = (uint64_t *)scratch(S_HITALL, (n + 1) * 8);
    size_t tb = scan_tmp_bytes(n);
    void *tmp = scratch(S_SCANALL, tb);
    HIP_CHECK(hipMemsetAsync(cap_all + n, 0, 8, stream_));
    k_caps<<<grid_for(n), kBlock, 0, stream_>>>(view_, d_o1, d_o2, n, cap_all);
    exclusive_scan(tmp, tb, cap_all, hit_all, n, stream_);
  }
  if (nsub > 1 || by_piece) {
    if (src) {
      for (size_t k = 0; k <= nsub; ++k) {
        const size_t at = k < nsub ? pieces[k].first : n;
        b1[k] = src->o1[at];
        if (paired) b2[k] = src->o2[at];
      }
    } else {
      // one launch that stores the pieces' boundaries (and their hit-list offsets) into pinned host memory, one synchronisation
      static_assert(kMaxSub + 1 <= 17, "PieceFirsts");
      unsigned long long *pin = (unsigned long long *)pinned((2 + kCtlWords * kMaxSub + 3 * (kMaxSub + 1)) * 8);
      uint64_t *pb1 = (uint64_t *)(pin + 2 + kCtlWords * kMaxSub), *pb2 = pb1 + (kMaxSub + 1), *ph = pb2 + (kMaxSub + 1);
      PieceFirsts pf;
      pf.n = (uint32_t)(nsub + 1);
      for (size_t k = 0; k <= nsub; ++k) pf.at[k] = k < nsub ? pieces[k].first : n;
      k_fetch_piece_scalars<<<1, 64, 0, stream_>>>(pf, d_o1, paired ? d_o2 : nullptr, hit_all, pb1, pb2, ph);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(stream_));
      for (size_t k = 0; k <= nsub; ++k) { b1[k] = pb1[k]; b2[k] = pb2[k]; if (k < nsub) hbase[k] = ph[k]; }
    }
    // one size for all pieces (the largest): the scratch buffers are then allocated once, not regrown under running kernels
    uint64_t m1 = 0, m2 = 0;
    for (size_t k = 0; k < nsub; ++k) { m1 = std::max(m1, b1[k + 1] - b1[k]); if (paired) m2 = std::max(m2, b2[k + 1] - b2[k]); }
    for (size_t k = 0; k < nsub; ++k) { pt1[k] = m1; pt2[k] = m2; }
  }
  const bool pack_late_now = pack_late && pack_pieces_on && nsub > 1;
  bool pack_rest_pending = false;
  if (pack_late && !pack_late_now) {
    // blocks [from, to) of a buffer on stream st
    auto part = [&](const uint8_t *db, uint64_t total, uint64_t *packed, uint64_t from, uint64_t to, hipStream_t st) {
      if (to > from) k_pack_reads<<<grid_for(to - from), kBlock, 0, st>>>(db + (from << 4), total - (from << 4), to - from, packed + from);
    };
    if (nsub > 1) {
      const uint64_t c1 = std::min(nblk1_, (b1[1] + 15) >> 4), c2 = paired ? std::min(nblk2_, (b2[1] + 15) >> 4) : 0;
      part(d_b1, total1, packed1_, 0, c1, stream_);
      if (paired) part(d_b2, total2, packed2_, 0, c2, stream_);
      // (the block the first two sub-batches share is packed by both launches, with the same bits)
      part(d_b1, total1, packed1_, b1[1] >> 4, nblk1_, h2d_stream_);
      if (paired) part(d_b2, total2, packed2_, b2[1] >> 4, nblk2_, h2d_stream_);
      HIP_CHECK(hipEventRecord(copied_[0], h2d_stream_));
      pack_rest_pending = true;
    } else {                                   // a single sub-batch: pack it now after all
      part(d_b1, total1, packed1_, 0, nblk1_, stream_);
      if (paired) part(d_b2, total2, packed2_, 0, nblk2_, stream_);
    }
    HIP_CHECK(hipGetLastError());
  }
  const bool fused = fused_tail_ && locate_direct();     // k_tail locates rows itself (memo / suffix array + step function / virtual rows)
  const bool one_launch = fused && stride > 0 && fused_post_ && !view_.prot.enabled;   // k_adjust_tail: no host round trip in a piece
  // streamed host inputs (classify_host): bases of piece k are copied on the h2d stream and packed right before its search
  std::vector<uint8_t> have_piece(nsub, by_piece ? 0 : 1);
  uint64_t sent_end1 = 0, sent_end2 = 0;       // packed host blocks: the first block of each mate's buffer that has not been copied up yet
  auto bring_piece = [&](size_t k) {
    if (!by_piece || have_piece[k]) return;
    const size_t lo = pieces[k].first, hi = lo + pieces[k].second;
    if (dust_pieces) {                          // device to device, then the mask kernel, both on the dust stream
      if (b1[k + 1] > b1[k]) HIP_CHECK(hipMemcpyAsync(const_cast<uint8_t *>(d_b1) + b1[k], orig_b1 + b1[k], b1[k + 1] - b1[k], hipMemcpyDeviceToDevice, dust_stream_));
      if (paired && b2[k + 1] > b2[k]) HIP_CHECK(hipMemcpyAsync(const_cast<uint8_t *>(d_b2) + b2[k], orig_b2 + b2[k], b2[k + 1] - b2[k], hipMemcpyDeviceToDevice, dust_stream_));
      dust_on_device(const_cast<uint8_t *>(d_b1), d_o1 + lo, hi - lo, dust_stream_);
      if (paired) dust_on_device(const_cast<uint8_t *>(d_b2), d_o2 + lo, hi - lo, dust_stream_);
      HIP_CHECK(hipEventRecord(h2d_done_[k], dust_stream_));
      have_piece[k] = 1;
      return;
    }
    auto one = [&](const uint8_t *hb, const uint64_t *hp, const uint64_t *ho, const uint8_t *db, uint64_t *packed, uint64_t &sent_end) {
      const uint64_t a = ho[lo], b = ho[hi];
      if (b <= a) return;
      if (hp) {
        // packed blocks: into the search kernel's own buffer - or, with SDUST, into a staging copy (the search's blocks are then packed
        // from the masked characters) - and unpacked, this sub-batch's characters only, for SDUST and the post stage.  The block a
        // sub-batch shares with the one before it came with that one (same stream, in order) and is not written again: the search
        // of the previous sub-batch may still be reading it.
        const uint64_t k0 = a >> 4, k1 = (b + 15) >> 4;
        const uint64_t kc = std::max(k0, std::min(sent_end, k1));
        if (k1 > kc) HIP_CHECK(hipMemcpyAsync(packed + kc, hp + kc, (k1 - kc) * 8, hipMemcpyHostToDevice, h2d_stream_));
        sent_end = k1;
        k_unpack_reads<<<grid_for(k1 - k0), kBlock, 0, h2d_stream_>>>(packed + k0, k1 - k0, const_cast<uint8_t *>(db) + (k0 << 4), a - (k0 << 4), b - (k0 << 4));
        HIP_CHECK(hipGetLastError());
      } else HIP_CHECK(hipMemcpyAsync(const_cast<uint8_t *>(db) + a, hb + a, b - a, hipMemcpyHostToDevice, h2d_stream_));
    };
    one(src->b1, src->p1, src->o1, d_b1, src->stage1 ? src->stage1 : packed1_, sent_end1);
    if (paired) one(src->b2, src->p2, src->o2, d_b2, src->stage2 ? src->stage2 : packed2_, sent_end2);
    if (dust_ && !view_.prot.enabled) {
      // masked on a stream of its own, behind the piece's copy: the copy stream goes straight on with the next piece (the link
      // is what bounds this entry: 187 MB per piece at ~47 GB/s = 4 ms, the mask kernel 1.1-1.4 ms - on the copy stream
      // itself every piece paid both)
      HIP_CHECK(hipEventRecord(copied_[k], h2d_stream_));
      HIP_CHECK(hipStreamWaitEvent(dust_stream_, copied_[k], 0));
      dust_on_device(const_cast<uint8_t *>(d_b1), d_o1 + lo, hi - lo, dust_stream_);
      if (paired) dust_on_device(const_cast<uint8_t *>(d_b2), d_o2 + lo, hi - lo, dust_stream_);
      HIP_CHECK(hipEventRecord(h2d_done_[k], dust_stream_));
    } else
    HIP_CHECK(hipEventRecord(h2d_done_[k], h2d_stream_));
    have_piece[k] = 1;
  };
  auto pack_piece = [&](size_t k) {            // on the main stream, behind the copy of the piece
    if (!by_piece && !pack_late_now) return;
    if (by_piece) HIP_CHECK(hipStreamWaitEvent(stream_, h2d_done_[k], 0));
    if (src && src->p1 && !src->stage1) return;       // the caller's packed blocks are in place (with SDUST they went to the staging copy and the masked characters are packed here)
    auto one = [&](uint64_t from, uint64_t to, const uint8_t *db, uint64_t total, uint64_t *packed) {
      const uint64_t b0 = from >> 4, b1x = (to + 15) >> 4;                // the blocks the piece touches (a block shared with the
      if (b1x > b0) k_pack_reads<<<grid_for(b1x - b0), kBlock, 0, stream_>>>(db + (b0 << 4), total - (b0 << 4), b1x - b0, packed + b0);   // next piece is packed again there)
    };
    one(b1[k], b1[k + 1], d_b1, total1, packed1_);
    if (paired) one(b2[k], b2[k + 1], d_b2, total2, packed2_);
    HIP_CHECK(hipGetLastError());
  };
  // the post stage beside the next sub-batch's search: pays when the post stage is long (reads over families of strains:
  // 20-strain workload 2.97e8 -> 3.3e8 reads/s) and costs when it is short (cfg2: the search runs 20 % slower with anything
  // beside it, 6.8e8 -> 6.3e8).  The share of reads the last call folded by teams decides (CFR_TAIL_STREAM=0/1 forces it).
  // Round 4 measured it again (tools/dbg/ab_dyn_tail.sh, profiles/r4f_ab_tail_overlap.txt): with the search held to 4 blocks per CU
  // the post stage beside it wins on every workload - cfg2 14.16 -> 13.19 ms per step (search 9.97 -> 10.95 ms, the 2.8 ms of post
  // stage gone), pairs 29.3 -> 27.9 - where round 3 had seen a loss with 5 blocks per CU and no room left for the post stage's
  // waves.  So it is the default whenever a batch has more than one sub-batch (CFR_TAIL_STREAM=0/1 forces it).
  const bool tail_overlap = tail_overlap_mode_ >= 0 ? tail_overlap_mode_ != 0 : true;
  overlap_now_ = tail_overlap && one_launch && nsub > 1;
  // (only where the sub-batches' inputs are ready before the first search - resident reads, offsets in one pass - and chains go by wave tiles)
  // Measured (profiles/r5_ab_post_and_tiles.txt, section 7): no gain - cfg2 12.0 against 11.9 ms, pairs / 20 / 200 strains unchanged - so it is
  // OFF unless CFR_SEARCH_TWO=1 asks for it: the launches of a step do not lose their time in the drain of the one before.
  bool two_search = false;
  if (const char *e = dbg_env("CFR_SEARCH_TWO")) two_search = (overlap_now_ && caps_once && !pack_late && !dust_pieces && !search_v1_ && !view_.prot.enabled) && atoi(e) != 0;
  two_search_now_ = two_search;
  bool prep_waited = false;
  if (src && !one_launch) throw HipError{"streamed host inputs need the one-launch post stage", -4};
  bring_piece(0);

  // compact layout: the reads that do not fit it are also kept in the wide one (cfr_compact_wide_reads)
  WideSide wide_side{nullptr, nullptr, nullptr, nullptr, 0};
  wide_idx_.clear(); wide_res_.clear(); wide_match_.clear(); wide_total_ = 0;
  if (compact) {
    wide_side.cap = kWideSideCap;
    wide_side.idx = (uint32_t *)scratch(S_WIDEIDX, kWideSideCap * 4);
    wide_side.res = (cfr_result *)scratch(S_WIDERES, kWideSideCap * sizeof(cfr_result));
    wide_side.match = (cfr_match *)scratch(S_WIDEMATCH, kWideSideCap * stride * sizeof(cfr_match));
    wide_side.cnt = (unsigned long long *)scratch(S_WIDECNT, 8);
    HIP_CHECK(hipMemsetAsync(wide_side.cnt, 0, 8, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));              // (the compaction kernels may run on another stream)
  }
  // results / matches of piece k leave through the buffer pair of its parity while piece k+1 computes
  auto copy_out = [&](size_t k, const cfr_result *d_res, const cfr_match *d_match, uint64_t extent, const void *d_flag, unsigned long long *h_flag, hipStream_t st) {
    const size_t lo = pieces[k].first, cnt = pieces[k].second;
    const int par = (int)(k & 1);
    if (compact) {                          // the narrow layout is made on the device; what is copied out are its arrays
      cfr_result_compact *c_res = (cfr_result_compact *)scratch(par ? S_CRES1 : S_CRES, std::max(cnt, sb) * sizeof(cfr_result_compact));
      cfr_match_compact *c_match = (cfr_match_compact *)scratch(par ? S_CMATCH1 : S_CMATCH, (stride * std::max(cnt, sb) + 1) * sizeof(cfr_match_compact));
      k_compact_results<<<grid_for(cnt), kBlock, 0, st>>>(d_res, d_match, cnt, stride, c_res, c_match, (uint64_t)lo, wide_side);
      HIP_CHECK(hipGetLastError());
      d_res = reinterpret_cast<const cfr_result *>(c_res);
      d_match = reinterpret_cast<const cfr_match *>(c_match);
    }
    HIP_CHECK(hipEventRecord(ev_[7], st));
    HIP_CHECK(hipEventRecord(tail_done_[par], st));
    HIP_CHECK(hipStreamWaitEvent(copy_stream_, tail_done_[par], 0));
    static const bool no_copy = dbg_env("CFR_NO_COPY_OUT") && atoi(dbg_env("CFR_NO_COPY_OUT"));     // diagnosis: the step without its D2H
    if (!no_copy) {
      HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char *>(results) + lo * res_bytes, d_res, cnt * res_bytes, hipMemcpyDeviceToHost, copy_stream_));
      if (extent) HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char *>(matches) + stride * lo * match_bytes, d_match, extent * match_bytes, hipMemcpyDeviceToHost, copy_stream_));
    }
    if (d_flag) HIP_CHECK(hipMemcpyAsync(h_flag, d_flag, kCtlWords * 8, hipMemcpyDeviceToHost, copy_stream_));     // overflow flag, the two team counts, the reads k_post_fast left to k_adjust_tail
    HIP_CHECK(hipEventRecord(copy_done_[par], copy_stream_));
  };
  auto out_buffers = [&](size_t k, uint64_t extent, cfr_result *&d_res, cfr_match *&d_match, hipStream_t st) {
    const int par = (int)(k & 1);
    d_res = (cfr_result *)scratch(par ? S_RESULTS1 : S_RESULTS, std::max(pieces[k].second, sb) * sizeof(cfr_result));
    d_match = (cfr_match *)scratch(par ? S_MATCHES1 : S_MATCHES, (std::max<uint64_t>(extent, stride * sb) + 1) * sizeof(cfr_match));
    if (k >= 2) HIP_CHECK(hipStreamWaitEvent(st, copy_done_[par], 0));           // the copy that read this buffer pair
  };

  // ---- one launch per piece behind the search: everything is enqueued at once
  std::vector<size_t> todo;                 // pieces still to do
  for (size_t k = 0; k < nsub; ++k) todo.push_back(k);
  if (one_launch) {
    unsigned long long *pin = (unsigned long long *)pinned((2 + kCtlWords * kMaxSub + 3 * (kMaxSub + 1)) * 8);    // one block: the pointers below stay valid
    // per sub-batch three words, fetched with ONE copy behind its post stage: pool-overflow flag, reads the small teams of
    // k_tail_heavy folded, reads the large teams folded (= ctl[1..3] of the sub-batch's control block)
    unsigned long long *pctl = pin + 2;
    for (size_t k = 0; k < kCtlWords * kMaxSub; ++k) pctl[k] = 0;
    if (!pool_cap_) pool_cap_ = std::max<uint64_t>(8ull * sb, 1ull << 20);
    const uint64_t pool_limit = std::max<uint64_t>(256ull * sb, 1ull << 26);      // ~10 GB at the default sub-batch
    for (int attempt = 0; attempt < 4 && !todo.empty(); ++attempt) {
      TailEntry *pool_e = (TailEntry *)scratch(S_POOL_E, pool_cap_ * sizeof(TailEntry));
      uint64_t *pool_v = (uint64_t *)scratch(S_POOL_V, pool_cap_ * 8);
      for (size_t k : todo) {
        const size_t lo = pieces[k].first, cnt = pieces[k].second;
        pctl[kCtlWords * k] = 0;
        ev_ = evs_[k];
        if (attempt == 0) { bring_piece(k); pack_piece(k); }
        // the post stage runs on its own stream: it is a chain of dependent gathers per read (4 fabric requests per read,
        // waves waiting two thirds of the time) and hides under the search of the next sub-batch, which is bound by the
        // fabric's request rate.  Two sets of search outputs: search k + 2 waits for the post stage of k.
        const int par = tail_overlap ? (int)(k & 1) : 0;
        hipStream_t ts = tail_overlap ? tail_stream_ : stream_;
        // two_search (round 5): the searches of odd sub-batches on a second stream, so that search k + 1 moves into the CUs as the blocks of
        // search k finish (a persistent grid ends with its slowest block; chains are handed out by wave tiles, so a block that starts late
        // simply takes fewer).  Everything else of a sub-batch follows its search as before.
        hipStream_t ss = two_search && par ? search2_stream_ : stream_;
        search_stream_ = ss;
        if (two_search && par && !prep_waited) { HIP_CHECK(hipEventRecord(prep_done_, stream_)); HIP_CHECK(hipStreamWaitEvent(search2_stream_, prep_done_, 0)); prep_waited = true; }
        if (tail_overlap) HIP_CHECK(hipStreamWaitEvent(ss, tail_done_[par], 0));
        if (pack_rest_pending && k >= 1) { HIP_CHECK(hipStreamWaitEvent(stream_, copied_[0], 0)); pack_rest_pending = false; }
        pre_hit_off_ = hit_all ? hit_all + lo : nullptr;
        pre_hit_base_ = hit_all ? hbase[k] : 0;
        const SearchBuf sbuf = launch_search(d_b1, d_o1 + lo, d_b2, paired ? d_o2 + lo : nullptr, cnt, pt1[k], pt2[k], par);
        pre_hit_off_ = nullptr;
        search_stream_ = nullptr;
        for (int e : {8, 3, 4, 5, 6}) HIP_CHECK(hipEventRecord(ev_[e], ss));
        if (tail_overlap) {
          HIP_CHECK(hipEventRecord(search_done_[par], ss));
          HIP_CHECK(hipStreamWaitEvent(ts, search_done_[par], 0));
        }
        unsigned long long *ctl = (unsigned long long *)scratch((k & 1) ? S_POOLCTL1 : S_POOLCTL, 64);    // pool cursor, overflow flag, heavy reads (two tiers), slow reads
        cfr_result *d_res;
        cfr_match *d_match;
        out_buffers(k, stride * cnt, d_res, d_match, ts);               // (also orders the memset below behind the copy of ctl)
        HIP_CHECK(hipMemsetAsync(ctl, 0, 64, ts));
        // reads whose fold does not fit the registers (many located rows) are listed and folded by teams of lanes afterwards
        uint64_t *heavy = team_tail_ ? (uint64_t *)scratch(par ? S_HEAVY1 : S_HEAVY, std::max(sb, cnt) * 32) : nullptr;
        // reads with more located rows than this go straight to the large teams (distinct ids <= rows: 48 fit the small teams' table; up to
        // 64 rows most reads still do - the same strains in every hit - and the few that do not are handed on): 200 strains per species
        // 96 -> 73 ms per step, 20 strains unchanged (profiles/r4n_strains_traces.txt).  0 = every read starts with the small teams
        static const uint32_t direct_rows = dbg_env("CFR_HEAVY_DIRECT_ROWS") ? (uint32_t)atoi(dbg_env("CFR_HEAVY_DIRECT_ROWS")) : 64u;
        uint64_t *heavy2 = team_tail_ ? (uint64_t *)scratch(par ? S_HEAVYB1 : S_HEAVYB, std::max(sb, cnt) * 32) : nullptr;
        // beside a search the post stage gets a few blocks per CU (grid-stride inside), alone the whole sub-batch at once
        const unsigned tail_grid = tail_overlap && tail_blocks_per_cu_ ? std::min<unsigned>(grid_for(cnt), (unsigned)(num_cus_ * tail_blocks_per_cu_)) : grid_for(cnt);
        // pairs: the image description behind a pointer (k_adjust_tail_p: no 1 KB copy into every lane's scratch); CFR_TAIL_VIEW_PTR=0: by value
        static const bool view_ptr = !(dbg_env("CFR_TAIL_VIEW_PTR") && atoi(dbg_env("CFR_TAIL_VIEW_PTR")) == 0);
        // CFR_POST_FAST=1 (round 5): the common read's post stage by k_post_fast - registers only, 94 of them, no scratch, so
        // that its waves fit beside four waves of the search - which lists the reads that need more: team folds, or the full lane program of
        // k_adjust_tail, which then runs over that list.  Built, parity-green (tests/test_gpu_variants.py) and NOT the default, because
        // it measures slower where it was meant to help: cfg2 12.1 against 11.85 ms per step, pairs 24.4 against 23.9 (20 strains 27.1
        // against 27.6).  What the measurements of profiles/r5_ab_post_and_tiles.txt say about the premise: with NO post stage at all the
        // searches of a step take 9.4 ms against 9.5 (the "8.7 ms alone" is one 2 M-read launch, not ten launches of a step), so the post
        // stage costs the search 0.15 ms, not 0.9; a post stage that co-resides MORE slows the search more (pairs: search 16.9 -> 19.9 ms
        // with the lean kernel beside it, 18.6 -> 24.9 with the post stage in the highest stream priority) - the 128-register kernel that
        // only gets the slots the search leaves is the better neighbour; and the ~70 reads per sub-batch whose boundary adjustment walks
        // the index character by character take ~0.1 ms whatever their number, which one kernel hides among 1.25 M other reads and a
        // second launch does not.
        static const bool post_fast = dbg_env("CFR_POST_FAST") && atoi(dbg_env("CFR_POST_FAST")) != 0;
        uint32_t *slow = nullptr;
        unsigned long long *slow_cnt = nullptr;
        unsigned fat_grid = tail_grid;
        static const bool skip_post = dbg_env("CFR_SKIP_POST") && atoi(dbg_env("CFR_SKIP_POST")) != 0;   // diagnosis only: NO post stage (results are not computed) - what the searches cost with nothing beside them
        if (skip_post) { copy_out(k, d_res, d_match, stride * cnt, ctl + 1, &pctl[kCtlWords * k], ts); if (attempt == 0 && k + 1 < nsub) bring_piece(k + 1); continue; }
        // (the last sub-batch has no search beside it and nothing behind it to hide the second launch - a handful of reads whose boundary
        // adjustment walks the index character by character, ~0.1 ms whatever their number: there the one kernel takes all reads)
        static const int post_fast_last = dbg_env("CFR_POST_FAST_LAST") ? atoi(dbg_env("CFR_POST_FAST_LAST")) : 0;
        if (post_fast && (k + 1 < nsub || post_fast_last || !tail_overlap)) {
          slow = (uint32_t *)scratch(par ? S_SLOW1 : S_SLOW, std::max(sb, cnt) * 4);
          slow_cnt = ctl + 4;
          if (paired) k_post_fast<4><<<tail_grid, kBlock, 0, ts>>>(view_, d_o1 + lo, d_o2 + lo, cnt, sbuf.hit_off, sbuf.raw, sbuf.chain_cnt, d_res, d_match, stride, stride * lo,
                                                               heavy, ctl + 2, heavy2, ctl + 3, direct_rows, slow, slow_cnt);
          else k_post_fast<2><<<tail_grid, kBlock, 0, ts>>>(view_, d_o1 + lo, nullptr, cnt, sbuf.hit_off, sbuf.raw, sbuf.chain_cnt, d_res, d_match, stride, stride * lo,
                                                          heavy, ctl + 2, heavy2, ctl + 3, direct_rows, slow, slow_cnt);
          fat_grid = std::min<unsigned>(grid_for(cnt), (unsigned)(num_cus_ * 2));          // (its list is short - or everything, with hits on both strands throughout)
        }
        if (paired && view_ptr && d_view_) k_adjust_tail_p<4><<<fat_grid, kBlock, 0, ts>>>(d_view_, d_b1, d_o1 + lo, d_b2, d_o2 + lo, cnt, sbuf.hit_off, sbuf.raw, sbuf.chain_cnt,
                                                                      pool_e, pool_v, ctl, pool_cap_, (uint32_t *)(ctl + 1), d_res, d_match, stride, stride * lo, heavy, ctl + 2, heavy2, ctl + 3, direct_rows, slow, slow_cnt);
        else if (paired) k_adjust_tail<4><<<fat_grid, kBlock, 0, ts>>>(view_, d_b1, d_o1 + lo, d_b2, d_o2 + lo, cnt, sbuf.hit_off, sbuf.raw, sbuf.chain_cnt,
                                                                      pool_e, pool_v, ctl, pool_cap_, (uint32_t *)(ctl + 1), d_res, d_match, stride, stride * lo, heavy, ctl + 2, heavy2, ctl + 3, direct_rows, slow, slow_cnt);
        else k_adjust_tail<2><<<fat_grid, kBlock, 0, ts>>>(view_, d_b1, d_o1 + lo, nullptr, nullptr, cnt, sbuf.hit_off, sbuf.raw, sbuf.chain_cnt,
                                                               pool_e, pool_v, ctl, pool_cap_, (uint32_t *)(ctl + 1), d_res, d_match, stride, stride * lo, heavy, ctl + 2, heavy2, ctl + 3, direct_rows, slow, slow_cnt);
        if (heavy) {
          // two tiers: teams of 8 lanes with 48 table entries, then - for the reads whose ids do not fit (hundreds of strains per
          // species) - teams of 32 lanes with 192; what is left after that takes the single-lane form with pool scratch
          const unsigned per_cu = (unsigned)(tail_overlap && tail_blocks_per_cu_ ? std::min(5, 2 * tail_blocks_per_cu_) : 5);
          const unsigned hb = std::min<unsigned>((unsigned)((cnt + kTeamsPerBlock - 1) / kTeamsPerBlock), (unsigned)num_cus_ * per_cu);
          const unsigned hb2 = std::min<unsigned>((unsigned)((cnt + kTeams2PerBlock - 1) / kTeams2PerBlock), (unsigned)num_cus_ * std::min(per_cu, 4u));
          if (paired) {
            k_tail_heavy<4, kTeam, kTeamSlots, kTeamsPerBlock, kTeamMaxEntries><<<hb, kTeam * kTeamsPerBlock, 0, ts>>>(
                view_, d_o1 + lo, d_o2 + lo, sbuf.hit_off, sbuf.raw, heavy, ctl + 2, pool_e, pool_v, ctl, pool_cap_, (uint32_t *)(ctl + 1), d_res, d_match, stride, stride * lo, heavy2, ctl + 3);
            k_tail_heavy<4, kTeam2, kTeam2Slots, kTeams2PerBlock, kTeam2MaxEntries><<<hb2, kTeam2 * kTeams2PerBlock, 0, ts>>>(
                view_, d_o1 + lo, d_o2 + lo, sbuf.hit_off, sbuf.raw, heavy2, ctl + 3, pool_e, pool_v, ctl, pool_cap_, (uint32_t *)(ctl + 1), d_res, d_match, stride, stride * lo, nullptr, nullptr);
          } else {
            k_tail_heavy<2, kTeam, kTeamSlots, kTeamsPerBlock, kTeamMaxEntries><<<hb, kTeam * kTeamsPerBlock, 0, ts>>>(
                view_, d_o1 + lo, nullptr, sbuf.hit_off, sbuf.raw, heavy, ctl + 2, pool_e, pool_v, ctl, pool_cap_, (uint32_t *)(ctl + 1), d_res, d_match, stride, stride * lo, heavy2, ctl + 3);
            k_tail_heavy<2, kTeam2, kTeam2Slots, kTeams2PerBlock, kTeam2MaxEntries><<<hb2, kTeam2 * kTeams2PerBlock, 0, ts>>>(
                view_, d_o1 + lo, nullptr, sbuf.hit_off, sbuf.raw, heavy2, ctl + 3, pool_e, pool_v, ctl, pool_cap_, (uint32_t *)(ctl + 1), d_res, d_match, stride, stride * lo, nullptr, nullptr);
          }
        }
        HIP_CHECK(hipGetLastError());
        copy_out(k, d_res, d_match, stride * cnt, ctl + 1, &pctl[kCtlWords * k], ts);
        if (attempt == 0) last_stats.n_chains += cnt * (size_t)(paired ? 4 : 2);
        if (attempt == 0 && k + 1 < nsub) bring_piece(k + 1);          // the host copies the next piece while this one computes
      }
      HIP_CHECK(hipStreamSynchronize(stream_));
      if (two_search) HIP_CHECK(hipStreamSynchronize(search2_stream_));
      HIP_CHECK(hipStreamSynchronize(tail_stream_));
      HIP_CHECK(hipStreamSynchronize(copy_stream_));
      if (attempt == 0) for (size_t k = 0; k < nsub; ++k) { ev_ = evs_[k]; finish_stats(true); }
      if (attempt == 0) {                    // what the next call's schedule goes by: the share of reads with a team fold
        unsigned long long hv = 0, slow_reads = 0;
        for (size_t k = 0; k < nsub; ++k) { hv += pctl[kCtlWords * k + 1] + pctl[kCtlWords * k + 2]; slow_reads += pctl[kCtlWords * k + 3]; }     // (a read the small teams handed on counts twice: it is a threshold)
        heavy_frac_ = (double)hv / (double)n;
        last_slow_reads_ = slow_reads; last_team_reads_ = hv;
        if (dbg_env("CFR_POST_STATS")) {                 // diagnostics: where the reads of this call went
          unsigned long long t1 = 0, t2 = 0;
          for (size_t k = 0; k < nsub; ++k) { t1 += pctl[kCtlWords * k + 1]; t2 += pctl[kCtlWords * k + 2]; }
          fprintf(stderr, "[cfr] post stage of %zu reads: %llu left to k_adjust_tail, %llu folded by small teams, %llu by large teams\n", n, slow_reads, t1, t2);
        }
      }
      std::vector<size_t> again;
      for (size_t k : todo) if (pctl[kCtlWords * k]) again.push_back(k);              // the scratch pool ran dry in these
      todo.swap(again);
      if (todo.empty() || pool_cap_ >= pool_limit || dbg_env("CFR_POOL_CAP")) break;
      pool_cap_ = std::min(pool_cap_ * 4, pool_limit);                   // kept for the calls that follow: the workload needs it
    }
  }
  const bool repeated = one_launch && !todo.empty();

  // ---- multi-kernel form: search, adjust/select, (compact, rows, locate,) tail; one or two 8-byte host syncs per piece
  if (pack_rest_pending) { HIP_CHECK(hipStreamWaitEvent(stream_, copied_[0], 0)); pack_rest_pending = false; }
  for (size_t k : todo) {
    const size_t lo = pieces[k].first, cnt = pieces[k].second;
    ev_ = evs_[k];
    Pipe p;
    run_device_stages(d_b1, d_o1 + lo, d_b2, paired ? d_o2 + lo : nullptr, cnt, pt1[k], pt2[k], true, p, nullptr, fused);
    const uint64_t extent = stride ? stride * cnt : p.nrows;
    if (!stride) {
      if (match_extent) *match_extent = extent;
      if (extent > match_cap) { HIP_CHECK(hipStreamSynchronize(stream_)); throw CapacityError{"match buffer too small"}; }
    }
    TailEntry *entries = (TailEntry *)scratch(S_ENTRIES, (p.nrows + 1) * sizeof(TailEntry));
    cfr_result *d_res;
    cfr_match *d_match;
    out_buffers(k, extent, d_res, d_match, stream_);
    if (fused) k_tail<true><<<grid_for(cnt), kBlock, 0, stream_>>>(view_, cnt, d_o1 + lo, paired ? d_o2 + lo : nullptr, p.hit_off, p.fin_off, p.hits,
                                                                  p.row_off, p.vals, entries, d_res, d_match, stride, stride * lo);
    else k_tail<false><<<grid_for(cnt), kBlock, 0, stream_>>>(view_, cnt, d_o1 + lo, paired ? d_o2 + lo : nullptr, p.fin_off, nullptr, p.hits,
                                                              p.row_off, p.vals, entries, d_res, d_match, stride, stride * lo);
    HIP_CHECK(hipGetLastError());
    copy_out(k, d_res, d_match, extent, nullptr, nullptr, stream_);
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  HIP_CHECK(hipStreamSynchronize(copy_stream_));
  if (!one_launch) for (size_t k = 0; k < nsub; ++k) { ev_ = evs_[k]; finish_stats(true); }
  if (compact) {                            // the wide form of the flagged reads (a handful, if any)
    unsigned long long cntw = 0;
    HIP_CHECK(hipMemcpy(&cntw, wide_side.cnt, 8, hipMemcpyDeviceToHost));
    wide_total_ = cntw;
    const size_t take = (size_t)std::min<unsigned long long>(cntw, kWideSideCap);
    if (take) {
      wide_idx_.resize(take); wide_res_.resize(take); wide_match_.resize(take * stride);
      HIP_CHECK(hipMemcpy(wide_idx_.data(), wide_side.idx, take * 4, hipMemcpyDeviceToHost));
      HIP_CHECK(hipMemcpy(wide_res_.data(), wide_side.res, take * sizeof(cfr_result), hipMemcpyDeviceToHost));
      HIP_CHECK(hipMemcpy(wide_match_.data(), wide_side.match, take * stride * sizeof(cfr_match), hipMemcpyDeviceToHost));
      // A sub-batch that was computed again (its scratch pool ran dry: a second attempt, or the multi-kernel form) appended its
      // flagged reads again; the entries stand in the order they were made, so the LAST one of a read is the one that belongs to
      // the results the caller got.  (match_begin points into wide_match_, which stays as it is.)
      if (cntw <= kWideSideCap) {
        std::vector<uint32_t> order(take);
        for (size_t j = 0; j < take; ++j) order[j] = (uint32_t)j;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return wide_idx_[a] < wide_idx_[b]; });
        std::vector<uint32_t> idx2;
        std::vector<cfr_result> res2;
        for (size_t j = 0; j < take; ++j) if (j + 1 == take || wide_idx_[order[j + 1]] != wide_idx_[order[j]]) { idx2.push_back(wide_idx_[order[j]]); res2.push_back(wide_res_[order[j]]); }
        wide_idx_.swap(idx2);
        wide_res_.swap(res2);
        wide_total_ = wide_idx_.size();
      }
    }
  }
  if (!repeated) {                          // wall time of the device work: first event of the first piece to the last of the last
    float t = 0;
    (void)hipEventElapsedTime(&t, evs_[0][0], evs_[nsub - 1][7]);
    last_stats.total_ms = t;
  }
  ev_ = evs_[0];
  overlap_now_ = false;                     // (the cap on the search's blocks belongs to this call's schedule, not to the entries that follow)
  if (!expand_end()) {
    // the pool of --expand-taxid records was too small: once more with one that holds what this run counted (it counted every record,
    // those of re-run sub-batches twice, so the second run fits).  Bounded: a pool that still does not fit after three runs is an error.
    if (++exp_attempt_ > 3) { exp_attempt_ = 0; throw HipError{"the pool of --expand-taxid records did not settle in three runs of the batch", -6}; }
    ++last_stats_exp_retries_;
    classify_device(orig_b1, d_o1, orig_b2, d_o2, n, total1, total2, results, matches, match_cap, match_extent, src, compact);
  } else exp_attempt_ = 0;
}

void DeviceIndex::classify_host(const uint8_t *b1, const uint64_t *o1, const uint8_t *b2, const uint64_t *o2, size_t n,
                                cfr_result *results, cfr_match *matches, size_t match_cap, size_t *match_extent) {
  HIP_CHECK(hipSetDevice(device_));
  static const bool stream_inputs = !(dbg_env("CFR_STREAM_INPUTS") && atoi(dbg_env("CFR_STREAM_INPUTS")) == 0);
  if (n && stream_inputs && !search_v1_ && view_.max_result > 0 && one_launch_ready()) {
    // streamed form: only the offsets go up front; the bases of sub-batch k+1 are copied (h2d stream) while sub-batch k computes
    const HostSrc src{b1, o1, b2, o2};
    const uint64_t t1 = o1[n], t2 = b2 ? o2[n] : 0;
    uint8_t *d_b1 = (uint8_t *)scratch(S_IN_B1, t1 + 16);
    uint64_t *d_o1 = (uint64_t *)scratch(S_IN_O1, (n + 1) * 8);
    // the offsets travel on the same stream as the bases: the SDUST kernel of a piece runs there and reads them, and the
    // main stream only touches a piece behind that stream's event
    HIP_CHECK(hipMemcpyAsync(d_o1, o1, (n + 1) * 8, hipMemcpyHostToDevice, h2d_stream_));
    uint8_t *d_b2 = nullptr;
    uint64_t *d_o2 = nullptr;
    if (b2) {
      d_b2 = (uint8_t *)scratch(S_IN_B2, t2 + 16);
      d_o2 = (uint64_t *)scratch(S_IN_O2, (n + 1) * 8);
      HIP_CHECK(hipMemcpyAsync(d_o2, o2, (n + 1) * 8, hipMemcpyHostToDevice, h2d_stream_));
    }
    classify_device(d_b1, d_o1, d_b2, d_o2, n, t1, t2, results, matches, match_cap, match_extent, &src);
    return;
  }
  Staged st = stage_inputs(b1, o1, b2, o2, n);
  classify_device(st.b1, st.o1, st.b2, st.o2, n, st.t1, st.t2, results, matches, match_cap, match_extent);
}

void DeviceIndex::classify_host_packed(const uint64_t *p1, const uint64_t *o1, const uint64_t *p2, const uint64_t *o2, size_t n,
                                       cfr_result *results, cfr_match *matches, size_t match_cap, size_t *match_extent) {
  HIP_CHECK(hipSetDevice(device_));
  if (n == 0) { if (match_extent) *match_extent = 0; last_stats = cfr_batch_stats{}; return; }
  const uint64_t t1 = o1[n], t2 = p2 ? o2[n] : 0;
  uint8_t *d_b1 = (uint8_t *)scratch(S_IN_B1, t1 + 32);
  uint64_t *d_o1 = (uint64_t *)scratch(S_IN_O1, (n + 1) * 8);
  uint8_t *d_b2 = nullptr;
  uint64_t *d_o2 = nullptr;
  if (p2) {
    d_b2 = (uint8_t *)scratch(S_IN_B2, t2 + 32);
    d_o2 = (uint64_t *)scratch(S_IN_O2, (n + 1) * 8);
  }
  static const bool stream_inputs = !(dbg_env("CFR_STREAM_INPUTS") && atoi(dbg_env("CFR_STREAM_INPUTS")) == 0);
  if (stream_inputs && !search_v1_ && view_.max_result > 0 && one_launch_ready()) {
    // streamed like classify_host: the offsets up front, the blocks of sub-batch k + 1 under the kernels of sub-batch k
    HIP_CHECK(hipMemcpyAsync(d_o1, o1, (n + 1) * 8, hipMemcpyHostToDevice, h2d_stream_));
    if (p2) HIP_CHECK(hipMemcpyAsync(d_o2, o2, (n + 1) * 8, hipMemcpyHostToDevice, h2d_stream_));
    HostSrc src{nullptr, o1, nullptr, o2};
    src.p1 = p1;
    src.p2 = p2;
    if (dust_ && !view_.prot.enabled) {
      src.stage1 = (uint64_t *)scratch(S_P0, ((t1 + 15) / 16 + 1) * 8);
      if (p2) src.stage2 = (uint64_t *)scratch(S_P1, ((t2 + 15) / 16 + 1) * 8);
    }
    classify_device(d_b1, d_o1, d_b2, d_o2, n, t1, t2, results, matches, match_cap, match_extent, &src);
    return;
  }
  // the other forms of the path (protein index, row-space matches, kernel v1): the whole batch is unpacked first
  auto whole = [&](const uint64_t *hp, uint8_t *db, uint64_t total, size_t slot) {
    const uint64_t nb = (total + 15) >> 4;
    uint64_t *tmp = (uint64_t *)scratch(slot, (nb + 1) * 8);
    if (nb) {
      HIP_CHECK(hipMemcpyAsync(tmp, hp, nb * 8, hipMemcpyHostToDevice, stream_));
      k_unpack_reads<<<grid_for(nb), kBlock, 0, stream_>>>(tmp, nb, db, 0, total);
      HIP_CHECK(hipGetLastError());
    }
  };
  whole(p1, d_b1, t1, S_P0);
  HIP_CHECK(hipMemcpyAsync(d_o1, o1, (n + 1) * 8, hipMemcpyHostToDevice, stream_));
  if (p2) {
    whole(p2, d_b2, t2, S_P1);
    HIP_CHECK(hipMemcpyAsync(d_o2, o2, (n + 1) * 8, hipMemcpyHostToDevice, stream_));
  }
  classify_device(d_b1, d_o1, d_b2, d_o2, n, t1, t2, results, matches, match_cap, match_extent);
}

void *DeviceIndex::pinned(size_t bytes) {
  if (pinned_cap_ < bytes) {
    if (pinned_) (void)hipHostFree(pinned_);
    pinned_ = nullptr;
    HIP_CHECK(hipHostMalloc(&pinned_, std::max<size_t>(bytes, 4096), hipHostMallocDefault));
    pinned_cap_ = std::max<size_t>(bytes, 4096);
  }
  return pinned_;
}

void *host_alloc_pinned(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void host_free_pinned(void *p) { if (p) (void)hipHostFree(p); }

}  // namespace cfr
