// cfr_hostlink.hpp — the PCIe link for callers that hand over ordinary (pageable) host buffers.
//
// The reference keeps its read batches in malloc'ed memory (ReadFiles.hpp, the batch buffers of CentrifugerClass.cpp:386-431), so a
// drop-in binding passes pageable pointers.  hipMemcpyAsync on such a pointer is staged by the runtime through ONE thread and
// blocks the caller: 16 GB/s for both directions together, a quarter of what the same entry reaches from cfr_host_alloc
// memory (profiles/r3m_bench.json: 8.1e7 against 3.3e8 reads/s).  HostLink does the staging itself: a few pinned chunks per
// direction, filled / emptied by a small crew of threads, the DMA of chunk i under the memcpy of chunk i + 1.
//
//   h2d(dst, src, bytes, stream): returns when the caller's bytes have left src (the DMAs may still be in flight on stream)
//   d2h(dst, src, bytes, stream): enqueues device -> pinned chunk copies on stream; the bytes reach dst in drain()
//   drain(all): empties the chunks whose DMA has finished (all = wait for every one); call with all = true before returning
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace cfr {

// N - 1 helper threads + the caller copy one block of memory in N slices
class CopyCrew {
 public:
  explicit CopyCrew(int threads) {
    for (int i = 1; i < threads; ++i) th_.emplace_back([this, i] { work(i); });
  }
  ~CopyCrew() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  CopyCrew(const CopyCrew &) = delete;
  CopyCrew &operator=(const CopyCrew &) = delete;
  void copy(void *dst, const void *src, size_t bytes) {
    const size_t n = th_.size() + 1;
    if (bytes < (1u << 20) || n == 1) { memcpy(dst, src, bytes); return; }
    {
      std::lock_guard<std::mutex> g(m_);
      dst_ = (char *)dst; src_ = (const char *)src; bytes_ = bytes; left_ = (int)th_.size(); ++gen_;
    }
    cv_.notify_all();
    slice(0);
    std::unique_lock<std::mutex> g(m_);
    done_cv_.wait(g, [this] { return left_ == 0; });
  }

 private:
  void slice(size_t i) {
    const size_t n = th_.size() + 1;
    const size_t per = ((bytes_ + n - 1) / n + 4095) & ~(size_t)4095;
    const size_t a = std::min(bytes_, i * per), b = std::min(bytes_, a + per);
    if (b > a) memcpy(dst_ + a, src_ + a, b - a);
  }
  void work(int i) {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
      }
      slice((size_t)i);
      bool last;
      { std::lock_guard<std::mutex> g(m_); last = --left_ == 0; }
      if (last) done_cv_.notify_one();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_cv_;
  uint64_t gen_ = 0;
  int left_ = 0;
  bool stop_ = false;
  char *dst_ = nullptr;
  const char *src_ = nullptr;
  size_t bytes_ = 0;
};

class HostLink {
 public:
  static constexpr size_t kChunk = 32u << 20;
  static constexpr int kUp = 3, kDown = 6;
  explicit HostLink(int threads) : crew_(threads) {}
  ~HostLink() {
    for (auto &c : up_) { if (c.ev) (void)hipEventDestroy(c.ev); if (c.p) (void)hipHostFree(c.p); }
    for (auto &c : down_) { if (c.ev) (void)hipEventDestroy(c.ev); if (c.p) (void)hipHostFree(c.p); }
  }
  HostLink(const HostLink &) = delete;
  HostLink &operator=(const HostLink &) = delete;

  // true for memory the runtime has never seen (malloc / new / numpy); false for cfr_host_alloc / hipHostRegister'ed memory
  static bool pageable(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return true; }
    return a.type == hipMemoryTypeUnregistered;
  }

  hipError_t h2d(void *dst, const void *src, size_t bytes, hipStream_t st) {
    for (size_t off = 0; off < bytes; off += kChunk) {
      const size_t len = std::min(kChunk, bytes - off);
      Chunk &c = up_[up_next_];
      up_next_ = (up_next_ + 1) % kUp;
      if (hipError_t e = ready(c)) return e;
      if (c.used) { if (hipError_t e = hipEventSynchronize(c.ev)) return e; }
      crew_.copy(c.p, (const char *)src + off, len);
      if (hipError_t e = hipMemcpyAsync((char *)dst + off, c.p, len, hipMemcpyHostToDevice, st)) return e;
      if (hipError_t e = hipEventRecord(c.ev, st)) return e;
      c.used = true;
    }
    return hipSuccess;
  }

  hipError_t d2h(void *dst, const void *src, size_t bytes, hipStream_t st) {
    for (size_t off = 0; off < bytes; off += kChunk) {
      const size_t len = std::min(kChunk, bytes - off);
      if ((int)pend_.size() == kDown) { if (hipError_t e = drain_one(true)) return e; }
      int idx = -1;
      for (int i = 0; i < kDown; ++i) if (!down_[i].used) { idx = i; break; }
      Chunk &c = down_[idx];
      if (hipError_t e = ready(c)) return e;
      if (hipError_t e = hipMemcpyAsync(c.p, (const char *)src + off, len, hipMemcpyDeviceToHost, st)) return e;
      if (hipError_t e = hipEventRecord(c.ev, st)) return e;
      c.used = true;
      pend_.push_back(Pend{idx, (char *)dst + off, len});
    }
    return hipSuccess;
  }

  hipError_t drain(bool all) {
    while (!pend_.empty()) {
      if (!all && hipEventQuery(down_[pend_.front().buf].ev) != hipSuccess) { (void)hipGetLastError(); break; }
      if (hipError_t e = drain_one(all)) return e;
    }
    return hipSuccess;
  }
  // after a failed call: forget what was in flight (the streams have been synchronised by the caller's error path or will be)
  void reset() {
    pend_.clear();
    for (auto &c : down_) c.used = false;
  }

 private:
  struct Chunk { void *p = nullptr; hipEvent_t ev = nullptr; bool used = false; };
  struct Pend { int buf; char *dst; size_t bytes; };
  hipError_t ready(Chunk &c) {
    if (!c.p) { if (hipError_t e = hipHostMalloc(&c.p, kChunk, hipHostMallocDefault)) { c.p = nullptr; return e; } }
    if (!c.ev) { if (hipError_t e = hipEventCreateWithFlags(&c.ev, hipEventDisableTiming)) { c.ev = nullptr; return e; } }
    return hipSuccess;
  }
  hipError_t drain_one(bool wait) {
    const Pend p = pend_.front();
    Chunk &c = down_[p.buf];
    if (wait) { if (hipError_t e = hipEventSynchronize(c.ev)) return e; }
    crew_.copy(p.dst, c.p, p.bytes);
    c.used = false;
    pend_.pop_front();
    return hipSuccess;
  }
  CopyCrew crew_;
  Chunk up_[kUp], down_[kDown];
  int up_next_ = 0;
  std::deque<Pend> pend_;
};

}  // namespace cfr
