// cfr_build_sa.hip — suffix array of a DNA text in HBM and what the index writer reads off it (gfx950).
//
// Replaces, for the sizes this repository measures on, the reference's blockwise suffix sorter
// (compactds/FMBuilder.hpp:444-811, SuffixArrayGenerator.hpp) - not its algorithm, its product: the plain lexicographic
// order of the suffixes of the concatenated genomes, a proper prefix first (FMBuilder.hpp:209-254).
//
// Algorithm (prefix doubling over sorted groups, Larsson-Sadakane style, sized for 288 GB of HBM):
//   text   2 bits per symbol, first symbol of a word in its top bits: the 32-mer at any position is two shifts
//   SA     5 bytes per row, RANK 5 bytes per text position (= SA index of the head of the position's group), head bit per row
//   phase 1   suffixes are sorted by their first 32 symbols in chunks of <= 2^28: the 14-bit top of the key cuts the key
//             space into contiguous chunks, each chunk is collected from the text, radix-sorted (hipCUB) and committed
//   phase 2   round h = 32, 64, ...: every group that is not a single row yet is sorted by RANK[pos + h]; only those rows
//             move (singletons are final).  Work is cut into spans of <= 2^28 rows that end on group boundaries; the keys
//             of a span are gathered before any of its ranks change, and a group is never split across spans, so the
//             in-place rank refinement between spans is the one Larsson-Sadakane rely on.
//   A suffix shorter than the comparison depth is padded with symbol 0; among the members of one group the padded ones
//   come first, shortest first (key n-1-pos < h <= every in-range key RANK + h): a proper prefix sorts first.
// Memory: 10.4 bytes per symbol + ~15 GB of chunk buffers: 8 Gbp = 98 GB, 16 Gbp = 182 GB.  Texts whose SA + RANK do not fit
// (above ~24 Gbp on a 288 GB device) keep the SA in HOST memory: RANK, text and head bits stay in HBM (5.4 bytes per symbol),
// a phase-1 chunk is copied out once it is committed and a phase-2 span is copied in, refined and copied back only when it
// still holds unresolved groups.  The composite key (group offset : second key) is 64 bits wide for n + 2^27 < 2^36.
#include "cfr_build.hpp"

#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>

#include "cfr_device.hpp"      // HipError

namespace cfr {
namespace {

inline void hip_check_b(hipError_t e, const char *what) {
  if (e != hipSuccess) throw HipError{std::string("index build: ") + what + ": " + hipGetErrorString(e), (int)e};
}
#define BCHECK(x) hip_check_b((x), #x)

struct __attribute__((packed)) U40b { uint32_t lo; uint8_t hi; };
__device__ __forceinline__ void put40(uint8_t *tab, uint64_t i, uint64_t val) {
  U40b e; e.lo = (uint32_t)val; e.hi = (uint8_t)(val >> 32);
  reinterpret_cast<U40b *>(tab)[i] = e;
}
__device__ __forceinline__ uint64_t get40(const uint8_t *tab, uint64_t i) {
  const U40b e = reinterpret_cast<const U40b *>(tab)[i];
  return (uint64_t)e.lo | ((uint64_t)e.hi << 32);
}

constexpr uint32_t kBinBits = 14;                 // phase-1 chunks are unions of 14-bit key prefixes
constexpr uint64_t kChunkDefault = 1ull << 28;    // rows per sort
// composite key of phase 2 = group offset in the span (<= 28 bits, top) : second key (rank_bits = bits of n + 2^27, low)

// ---- text ------------------------------------------------------------------------------------------------------------
// ASCII text (a piece starting at a multiple of 32) -> packed words; *bad is raised when a character is not an upper-case A,C,G,T
__global__ void k_pack_text(const uint8_t *text, uint64_t count, uint64_t *words, unsigned int *bad) {
  const uint64_t wI = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (wI * 32 >= count) return;
  uint64_t w = 0;
  const uint64_t base = wI * 32;
  for (uint32_t k = 0; k < 32 && base + k < count; ++k) {
    const uint32_t ch = text[base + k];
    const uint32_t c = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
    if (c > 3u) *bad = 1u;
    w |= (uint64_t)(c & 3u) << (62 - 2 * k);
  }
  words[wI] = w;
}
__device__ __forceinline__ uint64_t key32(const uint64_t *T, uint64_t pos) {
  const uint32_t o = (uint32_t)pos & 31u;
  const uint64_t w0 = T[pos >> 5];
  if (o == 0) return w0;
  return (w0 << (2 * o)) | (T[(pos >> 5) + 1] >> (64 - 2 * o));
}
__device__ __forceinline__ uint32_t sym_at(const uint64_t *T, uint64_t pos) { return (uint32_t)(T[pos >> 5] >> (62 - 2 * (pos & 31))) & 3u; }

// ---- phase 1 ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hist(const uint64_t *T, uint64_t n, unsigned long long *hist) {
  __shared__ uint32_t h[1u << kBinBits];
  for (uint32_t k = threadIdx.x; k < (1u << kBinBits); k += blockDim.x) h[k] = 0;
  __syncthreads();
  const uint64_t nwords = (n + 31) >> 5;
  for (uint64_t wI = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; wI < nwords; wI += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t w0 = T[wI], w1 = T[wI + 1];
    for (uint32_t o = 0; o < 32 && wI * 32 + o < n; ++o) {
      const uint64_t key = o ? (w0 << (2 * o)) | (w1 >> (64 - 2 * o)) : w0;
      atomicAdd(&h[key >> (64 - kBinBits)], 1u);
    }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < (1u << kBinBits); k += blockDim.x) if (h[k]) atomicAdd(hist + k, (unsigned long long)h[k]);
}
// every position whose key prefix lies in [bin_lo, bin_hi) -> (key, pos), in any order (a sort follows)
__global__ __launch_bounds__(256) void k_collect(const uint64_t *T, uint64_t n, uint32_t bin_lo, uint32_t bin_hi, unsigned long long *cursor,
                                                 uint64_t *keys, uint64_t *vals) {
  const uint64_t nwords = (n + 31) >> 5;
  const uint64_t rounds = (nwords + (uint64_t)gridDim.x * blockDim.x - 1) / ((uint64_t)gridDim.x * blockDim.x);
  const uint32_t lane = threadIdx.x & 63u;
  for (uint64_t r = 0; r < rounds; ++r) {
    const uint64_t wI = r * (uint64_t)gridDim.x * blockDim.x + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t mask = 0;
    uint64_t w0 = 0, w1 = 0;
    if (wI < nwords) {
      w0 = T[wI]; w1 = T[wI + 1];
      for (uint32_t o = 0; o < 32 && wI * 32 + o < n; ++o) {
        const uint64_t key = o ? (w0 << (2 * o)) | (w1 >> (64 - 2 * o)) : w0;
        const uint32_t bin = (uint32_t)(key >> (64 - kBinBits));
        if (bin >= bin_lo && bin < bin_hi) mask |= 1u << o;
      }
    }
    // one atomic per wave: prefix sum of the lanes' counts
    const uint32_t cnt = (uint32_t)__popc(mask);
    uint32_t incl = cnt;
    for (uint32_t d = 1; d < 64; d <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
      if (lane >= d) incl += up;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
    unsigned long long base = 0;
    if (total) {
      if (lane == 0) base = atomicAdd(cursor, (unsigned long long)total);
      base = (unsigned long long)__shfl((long long)base, 0, 64);
    }
    uint64_t at = base + incl - cnt;
    while (mask) {
      const uint32_t o = (uint32_t)__ffs((int)mask) - 1u;
      mask &= mask - 1u;
      keys[at] = o ? (w0 << (2 * o)) | (w1 >> (64 - 2 * o)) : w0;
      vals[at] = wI * 32 + o;
      ++at;
    }
  }
}

// ---- commit of a sorted chunk / span ------------------------------------------------------------------------------------
// slot(a) = slots ? slots[a] : slot_base + a.  isnew: first element of a group of equal keys.
__global__ void k_new_heads(const uint64_t *K, uint64_t m, const uint64_t *slots, uint64_t slot_base, uint64_t *headslot) {
  const uint64_t a = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= m) return;
  const bool isnew = a == 0 || K[a] != K[a - 1];
  headslot[a] = isnew ? (slots ? slots[a] : slot_base + a) + 1 : 0;        // +1: slot 0 must beat "not a head"
}
struct MaxOp { __host__ __device__ __forceinline__ uint64_t operator()(uint64_t a, uint64_t b) const { return a > b ? a : b; } };
// sa holds the rows from sa_off on (the whole array with sa_off = 0, or the buffer of one chunk / span when the SA lives on the host)
__global__ void k_commit(const uint64_t *P, const uint64_t *grp, const uint64_t *headslot_raw, uint64_t m, const uint64_t *slots, uint64_t slot_base,
                         uint8_t *sa, uint64_t sa_off, uint8_t *rank, unsigned long long *head_bits) {
  const uint64_t a = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= m) return;
  const uint64_t slot = slots ? slots[a] : slot_base + a;
  const uint64_t pos = P[a];
  put40(sa, slot - sa_off, pos);
  put40(rank, pos, grp[a] - 1);
  if (headslot_raw[a]) atomicOr(head_bits + (slot >> 6), 1ull << (slot & 63));
}

// ---- phase 2 ---------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ bool head_bit(const unsigned long long *hb, uint64_t j) { return (hb[j >> 6] >> (j & 63)) & 1ull; }
struct ActivePred {          // row j belongs to a group of more than one row
  const unsigned long long *hb;
  __host__ __device__ __forceinline__ bool operator()(const uint64_t &j) const { return !(head_bit(hb, j) && head_bit(hb, j + 1)); }
};
// first head bit at or after t (bit n is always set)
__global__ void k_next_head(const unsigned long long *hb, uint64_t t, uint64_t *out) {
  uint64_t w = t >> 6;
  unsigned long long x = hb[w] & (~0ull << (t & 63));
  while (!x) x = hb[++w];
  *out = (w << 6) + (uint64_t)__ffsll((long long)x) - 1;
}
__global__ void k_active_heads(const unsigned long long *hb, const uint64_t *slots, uint64_t m, uint64_t *headslot) {
  const uint64_t a = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= m) return;
  const uint64_t j = slots[a];
  headslot[a] = head_bit(hb, j) ? j + 1 : 0;
}
__global__ void k_gather_keys(const uint8_t *sa, uint64_t sa_off, const uint8_t *rank, const uint64_t *slots, const uint64_t *grp, uint64_t m, uint64_t span_lo,
                              uint64_t n, uint64_t h, uint32_t rank_bits, uint64_t *K, uint64_t *P) {
  const uint64_t a = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= m) return;
  const uint64_t pos = get40(sa, slots[a] - sa_off);
  const uint64_t k2 = pos + h >= n ? n - 1 - pos : get40(rank, pos + h) + h;
  K[a] = ((grp[a] - 1 - span_lo) << rank_bits) | k2;
  P[a] = pos;
}

// ---- products --------------------------------------------------------------------------------------------------------
// the products below work on the rows [lo, hi) whose SA entries sit in sa from sa_off on
__global__ void k_bwt(const uint64_t *T, const uint8_t *sa, uint64_t sa_off, uint64_t lo, uint64_t hi, uint64_t n, uint8_t *bwt, unsigned long long *first_isa) {
  for (uint64_t j = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < hi; j += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t pos = get40(sa, j - sa_off);
    if (pos == 0) { *first_isa = j; bwt[j - lo] = (uint8_t)sym_at(T, n - 1); }
    else bwt[j - lo] = (uint8_t)sym_at(T, pos - 1);
  }
}
// samples k_lo .. k_hi (row k * rate each) -> ids[k - k_lo]
__global__ void k_sampled(const uint8_t *sa, uint64_t sa_off, uint64_t n, uint32_t rate, uint32_t w, const uint64_t *psum, uint64_t nseq, uint64_t k_lo, uint64_t k_hi, uint32_t *ids) {
  for (uint64_t k = k_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < k_hi; k += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t pos = get40(sa, k * rate - sa_off);
    const uint64_t adj = pos + w + 1 < n ? pos + w + 1 : pos;       // the fuzzy boundary of Builder.hpp:27-51
    uint64_t lo = 0, hi = nseq + 1;                                  // upper_bound(psum, adj) - 1
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (psum[mid] <= adj) lo = mid + 1; else hi = mid; }
    ids[k - k_lo] = (uint32_t)(lo - 1);
  }
}
__global__ void k_rows_of(const uint8_t *rank, const uint64_t *want, uint64_t cnt, uint64_t *rows) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < cnt) rows[k] = get40(rank, want[k]);
}
// ftab: rows are in suffix order, so equal w-mers are contiguous among the rows that have w characters.  One lane walks
// 256 rows and flushes a (key, first row, count) run at every key change.
__global__ void k_ftab(const uint64_t *T, const uint8_t *sa, uint64_t sa_off, uint64_t row_lo, uint64_t row_hi, uint64_t n, uint32_t w,
                       unsigned long long *first, unsigned long long *count) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t lo = row_lo + t * 256, hi = lo + 256 < row_hi ? lo + 256 : row_hi;
  if (lo >= row_hi) return;
  uint64_t cur = ~0ull, cur_first = 0, cur_cnt = 0;
  for (uint64_t j = lo; j < hi; ++j) {
    const uint64_t pos = get40(sa, j - sa_off);
    if (pos + w > n) continue;
    uint64_t k = key32(T, pos) >> (64 - 2 * w);                     // first symbol in the top pair ...
    uint64_t r = __brevll(k) >> (64 - 2 * w);                       // ... PackRead wants it in the lowest (FMBuilder.hpp:256-283)
    r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
    if (r != cur) {
      if (cur_cnt) { atomicAdd(count + cur, (unsigned long long)cur_cnt); atomicMin(first + cur, (unsigned long long)cur_first); }
      cur = r; cur_first = j; cur_cnt = 0;
    }
    ++cur_cnt;
  }
  if (cur_cnt) { atomicAdd(count + cur, (unsigned long long)cur_cnt); atomicMin(first + cur, (unsigned long long)cur_first); }
}
__global__ void k_ftab_finish(uint64_t entries, const unsigned long long *first, const unsigned long long *count, uint64_t *ftab) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= entries) return;
  ftab[2 * k] = count[k] ? first[k] : 0;
  ftab[2 * k + 1] = count[k];
}

struct DevBuf {      // frees on scope exit, also when a HIP call throws
  void *p = nullptr;
  DevBuf() = default;
  explicit DevBuf(size_t bytes) { BCHECK(hipMalloc(&p, bytes ? bytes : 16)); }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  void alloc(size_t bytes) { release(); BCHECK(hipMalloc(&p, bytes ? bytes : 16)); }
  void release() { if (p) (void)hipFree(p); p = nullptr; }
  ~DevBuf() { release(); }
  template <class T> T *as() const { return (T *)p; }
};

inline unsigned grid_of(uint64_t n, unsigned block = 256) { return (unsigned)std::max<uint64_t>(1, (n + block - 1) / block); }

}  // namespace

void build_sa_products(const uint8_t *text, uint64_t n, int device, uint32_t sample_rate, uint32_t w,
                       const std::vector<uint64_t> &psum, const std::vector<uint64_t> &want_pos, SaProducts &out,
                       const std::function<void(const std::string &)> &log) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw HipError{"index build: no HIP device (the writer has no CPU path)", -1};
  if (device < 0 || device >= count) throw HipError{"index build: device ordinal out of range", -1};
  if (n < 64) throw HipError{"index build: text shorter than 64 symbols", -2};
  {
    // (test hook, behind CFR_DEBUG_ENV: CFR_BUILD_LIMIT_LOG2 lowers the size this writer refuses at, so that a test walks the refusal on a small text)
    uint64_t limit = 1ull << 36;
    if (getenv("CFR_DEBUG_ENV") && atoi(getenv("CFR_DEBUG_ENV")) && getenv("CFR_BUILD_LIMIT_LOG2")) limit = 1ull << std::min(36, std::max(8, atoi(getenv("CFR_BUILD_LIMIT_LOG2"))));
    if (n + (limit >> 9) >= limit) throw HipError{"index build: texts of 2^36 symbols and more are beyond this single-GPU writer", -2};
  }
  if (w < 1 || w > 16) throw HipError{"index build: ftab width must be in 1..16", -2};
  BCHECK(hipSetDevice(device));
  hipStream_t st = nullptr;      // default stream: everything here is sequential
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  const auto t_begin = now();
  char msg[256];

  // ---- text
  const uint64_t nwords = (n + 31) / 32;
  DevBuf d_text((nwords + 4) * 8);
  uint64_t *T = d_text.as<uint64_t>();
  BCHECK(hipMemsetAsync(T, 0, (nwords + 4) * 8, st));
  {
    const uint64_t piece = 1ull << 30;
    DevBuf stage(piece), d_bad(4);
    BCHECK(hipMemsetAsync(d_bad.p, 0, 4, st));
    for (uint64_t lo = 0; lo < n; lo += piece) {
      const uint64_t cnt = std::min(piece, n - lo);
      BCHECK(hipMemcpy(stage.p, text + lo, cnt, hipMemcpyHostToDevice));
      k_pack_text<<<grid_of((cnt + 31) / 32), 256, 0, st>>>(stage.as<uint8_t>(), cnt, T + lo / 32, d_bad.as<unsigned int>());
      BCHECK(hipGetLastError());
      BCHECK(hipStreamSynchronize(st));
    }
    unsigned int bad = 0;
    BCHECK(hipMemcpy(&bad, d_bad.p, 4, hipMemcpyDeviceToHost));
    if (bad) throw HipError{"index build: the text must be upper-case ACGT only", -2};
  }

  uint32_t rank_bits = 1;
  while ((n + (1ull << 27)) >> rank_bits) ++rank_bits;
  // test hooks (behind the CFR_DEBUG_ENV gate, like the classifier's switches): a small chunk makes small texts take many
  // chunks and spans; CFR_BUILD_HOST_SA forces the host-resident suffix array
  const bool dbg = getenv("CFR_DEBUG_ENV") && atoi(getenv("CFR_DEBUG_ENV"));
  uint64_t kChunk = kChunkDefault;
  if (dbg && getenv("CFR_BUILD_CHUNK_LOG2")) kChunk = 1ull << std::min(28, std::max(12, atoi(getenv("CFR_BUILD_CHUNK_LOG2"))));
  const uint64_t kMargin = std::min<uint64_t>(1ull << 20, kChunk / 2);      // a span ends on the first group boundary past chunk - margin
  // SA on the device when SA + RANK + text + head bits + chunk buffers fit, else in host memory (one chunk / span on the device)
  size_t free_b = 0, total_b = 0;
  BCHECK(hipMemGetInfo(&free_b, &total_b));
  const bool host_sa = (double)n * 10.2 + 16e9 > 0.95 * (double)free_b || (dbg && getenv("CFR_BUILD_HOST_SA") && atoi(getenv("CFR_BUILD_HOST_SA")));
  if (host_sa && (double)n * 5.2 + 22e9 > 0.95 * (double)free_b) throw HipError{"index build: the text does not fit this device even with the suffix array on the host", -3};
  std::unique_ptr<uint8_t[]> hsa_mem;       // (not a vector: 5 n bytes must not be zero-filled first)
  uint8_t *hsa = nullptr;
  if (host_sa) {
    hsa_mem.reset(new uint8_t[n * 5 + 16]);
    hsa = hsa_mem.get();
    if (log) log("suffix array kept in host memory (" + std::to_string((n * 5) >> 30) + " GiB); RANK, text and head bits in HBM");
  }
  DevBuf d_sa(host_sa ? kChunk * 5 + 16 : n * 5 + 16), d_rank(n * 5 + 16), d_head(((n + 1 + 63) / 64 + 2) * 8);
  uint8_t *SA = d_sa.as<uint8_t>(), *RANK = d_rank.as<uint8_t>();
  unsigned long long *HB = d_head.as<unsigned long long>();
  BCHECK(hipMemsetAsync(HB, 0, ((n + 1 + 63) / 64 + 2) * 8, st));

  // chunk buffers (shared by both phases)
  DevBuf bK0(kChunk * 8), bK1(kChunk * 8), bP0(kChunk * 8), bP1(kChunk * 8), bSlots(kChunk * 8), bHS(kChunk * 8), bGrp(kChunk * 8), bScalar(64);
  uint64_t *K0 = bK0.as<uint64_t>(), *K1 = bK1.as<uint64_t>(), *P0 = bP0.as<uint64_t>(), *P1 = bP1.as<uint64_t>();
  uint64_t *SL = bSlots.as<uint64_t>(), *HS = bHS.as<uint64_t>(), *GRP = bGrp.as<uint64_t>();
  unsigned long long *d_scalar = bScalar.as<unsigned long long>();
  size_t tmp_sort = 0, tmp_scan = 0, tmp_sel = 0;
  BCHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, K0, K1, P0, P1, (uint64_t)kChunk, 0, 64, st));
  BCHECK(hipcub::DeviceScan::InclusiveScan(nullptr, tmp_scan, HS, GRP, MaxOp(), (uint64_t)kChunk, st));
  {
    hipcub::CountingInputIterator<uint64_t> it(0);
    BCHECK(hipcub::DeviceSelect::If(nullptr, tmp_sel, it, SL, (uint64_t *)d_scalar, (int)kChunk, ActivePred{HB}, st));
  }
  const size_t tmp_bytes = std::max(tmp_sort, std::max(tmp_scan, tmp_sel));
  DevBuf bTmp(tmp_bytes);

  // sa_off: first row the SA buffer holds (0 with the whole array on the device)
  auto commit = [&](uint64_t *Ksorted, uint64_t *Psorted, uint64_t m, const uint64_t *slots, uint64_t slot_base, uint64_t sa_off) {
    k_new_heads<<<grid_of(m), 256, 0, st>>>(Ksorted, m, slots, slot_base, HS);
    size_t tb = tmp_bytes;
    BCHECK(hipcub::DeviceScan::InclusiveScan(bTmp.p, tb, HS, GRP, MaxOp(), m, st));
    k_commit<<<grid_of(m), 256, 0, st>>>(Psorted, GRP, HS, m, slots, slot_base, SA, sa_off, RANK, HB);
    BCHECK(hipGetLastError());
  };

  // ---- phase 1: order by the first 32 symbols
  {
    DevBuf d_hist((1u << kBinBits) * 8);
    BCHECK(hipMemsetAsync(d_hist.p, 0, (1u << kBinBits) * 8, st));
    k_hist<<<2048, 256, 0, st>>>(T, n, d_hist.as<unsigned long long>());
    BCHECK(hipGetLastError());
    std::vector<unsigned long long> hist(1u << kBinBits);
    BCHECK(hipMemcpy(hist.data(), d_hist.p, hist.size() * 8, hipMemcpyDeviceToHost));
    uint64_t base = 0;
    uint32_t bin = 0;
    int chunks = 0;
    while (bin < (1u << kBinBits)) {
      uint32_t hi = bin;
      uint64_t cnt = 0;
      while (hi < (1u << kBinBits) && cnt + hist[hi] <= kChunk) cnt += hist[hi++];
      if (hi == bin) throw HipError{"index build: more suffixes than one chunk holds share one 7-symbol prefix (text too skewed for this writer)", -2};
      if (cnt) {
        BCHECK(hipMemsetAsync(d_scalar, 0, 8, st));
        k_collect<<<4096, 256, 0, st>>>(T, n, bin, hi, d_scalar, K0, P0);
        BCHECK(hipGetLastError());
        size_t tb = tmp_bytes;
        BCHECK(hipcub::DeviceRadixSort::SortPairs(bTmp.p, tb, K0, K1, P0, P1, cnt, 0, 64, st));
        commit(K1, P1, cnt, nullptr, base, host_sa ? base : 0);
        if (host_sa) BCHECK(hipMemcpy(hsa + base * 5, SA, cnt * 5, hipMemcpyDeviceToHost));
        ++chunks;
      }
      base += cnt;
      bin = hi;
    }
    if (base != n) throw HipError{"index build: internal error (phase 1 lost suffixes)", -5};
    const unsigned long long one = 1ull << (n & 63);                  // bit n: the end of the last group
    BCHECK(hipStreamSynchronize(st));
    unsigned long long last = 0;
    BCHECK(hipMemcpy(&last, HB + (n >> 6), 8, hipMemcpyDeviceToHost));
    last |= one;
    BCHECK(hipMemcpy(HB + (n >> 6), &last, 8, hipMemcpyHostToDevice));
    snprintf(msg, sizeof(msg), "suffix array: %d chunks sorted by their first 32 symbols, %.1f s", chunks, secs(t_begin, now()));
    if (log) log(msg);
  }

  // ---- phase 2: doubling over the groups that are not single rows yet
  int rounds = 0;
  for (uint64_t h = 32;; h <<= 1) {
    if (h >= (1ull << 27)) throw HipError{"index build: repeats longer than 2^27 symbols (not supported by this writer)", -2};
    uint64_t active_total = 0;
    uint64_t lo = 0;
    while (lo < n) {
      uint64_t hi = n;
      if (n - lo > kChunk - kMargin) {      // a span ends on the first group boundary at or after lo + chunk - margin (groups are far smaller than the margin)
        k_next_head<<<1, 1, 0, st>>>(HB, lo + kChunk - kMargin, (uint64_t *)d_scalar + 1);
        BCHECK(hipMemcpy(&hi, (uint64_t *)d_scalar + 1, 8, hipMemcpyDeviceToHost));
        if (hi - lo > kChunk) throw HipError{"index build: a group of more than 2^20 equal prefixes (text too repetitive for this writer)", -2};
      }
      hipcub::CountingInputIterator<uint64_t> it(lo);
      size_t tb = tmp_bytes;
      BCHECK(hipcub::DeviceSelect::If(bTmp.p, tb, it, SL, (uint64_t *)d_scalar, (int)(hi - lo), ActivePred{HB}, st));
      uint64_t m = 0;
      BCHECK(hipMemcpy(&m, d_scalar, 8, hipMemcpyDeviceToHost));
      if (m) {
        if (host_sa) BCHECK(hipMemcpy(SA, hsa + lo * 5, (hi - lo) * 5, hipMemcpyHostToDevice));
        k_active_heads<<<grid_of(m), 256, 0, st>>>(HB, SL, m, HS);
        tb = tmp_bytes;
        BCHECK(hipcub::DeviceScan::InclusiveScan(bTmp.p, tb, HS, GRP, MaxOp(), m, st));
        k_gather_keys<<<grid_of(m), 256, 0, st>>>(SA, host_sa ? lo : 0, RANK, SL, GRP, m, lo, n, h, rank_bits, K0, P0);
        BCHECK(hipGetLastError());
        tb = tmp_bytes;
        BCHECK(hipcub::DeviceRadixSort::SortPairs(bTmp.p, tb, K0, K1, P0, P1, m, 0, 64, st));
        commit(K1, P1, m, SL, 0, host_sa ? lo : 0);
        if (host_sa) BCHECK(hipMemcpy(hsa + lo * 5, SA, (hi - lo) * 5, hipMemcpyDeviceToHost));
      }
      active_total += m;
      lo = hi;
    }
    BCHECK(hipStreamSynchronize(st));
    if (active_total == 0) break;
    ++rounds;
    snprintf(msg, sizeof(msg), "suffix array: round %d (depth %llu -> %llu): %llu rows in unresolved groups, %.1f s", rounds,
             (unsigned long long)h, (unsigned long long)(2 * h), (unsigned long long)active_total, secs(t_begin, now()));
    if (log) log(msg);
  }
  out.rounds = rounds;
  out.seconds_sa = secs(t_begin, now());
  const auto t_prod = now();

  // ---- products
  bK0.release(); bK1.release(); bP0.release(); bP1.release(); bSlots.release(); bHS.release(); bGrp.release(); bTmp.release();
  out.n = n;
  out.rows_of.assign(want_pos.size(), 0);
  if (!want_pos.empty()) {
    DevBuf d_want(want_pos.size() * 8), d_rows(want_pos.size() * 8);
    BCHECK(hipMemcpy(d_want.p, want_pos.data(), want_pos.size() * 8, hipMemcpyHostToDevice));
    k_rows_of<<<grid_of(want_pos.size()), 256, 0, st>>>(RANK, d_want.as<uint64_t>(), want_pos.size(), d_rows.as<uint64_t>());
    BCHECK(hipGetLastError());
    BCHECK(hipMemcpy(out.rows_of.data(), d_rows.p, want_pos.size() * 8, hipMemcpyDeviceToHost));
  }
  d_rank.release();
  d_head.release();
  {
    // rows in pieces: the whole array when the SA is on the device, one chunk at a time from the host copy otherwise
    const uint64_t piece = host_sa ? kChunk : n;
    const uint64_t nsamp = (n + sample_rate - 1) / sample_rate, entries = 1ull << (2 * w);
    DevBuf d_psum(psum.size() * 8), d_ids(std::min<uint64_t>(nsamp, piece / sample_rate + 2) * 4), d_first(entries * 8), d_count(entries * 8), d_ftab(entries * 16);
    DevBuf d_bwt(piece), d_fi(8);
    BCHECK(hipMemcpy(d_psum.p, psum.data(), psum.size() * 8, hipMemcpyHostToDevice));
    BCHECK(hipMemsetAsync(d_first.p, 0xff, entries * 8, st));
    BCHECK(hipMemsetAsync(d_count.p, 0, entries * 8, st));
    BCHECK(hipMemsetAsync(d_fi.p, 0, 8, st));
    out.bwt.resize(n);
    std::vector<uint32_t> &ids = out.sampled_ids;
    ids.resize(nsamp);
    for (uint64_t lo = 0; lo < n; lo += piece) {
      const uint64_t hi = std::min(n, lo + piece), sa_off = host_sa ? lo : 0;
      if (host_sa) BCHECK(hipMemcpy(SA, hsa + lo * 5, (hi - lo) * 5, hipMemcpyHostToDevice));
      const uint64_t k_lo = (lo + sample_rate - 1) / sample_rate, k_hi = (hi + sample_rate - 1) / sample_rate;      // samples with row in [lo, hi)
      if (k_hi > k_lo) {
        k_sampled<<<(unsigned)std::min<uint64_t>(grid_of(k_hi - k_lo), 1u << 20), 256, 0, st>>>(SA, sa_off, n, sample_rate, w, d_psum.as<uint64_t>(), psum.size() - 1, k_lo, k_hi, d_ids.as<uint32_t>());
        BCHECK(hipGetLastError());
        BCHECK(hipMemcpy(ids.data() + k_lo, d_ids.p, (k_hi - k_lo) * 4, hipMemcpyDeviceToHost));
      }
      k_ftab<<<grid_of((hi - lo + 255) / 256), 256, 0, st>>>(T, SA, sa_off, lo, hi, n, w, d_first.as<unsigned long long>(), d_count.as<unsigned long long>());
      k_bwt<<<(unsigned)std::min<uint64_t>(grid_of(hi - lo), 1u << 20), 256, 0, st>>>(T, SA, sa_off, lo, hi, n, d_bwt.as<uint8_t>(), d_fi.as<unsigned long long>());
      BCHECK(hipGetLastError());
      BCHECK(hipMemcpy(out.bwt.data() + lo, d_bwt.p, hi - lo, hipMemcpyDeviceToHost));
      if (host_sa) {      // this piece of the host copy is done with: give its pages back while the BWT grows
        const uintptr_t a0 = ((uintptr_t)(hsa + lo * 5) + 4095) & ~(uintptr_t)4095, a1 = (uintptr_t)(hsa + hi * 5) & ~(uintptr_t)4095;
        if (a1 > a0) (void)madvise((void *)a0, a1 - a0, MADV_DONTNEED);
      }
    }
    k_ftab_finish<<<grid_of(entries), 256, 0, st>>>(entries, d_first.as<unsigned long long>(), d_count.as<unsigned long long>(), d_ftab.as<uint64_t>());
    BCHECK(hipGetLastError());
    out.ftab.resize(entries * 2);
    BCHECK(hipMemcpy(out.ftab.data(), d_ftab.p, entries * 16, hipMemcpyDeviceToHost));
    unsigned long long fi = 0;
    BCHECK(hipMemcpy(&fi, d_fi.p, 8, hipMemcpyDeviceToHost));
    out.first_isa = fi;
  }
  out.seconds_products = secs(t_prod, now());
}

// ============================================================================================ byte texts (protein indexes)
// Suffix array of a text of byte codes, n < 2^32 (cfr_build.hpp: build_sa_bytes).  A protein text is 5 bits per symbol and three
// orders of magnitude shorter than the nucleotide texts above, so the plain form of prefix doubling serves: every position carries
// the rank of its first h symbols, one round sorts all positions by (rank[i], rank[i + h]) with one hipCUB radix sort and renumbers
// the groups with a flag pass and a scan; h = 12 (the first 12 symbols fit one 60-bit key), 24, 48, ... until every group is a
// single row.  A position past the end ranks 0, below every real rank: a proper prefix sorts first, as everywhere in this writer.
// Memory: 37 bytes per symbol.
namespace {

__global__ void k_bytes_key0(const uint8_t *text, uint64_t n, uint64_t *keys, uint32_t *pos) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t key = 0;
#pragma unroll
    for (uint32_t k = 0; k < 12; ++k) key = (key << 5) | (i + k < n ? (uint64_t)text[i + k] + 1ull : 0ull);
    keys[i] = key;
    pos[i] = (uint32_t)i;
  }
}
__global__ void k_bytes_key_h(const uint32_t *rank, uint64_t n, uint64_t h, uint64_t *keys, uint32_t *pos) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    keys[i] = ((uint64_t)rank[i] << 32) | (i + h < n ? (uint64_t)rank[i + h] : 0ull);
    pos[i] = (uint32_t)i;
  }
}
__global__ void k_bytes_flags(const uint64_t *keys, uint64_t n, uint32_t *flags) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
__global__ void k_bytes_scatter(const uint32_t *pos, const uint32_t *grp, uint64_t n, uint32_t *rank) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) rank[pos[i]] = grp[i];
}

}  // namespace

void build_sa_bytes(const uint8_t *codes, uint64_t n, int device, std::vector<uint32_t> &sa, double *seconds, int *rounds) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw HipError{"index build: no HIP device (the writer has no CPU path)", -1};
  if (device < 0 || device >= count) throw HipError{"index build: device ordinal out of range", -1};
  if (n == 0 || n >= 0xfffffff0ull) throw HipError{"index build: a byte text must hold between 1 and 2^32 - 16 symbols", -2};
  BCHECK(hipSetDevice(device));
  hipStream_t st = nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  DevBuf d_text(n + 16), bK0(n * 8), bK1(n * 8), bP0(n * 4), bP1(n * 4), bRank(n * 4), bFlag(n * 4), bGrp(n * 4);
  BCHECK(hipMemcpy(d_text.p, codes, n, hipMemcpyHostToDevice));
  uint64_t *K0 = bK0.as<uint64_t>(), *K1 = bK1.as<uint64_t>();
  uint32_t *P0 = bP0.as<uint32_t>(), *P1 = bP1.as<uint32_t>(), *RANK = bRank.as<uint32_t>(), *FLAG = bFlag.as<uint32_t>(), *GRP = bGrp.as<uint32_t>();
  size_t tmp_sort = 0, tmp_scan = 0;
  BCHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, K0, K1, P0, P1, n, 0, 64, st));
  BCHECK(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_scan, FLAG, GRP, n, st));
  size_t tmp_bytes = std::max(tmp_sort, tmp_scan);
  DevBuf bTmp(tmp_bytes);
  const unsigned grid = (unsigned)std::min<uint64_t>(grid_of(n), 1u << 16);
  uint32_t groups = 0;
  int round = 0;
  auto renumber = [&](int end_bit) {        // K0 / P0 hold the keys by position: sort, number the groups, hand every position its group
    size_t tb = tmp_bytes;
    BCHECK(hipcub::DeviceRadixSort::SortPairs(bTmp.p, tb, K0, K1, P0, P1, n, 0, end_bit, st));
    k_bytes_flags<<<grid, 256, 0, st>>>(K1, n, FLAG);
    BCHECK(hipGetLastError());
    tb = tmp_bytes;
    BCHECK(hipcub::DeviceScan::InclusiveSum(bTmp.p, tb, FLAG, GRP, n, st));
    k_bytes_scatter<<<grid, 256, 0, st>>>(P1, GRP, n, RANK);
    BCHECK(hipGetLastError());
    BCHECK(hipMemcpy(&groups, GRP + (n - 1), 4, hipMemcpyDeviceToHost));
    ++round;
  };
  k_bytes_key0<<<grid, 256, 0, st>>>(d_text.as<uint8_t>(), n, K0, P0);
  BCHECK(hipGetLastError());
  renumber(60);
  uint64_t h = 12;
  while ((uint64_t)groups < n) {
    if (h >= 2 * n + 24) throw HipError{"index build: the doubling rounds of a byte text did not separate its suffixes", -3};
    k_bytes_key_h<<<grid, 256, 0, st>>>(RANK, n, h, K0, P0);
    BCHECK(hipGetLastError());
    renumber(64);
    h *= 2;
  }
  sa.resize(n);
  BCHECK(hipMemcpy(sa.data(), P1, n * 4, hipMemcpyDeviceToHost));
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  if (rounds) *rounds = round;
}

}  // namespace cfr
