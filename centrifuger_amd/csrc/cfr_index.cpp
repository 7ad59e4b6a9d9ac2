// cfr_index.cpp — .cfr parser (see cfr_index.hpp).  Host C++ only, no HIP.
#include "cfr_index.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <thread>

namespace cfr {

namespace {

// Read-only cursor over an mmap'ed file.
class Cursor {
 public:
  explicit Cursor(const std::string &path) : path_(path) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw IoError{"cannot open " + path};
    struct stat st;
    if (fstat(fd_, &st) != 0) { ::close(fd_); throw IoError{"cannot stat " + path}; }
    size_ = (size_t)st.st_size;
    if (size_ > 0) {
      base_ = (const uint8_t *)mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
      if (base_ == MAP_FAILED) { ::close(fd_); throw IoError{"cannot mmap " + path}; }
    }
  }
  ~Cursor() {
    if (base_ && size_ && !kept_) munmap((void *)base_, size_);
    if (fd_ >= 0) ::close(fd_);
  }
  // the mapping outlives the cursor: whoever holds the returned pointer keeps the file's pages reachable (RawWords::map)
  std::shared_ptr<void> keep() {
    if (!kept_ && base_ && size_) {
      const size_t sz = size_;
      kept_ = std::shared_ptr<void>((void *)base_, [sz](void *q) { munmap(q, sz); });
    }
    return kept_;
  }
  // n words at the cursor: left in the mapping when the caller allows it, copied otherwise.  (Until round 5 only strings that stood 8-byte
  // aligned in the file were left there - and none does: the header's one-byte lastChr puts every bit string of a nucleotide index at an
  // odd offset, so every "mapped" open copied its 12 GB at 40 Gbp after all.  RawWords reads its words bytewise now.)
  void words(RawWords &w, size_t n, bool may_map, size_t extra_zero_words = 0) {
    need(n * 8);
    if (may_map && n && size_ - pos_ >= (n + extra_zero_words) * 8) {
      w.map(base_ + pos_, n + extra_zero_words);     // (the words behind it: whatever the file holds there - see the caller)
      pos_ += n * 8;
      return;
    }
    uint64_t *dst = w.alloc(n + extra_zero_words);
    for (size_t k = 0; k < extra_zero_words; ++k) dst[n + k] = 0;
    copy(dst, n * 8);
  }
  template <class T> T get() {
    T v;
    need(sizeof(T));
    memcpy(&v, base_ + pos_, sizeof(T));
    pos_ += sizeof(T);
    return v;
  }
  void copy(void *dst, size_t bytes) {
    need(bytes);
    if (bytes >= (64u << 20)) {
      // the large arrays of a multi-Gbp index (GBs of bitvector words): slices on several threads, which also spreads the
      // page faults of the mapping (one thread reads a 15 GB file at ~4 GB/s)
      const unsigned nt = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
      std::vector<std::thread> th;
      for (unsigned t = 0; t < nt; ++t) th.emplace_back([=]() {
        const size_t lo = bytes / nt * t, hi = t + 1 == nt ? bytes : bytes / nt * (t + 1);
        memcpy((char *)dst + lo, base_ + pos_ + lo, hi - lo);
      });
      for (auto &x : th) x.join();
    } else memcpy(dst, base_ + pos_, bytes);
    pos_ += bytes;
  }
  void skip(size_t bytes) { need(bytes); pos_ += bytes; }
  bool eof() const { return pos_ >= size_; }
  size_t remaining() const { return size_ - pos_; }
  size_t pos() const { return pos_; }
  size_t size() const { return size_; }
  const uint8_t *base() const { return base_; }
  const std::string &path() const { return path_; }
  int fd() const { return fd_; }

 private:
  void need(size_t bytes) const {
    if (bytes > size_ - pos_) throw FormatError{"truncated file " + path_};
  }
  std::string path_;
  int fd_ = -1;
  const uint8_t *base_ = nullptr;
  size_t size_ = 0, pos_ = 0;
  std::shared_ptr<void> kept_;
};

// CFR_INDEX_COPY=1 (a test / A-B switch behind CFR_DEBUG_ENV=1): every bit string copied into the process, as until round 4
bool want_copies() {
  const char *g = ::getenv("CFR_DEBUG_ENV"), *e = ::getenv("CFR_INDEX_COPY");
  return g && atoi(g) != 0 && e && atoi(e) != 0;
}

inline uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// ALPHABET := u64 space | i32 method | u64 n | [char list[n] | i32 code[256] | i16 codeLen[256]]
// returns the number of symbols; checks it is the plain 2-bit ACGT coder when non-empty.
uint64_t parse_alphabet(Cursor &c, bool must_be_acgt) {
  c.get<uint64_t>();
  int32_t method = c.get<int32_t>();
  uint64_t n = c.get<uint64_t>();
  if (n == 0) return 0;
  if (n > 255) throw FormatError{"alphabet too large"};
  char list[256];
  int32_t code[256];
  int16_t code_len[256];
  c.copy(list, n);
  c.copy(code, sizeof(code));
  c.copy(code_len, sizeof(code_len));
  if (must_be_acgt) {
    bool ok = method == 1 && n == 4 && memcmp(list, "ACGT", 4) == 0;
    for (int i = 0; ok && i < 4; ++i) ok = code[(unsigned char)list[i]] == i && code_len[(unsigned char)list[i]] == 2;
    if (!ok) throw FormatError{"only the plain-coded nucleotide alphabet ACGT is supported (protein indexes are out of scope)"};
  }
  return n;
}

// BITVEC := u64 space | u64 n | i32 rb | i32 sb | i32 selectSpeed | i32 selectType | [B | RANK9 | SELECT]
void parse_bitvector(Cursor &c, RawBitvector &bv, bool may_map = false) {
  c.get<uint64_t>();
  bv.n = c.get<uint64_t>();
  c.skip(4 * sizeof(int32_t));
  if (bv.n == 0) return;
  uint64_t words = ceil_div(bv.n, 64);
  bv.file_off = c.pos();
  c.words(bv.bits, words, may_map);
  c.get<uint64_t>();                        // rank9 _space
  uint64_t word_cnt = c.get<uint64_t>();
  if (word_cnt != words) throw FormatError{"rank9 word count mismatch"};
  c.skip(2 * ceil_div(word_cnt, 8) * 8);    // the rank9 blocks themselves: not used (rank lines are counted at image build)
  c.get<uint64_t>();                        // select _space
  uint64_t sel_n = c.get<uint64_t>();
  int32_t sel_speed = c.get<int32_t>();
  if (sel_speed != 0 && sel_n != 0) throw FormatError{"bitvector carries select structures (unexpected on this path)"};
}

// WAVELET := SEQHDR | i32 nodeCnt | i32 selectSpeed | NODE*   ; SEQHDR := u64 space | u64 n | ALPHABET
void parse_wavelet(Cursor &c, RawWavelet &w, bool may_map = false) {
  c.get<uint64_t>();
  w.n = c.get<uint64_t>();
  uint64_t asz = parse_alphabet(c, true);
  int32_t node_cnt = c.get<int32_t>();
  c.get<int32_t>();
  if (asz == 0) { w.node_cnt = 0; return; }      // never-initialised tree (--rbbwt-b 1)
  if (node_cnt != 3) throw FormatError{"wavelet tree with " + std::to_string(node_cnt) + " nodes (expected 3)"};
  w.node_cnt = 3;
  for (int i = 0; i < 3; ++i) {
    c.get<uint64_t>();      // prefix
    c.get<int32_t>();       // prefixLen
    w.children[i][0] = c.get<int32_t>();
    w.children[i][1] = c.get<int32_t>();
    parse_bitvector(c, w.node[i], may_map);
  }
  if (w.children[0][0] < 1 || w.children[0][0] > 2 || w.children[0][1] < 1 || w.children[0][1] > 2 ||
      w.children[0][0] == w.children[0][1])
    throw FormatError{"unexpected wavelet tree shape"};
  if (w.node[0].n != w.n) throw FormatError{"wavelet root length mismatch"};
}

inline unsigned bit_at(const RawBitvector &bv, uint64_t i) { return (unsigned)((bv.bits[i >> 6] >> (i & 63)) & 1); }

// The three run-block components must describe exactly n symbols (Sequence_RunBlock.hpp:15-20): one run-block symbol
// per set bit of useRunBlock, and the plain sequence holds what the other blocks cover.  (The BWT itself is never
// decoded on the host: the device expands its flat image from these components, cfr_device.hip.)
void check_run_block_lengths(const HostIndex &h) {
  uint64_t runs = 0;
  for (uint64_t w = 0; w * 64 < h.block_cnt; ++w) {
    uint64_t x = h.use_run_block.bits[w];
    if ((w + 1) * 64 > h.block_cnt) x &= (1ull << (h.block_cnt - w * 64)) - 1;
    runs += (uint64_t)__builtin_popcountll(x);
  }
  if (runs && h.run_block_seq.node_cnt == 0) throw FormatError{"run block without a run-block sequence"};
  const uint64_t last_len = h.n - (h.block_cnt - 1) * h.b;
  const bool last_is_run = h.block_cnt && bit_at(h.use_run_block, h.block_cnt - 1);
  const uint64_t covered_by_runs = runs * h.b - (last_is_run ? h.b - last_len : 0);
  if (covered_by_runs > h.n) throw FormatError{"decoded BWT length mismatch"};
  const uint64_t plain = h.n - covered_by_runs;
  if ((h.wavelet_seq.node_cnt ? h.wavelet_seq.n : 0) != plain || (h.run_block_seq.node_cnt && h.run_block_seq.n != runs))
    throw FormatError{"run-block component lengths do not add up"};
}

void parse_fm(const std::string &path, HostIndex &h) {
  Cursor c(path);
  h.n = c.get<uint64_t>();
  h.alphabet_bits = c.get<uint64_t>();
  h.first_isa = c.get<uint64_t>();
  h.last_chr = c.get<char>();
  // Sequence_RunBlock
  c.get<uint64_t>();
  uint64_t seq_n = c.get<uint64_t>();
  parse_alphabet(c, true);
  h.b = c.get<uint64_t>();
  h.block_cnt = c.get<uint64_t>();
  if (seq_n != h.n || h.b == 0 || h.block_cnt != ceil_div(h.n, h.b)) throw FormatError{"inconsistent run-block header"};
  const bool may_map = !want_copies();
  parse_bitvector(c, h.use_run_block, may_map);
  if (h.use_run_block.n != h.block_cnt) throw FormatError{"useRunBlock length mismatch"};
  parse_wavelet(c, h.wavelet_seq, may_map);
  parse_wavelet(c, h.run_block_seq, may_map);
  // alphabets + C[]
  parse_alphabet(c, true);
  uint64_t plain_n = parse_alphabet(c, true);
  if (plain_n != 4 || h.alphabet_bits != 2) throw FormatError{"unexpected alphabet"};
  c.copy(h.C, sizeof(h.C));
  const char *acgt = "ACGT";
  const char *p = strchr(acgt, h.last_chr);
  if (!p || !h.last_chr) throw FormatError{"lastChr not in ACGT"};
  h.last_code = (uint8_t)(p - acgt);
  // _FMIndexAuxData
  uint64_t aux_n = c.get<uint64_t>();
  c.get<int32_t>();                        // sampleStrategy
  h.sample_rate = c.get<int32_t>();
  h.sample_size = c.get<uint64_t>();
  h.precompute_width = c.get<uint64_t>();
  h.precompute_size = c.get<uint64_t>();
  h.adjusted_sa0 = c.get<uint64_t>();
  if (aux_n != h.n || h.sample_rate <= 0) throw FormatError{"inconsistent aux header"};
  if (h.precompute_width > 15 || h.precompute_size != (h.precompute_width ? (1ull << (2 * h.precompute_width)) : 0))
    throw FormatError{"unsupported ftab width"};
  c.get<uint64_t>();                       // FSEA _size (words allocated)
  h.sampled_bits = c.get<int32_t>();
  h.sampled_n = c.get<uint64_t>();
  if (h.sampled_bits <= 0 || h.sampled_bits > 64) throw FormatError{"bad sampledSA element width"};
  uint64_t sw = ceil_div(h.sampled_n * (uint64_t)h.sampled_bits, 64);
  // +2: the device reads two words unconditionally and masks what it takes - zeros in a copy, the file's next 16 bytes (the ftab's
  // first entry) in the mapping; no element's bits reach them
  c.words(h.sampled_words, sw, may_map, 2);
  h.ftab.resize(2 * h.precompute_size);
  c.copy(h.ftab.data(), h.ftab.size() * 8);
  uint64_t max_lcp = c.get<uint64_t>();
  if (max_lcp > 0) c.skip(2 * ceil_div(h.n, 64) * 8);
  uint64_t sel_cnt = c.get<uint64_t>();
  h.selected_filter_rate = c.get<int32_t>();
  h.selected_rows.resize(sel_cnt);
  h.selected_vals.resize(sel_cnt);
  for (uint64_t i = 0; i < sel_cnt; ++i) {
    h.selected_rows[i] = c.get<uint64_t>();
    h.selected_vals[i] = c.get<uint64_t>();
    if (i && h.selected_rows[i] <= h.selected_rows[i - 1]) throw FormatError{"selectedSA rows not ascending"};
  }
  h.has_end_marker = false;
  if (!c.eof()) h.has_end_marker = c.get<uint8_t>() != 0;   // absent in old indexes (FMIndex.hpp:178-181)
  if (h.has_end_marker) throw FormatError{"end-marker (protein) indexes are out of scope"};
  check_run_block_lengths(h);
  const bool any_mapped = h.use_run_block.bits.mapped() || h.sampled_words.mapped() || h.wavelet_seq.node[0].bits.mapped() || h.wavelet_seq.node[1].bits.mapped() ||
                          h.wavelet_seq.node[2].bits.mapped() || h.run_block_seq.node[0].bits.mapped() || h.run_block_seq.node[1].bits.mapped() || h.run_block_seq.node[2].bits.mapped();
  if (any_mapped) h.file_mapping = c.keep();
}

// ---- protein index: FMIndex<Sequence_RunBlockOneTree>::Load (FMIndex.hpp:588-606, Sequence_RunBlockOneTree.hpp:499-513)
struct GenAlphabet { int32_t method = 0; uint64_t n = 0; char list[256]; int32_t code[256]; int16_t code_len[256]; };
void parse_gen_alphabet(Cursor &c, GenAlphabet &a) {
  c.get<uint64_t>();
  a.method = c.get<int32_t>();
  a.n = c.get<uint64_t>();
  if (a.n == 0) return;
  if (a.n > 64) throw FormatError{"alphabet too large"};
  c.copy(a.list, a.n);
  c.copy(a.code, sizeof(a.code));
  c.copy(a.code_len, sizeof(a.code_len));
  if (a.method != 1) throw FormatError{"only plain-coded alphabets are supported"};
}
struct GenWavelet { uint64_t n = 0; GenAlphabet alphabet; std::vector<int32_t> child0, child1; std::vector<RawBitvector> node; };
void parse_gen_wavelet(Cursor &c, GenWavelet &w) {
  c.get<uint64_t>();
  w.n = c.get<uint64_t>();
  parse_gen_alphabet(c, w.alphabet);
  const int32_t node_cnt = c.get<int32_t>();
  c.get<int32_t>();
  if (w.alphabet.n == 0) return;
  if (node_cnt <= 0 || node_cnt > 255) throw FormatError{"unexpected wavelet node count"};
  w.node.resize((size_t)node_cnt);
  for (int i = 0; i < node_cnt; ++i) {
    c.get<uint64_t>();
    c.get<int32_t>();
    w.child0.push_back(c.get<int32_t>());
    w.child1.push_back(c.get<int32_t>());
    parse_bitvector(c, w.node[(size_t)i]);
  }
}

void parse_fm_protein(const std::string &path, HostIndex &h) {
  Cursor c(path);
  ProteinPart &P = h.prot;
  P.enabled = true;
  h.n = c.get<uint64_t>();
  h.alphabet_bits = c.get<uint64_t>();
  h.first_isa = c.get<uint64_t>();
  h.last_chr = c.get<char>();
  // Sequence_RunBlockOneTree
  c.get<uint64_t>();
  const uint64_t seq_n = c.get<uint64_t>();
  GenAlphabet seq_alpha;
  parse_gen_alphabet(c, seq_alpha);
  h.b = c.get<uint64_t>();
  h.block_cnt = c.get<uint64_t>();
  if (seq_n != h.n || h.b == 0 || h.block_cnt != ceil_div(h.n, h.b)) throw FormatError{"inconsistent run-block header"};
  parse_bitvector(c, h.use_run_block);
  if (h.use_run_block.n != h.block_cnt) throw FormatError{"useRunBlock length mismatch"};
  for (uint64_t i = 0; i < seq_alpha.n; ++i) { RawBitvector rb; parse_bitvector(c, rb); }   // _alphabetRB: only Rank needs them; the image is built from the decoded string
  GenWavelet comp;
  parse_gen_wavelet(c, comp);
  GenAlphabet fm_alpha, plain;
  parse_gen_alphabet(c, fm_alpha);
  parse_gen_alphabet(c, plain);
  if (plain.n < 2 || plain.n > 32 || fm_alpha.n != plain.n || h.alphabet_bits == 0 || h.alphabet_bits > 5 || (1ull << h.alphabet_bits) < plain.n)
    throw FormatError{"unexpected protein alphabet"};
  P.sigma = (uint32_t)plain.n;
  P.bits = (uint32_t)h.alphabet_bits;
  memset(P.code_of, 255, sizeof(P.code_of));
  for (uint32_t k = 0; k < P.sigma; ++k) {
    P.list[k] = plain.list[k];
    if (plain.code[(unsigned char)plain.list[k]] != (int32_t)k) throw FormatError{"plain coder is not the identity on its list"};
    P.code_of[(unsigned char)plain.list[k]] = (uint8_t)k;
  }
  for (uint64_t k = 0; k < fm_alpha.n; ++k) if (P.code_of[(unsigned char)fm_alpha.list[k]] == 255) throw FormatError{"alphabets of the index disagree"};
  P.C.resize(P.sigma + 1);
  c.copy(P.C.data(), (P.sigma + 1) * 8);
  if (P.code_of[(unsigned char)h.last_chr] == 255) throw FormatError{"lastChr not in the alphabet"};
  h.last_code = P.code_of[(unsigned char)h.last_chr];
  // _FMIndexAuxData
  const uint64_t aux_n = c.get<uint64_t>();
  c.get<int32_t>();
  h.sample_rate = c.get<int32_t>();
  h.sample_size = c.get<uint64_t>();
  h.precompute_width = c.get<uint64_t>();
  h.precompute_size = c.get<uint64_t>();
  h.adjusted_sa0 = c.get<uint64_t>();
  if (aux_n != h.n || h.sample_rate <= 0) throw FormatError{"inconsistent aux header"};
  if (h.precompute_width * P.bits > 30 || h.precompute_size != (h.precompute_width ? (1ull << (P.bits * h.precompute_width)) : 0))
    throw FormatError{"unsupported ftab width"};
  c.get<uint64_t>();
  h.sampled_bits = c.get<int32_t>();
  h.sampled_n = c.get<uint64_t>();
  if (h.sampled_bits <= 0 || h.sampled_bits > 64) throw FormatError{"bad sampledSA element width"};
  const uint64_t sw = ceil_div(h.sampled_n * (uint64_t)h.sampled_bits, 64);
  c.words(h.sampled_words, sw, false, 2);          // (a copy: the protein parser decodes everything into the process anyway)
  h.ftab.resize(2 * h.precompute_size);
  c.copy(h.ftab.data(), h.ftab.size() * 8);
  const uint64_t max_lcp = c.get<uint64_t>();
  if (max_lcp > 0) c.skip(2 * ceil_div(h.n, 64) * 8);
  const uint64_t sel_cnt = c.get<uint64_t>();
  h.selected_filter_rate = c.get<int32_t>();
  if (sel_cnt != 0) throw FormatError{"protein index with selectedSA rows (the reference never writes those, Builder.hpp:224)"};
  h.has_end_marker = !c.eof() && c.get<uint8_t>() != 0;
  if (h.has_end_marker) {
    c.get<uint64_t>();
    P.end_marker_bits = c.get<int32_t>();
    P.end_marker_n = c.get<uint64_t>();
    if (P.end_marker_bits <= 0 || P.end_marker_bits > 64) throw FormatError{"bad endMarkerSA element width"};
    const uint64_t ew = ceil_div(P.end_marker_n * (uint64_t)P.end_marker_bits, 64);
    P.end_marker_words.assign(ew + 2, 0);
    c.copy(P.end_marker_words.data(), ew * 8);
  }
  // ---- decode the BWT (Sequence_RunBlockOneTree::Access for i = 0, 1, ...: the compressed sequence is visited in order, so
  // every wavelet node is read front to back and no rank is needed).  Returns "" or what went wrong.
  auto decode = [&](std::vector<uint8_t> &bwt) -> std::string {
    std::vector<uint64_t> cur(comp.node.size(), 0);
    uint64_t produced = 0;
    std::string err;
    auto next_symbol = [&]() -> int {
      if (comp.node.empty()) { err = "protein index without a compressed sequence"; return -1; }
      int ti = 0;
      uint64_t code = 0;
      int depth = 0;
      while (ti != -1) {
        if (ti < 0 || (size_t)ti >= comp.node.size() || cur[(size_t)ti] >= comp.node[(size_t)ti].n || ++depth > 16) {
          err = "wavelet tree walk left the tree (node " + std::to_string(ti) + " of " + std::to_string(comp.node.size()) + ", bit " +
                std::to_string(ti >= 0 && (size_t)ti < comp.node.size() ? cur[(size_t)ti] : 0) + " of " +
                std::to_string(ti >= 0 && (size_t)ti < comp.node.size() ? comp.node[(size_t)ti].n : 0) + ", depth " + std::to_string(depth) +
                ", symbol " + std::to_string(produced) + " of " + std::to_string(comp.n) + ", n " + std::to_string(h.n) + ")";
          return -1;
        }
        const unsigned bit = bit_at(comp.node[(size_t)ti], cur[(size_t)ti]++);
        code = (code << 1) | bit;
        ti = bit ? comp.child1[(size_t)ti] : comp.child0[(size_t)ti];
      }
      ++produced;
      if (code >= comp.alphabet.n) { err = "wavelet code outside the alphabet"; return -1; }
      const uint8_t k = P.code_of[(unsigned char)comp.alphabet.list[code]];
      if (k == 255) { err = "wavelet symbol outside the plain alphabet"; return -1; }
      return k;
    };
    bwt.resize(h.n);
    for (uint64_t bi = 0; bi < h.block_cnt; ++bi) {
      const uint64_t lo = bi * h.b, hi = std::min(h.n, lo + h.b);
      if (bit_at(h.use_run_block, bi)) {
        const int k = next_symbol();
        if (k < 0) return err;
        for (uint64_t i = lo; i < hi; ++i) bwt[i] = (uint8_t)k;
      } else {
        for (uint64_t i = lo; i < hi; ++i) { const int k = next_symbol(); if (k < 0) return err; bwt[i] = (uint8_t)k; }
      }
    }
    if (produced != comp.n) return "run-block component lengths do not add up";
    // the partial sums must be those of the decoded string (FMIndex::Init, FMIndex.hpp:335-344)
    std::vector<uint64_t> cnt(P.sigma + 1, 0);
    for (uint64_t i = 0; i < h.n; ++i) ++cnt[bwt[i] + 1];
    for (uint32_t k = 1; k <= P.sigma; ++k) cnt[k] += cnt[k - 1];
    if (cnt != P.C) return "alphabet partial sums do not match the decoded BWT";
    return "";
  };
  const std::string derr = decode(P.bwt);
  if (!derr.empty()) {
    // A decode of a file that parsed field by field up to here should not fail; when it does, say what the failure was made of
    // (round 3 saw this check fail twice in ~110 GPU-side test runs on files that parsed before and after): does this process's
    // copy of the bit strings still equal the mapping, does the mapping equal what read() returns, does a second decode pass?
    std::string note = "; forensics:";
    size_t bad_copies = 0, first_bad = ~(size_t)0;
    auto check_copy = [&](const RawBitvector &bv, size_t which) {
      if (bv.n && memcmp(bv.bits.data(), c.base() + bv.file_off, ceil_div(bv.n, 64) * 8) != 0) { ++bad_copies; if (first_bad == ~(size_t)0) first_bad = which; }
    };
    check_copy(h.use_run_block, 0);
    for (size_t k = 0; k < comp.node.size(); ++k) check_copy(comp.node[k], k + 1);
    note += bad_copies ? " " + std::to_string(bad_copies) + " of " + std::to_string(comp.node.size() + 1) + " bit strings in memory differ from the file mapping (first: #" + std::to_string(first_bad) + ")"
                       : " in-memory bit strings equal the file mapping";
    {
      std::vector<uint8_t> buf(c.size());
      size_t got = 0;
      while (got < buf.size()) { const ssize_t r = pread(c.fd(), buf.data() + got, buf.size() - got, (off_t)got); if (r <= 0) break; got += (size_t)r; }
      if (got != buf.size()) note += ", pread() of the file came up short";
      else {
        size_t d = 0;
        while (d < buf.size() && buf[d] == c.base()[d]) ++d;
        note += d == buf.size() ? ", mapping equals pread()" : ", mapping differs from pread() at byte " + std::to_string(d);
      }
    }
    std::vector<uint8_t> again;
    const std::string e2 = decode(again);
    note += e2.empty() ? ", a second decode of the same memory PASSES" : ", a second decode fails too (" + e2 + ")";
    throw FormatError{derr + note};
  }
}

std::string get_string(Cursor &c) {
  uint64_t len = c.get<uint64_t>();
  std::string s(len, '\0');
  if (len) c.copy(&s[0], len);
  return s;
}

// rank enum of Taxonomy.hpp:25-59 and the rank-number table of :94-143
enum {
  R_UNKNOWN = 0, R_STRAIN, R_SPECIES, R_GENUS, R_FAMILY, R_ORDER, R_CLASS, R_PHYLUM, R_KINGDOM, R_DOMAIN, R_FORMA,
  R_INFRA_CLASS, R_INFRA_ORDER, R_PARV_ORDER, R_SUB_CLASS, R_SUB_FAMILY, R_SUB_GENUS, R_SUB_KINGDOM, R_SUB_ORDER,
  R_SUB_PHYLUM, R_SUB_SPECIES, R_SUB_TRIBE, R_SUPER_CLASS, R_SUPER_FAMILY, R_SUPER_KINGDOM, R_SUPER_ORDER,
  R_SUPER_PHYLUM, R_TRIBE, R_VARIETAS, R_LIFE, R_ACELLULAR_ROOT, R_MAX
};

void fill_rank_num(uint8_t *t) {
  struct { int level; std::vector<int> ranks; } groups[] = {
      {0, {R_SUB_SPECIES, R_STRAIN}}, {1, {R_SPECIES}}, {2, {R_SUB_GENUS, R_GENUS}},
      {3, {R_SUB_FAMILY, R_FAMILY, R_SUPER_FAMILY}},
      {4, {R_SUB_ORDER, R_INFRA_ORDER, R_PARV_ORDER, R_ORDER, R_SUPER_ORDER}},
      {5, {R_INFRA_CLASS, R_SUB_CLASS, R_CLASS, R_SUPER_CLASS}}, {6, {R_SUB_PHYLUM, R_PHYLUM, R_SUPER_PHYLUM}},
      {7, {R_SUB_KINGDOM, R_KINGDOM}}, {8, {R_SUPER_KINGDOM, R_ACELLULAR_ROOT, R_DOMAIN}},
      {9, {R_FORMA, R_SUB_TRIBE, R_TRIBE, R_VARIETAS, R_LIFE, R_UNKNOWN}}};
  for (auto &g : groups) for (int r : g.ranks) t[r] = (uint8_t)g.level;
}

void parse_taxonomy(const std::string &path, Taxonomy &t) {
  Cursor c(path);
  fill_rank_num(t.rank_num);
  t.node_cnt = c.get<uint64_t>();
  t.seq_cnt = c.get<uint64_t>();
  t.extra_seq_cnt = c.get<uint64_t>();
  t.parent.resize(t.node_cnt);
  t.rank.resize(t.node_cnt);
  for (uint64_t i = 0; i < t.node_cnt; ++i) {   // TaxonomyNode: u64 parent | u8 rank | u8 leaf | u8 pad[6]
    t.parent[i] = c.get<uint64_t>();
    t.rank[i] = c.get<uint8_t>();
    c.skip(7);
  }
  uint64_t map_n = c.get<uint64_t>();
  t.orig_taxid.resize(map_n);
  c.copy(t.orig_taxid.data(), map_n * 8);
  t.tax_name.reserve(t.node_cnt);
  for (uint64_t i = 0; i < t.node_cnt; ++i) t.tax_name.push_back(get_string(c));
  t.seq_to_tax.resize(t.seq_cnt);
  c.copy(t.seq_to_tax.data(), t.seq_cnt * 8);
  uint64_t names = t.seq_cnt + t.extra_seq_cnt;
  t.seq_name.reserve(names);
  for (uint64_t i = 0; i < names; ++i) t.seq_name.push_back(get_string(c));
  t.root = t.node_cnt;
  for (uint64_t i = 0; i < t.node_cnt; ++i) if (t.parent[i] == i) { t.root = i; break; }
  if (map_n < t.node_cnt || t.root >= t.node_cnt) throw FormatError{"taxonomy without a root / id map too short"};
}

bool is_protein(const std::string &prefix) {   // Classifier::IsProteinDatabase (Classifier.hpp:867-895)
  FILE *fp = fopen((prefix + ".4.cfr").c_str(), "r");
  if (!fp) return false;
  char key[128], val[128];
  bool ret = false;
  while (fscanf(fp, "%127s %127s", key, val) == 2)
    if (!strcmp(key, "sequence_type") && !strcmp(val, "amino_acid")) ret = true;
  fclose(fp);
  return ret;
}

}  // namespace

uint64_t index_digest(const HostIndex &h) {
  uint64_t x = 1469598103934665603ull;
  auto mix = [&](const void *p, size_t bytes) {
    const uint8_t *b = (const uint8_t *)p;
    // 8 bytes at a time (the bit strings of a large index are GBs): FNV-1a over words, then over the tail bytes
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) { uint64_t w; memcpy(&w, b + i, 8); x = (x ^ w) * 1099511628211ull; }
    for (; i < bytes; ++i) x = (x ^ b[i]) * 1099511628211ull;
  };
  auto num = [&](uint64_t v) { mix(&v, 8); };
  auto bitvec = [&](const RawBitvector &bv) { num(bv.n); if (bv.n) mix(bv.bits.data(), ceil_div(bv.n, 64) * 8); };
  auto wavelet = [&](const RawWavelet &w) { num(w.n); num((uint64_t)w.node_cnt); for (int k = 0; k < w.node_cnt; ++k) { num((uint64_t)w.children[k][0]); num((uint64_t)w.children[k][1]); bitvec(w.node[k]); } };
  num(h.n); num(h.alphabet_bits); num(h.first_isa); num((uint64_t)(unsigned char)h.last_chr); num(h.last_code);
  mix(h.C, sizeof(h.C));
  num(h.b); num(h.block_cnt);
  bitvec(h.use_run_block);
  if (!h.prot.enabled) { wavelet(h.wavelet_seq); wavelet(h.run_block_seq); }
  num((uint64_t)h.sample_rate); num(h.sample_size); num(h.precompute_width); num(h.precompute_size); num(h.adjusted_sa0);
  num((uint64_t)h.sampled_bits); num(h.sampled_n);
  mix(h.sampled_words.data(), ceil_div(h.sampled_n * (uint64_t)h.sampled_bits, 64) * 8);      // (not the two words behind them: zeros in a copy, the file's next bytes in the mapping)
  mix(h.ftab.data(), h.ftab.size() * 8);
  num((uint64_t)h.selected_filter_rate);
  mix(h.selected_rows.data(), h.selected_rows.size() * 8);
  mix(h.selected_vals.data(), h.selected_vals.size() * 8);
  num(h.has_end_marker);
  if (h.prot.enabled) {
    const ProteinPart &P = h.prot;
    num(P.sigma); num(P.bits); mix(P.list, sizeof(P.list)); mix(P.code_of, sizeof(P.code_of));
    mix(P.C.data(), P.C.size() * 8); mix(P.bwt.data(), P.bwt.size());
    mix(P.end_marker_words.data(), P.end_marker_words.size() * 8); num((uint64_t)P.end_marker_bits); num(P.end_marker_n);
  }
  const Taxonomy &t = h.tax;
  num(t.node_cnt); num(t.seq_cnt); num(t.extra_seq_cnt); num(t.root);
  mix(t.parent.data(), t.parent.size() * 8); mix(t.rank.data(), t.rank.size()); mix(t.orig_taxid.data(), t.orig_taxid.size() * 8);
  for (const auto &q : t.tax_name) { num(q.size()); mix(q.data(), q.size()); }
  mix(t.seq_to_tax.data(), t.seq_to_tax.size() * 8);
  for (const auto &q : t.seq_name) { num(q.size()); mix(q.data(), q.size()); }
  num((uint64_t)h.params.min_hit_len); num((uint64_t)h.score_hit_len_adjust);
  return x;
}

HostIndex *load_index(const std::string &prefix, const cfr_params *params) {
  const bool protein = is_protein(prefix);          // Classifier::IsProteinDatabase decides the sequence class (CentrifugerClass.cpp:1001-1004)
  HostIndex *h = new HostIndex();
  try {
    if (protein) parse_fm_protein(prefix + ".1.cfr", *h);
    else parse_fm(prefix + ".1.cfr", *h);
    parse_taxonomy(prefix + ".2.cfr", h->tax);
  } catch (...) {
    delete h;
    throw;
  }
  if (params) h->params = *params; else cfr_params_default(&h->params);
  if (protein) h->score_hit_len_adjust /= 3;         // Classifier.hpp:928-932
  if (h->params.min_hit_len <= 0 && protein) {
    // Classifier::InferMinHitLen (Classifier.hpp:113-129) with alphabet size sigma, starting at 11 (Kaiju's default)
    int m = 11;
    uint64_t kmerspace = 1;
    { uint64_t px = h->prot.sigma; int y = m; while (y) { if (y & 1) kmerspace *= px; px *= px; y >>= 1; } }   // Utils::PowerInt
    kmerspace /= 2;
    for (; m <= 32; ++m) {
      if (kmerspace >= 100 * h->n) break;
      kmerspace *= h->prot.sigma;
    }
    h->params.min_hit_len = m;
  }
  if (h->params.min_hit_len <= 0) {
    // Classifier::InferMinHitLen (Classifier.hpp:113-129): smallest m >= 23 with 4^m / 2 >= 100 n
    int m = 23;
    unsigned __int128 space = ((unsigned __int128)1 << (2 * m)) / 2;
    uint64_t kmerspace = (uint64_t)space;    // 4^23/2 fits in 64 bits
    for (; m <= 32; ++m) {
      if (kmerspace >= 100 * h->n) break;
      kmerspace *= 4;                        // wraps like the reference's uint64_t
    }
    h->params.min_hit_len = m;
  }
  return h;
}

}  // namespace cfr
