// cfr_index.hpp — host-side view of a Centrifuger index (<prefix>.1.cfr / .2.cfr / .4.cfr).
//
// Clean-room parser of the on-disk format written by the reference's FMIndex::Save
// (compactds/FMIndex.hpp:571-586), Sequence_RunBlock::Save (Sequence_RunBlock.hpp:468-476),
// Bitvector_Plain::Save (Bitvector_Plain.hpp:182-196), _FMIndexAuxData::Save (FMIndex.hpp:100-134)
// and Taxonomy::Save (Taxonomy.hpp:1238-1257); grammar in SURVEY.md Appendix A.
//
// The parser keeps the compressed components verbatim; the BWT string itself is never decoded on the host (the
// device expands its flat occurrence image from the uploaded components, cfr_device.hip / k_occ_expand).
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <cstring>
#include <vector>

#include "../../include/cfr_hip.h"

namespace cfr {

// Words of a bit string of the index.  Round 5: they stay where they are - in the read-only mapping of the .1.cfr file, which
// HostIndex keeps for its lifetime - instead of being copied into this process (a 40 Gbp index holds 15 GB of them: eight ranks of one
// node each made their own copy at once, 6.3 s per open against 2.5 s alone and 120 GB of host memory; mapped, the ranks share the
// page cache's pages and an open costs the headers).  Every string is mapped whatever its alignment in the file (words are read
// bytewise, see data() below); a caller that wants its own (CFR_INDEX_COPY=1 behind CFR_DEBUG_ENV, the protein parser) gets a copy
// as before: a vector whose resize() does not zero-fill first.  Move-only: a copy would keep p_ pointing into the source's buffer.
template <class T> struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  template <class U, class... A> void construct(U *p, A &&...a) {
    if constexpr (sizeof...(A) == 0) ::new ((void *)p) U; else ::new ((void *)p) U(std::forward<A>(a)...);
  }
};
class RawWords {
 public:
  RawWords() = default;
  RawWords(const RawWords &) = delete;
  RawWords &operator=(const RawWords &) = delete;
  RawWords(RawWords &&) noexcept = default;                  // (a vector's buffer moves with it: p_ stays valid)
  RawWords &operator=(RawWords &&) noexcept = default;
  // the words as BYTES: a mapped string stands where the file has it, and the .cfr format aligns nothing (a one-byte field in the header
  // leaves every bit string of a nucleotide index at an odd offset), so nothing may take this pointer for a uint64_t array - copies,
  // comparisons and uploads go through it bytewise, single words through operator[]
  const void *data() const { return p_; }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  uint64_t operator[](size_t i) const { uint64_t v; memcpy(&v, static_cast<const char *>(p_) + 8 * i, 8); return v; }     // (an unaligned load where the string is mapped)
  bool mapped() const { return n_ != 0 && own_.empty(); }
  void map(const void *q, size_t n) { own_.clear(); p_ = q; n_ = n; }                          // n words at q, which outlive this object
  uint64_t *alloc(size_t n) { own_.resize(n); p_ = own_.data(); n_ = n; return own_.data(); }   // n words of its own, not initialised
 private:
  const void *p_ = nullptr;
  size_t n_ = 0;
  std::vector<uint64_t, NoInitAlloc<uint64_t>> own_;
};

struct RawBitvector {          // Bitvector_Plain as stored (its DS_Rank9 blocks are checked for size and skipped: the device image
  uint64_t n = 0;              // bits                             counts its own rank lines)
  RawWords bits;               // ceil(n/64)
  uint64_t file_off = 0;       // where the words stand in the .1.cfr file (the decode's forensics compare a copy with the file again)
};

struct RawWavelet {            // Sequence_WaveletTree<Bitvector_Plain> for sigma=4: 3 nodes
  uint64_t n = 0;
  int node_cnt = 0;
  int children[3][2] = {{-1, -1}, {-1, -1}, {-1, -1}};
  RawBitvector node[3];
};

struct Taxonomy {
  uint64_t node_cnt = 0, seq_cnt = 0, extra_seq_cnt = 0, root = 0;
  std::vector<uint64_t> parent;
  std::vector<uint8_t> rank;
  std::vector<uint64_t> orig_taxid;
  std::vector<std::string> tax_name;
  std::vector<uint64_t> seq_to_tax;
  std::vector<std::string> seq_name;
  uint8_t rank_num[64] = {0};   // Taxonomy::_taxRankNum (Taxonomy.hpp:94-143)
};

// Protein index (FMIndex<Sequence_RunBlockOneTree>, CentrifugerClass.cpp:1001-1004): the parser checks the components
// (Sequence_RunBlockOneTree.hpp:499-513) and decodes the BWT once into plain codes; the device builds its occurrence image
// for the 21-symbol alphabet from those.
struct ProteinPart {
  bool enabled = false;
  uint32_t sigma = 0, bits = 0;          // alphabet size (21: '$' + 20 amino acids), bits per code in the ftab key
  char list[64] = {0};                   // plain coder: code -> character
  uint8_t code_of[256];                  // character -> code, 255 = not in the alphabet (Alphabet::IsIn false)
  std::vector<uint64_t> C;               // sigma + 1 partial sums (_plainAlphabetPartialSum)
  std::vector<uint8_t> bwt;              // n plain codes
  std::vector<uint64_t> end_marker_words;   // endMarkerSA (FixedSizeElemArray), rows [0, end_marker_n)
  int32_t end_marker_bits = 0;
  uint64_t end_marker_n = 0;
};

struct HostIndex {
  // FMIndex scalars
  uint64_t n = 0, alphabet_bits = 0, first_isa = 0;
  char last_chr = 0;
  uint8_t last_code = 0;
  uint64_t C[5] = {0, 0, 0, 0, 0};
  // run-block components
  uint64_t b = 0, block_cnt = 0;
  RawBitvector use_run_block;
  RawWavelet wavelet_seq, run_block_seq;
  // aux
  int32_t sample_rate = 0;
  uint64_t sample_size = 0, precompute_width = 0, precompute_size = 0, adjusted_sa0 = 0;
  int32_t sampled_bits = 0;
  uint64_t sampled_n = 0;
  RawWords sampled_words;
  std::vector<uint64_t> ftab;           // pairs (start, count)
  int32_t selected_filter_rate = 1024;
  std::vector<uint64_t> selected_rows, selected_vals;   // ascending rows
  bool has_end_marker = false;
  ProteinPart prot;
  // taxonomy + parameters
  Taxonomy tax;
  cfr_params params;
  int score_hit_len_adjust = 15;   // Classifier.hpp:848
  std::shared_ptr<void> file_mapping;   // the .1.cfr file, mapped read-only: the RawWords above point into it (empty: every string is a copy)
};

// Throws std::runtime_error with a message; cfr_capi.cpp maps it to cfr_status.
struct FormatError { std::string msg; };
struct IoError { std::string msg; };

HostIndex *load_index(const std::string &prefix, const cfr_params *params);
// FNV-1a over everything the parser produced (scalars, bit strings, tables, taxonomy): two opens of the same files must agree
uint64_t index_digest(const HostIndex &h);


}  // namespace cfr
