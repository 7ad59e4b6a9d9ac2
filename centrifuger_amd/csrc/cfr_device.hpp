// cfr_device.hpp — the index image in HBM and the batch pipeline that runs on it (gfx950).
//
// Data layout (profiles/HISTORY.md §3):
//   occ      : one 64-byte record per 128 BWT symbols
//                u64 mid[4]   #c in B[0 .. 128*r + 64)      (bit 63 of mid[0]: record holds a selectedSA row)
//                u64 lo0, hi0 bit planes of symbols   0..63  (bit k of lo = low code bit of symbol k)
//                u64 lo1, hi1 bit planes of symbols 64..127
//              => FMIndex::Rank(c, p) / Sequence::Access(p) touch exactly ONE 64-byte record:
//                 8 B (mid[c]) + 16 B (the half that holds symbol p).
//              Expanded ON THE DEVICE at load time from the uploaded run-block components (k_occ_expand + a scan).
//   ftab     : (start, count) u64 pairs, 4^w entries          (FMIndex.hpp:27)
//   ftabx    : DERIVED at load time, never on disk: for every K-mer (K = log4(n)+2, at most 16, > w) the exact state
//              (l, sp, ep) FMIndex::BackwardSearch reaches after its first K characters (ftab lookup +
//              K-w extends, including where it stopped).  One 16-byte gather replaces K-w+1 dependent ones.
//   sampled  : bit-packed seqIds, one per sample_rate rows    (FixedSizeElemArray.hpp:102-105)
//   sa/text2 : DERIVED at load time (list ranking over the LF permutation by rulers, k_ruler_walk/_jump/_fill):
//              SA[row] (u32 entries for n < 2^32, 36-bit packed entries up to 2^36) and the 2-bit text.  A BWT range of a few rows
//              is then extended by comparing the read with the text 48 bases per step (the rows' suffix positions move in
//              lock step), instead of one LF step per base.  A hit found that way is kept in TEXT-POSITION space: its rows
//              are "virtual rows" (the range the search had when it moved to the text, which of its rows survived, and how
//              many characters were matched since: virt_text_pos), so no inverse suffix array exists anywhere.
//   steps    : DERIVED at load time: what FMIndex::BackwardToSampledSA returns, as a step function of the TEXT POSITION of
//              the row it starts from (breakpoints: position 0 and the selectedSA positions; the premise is verified on
//              every sampled row, k_memo_check).  locate(row) = steps(SA[row]); locate(virtual row) = steps(SA[entry row] - matched).
//   loc_memo : DERIVED at load time when HBM allows: the value FMIndex::BackwardToSampledSA returns for every memo_rate-th row
//              (memo_rate = 1 when n*4 bytes fit the budget).  The LF-walk from row i passes through the same rows
//              as the reference's, so stopping at a memoised row returns exactly what the full walk would.
//   sel_rows / sel_vals : sorted selectedSA pairs             (FMIndex.hpp:34)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "cfr_index.hpp"

namespace cfr {

// Run-block image (CFR_LAYOUT=rb): the reference's own components kept compressed in HBM.  Every bitvector
// (useRunBlock; 3 wavelet nodes of the non-run symbols; 3 of the run symbols) is stored as 64-byte "rank lines":
// u64 #ones before the line + 7 payload words (448 bits), so one bit-rank / bit-access = one line
// (DS_Rank9 needs the counter pair and the word from two arrays).
struct RankLines { const uint64_t *lines; uint64_t nbits; };
struct RbView {
  RankLines use, plain[3], runs[3];
  uint64_t b;                        // run-block size (Sequence_RunBlock::_b)
  uint64_t block_cnt;
  const uint64_t *sel_filter;        // 1 bit per filter_rate rows (FMIndex.hpp:35-36, 166-176)
  uint32_t filter_rate;
  uint32_t enabled;                  // 0: flat occ image
};

// Protein image (FMIndex<Sequence_RunBlockOneTree>, sigma = 21): the decoded BWT as five bit planes per 64 symbols plus the
// per-block symbol counts.  n < 2^32 (and sigma <= 22): ONE 128-byte record per 64 symbols - 22 u32 counts (88 bytes) + the five
// planes (40 bytes) - so that Rank(c, p) and Access(p) touch one line (one fabric request).  Larger texts: the planes and u64
// counts in two arrays (two gathers per rank).
struct ProtView {
  uint32_t enabled, sigma, bits, endmarker_bits;
  const uint64_t *rec;       // 16 u64 per 64 symbols: u32 counts[22] (count of `code` in B[0, 64 blk)), then planes at word 11; nullptr: the two arrays below
  const uint64_t *planes;    // 8 u64 per 64 symbols: plane k = bit k of the plain code of every symbol (5 used)
  const uint64_t *counts;    // counts[blk * 32 + code] = number of `code` in B[0, 64 blk)
  const uint64_t *endmarker; // endMarkerSA (FixedSizeElemArray words): the sequence id stored at row i < endmarker_n
  uint64_t endmarker_n;
  const uint8_t *code_of;    // 256 entries: character -> plain code, 255 = not in the alphabet
  uint64_t C[33];            // _plainAlphabetPartialSum
  char list[32];             // code -> character
};

// rows >= kVirtRow are virtual: rows of a hit that was finished on the text (layout: virt_text_pos in cfr_kernels.hip.inc)
constexpr uint64_t kVirtRow = 1ull << 63;

// BackwardToSampledSA as a step function of the text position (see cfr_kernels.hip.inc, steps_value)
struct StepView {
  const uint64_t *pos, *val;   // breakpoints by ascending text position, pos[0] = 0 ; nullptr = not available
  const uint32_t *bucket;      // bucket[b] = index of the last breakpoint with pos <= (b << shift)
  uint64_t cnt, n;
  uint32_t shift;
};

struct DevView {            // passed by value to kernels
  uint64_t n, first_isa, adjusted_sa0;
  uint64_t C[5];
  RbView rb;                // run-block image (alternative to occ)
  ProtView prot;            // protein image (alternative to both)
  const uint64_t *occ;      // 8 u64 per record
  const uint64_t *ftab;     // 2 u64 per entry
  const uint64_t *ftabx;    // derived wide ftab: 2 u64 per K-mer = (sp, (count << 8) | l), or ONE u64 (ftabx_e8, see ftabx8_encode); nullptr = off
  uint32_t ftabx_width;     // K (> ftab_width)
  uint32_t ftabx_e8;        // 1: 8-byte entries
  const uint64_t *sampled;
  const uint32_t *loc_memo; // derived: memo[j / memo_rate] = BackwardToSampledSA(j) for j % memo_rate == 0; nullptr = off
  uint32_t memo_shift;      // log2(memo_rate)
  // derived text-mode tables: suffix array (u32 entries when n < 2^32, 36-bit packed entries otherwise) and the 2-bit text; nullptr = off
  const uint32_t *sa32;
  const uint32_t *sa36;     // entry i at bits [36 i, 36 i + 36) of the little-endian bit string
  StepView steps;           // locate as a function of the text position
  const uint64_t *text2;    // symbol p at bits 2(p%32) of word p/32
  const uint8_t *text8;     // protein index: the text as plain codes, one byte per symbol (16 zero bytes in front); nullptr otherwise
  uint32_t text_min_l;      // a search switches to text comparison once it has matched this many characters
  const uint64_t *sel_rows, *sel_vals;
  uint64_t sel_cnt;
  uint32_t last_code, ftab_width, sampled_bits, sample_rate;
  int32_t min_hit_len, score_adjust;
  uint64_t max_entries;     // (size_t)(maxResult * maxResultPerHitFactor)
  int32_t locate_all;       // factor <= 0 || maxResult <= 0
  // taxonomy side tables + tail parameters (Taxonomy.hpp:61-92; Classifier.hpp:17-38)
  int32_t max_result;
  const uint64_t *tax_parent, *tax_orig, *seq_to_tax;
  const uint8_t *tax_rank;
  const uint32_t *tax_depth;  // derived: steps from a node to the root (what Taxonomy::LCA's walk counts)
  uint64_t node_cnt, seq_cnt, tax_root;
  uint64_t secondary_hit_len;
  double secondary_factor;
  uint8_t rank_num[32];
  // --expand-taxid: the pool the tail appends its lists to (exp_append in cfr_kernels.hip.inc); nullptr = not wanted
  uint64_t *exp_pool;
  unsigned long long *exp_cursor;
  uint64_t exp_cap;
};

struct TailEntry { uint64_t seq_id, score; int32_t hit_length, k; };

struct HipError { std::string msg; int code; };
struct CapacityError { std::string msg; };
void *host_alloc_pinned(size_t bytes);
void host_free_pinned(void *p);

// host buffers of a batch whose bases are streamed to the device sub-batch by sub-batch (DeviceIndex::classify_host)
struct HostSrc { const uint8_t *b1; const uint64_t *o1; const uint8_t *b2; const uint64_t *o2;
                 const uint64_t *p1 = nullptr, *p2 = nullptr; uint64_t *stage1 = nullptr, *stage2 = nullptr; };      // p1 / p2: the bases in packed form (k_pack_reads' blocks) instead of b1 / b2; stage: where they land on the device when they are not the search's own blocks

class DeviceIndex {
 public:
  DeviceIndex(const HostIndex &h, int device, const cfr_device_options &opt);
  ~DeviceIndex();

  const HostIndex &host() const { return *host_; }
  int device() const { return device_; }
  uint64_t device_bytes() const { return device_bytes_; }
  hipStream_t stream() const { return stream_; }

  void rank_batch(const char *chars, const uint64_t *pos, const uint8_t *incl, size_t n, uint64_t *out_rank, char *out_access);
  void backward_search_batch(const uint8_t *bases, const uint64_t *offsets, const uint32_t *m, size_t n,
                             uint64_t *out_l, uint64_t *out_sp, uint64_t *out_ep);
  void locate_rows(const uint64_t *rows, size_t n, uint64_t *out_val, uint32_t *out_steps);

  // consistency of the derived tables (SA / ISA / text / locate memo) against the BWT itself; see k_selfcheck
  void selfcheck(uint64_t out[6]);

  // Runs search (+adjust, strand choice) for a batch whose reads are in device memory.
  // On return the per-read final hits are in host vectors (compacted), plus located seqIds per hit if want_rows.
  struct BatchOut {
    std::vector<uint64_t> hit_begin;     // n+1
    std::vector<cfr_hit> hits;
    std::vector<uint64_t> row_begin;     // per hit: hits.size()+1 (only when want_rows)
    std::vector<uint64_t> row_vals;      // located seqIds (only when want_rows)
    std::vector<int32_t> read_len;       // query length per read (r1 + r2)
  };
  void run_batch(const uint8_t *d_bases1, const uint64_t *d_offs1, const uint8_t *d_bases2, const uint64_t *d_offs2,
                 size_t n, uint64_t total1, uint64_t total2, bool want_rows, BatchOut &out);

  // Full Query on the device: search + locate + tail; results/matches land in caller memory
  // (fast when that memory came from cfr_host_alloc).  matches: stride entries per read.
  void classify_device(const uint8_t *d_bases1, const uint64_t *d_offs1, const uint8_t *d_bases2, const uint64_t *d_offs2,
                       size_t n, uint64_t total1, uint64_t total2, cfr_result *results, cfr_match *matches, size_t match_cap,
                       size_t *match_extent, const struct HostSrc *src = nullptr, bool compact = false);   // compact: results / matches are cfr_result_compact / cfr_match_compact arrays
  void classify_host(const uint8_t *bases1, const uint64_t *offs1, const uint8_t *bases2, const uint64_t *offs2, size_t n,
                     cfr_result *results, cfr_match *matches, size_t match_cap, size_t *match_extent);
  // the same with the bases as packed blocks (cfr_classify_batch_packed): half the bytes over PCIe, unpacked on the device
  void classify_host_packed(const uint64_t *packed1, const uint64_t *offs1, const uint64_t *packed2, const uint64_t *offs2, size_t n,
                            cfr_result *results, cfr_match *matches, size_t match_cap, size_t *match_extent);

  // convenience: host buffers -> device, then run_batch
  void run_batch_host(const uint8_t *bases1, const uint64_t *offs1, const uint8_t *bases2, const uint64_t *offs2,
                      size_t n, bool want_rows, BatchOut &out);

  // --expand-taxid: the records the last classify call appended (slot, count, ids ...), on the host; empty without output_expanded
  std::vector<uint64_t> expanded_raw_;
  cfr_batch_stats last_stats{};
  // reads of the last compact call whose values did not fit the narrow layout, in the wide one (cfr_compact_wide_reads)
  static constexpr size_t kWideSideCap = 65536;
  std::vector<uint32_t> wide_idx_;
  std::vector<cfr_result> wide_res_;
  std::vector<cfr_match> wide_match_;
  uint64_t wide_total_ = 0;

  // SDUST before the search of every classify call (off by default: the C-ABI contract is "already masked if the caller wants dust")
  void set_dust(bool on) { dust_ = on; }
  bool dust() const { return dust_; }
  // host buffers in/out: the device scan as a stand-alone entry (parity probe of k_dust)
  void dust_mask_host(uint8_t *bases, const uint64_t *offs, size_t n);

 private:
  void init(const HostIndex &h, const cfr_device_options &opt);
  void release();
  void *temp_alloc(size_t bytes);
  void temp_free(void *p);
  template <class T> T *dev_alloc(size_t count);
  template <class T> T *upload(const std::vector<T> &v);
  struct Staged { const uint8_t *b1; const uint64_t *o1; const uint8_t *b2; const uint64_t *o2; uint64_t t1, t2; };
  Staged stage_inputs(const uint8_t *b1, const uint64_t *o1, const uint8_t *b2, const uint64_t *o2, size_t n);
  // device stages shared by run_batch / classify_device; returns (nhits, nrows) and leaves device pointers in p_
  struct Pipe { uint64_t *hit_off, *fin_off, *row_off, *rows, *vals; cfr_hit *hits; uint64_t nhits, nrows; };
  void run_device_stages(const uint8_t *d_b1, const uint64_t *d_o1, const uint8_t *d_b2, const uint64_t *d_o2, size_t n,
                         uint64_t total1, uint64_t total2, bool want_rows, Pipe &p, std::vector<uint64_t> *hit_begin_host,
                         bool fused = false, bool row_space_only = false);
  struct SearchBuf { uint64_t *hit_off; cfr_hit *raw; uint32_t *chain_cnt; uint64_t cap_total; };
  // row_space_only: no text mode (every hit carries real BWT rows)
  SearchBuf launch_search(const uint8_t *d_b1, const uint64_t *d_o1, const uint8_t *d_b2, const uint64_t *d_o2, size_t n,
                          uint64_t total1, uint64_t total2, int par = 0, bool row_space_only = false);
  bool have_sa() const { return view_.sa32 != nullptr || view_.sa36 != nullptr; }
  // every row (real or virtual) of a hit can be located by one table access: memo at every row, or SA + step function
  bool locate_direct() const { return (view_.loc_memo && view_.memo_shift == 0) || (have_sa() && view_.steps.pos); }
  SearchBuf launch_search_protein(const uint8_t *d_b1, const uint64_t *d_o1, const uint8_t *d_b2, const uint64_t *d_o2, size_t n,
                                  uint64_t total1, uint64_t total2, bool row_space_only = false);
  void launch_post(const SearchBuf &sb, const uint8_t *d_b1, const uint64_t *d_o1, const uint8_t *d_b2, const uint64_t *d_o2, size_t n,
                   bool want_rows, Pipe &p, std::vector<uint64_t> *hit_begin_host, bool fused, hipStream_t st);
  std::vector<std::pair<size_t, size_t>> cut_pieces(size_t n, bool per_read_slots, size_t &sb, uint64_t total_bases = 0) const;
  void dust_on_device(uint8_t *d_bases, const uint64_t *d_offs, size_t n, hipStream_t st);
  void *scratch(size_t slot, size_t bytes);
  void *pinned(size_t bytes);
  void finish_stats(bool want_rows);
  void pack_inputs(const uint8_t *d_b1, uint64_t total1, const uint8_t *d_b2, uint64_t total2, bool pack_now = true);
  void expand_begin();                   // before the kernels of a classify call: pool in place, cursor zero, both views know it
  bool expand_end();                     // behind them: false = the pool was too small (it has been enlarged: run the call again)
  uint64_t exp_cap_ = 0;
  int exp_attempt_ = 0;                 // runs of the current batch that overflowed the --expand-taxid pool (classify_device repeats the batch: at most three)
  uint64_t last_stats_exp_retries_ = 0;  // such repeats since the image was made (diagnostics)
 public:
  uint64_t last_slow_reads_ = 0, last_team_reads_ = 0;   // of the last one-launch call: reads k_post_fast left to k_adjust_tail / to the team folds
 private:
  int prot_occ_[2] = {0, 0};             // resident blocks per CU of k_search_prot_sm<1|2, ..> on this device (asked once per image)
  bool one_launch_ready() const { return fused_tail_ && fused_post_ && locate_direct() && !view_.prot.enabled; }

  const HostIndex *host_;
  int device_;
  hipStream_t stream_ = nullptr;
  DevView view_{};
  DevView *d_view_ = nullptr;            // view_ in device memory (owned_): kernels that index its arrays take it by pointer
  std::vector<void *> owned_, temps_;     // device allocations of the image / load-time temporaries still alive
  uint64_t device_bytes_ = 0;
  struct Slot { void *p = nullptr; size_t cap = 0; };
  std::vector<Slot> slots_;
  static constexpr size_t kMaxSub = 16;
  hipEvent_t evs_[kMaxSub][9] = {};      // per sub-batch: 0-2 around the search, 8 and 3-7 around the stages behind it
  hipEvent_t *ev_ = nullptr;
  hipStream_t copy_stream_ = nullptr, h2d_stream_ = nullptr, tail_stream_ = nullptr, dust_stream_ = nullptr;
  hipStream_t search2_stream_ = nullptr, search_stream_ = nullptr;   // the second search stream; the stream launch_search enqueues on (nullptr: stream_)
  hipEvent_t prep_done_ = nullptr;
  bool two_search_now_ = false;
  hipEvent_t search_done_[2] = {};
  int tail_overlap_mode_ = -1;           // the post stage of a sub-batch beside the search of the next one: -1 = by heavy_frac_, 0 / 1
  bool overlap_now_ = false, blocks_forced_ = false;   // this call runs the post stage beside the next search; CFR_BLOCKS_PER_CU was given
  double heavy_frac_ = 0.0;              // share of the last call's reads that k_tail_heavy folded
  int tail_blocks_per_cu_ = 0;           // blocks per CU of the post stage when it runs beside a search (0: one lane per read)
  hipEvent_t tail_done_[2] = {}, copy_done_[2] = {}, h2d_done_[kMaxSub] = {}, copied_[kMaxSub] = {};
  size_t sub_batch_ = 1250000, taper_floor_ = 262144;
  const uint64_t *prot_o1_base_ = nullptr;      // offsets of the batch in flight (a sub-batch's read numbers are counted from here)
  uint64_t prot_reads_ = 0;
  uint64_t prot_total1_ = 0, prot_total2_ = 0;  // bases of the whole batch in flight (the translated codes of a protein search are indexed by absolute offsets)
  uint64_t piece_bases_max_ = 12000000000ull;   // bases a sub-batch may hold (its raw hit lists must fit beside the image)
  int num_cus_ = 256, blocks_per_cu_ = 7;
  uint64_t *packed1_ = nullptr, *packed2_ = nullptr;
  uint64_t nblk1_ = 0, nblk2_ = 0;
  double dust_mean_len_ = 0.0;          // mean read length of the batch being masked (dust_on_device: the screen is for short reads)
  bool search_split_default_ = false;   // the two-launch search (lookup stage + list stage) without CFR_SEARCH_SPLIT
  bool search_v1_ = false, fused_tail_ = true, fused_post_ = true, dust_ = false, team_tail_ = true;
  bool wide_ = false;                  // n >= 2^32: 36-bit SA entries and the WIDE search kernel
  uint64_t pool_cap_ = 0;              // scratch pool of k_adjust_tail in entries (0 = 8 per read of a sub-batch)
  void *pinned_ = nullptr;
  size_t pinned_cap_ = 0;
  const uint64_t *pre_hit_off_ = nullptr;   // launch_search: hit-list offsets of the sub-batch's reads computed for the whole batch already
  uint64_t pre_hit_base_ = 0;              //   (classify_device), and the offset of its first read (the raw buffer is the sub-batch's)
};

}  // namespace cfr
