// cfr_build.cpp — host half of the index writer: from the suffix-array products (cfr_build_sa.hip) to the four .cfr
// files, field for field what centrifuger-build writes:
//   run-block compression with the reference's automatic block size   Sequence_RunBlock.hpp:26-177, 231-358
//   wavelet trees (3 nodes, sigma = 4), plain bitvectors, rank9        Sequence_WaveletTree.hpp:303-310, Bitvector_Plain.hpp:182-196,
//                                                                      DS_Rank.hpp:206-247
//   FM-index scalars, sampled SA, ftab, selectedSA                     FMIndex.hpp:571-586, FMBuilder.hpp:209-313, Builder.hpp:224-234
//   taxonomy (compact ids in ascending original-id order)              Taxonomy.hpp:146-232, 1238-1257
// The `_space` bookkeeping fields of the reference's classes are written as 0: every loader ignores them.
#include "cfr_build.hpp"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <map>
#include <set>
#include <stdexcept>
#include <thread>

#include "cfr_index.hpp"     // IoError

namespace cfr {
namespace {

const char kAcgt[5] = "ACGT";

// ---- bit vectors ------------------------------------------------------------------------------------------------------
struct Bits {                      // append-only bit string, little endian inside u64 words
  std::vector<uint64_t> w;
  uint64_t n = 0;
  void push(uint32_t bit) {
    if ((n & 63) == 0) w.push_back(0);
    w.back() |= (uint64_t)(bit & 1u) << (n & 63);
    ++n;
  }
  void append(const Bits &o) {     // shift-copy of whole words
    if (o.n == 0) return;
    const uint32_t sh = (uint32_t)(n & 63);
    if (sh == 0) w.insert(w.end(), o.w.begin(), o.w.end());
    else {
      w.reserve(w.size() + o.w.size() + 1);
      for (size_t k = 0; k < o.w.size(); ++k) {
        w.back() |= o.w[k] << sh;
        w.push_back(o.w[k] >> (64 - sh));
      }
    }
    n += o.n;
    w.resize((n + 63) / 64);
  }
};

// DS_Rank9::Init (DS_Rank.hpp:206-247): per 8 words an absolute count and seven 9-bit relative counts
std::vector<uint64_t> rank9(const std::vector<uint64_t> &words) {
  const size_t nw = words.size(), nblk = (nw + 7) / 8;
  std::vector<uint64_t> out(nblk * 2, 0);
  uint64_t abs = 0;
  for (size_t b = 0; b < nblk; ++b) {
    out[2 * b] = abs;
    const size_t have = std::min<size_t>(8, nw - b * 8);
    uint64_t within = 0, rel = 0, tot = 0;
    for (size_t k = 0; k < have; ++k) tot += (uint64_t)__builtin_popcountll(words[b * 8 + k]);
    for (size_t br = 1; br < 8; ++br) {                      // entry br-1 = ones in words 0 .. br-1 of the block
      if (br - 1 < have) within += (uint64_t)__builtin_popcountll(words[b * 8 + br - 1]);
      uint64_t v;
      if (have == 8 || br < have) v = within;
      else v = have >= 2 ? tot : 0;                           // entries of words that do not exist: the block total, if the block has >= 2 words
      rel |= v << (9 * (br - 1));
    }
    out[2 * b + 1] = rel;
    abs += tot;
  }
  return out;
}

struct Out {
  FILE *f;
  std::string path;
  explicit Out(const std::string &p) : f(fopen(p.c_str(), "wb")), path(p) { if (!f) throw IoError{"cannot write " + p}; }
  ~Out() { if (f) fclose(f); }
  void raw(const void *p, size_t bytes) { if (bytes && fwrite(p, 1, bytes, f) != bytes) throw IoError{"short write to " + path}; }
  void u64(uint64_t v) { raw(&v, 8); }
  void i32(int32_t v) { raw(&v, 4); }
  void i16(int16_t v) { raw(&v, 2); }
  void close() { if (f && fclose(f) != 0) { f = nullptr; throw IoError{"cannot close " + path}; } f = nullptr; }
};

void write_alphabet(Out &o, bool empty) {      // Alphabet::Save (Alphabet.hpp:194-205), plain coding of ACGT
  if (empty) { o.u64(0); o.i32(0); o.u64(0); return; }
  o.u64(4); o.i32(1); o.u64(4);
  o.raw(kAcgt, 4);
  int32_t code[256] = {0};
  int16_t clen[256] = {0};
  for (int i = 0; i < 4; ++i) { code[(unsigned char)kAcgt[i]] = i; clen[(unsigned char)kAcgt[i]] = 2; }
  o.raw(code, sizeof(code));
  o.raw(clen, sizeof(clen));
}
void write_bitvector(Out &o, const Bits &b) {  // Bitvector_Plain::Save (Bitvector_Plain.hpp:182-196), select speed 0
  o.u64(0); o.u64(b.n); o.i32(0); o.i32(0); o.i32(0); o.i32(3);
  if (b.n == 0) return;
  o.raw(b.w.data(), b.w.size() * 8);
  const std::vector<uint64_t> r = rank9(b.w);
  o.u64(0); o.u64(b.w.size()); o.raw(r.data(), r.size() * 8);
  o.u64(0); o.u64(b.n); o.i32(0);
}
struct Wavelet { Bits root, lo0, lo1; uint64_t n = 0; };     // node 0: high code bit; node 1 / 2: low bit of the symbols with high bit 0 / 1
void write_wavelet(Out &o, const Wavelet &t) {  // Sequence_WaveletTree::Save (Sequence_WaveletTree.hpp:303-310)
  if (t.n == 0) { o.u64(0); o.u64(0); write_alphabet(o, true); o.i32(0); o.i32(3); return; }
  o.u64(0); o.u64(t.n); write_alphabet(o, false); o.i32(3); o.i32(0);
  o.u64(0); o.i32(0); o.i32(1); o.i32(2); write_bitvector(o, t.root);
  o.u64(0); o.i32(1); o.i32(-1); o.i32(-1); write_bitvector(o, t.lo0);
  o.u64(1); o.i32(1); o.i32(-1); o.i32(-1); write_bitvector(o, t.lo1);
}

// ---- automatic run-block size (Sequence_RunBlock.hpp:26-177) -------------------------------------------------------------
uint64_t run_block_len(const uint8_t *S, uint64_t n, uint64_t s, uint64_t e, uint64_t b) {   // GetRunBlockLength (:26-49)
  e = std::min(e, n - 1);
  if (s > e) return 0;
  uint64_t total = 0;
  for (uint64_t st = s; st <= e; st += b) {
    const uint64_t end = std::min(st + b, n);
    bool run = true;
    for (uint64_t k = st + 1; k < end; ++k) if (S[k] != S[st]) { run = false; break; }
    if (run) total += end - st;
  }
  return total;
}
uint64_t estimate_space(const uint8_t *S, uint64_t n, uint64_t b, uint64_t abits, uint64_t per_symbol_extra = 0) {   // EstimateSpace (:51-81; Sequence_RunBlockOneTree.hpp:52-82 charges one more bit per kept symbol)
  const uint64_t infer_len = 1024, cases = 1024;
  uint64_t rbl = 0, m = 0;
  if (infer_len * cases >= n) { rbl = run_block_len(S, n, 0, n - 1, b); m = n; }
  else {
    const uint64_t step = (n + cases - 1) / cases;
    for (uint64_t i = 0; i < n; i += step) {
      const uint64_t e = std::min(i + infer_len - 1, n - 1);
      rbl += run_block_len(S, n, i, i + infer_len - 1, b);
      m += e - i + 1;
    }
  }
  const uint64_t rbc = (rbl + b - 1) / b;
  if (b > 1) return (m + b - 1) / b + (abits + per_symbol_extra) * (rbc + m - rbl);
  return abits * m;
}
double avg_run_length(const uint8_t *S, uint64_t n) {                                          // EstimateAverageRunLength (:84-132)
  const uint64_t infer_len = 1024, cases = 1024;
  uint64_t r = 0, m = 0;
  auto runs_of = [&](uint64_t lo, uint64_t hi) { uint64_t c = 1; for (uint64_t k = lo + 1; k <= hi; ++k) c += S[k] != S[k - 1]; return c; };
  if (infer_len * cases >= n) return (double)n / (double)runs_of(0, n - 1);
  const uint64_t step = (n + cases - 1) / cases;
  for (uint64_t i = 0; i < n; i += step) {
    const uint64_t e = std::min(i + infer_len - 1, n - 1);
    r += runs_of(i, e);
    m += e - i + 1;
  }
  return (double)m / (double)r;
}
uint64_t compute_block_size(const uint8_t *S, uint64_t n, uint64_t abits = 2, uint64_t extra = 0) {      // ComputeBlockSize (:135-177), sigma = 4; Sequence_RunBlockOneTree.hpp:136-177 with (5, 1)
  uint64_t best_space = 0, best = 0;
  for (uint64_t i = 1; i <= 1024; i *= 2) {
    const uint64_t sp = estimate_space(S, n, i, abits, extra);
    if (best_space == 0 || sp < best_space) { best_space = sp; best = i; }
  }
  if (best >= 2) {
    const uint64_t sp = estimate_space(S, n, best / 2 * 3, abits, extra);
    if (sp < best_space) { best_space = sp; best = best / 2 * 3; }
  }
  const double x = std::sqrt(avg_run_length(S, n));
  const uint64_t test = (double)(uint64_t)x == x ? (uint64_t)x : (uint64_t)x + 1;
  if (test > 2) {
    const uint64_t sp = estimate_space(S, n, test, abits, extra);
    if (sp < best_space) { best_space = sp; best = test; }
  }
  return best;
}

// FixedSizeElemArray layout: element i at bits [i*l, (i+1)*l), LSB first
std::vector<uint64_t> pack_fixed(const std::vector<uint32_t> &vals, int bits) {
  const uint64_t nw = (vals.size() * (uint64_t)bits + 63) / 64;
  std::vector<uint64_t> out(nw + 1, 0);
  for (uint64_t i = 0; i < vals.size(); ++i) {
    const uint64_t pos = i * (uint64_t)bits, wi = pos >> 6, sh = pos & 63;
    out[wi] |= (uint64_t)vals[i] << sh;
    if (sh + (uint64_t)bits > 64) out[wi + 1] |= (uint64_t)vals[i] >> (64 - sh);
  }
  out.resize(nw);
  return out;
}

const char *kRanks[] = {"no rank", "strain", "species", "genus", "family", "order", "class", "phylum", "kingdom", "domain", "forma",
                        "infraclass", "infraorder", "parvorder", "subclass", "subfamily", "subgenus", "subkingdom", "suborder",
                        "subphylum", "subspecies", "subtribe", "superclass", "superfamily", "superkingdom", "superorder", "superphylum",
                        "tribe", "varietas", "life", "acellular root"};

// Taxonomy::Init + Save (Taxonomy.hpp:146-232, 1238-1257)
void write_taxonomy(const std::string &path, const BuildInput &in) {
  std::map<uint64_t, std::pair<uint64_t, int>> tree;            // tax id -> (parent, rank code); first mention wins
  for (const auto &nd : in.nodes) {
    if (tree.count(nd.taxid)) continue;
    int rk = 0;
    for (size_t k = 0; k < sizeof(kRanks) / sizeof(kRanks[0]); ++k) if (nd.rank == kRanks[k]) { rk = (int)k; break; }
    tree[nd.taxid] = {nd.parent, rk};
  }
  std::set<uint64_t> selected;                                   // the ids on the paths from the sequences' ids to the root
  std::set<uint64_t> present(in.taxids.begin(), in.taxids.end());
  present.insert(in.present_taxids.begin(), in.present_taxids.end());
  for (uint64_t tid : present) {
    uint64_t p = tid;
    if (!tree.count(p)) continue;
    while (!selected.count(p)) {
      selected.insert(p);
      auto it = tree.find(p);
      if (it == tree.end()) break;
      p = it->second.first;
      if (!tree.count(p)) break;
    }
  }
  std::vector<uint64_t> order(selected.begin(), selected.end());  // compact ids follow ascending original id (std::map order)
  std::map<uint64_t, uint64_t> cid;
  for (size_t i = 0; i < order.size(); ++i) cid[order[i]] = i;
  std::vector<uint8_t> leaf(order.size(), 1);
  std::vector<uint64_t> parent_c(order.size());
  for (size_t i = 0; i < order.size(); ++i) {
    const uint64_t par = tree[order[i]].first;
    auto it = cid.find(par);
    if (it != cid.end()) { parent_c[i] = it->second; leaf[it->second] = 0; }
    else parent_c[i] = i;
  }
  // a root is its own parent, which marked it a non-leaf above exactly like the reference's loop does
  std::map<uint64_t, std::string> sci;
  for (const auto &nm : in.tax_names) {
    if (!cid.count(nm.first)) continue;
    std::string s;                                               // "_".join(name.split())
    bool in_word = false;
    for (char ch : nm.second) {
      const bool sp = ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r' || ch == '\f' || ch == '\v';
      if (sp) { in_word = false; continue; }
      if (!in_word && !s.empty()) s += '_';
      s += ch;
      in_word = true;
    }
    sci[nm.first] = s;
  }
  Out o(path);
  o.u64(order.size()); o.u64(in.taxids.size()); o.u64(in.n_extra);          // _nodeCnt, _seqCnt, _extraSeqCnt
  for (size_t i = 0; i < order.size(); ++i) {
    o.u64(parent_c[i]);
    const uint8_t tail[8] = {(uint8_t)tree[order[i]].second, leaf[i], 0, 0, 0, 0, 0, 0};
    o.raw(tail, 8);
  }
  o.u64(order.size());
  for (uint64_t tid : order) o.u64(tid);
  for (uint64_t tid : order) { const std::string &s = sci[tid]; o.u64(s.size()); o.raw(s.data(), s.size()); }
  for (uint64_t tid : in.taxids) {
    // a tax id outside the tree: the reference warns and its MapID::Map (std::map operator[]) hands out compact id 0
    auto it = cid.find(tid);
    if (it == cid.end()) { fprintf(stderr, "WARNING: %llu is not in the taxonomy tree\n", (unsigned long long)tid); o.u64(0); }
    else o.u64(it->second);
  }
  for (const std::string &nm : in.names) { o.u64(nm.size()); o.raw(nm.data(), nm.size()); }
  o.close();
}

}  // namespace

// ================================================================================================ protein indexes
// centrifuger-build --protein (CentrifugerBuild.cpp:221-227): FMIndex<Sequence_RunBlockOneTree> over the alphabet
// "$ARNDCEQGHILKMFPSTWYV" with '$' behind every sequence (SequenceCompactor.hpp:59-87, Builder.hpp:95-101), no fuzzy boundary
// and no selectedSA (Builder.hpp:224-235), endMarkerSA = the sequence that FOLLOWS each '$' (Builder.hpp:55-67).
namespace {

const char kProtList[] = "$ARNDCEQGHILKMFPSTWYV";
constexpr uint32_t kProtSigma = 21, kProtBits = 5;

void write_alphabet_list(Out &o, const char *list, uint32_t sigma, uint32_t bits) {      // Alphabet::InitFromList + Save (Alphabet.hpp:53-69, 194-205)
  o.u64(sigma); o.i32(1); o.u64(sigma);
  o.raw(list, sigma);
  int32_t code[256] = {0};
  int16_t clen[256] = {0};
  for (uint32_t i = 0; i < sigma; ++i) { code[(unsigned char)list[i]] = (int32_t)i; clen[(unsigned char)list[i]] = (int16_t)bits; }
  o.raw(code, sizeof(code));
  o.raw(clen, sizeof(clen));
}

// Sequence_WaveletTree::BuildTree (Sequence_WaveletTree.hpp:104-133) for plain codes of `bits` bits: nodes numbered in preorder,
// a node's bit string holds bit (bits - 1 - depth) of its symbols; a node at the last bit or with no symbols is a leaf
struct GenNode { uint64_t prefix; int32_t depth, child0, child1; Bits v; };
int gen_tree(std::vector<GenNode> &nodes, const std::vector<uint8_t> &S, int depth, uint64_t prefix, uint32_t bits) {
  const int ti = (int)nodes.size();
  nodes.emplace_back();
  nodes[(size_t)ti].prefix = prefix; nodes[(size_t)ti].depth = depth; nodes[(size_t)ti].child0 = nodes[(size_t)ti].child1 = -1;
  {
    Bits &v = nodes[(size_t)ti].v;
    v.w.assign((S.size() + 63) / 64, 0);
    v.n = S.size();
    const uint32_t sh = bits - 1u - (uint32_t)depth;
    for (size_t i = 0; i < S.size(); ++i) v.w[i >> 6] |= (uint64_t)((S[i] >> sh) & 1u) << (i & 63);
  }
  if ((int)bits - depth == 1 || S.empty()) return ti;
  std::vector<uint8_t> left, right;
  {
    const uint32_t sh = bits - 1u - (uint32_t)depth;
    size_t ones = 0;
    for (uint8_t c : S) ones += (c >> sh) & 1u;
    left.reserve(S.size() - ones); right.reserve(ones);
    for (uint8_t c : S) ((c >> sh) & 1u ? right : left).push_back(c);
  }
  const int c0 = gen_tree(nodes, left, depth + 1, prefix << 1, bits);
  std::vector<uint8_t>().swap(left);
  const int c1 = gen_tree(nodes, right, depth + 1, (prefix << 1) | 1ull, bits);
  nodes[(size_t)ti].child0 = c0; nodes[(size_t)ti].child1 = c1;
  return ti;
}

std::vector<uint64_t> pack_fixed64(const std::vector<uint64_t> &vals, int bits) {
  const uint64_t nw = (vals.size() * (uint64_t)bits + 63) / 64;
  std::vector<uint64_t> out(nw + 1, 0);
  for (uint64_t i = 0; i < vals.size(); ++i) {
    const uint64_t pos = i * (uint64_t)bits, wi = pos >> 6, sh = pos & 63;
    out[wi] |= vals[i] << sh;
    if (sh + (uint64_t)bits > 64) out[wi + 1] |= vals[i] >> (64 - sh);
  }
  out.resize(nw);
  return out;
}
void write_fixed_array(Out &o, const std::vector<uint64_t> &vals) {       // FixedSizeElemArray::InitFromArray(0, ...) + Save (FixedSizeElemArray.hpp:72-90)
  uint64_t mx = 0;
  for (uint64_t v : vals) mx = std::max(mx, v);
  int bits = 1;
  while (bits < 64 && (mx >> bits)) ++bits;
  const std::vector<uint64_t> words = pack_fixed64(vals, bits);
  o.u64(words.size()); o.i32(bits); o.u64(vals.size()); o.raw(words.data(), words.size() * 8);
}

void build_protein_index_files(const BuildInput &in, const BuildOptions &opt, const std::string &prefix, BuildReport *rep) {
  const auto t0 = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  auto say = [&](const std::string &m) { if (opt.verbose) fprintf(stderr, "[cfr-build] %s\n", m.c_str()); };
  const uint32_t w = (uint32_t)opt.ftab_chars, rate = 1u << opt.offrate;
  if (opt.ftab_chars < 1 || opt.ftab_chars > 6) throw std::runtime_error("index build: --ftabchars of a protein index must be in 1..6 (5 bits per character)");
  if (opt.offrate < 0 || opt.offrate > 16) throw std::runtime_error("index build: --offrate must be in 0..16");
  const size_t G = in.lens.size();
  if (G == 0 || in.genome_seq.size() != G) throw std::invalid_argument("index build: the text needs at least one sequence, with one sequence id and one length each");
  if (in.names.size() != in.taxids.size() + in.n_extra) throw std::invalid_argument("index build: one tax id per conversion-table sequence, none for the extra names");
  std::vector<uint64_t> psum(G + 1, 0);                 // with the '$' of every sequence
  {
    std::vector<uint8_t> seen(in.names.size(), 0);
    for (size_t g = 0; g < G; ++g) {
      if (in.genome_seq[g] >= in.names.size()) throw std::invalid_argument("index build: a sequence id is outside the name list");
      if (seen[in.genome_seq[g]]) throw std::invalid_argument("index build: sequence " + in.names[in.genome_seq[g]] + " appears twice in the text");
      seen[in.genome_seq[g]] = 1;
      if (in.lens[g] + 1 < (uint64_t)w + 1) throw std::invalid_argument("index build: sequence " + in.names[in.genome_seq[g]] + " is shorter than --ftabchars");
      psum[g + 1] = psum[g] + in.lens[g] + 1;
    }
  }
  const uint64_t n = psum[G];
  if (n >= 0xfffffff0ull) throw std::invalid_argument("index build: a protein text of 2^32 symbols or more is outside this writer");
  // ---- the text as plain codes
  std::vector<uint8_t> T(n);
  {
    uint8_t code_of[256];
    memset(code_of, 255, sizeof(code_of));
    for (uint32_t k = 1; k < kProtSigma; ++k) code_of[(unsigned char)kProtList[k]] = (uint8_t)k;
    uint64_t src = 0;
    for (size_t g = 0; g < G; ++g) {
      uint8_t *dst = T.data() + psum[g];
      for (uint64_t k = 0; k < in.lens[g]; ++k) {
        const uint8_t c = code_of[in.text[src + k]];
        if (c == 255) throw std::runtime_error("index build: the text of a protein index must consist of the letters ARNDCEQGHILKMFPSTWYV only");
        dst[k] = c;
      }
      dst[in.lens[g]] = 0;
      src += in.lens[g];
    }
  }
  // ---- suffix array (device) and what is read off it
  std::vector<uint32_t> sa;
  double sa_seconds = 0;
  int rounds = 0;
  build_sa_bytes(T.data(), n, opt.device, sa, &sa_seconds, &rounds);
  say("suffix array: " + std::to_string(sa_seconds) + " s, " + std::to_string(rounds) + " rounds");
  auto seq_of = [&](uint64_t pos) -> uint64_t {          // PartialSum::Search: the sequence that holds text position pos
    size_t k = (size_t)(std::upper_bound(psum.begin(), psum.end(), pos) - psum.begin()) - 1;
    if (k >= G) k = G - 1;
    return in.genome_seq[k];
  };
  std::vector<uint8_t> B(n);
  uint64_t first_isa = 0;
  const uint64_t end_markers = G;                        // one '$' per sequence (the letters themselves never code to 0)
  const uint64_t nsamp = (n + rate - 1) / rate, nk = 1ull << (kProtBits * w);
  std::vector<uint64_t> sampled(nsamp), ftab(2 * nk, 0), end_sa(end_markers);
  {
    // Row ranges in parallel, ONE table (the reference allocates it once too: 2 x 32^w entries are 16 GiB at w = 6).  The rows of a key
    // - the suffixes that begin with its w symbols - stand together in the suffix array, so a key whose rows lie inside a thread's range
    // is that thread's alone and goes straight into the table; only the run a range begins in and the one it ends in can be shared with
    // a neighbour, and those two (key, first row, count) triples per thread are added afterwards in range order - "first" is then the
    // smallest row of a key exactly as in the reference's single pass.  (Suffixes shorter than w have no key and break no run.)
    int threads = opt.threads > 0 ? (int)std::min<uint64_t>((uint64_t)opt.threads, n)
                                  : (int)std::min<uint64_t>(std::min(32u, std::max(1u, std::thread::hardware_concurrency())), std::max<uint64_t>(1, n >> 20));
    struct Run { uint64_t key = ~0ull, first = 0, count = 0; };
    std::vector<Run> head((size_t)threads), tail((size_t)threads);
    std::vector<uint64_t> fi((size_t)threads, ~0ull);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back([&, t]() {
      const uint64_t lo = n * (uint64_t)t / (uint64_t)threads, hi = n * (uint64_t)(t + 1) / (uint64_t)threads;
      Run cur;
      bool first_run = true;
      auto flush = [&](bool last) {
        if (cur.key == ~0ull) return;
        if (first_run) head[(size_t)t] = cur;
        else if (last) tail[(size_t)t] = cur;
        else { ftab[2 * cur.key] = cur.first; ftab[2 * cur.key + 1] = cur.count; }
        first_run = false;
      };
      for (uint64_t i = lo; i < hi; ++i) {
        const uint64_t p = sa[i];
        if (p == 0) { fi[(size_t)t] = i; B[i] = T[n - 1]; } else B[i] = T[p - 1];
        if (i % rate == 0) sampled[i / rate] = seq_of(p);                      // (no fuzzy boundary, Builder.hpp:55-60)
        if (p + w <= n) {                                                       // FMBuilder.hpp:256-283: T.PackRead(p, w) - the first symbol in the low bits
          uint64_t key = 0;
          for (uint32_t k = 0; k < w; ++k) key |= (uint64_t)T[p + k] << (kProtBits * k);
          if (key != cur.key) { flush(false); cur.key = key; cur.first = i; cur.count = 0; }
          ++cur.count;
        }
        if (i < end_markers) end_sa[i] = seq_of(p + 1);                         // rows of the '$' suffixes come first (FMBuilder.hpp:306-311)
      }
      flush(true);
    });
    for (auto &x : th) x.join();
    for (int t = 0; t < threads; ++t) {
      if (fi[(size_t)t] != ~0ull) first_isa = fi[(size_t)t];
      for (const Run *r : {&head[(size_t)t], &tail[(size_t)t]}) if (r->key != ~0ull) {
        if (ftab[2 * r->key + 1] == 0) ftab[2 * r->key] = r->first;
        ftab[2 * r->key + 1] += r->count;
      }
    }
  }
  std::vector<uint32_t>().swap(sa);
  std::vector<uint64_t> C(kProtSigma + 1, 0);
  for (uint64_t i = 0; i < n; ++i) ++C[B[i] + 1u];
  for (uint32_t k = 1; k <= kProtSigma; ++k) C[k] += C[k - 1];
  const char last_chr = kProtList[B[first_isa]];
  std::vector<uint8_t>().swap(T);

  // ---- Sequence_RunBlockOneTree::Init (Sequence_RunBlockOneTree.hpp:227-370)
  uint64_t b = opt.rbbwt_b ? opt.rbbwt_b : compute_block_size(B.data(), n, kProtBits, 1);
  if (b == 1) b = n;
  const uint64_t nblk = (n + b - 1) / b;
  Bits use;
  std::vector<Bits> alpha_rb(kProtSigma);
  std::vector<uint8_t> mixed;
  mixed.reserve(n);
  {
    bool any_run = false;
    use.w.assign((nblk + 63) / 64, 0);
    use.n = nblk;
    for (uint64_t k = 0; k < nblk; ++k) {
      const uint64_t st = k * b, end = std::min(st + b, n);
      bool run = true;
      for (uint64_t q = st + 1; q < end; ++q) if (B[q] != B[st]) { run = false; break; }
      if (run) { use.w[k >> 6] |= 1ull << (k & 63); any_run = true; }
    }
    // b == n without a run: the per-symbol strings stay empty (:276-292, "let rank9 handle the special case")
    const bool keep_rb = b != n || any_run;
    for (uint64_t k = 0; k < nblk; ++k) {
      const uint64_t st = k * b, end = std::min(st + b, n);
      if ((use.w[k >> 6] >> (k & 63)) & 1ull) { mixed.push_back(B[st]); if (keep_rb) alpha_rb[B[st]].push(1); }
      else for (uint64_t q = st; q < end; ++q) { mixed.push_back(B[q]); if (keep_rb) alpha_rb[B[q]].push(0); }
    }
  }
  std::vector<uint8_t>().swap(B);
  std::vector<GenNode> nodes;
  nodes.reserve(32);
  gen_tree(nodes, mixed, 0, 0, kProtBits);
  const uint64_t mixed_n = mixed.size();
  std::vector<uint8_t>().swap(mixed);
  say("run-block (one tree): b = " + std::to_string(b) + ", " + std::to_string(nblk) + " blocks, " + std::to_string(mixed_n) + " symbols kept, " +
      std::to_string(nodes.size()) + " wavelet nodes (" + std::to_string(since()) + " s)");

  // ---- .1.cfr (FMIndex::Save, FMIndex.hpp:571-586; Sequence_RunBlockOneTree::Save :485-497)
  {
    Out o(prefix + ".1.cfr");
    o.u64(n); o.u64(kProtBits); o.u64(first_isa);
    o.raw(&last_chr, 1);
    o.u64(0); o.u64(n); write_alphabet_list(o, kProtList, kProtSigma, kProtBits);
    o.u64(b); o.u64(nblk);
    write_bitvector(o, use);
    for (uint32_t k = 0; k < kProtSigma; ++k) write_bitvector(o, alpha_rb[k]);
    o.u64(0); o.u64(mixed_n); write_alphabet_list(o, kProtList, kProtSigma, kProtBits);
    o.i32((int32_t)nodes.size()); o.i32(0);
    for (const GenNode &nd : nodes) { o.u64(nd.prefix); o.i32(nd.depth); o.i32(nd.child0); o.i32(nd.child1); write_bitvector(o, nd.v); }
    write_alphabet_list(o, kProtList, kProtSigma, kProtBits); write_alphabet_list(o, kProtList, kProtSigma, kProtBits);
    o.raw(C.data(), C.size() * 8);
    o.u64(n); o.i32(0); o.i32((int32_t)rate); o.u64(nsamp); o.u64(w); o.u64(nk); o.u64(0);       // adjustedSA0 stays 0 with end markers (FMBuilder.hpp:68)
    write_fixed_array(o, sampled);
    o.raw(ftab.data(), ftab.size() * 8);
    o.u64(0);                                                    // maxLcp
    o.u64(0); o.i32(1024);                                       // no selectedSA
    const char one = 1;
    o.raw(&one, 1);                                              // hasEndMarker
    write_fixed_array(o, end_sa);
    o.close();
  }
  write_taxonomy(prefix + ".2.cfr", in);
  {
    Out o(prefix + ".3.cfr");
    std::map<uint64_t, uint64_t> by_id;                        // lengths count the '$' (Builder.hpp:143-156: what Compact returned)
    for (size_t g = 0; g < G; ++g) by_id[in.genome_seq[g]] = in.lens[g] + 1;
    for (const auto &kv : by_id) { o.u64(kv.first); o.u64(kv.second); }
    o.close();
  }
  {
    Out o(prefix + ".4.cfr");
    char stime[128];
    const time_t now = time(nullptr);
    strftime(stime, sizeof(stime), "%c", localtime(&now));
    const std::string txt = "version\t1.1.3-r347\nSA_sample_rate\t" + std::to_string(rate) + "\nsequence_type\tamino_acid\nbuild_date\t" + stime;
    o.raw(txt.data(), txt.size());
    o.close();
  }
  say("protein index written to " + prefix + ".*.cfr in " + std::to_string(since()) + " s");
  if (rep) { rep->n = n; rep->block_size = b; rep->first_isa = first_isa; rep->seconds_sa = sa_seconds; rep->seconds_total = since(); rep->rounds = rounds; }
}

}  // namespace

void build_index_files(const BuildInput &in, const BuildOptions &opt, const std::string &prefix, BuildReport *rep) {
  if (opt.protein) {
    // the reference's rule (CentrifugerBuild.cpp:221-227): a protein index built with the nucleotide default of 10 initial-lookup
    // characters gets 4 - for every caller of the library, not only for the command line
    BuildOptions po = opt;
    if (po.ftab_chars == 10) po.ftab_chars = 4;
    build_protein_index_files(in, po, prefix, rep);
    return;
  }
  const auto t0 = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  auto say = [&](const std::string &m) { if (opt.verbose) fprintf(stderr, "[cfr-build] %s\n", m.c_str()); };
  const uint32_t w = (uint32_t)opt.ftab_chars, rate = 1u << opt.offrate;
  if (opt.ftab_chars < 1 || opt.ftab_chars > 16) throw std::runtime_error("index build: --ftabchars must be in 1..16");
  if (opt.offrate < 0 || opt.offrate > 16) throw std::runtime_error("index build: --offrate must be in 0..16");
  const size_t G = in.lens.size();
  if (G == 0 || in.genome_seq.size() != G) throw std::invalid_argument("index build: the text needs at least one genome, with one sequence id and one length each");
  if (in.names.size() != in.taxids.size() + in.n_extra) throw std::invalid_argument("index build: one tax id per conversion-table sequence, none for the extra names");
  {
    std::vector<uint8_t> seen(in.names.size(), 0);
    for (size_t g = 0; g < G; ++g) {
      if (in.genome_seq[g] >= in.names.size()) throw std::invalid_argument("index build: a genome's sequence id is outside the name list");
      if (seen[in.genome_seq[g]]) throw std::invalid_argument("index build: sequence " + in.names[in.genome_seq[g]] + " appears twice in the text");
      seen[in.genome_seq[g]] = 1;
      // (Builder.hpp:143-150 filters such a genome with a warning; here the caller does the filtering - the command line does)
      if (in.lens[g] < (uint64_t)w + 1) throw std::invalid_argument("index build: genome " + in.names[in.genome_seq[g]] + " is shorter than --ftabchars + 1");
    }
  }
  std::vector<uint64_t> psum(G + 1, 0);
  for (size_t g = 0; g < G; ++g) psum[g + 1] = psum[g] + in.lens[g];
  const uint64_t n = psum[G];
  if (n < 64) throw std::invalid_argument("index build: the text must hold at least 64 symbols");
  int threads = opt.threads > 0 ? opt.threads : (int)std::min(64u, std::max(1u, std::thread::hardware_concurrency()));

  // (the text must be upper-case ACGT only - SequenceCompactor.hpp:59-84 drops everything else before the text is formed, so does
  // the command line; the device checks it while packing the text, no host copy is made)
  const char *lp = strchr(kAcgt, (char)in.text[n - 1]);
  if (!lp || !in.text[n - 1]) throw std::runtime_error("index build: the text must be upper-case ACGT only");
  const uint8_t last_code = (uint8_t)(lp - kAcgt);

  // ---- suffix array and what is read off it (device)
  std::vector<uint64_t> want;              // selectedSA: the row of text position psum[g+1] - w - 1 for every genome boundary (Builder.hpp:224-234)
  for (size_t g = 0; g + 1 < G; ++g) if (psum[g + 1] >= (uint64_t)w + 1) want.push_back(psum[g + 1] - w - 1);
  SaProducts sa;
  build_sa_products(in.text, n, opt.device, rate, w, psum, want, sa, say);
  say("suffix array + products: " + std::to_string(sa.seconds_sa) + " + " + std::to_string(sa.seconds_products) + " s");
  std::map<uint64_t, uint64_t> sel;
  for (size_t k = 0; k < want.size(); ++k) {
    const uint64_t pos = want[k] + w + 1;
    const uint64_t id = (uint64_t)(std::upper_bound(psum.begin(), psum.end(), pos) - psum.begin()) - 1;
    sel[sa.rows_of[k]] = in.genome_seq[id];
  }
  // the device numbered the genomes in text order; the index stores sequence ids (Builder.hpp:27-51: genomeSeqIds[...])
  {
    bool identity = true;
    for (size_t g = 0; g < G; ++g) if (in.genome_seq[g] != g) { identity = false; break; }
    if (!identity) for (auto &x : sa.sampled_ids) x = (uint32_t)in.genome_seq[x];
  }
  const uint8_t *B = sa.bwt.data();

  // ---- C[]
  uint64_t C[5] = {0, 0, 0, 0, 0};
  {
    std::vector<std::array<uint64_t, 4>> part((size_t)threads, std::array<uint64_t, 4>{0, 0, 0, 0});
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back([&, t]() {
      const uint64_t lo = n * (uint64_t)t / (uint64_t)threads, hi = n * (uint64_t)(t + 1) / (uint64_t)threads;
      uint64_t c[4] = {0, 0, 0, 0};
      for (uint64_t i = lo; i < hi; ++i) ++c[B[i]];
      for (int k = 0; k < 4; ++k) part[(size_t)t][(size_t)k] = c[k];
    });
    for (auto &x : th) x.join();
    uint64_t cnt[4] = {0, 0, 0, 0};
    for (auto &p : part) for (int k = 0; k < 4; ++k) cnt[k] += p[(size_t)k];
    for (int k = 0; k < 4; ++k) C[k + 1] = C[k] + cnt[k];
  }

  // ---- run blocks (Sequence_RunBlock::Init, Sequence_RunBlock.hpp:231-358)
  uint64_t b = opt.rbbwt_b ? opt.rbbwt_b : compute_block_size(B, n);
  if (b == 1) b = n;
  const uint64_t nblk = (n + b - 1) / b;
  struct Part { Bits use; Wavelet plain, runs; };
  // a part owns a contiguous range of blocks; parts are concatenated afterwards (bit strings: shift-copy of words)
  const int parts_n = (int)std::min<uint64_t>((uint64_t)threads, std::max<uint64_t>(1, nblk / 4096));
  std::vector<Part> parts((size_t)parts_n);
  {
    std::vector<std::thread> th;
    for (int t = 0; t < parts_n; ++t) th.emplace_back([&, t]() {
      Part &P = parts[(size_t)t];
      const uint64_t blo = nblk * (uint64_t)t / (uint64_t)parts_n, bhi = nblk * (uint64_t)(t + 1) / (uint64_t)parts_n;
      auto push_sym = [](Wavelet &W, uint32_t s) { W.root.push(s >> 1); if (s >> 1) W.lo1.push(s & 1u); else W.lo0.push(s & 1u); ++W.n; };
      for (uint64_t k = blo; k < bhi; ++k) {
        const uint64_t st = k * b, end = std::min(st + b, n);          // the last block is judged on its real symbols only
        bool run = true;
        for (uint64_t q = st + 1; q < end; ++q) if (B[q] != B[st]) { run = false; break; }
        P.use.push(run ? 1u : 0u);
        if (run) push_sym(P.runs, B[st]);
        else for (uint64_t q = st; q < end; ++q) push_sym(P.plain, B[q]);
      }
    });
    for (auto &x : th) x.join();
  }
  Bits use;
  Wavelet plain, runs;
  {
    // seven independent concatenations
    auto cat = [&](Bits &dst, const std::function<const Bits &(const Part &)> &pick) { for (const Part &P : parts) dst.append(pick(P)); };
    std::vector<std::thread> th;
    th.emplace_back([&]() { cat(use, [](const Part &P) -> const Bits & { return P.use; }); });
    th.emplace_back([&]() { cat(plain.root, [](const Part &P) -> const Bits & { return P.plain.root; }); });
    th.emplace_back([&]() { cat(plain.lo0, [](const Part &P) -> const Bits & { return P.plain.lo0; }); });
    th.emplace_back([&]() { cat(plain.lo1, [](const Part &P) -> const Bits & { return P.plain.lo1; }); });
    th.emplace_back([&]() { cat(runs.root, [](const Part &P) -> const Bits & { return P.runs.root; }); });
    th.emplace_back([&]() { cat(runs.lo0, [](const Part &P) -> const Bits & { return P.runs.lo0; }); });
    th.emplace_back([&]() { cat(runs.lo1, [](const Part &P) -> const Bits & { return P.runs.lo1; }); });
    for (auto &x : th) x.join();
    for (const Part &P : parts) { plain.n += P.plain.n; runs.n += P.runs.n; }
    std::vector<Part>().swap(parts);
  }
  say("run-block: b = " + std::to_string(b) + ", " + std::to_string(runs.n) + " run blocks of " + std::to_string(nblk) + "; wavelet part " +
      std::to_string(plain.n) + " symbols (" + std::to_string(since()) + " s)");

  // ---- .1.cfr (FMIndex::Save, FMIndex.hpp:571-586)
  {
    Out o(prefix + ".1.cfr");
    o.u64(n); o.u64(2); o.u64(sa.first_isa);
    const char last_chr = kAcgt[last_code];
    o.raw(&last_chr, 1);
    o.u64(0); o.u64(n); write_alphabet(o, false);
    o.u64(b); o.u64(nblk);
    write_bitvector(o, use);
    write_wavelet(o, plain);
    write_wavelet(o, runs);
    write_alphabet(o, false); write_alphabet(o, false);
    o.raw(C, sizeof(C));
    const uint64_t nsamp = (n + rate - 1) / rate, nk = 1ull << (2 * w);
    o.u64(n); o.i32(0); o.i32((int32_t)rate); o.u64(nsamp); o.u64(w); o.u64(nk); o.u64(in.genome_seq[0]);   // n, strategy, rate, sample size, ftab width, ftab size, adjustedSA0 = genomeSeqIds[0]
    uint64_t mx = 0;
    for (uint64_t v : sa.sampled_ids) mx = std::max(mx, v);
    int bits = 1;
    while (bits < 64 && (mx >> bits)) ++bits;
    const std::vector<uint64_t> words = pack_fixed(sa.sampled_ids, bits);
    o.u64(words.size()); o.i32(bits); o.u64(sa.sampled_ids.size()); o.raw(words.data(), words.size() * 8);
    o.raw(sa.ftab.data(), sa.ftab.size() * 8);
    o.u64(0);                                                    // maxLcp
    o.u64(sel.size()); o.i32(1024);
    for (const auto &kv : sel) { o.u64(kv.first); o.u64(kv.second); }
    const char zero = 0;
    o.raw(&zero, 1);                                             // hasEndMarker = false
    o.close();
  }
  write_taxonomy(prefix + ".2.cfr", in);
  {
    Out o(prefix + ".3.cfr");
    std::map<uint64_t, uint64_t> by_id;                        // Builder.hpp:300-305: the std::map<seqId, length> in key order
    for (size_t g = 0; g < G; ++g) by_id[in.genome_seq[g]] = in.lens[g];
    for (const auto &kv : by_id) { o.u64(kv.first); o.u64(kv.second); }
    o.close();
  }
  {
    Out o(prefix + ".4.cfr");
    char stime[128];
    const time_t now = time(nullptr);
    strftime(stime, sizeof(stime), "%c", localtime(&now));
    const std::string txt = "version\t1.1.3-r347\nSA_sample_rate\t" + std::to_string(rate) + "\nsequence_type\tnucleotide\nbuild_date\t" + stime;
    o.raw(txt.data(), txt.size());
    o.close();
  }
  say("index written to " + prefix + ".*.cfr in " + std::to_string(since()) + " s");
  if (rep) {
    rep->n = n; rep->block_size = b; rep->first_isa = sa.first_isa; rep->seconds_sa = sa.seconds_sa; rep->seconds_total = since();
    rep->rounds = sa.rounds;
  }
}

}  // namespace cfr
