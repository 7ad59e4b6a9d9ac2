// cfr_build.hpp — the index writer: what centrifuger-build's FMBuilder / Builder produce (FMBuilder.hpp:209-313,
// Builder.hpp:86-313), built on one MI355X.  Not on the classification path; it exists so that indexes of the sizes the
// path is measured on (several Gbp; n >= 2^32) can be manufactured on the GPU box in a minute instead of the reference
// builder's half hour, and so that the on-disk contract is exercised from the writing side.
//
//   cfr_build_sa.hip  : the suffix array in HBM (prefix doubling over 32-mer-sorted groups, radix sorts by hipCUB) and what
//                       is read off it: BWT, firstISA, sampled sequence ids, selected rows, ftab
//   cfr_build.cpp     : run-block compression, wavelet trees, rank9 counters, taxonomy, the four .cfr files; C-ABI entry
#pragma once

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace cfr {

struct SaProducts {
  uint64_t n = 0, first_isa = 0;
  std::vector<uint8_t> bwt;              // n codes 0..3: B[i] = T[SA[i]-1], T[n-1] at the row of position 0 (FMBuilder.hpp:244-250)
  std::vector<uint32_t> sampled_ids;     // sequence id of SA[k * rate] with the fuzzy boundary (Builder.hpp:27-51)
  std::vector<uint64_t> rows_of;         // rows of the requested text positions (for selectedSA, Builder.hpp:224-234)
  std::vector<uint64_t> ftab;            // 4^w pairs (first row, count) over suffixes of >= w characters (FMBuilder.hpp:256-283)
  double seconds_sa = 0, seconds_products = 0;
  int rounds = 0;
};

// text: n upper-case A,C,G,T (host; read once, front to back).  psum: G+1 sequence start offsets.  want_pos: text positions
// whose rows are wanted.  Throws HipError (cfr_device.hpp) on device failure / unsupported size / other characters.
void build_sa_products(const uint8_t *text, uint64_t n, int device, uint32_t sample_rate, uint32_t ftab_width,
                       const std::vector<uint64_t> &psum, const std::vector<uint64_t> &want_pos, SaProducts &out,
                       const std::function<void(const std::string &)> &log);

// Suffix array of a text of byte codes (protein indexes: 21 codes, '$' = 0 closes every sequence), n < 2^32: prefix doubling with
// hipCUB radix sorts in HBM (cfr_build_sa.hip).  Same order as above: plain lexicographic, a proper prefix first.
void build_sa_bytes(const uint8_t *codes, uint64_t n, int device, std::vector<uint32_t> &sa, double *seconds, int *rounds);

struct TaxNode { uint64_t taxid, parent; std::string rank; };
struct BuildInput {
  std::vector<std::string> names;          // sequence names, conversion-table order = sequence ids; the last n_extra are extra names (no tax id)
  std::vector<uint64_t> taxids;            // original tax id of every conversion-table sequence
  std::vector<uint64_t> present_taxids;    // further ids the conversion table mentions (their lineages stay in the tree)
  uint64_t n_extra = 0;
  std::vector<uint64_t> genome_seq, lens;  // the genomes of the text, text order: sequence id and length (ACGT only)
  const uint8_t *text = nullptr;           // the genomes back to back, upper-case ACGT (BuildOptions::protein: the proteins back to back, letters of
                                           // "ARNDCEQGHILKMFPSTWYV" only; lens are residue counts, the writer puts '$' behind every protein)
  std::vector<TaxNode> nodes;              // nodes.dmp
  std::vector<std::pair<uint64_t, std::string>> tax_names;   // names.dmp, scientific names
};
struct BuildOptions { int ftab_chars = 10, offrate = 4, device = 0, threads = 0; uint64_t rbbwt_b = 0; bool verbose = false, protein = false; };
struct BuildReport { uint64_t n = 0, block_size = 0, first_isa = 0; double seconds_sa = 0, seconds_total = 0; int rounds = 0; };

// Writes <prefix>.{1,2,3,4}.cfr.  Throws HipError / IoError / std::runtime_error.
void build_index_files(const BuildInput &in, const BuildOptions &opt, const std::string &prefix, BuildReport *report);

}  // namespace cfr
