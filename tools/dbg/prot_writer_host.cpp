// tools/dbg/prot_writer_host.cpp — development check of the HOST half of the protein index writer (csrc/cfr_build.cpp) without a GPU:
// compiles cfr_build.cpp with g++, supplies a naive suffix sort in place of the device one (csrc/cfr_build_sa.hip), reads a FASTA
// of proteins + nodes.dmp / names.dmp / conversion table like centrifuger-build does and writes <prefix>.*.cfr, to be compared with
// what `oracle/_ref/centrifuger-build --protein` wrote for the same input (tools/dbg/prot_writer_check.sh).  Test infrastructure:
// never part of libcfr_hip.so.
//   g++ -O2 -std=c++17 -I centrifuger_amd/csrc -o /tmp/prot_writer_host tools/dbg/prot_writer_host.cpp centrifuger_amd/csrc/cfr_build.cpp -lpthread
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include "cfr_build.hpp"

namespace cfr {
void build_sa_products(const uint8_t *, uint64_t, int, uint32_t, uint32_t, const std::vector<uint64_t> &, const std::vector<uint64_t> &, SaProducts &,
                       const std::function<void(const std::string &)> &) { throw std::runtime_error("nucleotide texts need the device"); }
void build_sa_bytes(const uint8_t *codes, uint64_t n, int, std::vector<uint32_t> &sa, double *seconds, int *rounds) {
  sa.resize(n);
  for (uint64_t i = 0; i < n; ++i) sa[i] = (uint32_t)i;
  std::sort(sa.begin(), sa.end(), [&](uint32_t a, uint32_t b) {
    const uint64_t la = n - a, lb = n - b, m = std::min(la, lb);
    const int c = memcmp(codes + a, codes + b, m);
    if (c) return c < 0;
    return la < lb;                          // a proper prefix first
  });
  if (seconds) *seconds = 0;
  if (rounds) *rounds = 0;
}
}  // namespace cfr

int main(int argc, char **argv) {
  if (argc < 7) { fprintf(stderr, "usage: prot.fa nodes.dmp names.dmp conv.tsv out_prefix ftabchars [offrate] [rbbwt_b]\n"); return 2; }
  cfr::BuildInput in;
  std::map<std::string, size_t> idx;
  {
    std::ifstream f(argv[4]);
    std::string nm; unsigned long long t;
    while (f >> nm >> t) { idx[nm] = in.names.size(); in.names.push_back(nm); in.taxids.push_back(t); }
  }
  {
    std::ifstream f(argv[2]);
    std::string ln;
    while (std::getline(f, ln)) {
      unsigned long long a, b; char rk[256];
      if (sscanf(ln.c_str(), "%llu\t|\t%llu\t|\t%255[^\t]", &a, &b, rk) == 3) in.nodes.push_back(cfr::TaxNode{a, b, rk});
    }
    std::ifstream g(argv[3]);
    while (std::getline(g, ln)) {
      if (ln.find("scientific name") == std::string::npos) continue;
      unsigned long long a; char nm[1024];
      if (sscanf(ln.c_str(), "%llu\t|\t%1023[^\t]", &a, nm) == 2) in.tax_names.emplace_back(a, nm);
    }
  }
  static std::vector<uint8_t> text;
  {
    std::ifstream f(argv[1]);
    std::string ln, cur;
    size_t start = 0;
    bool have = false;
    auto close = [&]() { if (have) { in.genome_seq.push_back(idx.at(cur)); in.lens.push_back(text.size() - start); } };
    while (std::getline(f, ln)) {
      if (!ln.empty() && ln[0] == '>') { close(); cur = ln.substr(1, ln.find_first_of(" \t") - 1); start = text.size(); have = true; }
      else for (char c : ln) if (strchr("ARNDCEQGHILKMFPSTWYV", c) && c) text.push_back((uint8_t)c);
    }
    close();
  }
  in.text = text.data();
  cfr::BuildOptions opt;
  opt.protein = true; opt.verbose = true;
  opt.ftab_chars = atoi(argv[6]);
  if (argc > 7) opt.offrate = atoi(argv[7]);
  if (argc > 8) opt.rbbwt_b = strtoull(argv[8], nullptr, 10);
  if (argc > 9) opt.threads = atoi(argv[9]);
  cfr::BuildReport rep;
  cfr::build_index_files(in, opt, argv[5], &rep);
  return 0;
}
