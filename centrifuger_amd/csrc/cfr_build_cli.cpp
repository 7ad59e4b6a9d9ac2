// cfr_build_cli.cpp — `centrifuger-build`-compatible command line on top of cfr_build_index (include/cfr_hip.h).
//
// Same option names as the reference's builder (CentrifugerBuild.cpp:34-52) for what the MI355X writer covers:
// nucleotide references, one sequence per conversion-table line, default layout or --rbbwt-b / --offrate / --ftabchars.
// --bmax / --dcv / --build-mem steer the reference's blockwise sorter (FMBuilder.hpp:444-811) and have no meaning here
// (the suffix array is built in HBM): accepted and ignored.  --protein writes the amino-acid index (FMIndex<Sequence_RunBlockOneTree>,
// '$' behind every sequence).  Options of other parts of the builder (--subset-tax, --concat-tax-genome, --checkpoint, file-level conversion tables) are rejected with a message.
// Input handling follows Builder::Build (Builder.hpp:108-165) and Taxonomy::ReadSeqNameFile (Taxonomy.hpp:303-368): the text is
// formed in FASTA order; a conversion table may name sequences the FASTA does not hold (they keep their ids and tax ids in
// .2.cfr, have no length in .3.cfr); a FASTA sequence the table does not name is added as an extra name with a warning (or
// skipped under --ignore-uncategorized-genome); a repeated sequence id is taken once; a sequence shorter than --ftabchars + 1
// after dropping non-ACGT characters is filtered with a warning; a name listed twice in the table gets the lowest common
// ancestor of its tax ids.
#include <getopt.h>
#include <zlib.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <vector>

#include "../../include/cfr_hip.h"

namespace {

const char *kUsage =
    "./centrifuger-build [OPTIONS]:\n"
    "Required:\n"
    "\t-r FILE: reference sequence file (can use multiple -r to specify more than one input file)\n"
    "\t\tor\n"
    "\t-l FILE: list of reference sequence file stored in <file>, one sequence file per row\n"
    "\t--taxonomy-tree FILE: taxonomy tree, i.e., nodes.dmp file\n"
    "\t--name-table FILE: name table, i.e., names.dmp file\n"
    "\t--conversion-table FILE: a table that converts reference sequence id to taxonomy id\n"
    "Optional:\n"
    "\t-o STRING: output prefix [centrifuger]\n"
    "\t-t INT: number of host threads [automatic]\n"
    "\t--offrate INT: SA/offset is sampled every (2^<int>) BWT chars [4]\n"
    "\t--ftabchars INT: # of chars consumed in initial lookup [10]\n"
    "\t--rbbwt-b INT: block size for run-block compressed BWT. 0 for auto. 1 for no compression [0]\n"
    "\t--protein: the input is protein sequences [off]\n"
    "\t--ignore-uncategorized-genome: do not index a sequence that is missing from the conversion table\n"
    "\t--gpu INT: MI355X ordinal that builds the suffix array [0]\n"
    "\t--bmax / --dcv / --build-mem: accepted and ignored (they steer the reference's blockwise sorter)\n"
    "\t-h: print this usage message\n";

enum { O_BMAX = 1000, O_DCV, O_MEM, O_OFFRATE, O_FTAB, O_RBB, O_TREE, O_CONV, O_NAMES, O_GPU, O_IGNORE_UNCAT, O_PROTEIN, O_UNSUPPORTED };

void print_log(const char *fmt, ...) {
  char buffer[1024];
  va_list args;
  va_start(args, fmt);
  vsnprintf(buffer, sizeof(buffer), fmt, args);
  va_end(args);
  time_t now = time(nullptr);
  char stime[128];
  strftime(stime, sizeof(stime), "%c", localtime(&now));
  fprintf(stderr, "[%s] %s\n", stime, buffer);
}

std::vector<std::string> split_dmp(const std::string &line) {     // "a\t|\tb\t|\t..." -> fields, blanks trimmed
  std::vector<std::string> f;
  size_t at = 0;
  for (;;) {
    const size_t bar = line.find('|', at);
    std::string x = line.substr(at, bar == std::string::npos ? std::string::npos : bar - at);
    const size_t a = x.find_first_not_of(" \t\r\n"), b = x.find_last_not_of(" \t\r\n");
    f.push_back(a == std::string::npos ? "" : x.substr(a, b - a + 1));
    if (bar == std::string::npos) break;
    at = bar + 1;
  }
  return f;
}

bool read_lines(const char *path, std::vector<std::string> &lines) {
  gzFile fp = gzopen(path, "r");
  if (!fp) return false;
  std::string cur;
  char buf[1 << 16];
  int got;
  while ((got = gzread(fp, buf, sizeof(buf))) > 0) {
    for (int i = 0; i < got; ++i) {
      if (buf[i] == '\n') { lines.push_back(cur); cur.clear(); }
      else cur += buf[i];
    }
  }
  if (!cur.empty()) lines.push_back(cur);
  gzclose(fp);
  return true;
}

}  // namespace

int main(int argc, char *argv[]) {
  if (argc <= 1) { fprintf(stderr, "%s", kUsage); return 0; }
  static const char *short_options = "r:l:o:t:h";
  static struct option long_options[] = {
      {"bmax", required_argument, 0, O_BMAX}, {"dcv", required_argument, 0, O_DCV}, {"build-mem", required_argument, 0, O_MEM},
      {"offrate", required_argument, 0, O_OFFRATE}, {"ftabchars", required_argument, 0, O_FTAB}, {"rbbwt-b", required_argument, 0, O_RBB},
      {"taxonomy-tree", required_argument, 0, O_TREE}, {"conversion-table", required_argument, 0, O_CONV}, {"name-table", required_argument, 0, O_NAMES},
      {"gpu", required_argument, 0, O_GPU}, {"subset-tax", required_argument, 0, O_UNSUPPORTED}, {"concat-tax-genome", no_argument, 0, O_UNSUPPORTED},
      {"checkpoint", no_argument, 0, O_UNSUPPORTED}, {"protein", no_argument, 0, O_PROTEIN},
      {"ignore-uncategorized-genome", no_argument, 0, O_IGNORE_UNCAT}, {0, 0, 0, 0}};
  std::vector<std::string> fasta;
  std::string out_prefix = "centrifuger", tree, names_dmp, conv;
  bool ignore_uncategorized = false, ftab_given = false;
  cfr_build_options opt;
  cfr_build_options_default(&opt);
  opt.verbose = 1;
  int c, option_index = 0;
  while ((c = getopt_long(argc, argv, short_options, long_options, &option_index)) != -1) {
    switch (c) {
      case 'r': fasta.push_back(optarg); break;
      case 'l': {
        std::vector<std::string> lines;
        if (!read_lines(optarg, lines)) { print_log("ERROR: cannot open %s", optarg); return EXIT_FAILURE; }
        for (const std::string &ln : lines) {
          const size_t e = ln.find_first_of(" \t");
          if (e != std::string::npos && ln.find_first_not_of(" \t\r", e) != std::string::npos) {
            print_log("ERROR: a file list with a second column (file-level conversion) is outside this writer.");
            return EXIT_FAILURE;
          }
          const std::string f = ln.substr(0, e);
          if (!f.empty()) fasta.push_back(f);
        }
        break;
      }
      case 'o': out_prefix = optarg; break;
      case 't': opt.threads = atoi(optarg); break;
      case 'h': fprintf(stderr, "%s", kUsage); return 0;
      case O_BMAX: case O_DCV: case O_MEM: break;
      case O_OFFRATE: opt.offrate = atoi(optarg); break;
      case O_FTAB: opt.ftab_chars = atoi(optarg); ftab_given = true; break;
      case O_RBB: opt.rbbwt_b = strtoull(optarg, nullptr, 10); break;
      case O_TREE: tree = optarg; break;
      case O_CONV: conv = optarg; break;
      case O_NAMES: names_dmp = optarg; break;
      case O_GPU: opt.device = atoi(optarg); break;
      case O_IGNORE_UNCAT: ignore_uncategorized = true; break;
      case O_PROTEIN: opt.protein = 1; break;
      case O_UNSUPPORTED:
        print_log("ERROR: option --%s belongs to a part of centrifuger-build outside the MI355X writer and is not available in this build.",
                  long_options[option_index].name);
        return EXIT_FAILURE;
      default: fprintf(stderr, "%s", kUsage); return EXIT_FAILURE;
    }
  }
  // --protein: alphabet "$ARNDCEQGHILKMFPSTWYV", 4 characters in the initial lookup unless --ftabchars says otherwise (CentrifugerBuild.cpp:221-227:
  // the reference tests "width == 10", so an explicit --ftabchars 10 becomes 4 there too)
  if (opt.protein && (!ftab_given || opt.ftab_chars == 10)) opt.ftab_chars = 4;
  if (fasta.empty()) { print_log("Need to use -r/-l to specify the reference sequences."); return EXIT_FAILURE; }
  if (tree.empty() || names_dmp.empty() || conv.empty()) { print_log("Need to use --taxonomy-tree, --name-table and --conversion-table."); return EXIT_FAILURE; }

  // ---- taxonomy files (Taxonomy::Init, Taxonomy.hpp:146-180)
  std::vector<uint64_t> node_taxid, node_parent, name_taxid, seq_taxid, present_taxid;
  std::vector<std::string> node_rank, name_text, seq_name;
  {
    std::vector<std::string> lines;
    if (!read_lines(tree.c_str(), lines)) { print_log("ERROR: cannot open %s", tree.c_str()); return EXIT_FAILURE; }
    for (const std::string &ln : lines) {
      if (ln.empty() || ln[0] == '#') continue;
      const auto f = split_dmp(ln);
      if (f.size() < 3) continue;
      node_taxid.push_back(strtoull(f[0].c_str(), nullptr, 10));
      node_parent.push_back(strtoull(f[1].c_str(), nullptr, 10));
      node_rank.push_back(f[2]);
    }
    lines.clear();
    if (!read_lines(names_dmp.c_str(), lines)) { print_log("ERROR: cannot open %s", names_dmp.c_str()); return EXIT_FAILURE; }
    for (const std::string &ln : lines) {
      if (ln.find("scientific name") == std::string::npos) continue;
      const auto f = split_dmp(ln);
      if (f.size() < 2) continue;
      name_taxid.push_back(strtoull(f[0].c_str(), nullptr, 10));
      name_text.push_back(f[1]);
    }
    lines.clear();
    if (!read_lines(conv.c_str(), lines)) { print_log("ERROR: cannot open %s", conv.c_str()); return EXIT_FAILURE; }
    // a name listed with two tax ids keeps their lowest common ancestor (Taxonomy.hpp:327-352): lineages to the root, compared
    // from the root down; the root itself when even the top nodes differ
    std::map<uint64_t, uint64_t> parent_of;
    for (size_t i = 0; i < node_taxid.size(); ++i) parent_of.emplace(node_taxid[i], node_parent[i]);
    auto lineage = [&](uint64_t t) {
      std::vector<uint64_t> path;
      for (size_t guard = 0; guard <= parent_of.size(); ++guard) {
        path.push_back(t);
        auto it = parent_of.find(t);
        if (it == parent_of.end() || it->second == t) break;
        t = it->second;
      }
      return path;
    };
    uint64_t tree_root = 1;                  // Taxonomy::FindRoot (Taxonomy.hpp:426-433): the first node that is its own parent
    for (const auto &kv : parent_of) if (kv.second == kv.first) { tree_root = kv.first; break; }
    std::map<std::string, size_t> first_at;
    for (const std::string &ln : lines) {   // (every tax id the table mentions keeps its lineage in the tree, Taxonomy.hpp:275-300)
      if (ln.empty() || ln[0] == '#') continue;
      char nm[4096];
      unsigned long long tid;
      if (sscanf(ln.c_str(), "%4095s %llu", nm, &tid) != 2) continue;
      present_taxid.push_back(tid);
      auto it = first_at.find(nm);
      if (it == first_at.end()) {
        first_at.emplace(nm, seq_name.size());
        seq_name.push_back(nm);
        seq_taxid.push_back(tid);
      } else {
        // (an id the tree does not hold has the root as its whole lineage there, Taxonomy.hpp:977-984, and two lineages that part
        // at the very top meet in the root, :345-348)
        const uint64_t a = seq_taxid[it->second];
        if (!parent_of.count(a) || !parent_of.count(tid)) seq_taxid[it->second] = tree_root;
        else {
          const auto pa = lineage(a), pb = lineage(tid);
          long i = (long)pa.size() - 1, j = (long)pb.size() - 1;
          for (; i >= 0 && j >= 0; --i, --j) if (pa[(size_t)i] != pb[(size_t)j]) break;
          seq_taxid[it->second] = (i == (long)pa.size() - 1) ? tree_root : pa[(size_t)i + 1];
        }
      }
    }
  }
  const size_t n_table = seq_name.size();
  std::map<std::string, size_t> seq_index;
  for (size_t i = 0; i < seq_name.size(); ++i) seq_index.emplace(seq_name[i], i);

  // ---- sequences: FASTA (gz or plain), id = first word of the header; everything that is not an upper-case A,C,G,T is
  // dropped (SequenceCompactor::Compact, SequenceCompactor.hpp:59-84: no capitalisation, no replacement).  The text follows
  // the order of the records (Builder.hpp:108-165).
  std::vector<uint8_t> text;
  std::vector<uint64_t> genome_seq, genome_lens;
  std::vector<char> taken;                    // per sequence id: a record with this id is already part of the text
  taken.assign(seq_name.size(), 0);
  size_t n_extra = 0;
  {
    bool keep = false;                        // the record being read goes into the text
    size_t cur_id = 0, cur_start = 0;
    std::string cur_name;
    auto close_record = [&]() {
      if (!keep) return;
      const size_t len = text.size() - cur_start;
      if (len + (opt.protein ? 1 : 0) < (size_t)opt.ftab_chars + 1) {   // a genome too short (Builder.hpp:143-150; a protein's '$' counts): filtered, and its id is not taken
        fprintf(stderr, "WARNING: %s is filtered due to its short length (could be from masker)!\n", cur_name.c_str());
        text.resize(cur_start);
      } else {
        taken[cur_id] = 1;
        genome_seq.push_back(cur_id);
        genome_lens.push_back(len);
      }
      keep = false;
    };
    auto open_record = [&](const std::string &id) {
      close_record();
      cur_name = id;
      auto it = seq_index.find(id);
      if (it == seq_index.end()) {
        fprintf(stderr, "WARNING: taxonomy id doesn't exist for %s!\n", id.c_str());
        if (ignore_uncategorized) return;
        it = seq_index.emplace(id, seq_name.size()).first;      // Taxonomy::AddExtraSeqName: a new id behind the table's, no tax id
        seq_name.push_back(id);
        taken.push_back(0);
        ++n_extra;
      } else if (taken[it->second]) return;                       // a repeated sequence id is stored once (Builder.hpp:129-130)
      keep = true;
      cur_id = it->second;
      cur_start = text.size();
    };
    for (const std::string &path : fasta) {
      gzFile fp = gzopen(path.c_str(), "r");
      if (!fp) { print_log("ERROR: cannot open %s", path.c_str()); return EXIT_FAILURE; }
      gzbuffer(fp, 1 << 20);
      std::vector<char> buf(1 << 24);
      std::string header;
      bool in_header = false, at_line_start = true;
      int got;
      while ((got = gzread(fp, buf.data(), (unsigned)buf.size())) > 0) {
        for (int i = 0; i < got; ++i) {
          const char ch = buf[(size_t)i];
          if (in_header) {
            if (ch == '\n') {
              in_header = false; at_line_start = true;
              open_record(header.substr(0, header.find_first_of(" \t\r")));
            } else header += ch;
            continue;
          }
          if (at_line_start && ch == '>') { in_header = true; header.clear(); continue; }
          at_line_start = ch == '\n';
          if (keep && (opt.protein ? (ch > 0 && strchr("ARNDCEQGHILKMFPSTWYV$", ch) != nullptr) : (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'))) {
            if (ch == '$') { print_log("ERROR: a '$' inside a protein sequence is outside this writer."); return EXIT_FAILURE; }
            text.push_back((uint8_t)ch);
          }
        }
      }
      if (in_header) open_record(header.substr(0, header.find_first_of(" \t\r")));      // (a file that ends inside a header line)
      gzclose(fp);
      close_record();
    }
  }
  if (genome_seq.empty()) { print_log("ERROR: no sequence of the input is long enough to be indexed."); return EXIT_FAILURE; }
  print_log("Read %lu sequences, %lu bases (%lu names in the conversion table, %lu extra names).", (unsigned long)genome_seq.size(),
            (unsigned long)text.size(), (unsigned long)n_table, (unsigned long)n_extra);
  seq_taxid.resize(seq_name.size(), 0);       // (not read for the extra names)

  auto cptrs = [](const std::vector<std::string> &v) { std::vector<const char *> p; for (const auto &s : v) p.push_back(s.c_str()); return p; };
  const auto p_seq = cptrs(seq_name), p_rank = cptrs(node_rank), p_names = cptrs(name_text);
  cfr_build_input in;
  memset(&in, 0, sizeof(in));
  in.n_seqs = seq_name.size(); in.seq_names = p_seq.data(); in.seq_taxids = seq_taxid.data(); in.seq_lens = nullptr; in.text = text.data();
  in.n_genomes = genome_seq.size(); in.genome_seq = genome_seq.data(); in.genome_lens = genome_lens.data(); in.n_extra = n_extra;
  in.n_present_taxids = present_taxid.size(); in.present_taxids = present_taxid.data();
  in.n_nodes = node_taxid.size(); in.node_taxid = node_taxid.data(); in.node_parent = node_parent.data(); in.node_rank = p_rank.data();
  in.n_names = name_taxid.size(); in.name_taxid = name_taxid.data(); in.name_text = p_names.data();
  cfr_build_report rep;
  const cfr_status st = cfr_build_index(&in, &opt, out_prefix.c_str(), &rep);
  if (st != CFR_OK) { print_log("ERROR: index build failed (status %d): %s", st, cfr_last_error()); return EXIT_FAILURE; }
  print_log("Index of %lu bases written to %s.*.cfr (run-block size %lu, suffix array %.1f s, total %.1f s).", (unsigned long)rep.n, out_prefix.c_str(),
            (unsigned long)rep.block_size, rep.seconds_sa, rep.seconds_total);
  return 0;
}
