// cfr_cli.cpp — `centrifuger`-compatible command line on top of the C-ABI (include/cfr_hip.h).
//
// Drop-in for the reference's classification driver (CentrifugerClass.cpp:342-968) for the options that
// touch the accelerated path: same option names, same TSV on stdout (ResultWriter.hpp:186-242), same
// --un/--cl read dumps, same stderr summary lines.  What the reference does with
// pthread_create(ClassifyReads_Thread) per batch (CentrifugerClass.cpp:681-688) is one
// cfr_classify_batch call per batch here; batches round-robin over the GPUs given by --gpu, output stays
// in input order.  Additive options: --gpu LIST|all, --gpu-batch N, --gpu-throughput.
// Options of the reference that are outside this build (barcode/UMI/read-format/sample-sheet/
// merge-readpair) are rejected with a message instead of being silently ignored.
#include <fcntl.h>
#include <getopt.h>
#include <cerrno>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <atomic>
#include <chrono>
#include <functional>
#include <thread>
#include <vector>

#include "../../include/cfr_hip.h"

namespace {

const char *kUsage =
    "./centrifuger [OPTIONS] > output.tsv:\n"
    "Required:\n"
    "\t-x FILE: index prefix\n"
    "\t-1 FILE -2 FILE: paired-end read\n"
    "\t\tor\n"
    "\t-u FILE: single-end read\n"
    "\t\tor\n"
    "\t-i FILE: interleaved read file\n"
    "Optional:\n"
    "\t-t INT: number of host threads (dust masking, TSV formatting) [1]\n"
    "\t-k INT: report upto <int> distinct, primary assignments for each read pair [1]\n"
    "\t--un STR: output unclassified reads to files with the prefix of <str>\n"
    "\t--cl STR: output classified reads to files with the prefix of <str>\n"
    "\t--no-dust: do not DUST-mask low-complexity regions of reads [mask]\n"
    "\t--min-hitlen INT: minimum length of partial hits [auto]\n"
    "\t--hitk-factor INT: resolve at most <int>*k entries for each hit [40; use 0 for no restriction]\n"
    "\t--consider-secondary STR: in the format INT,FLOAT consider the secondary hit if its hitlen>=INT,score>=FLOAT*best_score [2000,0.995]\n"
    "\t--expand-taxid: output the tax IDs that are promoted to the final report tax ID [no]\n"
    "\t--gpu LIST: comma separated MI355X ordinals, or 'all' [0]\n"
    "\t--gpu-batch INT: reads per device batch [262144]\n"
    "\t--gpu-balanced: also derive the text-mode tables and the locate memo on the device (+0.4 s load per Gbp, faster kernels)\n"
    "\t--gpu-throughput: ... and the 68 GB K-mer table (longest load, fastest kernels) [default: neither, shortest load]\n"
    "\t--parse-threads INT: threads that parse plain single-end read files in pieces [min(-t,8); 1 = sequential reader]\n"
    "\t-h: print this usage message\n"
    "\t-v: print the version information and quit\n";

enum { OPT_UN = 1000, OPT_CL, OPT_NO_DUST, OPT_MIN_HITLEN, OPT_HITK, OPT_SECONDARY, OPT_GPU, OPT_GPU_BATCH, OPT_GPU_THROUGHPUT, OPT_GPU_FASTLOAD, OPT_GPU_BALANCED, OPT_PARSE_THREADS, OPT_EXPAND_TAXID, OPT_UNSUPPORTED };

void print_log(const char *fmt, ...) {   // Utils::PrintLog (compactds/Utils.hpp:369-381)
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

// Growable byte buffer for the bases of a batch; optionally in PINNED host memory (cfr_host_alloc: the bases then go to the
// device at PCIe rate, pageable memory is staged by the runtime at a third of that - but pinning 40 MB per batch object costs
// more than it returns below ~100 M reads, so it is off by default).
class ByteBuf {
 public:
  ByteBuf() = default;
  ByteBuf(const ByteBuf &) = delete;
  ByteBuf &operator=(const ByteBuf &) = delete;
  ~ByteBuf() { release(); }
  uint8_t *data() { return p_; }
  const uint8_t *data() const { return p_; }
  size_t size() const { return n_; }
  void clear() { n_ = 0; }
  void reserve(size_t want) {
    if (want <= cap_) return;
    size_t cap = cap_ ? cap_ : (1u << 20);
    while (cap < want) cap *= 2;
    bool pinned = use_pinned;
    uint8_t *q = pinned ? (uint8_t *)cfr_host_alloc(cap) : nullptr;
    if (!q) { pinned = false; q = (uint8_t *)malloc(cap); }
    if (!q) { fprintf(stderr, "out of memory\n"); exit(EXIT_FAILURE); }
    if (n_) memcpy(q, p_, n_);
    release();
    p_ = q; cap_ = cap; pinned_ = pinned;
  }
  void append(const uint8_t *src, size_t len) {
    if (n_ + len > cap_) { const size_t keep = n_; reserve(n_ + len); n_ = keep; }
    memcpy(p_ + n_, src, len);
    n_ += len;
  }
  static bool use_pinned;
 private:
  void release() {
    if (p_) { if (pinned_) cfr_host_free(p_); else free(p_); }
    p_ = nullptr; cap_ = 0;
  }
  uint8_t *p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
  bool pinned_ = false;
};
bool ByteBuf::use_pinned = false;     // measured: hipHostMalloc of 40 MB per batch costs more than the faster copies return (profiles/r2f_cli_timing.txt); CFR_CLI_PIN=1 turns it on

// ---- FASTA/FASTQ (optionally gz) record reader; id = first word of the header -----------------
// Block reads (16 MB) + memchr line splitting; sequence lines are appended straight into the batch's flat buffers
// (no per-record strings).  Record grammar as in the reference's kseq use (ReadFiles.hpp:94-160): multi-line FASTA and
// FASTQ, '>' or '@' headers, id up to the first blank, trailing /1 or /2 removed (ReadFiles.hpp:82-90).
class SeqReader {
 public:
  explicit SeqReader(const std::vector<std::string> &files) : files_(files), buf_(1u << 24) {}
  static int inflate_threads;          // threads that inflate the blocks of a BGZF file side by side (1: every .gz through gzread)
  // a byte range of a memory-mapped plain file that starts at a record header (the parallel path: ParallelFiles below)
  SeqReader(const char *mem, size_t len) : mem_(mem), pos_(0), end_(len), eof_(true) {}
  ~SeqReader() { close_file(); }

  // offset (in the memory range) at which the next record starts
  size_t next_start() const { return have_header_ ? header_off_ : pos_; }
  bool next_mem(std::vector<char> *ids, ByteBuf &seq, std::vector<char> *qual, bool &has_qual) { return read_record(ids, seq, qual, has_qual); }

  // Appends the next record: id (NUL-terminated) to ids, bases to seq, quality to qual when it is wanted.
  // returns false at the end of all files
  bool next(std::vector<char> *ids, ByteBuf &seq, std::vector<char> *qual, bool &has_qual) {
    for (;;) {
      if (!fp_ && !bgzf_base_) {
        if (file_idx_ >= files_.size()) return false;
        const std::string &f = files_[file_idx_++];
        have_header_ = false;
        pos_ = end_ = 0;
        eof_ = false;
        if (f != "-" && inflate_threads > 1 && open_bgzf(f)) start_bgzf_inflaters();
        else {
          fp_ = f == "-" ? gzdopen(fileno(stdin), "r") : gzopen(f.c_str(), "r");
          if (!fp_) { print_log("ERROR: cannot open read file %s", f.c_str()); exit(EXIT_FAILURE); }
          gzbuffer(fp_, 1 << 20);
          start_inflater();
        }
      }
      if (read_record(ids, seq, qual, has_qual)) return true;
      close_file();
    }
  }

 private:
  // next line without its line terminator; the view is valid until the next call.  false: nothing left.
  bool next_line(const char *&p, size_t &n) {
    for (;;) {
      const char *base = mem_ ? mem_ : buf_.data();
      line_off_ = pos_;
      if (const char *nl = (const char *)memchr(base + pos_, '\n', end_ - pos_)) {
        p = base + pos_;
        n = (size_t)(nl - p);
        pos_ = (size_t)(nl - base) + 1;
        while (n && p[n - 1] == '\r') --n;
        return true;
      }
      if (eof_) {
        if (pos_ == end_) return false;
        p = base + pos_;
        n = end_ - pos_;
        pos_ = end_;
        while (n && p[n - 1] == '\r') --n;
        return n > 0;
      }
      if (pos_) { memmove(buf_.data(), buf_.data() + pos_, end_ - pos_); end_ -= pos_; pos_ = 0; }
      if (end_ == buf_.size()) buf_.resize(buf_.size() * 2);
      const size_t got = pull(buf_.data() + end_, buf_.size() - end_);
      if (got == 0) eof_ = true; else end_ += got;
    }
  }
  bool read_record(std::vector<char> *ids, ByteBuf &seq, std::vector<char> *qual, bool &has_qual) {
    const char *p;
    size_t n;
    if (have_header_) { p = header_.data(); n = header_.size(); }
    else {
      do { if (!next_line(p, n)) return false; } while (n == 0 || (p[0] != '>' && p[0] != '@'));
    }
    have_header_ = false;
    const bool fastq = p[0] == '@';
    size_t e = 1;
    while (e < n && p[e] != ' ' && p[e] != '\t') ++e;
    size_t idn = e - 1;
    if (idn >= 2 && p[e - 2] == '/' && (p[e - 1] == '1' || p[e - 1] == '2')) idn -= 2;
    if (ids) { ids->insert(ids->end(), p + 1, p + 1 + idn); ids->push_back('\0'); }
    has_qual = false;
    size_t seq_n = 0;
    // kseq's grammar (kseq.h kseq_read): the sequence runs until a line that starts with '>', '@' or '+', whatever the
    // header character was and however many bases have been read; a '+' line switches to the quality block, which is
    // read line by line until it is at least as long as the sequence (at least one line, also for an empty sequence).
    (void)fastq;
    while (next_line(p, n)) {
      if (n == 0) continue;
      if (p[0] == '>' || p[0] == '@') { header_.assign(p, n); have_header_ = true; header_off_ = line_off_; return true; }
      if (p[0] == '+') {
        has_qual = true;
        size_t qn = 0;
        do {
          if (!next_line(p, n)) break;
          if (qual) qual->insert(qual->end(), p, p + n);
          qn += n;
        } while (qn < seq_n);
        return true;
      }
      seq.append((const uint8_t *)p, n);
      seq_n += n;
    }
    return true;
  }
  // ---- the inflater: gzread (one deflate stream is sequential: ~0.4 GB/s of text) runs on a thread of its own, 8 MB chunks ahead of
  // the line splitter, so that decompression and parsing overlap (a .gz input is bound by the inflate rate either way)
  struct Chunk { std::vector<char> data; size_t used = 0; };
  void start_inflater() {
    inf_stop_ = false;
    inf_eof_ = false;
    inflater_ = std::thread([this]() {
      for (;;) {
        Chunk c;
        c.data.resize(8u << 20);
        const int got = gzread(fp_, c.data.data(), (unsigned)c.data.size());
        std::unique_lock<std::mutex> lk(inf_mu_);
        if (got <= 0) { inf_eof_ = true; inf_cv_.notify_all(); return; }
        c.data.resize((size_t)got);
        inf_cv_.wait(lk, [&] { return inf_stop_ || ready_.size() < 4; });
        if (inf_stop_) return;
        ready_.push_back(std::move(c));
        inf_cv_.notify_all();
      }
    });
  }
  // ---- BGZF (bgzip, htslib): a .gz file made of independent deflate blocks of at most 64 KB whose compressed size stands in an extra
  // field of every block's header ('B' 'C', RFC 1952 FEXTRA; SAM specification section 4.1) - so the blocks can be found without
  // inflating anything, and inflated side by side.  The reference reads such a file like any .gz (kseq + gzread, ReadFiles.hpp:337): one
  // thread, ~0.4 GB/s of text; here a dispatcher walks the headers, groups of blocks (~8 MB of text) are inflated by inflate_threads
  // workers (raw inflate + CRC-32 + ISIZE of every block checked, like gzread does) and handed to the line splitter in file order.
  static bool bgzf_header(const uint8_t *b, size_t left, size_t &bsize) {        // b: start of a block; false: not a BGZF block
    if (left < 18 || b[0] != 0x1f || b[1] != 0x8b || b[2] != 8 || !(b[3] & 4)) return false;
    const size_t xlen = (size_t)b[10] | ((size_t)b[11] << 8);
    if (12 + xlen > left) return false;
    for (size_t o = 12; o + 4 <= 12 + xlen;) {
      const size_t slen = (size_t)b[o + 2] | ((size_t)b[o + 3] << 8);
      if (b[o] == 'B' && b[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) {
        bsize = ((size_t)b[o + 4] | ((size_t)b[o + 5] << 8)) + 1;
        return bsize >= 12 + xlen + 8 && bsize <= left;
      }
      o += 4 + slen;
    }
    return false;
  }
  bool open_bgzf(const std::string &path) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    uint8_t magic[4];
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 28 || pread(fd, magic, 4, 0) != 4 || magic[0] != 0x1f || magic[1] != 0x8b ||
        magic[2] != 8 || !(magic[3] & 4)) { ::close(fd); return false; }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) return false;
    // every member must be a BGZF block: a file that continues as a plain gzip member (`cat a.bgz b.gz`) is read like any .gz below,
    // which is what the reference does with it (kseq + gzread walks concatenated members).  The walk touches one header per block.
    for (size_t o = 0; o < (size_t)st.st_size;) {
      size_t bs = 0;
      if (!bgzf_header((const uint8_t *)m + o, (size_t)st.st_size - o, bs)) { munmap(m, (size_t)st.st_size); return false; }
      o += bs;
    }
    bgzf_base_ = (const uint8_t *)m;
    bgzf_size_ = (size_t)st.st_size;
    bgzf_path_ = path;
    return true;
  }
  struct BgzfTask { size_t id, off, end, text; };                               // blocks [off, end) of the file inflate to `text` bytes
  void start_bgzf_inflaters() {
    inf_stop_ = false;
    inf_eof_ = false;
    bgzf_next_deliver_ = 0;
    bgzf_issued_ = 0;
    bgzf_tasks_done_ = false;
    const int nt = inflate_threads;
    bgzf_workers_.emplace_back([this, nt]() {                                     // the dispatcher
      size_t off = 0, id = 0;
      while (off < bgzf_size_) {
        BgzfTask t{id, off, off, 0};
        while (t.end < bgzf_size_ && t.text < (8u << 20)) {
          size_t bs = 0;
          if (!bgzf_header(bgzf_base_ + t.end, bgzf_size_ - t.end, bs)) { print_log("ERROR: %s is not a well-formed BGZF file at byte %lu", bgzf_path_.c_str(), (unsigned long)t.end); exit(EXIT_FAILURE); }
          const uint8_t *tail = bgzf_base_ + t.end + bs - 4;
          t.text += (size_t)tail[0] | ((size_t)tail[1] << 8) | ((size_t)tail[2] << 16) | ((size_t)tail[3] << 24);
          t.end += bs;
        }
        off = t.end;
        std::unique_lock<std::mutex> lk(inf_mu_);
        inf_cv_.wait(lk, [&] { return inf_stop_ || bgzf_issued_ - bgzf_next_deliver_ < (size_t)(2 * nt + 2); });
        if (inf_stop_) return;
        bgzf_queue_.push_back(t);
        ++bgzf_issued_;
        ++id;
        inf_cv_.notify_all();
      }
      std::lock_guard<std::mutex> lk(inf_mu_);
      bgzf_tasks_done_ = true;
      inf_cv_.notify_all();
    });
    for (int w = 0; w < nt; ++w) bgzf_workers_.emplace_back([this]() {
      z_stream zs;
      memset(&zs, 0, sizeof(zs));
      if (inflateInit2(&zs, -15) != Z_OK) { print_log("ERROR: zlib inflateInit2 failed"); exit(EXIT_FAILURE); }
      for (;;) {
        BgzfTask t;
        {
          std::unique_lock<std::mutex> lk(inf_mu_);
          inf_cv_.wait(lk, [&] { return inf_stop_ || !bgzf_queue_.empty() || bgzf_tasks_done_; });
          if (inf_stop_ || bgzf_queue_.empty()) break;
          t = bgzf_queue_.front();
          bgzf_queue_.pop_front();
        }
        Chunk c;
        c.data.resize(t.text);
        size_t out = 0;
        for (size_t o = t.off; o < t.end;) {
          size_t bs = 0;
          bgzf_header(bgzf_base_ + o, bgzf_size_ - o, bs);
          const uint8_t *b = bgzf_base_ + o;
          const size_t xlen = (size_t)b[10] | ((size_t)b[11] << 8), hdr = 12 + xlen;
          const uint8_t *tail = b + bs - 8;
          const uint32_t crc = (uint32_t)tail[0] | ((uint32_t)tail[1] << 8) | ((uint32_t)tail[2] << 16) | ((uint32_t)tail[3] << 24);
          const size_t isize = (size_t)tail[4] | ((size_t)tail[5] << 8) | ((size_t)tail[6] << 16) | ((size_t)tail[7] << 24);
          inflateReset(&zs);
          zs.next_in = const_cast<Bytef *>(b + hdr);
          zs.avail_in = (uInt)(bs - hdr - 8);
          zs.next_out = (Bytef *)c.data.data() + out;
          zs.avail_out = (uInt)isize;
          const int rc = isize || zs.avail_in > 2 ? inflate(&zs, Z_FINISH) : Z_STREAM_END;
          if ((rc != Z_STREAM_END && !(rc == Z_OK && zs.avail_out == 0)) || zs.total_out != isize ||
              (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)c.data.data() + out, (uInt)isize) != crc) {
            print_log("ERROR: %s: a BGZF block at byte %lu does not inflate to what its trailer says", bgzf_path_.c_str(), (unsigned long)o);
            exit(EXIT_FAILURE);
          }
          out += isize;
          o += bs;
        }
        std::lock_guard<std::mutex> lk(inf_mu_);
        bgzf_done_.emplace(t.id, std::move(c));
        inf_cv_.notify_all();
      }
      inflateEnd(&zs);
    });
  }
  size_t pull_bgzf(char *dst, size_t cap) {
    std::unique_lock<std::mutex> lk(inf_mu_);
    for (;;) {
      inf_cv_.wait(lk, [&] { return bgzf_done_.count(bgzf_next_deliver_) || (bgzf_tasks_done_ && bgzf_next_deliver_ == bgzf_issued_); });
      auto it = bgzf_done_.find(bgzf_next_deliver_);
      if (it == bgzf_done_.end()) return 0;
      Chunk &c = it->second;
      const size_t n = std::min(cap, c.data.size() - c.used);
      memcpy(dst, c.data.data() + c.used, n);
      c.used += n;
      if (c.used == c.data.size()) { bgzf_done_.erase(it); ++bgzf_next_deliver_; inf_cv_.notify_all(); }
      if (n) return n;                                                           // (an empty group - the end-of-file marker block - is skipped)
    }
  }
  size_t pull(char *dst, size_t cap) {                    // up to cap bytes of decompressed text; 0 at the end of the file
    if (bgzf_base_) return pull_bgzf(dst, cap);
    std::unique_lock<std::mutex> lk(inf_mu_);
    inf_cv_.wait(lk, [&] { return !ready_.empty() || inf_eof_; });
    if (ready_.empty()) return 0;
    Chunk &c = ready_.front();
    const size_t n = std::min(cap, c.data.size() - c.used);
    memcpy(dst, c.data.data() + c.used, n);
    c.used += n;
    if (c.used == c.data.size()) { ready_.pop_front(); inf_cv_.notify_all(); }
    return n;
  }
  void close_file() {
    if (inflater_.joinable()) {
      { std::lock_guard<std::mutex> lk(inf_mu_); inf_stop_ = true; }
      inf_cv_.notify_all();
      inflater_.join();
    }
    if (!bgzf_workers_.empty()) {
      { std::lock_guard<std::mutex> lk(inf_mu_); inf_stop_ = true; }
      inf_cv_.notify_all();
      for (auto &t : bgzf_workers_) t.join();
      bgzf_workers_.clear();
      bgzf_queue_.clear();
      bgzf_done_.clear();
    }
    if (bgzf_base_) { munmap((void *)bgzf_base_, bgzf_size_); bgzf_base_ = nullptr; bgzf_size_ = 0; }
    ready_.clear();
    if (fp_) gzclose(fp_);
    fp_ = nullptr;
  }
  const uint8_t *bgzf_base_ = nullptr;
  size_t bgzf_size_ = 0, bgzf_next_deliver_ = 0, bgzf_issued_ = 0;
  bool bgzf_tasks_done_ = false;
  std::string bgzf_path_;
  std::vector<std::thread> bgzf_workers_;
  std::deque<BgzfTask> bgzf_queue_;
  std::map<size_t, Chunk> bgzf_done_;
  std::thread inflater_;
  std::mutex inf_mu_;
  std::condition_variable inf_cv_;
  std::deque<Chunk> ready_;
  bool inf_stop_ = false, inf_eof_ = false;
  std::vector<std::string> files_;
  size_t file_idx_ = 0;
  gzFile fp_ = nullptr;
  std::vector<char> buf_;
  const char *mem_ = nullptr;
  size_t pos_ = 0, end_ = 0, line_off_ = 0, header_off_ = 0;
  bool eof_ = false;
  std::string header_;
  bool have_header_ = false;
};

int SeqReader::inflate_threads = 1;

// A plain (not gz) regular read file mapped into memory, and the offsets at which it can be cut into independently parsable
// pieces.  A cut is a verified record start: FASTA - a line that starts with '>'; FASTQ - a line that starts with '@' whose
// third line starts with '+' and whose second and fourth lines have the same length (a quality line that starts with '@' fails
// that: the line two below it is a sequence).  Multi-line FASTQ never verifies; such files take the sequential reader.
struct MappedFile {
  const char *base = nullptr;
  size_t size = 0;
  int fd = -1;
  ~MappedFile() { if (base) munmap((void *)base, size); if (fd >= 0) close(fd); }
  bool open_plain(const std::string &path) {
    fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 4) return false;
    size = (size_t)st.st_size;
    void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) return false;
    base = (const char *)m;
    if ((unsigned char)base[0] == 0x1f && (unsigned char)base[1] == 0x8b) return false;      // gz
    return base[0] == '@' || base[0] == '>';
  }
  // first verified record start at or after `from` (a line start), or size
  size_t resync(size_t from) const {
    const char fmt = base[0];
    size_t at = from;
    if (at > 0 && base[at - 1] != '\n') {
      const char *nl = (const char *)memchr(base + at, '\n', size - at);
      if (!nl) return size;
      at = (size_t)(nl - base) + 1;
    }
    auto line_end = [&](size_t a) -> size_t { const char *nl = (const char *)memchr(base + a, '\n', size - a); return nl ? (size_t)(nl - base) : size; };
    auto trimmed = [&](size_t a, size_t e) -> size_t { while (e > a && base[e - 1] == '\r') --e; return e - a; };
    while (at < size) {
      const size_t e0 = line_end(at);
      if (base[at] == fmt) {
        if (fmt == '>') return at;
        const size_t l1 = e0 + 1;
        if (l1 < size) {
          const size_t e1 = line_end(l1), l2 = e1 + 1;
          if (l2 < size && base[l2] == '+') {
            const size_t e2 = line_end(l2), l3 = e2 + 1;
            if (l3 <= size) {
              const size_t e3 = l3 < size ? line_end(l3) : size;
              if (trimmed(l1, e1) == trimmed(l3, e3) && (e3 + 1 >= size || base[e3 + 1] == '@')) return at;
            }
          }
        }
      }
      at = e0 + 1;
    }
    return size;
  }
};

// Cuts at given RECORD numbers (pairs: both mate files must be cut at the same record, an interleaved file at an even one).
// The structure has to be regular for that - 4-line FASTQ (record r starts at line 4 r), or FASTA where every record starts
// with a '>' line: lines (FASTQ) or '>' lines (FASTA) are counted per chunk by several threads, then the wanted records are
// located inside their chunks.  Every cut is verified as a record start, the parsers check that a piece holds exactly the
// records it was cut for, and the caller falls back to the sequential reader when the file is not of that shape.
// every: records per piece.  cuts: byte offsets of records 0, every, 2 every, ... and the file size.  total: records in the file.
inline bool plan_record_cuts(const MappedFile &mf, size_t every, int threads, std::vector<size_t> &cuts, size_t &total) {
  const char fmt = mf.base[0];
  const bool fastq = fmt == '@';
  const size_t chunk = 4u << 20;       // (small chunks: locating a wanted record scans half a chunk line by line)
  const size_t nchunks = (mf.size + chunk - 1) / chunk;
  std::vector<size_t> units(nchunks + 1, 0);          // FASTQ: newlines in the chunk; FASTA: lines that start with '>'
  {
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    const int nt = (int)std::min<size_t>((size_t)std::max(1, threads), nchunks);
    for (int t = 0; t < nt; ++t) th.emplace_back([&]() {
      std::vector<char> buf(chunk + 1);
      for (;;) {
        const size_t k = next.fetch_add(1);
        if (k >= nchunks) return;
        const size_t lo = k * chunk, len = std::min(chunk, mf.size - lo);
        // one byte before the chunk decides whether its first byte starts a line
        const size_t from = lo ? lo - 1 : 0, want = len + (lo ? 1 : 0);
        size_t got = 0;
        while (got < want) {
          const ssize_t r = pread(mf.fd, buf.data() + got, want - got, (off_t)(from + got));
          if (r <= 0) break;
          got += (size_t)r;
        }
        if (got < want) { units[k + 1] = (size_t)-1; continue; }
        const char *p = buf.data() + (lo ? 1 : 0);
        size_t c = 0;
        if (fastq) { for (size_t i = 0; i < len; ++i) c += p[i] == '\n'; }
        else {
          if (p[0] == '>' && (lo == 0 || buf[0] == '\n')) ++c;
          for (size_t i = 1; i < len; ++i) c += (p[i] == '>') & (p[i - 1] == '\n');
        }
        units[k + 1] = c;
      }
    });
    for (auto &x : th) x.join();
  }
  for (size_t k = 0; k < nchunks; ++k) { if (units[k + 1] == (size_t)-1) return false; units[k + 1] += units[k]; }
  size_t lines = units[nchunks];
  if (fastq) {
    if (mf.base[mf.size - 1] != '\n') ++lines;
    if (lines % 4) return false;
    total = lines / 4;
  } else total = lines;
  auto verified = [&](size_t at) -> bool {
    if (at >= mf.size) return at == mf.size;
    if (mf.base[at] != fmt || (at && mf.base[at - 1] != '\n')) return false;
    if (!fastq) return true;
    const char *l1 = (const char *)memchr(mf.base + at, '\n', mf.size - at);
    if (!l1) return false;
    const char *l2 = (const char *)memchr(l1 + 1, '\n', mf.size - (size_t)(l1 + 1 - mf.base));
    return l2 && (size_t)(l2 + 1 - mf.base) < mf.size && l2[1] == '+';
  };
  cuts.clear();
  for (size_t rec = 0; rec < total; rec += every) {
    const size_t unit = fastq ? 4 * rec : rec;          // FASTQ: the record starts behind newline number `unit`; FASTA: at '>' line number `unit`
    if (fastq && unit == 0) { cuts.push_back(0); continue; }
    // FASTQ: the chunk that holds newline number `unit` (units[k] < unit <= units[k + 1]); FASTA: the chunk with '>' line number `unit`
    const size_t k = fastq ? (size_t)(std::lower_bound(units.begin(), units.end(), unit) - units.begin()) - 1
                           : (size_t)(std::upper_bound(units.begin(), units.end(), unit) - units.begin()) - 1;
    if (k >= nchunks) return false;
    size_t at = k * chunk, have = units[k];
    const size_t end = std::min(mf.size, at + chunk + 1);     // (+ 1: the record may start on the first byte of the next chunk)
    size_t pos = (size_t)-1;
    if (fastq) {
      // the byte behind newline number `unit` (1-based count of newlines seen = unit)
      while (have < unit && at < end) {
        const char *nl = (const char *)memchr(mf.base + at, '\n', end - at);
        if (!nl) break;
        ++have;
        at = (size_t)(nl - mf.base) + 1;
      }
      if (have == unit) pos = at;
    } else {
      while (at < end) {
        if (mf.base[at] == '>' && (at == 0 || mf.base[at - 1] == '\n')) { if (have == unit) { pos = at; break; } ++have; }
        const char *nl = (const char *)memchr(mf.base + at, '\n', end - at);
        if (!nl) break;
        at = (size_t)(nl - mf.base) + 1;
      }
    }
    if (pos == (size_t)-1 || !verified(pos)) return false;
    cuts.push_back(pos);
  }
  cuts.push_back(mf.size);
  return !cuts.empty() && cuts[0] == 0;
}

struct Batch {
  size_t seq_no = 0;
  size_t n = 0;
  bool paired = false;
  std::vector<char> ids;               // NUL-terminated ids back to back
  std::vector<size_t> id_off;
  std::vector<char> qual1, qual2;      // only filled when reads are dumped (--un / --cl)
  std::vector<size_t> q1_off, q2_off;
  std::vector<uint8_t> has_qual, has_qual2;
  ByteBuf bases1, bases2;
  std::vector<uint64_t> offs1, offs2;
  std::vector<cfr_result> results;
  std::vector<cfr_match> matches;
  std::vector<cfr_span> spans;         // --expand-taxid: per match slot, its list in exp_ids
  std::vector<uint64_t> exp_ids;
  std::vector<std::string> tsv_parts;  // the batch's TSV rows in order, as the formatter's workers made them (written part by part: no concatenation;
  std::vector<size_t> part_hits;       //  the strings keep their capacity across recycling); classified reads per part
  size_t n_parts = 0;
  bool done = false;
  const char *id(size_t i) const { return ids.data() + id_off[i]; }
  void reset() {   // keeps every buffer's capacity: batches are recycled, so steady state allocates (and page-faults) nothing
    n = 0; done = false;
    ids.clear(); id_off.clear(); qual1.clear(); qual2.clear(); q1_off.clear(); q2_off.clear(); has_qual.clear(); has_qual2.clear();
    bases1.clear(); bases2.clear(); offs1.clear(); offs2.clear(); n_parts = 0;
  }
};

// Persistent workers for the dust and the format stage: a 256 k-read batch is 6 ms of work on 64 threads, less than
// what starting 64 threads costs, so the threads are started once.
class WorkerPool {
 public:
  explicit WorkerPool(int n) : n_(n < 1 ? 1 : n) {
    for (int t = 1; t < n_; ++t) th_.emplace_back([this, t]() { loop(t); });
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto &x : th_) x.join();
  }
  int size() const { return n_; }
  // fn(t) for t in [0, parts): parts <= size(); the caller runs part 0 itself and returns when all are done
  void run(int parts, const std::function<void(int)> &fn) {
    if (parts <= 1) { fn(0); return; }
    { std::lock_guard<std::mutex> lk(mu_); fn_ = &fn; parts_ = parts; pending_ = parts - 1; ++gen_; }
    cv_.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&]() { return pending_ == 0; });
  }

 private:
  void loop(int t) {
    unsigned long seen = 0;
    for (;;) {
      const std::function<void(int)> *fn;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&]() { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        if (t >= parts_) continue;
        fn = fn_;
      }
      (*fn)(t);
      std::lock_guard<std::mutex> lk(mu_);
      if (--pending_ == 0) done_.notify_all();
    }
  }
  int n_;
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int)> *fn_ = nullptr;
  int parts_ = 0, pending_ = 0;
  unsigned long gen_ = 0;
  bool stop_ = false;
};

struct Options {
  std::string idx;
  std::vector<std::string> u, m1, m2, inter;
  int threads = 1;
  cfr_params params;
  bool dust = true;
  std::string un_prefix, cl_prefix;
  std::vector<int> gpus{0};
  bool all_gpus = false;
  size_t gpu_batch = 1u << 18;
  bool throughput_profile = false;
  int parse_threads = 0;               // 0 = automatic
  bool balanced_profile = false;
};

// gz read dumps (ResultWriter::SetOutputReads, ResultWriter.hpp:126-176)
struct ReadDump {
  gzFile fp[2] = {nullptr, nullptr};
  void open(const std::string &prefix, bool mate) {
    if (mate) {
      fp[0] = gzopen((prefix + "_1.fq.gz").c_str(), "w1");
      fp[1] = gzopen((prefix + "_2.fq.gz").c_str(), "w1");
    } else {
      fp[0] = gzopen((prefix + ".fq.gz").c_str(), "w1");
    }
  }
  // (gzprintf, like ResultWriter.hpp:254-265: zlib formats into its 8192-byte buffer and writes NOTHING for a record that does
  // not fit - the reference's dumps silently lack reads of ~8 kbp and more, and so do these; tests/test_gpu_cli.py)
  void put(int k, const char *id, const uint8_t *s, size_t n, const char *q, size_t qn) {
    if (!fp[k]) return;
    if (!q) gzprintf(fp[k], ">%s\n%.*s\n", id, (int)n, (const char *)s);
    else gzprintf(fp[k], "@%s\n%.*s\n+\n%.*s\n", id, (int)n, (const char *)s, (int)qn, q);
  }
  void close() { for (auto &f : fp) if (f) { gzclose(f); f = nullptr; } }
};

// CFR_CLI_TIMING=1: per-stage busy seconds on stderr at exit (stage threads overlap, so they do not add up to the wall clock)
struct StageClock {
  std::atomic<long long> ns[8];
  StageClock() { for (auto &x : ns) x = 0; }
  void add(int k, std::chrono::steady_clock::time_point t0) {
    ns[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  }
};
enum { T_OPEN = 0, T_DEVICE, T_PARSE, T_DUST, T_CLASSIFY, T_FORMAT, T_WRITE, T_WALL };
inline std::chrono::steady_clock::time_point tick() { return std::chrono::steady_clock::now(); }

[[noreturn]] void die_status(const char *what, cfr_status st) {
  print_log("ERROR: %s failed (status %d): %s", what, st, cfr_last_error());
  exit(EXIT_FAILURE);
}

}  // namespace

int main(int argc, char *argv[]) {
  if (argc <= 1) { fprintf(stderr, "%s", kUsage); return 0; }
  Options opt;
  cfr_params_default(&opt.params);
  static const char *short_options = "x:1:2:u:i:o:t:k:hv";
  static struct option long_options[] = {
      {"un", required_argument, 0, OPT_UN}, {"cl", required_argument, 0, OPT_CL}, {"no-dust", no_argument, 0, OPT_NO_DUST},
      {"min-hitlen", required_argument, 0, OPT_MIN_HITLEN}, {"hitk-factor", required_argument, 0, OPT_HITK},
      {"consider-secondary", required_argument, 0, OPT_SECONDARY}, {"gpu", required_argument, 0, OPT_GPU},
      {"gpu-batch", required_argument, 0, OPT_GPU_BATCH}, {"gpu-throughput", no_argument, 0, OPT_GPU_THROUGHPUT},
      {"gpu-fast-load", no_argument, 0, OPT_GPU_FASTLOAD}, {"gpu-balanced", no_argument, 0, OPT_GPU_BALANCED},
      {"parse-threads", required_argument, 0, OPT_PARSE_THREADS},
      {"sample-sheet", required_argument, 0, OPT_UNSUPPORTED}, {"merge-readpair", no_argument, 0, OPT_UNSUPPORTED},
      {"expand-taxid", no_argument, 0, OPT_EXPAND_TAXID}, {"read-format", required_argument, 0, OPT_UNSUPPORTED},
      {"barcode", required_argument, 0, OPT_UNSUPPORTED}, {"UMI", required_argument, 0, OPT_UNSUPPORTED},
      {"barcode-whitelist", required_argument, 0, OPT_UNSUPPORTED}, {"barcode-translate", required_argument, 0, OPT_UNSUPPORTED},
      {0, 0, 0, 0}};
  int c, option_index = 0;
  while ((c = getopt_long(argc, argv, short_options, long_options, &option_index)) != -1) {
    switch (c) {
      case 'x': opt.idx = optarg; break;
      case 'u': opt.u.push_back(optarg); break;
      case '1': opt.m1.push_back(optarg); break;
      case '2': opt.m2.push_back(optarg); break;
      case 'i': opt.inter.push_back(optarg); break;
      case 'o': break;   // accepted and unused, like the reference
      case 't': opt.threads = atoi(optarg); break;
      case 'k': opt.params.max_result = atoi(optarg); break;
      case 'v': printf("Centrifuger-MI355X %s\n", cfr_version()); return 0;
      case OPT_UN: opt.un_prefix = optarg; break;
      case OPT_CL: opt.cl_prefix = optarg; break;
      case OPT_NO_DUST: opt.dust = false; break;
      case OPT_EXPAND_TAXID: opt.params.output_expanded = 1; break;       // CentrifugerClass.cpp:453-455
      case OPT_MIN_HITLEN: opt.params.min_hit_len = atoi(optarg); break;
      case OPT_HITK: opt.params.max_result_per_hit_factor = atoi(optarg); break;
      case OPT_SECONDARY: {
        unsigned long hl; double f;
        if (sscanf(optarg, "%lu,%lf", &hl, &f) != 2) {
          print_log("Invalid format for --consider-secondary option. It should be in the format of INT,FLOAT");
          return EXIT_FAILURE;
        }
        opt.params.consider_secondary_hit_len = hl;
        opt.params.consider_secondary_score_factor = f;
        break;
      }
      case OPT_GPU:
        if (!strcmp(optarg, "all")) opt.all_gpus = true;
        else {
          opt.gpus.clear();
          for (char *tok = strtok(optarg, ","); tok; tok = strtok(nullptr, ",")) opt.gpus.push_back(atoi(tok));
        }
        break;
      case OPT_GPU_BATCH: opt.gpu_batch = strtoull(optarg, nullptr, 10); break;
      case OPT_GPU_THROUGHPUT: opt.throughput_profile = true; break;
      case OPT_GPU_FASTLOAD: break;                      // the default; accepted for symmetry
      case OPT_GPU_BALANCED: opt.balanced_profile = true; break;
      case OPT_PARSE_THREADS: opt.parse_threads = atoi(optarg); break;
      case OPT_UNSUPPORTED:
        print_log("ERROR: option --%s belongs to a part of Centrifuger outside the MI355X classification path and is not available in this build.",
                  long_options[option_index].name);
        return EXIT_FAILURE;
      default: fprintf(stderr, "%s", kUsage); return EXIT_FAILURE;
    }
  }
  print_log("Centrifuger-MI355X (%s) starts.", cfr_version());
  if (opt.idx.empty()) { print_log("Need to use -x to specify index prefix."); return EXIT_FAILURE; }
  if (opt.threads < 1) opt.threads = 1;
  SeqReader::inflate_threads = std::max(1, std::min(opt.threads, 16));
  if (opt.gpu_batch < 1) opt.gpu_batch = 1;
  {
    // a batch's match (24 B) and --expand-taxid span (16 B) buffers have max_result slots per read, on the host and on the device: with
    // -k 4096 the default batch of 262144 reads would be 43 GB of them, zero-filled per recycled batch.  Large -k therefore runs smaller
    // batches (2 GB of slots at most); the rows do not depend on where a batch ends.
    const size_t k_slots = (size_t)(opt.params.max_result > 0 ? opt.params.max_result : 4);
    const size_t by_k = std::max<size_t>(1024, (size_t)(2e9 / (40.0 * (double)k_slots)));
    if (opt.gpu_batch > by_k) opt.gpu_batch = by_k;
  }
  const bool paired = !opt.m1.empty() || !opt.inter.empty();
  if (opt.m1.size() != opt.m2.size()) { print_log("ERROR: -1 and -2 must be given the same number of times."); return EXIT_FAILURE; }
  if (opt.u.empty() && !paired) { print_log("Need to use -u/-1/-2/-i to specify input reads."); return EXIT_FAILURE; }
  if ((int)!opt.u.empty() + (int)!opt.m1.empty() + (int)!opt.inter.empty() > 1) {
    // the reference would walk a mixed file list; this build processes one kind of input per run and says so instead of dropping files
    print_log("ERROR: -u, -1/-2 and -i cannot be combined in one run of this build.");
    return EXIT_FAILURE;
  }

  if (const char *e = getenv("CFR_CLI_PARSE_ONLY")) if (atoi(e)) {
    // parser self-test hook (tests/test_host_cpu.py): records as "id<TAB>bases[<TAB>mate bases]<TAB>q|-" lines; no index, no device
    std::unique_ptr<SeqReader> r1, r2;
    const bool interleaved = !opt.inter.empty();
    if (interleaved) r1.reset(new SeqReader(opt.inter));
    else if (paired) { r1.reset(new SeqReader(opt.m1)); r2.reset(new SeqReader(opt.m2)); }
    else r1.reset(new SeqReader(opt.u));
    ByteBuf::use_pinned = false;
    std::vector<char> ids, q;
    ByteBuf s1, s2;
    bool hq = false, hq2 = false;
    if (atoi(e) == 3 && !paired) {   // the parallel cutter on every -u file: "cut offsets" then the records piece by piece
      for (const std::string &f : opt.u) {
        MappedFile mf;
        if (!mf.open_plain(f) || mf.resync(0) != 0) { puts("NOT_CUTTABLE"); continue; }
        const size_t piece = getenv("CFR_CLI_PIECE_BYTES") ? strtoull(getenv("CFR_CLI_PIECE_BYTES"), nullptr, 10) : 64;
        std::vector<size_t> cuts{0};
        while (cuts.back() + piece < mf.size) {
          const size_t c = mf.resync(cuts.back() + piece);
          if (c >= mf.size || c <= cuts.back()) break;
          cuts.push_back(c);
        }
        cuts.push_back(mf.size);
        for (size_t k = 0; k + 1 < cuts.size(); ++k) {
          SeqReader rd(mf.base + cuts[k], cuts[k + 1] - cuts[k]);
          const size_t span = cuts[k + 1] - cuts[k];
          for (;;) {
            ids.clear(); s1.clear(); q.clear();
            if (!(rd.next_start() < span && rd.next_mem(&ids, s1, &q, hq))) break;
            printf("%s\t%.*s\t%s%.*s\n", ids.data(), (int)s1.size(), (const char *)s1.data(), hq ? "q:" : "-", (int)q.size(), q.data());
          }
          if (rd.next_start() < span) puts("CUT_MISMATCH");
        }
      }
      return 0;
    }
    if (atoi(e) == 4 && paired) {   // the pair cutter (plan_record_cuts): pieces of CFR_CLI_PIECE_RECORDS pairs, records piece by piece
      const size_t every = getenv("CFR_CLI_PIECE_RECORDS") ? strtoull(getenv("CFR_CLI_PIECE_RECORDS"), nullptr, 10) : 7;
      MappedFile f1, f2;
      std::vector<size_t> c1, c2;
      size_t t1 = 0, t2 = 0;
      bool ok = interleaved ? (f1.open_plain(opt.inter[0]) && plan_record_cuts(f1, 2 * every, 3, c1, t1) && t1 % 2 == 0)
                            : (f1.open_plain(opt.m1[0]) && f2.open_plain(opt.m2[0]) && plan_record_cuts(f1, every, 3, c1, t1) && plan_record_cuts(f2, every, 3, c2, t2) && t1 == t2);
      if (!ok) { puts("NOT_CUTTABLE"); return 0; }
      for (size_t k = 0; k + 1 < c1.size(); ++k) {
        SeqReader rd1(f1.base + c1[k], c1[k + 1] - c1[k]);
        std::unique_ptr<SeqReader> rd2;
        if (!interleaved) rd2.reset(new SeqReader(f2.base + c2[k], c2[k + 1] - c2[k]));
        size_t got = 0;
        for (;;) {
          ids.clear(); s1.clear(); s2.clear(); q.clear();
          if (!(rd1.next_start() < c1[k + 1] - c1[k] && rd1.next_mem(&ids, s1, &q, hq))) break;
          if (!(interleaved ? rd1.next_mem(nullptr, s2, nullptr, hq2) : rd2->next_mem(nullptr, s2, nullptr, hq2))) { puts("MATE_MISSING"); break; }
          printf("%s\t%.*s\t%.*s\t%s%.*s\n", ids.data(), (int)s1.size(), (const char *)s1.data(), (int)s2.size(), (const char *)s2.data(), hq ? "q:" : "-", (int)q.size(), q.data());
          ++got;
        }
        if (k + 2 < c1.size() && got != every) puts("PIECE_SIZE_MISMATCH");
      }
      return 0;
    }
    if (atoi(e) == 2) {       // count only (parser throughput)
      size_t nrec = 0, nbase = 0;
      const auto tp = tick();
      for (;;) {
        if (s1.size() > (1u << 26)) { ids.clear(); s1.clear(); s2.clear(); }
        if (!r1->next(&ids, s1, nullptr, hq)) break;
        if (paired && !(interleaved ? r1->next(nullptr, s2, nullptr, hq2) : r2->next(nullptr, s2, nullptr, hq2))) break;
        ++nrec;
      }
      nbase = s1.size();
      const double sec = std::chrono::duration<double>(tick() - tp).count();
      printf("%zu records in %.3f s = %.2f M records/s (%zu)\n", nrec, sec, (double)nrec / sec * 1e-6, nbase);
      return 0;
    }
    for (;;) {
      ids.clear(); s1.clear(); s2.clear(); q.clear();
      if (!r1->next(&ids, s1, &q, hq)) break;
      if (paired && !(interleaved ? r1->next(nullptr, s2, nullptr, hq2) : r2->next(nullptr, s2, nullptr, hq2))) { fputs("MATE_MISSING\n", stdout); break; }
      printf("%s\t%.*s", ids.data(), (int)s1.size(), (const char *)s1.data());
      if (paired) printf("\t%.*s", (int)s2.size(), (const char *)s2.data());
      printf("\t%s%.*s\n", hq ? "q:" : "-", (int)q.size(), q.data());
    }
    return 0;
  }
  {   // pinned batch buffers pay off once the input is large (they cost ~0.4 s of allocation per GB of buffers)
    unsigned long long in_bytes = 0;
    for (const auto *lst : {&opt.u, &opt.m1, &opt.m2, &opt.inter}) for (const std::string &f : *lst) { struct stat st; if (f != "-" && stat(f.c_str(), &st) == 0) in_bytes += (unsigned long long)st.st_size; }
    ByteBuf::use_pinned = in_bytes > 20ull << 30;
  }
  if (const char *e = getenv("CFR_CLI_PIN")) ByteBuf::use_pinned = atoi(e) != 0;
  StageClock clk;
  const auto t_wall = tick();
  ReadDump un, cl;
  if (!opt.un_prefix.empty()) un.open(opt.un_prefix, paired);
  if (!opt.cl_prefix.empty()) cl.open(opt.cl_prefix, paired);
  if (!opt.all_gpus && opt.gpus.empty()) { print_log("ERROR: no MI355X device selected."); return EXIT_FAILURE; }
  if (opt.all_gpus) {
    int cnt = 0;
    cfr_device_count(&cnt);
    opt.gpus.clear();
    for (int g = 0; g < cnt; ++g) opt.gpus.push_back(g);
    if (opt.gpus.empty()) { print_log("ERROR: no MI355X device found (this build has no CPU fallback)."); return EXIT_FAILURE; }
  }

  // ---- pipeline: reader -> dust -> device workers (one per GPU) -> TSV formatter -> ordered writer (this thread)
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::shared_ptr<Batch>> pending;      // parsed, waiting for the dust stage
  std::deque<std::shared_ptr<Batch>> dusted;       // masked, waiting for a device
  std::deque<std::shared_ptr<Batch>> classified_q; // classified, waiting for the TSV formatter
  bool dust_done = false;
  size_t workers_finished = 0;
  std::deque<std::shared_ptr<Batch>> in_order;     // every batch in input order, for the writer
  std::deque<std::shared_ptr<Batch>> recycled;     // written out; their buffers serve the next batches
  bool reader_done = false;
  const size_t max_inflight = 16;                  // parsed batches waiting (the reader runs ahead of the index load)

  std::thread reader([&]() {
    const bool interleaved = !opt.inter.empty();
    const bool keep_qual = !opt.un_prefix.empty() || !opt.cl_prefix.empty();
    size_t seq_no = 0;
    std::atomic<size_t> bases_hint{0};
    auto fresh_batch = [&]() {
      std::shared_ptr<Batch> b;
      {
        std::lock_guard<std::mutex> lk(mu);
        if (!recycled.empty()) { b = recycled.front(); recycled.pop_front(); }
      }
      if (!b) b = std::make_shared<Batch>();
      b->reset();
      b->bases1.reserve(bases_hint.load());                 // one allocation per batch object instead of a doubling series
      if (paired) b->bases2.reserve(bases_hint.load());
      b->paired = paired;
      b->offs1.push_back(0);
      if (paired) b->offs2.push_back(0);
      b->q1_off.push_back(0);
      b->q2_off.push_back(0);
      return b;
    };
    // one record (+ mate) into b; false at the end of the input
    auto take = [&](Batch &b, SeqReader *r1, SeqReader *r2, bool from_memory) -> bool {
      bool hq = false, hq2 = false;
      const size_t id_at = b.ids.size();
      const bool ok1 = from_memory ? r1->next_mem(&b.ids, b.bases1, keep_qual ? &b.qual1 : nullptr, hq)
                                   : r1->next(&b.ids, b.bases1, keep_qual ? &b.qual1 : nullptr, hq);
      if (!ok1) return false;
      if (paired) {
        SeqReader *rm = interleaved ? r1 : r2;
        const bool ok = from_memory ? rm->next_mem(nullptr, b.bases2, keep_qual ? &b.qual2 : nullptr, hq2)
                                    : rm->next(nullptr, b.bases2, keep_qual ? &b.qual2 : nullptr, hq2);
        if (!ok) { print_log("ERROR: The two mate-pair read files have different number of reads."); exit(EXIT_FAILURE); }
        b.offs2.push_back(b.bases2.size());
        if (keep_qual) { b.q2_off.push_back(b.qual2.size()); b.has_qual2.push_back(hq2 ? 1 : 0); }
      }
      b.id_off.push_back(id_at);
      b.offs1.push_back(b.bases1.size());
      if (keep_qual) { b.q1_off.push_back(b.qual1.size()); b.has_qual.push_back(hq ? 1 : 0); }
      ++b.n;
      return true;
    };
    auto note_size = [&](const Batch &b) {
      const size_t want = std::max(b.bases1.size(), b.bases2.size()) + (std::max(b.bases1.size(), b.bases2.size()) >> 3);
      size_t cur = bases_hint.load();
      while (want > cur && !bases_hint.compare_exchange_weak(cur, want)) {}
    };
    auto publish = [&](const std::shared_ptr<Batch> &b) {
      note_size(*b);
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&]() { return in_order.size() < max_inflight; });
      pending.push_back(b);
      in_order.push_back(b);
      cv.notify_all();
    };
    auto read_sequential = [&](SeqReader *r1, SeqReader *r2) {
      bool more = true;
      while (more) {
        const auto tp = tick();
        std::shared_ptr<Batch> b = fresh_batch();
        b->seq_no = seq_no++;
        while (b->n < opt.gpu_batch) if (!take(*b, r1, r2, false)) { more = false; break; }
        if (!more && paired && !interleaved) {
          ByteBuf extra;
          bool hq2 = false;
          if (r2->next(nullptr, extra, nullptr, hq2)) { print_log("ERROR: The two mate-pair read files have different number of reads."); exit(EXIT_FAILURE); }
        }
        clk.add(T_PARSE, tp);
        if (b->n == 0) break;
        publish(b);
      }
    };
    // Plain single-end files are cut at verified record starts and parsed by several threads (MappedFile); the pieces are
    // published in file order.  Anything else (gz, stdin, pairs, files that do not verify) takes the sequential reader.
    auto read_parallel = [&](const MappedFile &mf, const std::vector<size_t> &cuts) {
      const size_t nchunks = cuts.size() - 1;
      std::atomic<size_t> next_chunk{0};
      size_t publish_next = 0;
      const int workers = std::max(1, std::min<int>(opt.parse_threads > 0 ? opt.parse_threads : std::min(opt.threads, 8), (int)nchunks));
      std::vector<std::thread> th;
      for (int w = 0; w < workers; ++w) th.emplace_back([&]() {
        std::vector<char> piece;      // the worker's private copy of its piece: pread, not page faults on a mapping every thread shares
        for (;;) {
          const size_t k = next_chunk.fetch_add(1);
          if (k >= nchunks) return;
          const auto tp = tick();
          std::shared_ptr<Batch> b = fresh_batch();
          const size_t span = cuts[k + 1] - cuts[k];
          if (piece.size() < span) piece.resize(span + (span >> 3));
          for (size_t got = 0; got < span;) {
            const ssize_t r = pread(mf.fd, piece.data() + got, span - got, (off_t)(cuts[k] + got));
            if (r <= 0) { print_log("ERROR: cannot read the read file."); exit(EXIT_FAILURE); }
            got += (size_t)r;
          }
          SeqReader rd(piece.data(), span);
          while (rd.next_start() < span && take(*b, &rd, nullptr, true)) {}
          if (rd.next_start() < span) { print_log("ERROR: read file could not be cut at byte %lu; rerun with --parse-threads 1.", (unsigned long)cuts[k + 1]); exit(EXIT_FAILURE); }
          clk.add(T_PARSE, tp);
          note_size(*b);
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&]() { return publish_next == k && in_order.size() < max_inflight; });
          if (b->n) { pending.push_back(b); in_order.push_back(b); }
          ++publish_next;
          cv.notify_all();
        }
      });
      for (auto &x : th) x.join();
      seq_no += nchunks;
    };
    // Pairs from plain files: both mate files (or the interleaved file) are cut at the same RECORD numbers (plan_record_cuts)
    // and the pieces parsed by several threads; a piece that does not hold exactly the records it was cut for stops the run
    // (never a silent shift between mates).
    auto read_parallel_pairs = [&](const MappedFile &f1, const std::vector<size_t> &c1, const MappedFile *f2, const std::vector<size_t> &c2,
                                   size_t total_pairs) {
      const size_t nchunks = c1.size() - 1;
      std::atomic<size_t> next_chunk{0};
      size_t publish_next = 0;
      const int workers = std::max(1, std::min<int>(opt.parse_threads > 0 ? opt.parse_threads : std::min(opt.threads, 8), (int)nchunks));
      std::vector<std::thread> th;
      for (int w = 0; w < workers; ++w) th.emplace_back([&]() {
        std::vector<char> p1, p2;
        auto fetch = [&](const MappedFile &mf, size_t lo, size_t hi, std::vector<char> &dst) {
          const size_t span = hi - lo;
          if (dst.size() < span) dst.resize(span + (span >> 3));
          for (size_t got = 0; got < span;) {
            const ssize_t r = pread(mf.fd, dst.data() + got, span - got, (off_t)(lo + got));
            if (r <= 0) { print_log("ERROR: cannot read the read file."); exit(EXIT_FAILURE); }
            got += (size_t)r;
          }
          return span;
        };
        for (;;) {
          const size_t k = next_chunk.fetch_add(1);
          if (k >= nchunks) return;
          const auto tp = tick();
          std::shared_ptr<Batch> b = fresh_batch();
          const size_t want = std::min(opt.gpu_batch, total_pairs - k * opt.gpu_batch);
          const size_t s1 = fetch(f1, c1[k], c1[k + 1], p1);
          SeqReader rd1(p1.data(), s1);
          if (f2) {
            const size_t s2 = fetch(*f2, c2[k], c2[k + 1], p2);
            SeqReader rd2(p2.data(), s2);
            while (b->n < want && take(*b, &rd1, &rd2, true)) {}
            if (b->n != want || rd1.next_start() < s1 || rd2.next_start() < s2) {
              print_log("ERROR: the mate files are not made of regular records around pair %lu; rerun with --parse-threads 1.", (unsigned long)(k * opt.gpu_batch + b->n));
              exit(EXIT_FAILURE);
            }
          } else {
            while (b->n < want && take(*b, &rd1, nullptr, true)) {}
            if (b->n != want || rd1.next_start() < s1) {
              print_log("ERROR: the interleaved file is not made of regular records around pair %lu; rerun with --parse-threads 1.", (unsigned long)(k * opt.gpu_batch + b->n));
              exit(EXIT_FAILURE);
            }
          }
          clk.add(T_PARSE, tp);
          note_size(*b);
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&]() { return publish_next == k && in_order.size() < max_inflight; });
          if (b->n) { pending.push_back(b); in_order.push_back(b); }
          ++publish_next;
          cv.notify_all();
        }
      });
      for (auto &x : th) x.join();
      seq_no += nchunks;
    };
    bool pairs_done = false;
    if (paired && opt.parse_threads != 1 && (interleaved ? opt.inter.size() == 1 : (opt.m1.size() == 1 && opt.m2.size() == 1))) {
      MappedFile f1, f2;
      std::vector<size_t> c1, c2;
      size_t t1 = 0, t2 = 0;
      const int pt = opt.parse_threads > 0 ? opt.parse_threads : std::min(opt.threads, 8);
      if (interleaved) {
        if (opt.inter[0] != "-" && f1.open_plain(opt.inter[0]) && plan_record_cuts(f1, 2 * opt.gpu_batch, pt, c1, t1) && t1 % 2 == 0 && c1.size() >= 3) {
          read_parallel_pairs(f1, c1, nullptr, c2, t1 / 2);
          pairs_done = true;
        }
      } else if (opt.m1[0] != "-" && opt.m2[0] != "-" && f1.open_plain(opt.m1[0]) && f2.open_plain(opt.m2[0]) &&
                 plan_record_cuts(f1, opt.gpu_batch, pt, c1, t1) && plan_record_cuts(f2, opt.gpu_batch, pt, c2, t2) && c1.size() >= 3) {
        if (t1 != t2) { print_log("ERROR: The two mate-pair read files have different number of reads."); exit(EXIT_FAILURE); }
        read_parallel_pairs(f1, c1, &f2, c2, t1);
        pairs_done = true;
      }
    }
    if (pairs_done) {
    } else
    if (!paired && opt.parse_threads != 1) {
      for (const std::string &f : opt.u) {
        MappedFile mf;
        std::vector<size_t> cuts;
        if (f != "-" && mf.open_plain(f) && mf.resync(0) == 0) {
          // pieces of about one device batch: bytes per record from the first record
          const size_t second = mf.resync(1);
          const size_t rec_bytes = std::max<size_t>(16, second);
          const size_t piece = std::max<size_t>(1u << 16, rec_bytes * opt.gpu_batch);
          cuts.push_back(0);
          while (cuts.back() + piece < mf.size) {
            const size_t c = mf.resync(cuts.back() + piece);
            if (c >= mf.size || c <= cuts.back()) break;
            cuts.push_back(c);
          }
          cuts.push_back(mf.size);
        }
        if (cuts.size() >= 3) read_parallel(mf, cuts);
        else { SeqReader r1(std::vector<std::string>{f}); read_sequential(&r1, nullptr); }
      }
    } else {
      std::unique_ptr<SeqReader> r1, r2;
      if (interleaved) r1.reset(new SeqReader(opt.inter));
      else if (paired) { r1.reset(new SeqReader(opt.m1)); r2.reset(new SeqReader(opt.m2)); }
      else r1.reset(new SeqReader(opt.u));
      read_sequential(r1.get(), r2.get());
    }
    std::lock_guard<std::mutex> lk(mu);
    reader_done = true;
    cv.notify_all();
  });

  // dust stage (CentrifugerClass.cpp:276-316): needs no index either
  // The masking itself runs on the device inside the classify call (cfr_device_index_set_dust); the host stage only works
  // when the masked reads are needed on the host as well (--un / --cl write them).
  // a protein index (.4.cfr says sequence_type amino_acid, Classifier::IsProteinDatabase) is searched translated and never
  // dust-masked (CentrifugerClass.cpp:248, 276); the stage below must know before the index itself is loaded
  bool protein = false;
  if (FILE *f4 = fopen((opt.idx + ".4.cfr").c_str(), "r")) {
    char key[128], val[128];
    while (fscanf(f4, "%127s %127s", key, val) == 2) if (!strcmp(key, "sequence_type") && !strcmp(val, "amino_acid")) protein = true;
    fclose(f4);
  }
  if (protein) opt.dust = false;
  const bool host_dust = opt.dust && (!opt.un_prefix.empty() || !opt.cl_prefix.empty());
  WorkerPool dust_pool(host_dust ? opt.threads : 1), format_pool(opt.threads);
  std::thread duster([&]() {
    for (;;) {
      std::shared_ptr<Batch> b;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return !pending.empty() || reader_done; });
        if (pending.empty()) break;
        b = pending.front();
        pending.pop_front();
      }
      const auto ts = tick();
      if (opt.dust && host_dust) {      // slices of the batch on the stage's own workers (offsets are absolute, so a slice is just an offset window)
        const int parts = (int)std::min<size_t>((size_t)dust_pool.size(), std::max<size_t>(1, b->n / 2048));
        dust_pool.run(parts, [&](int t) {
          const size_t lo = b->n * (size_t)t / (size_t)parts, hi = b->n * (size_t)(t + 1) / (size_t)parts;
          cfr_dust_mask_batch(b->bases1.data(), b->offs1.data() + lo, hi - lo, 1);
          if (b->paired) cfr_dust_mask_batch(b->bases2.data(), b->offs2.data() + lo, hi - lo, 1);
        });
      }
      clk.add(T_DUST, ts);
      std::lock_guard<std::mutex> lk(mu);
      dusted.push_back(b);
      cv.notify_all();
    }
    std::lock_guard<std::mutex> lk(mu);
    dust_done = true;
    cv.notify_all();
  });

  // ---- index load and device image, while the reader and dust threads already work on the first batches
  auto t0 = tick();
  cfr_index *idx = nullptr;
  cfr_status st = cfr_index_open(opt.idx.c_str(), &opt.params, &idx);
  if (st != CFR_OK) die_status("loading the index", st);
  clk.add(T_OPEN, t0);
  cfr_index_info info;
  cfr_index_get_info(idx, &info);
  if (info.is_protein) print_log("This is a protein database and will use translated search.");
  print_log("Finishes loading index.");
  if (opt.params.min_hit_len <= 0) print_log("Inferred --min-hitlen: %d", info.min_hit_len);
  std::vector<cfr_dev_index *> devs;
  t0 = tick();
  cfr_device_options dopt;
  cfr_device_options_default(&dopt);
  // default: the fast-load image.  This program hands the device pageable host buffers, which bounds the device stage at
  // ~30 M reads/s whatever the tables (profiles/r2f_cli_timing.txt): what a run feels is the load time.  --gpu-balanced adds the
  // text-mode tables and the locate memo (+0.4 s per Gbp), --gpu-throughput the 68 GB K-mer table as well.
  dopt.profile = opt.throughput_profile ? CFR_PROFILE_THROUGHPUT : opt.balanced_profile ? CFR_PROFILE_BALANCED : CFR_PROFILE_FAST_LOAD;
  if (!opt.throughput_profile && !opt.balanced_profile) {
    // no profile asked for: by the size of the input.  The fast-load image classifies ~30 M reads/s; the text-mode tables of the
    // balanced image cost ~0.4 s per Gbp of index once and lift that several-fold, which pays from a few GB of reads on (plain
    // files: bytes on disk; gz: ~4x that when inflated; stdin: unknown, stays fast-load).
    unsigned long long bytes = 0;
    for (const auto *lst : {&opt.u, &opt.m1, &opt.m2, &opt.inter})
      for (const std::string &f : *lst) {
        struct stat st;
        if (f != "-" && stat(f.c_str(), &st) == 0 && S_ISREG(st.st_mode)) {
          const bool gz = f.size() > 3 && f.compare(f.size() - 3, 3, ".gz") == 0;
          bytes += (unsigned long long)st.st_size * (gz ? 4ull : 1ull);
        }
      }
    if (bytes >= (8ull << 30)) dopt.profile = CFR_PROFILE_BALANCED;
  }
  for (int g : opt.gpus) {
    cfr_dev_index *d = nullptr;
    st = cfr_device_index_create_ex(idx, g, &dopt, &d);
    if (st != CFR_OK) die_status("creating the device index (this build has no CPU fallback)", st);
    if (opt.dust && !host_dust) cfr_device_index_set_dust(d, 1);
    devs.push_back(d);
  }
  clk.add(T_DEVICE, t0);
  const bool expand = opt.params.output_expanded != 0;
  fputs(expand ? cfr_tsv_header_expanded() : cfr_tsv_header(), stdout);          // only once the index and the devices are up: a failed load prints no TSV at all


  // device stage: one thread per GPU takes dust-masked batches
  auto worker = [&](cfr_dev_index *dev) {
    for (;;) {
      std::shared_ptr<Batch> b;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return !dusted.empty() || dust_done; });
        if (dusted.empty()) break;
        b = dusted.front();
        dusted.pop_front();
      }
      const auto ts = tick();
      b->results.resize(b->n);
      size_t cap = b->n * (size_t)(opt.params.max_result > 0 ? opt.params.max_result : 4) + 16, used = 0;
      size_t ids_cap = expand ? std::max<size_t>(b->exp_ids.size(), 4 * b->n + 16) : 0, ids_used = 0;
      for (;;) {
        b->matches.resize(cap);
        cfr_status s;
        if (expand) {
          b->spans.resize(cap);
          b->exp_ids.resize(ids_cap);
          s = cfr_classify_batch_expanded(dev, b->bases1.data(), b->offs1.data(), b->paired ? b->bases2.data() : nullptr,
                                          b->paired ? b->offs2.data() : nullptr, b->n, b->results.data(), b->matches.data(), b->spans.data(), cap, &used,
                                          b->exp_ids.data(), ids_cap, &ids_used);
        } else
        s = cfr_classify_batch(dev, b->bases1.data(), b->offs1.data(), b->paired ? b->bases2.data() : nullptr,
                               b->paired ? b->offs2.data() : nullptr, b->n, b->results.data(), b->matches.data(), cap, &used);
        if (s == CFR_ERR_CAPACITY) { if (used > cap) cap = used + 16; if (ids_used > ids_cap) ids_cap = ids_used + 16; continue; }
        if (s != CFR_OK) die_status("cfr_classify_batch", s);
        break;
      }
      clk.add(T_CLASSIFY, ts);
      std::lock_guard<std::mutex> lk(mu);
      classified_q.push_back(b);
      cv.notify_all();
    }
    std::lock_guard<std::mutex> lk(mu);
    ++workers_finished;
    cv.notify_all();
  };
  std::vector<std::thread> workers;
  for (cfr_dev_index *d : devs) workers.emplace_back(worker, d);

  // format stage: TSV rows (ResultWriter::Output), formatted in parallel slices then concatenated in order
  std::thread formatter([&]() {
    for (;;) {
      std::shared_ptr<Batch> b;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return !classified_q.empty() || workers_finished == devs.size(); });
        if (classified_q.empty()) return;
        b = classified_q.front();
        classified_q.pop_front();
      }
      const auto ts = tick();
      const int nt = (int)std::min<size_t>((size_t)format_pool.size(), std::max<size_t>(1, b->n / 4096));
      if (b->tsv_parts.size() < (size_t)nt) { b->tsv_parts.resize((size_t)nt); b->part_hits.resize((size_t)nt); }
      b->n_parts = (size_t)nt;
      format_pool.run(nt, [&](int t) {
        const size_t lo = b->n * (size_t)t / (size_t)nt, hi = b->n * (size_t)(t + 1) / (size_t)nt;
        std::string &out = b->tsv_parts[(size_t)t];
        out.clear();
        if (out.capacity() < (hi - lo) * 64) out.reserve((hi - lo) * 96);
        size_t hits = 0;
        for (size_t i = lo; i < hi; ++i) hits += b->results[i].n_match > 0 ? 1 : 0;
        b->part_hits[(size_t)t] = hits;
        char buf[8192];
        auto row = [&](size_t i, char *dst, size_t cap) {
          return expand ? cfr_format_tsv_expanded(idx, b->id(i), &b->results[i], b->matches.data(), b->spans.data(), b->exp_ids.data(), dst, cap)
                        : cfr_format_tsv(idx, b->id(i), &b->results[i], b->matches.data(), dst, cap);
        };
        for (size_t i = lo; i < hi; ++i) {
          size_t w = row(i, buf, sizeof(buf));
          if (w < sizeof(buf)) out.append(buf, w);
          else {
            std::string big(w + 1, '\0');
            row(i, &big[0], big.size());
            out.append(big.data(), w);
          }
        }
      });
      clk.add(T_FORMAT, ts);
      std::lock_guard<std::mutex> lk(mu);
      b->done = true;
      cv.notify_all();
    }
  });

  size_t total = 0, classified = 0;
  for (;;) {
    std::shared_ptr<Batch> b;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&]() { return (!in_order.empty() && in_order.front()->done) || (reader_done && in_order.empty()); });
      if (in_order.empty()) break;
      b = in_order.front();
      in_order.pop_front();
      cv.notify_all();
    }
    const auto tw = tick();
    for (size_t t = 0; t < b->n_parts; ++t) { fwrite(b->tsv_parts[t].data(), 1, b->tsv_parts[t].size(), stdout); classified += b->part_hits[t]; }
    total += b->n;
    if (cl.fp[0] || un.fp[0]) for (size_t i = 0; i < b->n; ++i) {
      const bool hit = b->results[i].n_match > 0;
      ReadDump *dump = hit ? (cl.fp[0] ? &cl : nullptr) : (un.fp[0] ? &un : nullptr);
      if (!dump) continue;
      dump->put(0, b->id(i), b->bases1.data() + b->offs1[i], b->offs1[i + 1] - b->offs1[i],
                b->has_qual[i] ? b->qual1.data() + b->q1_off[i] : nullptr, b->q1_off[i + 1] - b->q1_off[i]);
      if (b->paired)
        dump->put(1, b->id(i), b->bases2.data() + b->offs2[i], b->offs2[i + 1] - b->offs2[i],
                  b->has_qual2[i] ? b->qual2.data() + b->q2_off[i] : nullptr, b->q2_off[i + 1] - b->q2_off[i]);
    }
    clk.add(T_WRITE, tw);
    std::lock_guard<std::mutex> lk(mu);
    recycled.push_back(b);
  }
  reader.join();
  duster.join();
  for (auto &w : workers) w.join();
  formatter.join();
  un.close();
  cl.close();
  fflush(stdout);
  // ResultWriter::Finalize (ResultWriter.hpp:279-283)
  print_log("Processed %lu read fragments, and %lu (%.2lf%%) can be classified.", (unsigned long)total, (unsigned long)classified,
            total ? (double)classified / (double)total * 100.0 : 0.0);
  for (auto *d : devs) cfr_device_index_destroy(d);
  cfr_index_destroy(idx);
  clk.add(T_WALL, t_wall);
  if (const char *e = getenv("CFR_CLI_TIMING")) if (atoi(e)) {
    static const char *names[] = {"index_open", "device_index", "parse", "dust", "classify", "format", "write", "wall"};
    for (int k = 0; k < 8; ++k) fprintf(stderr, "[timing] %-12s %8.3f s\n", names[k], (double)clk.ns[k].load() * 1e-9);
  }
  // the TSV must have reached its destination before success is reported (a full disk or a closed pipe otherwise ends in a
  // truncated file with exit status 0); only then is the runtime's teardown skipped (~0.3 s of a sub-second run)
  if (fflush(stdout) != 0 || ferror(stdout)) {
    print_log("ERROR: writing the classification output failed (%s).", strerror(errno));
    fflush(stderr);
    _exit(EXIT_FAILURE);
  }
  print_log("Centrifuger finishes.");
  fflush(stderr);
  _exit(0);
}
