// cfr_capi.cpp — the extern "C" surface declared in include/cfr_hip.h.
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "cfr_build.hpp"
#include "cfr_device.hpp"
#include "cfr_tail.hpp"

struct cfr_index { cfr::HostIndex *h; };
// One call at a time per device image: `busy` is held for the length of every entry that touches the image (try_lock:
// CFR_ERR_BUSY for the second caller) and by the worker thread while it runs a submitted batch.
struct cfr_async_job { std::function<cfr_status(std::string &, size_t &)> run; uint64_t ticket; };
struct cfr_async_done { cfr_status status; std::string err; size_t n_matches; };
struct cfr_dev_index {
  cfr::DeviceIndex *d; const cfr_index *host; int tail_threads;
  std::mutex busy;
  // submitted batches (cfr_classify_batch_submit / _wait)
  std::mutex qmu;
  std::condition_variable qcv, dcv;
  std::deque<cfr_async_job> queue;
  std::map<uint64_t, cfr_async_done> done;
  std::set<uint64_t> live;                               // tickets handed out and not yet waited for
  uint64_t next_ticket = 1;
  bool stop = false, worker_started = false;
  std::thread worker;
  cfr_dev_index(cfr::DeviceIndex *dd, const cfr_index *h, int t) : d(dd), host(h), tail_threads(t) {}
};

namespace {
thread_local std::string g_err;

template <class F> cfr_status guarded(F &&f) {
  try {
    return f();
  } catch (const cfr::IoError &e) {
    g_err = e.msg; return CFR_ERR_IO;
  } catch (const cfr::FormatError &e) {
    g_err = e.msg; return CFR_ERR_FORMAT;
  } catch (const cfr::CapacityError &e) {
    g_err = e.msg; return CFR_ERR_CAPACITY;
  } catch (const cfr::HipError &e) {
    g_err = e.msg; return e.code == -1 ? CFR_ERR_NO_DEVICE : CFR_ERR_HIP;
  } catch (const std::exception &e) {
    g_err = e.what(); return CFR_ERR_ARG;
  }
}
cfr_status bad_arg(const char *m) { g_err = m; return CFR_ERR_ARG; }

// entry guard: the image's buffers, streams and statistics belong to one call at a time
struct BusyGuard {
  std::unique_lock<std::mutex> lk;
  explicit BusyGuard(cfr_dev_index *d) : lk(d->busy, std::try_to_lock) {}
  bool ok() const { return lk.owns_lock(); }
};
#define CFR_ENTER(d, what)                                                                                                   \
  BusyGuard busy_guard_(d);                                                                                                  \
  if (!busy_guard_.ok()) { g_err = what ": this cfr_dev_index is in use by another call (one call at a time per device image)"; return CFR_ERR_BUSY; }

void worker_loop(cfr_dev_index *d) {
  for (;;) {
    cfr_async_job job;
    {
      std::unique_lock<std::mutex> lk(d->qmu);
      d->qcv.wait(lk, [&] { return d->stop || !d->queue.empty(); });
      if (d->queue.empty()) return;                      // stop, nothing left to run
      job = std::move(d->queue.front());
      d->queue.pop_front();
    }
    cfr_async_done r{CFR_OK, "", 0};
    {
      std::lock_guard<std::mutex> run(d->busy);          // (waits for a synchronous call that is inside the image)
      r.status = job.run(r.err, r.n_matches);
    }
    {
      std::lock_guard<std::mutex> lk(d->qmu);
      d->done.emplace(job.ticket, std::move(r));
    }
    d->dcv.notify_all();
  }
}

void fill_info(const cfr::HostIndex &h, cfr_index_info *info, uint64_t dev_bytes) {
  memset(info, 0, sizeof(*info));
  info->n = h.n; info->first_isa = h.first_isa; info->block_size = h.b; info->precompute_width = h.precompute_width;
  info->sample_rate = (uint64_t)h.sample_rate; info->selected_cnt = h.selected_rows.size();
  info->seq_cnt = h.tax.seq_cnt; info->node_cnt = h.tax.node_cnt; info->min_hit_len = h.params.min_hit_len;
  info->last_chr = h.last_chr; info->is_protein = h.prot.enabled ? 1 : 0; info->device_bytes = dev_bytes;
}

int default_tail_threads() {
  unsigned hc = std::thread::hardware_concurrency();
  if (hc == 0) hc = 1;
  return (int)std::min(hc, 64u);
}
}  // namespace

extern "C" {

void cfr_params_default(cfr_params *p) {   // _classifierParam() (Classifier.hpp:28-37)
  p->max_result = 1; p->min_hit_len = 0; p->max_result_per_hit_factor = 40; p->output_expanded = 0;
  p->consider_secondary_hit_len = 2000; p->consider_secondary_score_factor = 0.995;
}
const char *cfr_last_error(void) { return g_err.c_str(); }
const char *cfr_version(void) { return "centrifuger_amd 0.1 (gfx950); path parity with Centrifuger v1.1.3-r347"; }

cfr_status cfr_index_open(const char *idx_prefix, const cfr_params *params, cfr_index **out) {
  if (!idx_prefix || !out) return bad_arg("cfr_index_open: null argument");
  return guarded([&]() -> cfr_status {
    cfr::HostIndex *h = cfr::load_index(idx_prefix, params);
    *out = new cfr_index{h};
    return CFR_OK;
  });
}
void cfr_index_destroy(cfr_index *idx) { if (idx) { delete idx->h; delete idx; } }
cfr_status cfr_index_get_info(const cfr_index *idx, cfr_index_info *info) {
  if (!idx || !info) return bad_arg("cfr_index_get_info: null argument");
  fill_info(*idx->h, info, 0);
  return CFR_OK;
}

cfr_status cfr_index_digest(const cfr_index *idx, uint64_t *digest) {
  if (!idx || !digest) return bad_arg("cfr_index_digest: null argument");
  *digest = cfr::index_digest(*idx->h);
  return CFR_OK;
}

cfr_status cfr_index_mapped_bytes(const cfr_index *idx, uint64_t *mapped, uint64_t *copied) {
  if (!idx || !mapped || !copied) return bad_arg("cfr_index_mapped_bytes: null argument");
  const cfr::HostIndex &h = *idx->h;
  uint64_t m = 0, c = 0;
  auto add = [&](const cfr::RawWords &w) { (w.mapped() ? m : c) += (uint64_t)w.size() * 8; };
  add(h.use_run_block.bits);
  for (int k = 0; k < 3; ++k) { add(h.wavelet_seq.node[k].bits); add(h.run_block_seq.node[k].bits); }
  add(h.sampled_words);
  *mapped = m;
  *copied = c;
  return CFR_OK;
}

cfr_status cfr_device_count(int *count) {
  if (!count) return bad_arg("cfr_device_count: null argument");
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) { *count = 0; g_err = "hipGetDeviceCount failed"; return CFR_ERR_NO_DEVICE; }
  *count = c;
  return CFR_OK;
}
void cfr_device_options_default(cfr_device_options *o) {
  if (!o) return;
  o->profile = CFR_PROFILE_THROUGHPUT;
  o->ftabx_width = -1;
  o->text_mode = -1;
  o->run_block_layout = 0;
  o->loc_memo_gb = -1.0;
  o->sub_batch = 0;
}
cfr_status cfr_device_index_create_ex(const cfr_index *idx, int device, const cfr_device_options *options, cfr_dev_index **out) {
  if (!idx || !out) return bad_arg("cfr_device_index_create: null argument");
  cfr_device_options o;
  cfr_device_options_default(&o);
  if (options) o = *options;
  if (o.profile != CFR_PROFILE_THROUGHPUT && o.profile != CFR_PROFILE_FAST_LOAD && o.profile != CFR_PROFILE_BALANCED) return bad_arg("cfr_device_index_create_ex: unknown profile");
  if (o.ftabx_width < -1 || o.ftabx_width > 16) return bad_arg("cfr_device_index_create_ex: ftabx_width out of range");
  return guarded([&]() -> cfr_status {
    cfr::DeviceIndex *d = new cfr::DeviceIndex(*idx->h, device, o);
    *out = new cfr_dev_index(d, idx, default_tail_threads());
    return CFR_OK;
  });
}
cfr_status cfr_device_index_create(const cfr_index *idx, int device, cfr_dev_index **out) {
  return cfr_device_index_create_ex(idx, device, nullptr, out);
}
void cfr_device_index_destroy(cfr_dev_index *d) {
  if (!d) return;
  if (d->worker_started) {                               // queued batches run to completion first (their buffers are the caller's)
    { std::lock_guard<std::mutex> lk(d->qmu); d->stop = true; }
    d->qcv.notify_all();
    d->worker.join();
  }
  { std::lock_guard<std::mutex> wait_for_a_running_call(d->busy); }
  delete d->d;
  delete d;
}
cfr_status cfr_device_index_get_info(const cfr_dev_index *d, cfr_index_info *info) {
  if (!d || !info) return bad_arg("cfr_device_index_get_info: null argument");
  fill_info(d->d->host(), info, d->d->device_bytes());
  return CFR_OK;
}

cfr_status cfr_rank_batch(cfr_dev_index *d, const char *chars, const uint64_t *pos, const uint8_t *inclusive, size_t n,
                          uint64_t *out_rank, char *out_access) {
  if (!d || (n && (!chars || !pos || !inclusive))) return bad_arg("cfr_rank_batch: null argument");
  for (size_t i = 0; i < n; ++i) if (pos[i] >= d->d->host().n) return bad_arg("cfr_rank_batch: position out of range");
  CFR_ENTER(d, "cfr_rank_batch");
  return guarded([&]() -> cfr_status { d->d->rank_batch(chars, pos, inclusive, n, out_rank, out_access); return CFR_OK; });
}
cfr_status cfr_backward_search_batch(cfr_dev_index *d, const uint8_t *bases, const uint64_t *offsets, const uint32_t *m, size_t n,
                                     uint64_t *out_l, uint64_t *out_sp, uint64_t *out_ep) {
  if (!d || (n && (!bases || !offsets || !m || !out_l || !out_sp || !out_ep))) return bad_arg("cfr_backward_search_batch: null argument");
  for (size_t i = 0; i < n; ++i) if (m[i] > offsets[i + 1] - offsets[i]) return bad_arg("cfr_backward_search_batch: m exceeds read length");
  CFR_ENTER(d, "cfr_backward_search_batch");
  return guarded([&]() -> cfr_status { d->d->backward_search_batch(bases, offsets, m, n, out_l, out_sp, out_ep); return CFR_OK; });
}
cfr_status cfr_locate_rows(cfr_dev_index *d, const uint64_t *rows, size_t n, uint64_t *out_val, uint32_t *out_steps) {
  if (!d || (n && (!rows || !out_val))) return bad_arg("cfr_locate_rows: null argument");
  for (size_t i = 0; i < n; ++i) if (rows[i] >= d->d->host().n) return bad_arg("cfr_locate_rows: row out of range");
  CFR_ENTER(d, "cfr_locate_rows");
  return guarded([&]() -> cfr_status { d->d->locate_rows(rows, n, out_val, out_steps); return CFR_OK; });
}

void cfr_build_options_default(cfr_build_options *o) {
  if (!o) return;
  memset(o, 0, sizeof(*o));
  o->ftab_chars = 10; o->offrate = 4;
}
cfr_status cfr_build_index(const cfr_build_input *in, const cfr_build_options *opt, const char *out_prefix, cfr_build_report *report) {
  if (!in || !out_prefix) return bad_arg("cfr_build_index: null argument");
  if (!in->n_seqs || !in->seq_names || !in->seq_taxids || !in->text) return bad_arg("cfr_build_index: empty input");
  if (in->n_genomes ? (!in->genome_seq || !in->genome_lens) : !in->seq_lens) return bad_arg("cfr_build_index: genome lengths missing");
  if (in->n_extra > in->n_seqs || (!in->n_genomes && in->n_extra)) return bad_arg("cfr_build_index: n_extra needs the genome list and cannot exceed n_seqs");
  if ((in->n_nodes && (!in->node_taxid || !in->node_parent || !in->node_rank)) || (in->n_names && (!in->name_taxid || !in->name_text)))
    return bad_arg("cfr_build_index: taxonomy arrays missing");
  return guarded([&]() -> cfr_status {
    cfr::BuildInput bi;
    for (uint64_t i = 0; i < in->n_seqs; ++i) {
      bi.names.emplace_back(in->seq_names[i]);
      if (i < in->n_seqs - in->n_extra) bi.taxids.push_back(in->seq_taxids[i]);
    }
    bi.n_extra = in->n_extra;
    if (in->n_present_taxids && in->present_taxids) bi.present_taxids.assign(in->present_taxids, in->present_taxids + in->n_present_taxids);
    if (in->n_genomes) {
      bi.genome_seq.assign(in->genome_seq, in->genome_seq + in->n_genomes);
      bi.lens.assign(in->genome_lens, in->genome_lens + in->n_genomes);
    } else {
      for (uint64_t i = 0; i < in->n_seqs; ++i) { bi.genome_seq.push_back(i); bi.lens.push_back(in->seq_lens[i]); }
    }
    bi.text = in->text;
    for (uint64_t i = 0; i < in->n_nodes; ++i) bi.nodes.push_back(cfr::TaxNode{in->node_taxid[i], in->node_parent[i], in->node_rank[i] ? in->node_rank[i] : ""});
    for (uint64_t i = 0; i < in->n_names; ++i) bi.tax_names.emplace_back(in->name_taxid[i], in->name_text[i] ? in->name_text[i] : "");
    cfr::BuildOptions bo;
    if (opt) { bo.ftab_chars = opt->ftab_chars; bo.offrate = opt->offrate; bo.device = opt->device; bo.threads = opt->threads; bo.rbbwt_b = opt->rbbwt_b; bo.verbose = opt->verbose != 0; bo.protein = opt->protein != 0; }
    cfr::BuildReport rep;
    cfr::build_index_files(bi, bo, out_prefix, &rep);
    if (report) {
      report->n = rep.n; report->block_size = rep.block_size; report->first_isa = rep.first_isa;
      report->seconds_sa = rep.seconds_sa; report->seconds_total = rep.seconds_total; report->rounds = rep.rounds; report->pad = 0;
    }
    return CFR_OK;
  });
}

cfr_status cfr_selfcheck_tables(cfr_dev_index *d, uint64_t out[6]) {
  if (!d || !out) return bad_arg("cfr_selfcheck_tables: null argument");
  CFR_ENTER(d, "cfr_selfcheck_tables");
  return guarded([&]() -> cfr_status { d->d->selfcheck(out); return CFR_OK; });
}

cfr_status cfr_search_batch(cfr_dev_index *d, const uint8_t *bases1, const uint64_t *offsets1, const uint8_t *bases2,
                            const uint64_t *offsets2, size_t n, cfr_hit *out_hits, size_t hit_cap, uint64_t *hit_begin) {
  if (!d || !hit_begin || (n && (!bases1 || !offsets1))) return bad_arg("cfr_search_batch: null argument");
  if ((bases2 == nullptr) != (offsets2 == nullptr)) return bad_arg("cfr_search_batch: bases2/offsets2 must both be given");
  CFR_ENTER(d, "cfr_search_batch");
  return guarded([&]() -> cfr_status {
    cfr::DeviceIndex::BatchOut out;
    d->d->run_batch_host(bases1, offsets1, bases2, offsets2, n, false, out);
    memcpy(hit_begin, out.hit_begin.data(), (n + 1) * 8);
    if (out.hits.size() > hit_cap) { g_err = "cfr_search_batch: hit buffer too small"; return CFR_ERR_CAPACITY; }
    if (!out.hits.empty()) memcpy(out_hits, out.hits.data(), out.hits.size() * sizeof(cfr_hit));
    return CFR_OK;
  });
}

cfr_status cfr_classify_batch(cfr_dev_index *d, const uint8_t *bases1, const uint64_t *offsets1, const uint8_t *bases2,
                              const uint64_t *offsets2, size_t n, cfr_result *results, cfr_match *matches, size_t match_cap,
                              size_t *n_matches) {
  if (!d || (n && (!bases1 || !offsets1 || !results))) return bad_arg("cfr_classify_batch: null argument");
  if ((bases2 == nullptr) != (offsets2 == nullptr)) return bad_arg("cfr_classify_batch: bases2/offsets2 must both be given");
  CFR_ENTER(d, "cfr_classify_batch");
  return guarded([&]() -> cfr_status {
    d->d->classify_host(bases1, offsets1, bases2, offsets2, n, results, matches, match_cap, n_matches);
    return CFR_OK;
  });
}

// --expand-taxid: the records the tail appended (slot, count, ids ...) in whatever order its atomics fell, made into spans over an id
// array in match-slot order.  A sub-batch that ran twice (scratch pool too small the first time) left two records for a slot, with
// the same ids: the last one counts.
static cfr_status expanded_out(const std::vector<uint64_t> &raw, cfr_span *spans, size_t match_cap, uint64_t *ids, size_t ids_cap, size_t *n_ids) {
  for (size_t m = 0; m < match_cap; ++m) spans[m] = cfr_span{0, 0};
  std::vector<std::pair<uint64_t, size_t>> recs;             // (slot, position of the record)
  for (size_t at = 0; at + 2 <= raw.size(); at += 2 + (size_t)raw[at + 1]) recs.emplace_back(raw[at], at);
  std::stable_sort(recs.begin(), recs.end(), [](const std::pair<uint64_t, size_t> &a, const std::pair<uint64_t, size_t> &b) { return a.first < b.first; });
  size_t need = 0;
  for (size_t j = 0; j < recs.size(); ++j) if (j + 1 == recs.size() || recs[j + 1].first != recs[j].first) need += (size_t)raw[recs[j].second + 1];
  if (n_ids) *n_ids = need;
  if (need > ids_cap) { g_err = "cfr_classify_batch_expanded: id buffer too small"; return CFR_ERR_CAPACITY; }
  size_t out = 0;
  for (size_t j = 0; j < recs.size(); ++j) {
    if (j + 1 < recs.size() && recs[j + 1].first == recs[j].first) continue;
    const size_t at = recs[j].second, cnt = (size_t)raw[at + 1];
    if (recs[j].first >= match_cap) { g_err = "cfr_classify_batch_expanded: a list for a match slot outside the match buffer"; return CFR_ERR_HIP; }
    spans[recs[j].first] = cfr_span{out, cnt};
    memcpy(ids + out, raw.data() + at + 2, cnt * 8);
    out += cnt;
  }
  return CFR_OK;
}

cfr_status cfr_classify_batch_expanded(cfr_dev_index *d, const uint8_t *bases1, const uint64_t *offsets1, const uint8_t *bases2,
                                       const uint64_t *offsets2, size_t n, cfr_result *results, cfr_match *matches, cfr_span *spans,
                                       size_t match_cap, size_t *n_matches, uint64_t *ids, size_t ids_cap, size_t *n_ids) {
  if (!d || (n && (!bases1 || !offsets1 || !results)) || (match_cap && !spans) || (ids_cap && !ids)) return bad_arg("cfr_classify_batch_expanded: null argument");
  if ((bases2 == nullptr) != (offsets2 == nullptr)) return bad_arg("cfr_classify_batch_expanded: bases2/offsets2 must both be given");
  if (!d->d->host().params.output_expanded) return bad_arg("cfr_classify_batch_expanded: the index was opened without cfr_params.output_expanded");
  CFR_ENTER(d, "cfr_classify_batch_expanded");
  return guarded([&]() -> cfr_status {
    d->d->classify_host(bases1, offsets1, bases2, offsets2, n, results, matches, match_cap, n_matches);
    return expanded_out(d->d->expanded_raw_, spans, match_cap, ids, ids_cap, n_ids);
  });
}

// cfr_pack_reads: the packed form of a read buffer (what k_pack_reads makes on the device), on host threads
cfr_status cfr_pack_reads(const uint8_t *bases, uint64_t total, int threads, uint64_t *packed) {
  if ((total && !bases) || !packed) return bad_arg("cfr_pack_reads: null argument");
  const uint64_t nblk = (total + 15) / 16;
  if (threads < 1) threads = 1;
  auto conv4 = [](uint32_t x, uint32_t &code8, uint32_t &valid4) {        // 4 ASCII bytes -> 4 two-bit codes + 4 validity bits (the device's conv4)
    uint32_t k = (x >> 1) & 0x03030303u;
    k ^= (k >> 1) & 0x01010101u;
    const uint32_t b0 = k & 0x01010101u, b1 = (k >> 1) & 0x01010101u;
    const uint32_t expect = 0x41414141u + b0 * 2u + b1 * 6u + (b0 & b1) * 11u;
    const uint32_t d = x ^ expect;
    const uint32_t nz = (((d & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d) & 0x80808080u;
    const uint32_t ok = ~nz & 0x80808080u;
    valid4 = ((ok >> 7) | (ok >> 14) | (ok >> 21) | (ok >> 28)) & 0xfu;
    code8 = (k | (k >> 6) | (k >> 12) | (k >> 18)) & 0xffu;
  };
  auto work = [&](int tid) {
    const uint64_t lo = nblk * (uint64_t)tid / (uint64_t)threads, hi = nblk * (uint64_t)(tid + 1) / (uint64_t)threads;
    for (uint64_t b = lo; b < hi; ++b) {
      uint32_t w[4] = {0, 0, 0, 0};
      const uint64_t a = b << 4;
      if (a + 16 <= total) memcpy(w, bases + a, 16);
      else memcpy(w, bases + a, (size_t)(total - a));               // bytes past the end: zero = not a symbol
      uint32_t c = 0, vv = 0;
      for (int q = 0; q < 4; ++q) { uint32_t c8, v4; conv4(w[q], c8, v4); c |= c8 << (8 * q); vv |= v4 << (4 * q); }
      packed[b] = (uint64_t)c | ((uint64_t)vv << 32);
    }
  };
  if (threads == 1 || nblk < 4096) { threads = 1; work(0); }
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  return CFR_OK;
}

cfr_status cfr_classify_batch_packed(cfr_dev_index *d, const uint64_t *packed1, const uint64_t *offsets1, const uint64_t *packed2,
                                     const uint64_t *offsets2, size_t n, cfr_result *results, cfr_match *matches, size_t match_cap,
                                     size_t *n_matches) {
  if (!d || (n && (!packed1 || !offsets1 || !results))) return bad_arg("cfr_classify_batch_packed: null argument");
  if ((packed2 == nullptr) != (offsets2 == nullptr)) return bad_arg("cfr_classify_batch_packed: packed2/offsets2 must both be given");
  if (d->d->host().prot.enabled) return bad_arg("cfr_classify_batch_packed: a protein index needs the characters themselves (DnaToAa tells non-symbols apart): use cfr_classify_batch");
  CFR_ENTER(d, "cfr_classify_batch_packed");
  return guarded([&]() -> cfr_status {
    d->d->classify_host_packed(packed1, offsets1, packed2, offsets2, n, results, matches, match_cap, n_matches);
    return CFR_OK;
  });
}

cfr_status cfr_classify_batch_submit(cfr_dev_index *d, const uint8_t *bases1, const uint64_t *offsets1, const uint8_t *bases2,
                                     const uint64_t *offsets2, size_t n, cfr_result *results, cfr_match *matches, size_t match_cap,
                                     cfr_ticket *ticket) {
  if (!d || !ticket || (n && (!bases1 || !offsets1 || !results))) return bad_arg("cfr_classify_batch_submit: null argument");
  if ((bases2 == nullptr) != (offsets2 == nullptr)) return bad_arg("cfr_classify_batch_submit: bases2/offsets2 must both be given");
  std::lock_guard<std::mutex> lk(d->qmu);
  if (d->live.size() >= CFR_MAX_PENDING) { g_err = "cfr_classify_batch_submit: CFR_MAX_PENDING batches are outstanding on this cfr_dev_index"; return CFR_ERR_BUSY; }
  if (!d->worker_started) { d->worker = std::thread(worker_loop, d); d->worker_started = true; }
  cfr_async_job job;
  job.ticket = d->next_ticket++;
  job.run = [=](std::string &err, size_t &nm) -> cfr_status {
    const cfr_status st = guarded([&]() -> cfr_status {
      d->d->classify_host(bases1, offsets1, bases2, offsets2, n, results, matches, match_cap, &nm);
      return CFR_OK;
    });
    if (st != CFR_OK) err = g_err;                       // (the worker's thread-local message travels with the ticket)
    return st;
  };
  *ticket = job.ticket;
  d->live.insert(job.ticket);
  d->queue.push_back(std::move(job));
  d->qcv.notify_one();
  return CFR_OK;
}
cfr_status cfr_classify_batch_wait(cfr_dev_index *d, cfr_ticket ticket, size_t *n_matches) {
  if (!d) return bad_arg("cfr_classify_batch_wait: null argument");
  std::unique_lock<std::mutex> lk(d->qmu);
  if (!d->live.count(ticket)) { g_err = "cfr_classify_batch_wait: unknown ticket, or one that has been waited for already"; return CFR_ERR_ARG; }
  d->live.erase(ticket);                                 // (a second waiter for the same ticket gets the error above)
  d->dcv.wait(lk, [&] { return d->done.count(ticket) != 0; });
  cfr_async_done r = std::move(d->done[ticket]);
  d->done.erase(ticket);
  lk.unlock();
  if (n_matches) *n_matches = r.n_matches;
  if (r.status != CFR_OK) g_err = r.err;
  return r.status;
}

cfr_status cfr_classify_batch_resident(cfr_dev_index *d, const void *d_bases1, const void *d_offsets1, const void *d_bases2,
                                       const void *d_offsets2, size_t n, uint64_t total_bases1, uint64_t total_bases2,
                                       cfr_result *results, cfr_match *matches, size_t match_cap, size_t *n_matches) {
  if (!d || (n && (!d_bases1 || !d_offsets1 || !results))) return bad_arg("cfr_classify_batch_resident: null argument");
  if ((d_bases2 == nullptr) != (d_offsets2 == nullptr)) return bad_arg("cfr_classify_batch_resident: mate buffers must both be given");
  CFR_ENTER(d, "cfr_classify_batch_resident");
  return guarded([&]() -> cfr_status {
    d->d->classify_device((const uint8_t *)d_bases1, (const uint64_t *)d_offsets1, (const uint8_t *)d_bases2,
                          (const uint64_t *)d_offsets2, n, total_bases1, total_bases2, results, matches, match_cap, n_matches);
    return CFR_OK;
  });
}

cfr_status cfr_classify_batch_resident_compact(cfr_dev_index *d, const void *d_bases1, const void *d_offsets1, const void *d_bases2,
                                               const void *d_offsets2, size_t n, uint64_t total_bases1, uint64_t total_bases2,
                                               cfr_result_compact *results, cfr_match_compact *matches, size_t match_cap, size_t *n_matches) {
  if (!d || (n && (!d_bases1 || !d_offsets1 || !results || !matches))) return bad_arg("cfr_classify_batch_resident_compact: null argument");
  if ((d_bases2 == nullptr) != (d_offsets2 == nullptr)) return bad_arg("cfr_classify_batch_resident_compact: mate buffers must both be given");
  CFR_ENTER(d, "cfr_classify_batch_resident_compact");
  return guarded([&]() -> cfr_status {
    d->d->classify_device((const uint8_t *)d_bases1, (const uint64_t *)d_offsets1, (const uint8_t *)d_bases2,
                          (const uint64_t *)d_offsets2, n, total_bases1, total_bases2, reinterpret_cast<cfr_result *>(results),
                          reinterpret_cast<cfr_match *>(matches), match_cap, n_matches, nullptr, /*compact=*/true);
    return CFR_OK;
  });
}

cfr_status cfr_compact_wide_reads(cfr_dev_index *d, size_t *n, const uint32_t **read_index, const cfr_result **results, const cfr_match **matches) {
  if (!d || !n) return bad_arg("cfr_compact_wide_reads: null argument");
  CFR_ENTER(d, "cfr_compact_wide_reads");
  const cfr::DeviceIndex &D = *d->d;
  *n = (size_t)D.wide_total_;
  if (read_index) *read_index = D.wide_idx_.data();
  if (results) *results = D.wide_res_.data();
  if (matches) *matches = D.wide_match_.data();
  if (D.wide_total_ > D.wide_idx_.size()) { g_err = "cfr_compact_wide_reads: more flagged reads than the side list holds"; return CFR_ERR_CAPACITY; }
  return CFR_OK;
}

void *cfr_host_alloc(size_t bytes) { return cfr::host_alloc_pinned(bytes); }
void cfr_host_free(void *p) { cfr::host_free_pinned(p); }

cfr_status cfr_classify_from_hits(const cfr_index *idx, const cfr_hit *hits, const uint64_t *hit_begin, const uint64_t *row_begin,
                                  const uint64_t *row_vals, const int32_t *query_len, size_t n, int threads, cfr_result *results,
                                  cfr_match *matches, size_t match_cap, size_t *n_matches) {
  if (!idx || !hit_begin || (n && (!results || !query_len))) return bad_arg("cfr_classify_from_hits: null argument");
  return guarded([&]() -> cfr_status {
    cfr::DeviceIndex::BatchOut out;
    out.hit_begin.assign(hit_begin, hit_begin + n + 1);
    const uint64_t nh = hit_begin[n];
    out.hits.assign(hits, hits + nh);
    out.row_begin.assign(row_begin, row_begin + nh + 1);
    out.row_vals.assign(row_vals, row_vals + row_begin[nh]);
    out.read_len.assign(query_len, query_len + n);
    std::vector<cfr_match> mv;
    cfr::classify_batch_tail(*idx->h, out, n, threads, results, mv);
    if (n_matches) *n_matches = mv.size();
    if (mv.size() > match_cap) { g_err = "cfr_classify_from_hits: match buffer too small"; return CFR_ERR_CAPACITY; }
    if (!mv.empty()) memcpy(matches, mv.data(), mv.size() * sizeof(cfr_match));
    return CFR_OK;
  });
}

cfr_status cfr_classify_from_hits_expanded(const cfr_index *idx, const cfr_hit *hits, const uint64_t *hit_begin, const uint64_t *row_begin,
                                           const uint64_t *row_vals, const int32_t *query_len, size_t n, int threads, cfr_result *results,
                                           cfr_match *matches, cfr_span *spans, size_t match_cap, size_t *n_matches, uint64_t *ids, size_t ids_cap,
                                           size_t *n_ids) {
  if (!idx || !hit_begin || (n && (!results || !query_len)) || (match_cap && !spans) || (ids_cap && !ids)) return bad_arg("cfr_classify_from_hits_expanded: null argument");
  if (!idx->h->params.output_expanded) return bad_arg("cfr_classify_from_hits_expanded: the index was opened without cfr_params.output_expanded");
  return guarded([&]() -> cfr_status {
    cfr::DeviceIndex::BatchOut out;
    out.hit_begin.assign(hit_begin, hit_begin + n + 1);
    const uint64_t nh = hit_begin[n];
    out.hits.assign(hits, hits + nh);
    out.row_begin.assign(row_begin, row_begin + nh + 1);
    out.row_vals.assign(row_vals, row_vals + row_begin[nh]);
    out.read_len.assign(query_len, query_len + n);
    std::vector<cfr_match> mv;
    cfr::ExpandedLists x;
    cfr::classify_batch_tail(*idx->h, out, n, threads, results, mv, &x);
    if (n_matches) *n_matches = mv.size();
    if (n_ids) *n_ids = x.ids.size();
    if (mv.size() > match_cap) { g_err = "cfr_classify_from_hits_expanded: match buffer too small"; return CFR_ERR_CAPACITY; }
    if (x.ids.size() > ids_cap) { g_err = "cfr_classify_from_hits_expanded: id buffer too small"; return CFR_ERR_CAPACITY; }
    if (!mv.empty()) memcpy(matches, mv.data(), mv.size() * sizeof(cfr_match));
    for (size_t m = 0; m < match_cap; ++m) spans[m] = m < x.spans.size() ? x.spans[m] : cfr_span{0, 0};
    if (!x.ids.empty()) memcpy(ids, x.ids.data(), x.ids.size() * 8);
    return CFR_OK;
  });
}

cfr_status cfr_last_batch_stats(const cfr_dev_index *d, cfr_batch_stats *st) {
  if (!d || !st) return bad_arg("cfr_last_batch_stats: null argument");
  *st = d->d->last_stats;
  return CFR_OK;
}

static cfr_status dust_batch(uint8_t *bases, const uint64_t *offsets, size_t n, int threads, void (*mask)(uint8_t *, size_t)) {
  if (n && (!bases || !offsets)) return bad_arg("cfr_dust_mask_batch: null argument");
  if (threads < 1) threads = 1;
  auto work = [&](int tid) {
    // a contiguous slice per thread (reads are independent; the reference strides them, the result is the same)
    const size_t lo = n * (size_t)tid / (size_t)threads, hi = n * (size_t)(tid + 1) / (size_t)threads;
    for (size_t i = lo; i < hi; ++i) mask(bases + offsets[i], offsets[i + 1] - offsets[i]);
  };
  if (threads == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  return CFR_OK;
}
cfr_status cfr_dust_mask_batch(uint8_t *bases, const uint64_t *offsets, size_t n, int threads) {
  return dust_batch(bases, offsets, n, threads, cfr::dust_mask);
}
cfr_status cfr_dust_mask_batch_literal(uint8_t *bases, const uint64_t *offsets, size_t n, int threads) {
  return dust_batch(bases, offsets, n, threads, cfr::dust_mask_literal);
}

cfr_status cfr_device_index_set_dust(cfr_dev_index *d, int on) {
  if (!d) return bad_arg("cfr_device_index_set_dust: null argument");
  CFR_ENTER(d, "cfr_device_index_set_dust");
  d->d->set_dust(on != 0);
  return CFR_OK;
}
cfr_status cfr_dust_mask_device(cfr_dev_index *d, uint8_t *bases, const uint64_t *offsets, size_t n) {
  if (!d || (n && (!bases || !offsets))) return bad_arg("cfr_dust_mask_device: null argument");
  CFR_ENTER(d, "cfr_dust_mask_device");
  return guarded([&]() -> cfr_status {
    d->d->dust_mask_host(bases, offsets, n);
    return CFR_OK;
  });
}

const char *cfr_tsv_header(void) {   // ResultWriter::OutputHeader (ResultWriter.hpp:186-197), no barcode/UMI columns
  return "readID\tseqID\ttaxID\tscore\t2ndBestScore\thitLength\tqueryLength\tnumMatches\n";
}

const char *cfr_tsv_header_expanded(void) {   // ... with _outputExpandedTaxIds (ResultWriter.hpp:194-195)
  return "readID\tseqID\ttaxID\tscore\t2ndBestScore\thitLength\tqueryLength\tnumMatches\texpandedTaxIDs\n";
}

static size_t format_tsv(const cfr_index *idx, const char *read_id, const cfr_result *r, const cfr_match *matches, bool expanded, const cfr_span *spans,
                         const uint64_t *ids, char *buf, size_t cap);
size_t cfr_format_tsv(const cfr_index *idx, const char *read_id, const cfr_result *r, const cfr_match *matches, char *buf, size_t cap) {
  return format_tsv(idx, read_id, r, matches, false, nullptr, nullptr, buf, cap);
}
size_t cfr_format_tsv_expanded(const cfr_index *idx, const char *read_id, const cfr_result *r, const cfr_match *matches, const cfr_span *spans,
                               const uint64_t *ids, char *buf, size_t cap) {
  return format_tsv(idx, read_id, r, matches, true, spans, ids, buf, cap);
}

static size_t format_tsv(const cfr_index *idx, const char *read_id, const cfr_result *r, const cfr_match *matches, bool expanded, const cfr_span *spans,
                         const uint64_t *ids, char *buf, size_t cap) {
  // ResultWriter::Output (ResultWriter.hpp:209-240): "%s\t%s\t%lu\t%lu\t%lu\t%d\t%d\t%d" + PrintExtraCol(expandedTaxIdStrings[i]) / PrintExtraCol("")
  // (:226-227, :239-240) - the same bytes, put together by hand: at a hundred million rows per run snprintf's format parsing was the
  // command line's slowest stage (profiles/r5_cli_timing_100m.txt)
  struct Out {
    char *buf; size_t cap, off;
    void put(const char *p, size_t n) { if (buf && off < cap) memcpy(buf + off, p, n < cap - off ? n : cap - off); off += n; }
    void ch(char c) { if (buf && off < cap) buf[off] = c; ++off; }
    void u64(uint64_t v) { char t[24]; int k = 24; do { t[--k] = (char)('0' + v % 10); v /= 10; } while (v); put(t + k, (size_t)(24 - k)); }
    void i32(int32_t v) { if (v < 0) { ch('-'); u64((uint64_t)(-(int64_t)v)); } else u64((uint64_t)v); }
  } o{buf, cap, 0};
  const cfr::Taxonomy &t = idx->h->tax;
  const size_t idn = strlen(read_id);
  if (r->n_match > 0) {
    for (int i = 0; i < r->n_match; ++i) {
      const cfr_match &m = matches[r->match_begin + (uint64_t)i];
      const char *name;
      if (m.kind == 0) name = m.id < t.seq_name.size() ? t.seq_name[m.id].c_str() : "";
      else name = cfr::tax_rank_string(m.id < t.node_cnt ? t.rank[m.id] : 0);
      o.put(read_id, idn); o.ch('\t'); o.put(name, strlen(name)); o.ch('\t'); o.u64(m.taxid); o.ch('\t'); o.u64(r->score); o.ch('\t');
      o.u64(r->secondary_score); o.ch('\t'); o.i32(r->hit_length); o.ch('\t'); o.i32(r->query_length); o.ch('\t'); o.i32(r->n_match);
      if (expanded) {
        o.ch('\t');
        const cfr_span sp = spans ? spans[r->match_begin + (uint64_t)i] : cfr_span{0, 0};
        for (uint64_t j = 0; j < sp.count; ++j) { if (j) o.ch(','); o.u64(ids[sp.begin + j]); }
      }
      o.ch('\n');
    }
  } else {
    o.put(read_id, idn); { static const char kUn[] = "\tunclassified\t0\t0\t0\t0\t"; o.put(kUn, sizeof(kUn) - 1); } o.i32(r->query_length); o.put("\t1", 2);
    if (expanded) o.ch('\t');
    o.ch('\n');
  }
  if (buf && cap) buf[o.off < cap ? o.off : cap - 1] = '\0';      // always a C string: a truncated call (return value >= cap: retry with that much + 1) ends at cap - 1
  return o.off;
}

}  // extern "C"
