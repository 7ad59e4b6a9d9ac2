/*
 * cfr_hip.h — C-ABI of libcfr_hip.so: the MI355X (gfx950) replacement for Centrifuger's
 * per-read classification hot path.
 *
 * The reference has no FFI layer; the seam this library replaces is the per-batch thread
 * fan-out around Classifier<Sequence_RunBlock>::Query:
 *
 *   reference                                                   this library
 *   ---------------------------------------------------------   ---------------------------
 *   Classifier::Init(idxPrefix, param)   Classifier.hpp:902-947  cfr_index_open + cfr_device_index_create
 *   FMIndex::Load / Taxonomy::Load       FMIndex.hpp:588-606,    (inside cfr_index_open)
 *                                        Taxonomy.hpp:1259-1287
 *   pthread_create(ClassifyReads_Thread) CentrifugerClass.cpp:   cfr_classify_batch
 *     ... Query(r1, r2, result) ...        681-688, 240-340,
 *                                          Classifier.hpp:950-961
 *   SearchForwardAndReverse              Classifier.hpp:509-583  cfr_search_batch  (device)
 *   FMIndex::BackwardToSampledSA         FMIndex.hpp:514-524     cfr_locate_rows   (device)
 *   FMIndex::Rank / Sequence::Access     FMIndex.hpp:352-362,    cfr_rank_batch    (device; parity probe of
 *                                        Sequence.hpp:44-55                          the "operator API" of the path)
 *   FMIndex::BackwardSearch              FMIndex.hpp:487-510     cfr_backward_search_batch (device probe)
 *   ResultWriter::Output                 ResultWriter.hpp:199-242 cfr_format_tsv
 *
 * Conventions: plain pointers and sizes only; the caller owns every host buffer; the library
 * owns device memory.  Every call returns a cfr_status; cfr_last_error() gives the message of
 * the last failing call on this thread.  The library never calls exit().  There is NO CPU
 * fallback: without a usable HIP device cfr_device_index_create fails with CFR_ERR_NO_DEVICE.
 *
 * Reads are passed as one flat ASCII buffer plus n+1 byte offsets (no terminators), already
 * dust-masked if the caller wants dust (the reference masks before Query,
 * CentrifugerClass.cpp:276-316; cfr_dust_mask_batch does the same on the host).
 */
#ifndef CFR_HIP_H
#define CFR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int cfr_status;
enum {
  CFR_OK = 0,
  CFR_ERR_IO = 1,          /* cannot open / short read */
  CFR_ERR_FORMAT = 2,      /* malformed or unsupported .cfr content */
  CFR_ERR_NO_DEVICE = 3,   /* no HIP device / HIP runtime error at setup */
  CFR_ERR_HIP = 4,         /* HIP runtime error during a batch */
  CFR_ERR_ARG = 5,         /* bad argument */
  CFR_ERR_CAPACITY = 6,    /* caller-provided output buffer too small */
  CFR_ERR_BUSY = 7         /* another thread is inside a call on this device index, or too many batches are queued */
};

typedef struct cfr_index cfr_index;           /* host copy of <prefix>.{1,2,4}.cfr */
typedef struct cfr_dev_index cfr_dev_index;   /* flat image of the index in one GPU's HBM */

/* _classifierParam (Classifier.hpp:17-38) */
typedef struct {
  int32_t max_result;               /* -k, default 1 */
  int32_t min_hit_len;              /* --min-hitlen, <=0: infer (Classifier.hpp:113-129) */
  int32_t max_result_per_hit_factor;/* --hitk-factor, default 40 */
  int32_t output_expanded;          /* --expand-taxid (outputExpandedResult, Classifier.hpp:22): keep, for every reported tax id, the ids that
                                       ReduceTaxIds / LCA promoted into it; they are handed out by cfr_classify_batch_expanded */
  uint64_t consider_secondary_hit_len;     /* 2000 */
  double consider_secondary_score_factor;  /* 0.995 */
} cfr_params;

/* _BWTHit (Classifier.hpp:70-85) */
typedef struct {
  uint64_t sp, ep;
  int32_t l, strand, offset, pad;
} cfr_hit;

/* _classifierResult (Classifier.hpp:41-59) as POD.  Matches live in a separate array:
 * read i owns matches [match_begin, match_begin + n_match). */
typedef struct {
  uint64_t score, secondary_score;
  int32_t hit_length, query_length;
  int32_t n_match;            /* 0 = unclassified */
  int32_t pad;
  uint64_t match_begin;
} cfr_result;

typedef struct {
  uint64_t id;       /* kind 0: seqId (name = sequence name); kind 1: compact tax id (name = rank string) */
  uint64_t taxid;    /* ORIGINAL tax id (column 3 of the TSV) */
  int32_t kind, pad;
} cfr_match;

/* index geometry, for tests and sizing */
typedef struct {
  uint64_t n;                  /* BWT length */
  uint64_t first_isa;
  uint64_t block_size;         /* run-block b (== n when built with --rbbwt-b 1) */
  uint64_t precompute_width;   /* ftab chars */
  uint64_t sample_rate;
  uint64_t selected_cnt;
  uint64_t seq_cnt, node_cnt;
  int32_t min_hit_len;         /* after inference */
  char last_chr;
  uint8_t is_protein;          /* 1: amino-acid index (.4.cfr sequence_type amino_acid): reads are searched translated, never dust-masked */
  char pad[2];
  uint64_t device_bytes;       /* 0 for a host index */
} cfr_index_info;

/* kernel timing of the last batch call on a device index (HIP events on the library's stream) */
typedef struct {
  float pack_ms, search_ms, adjust_ms, rows_ms, locate_ms, tail_ms, total_ms;
  uint64_t n_chains, n_hits, n_rows;   /* n_hits / n_rows stay 0 on the paths that never need those totals on the host (fused tail; one-launch post stage) */
} cfr_batch_stats;

void cfr_params_default(cfr_params *p);
const char *cfr_last_error(void);
const char *cfr_version(void);

/* ---- index lifetime ---- */
cfr_status cfr_index_open(const char *idx_prefix, const cfr_params *params, cfr_index **out);
void cfr_index_destroy(cfr_index *idx);
cfr_status cfr_index_get_info(const cfr_index *idx, cfr_index_info *info);
/* A 64-bit digest (FNV-1a) of everything cfr_index_open parsed - scalars, bit strings, tables, taxonomy, inferred parameters.
 * Two opens of the same files with the same parameters give the same value (what the open/destroy stress test asserts); no
 * reference counterpart. */
cfr_status cfr_index_digest(const cfr_index *idx, uint64_t *digest);
/* How many bytes of the index's bit strings this handle reads straight from the read-only mapping of the .1.cfr file (the ranks of a node
 * share those pages through the page cache) and how many it holds as private copies.  A nucleotide index opened normally maps all of
 * them; a protein index is decoded into the process and maps none.  No reference counterpart (FMIndex::Load reads everything,
 * FMIndex.hpp:588-606). */
cfr_status cfr_index_mapped_bytes(const cfr_index *idx, uint64_t *mapped, uint64_t *copied);

cfr_status cfr_device_count(int *count);
cfr_status cfr_device_index_create(const cfr_index *idx, int device, cfr_dev_index **out);

/* What the device image is built with.  Nothing here changes a result; it trades load time and HBM for throughput.
 * cfr_device_index_create == cfr_device_index_create_ex with the defaults.  (The CFR_* environment variables listed in
 * profiles/HISTORY.md section 5 override these fields when set: they exist for A/B runs and for the variant tests.) */
typedef enum { CFR_PROFILE_THROUGHPUT = 0, CFR_PROFILE_FAST_LOAD = 1, CFR_PROFILE_BALANCED = 2 } cfr_profile;
typedef struct {
  int32_t profile;        /* cfr_profile.  FAST_LOAD: K-mer table of at most 4^13 entries, no text-mode tables, no locate memo
                             (shortest load).  BALANCED: the 4^13 table WITH the text-mode tables and the locate memo (no 68 GB
                             allocation: what the command line uses).  THROUGHPUT: everything, K up to 17 */
  int32_t ftabx_width;    /* K of the derived K-mer table; -1 = automatic (17 with 8-byte entries - 137 GB - for indexes of 2.7e8 symbols and more when it fits, else log4(n)+2, at most 16), 0 = none */
  int32_t text_mode;      /* derived SA / ISA / 2-bit text (n < 2^32): -1 = by profile, 0 = off, 1 = on */
  int32_t run_block_layout; /* 1 = keep the run-block components compressed in HBM instead of the flat occurrence image */
  double loc_memo_gb;     /* byte budget of the locate memo in GB; negative = by profile (16 / none), 0 = none */
  uint64_t sub_batch;     /* reads per sub-batch of one batch call; 0 = default */
} cfr_device_options;
void cfr_device_options_default(cfr_device_options *o);
cfr_status cfr_device_index_create_ex(const cfr_index *idx, int device, const cfr_device_options *options, cfr_dev_index **out);
void cfr_device_index_destroy(cfr_dev_index *d);
cfr_status cfr_device_index_get_info(const cfr_dev_index *d, cfr_index_info *info);

/* ---- primitive probes (device): used by the parity tests of L0-L2 ---- */
/* FMIndex::Rank(c, pos, inclusive) and Sequence::Access(pos) for n queries.
 * chars[i] in "ACGT"; out_rank may be NULL, out_access may be NULL. */
cfr_status cfr_rank_batch(cfr_dev_index *d, const char *chars, const uint64_t *pos, const uint8_t *inclusive,
                          size_t n, uint64_t *out_rank, char *out_access);
/* FMIndex::BackwardSearch(s = read[0..m), m): out_l / out_sp / out_ep (sp,ep untouched -> value 0 when l==0 path) */
cfr_status cfr_backward_search_batch(cfr_dev_index *d, const uint8_t *bases, const uint64_t *offsets,
                                     const uint32_t *m, size_t n, uint64_t *out_l, uint64_t *out_sp, uint64_t *out_ep);
/* FMIndex::BackwardToSampledSA(row): value (a sequence id) and LF-step count */
cfr_status cfr_locate_rows(cfr_dev_index *d, const uint64_t *rows, size_t n, uint64_t *out_val, uint32_t *out_steps);

/* Self-check of the tables the device image derives at load time, against the BWT itself (no reference counterpart: the
 * reference has no derived tables).  out[0..3] = number of rows where the suffix array (SA[i] < n, 0 exactly at the row of text
 * position 0), the text (symbol left of SA[i] = B[i]), the LF order (SA[LF(i)] = SA[i] - 1) and the direct locate (memo, or
 * suffix array + step function; every 61st row, against the plain FMIndex::BackwardToSampledSA walk) disagree - all 0 on a sound
 * image; out[4] = 1 if the text-mode tables exist, out[5] = 0 without a locate memo, else 1 + log2 of its row rate. */
cfr_status cfr_selfcheck_tables(cfr_dev_index *d, uint64_t out[6]);

/* ---- the path ---- */
/* SearchForwardAndReverse for n reads (mates optional: bases2/offsets2 == NULL for single-end).
 * hits of read i are out_hits[hit_begin[i] .. hit_begin[i+1]) ; hit_begin has n+1 entries.
 * out_hits capacity in elements = hit_cap ; CFR_ERR_CAPACITY if too small (needed size in hit_begin[n]). */
cfr_status cfr_search_batch(cfr_dev_index *d, const uint8_t *bases1, const uint64_t *offsets1,
                            const uint8_t *bases2, const uint64_t *offsets2, size_t n,
                            cfr_hit *out_hits, size_t hit_cap, uint64_t *hit_begin);

/* Query for n reads, entirely on the device (search, locate, scoring, LCA / tax-id reduction).
 * results: n entries.  Matches of read i are matches[results[i].match_begin .. +n_match):
 *   max_result  > 0: match_begin = i * max_result (slots a read does not use are left untouched);
 *   max_result <= 0: match_begin = the read's offset in the located-row space.
 * *n_matches receives the number of match SLOTS the call needs (the extent of the matches array);
 * CFR_ERR_CAPACITY if match_cap is smaller.  Host buffers in, host buffers out (any host memory works;
 * memory from cfr_host_alloc is pinned and transfers at PCIe rate). */
cfr_status cfr_classify_batch(cfr_dev_index *d, const uint8_t *bases1, const uint64_t *offsets1,
                              const uint8_t *bases2, const uint64_t *offsets2, size_t n,
                              cfr_result *results, cfr_match *matches, size_t match_cap, size_t *n_matches);

/* The same call with the bases in PACKED form: block b (one uint64_t) holds the 16 characters [16 b, 16 b + 16) of the flat buffer
 * the offsets refer to - character j of the block as a 2-bit code (A 0, C 1, G 2, T 3) at bits 2j..2j+1, and bit 32 + j set when
 * the character is one of the upper-case letters A, C, G, T (anything else - N, lower case, IUPAC codes, bytes past the end of the
 * buffer - has the bit clear and code bits that do not matter).  (total + 15) / 16 blocks per buffer.  Half the bytes of the ASCII
 * form cross PCIe (the ASCII entry is bound by its 1 byte per base: 3.2e8 reads/s of 150 bp from pinned memory); results are those of
 * cfr_classify_batch on the same reads, SDUST on the device included when it is switched on - nothing on the path tells two
 * non-symbols apart (a non-symbol ends a match, FMIndex.hpp:396-401; is SDUST's fifth code; reverse-complements to 'N',
 * Classifier.hpp:846-856).  Nucleotide indexes only: the translated search of a protein index does tell them apart (DnaToAa,
 * Classifier.hpp:131-241), so the call returns CFR_ERR_ARG there.  cfr_pack_reads makes the blocks from an ASCII buffer on `threads` host threads (a parser can also
 * emit them directly).  The reference reads ASCII through kseq (ReadFiles.hpp:337); there is no packed form there. */
cfr_status cfr_pack_reads(const uint8_t *bases, uint64_t total, int threads, uint64_t *packed);
cfr_status cfr_classify_batch_packed(cfr_dev_index *d, const uint64_t *packed1, const uint64_t *offsets1,
                                     const uint64_t *packed2, const uint64_t *offsets2, size_t n,
                                     cfr_result *results, cfr_match *matches, size_t match_cap, size_t *n_matches);

/* cfr_classify_batch with the lists of `--expand-taxid` (Classifier.hpp:792-838; Taxonomy::ReduceTaxIds / LCA with
 * promotedChildTaxIds, Taxonomy.hpp:733-973): for every match slot m of `matches` the ORIGINAL tax ids (GetOrigTaxId) of the
 * children that were promoted into matches[m] are ids[spans[m].begin .. spans[m].begin + spans[m].count), in the order the
 * reference prints them; count 0 = the empty string (every sequence-level match, Classifier.hpp:792-795).  The index must have
 * been opened with cfr_params.output_expanded = 1 (CFR_ERR_ARG otherwise).  spans: match_cap entries, every one is written.
 * *n_ids = ids used; CFR_ERR_CAPACITY (and *n_ids = the number needed) when ids_cap is too small.  SDUST on the device and the
 * streamed upload work as in cfr_classify_batch. */
typedef struct { uint64_t begin, count; } cfr_span;
cfr_status cfr_classify_batch_expanded(cfr_dev_index *d, const uint8_t *bases1, const uint64_t *offsets1,
                                       const uint8_t *bases2, const uint64_t *offsets2, size_t n,
                                       cfr_result *results, cfr_match *matches, cfr_span *spans, size_t match_cap, size_t *n_matches,
                                       uint64_t *ids, size_t ids_cap, size_t *n_ids);

/* Asynchronous form of cfr_classify_batch.  The reference overlaps reading, classification and output of consecutive batches
 * (CentrifugerClass.cpp:776-800: the next batch is read while the threads classify this one); a caller of this library gets the
 * same by submitting batch k+1 before it waits for batch k:
 *   cfr_classify_batch_submit  queues the call and returns at once with a ticket.  Every buffer handed over (reads, offsets,
 *                              results, matches) must stay valid and untouched until the ticket has been waited for.
 *   cfr_classify_batch_wait    blocks until that batch is done, returns ITS status (cfr_last_error() then holds its message)
 *                              and, through n_matches, what cfr_classify_batch would have stored there.  A ticket is waited
 *                              for exactly once.
 * Batches of one cfr_dev_index run in submission order on the index's own worker thread; at most CFR_MAX_PENDING may be
 * outstanding (CFR_ERR_BUSY beyond that).  cfr_device_index_destroy waits for queued batches first.
 * One cfr_dev_index serves one call at a time: a synchronous entry (classify / search / probes / dust) called while another
 * thread - or a queued batch - is inside the same cfr_dev_index returns CFR_ERR_BUSY instead of racing on its buffers. */
#define CFR_MAX_PENDING 8
typedef uint64_t cfr_ticket;
cfr_status cfr_classify_batch_submit(cfr_dev_index *d, const uint8_t *bases1, const uint64_t *offsets1,
                                     const uint8_t *bases2, const uint64_t *offsets2, size_t n,
                                     cfr_result *results, cfr_match *matches, size_t match_cap, cfr_ticket *ticket);
cfr_status cfr_classify_batch_wait(cfr_dev_index *d, cfr_ticket ticket, size_t *n_matches);

/* Same, with the read buffers ALREADY RESIDENT in this device's HBM (device pointers; the caller has
 * synchronised whatever produced them).  This is the entry bench.py times. */
cfr_status cfr_classify_batch_resident(cfr_dev_index *d, const void *d_bases1, const void *d_offsets1,
                                       const void *d_bases2, const void *d_offsets2, size_t n,
                                       uint64_t total_bases1, uint64_t total_bases2,
                                       cfr_result *results, cfr_match *matches, size_t match_cap, size_t *n_matches);

/* The same call with the results in a narrow layout: 20 + 12 bytes per read and match slot instead of 40 + 24.  At several
 * hundred million reads per second the step is co-limited by the device-to-host copy of its results (64 bytes per single-end
 * read, 160 per pair with -k 5); the fields are the same, in the widths they need for reads below 64 k bases.
 *   max_result > 0 only; the match slots of read i are [i * max_result, i * max_result + n_match) (no match_begin field).
 *   The match slots a read does not use are zero.
 *   A read one of whose values does not fit (score >= 2^32, sequence id >= 2^31 ...) has CFR_COMPACT_WIDE set in `flags`: its
 *   other fields are not meaningful; cfr_compact_wide_reads hands out such reads in the wide layout (no second pass over the batch). */
#define CFR_COMPACT_WIDE 1u
typedef struct {
  uint32_t score, secondary_score;
  uint32_t hit_length, query_length;
  uint8_t n_match, flags;
  uint16_t pad;
} cfr_result_compact;                 /* 20 bytes */
typedef struct {
  uint32_t id_kind;                   /* bit 31: kind (cfr_match.kind), bits 0..30: id */
  uint32_t taxid_lo, taxid_hi;        /* ORIGINAL tax id */
} cfr_match_compact;                  /* 12 bytes */
cfr_status cfr_classify_batch_resident_compact(cfr_dev_index *d, const void *d_bases1, const void *d_offsets1,
                                               const void *d_bases2, const void *d_offsets2, size_t n,
                                               uint64_t total_bases1, uint64_t total_bases2,
                                               cfr_result_compact *results, cfr_match_compact *matches, size_t match_cap, size_t *n_matches);
/* The reads of the LAST cfr_classify_batch_resident_compact call on d that carry CFR_COMPACT_WIDE, in the wide layout: *n of them,
 * read_index[j] = which read of the batch, results[j] = its cfr_result with match_begin pointing into matches[] (max_result slots
 * per entry).  The arrays belong to d and stay valid until its next classify call.  CFR_ERR_CAPACITY (and *n = their number)
 * if the batch held more than 65536 such reads: take cfr_classify_batch_resident for it. */
cfr_status cfr_compact_wide_reads(cfr_dev_index *d, size_t *n, const uint32_t **read_index, const cfr_result **results, const cfr_match **matches);

/* pinned host memory for result buffers (hipHostMalloc); NULL on failure */
void *cfr_host_alloc(size_t bytes);
void cfr_host_free(void *p);

cfr_status cfr_last_batch_stats(const cfr_dev_index *d, cfr_batch_stats *st);

/* Host-only tail of Query: GetClassificationFromHits (Classifier.hpp:585-843) once the device has
 * produced the hits and the located sequence ids.  hits of read i: [hit_begin[i], hit_begin[i+1]);
 * located ids of hit h: row_vals[row_begin[h] .. row_begin[h+1]).  query_len[i] = strlen(r1)+strlen(r2). */
cfr_status cfr_classify_from_hits(const cfr_index *idx, const cfr_hit *hits, const uint64_t *hit_begin,
                                  const uint64_t *row_begin, const uint64_t *row_vals, const int32_t *query_len,
                                  size_t n, int threads, cfr_result *results, cfr_match *matches, size_t match_cap,
                                  size_t *n_matches);

/* cfr_classify_from_hits with the lists of --expand-taxid (see cfr_classify_batch_expanded; idx opened with output_expanded = 1) */
cfr_status cfr_classify_from_hits_expanded(const cfr_index *idx, const cfr_hit *hits, const uint64_t *hit_begin,
                                           const uint64_t *row_begin, const uint64_t *row_vals, const int32_t *query_len,
                                           size_t n, int threads, cfr_result *results, cfr_match *matches, cfr_span *spans, size_t match_cap,
                                           size_t *n_matches, uint64_t *ids, size_t ids_cap, size_t *n_ids);

/* ---- host helpers around the path ---- */
/* SDUST pre-step (Dustmasker.hpp:357-421 + CentrifugerClass.cpp:283-289): masked bases -> 'N', in place */
cfr_status cfr_dust_mask_batch(uint8_t *bases, const uint64_t *offsets, size_t n, int threads);
/* the same masks computed with the reference's data structure as it is (a list of perfect intervals that reaches 1711 entries on a
 * homopolymer and is rescanned per window suffix: milliseconds per poly-A read); the anchor the two fast forms are tested against */
cfr_status cfr_dust_mask_batch_literal(uint8_t *bases, const uint64_t *offsets, size_t n, int threads);

/* The same masking on the device.  cfr_device_index_set_dust(d, 1): every following cfr_classify_batch /
 * cfr_classify_batch_resident call on d masks its reads in HBM before searching them (the caller's buffers are not modified),
 * so unmasked reads can be handed over and the host pre-step disappears; results equal cfr_dust_mask_batch + classify.
 * cfr_dust_mask_device: the device scan alone, host buffer in and out (parity probe). */
cfr_status cfr_device_index_set_dust(cfr_dev_index *d, int on);
cfr_status cfr_dust_mask_device(cfr_dev_index *d, uint8_t *bases, const uint64_t *offsets, size_t n);

/* ResultWriter::Output rows for one read (ResultWriter.hpp:209-240).  Returns bytes needed (without the NUL); writes at most cap and always
 * NUL-terminates a non-empty buffer: a return value >= cap means the text was cut at cap - 1 and the caller retries with value + 1. */
size_t cfr_format_tsv(const cfr_index *idx, const char *read_id, const cfr_result *r, const cfr_match *matches,
                      char *buf, size_t cap);
const char *cfr_tsv_header(void);
/* the same rows with the expandedTaxIDs column of --expand-taxid (ResultWriter.hpp:194-195, 226-227, 239-240): spans / ids as
 * cfr_classify_batch_expanded filled them */
size_t cfr_format_tsv_expanded(const cfr_index *idx, const char *read_id, const cfr_result *r, const cfr_match *matches,
                               const cfr_span *spans, const uint64_t *ids, char *buf, size_t cap);
const char *cfr_tsv_header_expanded(void);

/* ---- index writer (outside the classification path) ----
 * What `centrifuger-build` produces (Builder::Build + FMBuilder, Builder.hpp:86-313, compactds/FMBuilder.hpp:209-313):
 * <out_prefix>.{1,2,3,4}.cfr for nucleotide sequences, default layout options (--rbbwt-b / --offrate / --ftabchars
 * honoured).  The suffix array is built in the HBM of one MI355X (texts below 2^34 symbols); there is no CPU path.
 * bin/centrifuger-build is the command line on top of it. */
typedef struct {
  uint64_t n_seqs;
  const char *const *seq_names;     /* conversion-table order = sequence ids */
  const uint64_t *seq_taxids;       /* original tax id of every sequence */
  const uint64_t *seq_lens;         /* length of every sequence */
  const uint8_t *text;              /* the sequences back to back: upper-case A,C,G,T only (SequenceCompactor.hpp:59-84 drops the rest) */
  uint64_t n_nodes;                 /* nodes.dmp */
  const uint64_t *node_taxid, *node_parent;
  const char *const *node_rank;
  uint64_t n_names;                 /* names.dmp, scientific names */
  const uint64_t *name_taxid;
  const char *const *name_text;
  /* Genomes of the text when they are not simply "every sequence, in conversion-table order" (Builder.hpp:108-165: the text
   * follows the FASTA order, a conversion table may name sequences the FASTA does not hold, and a FASTA sequence the table
   * does not name is added as an extra name without a tax id).  n_genomes = 0: genome g = sequence g with seq_lens[g]. */
  uint64_t n_genomes;
  const uint64_t *genome_seq;       /* sequence id (index into seq_names) of every genome of the text, in text order; no id twice */
  const uint64_t *genome_lens;      /* their lengths; seq_lens is not read when n_genomes > 0 */
  uint64_t n_extra;                 /* the LAST n_extra entries of seq_names are extra names (Taxonomy::AddExtraSeqName): seq_taxids not read for them */
  /* further tax ids whose lineages stay in the tree: Taxonomy::Init keeps every id the conversion table MENTIONS
   * (Taxonomy.hpp:275-300), also one that no sequence ends up with (a name listed twice gets the LCA of its ids) */
  uint64_t n_present_taxids;
  const uint64_t *present_taxids;
} cfr_build_input;
typedef struct {
  int32_t ftab_chars;   /* --ftabchars, default 10 */
  int32_t offrate;      /* --offrate, default 4: SA sampled every 2^offrate rows */
  int32_t device;       /* HIP device ordinal */
  int32_t threads;      /* host threads of the compression / writing half, 0 = automatic */
  uint64_t rbbwt_b;     /* --rbbwt-b, 0 = automatic block size */
  int32_t verbose;
  int32_t protein;      /* --protein (CentrifugerBuild.cpp:221-227, Builder.hpp:95-101): `text` holds the proteins back to back, letters of
                         * "ARNDCEQGHILKMFPSTWYV" only, lengths in residues; the writer closes every protein with '$' and writes an
                         * FMIndex<Sequence_RunBlockOneTree> with endMarkerSA.  ftab_chars <= 6 (the reference's default is 4), n < 2^32 */
} cfr_build_options;
typedef struct {
  uint64_t n, block_size, first_isa;
  double seconds_sa, seconds_total;
  int32_t rounds, pad;
} cfr_build_report;
void cfr_build_options_default(cfr_build_options *o);
cfr_status cfr_build_index(const cfr_build_input *in, const cfr_build_options *opt, const char *out_prefix, cfr_build_report *report);

#ifdef __cplusplus
}
#endif
#endif /* CFR_HIP_H */
