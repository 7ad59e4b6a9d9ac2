/*
 * cfr_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's per-read classification path
 * (mourisl/centrifuger @ v1.1.3-r347): .cfr loading, rank9, wavelet tree, run-block
 * BWT rank/access, FM-index backward search / locate, Classifier::Query, the taxonomy
 * tail and the SDUST pre-step.  Every function cites the reference file:line it restates.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this
 * library, and only as the checker.  Nothing under centrifuger_amd/ links or calls it.
 *
 * Parity pinning: validated against the REAL reference compiled in the dev container
 * (oracle/_ref, built by oracle/Makefile from /root/reference) — TSV byte-identity on
 * synthetic read sets and exhaustive Rank/Access/BackwardSearch/locate vectors
 * (live: tests/test_oracle_vs_ref.py; committed fixtures: tests/test_oracle_golden.py + tests/golden/).
 */
#ifndef CFR_ORACLE_H
#define CFR_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- L0: plain bitvector + rank9 (Bitvector_Plain.hpp, DS_Rank.hpp:146-297) ---- */
typedef struct {
  uint64_t n;        /* bits */
  uint64_t *B;       /* ceil(n/64) words */
  uint64_t wordCnt;
  uint64_t *R;       /* 2*ceil(wordCnt/8) words */
} ora_bitvec;

/* ---- alphabet (Alphabet.hpp:194-224) ---- */
typedef struct {
  int method;
  uint64_t n;
  char list[256];
  int32_t code[256];
  int16_t codeLen[256];
} ora_alphabet;

/* ---- L1: wavelet tree (Sequence_WaveletTree.hpp) ---- */
typedef struct {
  uint64_t prefix;
  int32_t prefixLen;
  int32_t children[2];
  ora_bitvec v;
} ora_wt_node;

typedef struct {
  uint64_t n;
  ora_alphabet alphabet;
  int32_t nodeCnt;
  ora_wt_node *T;
} ora_wavelet;

/* ---- L1: run-block sequence (Sequence_RunBlock.hpp:15-20) ---- */
typedef struct {
  uint64_t n;
  ora_alphabet alphabet;
  uint64_t b, blockCnt;
  ora_bitvec useRunBlock;
  ora_wavelet waveletSeq, runBlockSeq;
} ora_runblock;

/* ---- L1: run-block sequence over ONE wavelet tree (Sequence_RunBlockOneTree.hpp:15-21): protein indexes ---- */
typedef struct {
  uint64_t n;
  ora_alphabet alphabet;
  uint64_t b, blockCnt;
  ora_bitvec useRunBlock;
  ora_bitvec *alphabetRB;     /* one per alphabet symbol */
  ora_wavelet compressedSeq;
} ora_runblock1;

/* ---- FixedSizeElemArray (FixedSizeElemArray.hpp) ---- */
typedef struct {
  uint64_t size;   /* words */
  int32_t l;       /* bits per element */
  uint64_t n;
  uint64_t *W;
} ora_fsea;

/* ---- L2: FM index (FMIndex.hpp:13-63, 188-199) ---- */
typedef struct {
  uint64_t n, plainAlphabetBits, firstISA;
  char lastChr;
  int oneTree;              /* FMIndex<Sequence_RunBlockOneTree> (protein, CentrifugerClass.cpp:1001-1004) */
  ora_runblock bwt;
  ora_runblock1 bwt1;
  ora_alphabet alphabets, plainCoder;
  uint64_t C[257];
  /* aux data */
  uint64_t auxN;
  int32_t sampleStrategy, sampleRate;
  uint64_t sampleSize, precomputeWidth, precomputeSize, adjustedSA0;
  ora_fsea sampledSA;
  uint64_t *precomputedRange;   /* pairs (first,second) */
  uint64_t maxLcp;
  uint64_t selectedCnt;
  int32_t selectedFilterRate;
  uint64_t *selectedRows, *selectedVals;  /* sorted by row (std::map order) */
  uint64_t *selectedFilter;               /* NULL when selectedCnt == 0 */
  int hasEndMarker;
  ora_fsea endMarkerSA;
} ora_fm;

/* ---- taxonomy (.2.cfr; Taxonomy.hpp:1259-1287) ---- */
typedef struct {
  uint64_t nodeCnt, seqCnt, extraSeqCnt, rootCTaxId;
  uint64_t *parent;
  uint8_t *rank;
  uint64_t *origTaxId;   /* MapID inverse */
  uint64_t origCnt;
  char **taxName;
  uint64_t *seqIdToTaxId;
  char **seqName;        /* seqCnt + extraSeqCnt */
  uint8_t taxRankNum[64];
} ora_taxonomy;

typedef struct {
  int maxResult;               /* -k, default 1 */
  int minHitLen;               /* <=0: infer */
  int maxResultPerHitFactor;   /* 40 */
  uint64_t considerSecondaryHitLen;     /* 2000 */
  double considerSecondaryScoreFactor;  /* 0.995 */
  int outputExpandedResult;    /* --expand-taxid (Classifier.hpp:22): keep the children that ReduceTaxIds promoted */
} ora_param;

typedef struct {
  ora_fm fm;
  ora_taxonomy tax;
  ora_param param;
  int scoreHitLenAdjust;  /* 15; 5 for a protein index (Classifier.hpp:928-932) */
  int protein;            /* Classifier::_protein: translated search, no dust */
} ora_index;

/* operation counters: the N's of SURVEY.md §8(d) "algorithmic bytes" */
typedef struct {
  uint64_t bitrank, bitaccess, ftab, sampled, filter, hits, bs_calls, extends, lf_steps, locates, read_bases;
  uint64_t bitrank_locate, bitaccess_locate;   /* the part of bitrank/bitaccess spent inside BackwardToSampledSA */
  uint64_t ext_single_row, ext_two_records;    /* extends with sp==ep / with sp,ep in different 128-row blocks */
} ora_counters;

typedef struct {
  uint64_t sp, ep;
  int32_t l, strand, offset;
} ora_hit;

typedef struct {
  ora_hit *a;
  size_t n, cap;
} ora_hitvec;

#define ORA_MAX_MATCH 64
typedef struct {
  uint64_t score, secondaryScore;
  int32_t hitLength, queryLength;
  int32_t nmatch;             /* number of output rows (0 = unclassified) */
  /* kind 0: seqId (name = sequence name) ; kind 1: compact taxid (name = rank string) */
  int32_t kind[ORA_MAX_MATCH];
  uint64_t id[ORA_MAX_MATCH];
  uint64_t taxid[ORA_MAX_MATCH];   /* ORIGINAL tax id printed in column 3 */
  /* --expand-taxid (Classifier.hpp:49, 792-838): ORIGINAL tax ids of the children promoted into match i =
   * expanded[expOff[i] .. expOff[i+1]); expanded == NULL when every string is empty.  malloc'ed by the classification,
   * released by ora_results_free (a result must start zeroed: calloc / ctypes arrays do). */
  uint64_t *expanded;
  int32_t expOff[ORA_MAX_MATCH + 1];
} ora_result;

/* default parameter block (Classifier.hpp:28-37) */
void ora_param_default(ora_param *p);

/* load <prefix>.1.cfr + <prefix>.2.cfr ; returns NULL on failure (message on stderr) */
ora_index *ora_index_load(const char *prefix, const ora_param *param);
void ora_index_free(ora_index *idx);
int ora_is_protein_index(const char *prefix);

/* L0-L2 primitives (counters may be NULL) */
uint64_t ora_bv_rank1(const ora_bitvec *bv, uint64_t i, int inclusive, ora_counters *c);
int ora_bv_access(const ora_bitvec *bv, uint64_t i, ora_counters *c);
uint64_t ora_bv_rank(const ora_bitvec *bv, int type, uint64_t i, int inclusive, ora_counters *c);
uint64_t ora_wt_rank(const ora_wavelet *w, char ch, uint64_t i, int inclusive, ora_counters *c);
uint64_t ora_wt_rank_and_test(const ora_wavelet *w, char ch, uint64_t i, int *isC, ora_counters *c);
char ora_wt_access(const ora_wavelet *w, uint64_t i, ora_counters *c);
uint64_t ora_rb_rank(const ora_runblock *s, char ch, uint64_t i, int inclusive, ora_counters *c);
char ora_rb_access(const ora_runblock *s, uint64_t i, ora_counters *c);
uint64_t ora_rb1_rank(const ora_runblock1 *s, char ch, uint64_t i, int inclusive, ora_counters *c);
char ora_rb1_access(const ora_runblock1 *s, uint64_t i, ora_counters *c);
char ora_fm_access(const ora_fm *fm, uint64_t i, ora_counters *c);    /* _BWT.Access on whichever sequence class the index holds */
char ora_dna_to_aa(char a, char b, char c);                           /* Classifier::DnaToAa (Classifier.hpp:131-241) */
uint64_t ora_fsea_read(const ora_fsea *a, uint64_t i);

uint64_t ora_fm_rank(const ora_fm *fm, char ch, uint64_t p, int inclusive, ora_counters *c);
void ora_fm_backward_extend(const ora_fm *fm, char ch, uint64_t sp, uint64_t ep,
                            uint64_t *nsp, uint64_t *nep, ora_counters *c);
uint64_t ora_fm_lf(const ora_fm *fm, char ch, uint64_t p, ora_counters *c);
uint64_t ora_fm_backward_search(const ora_fm *fm, const char *s, uint64_t m,
                                uint64_t *sp, uint64_t *ep, ora_counters *c);
uint64_t ora_fm_backward_to_sampled_sa(const ora_fm *fm, uint64_t i, uint64_t *l, ora_counters *c);

/* L3 */
size_t ora_get_hits_from_read(const ora_index *idx, const char *r, size_t len, ora_hitvec *hits, ora_counters *c);
void ora_adjust_hit_boundary(const ora_index *idx, const char *r, const char *rc, int len,
                             ora_hitvec strandHits[2], ora_counters *c);
size_t ora_search_forward_and_reverse(const ora_index *idx, const char *r1, const char *r2,
                                      ora_hitvec *hits, ora_counters *c);
size_t ora_get_classification_from_hits(const ora_index *idx, const ora_hitvec *hits,
                                        ora_result *res, ora_counters *c);
void ora_query(const ora_index *idx, const char *r1, const char *r2, ora_result *res, ora_counters *c);
void ora_hitvec_free(ora_hitvec *v);

/* SDUST pre-step (Dustmasker.hpp; CentrifugerClass.cpp:276-316): overwrite masked bases with 'N' */
void ora_dust_mask_inplace(char *s, size_t n);

/* batch: reads given as flat buffers (bases + n+1 offsets); mates optional (NULL).
 * threads follow the reference's i % threadCnt striding (CentrifugerClass.cpp:251-254).
 * dust != 0 applies ora_dust_mask_inplace on a private copy first. */
void ora_classify_batch(const ora_index *idx, const uint8_t *bases1, const uint64_t *offs1,
                        const uint8_t *bases2, const uint64_t *offs2, size_t nreads,
                        int dust, int nthreads, ora_result *results, ora_counters *total);

/* the hit list Query builds for one read (for kernel-level parity tests). returns count; fills up to cap */
size_t ora_query_hits(const ora_index *idx, const char *r1, const char *r2, ora_hit *out, size_t cap);

/* TSV (ResultWriter.hpp:186-242).  returns bytes written into buf (needs cap) or required size */
size_t ora_format_result(const ora_index *idx, const char *readid, const ora_result *r, char *buf, size_t cap);
const char *ora_tsv_header(void);
const char *ora_tsv_header_for(const ora_index *idx);   /* with the expandedTaxIDs column when the index was loaded with outputExpandedResult */
void ora_results_free(ora_result *r, size_t n);
const char *ora_tax_rank_string(uint8_t rank);

/* taxonomy helpers exposed for tests */
uint64_t ora_tax_lca(const ora_taxonomy *t, const uint64_t *taxIds, int n);
int ora_tax_reduce(const ora_taxonomy *t, const uint64_t *taxIds, int n, int k, uint64_t *out, int outCap);

#ifdef __cplusplus
}
#endif
#endif
