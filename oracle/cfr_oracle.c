/*
 * cfr_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see cfr_oracle.h).
 *
 * CPU restatement (plain C11) of the reference path.  Citations are into /root/reference
 * (mourisl/centrifuger v1.1.3-r347).  Data structures mirror the reference's on purpose
 * (rank9 / 3-node wavelet / run-block) so that the operation counters equal the
 * reference's operation counts (SURVEY.md §8(d) algorithmic bytes).
 */
#include "cfr_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define DIV_CEIL(x, y) (((x) % (y)) ? ((x) / (y) + 1) : ((x) / (y)))

static void *xmalloc(size_t n) {
  void *p = malloc(n ? n : 1);
  if (!p) { fprintf(stderr, "oracle: out of memory (%zu)\n", n); abort(); }
  return p;
}
static void *xcalloc(size_t n, size_t s) {
  void *p = calloc(n ? n : 1, s ? s : 1);
  if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
  return p;
}
static int rd(FILE *fp, void *dst, size_t sz, size_t cnt) {
  return fread(dst, sz, cnt, fp) == cnt ? 0 : -1;
}
#define RD(fp, x) do { if (rd(fp, &(x), sizeof(x), 1)) return -1; } while (0)

/* ======================================================================== L0 */

/* Bitvector_Plain::Load (Bitvector_Plain.hpp:198-221), DS_Rank9::Load (DS_Rank.hpp:284-296),
 * DS_Select stub (DS_Select.hpp:679-686; speed must be 0 on this path). */
static int bv_load(FILE *fp, ora_bitvec *bv) {
  uint64_t space; int32_t rb, sb, selSpeed, selType;
  memset(bv, 0, sizeof(*bv));
  RD(fp, space); RD(fp, bv->n); RD(fp, rb); RD(fp, sb); RD(fp, selSpeed); RD(fp, selType);
  if (bv->n > 0) {
    uint64_t words = DIV_CEIL(bv->n, 64);
    bv->B = xmalloc(words * 8);
    if (rd(fp, bv->B, 8, words)) return -1;
    uint64_t rspace; RD(fp, rspace); RD(fp, bv->wordCnt);
    uint64_t blk = DIV_CEIL(bv->wordCnt, 8);
    bv->R = xmalloc(blk * 2 * 8);
    if (rd(fp, bv->R, 8, blk * 2)) return -1;
    uint64_t sspace, sn; int32_t sspeed;
    RD(fp, sspace); RD(fp, sn); RD(fp, sspeed);
    if (sspeed != 0 && sn != 0) { fprintf(stderr, "oracle: select structures not supported (speed=%d)\n", sspeed); return -1; }
  }
  return 0;
}
static void bv_free(ora_bitvec *bv) { free(bv->B); free(bv->R); memset(bv, 0, sizeof(*bv)); }

/* DS_Rank9::Query (DS_Rank.hpp:255-273) via Bitvector_Plain::Rank1 (Bitvector_Plain.hpp:134-137) */
uint64_t ora_bv_rank1(const ora_bitvec *bv, uint64_t i, int inclusive, ora_counters *c) {
  if (i >= bv->n) i = bv->n - 1;                 /* DS_Rank.hpp:259-260 */
  if (c) c->bitrank++;
  const uint64_t wi = i >> 6;
  const uint64_t ri = (wi >> 3) * 2;
  const uint64_t t = (wi & 7) - 1;
  const uint64_t mask = ((((uint64_t)1 << (i & 63)) - 1) << inclusive) + (uint64_t)inclusive;
  return bv->R[ri] + ((bv->R[ri + 1] >> ((t + ((t >> 60) & 8)) * 9)) & 0x1ff)
       + (uint64_t)__builtin_popcountll(bv->B[wi] & mask);
}
/* Bitvector_Plain::Access (:128-131) */
int ora_bv_access(const ora_bitvec *bv, uint64_t i, ora_counters *c) {
  if (c) c->bitaccess++;
  return (int)((bv->B[i >> 6] >> (i & 63)) & 1);
}
/* Bitvector::Rank0 / Rank (Bitvector.hpp:45-57): Rank0 does not re-clamp i */
uint64_t ora_bv_rank(const ora_bitvec *bv, int type, uint64_t i, int inclusive, ora_counters *c) {
  uint64_t r1 = ora_bv_rank1(bv, i, inclusive, c);
  return type == 1 ? r1 : i + (uint64_t)inclusive - r1;
}

/* Alphabet::Load (Alphabet.hpp:208-221) */
static int alphabet_load(FILE *fp, ora_alphabet *a) {
  uint64_t space;
  memset(a, 0, sizeof(*a));
  RD(fp, space); RD(fp, a->method); RD(fp, a->n);
  if (a->n != 0) {
    if (a->n > 255) return -1;
    if (rd(fp, a->list, 1, a->n)) return -1;
    if (rd(fp, a->code, 4, 256)) return -1;
    if (rd(fp, a->codeLen, 2, 256)) return -1;
  }
  return 0;
}
/* Alphabet::IsIn (Alphabet.hpp:169-176) */
static inline int alphabet_is_in(const ora_alphabet *a, char ch) {
  for (uint64_t i = 0; i < a->n; ++i) if (a->list[i] == ch) return 1;
  return 0;
}

/* ======================================================================== L1 wavelet */

/* Sequence::Load (Sequence.hpp:31-36) + Sequence_WaveletTree::Load (:312-327) */
static int wt_load(FILE *fp, ora_wavelet *w) {
  uint64_t space; int32_t selSpeed;
  memset(w, 0, sizeof(*w));
  RD(fp, space); RD(fp, w->n);
  if (alphabet_load(fp, &w->alphabet)) return -1;
  RD(fp, w->nodeCnt); RD(fp, selSpeed);
  if (w->alphabet.n == 0) { w->nodeCnt = 0; return 0; }   /* empty tree (:320-321) */
  w->T = xcalloc((size_t)w->nodeCnt, sizeof(ora_wt_node));
  for (int i = 0; i < w->nodeCnt; ++i) {
    RD(fp, w->T[i].prefix); RD(fp, w->T[i].prefixLen);
    if (rd(fp, w->T[i].children, 4, 2)) return -1;
    if (bv_load(fp, &w->T[i].v)) return -1;
  }
  return 0;
}
static void wt_free(ora_wavelet *w) {
  for (int i = 0; i < w->nodeCnt; ++i) bv_free(&w->T[i].v);
  free(w->T); memset(w, 0, sizeof(*w));
}

/* Sequence_WaveletTree::Rank (:235-264) */
uint64_t ora_wt_rank(const ora_wavelet *w, char ch, uint64_t i, int inclusive, ora_counters *c) {
  int l = w->alphabet.codeLen[(unsigned char)ch];
  uint64_t code = (uint64_t)w->alphabet.code[(unsigned char)ch];
  int ti = 0;
  if (!inclusive) { if (i == 0) return 0; --i; }
  for (int depth = 0; depth < l; ++depth) {
    int b = (int)((code >> (l - depth - 1)) & 1);
    i = ora_bv_rank(&w->T[ti].v, b, i, 1, c);
    if (i == 0 || depth == l - 1) break;
    --i;
    ti = w->T[ti].children[b];
  }
  return i;
}
/* Sequence_WaveletTree::RankAndTest (:268-293) */
uint64_t ora_wt_rank_and_test(const ora_wavelet *w, char ch, uint64_t i, int *isC, ora_counters *c) {
  int l = w->alphabet.codeLen[(unsigned char)ch];
  uint64_t code = (uint64_t)w->alphabet.code[(unsigned char)ch];
  int ti = 0;
  *isC = 1;
  for (int depth = 0; depth < l; ++depth) {
    int b = (int)((code >> (l - depth - 1)) & 1);
    if (*isC && b != ora_bv_access(&w->T[ti].v, i, c)) *isC = 0;
    i = ora_bv_rank(&w->T[ti].v, b, i, 1, c);
    if (i == 0 || depth == l - 1) break;
    --i;
    ti = w->T[ti].children[b];
  }
  return i;
}
/* Sequence_WaveletTree::Access (:215-232); Alphabet::Decode plain (Alphabet.hpp:128-142) */
char ora_wt_access(const ora_wavelet *w, uint64_t i, ora_counters *c) {
  uint64_t code = 0;
  int ti = 0;
  for (; ti != -1;) {
    int b = ora_bv_access(&w->T[ti].v, i, c);
    code = (code << 1) | (uint64_t)b;
    i = ora_bv_rank(&w->T[ti].v, b, i, 1, c) - 1;
    ti = w->T[ti].children[b];
  }
  return w->alphabet.list[code];
}

/* ======================================================================== L1 run-block */

/* Sequence_RunBlock::Load (Sequence_RunBlock.hpp:478-488) */
static int rb_load(FILE *fp, ora_runblock *s) {
  uint64_t space;
  memset(s, 0, sizeof(*s));
  RD(fp, space); RD(fp, s->n);
  if (alphabet_load(fp, &s->alphabet)) return -1;
  RD(fp, s->b); RD(fp, s->blockCnt);
  if (bv_load(fp, &s->useRunBlock)) return -1;
  if (wt_load(fp, &s->waveletSeq)) return -1;
  if (wt_load(fp, &s->runBlockSeq)) return -1;
  return 0;
}
static void rb_free(ora_runblock *s) {
  bv_free(&s->useRunBlock); wt_free(&s->waveletSeq); wt_free(&s->runBlockSeq);
}

/* Sequence_RunBlock::Access (:360-376) */
char ora_rb_access(const ora_runblock *s, uint64_t i, ora_counters *c) {
  uint64_t bi = i / s->b;
  int type = ora_bv_access(&s->useRunBlock, bi, c);
  if (type == 0) {
    uint64_t r = ora_bv_rank(&s->useRunBlock, 1, bi, 1, c);
    i -= s->b * r;
    return ora_wt_access(&s->waveletSeq, i, c);
  } else {
    uint64_t r = ora_bv_rank(&s->useRunBlock, 0, bi, 1, c);
    i -= s->b * r;
    return ora_wt_access(&s->runBlockSeq, i / s->b, c);
  }
}
/* Sequence_RunBlock::Rank (:378-416) */
uint64_t ora_rb_rank(const ora_runblock *s, char ch, uint64_t i, int inclusive, ora_counters *c) {
  if (!inclusive) { if (i == 0) return 0; --i; }
  uint64_t bi = i / s->b;
  int type = ora_bv_access(&s->useRunBlock, bi, c);
  uint64_t ranki = s->b < s->n ? ora_bv_rank(&s->useRunBlock, type, bi, 1, c) : 1;
  uint64_t otherRanki = (bi + 1) - ranki;
  uint64_t ret;
  if (type == 0)
    ret = ora_wt_rank(&s->waveletSeq, ch, (ranki - 1) * s->b + i % s->b, 1, c);
  else {
    int inRun = 1;
    uint64_t rbRank = ora_wt_rank_and_test(&s->runBlockSeq, ch, ranki - 1, &inRun, c);
    ret = inRun ? (rbRank - 1) * s->b + i % s->b + 1 : rbRank * s->b;
  }
  if (otherRanki == 0) return ret;
  if (type == 0) ret += ora_wt_rank(&s->runBlockSeq, ch, otherRanki - 1, 1, c) * s->b;
  else ret += ora_wt_rank(&s->waveletSeq, ch, otherRanki * s->b - 1, 1, c);
  return ret;
}

/* Sequence_RunBlockOneTree::Load (Sequence_RunBlockOneTree.hpp:499-513) */
static int rb1_load(FILE *fp, ora_runblock1 *s) {
  uint64_t space;
  memset(s, 0, sizeof(*s));
  RD(fp, space); RD(fp, s->n);
  if (alphabet_load(fp, &s->alphabet)) return -1;
  RD(fp, s->b); RD(fp, s->blockCnt);
  if (bv_load(fp, &s->useRunBlock)) return -1;
  s->alphabetRB = xcalloc(s->alphabet.n ? s->alphabet.n : 1, sizeof(ora_bitvec));
  for (uint64_t i = 0; i < s->alphabet.n; ++i) if (bv_load(fp, &s->alphabetRB[i])) return -1;
  if (wt_load(fp, &s->compressedSeq)) return -1;
  return 0;
}
static void rb1_free(ora_runblock1 *s) {
  bv_free(&s->useRunBlock);
  if (s->alphabetRB) for (uint64_t i = 0; i < s->alphabet.n; ++i) bv_free(&s->alphabetRB[i]);
  free(s->alphabetRB);
  wt_free(&s->compressedSeq);
}
/* Sequence_RunBlockOneTree::Access (:382-396) */
char ora_rb1_access(const ora_runblock1 *s, uint64_t i, ora_counters *c) {
  uint64_t bi = i / s->b;
  uint64_t r = ora_bv_rank(&s->useRunBlock, 1, bi, 0, c);
  int type = ora_bv_access(&s->useRunBlock, bi, c);
  if (type == 1) i -= i % s->b;            /* the "representative" position of a run block */
  i -= (s->b - 1) * r;                      /* every run block before takes b-1 positions out */
  return ora_wt_access(&s->compressedSeq, i, c);
}
/* Sequence_RunBlockOneTree::Rank (:398-435) */
uint64_t ora_rb1_rank(const ora_runblock1 *s, char ch, uint64_t i, int inclusive, ora_counters *c) {
  if (!inclusive) { if (i == 0) return 0; --i; }
  uint64_t bi = i / s->b;
  uint64_t r = ora_bv_rank(&s->useRunBlock, 1, bi, 0, c);
  int type = ora_bv_access(&s->useRunBlock, bi, c);
  uint64_t remainder = 0, ci = i;
  if (type == 1) { remainder = i % s->b; ci -= i % s->b; }
  ci -= (s->b - 1) * r;
  uint64_t ret;
  int inRun = 0;
  if (type == 1) ret = ora_wt_rank_and_test(&s->compressedSeq, ch, ci, &inRun, c);
  else ret = ora_wt_rank(&s->compressedSeq, ch, ci, 1, c);
  if (ret > 0 && (s->b < s->n || inRun))
    ret += ora_bv_rank(&s->alphabetRB[s->alphabet.code[(unsigned char)ch]], 1, ret - 1, 1, c) * (s->b - 1);
  if (inRun) ret = ret - (s->b - 1) + remainder;
  return ret;
}

/* ======================================================================== FixedSizeElemArray */

/* FixedSizeElemArray::Load (:396-404) */
static int fsea_load(FILE *fp, ora_fsea *a) {
  memset(a, 0, sizeof(*a));
  RD(fp, a->size); RD(fp, a->l); RD(fp, a->n);
  uint64_t words = DIV_CEIL(a->n * (uint64_t)a->l, 64);
  uint64_t alloc = a->size > words ? a->size : words;
  a->W = xcalloc(alloc + 1, 8);
  if (rd(fp, a->W, 8, words)) return -1;
  return 0;
}
/* FixedSizeElemArray::Read (:102-105) = Utils::BitsRead (Utils.hpp:197-219) */
uint64_t ora_fsea_read(const ora_fsea *a, uint64_t i) {
  const uint64_t s = i * (uint64_t)a->l, e = (i + 1) * (uint64_t)a->l - 1;
  const uint64_t is = s >> 6, ie = e >> 6;
  const int rs = (int)(s & 63);
  if (is == ie) {
    uint64_t len = e - s + 1;
    uint64_t m = len >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << len) - 1);
    return (a->W[is] >> rs) & m;
  } else {
    const int re = (int)(e & 63);
    return (a->W[is] >> rs) | ((a->W[ie] & (((uint64_t)1 << (re + 1)) - 1)) << (64 - rs));
  }
}

/* ======================================================================== L2 FM index */

/* _FMIndexAuxData::Load (FMIndex.hpp:136-185) */
static int aux_load(FILE *fp, ora_fm *fm) {
  RD(fp, fm->auxN); RD(fp, fm->sampleStrategy); RD(fp, fm->sampleRate); RD(fp, fm->sampleSize);
  RD(fp, fm->precomputeWidth); RD(fp, fm->precomputeSize); RD(fp, fm->adjustedSA0);
  if (fsea_load(fp, &fm->sampledSA)) return -1;
  fm->precomputedRange = xmalloc(fm->precomputeSize * 16);
  if (rd(fp, fm->precomputedRange, 16, fm->precomputeSize)) return -1;
  RD(fp, fm->maxLcp);
  if (fm->maxLcp > 0) {   /* semiLcp arrays: not used on this path, skip */
    uint64_t words = DIV_CEIL(fm->auxN, 64);
    if (fseek(fp, (long)(words * 16), SEEK_CUR)) return -1;
  }
  RD(fp, fm->selectedCnt); RD(fp, fm->selectedFilterRate);
  if (fm->selectedCnt > 0) {
    fm->selectedRows = xmalloc(fm->selectedCnt * 8);
    fm->selectedVals = xmalloc(fm->selectedCnt * 8);
    uint64_t fbits = DIV_CEIL(fm->auxN, (uint64_t)fm->selectedFilterRate);
    fm->selectedFilter = xcalloc(DIV_CEIL(fbits, 64), 8);
    for (uint64_t i = 0; i < fm->selectedCnt; ++i) {
      uint64_t pr[2];
      if (rd(fp, pr, 8, 2)) return -1;
      fm->selectedRows[i] = pr[0]; fm->selectedVals[i] = pr[1];
      uint64_t fb = pr[0] / (uint64_t)fm->selectedFilterRate;
      fm->selectedFilter[fb >> 6] |= (uint64_t)1 << (fb & 63);
    }
    /* the file is written in std::map order (ascending row); keep a check */
    for (uint64_t i = 1; i < fm->selectedCnt; ++i)
      if (fm->selectedRows[i] <= fm->selectedRows[i - 1]) { fprintf(stderr, "oracle: selectedSA not sorted\n"); return -1; }
  }
  uint8_t hem = 0;
  if (fread(&hem, 1, 1, fp) != 1) hem = 0;       /* old indexes: FMIndex.hpp:178-181 */
  fm->hasEndMarker = hem != 0;
  if (fm->hasEndMarker && fsea_load(fp, &fm->endMarkerSA)) return -1;
  return 0;
}

/* FMIndex::Load (FMIndex.hpp:588-606) */
static int fm_load(FILE *fp, ora_fm *fm, int oneTree) {
  memset(fm, 0, sizeof(*fm));
  fm->oneTree = oneTree;
  RD(fp, fm->n); RD(fp, fm->plainAlphabetBits); RD(fp, fm->firstISA); RD(fp, fm->lastChr);
  if (oneTree ? rb1_load(fp, &fm->bwt1) : rb_load(fp, &fm->bwt)) return -1;
  if (alphabet_load(fp, &fm->alphabets)) return -1;
  if (alphabet_load(fp, &fm->plainCoder)) return -1;
  if (rd(fp, fm->C, 8, fm->plainCoder.n + 1)) return -1;
  return aux_load(fp, fm);
}
static void fm_free(ora_fm *fm) {
  rb_free(&fm->bwt); rb1_free(&fm->bwt1); free(fm->sampledSA.W); free(fm->precomputedRange);
  free(fm->selectedRows); free(fm->selectedVals); free(fm->selectedFilter); free(fm->endMarkerSA.W);
}

/* FMIndex::Rank (:352-362) */
char ora_fm_access(const ora_fm *fm, uint64_t i, ora_counters *c) {
  return fm->oneTree ? ora_rb1_access(&fm->bwt1, i, c) : ora_rb_access(&fm->bwt, i, c);
}
uint64_t ora_fm_rank(const ora_fm *fm, char ch, uint64_t p, int inclusive, ora_counters *c) {
  uint64_t ret = fm->oneTree ? ora_rb1_rank(&fm->bwt1, ch, p, inclusive, c) : ora_rb_rank(&fm->bwt, ch, p, inclusive, c);
  if (ch == fm->lastChr && (p < fm->firstISA || (!inclusive && p == fm->firstISA))) ++ret;
  return ret;
}
/* FMIndex::BackwardExtend range form (:364-379) */
void ora_fm_backward_extend(const ora_fm *fm, char ch, uint64_t sp, uint64_t ep,
                            uint64_t *nsp, uint64_t *nep, ora_counters *c) {
  uint64_t offset = fm->C[fm->plainCoder.code[(unsigned char)ch]];
  if (c) { c->extends++; if (sp == ep) c->ext_single_row++; else if ((sp >> 7) != (ep >> 7)) c->ext_two_records++; }
  *nsp = offset + ora_fm_rank(fm, ch, sp, 0, c) + 1 - 1;
  if (sp != ep) *nep = offset + ora_fm_rank(fm, ch, ep, 1, c) - 1;
  else *nep = *nsp + ((ora_fm_access(fm, ep, c) == ch) ? 0 : (uint64_t)-1);
}
/* FMIndex::BackwardExtend LF form (:382-386) */
uint64_t ora_fm_lf(const ora_fm *fm, char ch, uint64_t p, ora_counters *c) {
  uint64_t offset = fm->C[fm->plainCoder.code[(unsigned char)ch]];
  return offset + ora_fm_rank(fm, ch, p, 1, c) - 1;
}
/* FMIndex::GetBackwardSearchInitialRange (:388-422) */
static uint64_t fm_initial_range(const ora_fm *fm, const char *s, uint64_t m, uint64_t *sp, uint64_t *ep, ora_counters *c) {
  if (fm->precomputeWidth > 0) {
    uint64_t initW = 0;
    for (uint64_t i = 0; i < fm->precomputeWidth; ++i) {
      if (!alphabet_is_in(&fm->alphabets, s[m - 1 - i])) { *sp = 1; *ep = 0; return i; }
      initW = (initW << fm->plainAlphabetBits) | (uint64_t)fm->plainCoder.code[(unsigned char)s[m - 1 - i]];
    }
    if (c) c->ftab++;
    if (fm->precomputedRange[2 * initW + 1] == 0) { *sp = 1; *ep = 0; return fm->precomputeWidth - 1; }
    *sp = fm->precomputedRange[2 * initW];
    *ep = *sp + fm->precomputedRange[2 * initW + 1] - 1;
    return fm->precomputeWidth;
  }
  *sp = 0; *ep = fm->n - 1;
  return 0;
}
/* FMIndex::BackwardSearch (:487-510) */
uint64_t ora_fm_backward_search(const ora_fm *fm, const char *s, uint64_t m, uint64_t *sp, uint64_t *ep, ora_counters *c) {
  if (c) c->bs_calls++;
  if (m < fm->precomputeWidth) return 0;
  uint64_t l = fm_initial_range(fm, s, m, sp, ep, c);
  if (l < fm->precomputeWidth) return l;
  uint64_t nsp = *sp, nep = *ep;
  while (l < m) {
    if (!alphabet_is_in(&fm->alphabets, s[m - 1 - l])) break;
    ora_fm_backward_extend(fm, s[m - 1 - l], *sp, *ep, &nsp, &nep, c);
    if (nsp > nep || nep > fm->n) break;
    *sp = nsp; *ep = nep; ++l;
  }
  return l;
}
/* FMIndex::GetSampledSA (:203-231) */
static int fm_get_sampled_sa(const ora_fm *fm, uint64_t i, uint64_t *sa, ora_counters *c) {
  if (i == fm->firstISA) { *sa = fm->adjustedSA0; return 1; }
  else if (i % (uint64_t)fm->sampleRate == 0) {
    if (c) c->sampled++;
    *sa = ora_fsea_read(&fm->sampledSA, i / (uint64_t)fm->sampleRate);
    return 1;
  } else if (fm->selectedFilter) {
    uint64_t fb = i / (uint64_t)fm->selectedFilterRate;
    if (c) c->filter++;
    if ((fm->selectedFilter[fb >> 6] >> (fb & 63)) & 1) {
      uint64_t lo = 0, hi = fm->selectedCnt;       /* std::map::find */
      while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (fm->selectedRows[mid] < i) lo = mid + 1; else hi = mid; }
      if (lo < fm->selectedCnt && fm->selectedRows[lo] == i) { *sa = fm->selectedVals[lo]; return 1; }
    }
  } else if (fm->hasEndMarker && i < fm->endMarkerSA.n) {
    *sa = ora_fsea_read(&fm->endMarkerSA, i);
    return 1;
  }
  return 0;
}
/* FMIndex::BackwardToSampledSA (:514-524) */
uint64_t ora_fm_backward_to_sampled_sa(const ora_fm *fm, uint64_t i, uint64_t *l, ora_counters *c) {
  uint64_t ret = 0;
  *l = 0;
  uint64_t r0 = 0, a0 = 0;
  if (c) { c->locates++; r0 = c->bitrank; a0 = c->bitaccess; }
  while (!fm_get_sampled_sa(fm, i, &ret, c)) {
    i = ora_fm_lf(fm, ora_fm_access(fm, i, c), i, c);
    if (c) c->lf_steps++;
    ++*l;
  }
  if (c) { c->bitrank_locate += c->bitrank - r0; c->bitaccess_locate += c->bitaccess - a0; }
  return ret;
}

/* ======================================================================== taxonomy */

/* Taxonomy::InitTaxRankNum (Taxonomy.hpp:94-143); rank enum (:25-59) */
enum {
  RANK_UNKNOWN = 0, RANK_STRAIN, RANK_SPECIES, RANK_GENUS, RANK_FAMILY, RANK_ORDER, RANK_CLASS, RANK_PHYLUM,
  RANK_KINGDOM, RANK_DOMAIN, RANK_FORMA, RANK_INFRA_CLASS, RANK_INFRA_ORDER, RANK_PARV_ORDER, RANK_SUB_CLASS,
  RANK_SUB_FAMILY, RANK_SUB_GENUS, RANK_SUB_KINGDOM, RANK_SUB_ORDER, RANK_SUB_PHYLUM, RANK_SUB_SPECIES,
  RANK_SUB_TRIBE, RANK_SUPER_CLASS, RANK_SUPER_FAMILY, RANK_SUPER_KINGDOM, RANK_SUPER_ORDER, RANK_SUPER_PHYLUM,
  RANK_TRIBE, RANK_VARIETAS, RANK_LIFE, RANK_ACELLULAR_ROOT, RANK_MAX
};
static void tax_init_rank_num(uint8_t *t) {
  uint8_t rank = 0;
  memset(t, 0, 64);
  t[RANK_SUB_SPECIES] = rank; t[RANK_STRAIN] = rank++;
  t[RANK_SPECIES] = rank++;
  t[RANK_SUB_GENUS] = rank; t[RANK_GENUS] = rank++;
  t[RANK_SUB_FAMILY] = rank; t[RANK_FAMILY] = rank; t[RANK_SUPER_FAMILY] = rank++;
  t[RANK_SUB_ORDER] = rank; t[RANK_INFRA_ORDER] = rank; t[RANK_PARV_ORDER] = rank; t[RANK_ORDER] = rank; t[RANK_SUPER_ORDER] = rank++;
  t[RANK_INFRA_CLASS] = rank; t[RANK_SUB_CLASS] = rank; t[RANK_CLASS] = rank; t[RANK_SUPER_CLASS] = rank++;
  t[RANK_SUB_PHYLUM] = rank; t[RANK_PHYLUM] = rank; t[RANK_SUPER_PHYLUM] = rank++;
  t[RANK_SUB_KINGDOM] = rank; t[RANK_KINGDOM] = rank++;
  t[RANK_SUPER_KINGDOM] = rank; t[RANK_ACELLULAR_ROOT] = rank; t[RANK_DOMAIN] = rank++;
  t[RANK_FORMA] = rank; t[RANK_SUB_TRIBE] = rank; t[RANK_TRIBE] = rank; t[RANK_VARIETAS] = rank; t[RANK_LIFE] = rank;
  t[RANK_UNKNOWN] = rank;
}
/* Taxonomy::GetTaxRankString (Taxonomy.hpp:497-533) */
const char *ora_tax_rank_string(uint8_t rank) {
  switch (rank) {
    case RANK_STRAIN: return "strain"; case RANK_SPECIES: return "species"; case RANK_GENUS: return "genus";
    case RANK_FAMILY: return "family"; case RANK_ORDER: return "order"; case RANK_CLASS: return "class";
    case RANK_PHYLUM: return "phylum"; case RANK_KINGDOM: return "kingdom"; case RANK_DOMAIN: return "domain";
    case RANK_ACELLULAR_ROOT: return "acellular root"; case RANK_FORMA: return "forma";
    case RANK_INFRA_CLASS: return "infraclass"; case RANK_INFRA_ORDER: return "infraorder";
    case RANK_PARV_ORDER: return "parvorder"; case RANK_SUB_CLASS: return "subclass";
    case RANK_SUB_FAMILY: return "subfamily"; case RANK_SUB_GENUS: return "subgenus";
    case RANK_SUB_KINGDOM: return "subkingdom"; case RANK_SUB_ORDER: return "suborder";
    case RANK_SUB_PHYLUM: return "subphylum"; case RANK_SUB_SPECIES: return "subspecies";
    case RANK_SUB_TRIBE: return "subtribe"; case RANK_SUPER_CLASS: return "superclass";
    case RANK_SUPER_FAMILY: return "superfamily"; case RANK_SUPER_KINGDOM: return "superkingdom";
    case RANK_SUPER_ORDER: return "superorder"; case RANK_SUPER_PHYLUM: return "superphylum";
    case RANK_TRIBE: return "tribe"; case RANK_VARIETAS: return "varietas"; case RANK_LIFE: return "life";
    default: return "no rank";
  }
}

static char *load_string(FILE *fp) {   /* Taxonomy::LoadString (Taxonomy.hpp:411-420) */
  uint64_t len;
  if (fread(&len, 8, 1, fp) != 1) return NULL;
  char *s = xmalloc(len + 1);
  if (len && fread(s, 1, len, fp) != len) { free(s); return NULL; }
  s[len] = 0;
  return s;
}
/* Taxonomy::Load (Taxonomy.hpp:1259-1287); TaxonomyNode layout (:61-82); MapID::Load (MapID.hpp:83-99) */
static int tax_load(FILE *fp, ora_taxonomy *t) {
  memset(t, 0, sizeof(*t));
  tax_init_rank_num(t->taxRankNum);
  RD(fp, t->nodeCnt); RD(fp, t->seqCnt); RD(fp, t->extraSeqCnt);
  t->parent = xmalloc(t->nodeCnt * 8); t->rank = xmalloc(t->nodeCnt);
  for (uint64_t i = 0; i < t->nodeCnt; ++i) {
    uint8_t node[16];
    if (rd(fp, node, 16, 1)) return -1;
    memcpy(&t->parent[i], node, 8); t->rank[i] = node[8];
  }
  RD(fp, t->origCnt);
  t->origTaxId = xmalloc(t->origCnt * 8);
  if (rd(fp, t->origTaxId, 8, t->origCnt)) return -1;
  t->taxName = xcalloc(t->nodeCnt, sizeof(char *));
  for (uint64_t i = 0; i < t->nodeCnt; ++i) if (!(t->taxName[i] = load_string(fp))) return -1;
  t->seqIdToTaxId = xmalloc(t->seqCnt * 8);
  if (rd(fp, t->seqIdToTaxId, 8, t->seqCnt)) return -1;
  uint64_t ns = t->seqCnt + t->extraSeqCnt;
  t->seqName = xcalloc(ns, sizeof(char *));
  for (uint64_t i = 0; i < ns; ++i) if (!(t->seqName[i] = load_string(fp))) return -1;
  t->rootCTaxId = t->nodeCnt;     /* FindRoot (:422-429) */
  for (uint64_t i = 0; i < t->nodeCnt; ++i) if (t->parent[i] == i) { t->rootCTaxId = i; break; }
  return 0;
}
static void tax_free(ora_taxonomy *t) {
  if (t->taxName) for (uint64_t i = 0; i < t->nodeCnt; ++i) free(t->taxName[i]);
  if (t->seqName) for (uint64_t i = 0; i < t->seqCnt + t->extraSeqCnt; ++i) free(t->seqName[i]);
  free(t->taxName); free(t->seqName); free(t->parent); free(t->rank); free(t->origTaxId); free(t->seqIdToTaxId);
}
/* Taxonomy::GetOrigTaxId (:633-639), SeqIdToTaxId (:718-724), GetTaxIdRank (:658-664) */
static uint64_t tax_orig(const ora_taxonomy *t, uint64_t ctid) {
  return t->origTaxId[ctid >= t->nodeCnt ? t->rootCTaxId : ctid];
}
static uint64_t tax_seq_to_tax(const ora_taxonomy *t, uint64_t seqId) {
  return seqId < t->seqCnt ? t->seqIdToTaxId[seqId] : t->nodeCnt;
}
static uint8_t tax_rank(const ora_taxonomy *t, uint64_t ctid) {
  return ctid >= t->nodeCnt ? RANK_UNKNOWN : t->rank[ctid];
}

typedef struct { uint64_t *a; size_t n, cap; } u64vec;
static void u64_push(u64vec *v, uint64_t x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 16; v->a = realloc(v->a, v->cap * 8); if (!v->a) abort(); }
  v->a[v->n++] = x;
}
/* sorted-unique set insert; returns 1 when newly inserted (std::map<size_t,int> used as a set) */
static int set_insert(u64vec *v, uint64_t x) {
  size_t lo = 0, hi = v->n;
  while (lo < hi) { size_t mid = (lo + hi) / 2; if (v->a[mid] < x) lo = mid + 1; else hi = mid; }
  if (lo < v->n && v->a[lo] == x) return 0;
  u64_push(v, 0);
  memmove(v->a + lo + 1, v->a + lo, (v->n - 1 - lo) * 8);
  v->a[lo] = x;
  return 1;
}
static int set_has(const u64vec *v, uint64_t x) {
  size_t lo = 0, hi = v->n;
  while (lo < hi) { size_t mid = (lo + hi) / 2; if (v->a[mid] < x) lo = mid + 1; else hi = mid; }
  return lo < v->n && v->a[lo] == x;
}

/* Taxonomy::LCA (Taxonomy.hpp:733-836).  children != NULL: the lcaChildTaxIds bookkeeping (:771-777, :806-813, :822-830) -
 * one ordered set per backbone node, the LCA's set handed back (compact ids, ascending = std::map order) */
static uint64_t tax_lca(const ora_taxonomy *t, const uint64_t *taxIds, int taxCnt, u64vec *children) {
  int i, j, k;
  if (children) children->n = 0;
  for (i = 0; i < taxCnt; ++i) if (taxIds[i] != t->rootCTaxId) break;
  if (i < taxCnt) k = i; else return t->rootCTaxId;
  u64vec path = {0}, cnt = {0}, tmp = {0};
  uint64_t x = taxIds[k];
  do { u64_push(&path, x); u64_push(&cnt, 1); x = t->parent[x]; } while (x != t->parent[x]);
  u64_push(&path, t->rootCTaxId); u64_push(&cnt, 1);
  int backboneLen = (int)path.n;
  u64vec *bbChild = NULL;                                   /* backboneChildTaxIds (:771-777) */
  if (children) {
    bbChild = xcalloc((size_t)backboneLen, sizeof(u64vec));
    for (j = 1; j < backboneLen; ++j) set_insert(&bbChild[j], path.a[j - 1]);
  }
  int rootCount = 0;
  for (i = 0; i < taxCnt; ++i) {
    if (i == k) continue;
    tmp.n = 0;
    x = taxIds[i];
    if (x == t->parent[x]) { ++rootCount; continue; }
    do { u64_push(&tmp, x); x = t->parent[x]; } while (x != t->parent[x]);
    u64_push(&tmp, t->rootCTaxId);
    int ib, it;
    for (ib = backboneLen - 1, it = (int)tmp.n - 1; ib >= 0 && it >= 0; --ib, --it) {
      if (tmp.a[it] != path.a[ib]) break;
      cnt.a[ib] += 1;
    }
    if (children && it >= 0 && ib + 1 < backboneLen) set_insert(&bbChild[ib + 1], tmp.a[it]);   /* :812-813: where the two paths part */
  }
  for (j = 0; j < backboneLen; ++j) if ((int)cnt.a[j] == taxCnt - rootCount) break;
  uint64_t ret = j >= backboneLen ? t->rootCTaxId : path.a[j];
  if (children) {
    if (j < backboneLen) for (size_t q = 0; q < bbChild[j].n; ++q) u64_push(children, bbChild[j].a[q]);
    for (j = 0; j < backboneLen; ++j) free(bbChild[j].a);
    free(bbChild);
  }
  free(path.a); free(cnt.a); free(tmp.a);
  return ret;
}
uint64_t ora_tax_lca(const ora_taxonomy *t, const uint64_t *taxIds, int taxCnt) { return tax_lca(t, taxIds, taxCnt, NULL); }

/* Taxonomy::ReduceTaxIds (Taxonomy.hpp:839-973).  returns number of promoted ids written to out.
 * child != NULL (promotedChildTaxIds): child[0..*nchild) receives one list per pushed vector - the caller compares *nchild with
 * the return value exactly as Classifier.hpp:823 does; the lists hold COMPACT ids in the order the reference pushes them. */
static int tax_reduce(const ora_taxonomy *t, const uint64_t *taxIds, int taxCnt, int k, uint64_t *out, int outCap,
                      u64vec *child, int *nchild) {
  int i, n = 0;
  if (nchild) *nchild = 0;
  if (taxCnt <= k) {
    for (i = 0; i < taxCnt && i < outCap; ++i) out[i] = taxIds[i];
    return taxCnt;
  }
  for (i = 0; i < taxCnt; ++i)
    if (taxIds[i] >= t->nodeCnt) {
      out[0] = t->nodeCnt;
      if (child) {                                         /* :866-872: every input id, in input order */
        child[0].n = 0;
        for (int j = 0; j < taxCnt; ++j) u64_push(&child[0], taxIds[j]);
        *nchild = 1;
      }
      return 1;
    }
  if (k == 1) {
    if (child) { out[0] = tax_lca(t, taxIds, taxCnt, &child[0]); *nchild = 1; }
    else out[0] = tax_lca(t, taxIds, taxCnt, NULL);
    return 1;
  }

  u64vec inRank[RANK_MAX];
  memset(inRank, 0, sizeof(inRank));
  for (i = 0; i < taxCnt; ++i) {
    uint64_t x = taxIds[i];
    uint8_t prevRankNum = 0, ri;
    set_insert(&inRank[prevRankNum], x);
    do {
      uint8_t rankNum = t->taxRankNum[t->rank[x]];
      if (rankNum != t->taxRankNum[RANK_UNKNOWN] && rankNum > prevRankNum) {
        for (ri = rankNum - 1; ri > prevRankNum; --ri) set_insert(&inRank[ri], x);
        if (!set_insert(&inRank[rankNum], x)) break;
        prevRankNum = rankNum;
      }
      x = t->parent[x];
    } while (x != t->parent[x]);
  }
  uint8_t ri;
  for (ri = 0; ri < t->taxRankNum[RANK_UNKNOWN]; ++ri) if ((int)inRank[ri].n <= k) break;
  for (size_t q = 0; q < inRank[ri].n && n < outCap; ++q) out[n++] = inRank[ri].a[q];
  if (n == 0) out[n++] = t->rootCTaxId;
  else if (child && ri > 0) {                              /* :939-971 */
    for (i = 0; i < n; ++i) child[i].n = 0;
    *nchild = n;
    for (size_t q = 0; q < inRank[ri - 1].n; ++q) {
      uint64_t x = inRank[ri - 1].a[q];
      while (x != t->parent[x]) {
        x = t->parent[x];
        if (t->taxRankNum[t->rank[x]] > ri) break;
        else if (t->taxRankNum[t->rank[x]] == ri) {
          for (i = 0; i < n; ++i) if (out[i] == x) { u64_push(&child[i], inRank[ri - 1].a[q]); break; }   /* promotedTaxIdIdx */
          break;
        }
      }
    }
  }
  for (i = 0; i < RANK_MAX; ++i) free(inRank[i].a);
  return n;
}
int ora_tax_reduce(const ora_taxonomy *t, const uint64_t *taxIds, int taxCnt, int k, uint64_t *out, int outCap) {
  return tax_reduce(t, taxIds, taxCnt, k, out, outCap, NULL, NULL);
}

/* ======================================================================== L3 classifier */

void ora_param_default(ora_param *p) {   /* Classifier.hpp:28-37 */
  p->outputExpandedResult = 0;
  p->maxResult = 1; p->minHitLen = 0; p->maxResultPerHitFactor = 40;
  p->considerSecondaryHitLen = 2000; p->considerSecondaryScoreFactor = 0.995;
}

int ora_is_protein_index(const char *prefix) {   /* Classifier::IsProteinDatabase (:867-895) */
  char *name = xmalloc(strlen(prefix) + 17);
  sprintf(name, "%s.4.cfr", prefix);
  FILE *fp = fopen(name, "r");
  free(name);
  if (!fp) return 0;
  char key[128], val[128];
  int ret = 0;
  while (fscanf(fp, "%127s %127s", key, val) != EOF)
    if (!strcmp(key, "sequence_type") && !strcmp(val, "amino_acid")) ret = 1;
  fclose(fp);
  return ret;
}

static uint64_t power_int(int x, int y) {   /* Utils::PowerInt (Utils.hpp:164-176) */
  uint64_t ret = 1, px = (uint64_t)x;
  while (y) { if (y & 1) ret *= px; px *= px; y >>= 1; }
  return ret;
}

ora_index *ora_index_load(const char *prefix, const ora_param *param) {   /* Classifier::Init (:902-947) */
  ora_index *idx = xcalloc(1, sizeof(*idx));
  char *name = xmalloc(strlen(prefix) + 17);
  idx->protein = ora_is_protein_index(prefix);
  sprintf(name, "%s.1.cfr", prefix);
  FILE *fp = fopen(name, "rb");
  if (!fp) { fprintf(stderr, "oracle: cannot open %s\n", name); goto fail; }
  int rc = fm_load(fp, &idx->fm, idx->protein);
  fclose(fp);
  if (rc) { fprintf(stderr, "oracle: malformed %s\n", name); goto fail; }
  sprintf(name, "%s.2.cfr", prefix);
  fp = fopen(name, "rb");
  if (!fp) { fprintf(stderr, "oracle: cannot open %s\n", name); goto fail; }
  rc = tax_load(fp, &idx->tax);
  fclose(fp);
  if (rc) { fprintf(stderr, "oracle: malformed %s\n", name); goto fail; }
  free(name);
  if (param) idx->param = *param; else ora_param_default(&idx->param);
  idx->scoreHitLenAdjust = 15;
  if (idx->protein) idx->scoreHitLenAdjust /= 3;      /* Classifier.hpp:928-932 */
  if (idx->param.minHitLen <= 0) {   /* InferMinHitLen (:113-129) */
    int mhl = idx->protein ? 11 : 23;
    int asz = (int)idx->fm.alphabets.n;
    uint64_t kmerspace = power_int(asz, mhl) / 2;
    for (; mhl <= 32; ++mhl) { if (kmerspace >= 100 * idx->fm.n) break; kmerspace *= (uint64_t)asz; }
    idx->param.minHitLen = mhl;
  }
  return idx;
fail:
  free(name);
  ora_index_free(idx);
  return NULL;
}
void ora_index_free(ora_index *idx) {
  if (!idx) return;
  fm_free(&idx->fm); tax_free(&idx->tax); free(idx);
}

static void hit_push(ora_hitvec *v, ora_hit h) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 8; v->a = realloc(v->a, v->cap * sizeof(ora_hit)); if (!v->a) abort(); }
  v->a[v->n++] = h;
}
static void hit_append(ora_hitvec *dst, const ora_hitvec *src) { for (size_t i = 0; i < src->n; ++i) hit_push(dst, src->a[i]); }
void ora_hitvec_free(ora_hitvec *v) { free(v->a); v->a = NULL; v->n = v->cap = 0; }

/* Classifier::CalculateHitScore (:243-252) */
static uint64_t hit_score(const ora_index *idx, int l) {
  if (l < idx->param.minHitLen) return 0;
  return (uint64_t)(l - idx->scoreHitLenAdjust) * (uint64_t)(l - idx->scoreHitLenAdjust);
}
static uint64_t hits_score(const ora_index *idx, const ora_hitvec *h) {   /* CalculateHitsScore (:261-271) */
  uint64_t s = 0;
  for (size_t i = 0; i < h->n; ++i) s += hit_score(idx, h->a[i].l);
  return s;
}

/* Classifier::GetHitsFromRead (:274-293) */
size_t ora_get_hits_from_read(const ora_index *idx, const char *r, size_t len, ora_hitvec *hits, ora_counters *c) {
  uint64_t sp = 0, ep = 0;
  int l = 0;
  int remaining = (int)len;
  while (remaining >= idx->param.minHitLen) {
    l = (int)ora_fm_backward_search(&idx->fm, r, (uint64_t)remaining, &sp, &ep, c);
    if (l >= idx->param.minHitLen && sp <= ep) {
      ora_hit nh = {sp, ep, l, 0, (int)len - remaining};
      hit_push(hits, nh);
    }
    remaining -= (l + 1);
  }
  return hits->n;
}

/* Classifier::AdjustHitBoundaryFromStrandHits (:303-401) */
void ora_adjust_hit_boundary(const ora_index *idx, const char *r, const char *rc, int len,
                             ora_hitvec strandHits[2], ora_counters *c) {
  int i, j, k;
  if (!strandHits[0].n || !strandHits[1].n) return;
  int hitSize[2] = {(int)strandHits[0].n, (int)strandHits[1].n};
  uint64_t sp = 0, ep = 0;
  int l;
  j = hitSize[0] - 1;
  int needFix[2] = {0, 0};
  for (i = 0; i < hitSize[1]; ++i) {
    int right = len - strandHits[1].a[i].offset - 1;
    int left = right - strandHits[1].a[i].l + 1;
    for (; j >= 0; --j) {
      int rcLeft = strandHits[0].a[j].offset;
      int rcRight = rcLeft + strandHits[0].a[j].l - 1;
      if (rcLeft >= right) continue;
      if (left >= rcRight) break;
      if (left == rcLeft && right == rcRight) break;
      if (left < rcLeft && rcRight < right) break;
      if (rcLeft < left && right < rcRight) break;
      if (rcRight > right) {
        l = (int)ora_fm_backward_search(&idx->fm, r, (uint64_t)(rcRight + 1), &sp, &ep, c);
        if (rcRight - l + 1 == left && sp <= ep) {
          ora_hit nh = {sp, ep, l, 1, len - rcRight - 1};
          strandHits[1].a[i] = nh;
          needFix[1] = 1;
        }
      }
      if (left < rcLeft) {
        l = (int)ora_fm_backward_search(&idx->fm, rc, (uint64_t)(len - left), &sp, &ep, c);
        if (left + l - 1 == rcRight && sp <= ep) {
          ora_hit nh = {sp, ep, l, -1, left};
          strandHits[0].a[j] = nh;
          needFix[0] = 1;
        }
      }
    }
  }
  for (k = 0; k <= 1; ++k) {
    if (!needFix[k]) continue;
    ora_hit *h = strandHits[k].a;
    for (i = 0; i < hitSize[k] - 1; ++i) {
      int starti = h[i].offset;
      int endi = starti + h[i].l - 1;
      for (j = i + 1; j < hitSize[k]; ++j) {
        int startj = h[j].offset;
        if (startj > endi) break;
        int endj = startj + h[j].l - 1;
        if (h[j].l >= h[i].l) { h[i].l = startj - starti; break; }
        else {
          if (endj <= endi) h[j].l = 0;
          else { h[j].offset = endi + 1; h[j].l = endj - (endi + 1) + 1; break; }
        }
      }
    }
  }
}

/* Classifier::ReverseComplement (:99-111) with _compChar (:846-856) */
static char *revcomp_dup(const char *r, int len) {
  char *rc = xmalloc((size_t)len + 1);
  for (int i = 0; i < len; ++i) {
    char ch = r[len - 1 - i], o;
    switch (ch) { case 'A': o = 'T'; break; case 'C': o = 'G'; break; case 'G': o = 'C'; break; case 'T': o = 'A'; break; default: o = 'N'; }
    rc[i] = o;
  }
  rc[len] = 0;
  return rc;
}

/* Classifier::DnaToAa (:131-241): an if/else ladder, so every character that is not one of the letters it tests takes the
 * last branch of its level (a == anything but A,C,G behaves as T, and so on); only 'N' gives '?' */
char ora_dna_to_aa(char a, char b, char c) {
  if (a == 'N' || b == 'N' || c == 'N') return '?';
  const int ag = (c == 'A' || c == 'G');
  if (a == 'A') {
    if (b == 'A') return ag ? 'K' : 'N';
    if (b == 'C') return 'T';
    if (b == 'G') return ag ? 'R' : 'S';
    return c == 'G' ? 'M' : 'I';
  } else if (a == 'C') {
    if (b == 'A') return ag ? 'Q' : 'H';
    if (b == 'C') return 'P';
    if (b == 'G') return 'R';
    return 'L';
  } else if (a == 'G') {
    if (b == 'A') return ag ? 'E' : 'D';
    if (b == 'C') return 'A';
    if (b == 'G') return 'G';
    return 'V';
  } else {
    if (b == 'A') return ag ? '_' : 'Y';
    if (b == 'C') return 'S';
    if (b == 'G') return c == 'A' ? '_' : (c == 'G' ? 'W' : 'C');
    return ag ? 'L' : 'F';
  }
}
/* Classifier::TranslatedSearch (:463-506): the three frames of r, the frame with the highest "score" wins - where the score
 * of a frame is (number of its hits) x (sum of its hit scores): the loop at :489-493 adds the whole list's score once per hit */
static size_t translated_search(const ora_index *idx, const char *r, int rlen, ora_hitvec *hits, ora_counters *c) {
  char *aa = xmalloc((size_t)rlen + 1);
  ora_hitvec frameHits[3] = {{0}, {0}, {0}};
  for (int frame = 0; frame < 3; ++frame) {
    int k = 0;
    for (int i = frame; i + 2 < rlen; i += 3) aa[k++] = ora_dna_to_aa(r[i], r[i + 1], r[i + 2]);
    aa[k] = 0;
    ora_get_hits_from_read(idx, aa, (size_t)k, &frameHits[frame], c);
  }
  uint64_t maxScore = 0;
  int maxTag = 0;
  for (int frame = 0; frame < 3; ++frame) {
    uint64_t score = 0;
    for (size_t i = 0; i < frameHits[frame].n; ++i) score += hits_score(idx, &frameHits[frame]);
    if (score > maxScore) { maxScore = score; maxTag = frame; }
  }
  size_t ret = frameHits[maxTag].n;
  hit_append(hits, &frameHits[maxTag]);
  for (int frame = 0; frame < 3; ++frame) ora_hitvec_free(&frameHits[frame]);
  free(aa);
  return ret;
}

/* Classifier::SearchForwardAndReverse (:509-583) */
size_t ora_search_forward_and_reverse(const ora_index *idx, const char *r1, const char *r2, ora_hitvec *hits, ora_counters *c) {
  int r1len = (int)strlen(r1);
  char *rcR1 = revcomp_dup(r1, r1len), *rcR2 = NULL;
  ora_hitvec strandHits[2] = {{0}, {0}};
  if (c) c->read_bases += (uint64_t)r1len;
  if (idx->protein) {
    translated_search(idx, r1, r1len, &strandHits[1], c);
    translated_search(idx, rcR1, r1len, &strandHits[0], c);
  } else {
  ora_get_hits_from_read(idx, r1, (size_t)r1len, &strandHits[1], c);
  ora_get_hits_from_read(idx, rcR1, (size_t)r1len, &strandHits[0], c);
  ora_adjust_hit_boundary(idx, r1, rcR1, r1len, strandHits, c);
  }
  if (r2) {
    int r2len = (int)strlen(r2);
    rcR2 = revcomp_dup(r2, r2len);
    if (c) c->read_bases += (uint64_t)r2len;
    ora_hitvec r2Hits[2] = {{0}, {0}};
    if (idx->protein) {
      translated_search(idx, r2, r2len, &r2Hits[1], c);
      translated_search(idx, rcR2, r2len, &r2Hits[0], c);
    } else {
    ora_get_hits_from_read(idx, r2, (size_t)r2len, &r2Hits[1], c);
    ora_get_hits_from_read(idx, rcR2, (size_t)r2len, &r2Hits[0], c);
    ora_adjust_hit_boundary(idx, r2, rcR2, r2len, r2Hits, c);
    }
    for (int i = 0; i <= 1; ++i) hit_append(&strandHits[i], &r2Hits[1 - i]);
    ora_hitvec_free(&r2Hits[0]); ora_hitvec_free(&r2Hits[1]);
  }
  uint64_t strandScore[2];
  for (int k = 0; k < 2; ++k) {
    for (size_t i = 0; i < strandHits[k].n; ++i) strandHits[k].a[i].strand = 2 * k - 1;
    strandScore[k] = hits_score(idx, &strandHits[k]);
  }
  hits->n = 0;
  if (strandScore[1] > strandScore[0] + strandScore[0] / 100) hit_append(hits, &strandHits[1]);
  else if (strandScore[0] > strandScore[1] + strandScore[1] / 100) hit_append(hits, &strandHits[0]);
  else { hit_append(hits, &strandHits[1]); hit_append(hits, &strandHits[0]); }
  free(rcR1); free(rcR2);
  ora_hitvec_free(&strandHits[0]); ora_hitvec_free(&strandHits[1]);
  if (c) c->hits += hits->n;
  return hits->n;
}

/* seqId -> record map (std::map<size_t,_seqHitRecord>): sorted array keyed by seqId */
typedef struct { uint64_t seqId, score; int hitLength; } seq_rec;
typedef struct { seq_rec *a; size_t n, cap; } recmap;
/* operator[]: find or insert a value-initialised (all-zero) record; the KEY is the map key,
 * the record's own seqId field stays 0 until assigned (Classifier.hpp:679-692). */
static seq_rec *rec_get(recmap *m, uint64_t key, int *created) {
  size_t lo = 0, hi = m->n;
  while (lo < hi) { size_t mid = (lo + hi) / 2; if (m->a[mid].seqId < key) lo = mid + 1; else hi = mid; }
  if (lo < m->n && m->a[lo].seqId == key) { if (created) *created = 0; return &m->a[lo]; }
  if (m->n == m->cap) { m->cap = m->cap ? m->cap * 2 : 16; m->a = realloc(m->a, m->cap * sizeof(seq_rec)); if (!m->a) abort(); }
  memmove(m->a + lo + 1, m->a + lo, (m->n - lo) * sizeof(seq_rec));
  m->n++;
  m->a[lo].seqId = key; m->a[lo].score = 0; m->a[lo].hitLength = 0;
  if (created) *created = 1;
  return &m->a[lo];
}

/* Classifier::GetClassificationFromHits (:585-843) */
size_t ora_get_classification_from_hits(const ora_index *idx, const ora_hitvec *hitv, ora_result *result, ora_counters *c) {
  const ora_param *P = &idx->param;
  const ora_hit *hits = hitv->a;
  int i, k;
  uint64_t j;
  int hitCnt = (int)hitv->n;
  recmap rec[2] = {{0}, {0}};
  seq_rec prevUniq = {0, 0, 0};
  int mixStrand = 0;
  for (i = 1; i < hitCnt; ++i) if (hits[i].strand != hits[i - 1].strand) { mixStrand = 1; break; }

  u64vec local = {0};
  for (i = 0; i < hitCnt; ++i) {
    if (hits[i].l < P->minHitLen) continue;
    uint64_t score = hit_score(idx, hits[i].l);
    local.n = 0;
    k = (hits[i].strand + 1) / 2;
    const uint64_t maxEntries = (uint64_t)(int64_t)(P->maxResult * P->maxResultPerHitFactor);   /* int*int -> size_t (:620) */
    if (hits[i].ep - hits[i].sp + 1 <= maxEntries || P->maxResultPerHitFactor <= 0 || P->maxResult <= 0) {
      for (j = hits[i].sp; j <= hits[i].ep; ++j) {
        uint64_t bl = 0;
        set_insert(&local, ora_fm_backward_to_sampled_sa(&idx->fm, j, &bl, c));
      }
    } else {
      uint64_t rangeSize = hits[i].ep - hits[i].sp + 1;
      uint64_t step = DIV_CEIL(rangeSize, maxEntries);
      uint64_t resolvedCnt = 0;
      for (j = hits[i].sp; j <= hits[i].ep; j += step) {
        uint64_t bl = 0;
        set_insert(&local, ora_fm_backward_to_sampled_sa(&idx->fm, j, &bl, c));
        ++resolvedCnt;
      }
      for (j = hits[i].ep; j >= hits[i].sp && j <= hits[i].ep; j -= step) {
        uint64_t bl = 0;
        set_insert(&local, ora_fm_backward_to_sampled_sa(&idx->fm, j, &bl, c));
        ++resolvedCnt;
        if (resolvedCnt >= maxEntries) break;
      }
    }
    for (size_t q = 0; q < local.n; ++q) {
      uint64_t seqId = local.a[q];
      if (!mixStrand && i > 0 && hits[i].ep == hits[i].sp && hits[i - 1].ep == hits[i - 1].sp &&
          hits[i - 1].strand == hits[i].strand &&
          hits[i - 1].offset + hits[i - 1].l + 1 == hits[i].offset && seqId == prevUniq.seqId) {
        seq_rec *r = rec_get(&rec[k], seqId, NULL);
        r->score -= prevUniq.score;
        prevUniq.hitLength += hits[i].l;
        prevUniq.score = hit_score(idx, prevUniq.hitLength);
        r->score += prevUniq.score;
        r->hitLength += hits[i].l;
      } else {
        int created;
        seq_rec *r = rec_get(&rec[k], seqId, &created);
        if (created) { r->score = score; r->hitLength = hits[i].l; }
        else { r->score += score; r->hitLength += hits[i].l; }
        if (hits[i].ep == hits[i].sp) { prevUniq.seqId = seqId; prevUniq.score = score; prevUniq.hitLength = hits[i].l; }
      }
    }
  }
  free(local.a);

  uint64_t bestScore = 0, secondBestScore = 0, bestLen = 0, secondLen = 0;
  for (k = 0; k <= 1; ++k)
    for (size_t q = 0; q < rec[k].n; ++q) {
      const seq_rec *r = &rec[k].a[q];
      if (r->score > bestScore) {
        secondBestScore = bestScore; secondLen = bestLen;
        bestScore = r->score; bestLen = (uint64_t)(int64_t)r->hitLength;
      } else if (r->score > secondBestScore) {
        secondBestScore = r->score; secondLen = (uint64_t)(int64_t)r->hitLength;
      }
    }
  result->score = bestScore;
  result->secondaryScore = secondBestScore;
  result->hitLength = (int32_t)bestLen;

  u64vec best = {0}, used = {0};
  for (k = 0; k <= 1; ++k)
    for (size_t q = 0; q < rec[k].n; ++q)
      if (rec[k].a[q].score == bestScore && !set_has(&used, rec[k].a[q].seqId)) {
        u64_push(&best, rec[k].a[q].seqId); set_insert(&used, rec[k].a[q].seqId);
      }
  if (best.n > 1) result->secondaryScore = bestScore;
  if (secondLen >= P->considerSecondaryHitLen && secondBestScore < bestScore &&
      secondBestScore >= (uint64_t)(P->considerSecondaryScoreFactor * (double)bestScore)) {
    for (k = 0; k <= 1; ++k)
      for (size_t q = 0; q < rec[k].n; ++q)
        if (rec[k].a[q].score == secondBestScore && !set_has(&used, rec[k].a[q].seqId)) {
          u64_push(&best, rec[k].a[q].seqId); set_insert(&used, rec[k].a[q].seqId);
        }
    result->secondaryScore = secondBestScore;
  }

  result->nmatch = 0;
  free(result->expanded); result->expanded = NULL;
  memset(result->expOff, 0, sizeof(result->expOff));
  if ((int)best.n <= P->maxResult || P->maxResult <= 0) {
    if (best.n > ORA_MAX_MATCH) { fprintf(stderr, "oracle: more than %d matches for a read\n", ORA_MAX_MATCH); abort(); }
    for (size_t q = 0; q < best.n; ++q) {
      result->kind[q] = 0; result->id[q] = best.a[q];
      result->taxid[q] = tax_orig(&idx->tax, tax_seq_to_tax(&idx->tax, best.a[q]));
    }                                                      /* (:792-795: an empty expanded string per sequence-level match) */
    result->nmatch = (int32_t)best.n;
  } else {
    uint64_t *tids = xmalloc(best.n * 8);
    for (size_t q = 0; q < best.n; ++q) tids[q] = tax_seq_to_tax(&idx->tax, best.a[q]);
    uint64_t out[ORA_MAX_MATCH];
    u64vec child[ORA_MAX_MATCH];
    int nchild = 0;
    memset(child, 0, sizeof(child));
    int n = tax_reduce(&idx->tax, tids, (int)best.n, P->maxResult, out, ORA_MAX_MATCH, P->outputExpandedResult ? child : NULL, &nchild);
    for (int q = 0; q < n; ++q) {
      result->kind[q] = 1; result->id[q] = out[q];
      result->taxid[q] = tax_orig(&idx->tax, out[q]);
    }
    if (P->outputExpandedResult && nchild == n) {          /* :821-839: only when one list per promoted id came back */
      size_t total = 0;
      for (int q = 0; q < n; ++q) total += child[q].n;
      result->expanded = xmalloc((total ? total : 1) * 8);
      size_t at = 0;
      for (int q = 0; q < n; ++q) {
        result->expOff[q] = (int32_t)at;
        for (size_t j = 0; j < child[q].n; ++j) result->expanded[at++] = tax_orig(&idx->tax, child[q].a[j]);   /* GetOrigTaxId (:833) */
      }
      for (int q = n; q <= ORA_MAX_MATCH; ++q) result->expOff[q] = (int32_t)at;
    }
    for (int q = 0; q < ORA_MAX_MATCH; ++q) free(child[q].a);
    result->nmatch = n;
    free(tids);
  }
  free(best.a); free(used.a); free(rec[0].a); free(rec[1].a);
  return (size_t)result->nmatch;
}

/* Classifier::Query (:950-961) */
void ora_query(const ora_index *idx, const char *r1, const char *r2, ora_result *res, ora_counters *c) {
  ora_hitvec hits = {0};
  memset(res, 0, sizeof(*res));
  ora_search_forward_and_reverse(idx, r1, r2, &hits, c);
  ora_get_classification_from_hits(idx, &hits, res, c);
  res->queryLength = (int32_t)strlen(r1);
  if (r2) res->queryLength += (int32_t)strlen(r2);
  ora_hitvec_free(&hits);
}

size_t ora_query_hits(const ora_index *idx, const char *r1, const char *r2, ora_hit *out, size_t cap) {
  ora_hitvec hits = {0};
  ora_search_forward_and_reverse(idx, r1, r2, &hits, NULL);
  size_t n = hits.n;
  for (size_t i = 0; i < n && i < cap; ++i) out[i] = hits.a[i];
  ora_hitvec_free(&hits);
  return n;
}

/* ======================================================================== SDUST */
/* Dustmasker (Dustmasker.hpp), parameters w=64, T=20, linker=1 (:247-249), alphabet "ACGT" */

typedef struct { uint64_t start, end; int score; } dust_iv;
typedef struct { dust_iv *a; size_t n, cap; } ivvec;
static void iv_push(ivvec *v, dust_iv x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 8; v->a = realloc(v->a, v->cap * sizeof(dust_iv)); if (!v->a) abort(); }
  v->a[v->n++] = x;
}
static void iv_insert(ivvec *v, size_t pos, dust_iv x) {
  iv_push(v, x);
  memmove(v->a + pos + 1, v->a + pos, (v->n - 1 - pos) * sizeof(dust_iv));
  v->a[pos] = x;
}
typedef struct { int head, tail, mask; int s[128]; } dust_queue;   /* Dustmasker_Queue (:33-90), capacity 128 for w=64 */
static inline int dq_size(const dust_queue *q) { return (q->tail - q->head) & q->mask; }
static inline void dq_push(dust_queue *q, int t) { q->s[q->tail] = t; q->tail = (q->tail + 1) & q->mask; }
static inline int dq_pop(dust_queue *q) { int t = q->s[q->head]; q->head = (q->head + 1) & q->mask; return t; }
static inline int dq_at(const dust_queue *q, int i) { return q->s[(q->head + i) & q->mask]; }

enum { DUST_W = 64, DUST_T = 20, DUST_ABIT = 3, DUST_ASIZE = 5 };
static inline int dust_code(char ch) { switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; } }

static inline void dust_add(int t, int *count, int *r) { *r += count[t]; ++count[t]; }       /* :93-97 */
static inline void dust_rem(int t, int *count, int *r) { --count[t]; *r -= count[t]; }      /* :99-103 */

/* ShiftWindow (:106-138) */
static void dust_shift(int t, dust_queue *w, int *lv, int *rw, int *rv, int *cw, int *cv) {
  if (dq_size(w) >= DUST_W - 2) {
    int old = w->s[w->head];
    dust_rem(old, cw, rw);
    dq_pop(w);
    if (*lv > dq_size(w)) { dust_rem(old, cv, rv); --*lv; }
  }
  dq_push(w, t);
  ++*lv;
  dust_add(t, cw, rw);
  dust_add(t, cv, rv);
  if (cv[t] * 10 > 2 * DUST_T) {
    for (;;) {
      int s = dq_at(w, dq_size(w) - *lv);
      dust_rem(s, cv, rv);
      --*lv;
      if (s == t) break;
    }
  }
}
/* SaveMaskedRegions (:141-167) */
static void dust_save(ivvec *result, ivvec *P, uint64_t windowStart) {
  if (P->n > 0 && P->a[P->n - 1].start < windowStart) {
    dust_iv lastP = P->a[P->n - 1];
    size_t l = result->n;
    if (l > 0) {
      if (lastP.start <= result->a[l - 1].end + 1) {
        if (lastP.end > result->a[l - 1].end) result->a[l - 1].end = lastP.end;
      } else iv_push(result, lastP);
    } else iv_push(result, lastP);
    while (P->n > 0 && P->a[P->n - 1].start < windowStart) P->n--;
  }
}
/* FindPerfect (:172-243) */
static void dust_find_perfect(ivvec *P, dust_queue *w, uint64_t windowStart, int lv, int rv, int *cv) {
  int i;
  int maxScore = 0, maxScoreTripletCount = 1;
  size_t it;
  for (i = dq_size(w) - lv - 1; i >= 0; --i) {
    int t = dq_at(w, i);
    dust_add(t, cv, &rv);
    it = 0;   /* std::vector::iterator it = P.begin() (:188) */
    if (rv * 10 > DUST_T * (dq_size(w) - i - 1)) {
      while (it != P->n && P->a[it].start >= (uint64_t)i + windowStart) {
        if ((uint64_t)(int64_t)P->a[it].score * (uint64_t)(int64_t)maxScoreTripletCount >
            (uint64_t)(int64_t)maxScore * (P->a[it].end - P->a[it].start - 2)) {
          maxScore = P->a[it].score;
          maxScoreTripletCount = (int)(P->a[it].end - P->a[it].start - 2);
        }
        ++it;
      }
      if (rv * maxScoreTripletCount >= maxScore * (dq_size(w) - i - 1)) {
        maxScore = rv;
        maxScoreTripletCount = dq_size(w) - i - 1;
        dust_iv nv = {(uint64_t)i + windowStart, windowStart + (uint64_t)dq_size(w) + 1, rv};
        iv_insert(P, it, nv);
      }
    }
  }
  for (i = dq_size(w) - lv - 1; i >= 0; --i) dust_rem(dq_at(w, i), cv, &rv);
}
/* SDust (:312-354) */
static void dust_sdust(const char *S, size_t n, ivvec *result) {
  if (n < 3) return;
  size_t wstart, wfinish;
  int triplet;
  const int tripletMask = (1 << (3 * DUST_ABIT)) - 1;
  int countV[512], countW[512];
  memset(countV, 0, sizeof(countV)); memset(countW, 0, sizeof(countW));
  int rv = 0, rw = 0, lv = 0;
  dust_queue window; window.head = window.tail = 0; window.mask = 127;
  ivvec P = {0};
  triplet = (dust_code(S[0]) << DUST_ABIT) + dust_code(S[1]);
  for (wfinish = 2; wfinish < n; ++wfinish) {
    wstart = 0;
    if (wfinish + 1 > (size_t)DUST_W) wstart = wfinish + 1 - DUST_W;
    dust_save(result, &P, wstart);
    triplet = ((triplet << DUST_ABIT) & tripletMask) + dust_code(S[wfinish]);
    dust_shift(triplet, &window, &lv, &rw, &rv, countW, countV);
    if (rw * 10 > lv * DUST_T) dust_find_perfect(&P, &window, wstart, lv, rv, countV);
  }
  wstart = 0;
  if (wfinish + 1 > (size_t)DUST_W) wstart = wfinish + 1 - DUST_W;
  while (P.n > 0) { dust_save(result, &P, wstart); ++wstart; }
  free(P.a);
}
/* MaskWithBuffer (:357-421) + the overwrite loop of ClassifyReads_Thread (CentrifugerClass.cpp:283-289) */
void ora_dust_mask_inplace(char *S, size_t n) {
  size_t i, j;
  if (n < 3) return;
  ivvec result = {0}, win = {0};
  for (i = 0; i < n && dust_code(S[i]) == DUST_ASIZE - 1; ++i) ;
  for (; i < n;) {
    size_t nCount = 0, lastValidPos = i;
    for (j = i; j < n; ++j) {
      if (dust_code(S[j]) == DUST_ASIZE - 1) ++nCount;
      else { if (nCount > (size_t)DUST_W) break; lastValidPos = j; nCount = 0; }
    }
    if (lastValidPos > i) {
      win.n = 0;
      dust_sdust(S + i, lastValidPos - i + 1, &win);
      for (size_t k = 0; k < win.n; ++k) { dust_iv v = win.a[k]; v.start += i; v.end += i; iv_push(&result, v); }
    }
    i = j;
  }
  /* linker _l == 1: the merge block (:400-416) is skipped */
  for (size_t k = 0; k < result.n; ++k)
    for (uint64_t p = result.a[k].start; p <= result.a[k].end; ++p) S[p] = 'N';
  free(result.a); free(win.a);
}

/* ======================================================================== batch + TSV */

typedef struct {
  const ora_index *idx;
  const uint8_t *b1, *b2;
  const uint64_t *o1, *o2;
  size_t n;
  int dust, tid, nthreads;
  ora_result *res;
  ora_counters cnt;
} batch_arg;

static void *batch_thread(void *p) {   /* ClassifyReads_Thread (CentrifugerClass.cpp:240-340) */
  batch_arg *a = p;
  size_t cap = 1024;
  char *s1 = xmalloc(cap), *s2 = xmalloc(cap);
  for (size_t i = 0; i < a->n; ++i) {
    if ((int)(i % (size_t)a->nthreads) != a->tid) continue;
    size_t l1 = a->o1[i + 1] - a->o1[i], l2 = a->b2 ? a->o2[i + 1] - a->o2[i] : 0;
    if (l1 + 1 > cap || l2 + 1 > cap) {
      cap = (l1 > l2 ? l1 : l2) * 2 + 2;
      s1 = realloc(s1, cap); s2 = realloc(s2, cap);
      if (!s1 || !s2) abort();
    }
    memcpy(s1, a->b1 + a->o1[i], l1); s1[l1] = 0;
    if (a->b2) { memcpy(s2, a->b2 + a->o2[i], l2); s2[l2] = 0; }
    if (a->dust && !a->idx->protein) { ora_dust_mask_inplace(s1, l1); if (a->b2) ora_dust_mask_inplace(s2, l2); }   /* CentrifugerClass.cpp:276 */
    ora_query(a->idx, s1, a->b2 ? s2 : NULL, &a->res[i], &a->cnt);
  }
  free(s1); free(s2);
  return NULL;
}

void ora_classify_batch(const ora_index *idx, const uint8_t *bases1, const uint64_t *offs1,
                        const uint8_t *bases2, const uint64_t *offs2, size_t nreads,
                        int dust, int nthreads, ora_result *results, ora_counters *total) {
  if (nthreads < 1) nthreads = 1;
  batch_arg *args = xcalloc((size_t)nthreads, sizeof(batch_arg));
  pthread_t *th = xcalloc((size_t)nthreads, sizeof(pthread_t));
  for (int t = 0; t < nthreads; ++t) {
    batch_arg a = {idx, bases1, bases2, offs1, offs2, nreads, dust, t, nthreads, results, {0}};
    args[t] = a;
    if (nthreads == 1) batch_thread(&args[t]);
    else pthread_create(&th[t], NULL, batch_thread, &args[t]);
  }
  if (nthreads > 1) for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  if (total) {
    memset(total, 0, sizeof(*total));
    for (int t = 0; t < nthreads; ++t) {
      const uint64_t *s = (const uint64_t *)&args[t].cnt;
      uint64_t *d = (uint64_t *)total;
      for (size_t q = 0; q < sizeof(ora_counters) / 8; ++q) d[q] += s[q];
    }
  }
  free(args); free(th);
}

const char *ora_tsv_header(void) {   /* ResultWriter::OutputHeader (ResultWriter.hpp:186-197) */
  return "readID\tseqID\ttaxID\tscore\t2ndBestScore\thitLength\tqueryLength\tnumMatches\n";
}
const char *ora_tsv_header_for(const ora_index *idx) {   /* ... with the expandedTaxIDs column of --expand-taxid (:194-195) */
  return idx->param.outputExpandedResult
           ? "readID\tseqID\ttaxID\tscore\t2ndBestScore\thitLength\tqueryLength\tnumMatches\texpandedTaxIDs\n" : ora_tsv_header();
}
void ora_results_free(ora_result *r, size_t n) {
  for (size_t i = 0; i < n; ++i) { free(r[i].expanded); r[i].expanded = NULL; }
}
/* ResultWriter::Output (ResultWriter.hpp:199-242) */
size_t ora_format_result(const ora_index *idx, const char *readid, const ora_result *r, char *buf, size_t cap) {
  size_t off = 0;
  const int expand = idx->param.outputExpandedResult;
#define ORA_PUT(...) { int w_ = snprintf(buf ? buf + off : NULL, buf && cap > off ? cap - off : 0, __VA_ARGS__); off += (size_t)w_; }
  if (r->nmatch > 0) {
    for (int i = 0; i < r->nmatch; ++i) {
      const char *name = r->kind[i] == 0 ? idx->tax.seqName[r->id[i]] : ora_tax_rank_string(tax_rank(&idx->tax, r->id[i]));
      ORA_PUT("%s\t%s\t%lu\t%lu\t%lu\t%d\t%d\t%d", readid, name, (unsigned long)r->taxid[i], (unsigned long)r->score,
              (unsigned long)r->secondaryScore, r->hitLength, r->queryLength, r->nmatch);
      if (expand) {                                        /* PrintExtraCol(expandedTaxIdStrings[i]) (:226-227) */
        ORA_PUT("\t");
        if (r->expanded)
          for (int32_t j = r->expOff[i]; j < r->expOff[i + 1]; ++j) ORA_PUT(j == r->expOff[i] ? "%lu" : ",%lu", (unsigned long)r->expanded[j]);
      }
      ORA_PUT("\n");
    }
  } else {
    ORA_PUT("%s\tunclassified\t0\t0\t0\t0\t%d\t1", readid, r->queryLength);
    if (expand) ORA_PUT("\t");
    ORA_PUT("\n");
  }
#undef ORA_PUT
  return off;
}
