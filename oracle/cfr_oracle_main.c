/*
 * cfr_oracle_main.c — TEST INFRASTRUCTURE: command-line front end of the CPU restatement.
 *
 *   cfr_oracle classify -x IDX (-u R | -1 R1 -2 R2) [-k K] [-t T] [--no-dust] [--min-hitlen N] [--hitk-factor N] [--expand-taxid]
 *        -> TSV on stdout, same bytes as the reference `centrifuger` (ResultWriter.hpp:186-242)
 *   cfr_oracle dump-rank   -x IDX [--step S]     -> "i acc rA rC rG rT eA eC eG eT" (FMIndex::Rank incl/excl, Access)
 *   cfr_oracle dump-bs     -x IDX -u R           -> per read/strand/prefix: "l sp ep"
 *   cfr_oracle dump-locate -x IDX [--step S]     -> "row seqId steps"
 *   cfr_oracle counters    -x IDX -u R [...]     -> operation counters (algorithmic bytes)
 * The dump formats are mirrored by oracle/ref_dump.cpp, which drives the REAL reference headers.
 */
#include "cfr_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef struct { char **id; uint8_t *bases; uint64_t *offs; size_t n, cap, bcap; } readset;

static void rs_add(readset *rs, const char *id, const char *seq, size_t len) {
  if (rs->n + 2 > rs->cap) {
    rs->cap = rs->cap ? rs->cap * 2 : 1024;
    rs->id = realloc(rs->id, rs->cap * sizeof(char *));
    rs->offs = realloc(rs->offs, (rs->cap + 1) * 8);
    if (rs->n == 0) rs->offs[0] = 0;
  }
  uint64_t o = rs->offs[rs->n];
  if (o + len + 1 > rs->bcap) { rs->bcap = (o + len + 1) * 2; rs->bases = realloc(rs->bases, rs->bcap); }
  memcpy(rs->bases + o, seq, len);
  rs->id[rs->n] = strdup(id);
  rs->offs[++rs->n] = o + len;
}

/* FASTA/FASTQ reader; id = first word, trailing /1 /2 stripped (ReadFiles.hpp:82-90) */
static int read_file(const char *path, readset *rs) {
  FILE *fp = fopen(path, "r");
  if (!fp) { fprintf(stderr, "cannot open %s\n", path); return -1; }
  char *line = NULL; size_t lcap = 0; ssize_t n;
  char *seq = NULL; size_t scap = 0, slen = 0;
  char id[4096]; int have = 0, fastq = 0;
  while ((n = getline(&line, &lcap, fp)) >= 0) {
    while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
    if (line[0] == '>' || (line[0] == '@' && !have)) {
      if (have) rs_add(rs, id, seq ? seq : "", slen);
      fastq = line[0] == '@';
      size_t k = 0;
      while (line[1 + k] && line[1 + k] != ' ' && line[1 + k] != '\t' && k < sizeof(id) - 1) { id[k] = line[1 + k]; ++k; }
      id[k] = 0;
      if (k >= 2 && id[k - 2] == '/' && (id[k - 1] == '1' || id[k - 1] == '2')) id[k - 2] = 0;
      have = 1; slen = 0;
      if (fastq) {   /* 4-line records */
        if ((n = getline(&line, &lcap, fp)) < 0) break;
        while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
        if ((size_t)n + 1 > scap) { scap = (size_t)n * 2 + 16; seq = realloc(seq, scap); }
        memcpy(seq, line, (size_t)n + 1); slen = (size_t)n;
        if (getline(&line, &lcap, fp) < 0) break;   /* + */
        if (getline(&line, &lcap, fp) < 0) break;   /* qual */
        rs_add(rs, id, seq, slen);
        have = 0;
      }
    } else if (have) {
      if (slen + (size_t)n + 1 > scap) { scap = (slen + (size_t)n) * 2 + 16; seq = realloc(seq, scap); }
      memcpy(seq + slen, line, (size_t)n + 1); slen += (size_t)n;
    }
  }
  if (have) rs_add(rs, id, seq ? seq : "", slen);
  free(line); free(seq); fclose(fp);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: cfr_oracle <classify|dump-rank|dump-bs|dump-locate|counters> ...\n"); return 1; }
  const char *cmd = argv[1], *idxp = NULL, *u = NULL, *m1 = NULL, *m2 = NULL;
  ora_param P; ora_param_default(&P);
  int threads = 1, dust = 1; uint64_t step = 1;
  for (int i = 2; i < argc; ++i) {
    if (!strcmp(argv[i], "-x")) idxp = argv[++i];
    else if (!strcmp(argv[i], "-u")) u = argv[++i];
    else if (!strcmp(argv[i], "-1")) m1 = argv[++i];
    else if (!strcmp(argv[i], "-2")) m2 = argv[++i];
    else if (!strcmp(argv[i], "-k")) P.maxResult = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-t")) threads = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--no-dust")) dust = 0;
    else if (!strcmp(argv[i], "--expand-taxid")) P.outputExpandedResult = 1;
    else if (!strcmp(argv[i], "--min-hitlen")) P.minHitLen = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--hitk-factor")) P.maxResultPerHitFactor = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--step")) step = strtoull(argv[++i], NULL, 10);
    else { fprintf(stderr, "unknown option %s\n", argv[i]); return 1; }
  }
  if (!idxp) { fprintf(stderr, "need -x\n"); return 1; }
  ora_index *idx = ora_index_load(idxp, &P);
  if (!idx) return 1;
  const ora_fm *fm = &idx->fm;

  if (!strcmp(cmd, "dump-rank")) {
    const char *A = idx->protein ? "$ARNDCEQGHILKMFPSTWYV" : "ACGT";      /* protein: ref_dump's prank stream */
    const int na = (int)strlen(A);
    for (uint64_t i = 0; i < fm->n; i += step) {
      printf("%lu %c", (unsigned long)i, ora_fm_access(fm, i, NULL));
      for (int inc = 1; inc >= 0; --inc)
        for (int c = 0; c < na; ++c) printf(" %lu", (unsigned long)ora_fm_rank(fm, A[c], i, inc, NULL));
      printf("\n");
    }
    return 0;
  }
  if (!strcmp(cmd, "dump-locate")) {
    for (uint64_t i = 0; i < fm->n; i += step) {
      uint64_t l, v = ora_fm_backward_to_sampled_sa(fm, i, &l, NULL);
      printf("%lu %lu %lu\n", (unsigned long)i, (unsigned long)v, (unsigned long)l);
    }
    return 0;
  }

  readset r1 = {0}, r2 = {0};
  if (u) { if (read_file(u, &r1)) return 1; }
  else if (m1 && m2) { if (read_file(m1, &r1) || read_file(m2, &r2)) return 1; }
  else { fprintf(stderr, "need -u or -1/-2\n"); return 1; }
  if (m2 && r1.n != r2.n) { fprintf(stderr, "mate files differ in length\n"); return 1; }

  if (!strcmp(cmd, "dump-bs")) {
    for (size_t i = 0; i < r1.n; ++i) {
      size_t len = r1.offs[i + 1] - r1.offs[i];
      char *s = malloc(len + 1); memcpy(s, r1.bases + r1.offs[i], len); s[len] = 0;
      for (size_t m = len; m > 0; m = m > 13 ? m - 13 : 0) {
        uint64_t sp = 7, ep = 3;
        uint64_t l = ora_fm_backward_search(fm, s, m, &sp, &ep, NULL);
        printf("%zu %zu %lu %lu %lu\n", i, m, (unsigned long)l, (unsigned long)sp, (unsigned long)ep);
      }
      free(s);
    }
    return 0;
  }

  ora_result *res = calloc(r1.n ? r1.n : 1, sizeof(ora_result));
  ora_counters cnt;
  ora_classify_batch(idx, r1.bases, r1.offs, m2 ? r2.bases : NULL, m2 ? r2.offs : NULL, r1.n, dust, threads, res, &cnt);
  if (!strcmp(cmd, "classify")) {
    fputs(ora_tsv_header_for(idx), stdout);
    char buf[1 << 16];
    for (size_t i = 0; i < r1.n; ++i) {
      size_t w = ora_format_result(idx, r1.id[i], &res[i], buf, sizeof(buf));
      fwrite(buf, 1, w, stdout);
    }
  } else if (!strcmp(cmd, "counters")) {
    printf("reads %zu\nbitrank %lu\nbitaccess %lu\nftab %lu\nsampled %lu\nfilter %lu\nhits %lu\nbs_calls %lu\nextends %lu\nlf_steps %lu\nlocates %lu\nread_bases %lu\n",
           r1.n, (unsigned long)cnt.bitrank, (unsigned long)cnt.bitaccess, (unsigned long)cnt.ftab, (unsigned long)cnt.sampled,
           (unsigned long)cnt.filter, (unsigned long)cnt.hits, (unsigned long)cnt.bs_calls, (unsigned long)cnt.extends,
           (unsigned long)cnt.lf_steps, (unsigned long)cnt.locates, (unsigned long)cnt.read_bases);
  } else { fprintf(stderr, "unknown command %s\n", cmd); return 1; }
  ora_results_free(res, r1.n);
  free(res);
  ora_index_free(idx);
  return 0;
}
