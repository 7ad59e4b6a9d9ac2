// ref_dump.cpp — TEST INFRASTRUCTURE.  Our own driver around the REAL reference headers
// (included in place from /root/reference via -I; nothing is copied).  Emits the same
// intermediate-vector streams as `cfr_oracle dump-*` so the C restatement can be pinned
// below the TSV level: FMIndex::Rank / Sequence_RunBlock::Access, BackwardSearch tuples,
// BackwardToSampledSA values.  Built only where /root/reference exists (oracle/Makefile).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <map>

char nucToNum[26] ;
char numToNuc[26] ;

#include "compactds/FMIndex.hpp"
#include "compactds/Sequence_RunBlock.hpp"
#include "compactds/Sequence_RunBlockOneTree.hpp"

using namespace compactds ;

static std::vector<std::string> ReadSeqs(const char *path)
{
  std::vector<std::string> out ;
  FILE *fp = fopen(path, "r") ;
  if (!fp) { fprintf(stderr, "cannot open %s\n", path) ; exit(1) ; }
  char *line = NULL ; size_t cap = 0 ; ssize_t n ;
  int state = 0 ; bool fastq = false ;
  while ((n = getline(&line, &cap, fp)) >= 0)
  {
    while (n > 0 && (line[n-1] == '\n' || line[n-1] == '\r')) line[--n] = 0 ;
    if (state == 0 && (line[0] == '>' || line[0] == '@')) { fastq = line[0] == '@' ; state = 1 ; out.push_back("") ; }
    else if (state == 1) { out.back() += line ; if (fastq) state = 2 ; else state = 0 ; }
    else if (state == 2) state = 3 ;
    else if (state == 3) state = 0 ;
  }
  free(line) ; fclose(fp) ;
  return out ;
}

int main(int argc, char *argv[])
{
  if (argc < 3) { fprintf(stderr, "usage: ref_dump <rank|locate|bs|prank|plocate> idx.1.cfr [step|reads]\n") ; return 1 ; }
  if (argv[1][0] == 'p')     // protein index: FMIndex<Sequence_RunBlockOneTree> (CentrifugerClass.cpp:1001-1004)
  {
    FMIndex<Sequence_RunBlockOneTree> pfm ;
    FILE *pfp = fopen(argv[2], "r") ;
    if (!pfp) { fprintf(stderr, "cannot open %s\n", argv[2]) ; return 1 ; }
    pfm.Load(pfp) ;
    fclose(pfp) ;
    const size_t pn = pfm.GetSize() ;
    const char *PA = "$ARNDCEQGHILKMFPSTWYV" ;
    const size_t step = argc > 3 ? strtoull(argv[3], NULL, 10) : 1 ;
    if (!strcmp(argv[1], "prank"))
      for (size_t i = 0 ; i < pn ; i += step)
      {
        printf("%lu %c", i, pfm.GetBWT()->Access(i)) ;
        for (int inc = 1 ; inc >= 0 ; --inc)
          for (int c = 0 ; c < 21 ; ++c)
            printf(" %lu", pfm.Rank(PA[c], i, inc)) ;
        printf("\n") ;
      }
    else
      for (size_t i = 0 ; i < pn ; i += step)
      {
        size_t l ;
        size_t v = pfm.BackwardToSampledSA(i, l) ;
        printf("%lu %lu %lu\n", i, v, l) ;
      }
    return 0 ;
  }
  FMIndex<Sequence_RunBlock> fm ;
  FILE *fp = fopen(argv[2], "r") ;
  if (!fp) { fprintf(stderr, "cannot open %s\n", argv[2]) ; return 1 ; }
  fm.Load(fp) ;
  fclose(fp) ;
  size_t n = fm.GetSize() ;
  const char *A = "ACGT" ;
  if (!strcmp(argv[1], "rank"))
  {
    size_t step = argc > 3 ? strtoull(argv[3], NULL, 10) : 1 ;
    for (size_t i = 0 ; i < n ; i += step)
    {
      printf("%lu %c", i, fm.GetBWT()->Access(i)) ;
      for (int inc = 1 ; inc >= 0 ; --inc)
        for (int c = 0 ; c < 4 ; ++c)
          printf(" %lu", fm.Rank(A[c], i, inc)) ;
      printf("\n") ;
    }
  }
  else if (!strcmp(argv[1], "locate"))
  {
    size_t step = argc > 3 ? strtoull(argv[3], NULL, 10) : 1 ;
    for (size_t i = 0 ; i < n ; i += step)
    {
      size_t l ;
      size_t v = fm.BackwardToSampledSA(i, l) ;
      printf("%lu %lu %lu\n", i, v, l) ;
    }
  }
  else if (!strcmp(argv[1], "bs"))
  {
    std::vector<std::string> reads = ReadSeqs(argv[3]) ;
    for (size_t i = 0 ; i < reads.size() ; ++i)
    {
      size_t len = reads[i].length() ;
      char *s = strdup(reads[i].c_str()) ;
      for (size_t m = len ; m > 0 ; m = m > 13 ? m - 13 : 0)
      {
        size_t sp = 7, ep = 3 ;
        size_t l = fm.BackwardSearch(s, m, sp, ep) ;
        printf("%lu %lu %lu %lu %lu\n", i, m, l, sp, ep) ;
      }
      free(s) ;
    }
  }
  return 0 ;
}
