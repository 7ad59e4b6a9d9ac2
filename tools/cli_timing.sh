#!/bin/bash
export CFR_DEBUG_ENV=1   # the gate behind which the library reads its CFR_* A/B switches
# Stage timing of the drop-in command line on the bench sample (run on the GPU box; fills the bench cache first).
mkdir -p gpurun_out
python bench.py --steps 2 --warmup 1 --no-pmc --no-extra-configs > gpurun_out/cli_bench.json 2> gpurun_out/cli_bench.err
idx=$(ls /tmp/cfr_bench/*/idx.1.cfr 2>/dev/null | head -1); idx=${idx%.1.cfr}
fa=$(ls /tmp/cfr_bench/*/sample_0.fa 2>/dev/null | head -1)
out=gpurun_out/cli_timing.txt
echo "index $idx reads $fa ($(grep -c '>' $fa) reads)" > $out
big=/tmp/big10m.fa
for i in 1 2 3 4 5; do cat $fa; done > $big
for prof in "" "--gpu-balanced" "--gpu-throughput"; do
  for pt in 0 1; do
    echo "== 10 M reads, -t 64 $prof --parse-threads $pt" | tee -a $out
    ( time CFR_CLI_TIMING=1 centrifuger_amd/bin/centrifuger -x $idx -u $big -t 64 $prof --parse-threads $pt > /tmp/cli_big.tsv ) 2>&1 | grep -E "timing|real" | tee -a $out
    md5sum /tmp/cli_big.tsv | tee -a $out
  done
done
echo "== 2 M reads, -t 64 (default profile)" | tee -a $out
( time CFR_CLI_TIMING=1 centrifuger_amd/bin/centrifuger -x $idx -u $fa -t 64 > /tmp/cli.tsv ) 2>&1 | grep -E "timing|real" | tee -a $out
md5sum /tmp/cli.tsv | tee -a $out
if [ -x oracle/_ref/centrifuger ]; then
  echo "== 10 M reads, reference -t 64" | tee -a $out
  ( time oracle/_ref/centrifuger -x $idx -u $big -t 64 > /tmp/ref_big.tsv 2>/dev/null ) 2>&1 | grep real | tee -a $out
  md5sum /tmp/ref_big.tsv | tee -a $out
fi
# ---- compressed input: the same 10 M reads as fastq.gz (one deflate stream: the inflate thread bounds the run; the reference reads it through kseq + gzread)
fq=/tmp/big10m.fq.gz
awk 'NR % 2 == 1 { print "@" substr($0, 2) } NR % 2 == 0 { print; print "+"; q = $0; gsub(/./, "I", q); print q }' $big | gzip -1 > $fq
echo "== 10 M reads as fastq.gz ($(du -h $fq | cut -f1)), -t 64" | tee -a $out
( time CFR_CLI_TIMING=1 centrifuger_amd/bin/centrifuger -x $idx -u $fq -t 64 > /tmp/cli_gz.tsv ) 2>&1 | grep -E "timing|real" | tee -a $out
md5sum /tmp/cli_gz.tsv | tee -a $out
fq2=/tmp/sample2m.fq.gz
awk 'NR % 2 == 1 { print "@" substr($0, 2) } NR % 2 == 0 { print; print "+"; q = $0; gsub(/./, "I", q); print q }' $fa | gzip -1 > $fq2
if [ -x oracle/_ref/centrifuger ]; then
  echo "== 2 M reads as fastq.gz: reference -t 64 vs this command line" | tee -a $out
  ( time oracle/_ref/centrifuger -x $idx -u $fq2 -t 64 > /tmp/ref_gz.tsv 2>/dev/null ) 2>&1 | grep real | tee -a $out
  ( time centrifuger_amd/bin/centrifuger -x $idx -u $fq2 -t 64 > /tmp/own_gz.tsv 2>/dev/null ) 2>&1 | grep real | tee -a $out
  md5sum /tmp/ref_gz.tsv /tmp/own_gz.tsv | tee -a $out
fi
# ---- pairs: 10 M pairs (-1/-2, -k 5) and the same as one interleaved file; plain files are cut at equal record numbers and parsed in pieces
python bench.py --mode pe --steps 1 --warmup 1 --no-pmc --no-extra-configs > gpurun_out/cli_bench_pe.json 2> gpurun_out/cli_bench_pe.err
f1=$(ls /tmp/cfr_bench/*/sample_0.fa 2>/dev/null | head -1); f2=${f1%.fa}_2.fa
if [ -f "$f2" ]; then
  for i in 1 2 3 4 5; do cat $f1; done > /tmp/big10m_1.fa
  for i in 1 2 3 4 5; do cat $f2; done > /tmp/big10m_2.fa
  for pt in 0 1; do
    echo "== 10 M pairs -k 5, -t 64 --parse-threads $pt" | tee -a $out
    ( time CFR_CLI_TIMING=1 centrifuger_amd/bin/centrifuger -x $idx -1 /tmp/big10m_1.fa -2 /tmp/big10m_2.fa -k 5 -t 64 --parse-threads $pt > /tmp/cli_pe.tsv ) 2>&1 | grep -E "timing|real" | tee -a $out
    md5sum /tmp/cli_pe.tsv | tee -a $out
  done
  if [ -x oracle/_ref/centrifuger ]; then
    echo "== 2 M pairs -k 5: reference -t 64 vs this command line" | tee -a $out
    ( time oracle/_ref/centrifuger -x $idx -1 $f1 -2 $f2 -k 5 -t 64 > /tmp/ref_pe.tsv 2>/dev/null ) 2>&1 | grep real | tee -a $out
    ( time centrifuger_amd/bin/centrifuger -x $idx -1 $f1 -2 $f2 -k 5 -t 64 > /tmp/own_pe.tsv 2>/dev/null ) 2>&1 | grep real | tee -a $out
    md5sum /tmp/ref_pe.tsv /tmp/own_pe.tsv | tee -a $out
  fi
fi
