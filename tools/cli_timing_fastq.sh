#!/bin/bash
# End to end on FASTQ (what sequencers write): M million x 150 bp reads as plain FASTQ through the drop-in command line, wall clock including
# process start, index load and device image, for several --parse-threads; the reference's rows of the 2 M-read sample give the md5.
# Run on the GPU box:  tools/cli_timing_fastq.sh [reads in millions, default 100]   -> gpurun_out/cli_timing_fastq.txt
export CFR_DEBUG_ENV=1
M=${1:-100}
mkdir -p gpurun_out
python bench.py --steps 2 --warmup 1 --no-pmc --no-extra-configs --no-40gbp > gpurun_out/clifq_bench.json 2> gpurun_out/clifq_bench.err
idx=$(ls /tmp/cfr_bench/*/idx.1.cfr 2>/dev/null | head -1); idx=${idx%.1.cfr}
fa=$(ls /tmp/cfr_bench/*/sample_0.fa 2>/dev/null | head -1)
out=gpurun_out/cli_timing_fastq.txt
n2=$(grep -c '>' $fa)
reps=$(( M * 1000000 / n2 ))
awk 'NR % 2 == 1 { print "@" substr($0, 2) } NR % 2 == 0 { print; print "+"; q = $0; gsub(/./, "I", q); print q }' $fa > /tmp/sample.fq
big=/tmp/big.fq
for i in $(seq $reps); do cat /tmp/sample.fq; done > $big
echo "index $idx; $n2 reads x $reps = $(( reps * n2 )) reads as FASTQ: $(du -h $big | cut -f1)" | tee $out
run() {
  local label=$1; shift
  local t0=$(date +%s.%N)
  CFR_CLI_TIMING=1 centrifuger_amd/bin/centrifuger -x $idx "$@" > /tmp/cli_fq.tsv 2> /tmp/cli_fq.err
  local t1=$(date +%s.%N)
  local rows=$(( $(wc -l < /tmp/cli_fq.tsv) - 1 ))
  echo "== $label: $(python -c "el=$t1-$t0; print('process wall %.2f s, %d rows, %.1f M reads/s' % (el, $rows, $rows/el/1e6))"), md5 $(md5sum < /tmp/cli_fq.tsv | cut -c1-12)" | tee -a $out
  grep timing /tmp/cli_fq.err | tr '\n' ' ' | tee -a $out; echo | tee -a $out
}
for pt in ${PTS:-0 16 32}; do
  run "plain FASTQ -t 64 --parse-threads $pt" -u $big -t 64 --parse-threads $pt
done
run "plain FASTQ -t 64 again (page cache warm)" -u $big -t 64
if [ -x oracle/_ref/centrifuger ]; then
  oracle/_ref/centrifuger -x $idx -u $fa -t 64 > /tmp/ref_2m.tsv 2>/dev/null
  ( head -1 /tmp/ref_2m.tsv; for i in $(seq $reps); do tail -n +2 /tmp/ref_2m.tsv; done ) | md5sum | cut -c1-12 | sed 's/^/reference rows (2 M reads x reps): md5 /' | tee -a $out
fi
rm -f $big /tmp/sample.fq
