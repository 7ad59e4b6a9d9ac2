#!/bin/bash
# End to end at cfg4's read count: 100 M x 150 bp reads through the drop-in command line (plain FASTA, and the same reads as BGZF / plain
# gz FASTQ), wall clock including process start, index load and device image; the reference on 2 M of the same reads for the md5.
# Run on the GPU box:  tools/cli_timing_100m.sh [reads in millions, default 100]   -> gpurun_out/cli_timing_100m.txt
export CFR_DEBUG_ENV=1
M=${1:-100}
mkdir -p gpurun_out
python bench.py --steps 2 --warmup 1 --no-pmc --no-extra-configs --no-40gbp > gpurun_out/cli100_bench.json 2> gpurun_out/cli100_bench.err
idx=$(ls /tmp/cfr_bench/*/idx.1.cfr 2>/dev/null | head -1); idx=${idx%.1.cfr}
fa=$(ls /tmp/cfr_bench/*/sample_0.fa 2>/dev/null | head -1)
out=gpurun_out/cli_timing_100m.txt
n2=$(grep -c '>' $fa)
reps=$(( M * 1000000 / n2 ))
echo "index $idx; $fa holds $n2 reads, x $reps = $(( reps * n2 )) reads" > $out
big=/tmp/big100m.fa
for i in $(seq $reps); do cat $fa; done > $big; echo "wrote $(du -h $big | cut -f1)" | tee -a $out
run() {  # label, then the command line's arguments
  local label=$1; shift
  local t0=$(date +%s.%N)
  CFR_CLI_TIMING=1 centrifuger_amd/bin/centrifuger -x $idx "$@" > /tmp/cli_100m.tsv 2> /tmp/cli_100m.err
  local t1=$(date +%s.%N)
  local rows=$(( $(wc -l < /tmp/cli_100m.tsv) - 1 ))
  echo "== $label: $(python -c "el=$t1-$t0; print('process wall %.2f s, %d rows, %.1f M reads/s' % (el, $rows, $rows/el/1e6))"), md5 $(md5sum < /tmp/cli_100m.tsv | cut -c1-12)" | tee -a $out
  grep timing /tmp/cli_100m.err | tr '\n' ' ' | tee -a $out; echo | tee -a $out
}
for prof in "" "--gpu-throughput"; do
  for pt in ${PTS:-0 16}; do
    run "plain FASTA -t 64 $prof --parse-threads $pt" -u $big -t 64 $prof --parse-threads $pt
  done
done
# reference rows for the md5: the first file's rows x reps (the ids repeat) behind one header
if [ -x oracle/_ref/centrifuger ]; then
  oracle/_ref/centrifuger -x $idx -u $fa -t 64 > /tmp/ref_2m.tsv 2>/dev/null
  ( head -1 /tmp/ref_2m.tsv; for i in $(seq $reps); do tail -n +2 /tmp/ref_2m.tsv; done ) | md5sum | cut -c1-12 | sed 's/^/reference rows (2 M reads x reps): md5 /' | tee -a $out
fi
# compressed: BGZF (blocks inflated by several threads) and one plain deflate stream, 20 M reads each
n20=$(( 20000000 / n2 ))
for i in $(seq $n20); do cat $fa; done | awk 'NR % 2 == 1 { print "@" substr($0, 2) } NR % 2 == 0 { print; print "+"; q = $0; gsub(/./, "I", q); print q }' > /tmp/big20m.fq
python - <<'PY'
import sys, os
sys.path.insert(0, "tests")
os.environ.setdefault("CFR_DEBUG_ENV", "1")
from test_host_cpu import write_bgzf
import mmap
f = open("/tmp/big20m.fq", "rb"); m = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
write_bgzf("/tmp/big20m.bgzf.fq.gz", m)
PY
gzip -1 -c /tmp/big20m.fq > /tmp/big20m.plain.fq.gz
echo "20 M reads as FASTQ: $(du -h /tmp/big20m.fq | cut -f1), BGZF $(du -h /tmp/big20m.bgzf.fq.gz | cut -f1), gzip -1 $(du -h /tmp/big20m.plain.fq.gz | cut -f1)" | tee -a $out
run "20 M reads plain FASTQ -t 64 --gpu-throughput" -u /tmp/big20m.fq -t 64 --gpu-throughput
run "20 M reads BGZF -t 64 --gpu-throughput" -u /tmp/big20m.bgzf.fq.gz -t 64 --gpu-throughput
run "20 M reads BGZF -t 64 (default profile)" -u /tmp/big20m.bgzf.fq.gz -t 64
run "20 M reads gzip -1 (one stream) -t 64 --gpu-throughput" -u /tmp/big20m.plain.fq.gz -t 64 --gpu-throughput
rm -f /tmp/big100m.fa /tmp/big20m.fq /tmp/big20m.*.gz
