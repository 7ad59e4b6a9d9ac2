#!/bin/bash
# In-process stress of cfr_index_open beside live device images (csrc/cfr_stress.cpp), on the GPU box:
#   tools/stress_open.sh <tag> [opens] [plain rounds] [pytest loops]
# plain build x rounds, then the ASan+UBSan and TSan variants of the host code (make SAN=...), then the protein test file in a loop
# (the place the round-3 failure was seen).  Everything lands in gpurun_out/<tag>/.
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r4a}; OPENS=${2:-2000}; ROUNDS=${3:-5}; LOOPS=${4:-10}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
T=$(mktemp -d /tmp/cfr_stress.XXXXXX)
G=$PWD/tests/golden
gunzip -c $G/f10.1.cfr.gz > $T/f10.1.cfr; cp $G/f10.2.cfr $G/f10.4.cfr $T/
gunzip -c $G/prot/p4.1.cfr.gz > $T/p4.1.cfr; cp $G/prot/p4.2.cfr $G/prot/p4.4.cfr $T/
LIST=$G/f6,$G/f6_b1,$G/f6_b8,$G/f6_off3,$T/f10,$G/prot/p2,$G/prot/p2_b1_off2,$G/prot/p3_b4,$T/p4
B=centrifuger_amd/bin
for r in $(seq 1 $ROUNDS); do
  timeout 900 $B/cfr_stress --load $T/f10,$G/prot/p3_b4 --open $LIST --reads $G/se.fq --opens $OPENS > $OUT/stress_plain_$r.log 2>&1
  echo "exit $?" >> $OUT/stress_plain_$r.log
done
if [ -x $B/cfr_stress_asan ]; then
  ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
    timeout 1200 $B/cfr_stress_asan --load $T/f10,$G/prot/p3_b4 --open $LIST --reads $G/se.fq --opens $((OPENS / 4)) > $OUT/stress_asan.log 2>&1
  echo "exit $?" >> $OUT/stress_asan.log
fi
if [ -x $B/cfr_stress_tsan ]; then
  TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 second_deadlock_stack=1" \
    timeout 1200 $B/cfr_stress_tsan --load $T/f10,$G/prot/p3_b4 --open $LIST --reads $G/se.fq --opens $((OPENS / 10)) > $OUT/stress_tsan.log 2>&1
  echo "exit $?" >> $OUT/stress_tsan.log
  # the reports that involve this library's own frames (the HIP runtime is not instrumented: its internals do not count)
  grep -c "WARNING: ThreadSanitizer" $OUT/stress_tsan.log > $OUT/stress_tsan_summary.txt
  grep -B2 -A25 "WARNING: ThreadSanitizer" $OUT/stress_tsan.log | grep -E "cfr::|cfr_[a-z_]+ " | sort | uniq -c | sort -rn | head -40 >> $OUT/stress_tsan_summary.txt
fi
for k in $(seq 1 $LOOPS); do
  timeout 600 python -m pytest tests/test_gpu_protein.py -q -m gpu -x > $OUT/protein_loop_$k.log 2>&1
  echo "exit $?" >> $OUT/protein_loop_$k.log
done
tail -n 2 $OUT/stress_plain_*.log $OUT/stress_asan.log $OUT/stress_tsan.log 2>/dev/null
for k in $(seq 1 $LOOPS); do tail -n 2 $OUT/protein_loop_$k.log | head -1; done
rm -rf $T
