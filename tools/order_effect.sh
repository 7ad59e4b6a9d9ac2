#!/bin/bash
# Does the position of a bench process within one box session change its speed?  (first process fast, later ones slower?)
mkdir -p gpurun_out; out=gpurun_out/order.txt; : > $out
one() { line=$(env $1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1); echo "$2 [$1] $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f M/s ms/step %.2f tail %.2f search %.2f' % (d['value']/1e6, d['ms_per_step'], d['stage_ms']['tail_ms'], d['stage_ms']['search_ms']))")" >> $out; }
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --reads 1000 > /dev/null 2>&1    # index cache only (tiny batch)
one "X=1" "run1"; one "X=1" "run2"; one "HSA_ENABLE_SDMA=0" "run3"; one "HSA_ENABLE_SDMA=0" "run4"; one "X=1" "run5"; one "X=1" "run6"
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" >> $out
cat $out
