#!/bin/bash
# NOTE: tools/dbg/libcfr_hip_old.so is not kept in the tree: build it from the commit to compare with (git archive <commit> centrifuger_amd/csrc | tar -x -C /tmp/old; hipcc -c cfr_device.hip there; link with today's other objects as tools/dbg/build_variant.sh does)
# round 5: the search kernel of this commit against the one of the round's start (tools/dbg/libcfr_hip_old.so: the device translation unit of
# d726953 linked with today's other objects), alternating on one box: cfg2, pairs, long reads, 20 strains, and the 36-bit kernel on an 8 Gbp
# index (150 bp and long reads).  First the iteration mix of today's kernel.
export CFR_DEBUG_ENV=1
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g /s  search %.2f ms  post %.2f ms  total %.2f ms  step %.2f ms' % (d['value'], d['stage_ms']['search_ms'], d['stage_ms']['tail_ms'], d['stage_ms']['total_ms'], d['ms_per_step']))"; }
legs() {
  echo -n "cfg2 se: "; run; echo -n "cfg3 pe: "; run --mode pe; echo -n "long: "; run --mode long
  echo -n "strains20: "; run --workload strains20
  echo -n "8 Gbp (36-bit kernel): "; run --index-gbp 8
  echo -n "8 Gbp long reads: "; run --index-gbp 8 --mode long
}
echo "== iteration mix, this commit"
CFR_SEARCH_PROF=1 python bench.py --no-cpu-baseline --no-pmc --no-extra-configs --steps 1 --warmup 0 2>&1 >/dev/null | grep "search prof" | head -1
echo "== this commit"; legs
cp centrifuger_amd/libcfr_hip.so /tmp/shipped.so; cp tools/dbg/libcfr_hip_old.so centrifuger_amd/libcfr_hip.so
echo "== round start (d726953)"; legs
cp /tmp/shipped.so centrifuger_amd/libcfr_hip.so
echo "== this commit once more"; legs
