"""the numbers of one bench.py line that matter when two builds are compared: python tools/dbg/bench_brief.py bench.json [...]"""
import json, sys
for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    st = d.get("stage_ms", {})
    print(f"{path}: {d['value']:.4g} {d.get('unit','')} {d['ms_per_step']:.2f} ms/step  search {st.get('search_ms',0):.2f} tail {st.get('tail_ms',0):.2f} total {st.get('total_ms',0):.2f}"
          f"  parity {d.get('parity')} oracle {d.get('parity_oracle', {}).get('ok') if isinstance(d.get('parity_oracle'), dict) else d.get('parity_oracle')}")
    sd = d.get("with_device_sdust")
    if sd: print(f"   sdust {sd['value']:.4g} {sd['ms_per_step']:.2f} ms  parity {sd.get('parity')}")
    o = d.get("other_result_layout")
    if o: print(f"   wide layout {o['value']:.4g}")
    for k, v in d.get("other_configs", {}).items():
        if not isinstance(v, dict) or "value" not in v: print("  ", k, str(v)[:100]); continue
        par = {a: v[a] for a in v if "parity" in a}
        print(f"   {k}: {v['value']:.4g} {v['ms_per_step']:.2f} ms  search {v.get('search_ms',0):.2f} tail {v.get('tail_ms',0):.2f} {par}")
    pf = d.get("post_stage")
    if pf: print("   post_stage", json.dumps(pf)[:400])
