"""Where do the kernels touch their scratch (spill) memory - inside the loops they live in, or outside?
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only -S -o dev.s centrifuger_amd/csrc/cfr_device.hip
       python tools/dbg/scratch_sites.py dev.s 'k_search_chains_v2<2, false, false, false>' 'k_adjust_tail<2>' ..."""
import re, subprocess, sys
lines = open(sys.argv[1]).read().split('\n')
want = sys.argv[2:]
starts = [(i, re.match(r'^(_Z\w+):', l).group(1)) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
names = subprocess.run(['c++filt'], input='\n'.join(n for _, n in starts).encode(), stdout=subprocess.PIPE).stdout.decode().split('\n')
for idx, (i, name) in enumerate(starts):
    d = names[idx]
    if not any(k in d for k in want):
        continue
    end = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
    body = lines[i:end]
    for j, l in enumerate(body):
        if 's_endpgm' in l:
            body = body[:j + 1]
            break
    labels = {}
    for j, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = j
    back = []
    for j, l in enumerate(body):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < j:
                back.append((labels[t], j))
    isinst = lambda l: l.startswith('\t') and not l.strip().startswith(('.', ';'))
    ninst = sum(1 for l in body if isinst(l))
    sc = [(j, l.strip()) for j, l in enumerate(body) if re.search(r'\bscratch_(load|store)', l)]
    back.sort(key=lambda x: x[1] - x[0], reverse=True)
    print(d.split('(')[0], '| instructions', ninst, '| scratch ops', len(sc), '| loops', len(back))
    if back:
        lo, hi = back[0]
        print('  largest loop: lines', lo, '-', hi, '=', sum(1 for l in body[lo:hi] if isinst(l)), 'instructions')
    for j, l in sc:
        inl = [(a, b) for a, b in back if a <= j <= b]
        inner = min(inl, key=lambda x: x[1] - x[0]) if inl else None
        where = 'outside every loop' if not inl else 'in %d loop(s), innermost of %d instructions%s' % (len(inl), sum(1 for x in body[inner[0]:inner[1]] if isinst(x)), ' (= the largest)' if back and inner == back[0] else '')
        print('   line %6d  %-46s %s' % (j, l.split(';')[0].strip()[:46], where))
