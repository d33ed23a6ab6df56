"""Developer check: where a kernel's scratch (spill) accesses sit relative to its MFMA loops.
usage: hipcc ... -S --cuda-device-only file.hip -o file.s ; python tools/spill_map.py file.s"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
starts.append((len(lines), 'END'))
for (i, name), (j, _) in zip(starts, starts[1:]):
    body = lines[i:j]
    sc = [k for k, l in enumerate(body) if re.search(r'\bscratch_(load|store)', l)]
    labels = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r'^(\.LBB\w+):', l)] if m}
    loops = []
    for k, l in enumerate(body):
        m = re.search(r's_c?branch\w*\s+(\.LBB\w+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            loops.append((labels[m.group(1)], k))
    mf = [k for k, l in enumerate(body) if 'v_mfma' in l]
    hot = [(a, b) for a, b in loops if sum(1 for q in mf if a <= q <= b) > 8]
    inloop = sum(1 for s in sc if any(a <= s <= b for a, b in hot))
    print(f"{name[:100]:100s} insts={len(body):6d} mfma={len(mf):5d} scratch={len(sc):4d} in-mfma-loop={inloop:4d} loops={[(b - a, sum(1 for q in mf if a <= q <= b)) for a, b in hot][:3]}")
