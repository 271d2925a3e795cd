#!/usr/bin/env python3
"""Static instruction histogram per kernel of a gfx950 .s file (whole function, all exits)."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "."
for m in re.finditer(r'^(\w+):\s+; @\1\n(.*?)^\.Lfunc_end\d+:', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not re.search(pat, name):
        continue
    ins = [l.strip().split()[0] for l in body.split('\n') if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':')]
    c = Counter(ins)
    grp = lambda p: sum(n for k, n in c.items() if k.startswith(p))
    print(f"{name}\n  total {len(ins)} VALU {grp('v_')} SALU {grp('s_')} | " + " ".join(f"{k}:{n}" for k, n in sorted(c.items()) if k.startswith(('global', 'scratch', 'ds_', 'buffer', 'flat', 's_barrier', 's_waitcnt', 's_nop', 's_setprio'))))
    if len(sys.argv) > 3:
        print("  top:", c.most_common(25))
