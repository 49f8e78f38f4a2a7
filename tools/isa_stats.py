#!/usr/bin/env python3
"""Dev helper: per-kernel instruction histogram from a gfx950 .s (hipcc --offload-device-only -S)."""
import collections
import re
import sys

s = open(sys.argv[1]).read()
keys = sys.argv[2:]
for key in keys:
    m = re.search(r'^(_Z\S*' + re.escape(key) + r'\S*):', s, re.M)
    if not m:
        print(key, "not found"); continue
    i = m.end(); j = s.index('.Lfunc_end', i)
    f = s[i:j]
    ops = collections.Counter(mm.group(1) for mm in re.finditer(r'^\s+([a-z_0-9]+)\s', f, re.M))
    valu = sum(v for k, v in ops.items() if k.startswith('v_'))
    print(key, 'VALU', valu, 'SALU', sum(v for k, v in ops.items() if k.startswith('s_')),
          'scratch', sum(v for k, v in ops.items() if 'scratch' in k),
          'vgpr', re.search(re.escape(m.group(1)) + r'\.num_vgpr, (\d+)', s).group(1))
    print('  ', ops.most_common(28))
