#!/usr/bin/env python3
"""Static instruction mix per kernel of a hipcc -S listing (lab tool: `hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only`)."""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    ops = collections.Counter()
    for line in body.split('\n'):
        line = line.strip()
        if not line or line.startswith((';', '.')) or line.endswith(':'):
            continue
        ops[line.split()[0]] += 1
    pick = lambda f: sum(v for k, v in ops.items() if f(k))
    print(name[:70], 'total', sum(ops.values()), 'f64', pick(lambda k: 'f64' in k), 'bperm', ops['ds_bpermute_b32'], 'cndmask', pick(lambda k: k.startswith('v_cndmask')),
          'rcp', pick(lambda k: 'rcp_f64' in k), 'div_scale', pick(lambda k: 'div_scale' in k), 'dpp', pick(lambda k: 'dpp' in k), 'mfma', pick(lambda k: 'mfma' in k),
          'waitcnt', ops['s_waitcnt'], 'vmem', pick(lambda k: k.startswith(('global_', 'buffer_', 'scratch_'))), 'lds', pick(lambda k: k.startswith('ds_')))
