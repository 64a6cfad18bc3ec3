#!/usr/bin/env python3
"""count the instructions of selected kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only): a static view of where a
kernel's instruction stream goes (VALU / SALU / LDS / VMEM, quarter-rate integer multiplies, DPP moves)
usage: tools/isa_count.py <file.s> <substring of the mangled name> ..."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
for name in sys.argv[2:]:
    for m in re.finditer(r'^(_ZN3ngp\d+[^:\n]*' + re.escape(name) + r'[^:\n]*):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
        body = m.group(2)
        ins = [l.strip().split()[0] for l in body.splitlines() if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
        c = Counter()
        for i in ins:
            c['valu' if i.startswith('v_') else 'salu' if i.startswith('s_') else 'lds' if i.startswith('ds_') else
              'vmem' if i.startswith(('global_', 'buffer_', 'flat_')) else 'other'] += 1
        print(m.group(1)[:70], len(ins), dict(c), 'v_mul_lo', sum(i.startswith('v_mul_lo') for i in ins), 'dpp', body.count('row_sh') + body.count('row_bcast'))
