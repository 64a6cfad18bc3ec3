#!/usr/bin/env python3
"""count the instructions of selected kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only): a static view of where a
kernel's instruction stream goes -- VALU / SALU / LDS / VMEM, MFMA, conversions, moves (incl. accumulator reads), DPP, waits, quarter-rate
integer multiplies.  A wave64 VALU instruction occupies its SIMD for 4 cycles, a 32x32x16 MFMA for 32: the last column is the static
ratio of VALU issue cycles to MFMA cycles (how busy the matrix pipe can be at best when nothing else stalls).
usage: tools/isa_count.py <file.s> <substring of the mangled name> ...     (--md: a markdown table)"""
import re
import sys
from collections import Counter

md = '--md' in sys.argv
args = [a for a in sys.argv[1:] if a != '--md']
s = open(args[0]).read()
rows = []
for name in args[1:]:
    for m in re.finditer(r'^(_ZN3ngp\d+[^:\n]*' + re.escape(name) + r'[^:\n]*):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
        body = m.group(2)
        ins = [l.strip().split()[0] for l in body.splitlines() if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
        c = Counter()
        for i in ins:
            c['valu' if i.startswith('v_') and not i.startswith('v_mfma') else 'mfma' if i.startswith('v_mfma') else 'salu' if i.startswith('s_') else
              'lds' if i.startswith('ds_') else 'vmem' if i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other'] += 1
        cvt = sum(i.startswith('v_cvt') for i in ins)
        mov = sum(i.startswith(('v_mov', 'v_accvgpr')) for i in ins)
        wait = sum(i.startswith('s_waitcnt') for i in ins)
        nop = sum(i.startswith('s_nop') for i in ins)
        pk = sum(i.startswith('v_pk_') for i in ins)
        mul_lo = sum(i.startswith('v_mul_lo') for i in ins)
        dpp = body.count('row_sh') + body.count('row_bcast') + body.count('wave_sh') + body.count('quad_perm')
        ratio = (4.0 * c['valu']) / (32.0 * c['mfma']) if c['mfma'] else float('nan')
        rows.append((m.group(1), len(ins), c, cvt, mov, pk, wait, nop, mul_lo, dpp, ratio))
if md:
    print('| kernel | instructions | VALU | MFMA | SALU | LDS | VMEM | v_cvt | v_mov / accvgpr | v_pk | s_waitcnt | s_nop | DPP | VALU issue cycles : MFMA cycles |')
    print('|---|---|---|---|---|---|---|---|---|---|---|---|---|---|')
    for n, tot, c, cvt, mov, pk, wait, nop, mul_lo, dpp, ratio in rows:
        print(f"| `{n[:90]}` | {tot} | {c['valu']} | {c['mfma']} | {c['salu']} | {c['lds']} | {c['vmem']} | {cvt} | {mov} | {pk} | {wait} | {nop} | {dpp} | {ratio:.2f} |")
else:
    for n, tot, c, cvt, mov, pk, wait, nop, mul_lo, dpp, ratio in rows:
        print(n[:70], tot, dict(c), 'v_cvt', cvt, 'v_mov', mov, 'v_pk', pk, 's_waitcnt', wait, 's_nop', nop, 'v_mul_lo', mul_lo, 'dpp', dpp, f'valu:mfma cycles {ratio:.2f}')
