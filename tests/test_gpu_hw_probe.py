"""The gfx950 finding behind tests/test_isa_invariants.py's 64-bit-shift rule, re-measured on whatever box runs the GPU suite: the stand-alone
probe tools/probes/vgpr_last_probe.hip is compiled with hipcc and run (~10 s).  Hard assertions only on the CONTROL rows (a larger allocation,
another register, the unaffected instructions: must be exact); the rows that show the misread are reported, not required -- a part or a
firmware without it would be good news, and the ISA rule would merely be stricter than needed."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = '/opt/rocm/bin/hipcc'


@pytest.mark.gpu
def test_last_register_shift_probe(tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip('no hipcc on this box')
    exe = str(tmp_path / 'vgpr_last')
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-Wno-unused-value', os.path.join(ROOT, 'tools', 'probes', 'vgpr_last_probe.hip'), '-o', exe],
                          stderr=subprocess.DEVNULL, timeout=300)
    out = subprocess.check_output([exe], text=True, timeout=120)
    rows = []
    for line in out.splitlines():
        m = re.match(r'(\S+)\s+32-bit operand in v(\d+), kernel allocates\s+(\d+) VGPRs: wrong results\s+(\d+) of (\d+) \(operand register read back wrong: (\d+)\)', line)
        if m:
            rows.append((m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(6))))
    assert len(rows) >= 18, out
    shifts = ('v_lshlrev_b64', 'v_lshrrev_b64', 'v_ashrrev_i64')
    exposed = [r for r in rows if r[0] in shifts and r[1] == r[2] - 1]       # the amount sits in the allocation's last register
    control = [r for r in rows if r not in exposed]
    assert len(exposed) >= 5 and len(control) >= 12
    for op, reg, alloc, wrong, wrong_reg in rows:
        assert wrong_reg == 0, f'{op}: the operand register itself read back wrong ({wrong_reg})'
    for op, reg, alloc, wrong, _ in control:
        assert wrong == 0, f'{op} with its operand in v{reg} of a {alloc}-register kernel: {wrong} wrong results -- the finding is wider than the ISA rule assumes'
    tail = [l for l in out.splitlines() if l.startswith('in range')]
    assert tail and ': 0 wrong; ' in tail[0] and tail[0].count(' 0 wrong') == 2, tail
    seen = sum(1 for r in exposed if r[3] > 0)
    print(f'\nlast-register 64-bit shifts misread on this box: {seen} of {len(exposed)} exposed rows ' + ', '.join(f'{r[0]} v{r[1]}/{r[2]}: {r[3]}' for r in exposed))
    dst = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(dst):
        with open(os.path.join(dst, 'vgpr_last_probe.txt'), 'w') as f:
            f.write(out)
