#!/usr/bin/env python3
"""Scan built objects (torch-ngp_amd/csrc/_obj*/<unit>.o) for an instruction pattern the compiler does not know is unsafe on gfx950:

  v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64 whose 32-bit shift amount sits in the LAST register of the wave's VGPR allocation
  (v39 of 40, v47 of 48, v63 of 64, ...): ~2.5 % of its executions return a wrong result (tools/probes/vgpr_last_probe.hip,
  profiles/r06_vgpr_last_probe.txt, EXPERIMENTS.md round 6).

Used by __graft_entry__.build() (a build with the pattern fails) and tests/test_isa_invariants.py.  When it trips: give that kernel one more
allocation granule (an `asm volatile("" ::: "v<N>")` clobber of a register 8 above its count) or reorder the source until the operand moves.
  python tools/check_isa_hazards.py [objects...]      (default: every object of the in-tree build that holds device code)"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
TOOLS = [os.path.join(LLVM, t) for t in ('llvm-objcopy', 'clang-offload-bundler', 'llvm-readelf', 'llvm-objdump')]
SHIFT64 = re.compile(r'^(v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64)\s+v\[\d+:\d+\],\s*v(\d+),')


def tools_present():
    return all(os.path.exists(t) for t in TOOLS)


def code_object(obj, directory):
    """the gfx950 code object of a hipcc object file -> path (None when the object holds no device code)"""
    fat, co = os.path.join(directory, 'fat.bin'), os.path.join(directory, 'dev.co')
    subprocess.check_call([TOOLS[0], '-O', 'binary', '--only-section=.hip_fatbin', obj, fat])
    if not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return None
    rc = subprocess.call([TOOLS[1], '--unbundle', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--input=' + fat, '--output=' + co],
                         stderr=subprocess.DEVNULL)
    return co if rc == 0 and os.path.exists(co) else None


def kernel_metadata(co):
    """{kernel symbol: {private_segment_fixed_size, vgpr_count, agpr_count}} from the code object's metadata notes"""
    text = subprocess.check_output([TOOLS[2], '--notes', co], text=True)
    out, cur = {}, {}
    for line in text.splitlines():   # one YAML map per kernel, keys in alphabetical order: .agpr_count opens it, .wavefront_size closes it
        m = re.match(r'\s+(?:- )?\.(\w+):\s+(\S+)', line)
        if not m:
            continue
        key, value = m.groups()
        if key == 'agpr_count':
            cur = {}
        cur[key] = value
        if key == 'wavefront_size' and 'symbol' in cur:
            out[cur['symbol'][:-len('.kd')]] = {k: int(cur[k]) for k in ('private_segment_fixed_size', 'vgpr_count', 'agpr_count')}
    return out


def disassembly(co):
    """{kernel symbol: [instruction lines]}"""
    text = subprocess.check_output([TOOLS[3], '-d', '--no-show-raw-insn', co], text=True)
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.match(r'^[0-9a-f]+ <(\S+)>:$', line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
        elif cur is not None and line.startswith('\t'):
            cur.append(line.split('//')[0].strip())
    return kernels


def last_register_shifts(meta, kernels):
    """[(kernel, vgpr_count, instruction)]: 64-bit shifts that take their amount from the allocation's last register"""
    hits = []
    for sym, ins in kernels.items():
        m = meta.get(sym)
        if m is None or m['agpr_count']:   # (accumulation registers sit above the vector registers: no vector register is the allocation's last)
            continue
        last = (m['vgpr_count'] + 7) // 8 * 8 - 1
        for i in ins:
            h = SHIFT64.match(i)
            if h and int(h.group(2)) >= last:
                hits.append((sym, m['vgpr_count'], i))
    return hits


def scan_object(obj):
    """-> (kernels checked, hits) of one object file"""
    d = tempfile.mkdtemp(prefix='isa_hazard_')
    try:
        co = code_object(obj, d)
        if co is None:
            return 0, []
        meta, kernels = kernel_metadata(co), disassembly(co)
        return sum(1 for k in kernels if k in meta), last_register_shifts(meta, kernels)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def scan_build(obj_dir):
    """every object of a build directory -> (kernels checked, hits)"""
    checked, hits = 0, []
    for obj in sorted(glob.glob(os.path.join(obj_dir, '*.o'))):
        n, h = scan_object(obj)
        checked += n
        hits += h
    return checked, hits


if __name__ == '__main__':
    if not tools_present():
        raise SystemExit('the LLVM tools under /opt/rocm/lib/llvm/bin are missing')
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, 'torch-ngp_amd', 'csrc', '_obj', '*.o')))
    total, bad = 0, []
    for o in objs:
        n, h = scan_object(o)
        total += n
        bad += h
        print(f'{os.path.relpath(o, ROOT)}: {n} kernels, {len(h)} 64-bit shifts with the amount in the last allocated register')
    for sym, count, ins in bad:
        print(f'  {sym} ({count} VGPRs): {ins}')
    sys.exit(1 if bad else 0)
