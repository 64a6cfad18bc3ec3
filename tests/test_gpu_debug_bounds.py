"""GPU: the core cases of the grid, pipeline and graph suites once more against libngp_hip_dbg.so -- the same sources built with
`make DEBUG_BOUNDS=1` (csrc/common.h: NGP_BOUNDS), i.e. with device-side range traps on the indices the fast paths trust: the encoder's
per-XCD work lists and table indices, the record sort's staging slots / descriptors / bin counters, the accumulate's record and entry
indices, the marcher's sample rows.  A violated bound aborts the kernel and the child process; the product build compiles the checks away.
(Round 3 saw one pool-wide `Memory access fault` that was never reproduced: this build is the standing guard against an out-of-range
index that only some memory layout exposes.)  NGP_HIP_LIBRARY selects the library; the operator packages then bind through ctypes (the
compiled `_gridencoder ...` modules link the in-tree product library)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.path.join(ROOT, 'torch-ngp_amd', 'libngp_hip_dbg.so')

CASES = [
    'tests/test_gpu_grid.py::test_forward_lego_config',
    'tests/test_gpu_grid.py::test_backward_binned_matches_oracle_and_is_reproducible',
    'tests/test_gpu_grid.py::test_backward_binned_accumulates_poisons_and_survives_bin_overflow',
    'tests/test_gpu_grid.py::test_backward_binned_all_index_modes',
    'tests/test_gpu_grid.py::test_forward_balanced_work_lists_are_scheduling_only',
    'tests/test_gpu_pipeline.py',
    'tests/test_gpu_graph.py',
]


def test_core_cases_pass_with_range_traps():
    assert os.path.isfile(DBG), 'libngp_hip_dbg.so is missing: python __graft_entry__.py (or make -C torch-ngp_amd/csrc DEBUG_BOUNDS=1)'
    env = dict(os.environ, NGP_HIP_LIBRARY=DBG)
    res = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider'] + CASES, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=1500)
    tail = (res.stdout + res.stderr)[-4000:]
    assert res.returncode == 0, tail
    assert ' passed' in res.stdout and 'failed' not in res.stdout.splitlines()[-1], tail


def test_the_debug_library_really_traps():
    """self-test of the traps: `ngp_debug_forward_bad_tile` (exported by the debug library only) launches the encoder with a work list that
    names a tile behind the last point; the range trap must abort the child process instead of letting the kernel run"""
    code = r'''
import os, sys, ctypes
sys.path[:0] = [os.path.join(sys.argv[1], "torch-ngp_amd"), sys.argv[1]]
import torch
import _ngp_capi as capi
B = 4096
x = torch.rand(B, 3, device="cuda")
emb = torch.zeros(4913 + 12167, 2, device="cuda", dtype=torch.half)
offs = torch.tensor([0, 4913, 4913 + 12167], dtype=torch.int32, device="cuda")
out = torch.empty(2, B, 2, device="cuda", dtype=torch.half)
capi.lib.ngp_debug_forward_bad_tile.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_uint32, ctypes.c_void_p]
rc = capi.lib.ngp_debug_forward_bad_tile(x.data_ptr(), emb.data_ptr(), offs.data_ptr(), out.data_ptr(), B, capi.stream())
torch.cuda.synchronize()
print("survived", rc)
'''
    if not os.path.isfile(DBG):
        pytest.skip('debug library not built')
    env = dict(os.environ, NGP_HIP_LIBRARY=DBG)
    res = subprocess.run([sys.executable, '-c', code, ROOT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and 'survived' not in res.stdout, (res.stdout + res.stderr)[-2000:]
