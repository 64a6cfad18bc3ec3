"""CPU checks of the drop-in boundary: the shared library loads without a GPU and exports every symbol
include/ngp_hip.h declares; argument validation that happens before any device work returns the documented
error codes/messages; host-side wrapper logic (offset tables, padding rules, parameter layout)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'ngp_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ngp_[a-zA-Z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    import _ngp_capi as capi
    declared = _declared_symbols()
    assert len(declared) >= 28
    for name in declared:
        assert hasattr(capi.lib, name), f'{name} declared in ngp_hip.h but not exported by libngp_hip.so'
    assert capi.lib.ngp_abi_version() == capi.ABI_VERSION
    assert capi.lib.ngp_target_arch() == b'gfx950'
    # the ctypes table binds exactly the declared set
    assert set(capi.EXPORTED) == set(declared)


def test_ctypes_signatures_match_the_header_prototypes():
    """every prototype of include/ngp_hip.h against the ctypes argtypes of _ngp_capi: same parameter count, and per parameter the same
    kind (pointer / uint32 / int / float / size_t) -- a missing argtype lets ctypes pass a 64-bit stream handle as a 32-bit int"""
    import _ngp_capi as capi
    text = open(os.path.join(ROOT, 'include', 'ngp_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    text = re.sub(r'^\s*#.*$', '', text, flags=re.M)
    kinds = {ctypes.c_void_p: 'ptr', ctypes.c_char_p: 'ptr', ctypes.c_uint32: 'u32', ctypes.c_int: 'int', ctypes.c_float: 'f32',
             ctypes.c_size_t: 'w64', ctypes.c_uint64: 'w64'}   # (c_size_t IS c_uint64 on this platform)

    def kind(param):
        param = param.strip()
        if '*' in param or param.startswith('ngp_stream_t'):
            return 'ptr'
        base = param.replace('const ', '').split()[0]
        return {'uint32_t': 'u32', 'int': 'int', 'int32_t': 'int', 'float': 'f32', 'size_t': 'w64', 'uint64_t': 'w64'}[base]

    protos = re.findall(r'\b(?:int|size_t|uint32_t|const char\s*\*)\s*(ngp_[a-zA-Z0-9_]+)\s*\(([^;]*?)\)\s*;', text, flags=re.S)
    assert len(protos) >= 60
    checked = 0
    for name, params in protos:
        fn = getattr(capi.lib, name)
        params = params.strip()
        want = [] if params in ('', 'void') else [kind(p) for p in params.split(',')]
        if fn.argtypes is None:
            assert want == [], f'{name}: no argtypes bound for {len(want)} parameters'
            continue
        got = [kinds[t] for t in fn.argtypes]
        assert got == want, f'{name}: ctypes {got} != header {want}'
        checked += 1
    assert checked >= 55


def test_host_side_argument_validation_needs_no_gpu():
    import _ngp_capi as capi
    lib = capi.lib
    one = ctypes.c_void_p(16)  # never dereferenced: validation fails first
    rc = lib.ngp_grid_encode_forward(one, one, one, one, 8, 3, 3, 2, 1.0, 4, None, 0, 0, 0, capi.NGP_F32, None)
    assert rc == 1 and b'C must be 1, 2, 4, or 8' in lib.ngp_last_error()
    rc = lib.ngp_grid_encode_forward(one, one, one, one, 8, 7, 2, 2, 1.0, 4, None, 0, 0, 0, capi.NGP_F32, None)
    assert rc == 1 and b'input dim' in lib.ngp_last_error()
    rc = lib.ngp_sh_encode_forward(one, one, 8, 3, 9, None, capi.NGP_F32, None)
    assert rc == 1 and b'degree in [1, 8]' in lib.ngp_last_error()
    rc = lib.ngp_ffmlp_forward(one, one, 128, 32, 16, 48, 2, 0, 6, one, one, None)
    assert rc == 1 and b'hidden_dim' in lib.ngp_last_error()
    rc = lib.ngp_ffmlp_forward(one, one, 100, 32, 16, 64, 2, 0, 6, one, one, None)
    assert rc == 1 and b'128' in lib.ngp_last_error()
    rc = lib.ngp_march_rays_train(one, one, one, 1.0, 0.0, 1024, 8, 9, 128, 64, one, one, one, one, one, one, one, one, one, None)
    assert rc == 1 and b'cascade' in lib.ngp_last_error()
    with pytest.raises(RuntimeError, match='cascade'):
        capi.check(rc)
    assert lib.ngp_allocate_splitk(3) == 0 and lib.ngp_free_splitk() == 0
    # the fused compositor reads and writes group tickets at the END of the marcher's workspace: a buffer sized by an older formula is refused
    rc = lib.ngp_composite_train_loss_backward(one, one, one, one, 100, 16, 1e-4, 1, 1.0, None, one, one, one, None, one, one, one, one, one, one, one, one,
                                               256, None)
    assert rc == 1 and b'group tickets' in lib.ngp_last_error()
    # the shard-poisoning verdict of the data-parallel update validates before it launches
    assert lib.ngp_optim_poison_shards(None, 2, 8, one, None) == 1 and lib.ngp_optim_shard_verdict(one, None, None, 1, 8, None) == 1
    words = 2 + 4096 + 2 * 4096 * 20  # fit_end, final ticket, windows per ray, emit masks ...
    assert lib.ngp_march_rays_train_workspace_bytes(4096) == 4 * ((words + 31) // 32 * 32 + 32 * 32)  # ... and 32 group tickets, 128 bytes apart


def test_level_table_is_the_oracle_recipe():
    import _ngp_capi as capi
    import oracle
    for L, pls, H in [(16, 1.3819128800392151, 16), (16, 1.5874010519681994, 16), (4, 2.0, 4), (12, 1.26, 7)]:
        S = float(np.log2(pls))
        sc = (ctypes.c_float * 32)()
        rs = (ctypes.c_uint32 * 32)()
        assert capi.lib.ngp_grid_level_table(L, S, H, ctypes.cast(sc, ctypes.c_void_p), ctypes.cast(rs, ctypes.c_void_p)) == 0
        so, ro = oracle.grid_level_table(L, S, H)
        assert np.array_equal(np.array(sc[:L], np.float32), so) and np.array_equal(np.array(rs[:L], np.uint32), ro)


def test_grid_encoder_module_matches_reference_ctor(golden_dir):
    from gridencoder import GridEncoder
    for rec in json.load(open(os.path.join(golden_dir, 'grid_offsets_ref.json'))):
        enc = GridEncoder(**rec['cfg'])
        assert enc.offsets.tolist() == rec['offsets'] and enc.offsets.dtype == torch.int32
        assert list(enc.embeddings.shape) == rec['embeddings_shape']
        assert float(enc.per_level_scale) == rec['per_level_scale']
        assert enc.embeddings.abs().max() <= 1e-4
        assert enc.output_dim == rec['cfg']['num_levels'] * rec['cfg']['level_dim']


def test_ffmlp_module_layout_and_asserts():
    from ffmlp import FFMLP
    from ffmlp.ffmlp import convert_activation
    net = FFMLP(32, 16, 64, 2)
    assert net.weights.shape == (64 * (32 + 64 + 16),) and net.activation == 0 and net.output_activation == 6
    assert [convert_activation(a) for a in ('relu', 'exponential', 'sine', 'sigmoid', 'squareplus', 'softplus', 'none', 'foo')] == [0, 1, 2, 3, 4, 5, 6, 6]
    for bad in (dict(input_dim=30, output_dim=3, hidden_dim=64, num_layers=2), dict(input_dim=32, output_dim=17, hidden_dim=64, num_layers=2),
                dict(input_dim=32, output_dim=3, hidden_dim=48, num_layers=2), dict(input_dim=32, output_dim=3, hidden_dim=64, num_layers=1)):
        with pytest.raises(AssertionError):
            FFMLP(**bad)


def test_network_state_dict_names_and_shapes():
    from nerf.network_ff import NeRFNetwork
    m = NeRFNetwork(bound=1, cuda_ray=True)
    sd = m.state_dict()
    assert tuple(sd['encoder.embeddings'].shape) == (6119864, 2) and sd['encoder.offsets'].shape == (17,)
    assert sd['sigma_net.weights'].shape == (7168,) and sd['color_net.weights'].shape == (11264,)
    assert sd['density_grid'].shape == (1, 128 ** 3) and sd['density_bitfield'].shape == (128 ** 3 // 8,)
    assert sd['step_counter'].shape == (16, 2) and sd['aabb_train'].tolist() == [-1, -1, -1, 1, 1, 1]
    m8 = NeRFNetwork(bound=8, cuda_ray=True)
    assert m8.cascade == 4 and tuple(m8.encoder.embeddings.shape) == (6664784, 2)


def test_ops_fail_loudly_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from gridencoder import GridEncoder
    enc = GridEncoder(num_levels=2, base_resolution=4, log2_hashmap_size=8)
    with pytest.raises(RuntimeError, match='CUDA tensor'):
        enc(torch.rand(8, 3))
    import raymarching
    with pytest.raises((RuntimeError, AssertionError)):
        raymarching.near_far_from_aabb(torch.rand(4, 3), torch.rand(4, 3), torch.tensor([-1., -1, -1, 1, 1, 1]), 0.2)


def test_grid_backward_workspace_plan_is_host_only():
    """ngp_grid_backward_workspace_bytes plans the atomic-free scatter on the host: which calls are eligible (fp16 C = 2 tables, D <= 3,
    large batches) and how much scratch they need (64 B per sample and level + one descriptor word per (slice, chunk))."""
    import _ngp_capi as capi
    import oracle
    offs, pls = oracle.grid_offsets(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048)
    S = float(np.log2(pls))
    arr = (ctypes.c_int32 * len(offs))(*[int(v) for v in offs])
    ptr = ctypes.cast(arr, ctypes.c_void_p)
    ws = lambda B, D=3, C=2, dtype=capi.NGP_F16, gridtype=0: int(capi.lib.ngp_grid_backward_workspace_bytes(ptr, B, D, C, 16, S, 16, gridtype, 0, dtype))
    n_levels = 16                                                   # 11 hashed levels (128 slices of 4096 entries) + 5 dense ones (128 round-robin bins)
    B = 1 << 18
    chunks = B // 512
    records = n_levels * chunks * 512 * 8 * 8                       # 8 corners x 8 bytes per sample and level
    descriptors = n_levels * 128 * chunks * 4                       # one word per (slice, chunk)
    assert ws(B) == ((descriptors + 255) // 256) * 256 + records + 64
    assert ws(1 << 12) == 0                                         # small batch: atomic path
    assert ws(B, dtype=capi.NGP_F32) == 0 and ws(B, C=4) == 0 and ws(B, D=4) == 0
    assert ws(B, gridtype=1) > 0                                    # tiled grids: every level dense -> round-robin bins
    assert ws(2 * B) > ws(B)
    assert int(capi.lib.ngp_grid_backward_workspace_bytes(None, B, 3, 2, 16, S, 16, 0, 0, capi.NGP_F16)) == 0
    # host copy of the offsets used by the Python wrappers: cached on the tensor, refreshed when it changes
    t = torch.from_numpy(offs.astype(np.int32))
    a1 = capi.host_offsets(t)
    assert list(a1) == [int(v) for v in offs] and capi.host_offsets(t) is a1
    t[1] += 8
    assert list(capi.host_offsets(t))[1] == int(offs[1]) + 8
