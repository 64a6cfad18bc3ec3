"""CPU: the compiled Python bindings over the C ABI (torch-ngp_amd/_gridencoder.so, _shencoder.so, _freqencoder.so, _raymarching.so,
_ffmlp.so -- the module names the reference imports first, e.g. gridencoder/grid.py:9-12) load, export exactly the callables of the
reference's pybind tables (gridencoder/src/bindings.cpp:5-9, shencoder/src/bindings.cpp:5-8, raymarching/src/bindings.cpp:5-19,
ffmlp/src/bindings.cpp:5-11, freqencoder/src/bindings.cpp:5-8) plus the documented extensions, and fail the way TORCH_CHECK does
(RuntimeError, the reference's message) -- no compute calls without a GPU."""
import importlib
import os

import pytest
import torch  # noqa: F401  (the modules link against libtorch)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REFERENCE_TABLES = {
    '_gridencoder': ['grid_encode_forward', 'grid_encode_backward', 'grad_total_variation'],
    '_shencoder': ['sh_encode_forward', 'sh_encode_backward'],
    '_freqencoder': ['freq_encode_forward', 'freq_encode_backward'],
    '_raymarching': ['packbits', 'near_far_from_aabb', 'sph_from_ray', 'morton3D', 'morton3D_invert', 'march_rays_train',
                     'composite_rays_train_forward', 'composite_rays_train_backward', 'march_rays', 'composite_rays'],
    '_ffmlp': ['ffmlp_forward', 'ffmlp_inference', 'ffmlp_backward', 'allocate_splitk', 'free_splitk'],
}
EXTENSIONS = {
    '_gridencoder': ['grid_corner_indices'],
    '_raymarching': ['packbits_capped', 'march_rays_ex', 'compact_rays', 'march_rays_dev', 'composite_rays_dev', 'compact_rays_dev',
                     'density_grid_update', 'density_grid_update_workspace_bytes', 'coarse_occupancy', 'cull_rays'],
}


@pytest.mark.parametrize('name', sorted(REFERENCE_TABLES))
def test_module_loads_in_tree_and_exports_the_reference_table(name):
    mod = importlib.import_module(name)
    assert os.path.dirname(os.path.abspath(mod.__file__)) == os.path.join(ROOT, 'torch-ngp_amd'), 'the binding must be the in-tree build'
    public = sorted(n for n in dir(mod) if not n.startswith('_'))
    assert public == sorted(REFERENCE_TABLES[name] + EXTENSIONS.get(name, []))
    for fn in public:
        assert callable(getattr(mod, fn))


def test_reference_tables_match_the_reference_sources_when_present():
    """in the build container: the tables above are what the reference's bindings.cpp files m.def()"""
    import re
    ref = '/root/reference'
    if not os.path.isdir(ref):
        pytest.skip('reference checkout not present (GPU box)')
    for name, table in REFERENCE_TABLES.items():
        src = open(os.path.join(ref, name[1:], 'src', 'bindings.cpp')).read()
        assert sorted(re.findall(r'm\.def\("(\w+)"', src)) == sorted(table), name


def test_mirror_packages_bind_to_the_compiled_modules():
    import ffmlp.ffmlp
    import freqencoder.freq
    import gridencoder.grid
    import raymarching.raymarching
    import shencoder.sphere_harmonics
    for pkg, name in ((gridencoder.grid, '_gridencoder'), (shencoder.sphere_harmonics, '_shencoder'), (raymarching.raymarching, '_raymarching'),
                      (ffmlp.ffmlp, '_ffmlp'), (freqencoder.freq, '_freqencoder')):
        assert pkg._backend.__name__ == name


def test_errors_are_runtime_errors_with_the_reference_wording():
    import _ffmlp
    import _gridencoder
    import _raymarching
    import _shencoder
    x = torch.zeros(8, 3)
    offs = torch.zeros(3, dtype=torch.int32)
    with pytest.raises(RuntimeError, match='inputs must be a CUDA tensor'):      # CHECK_CUDA, gridencoder.cu:15
        _gridencoder.grid_encode_forward(x, x, offs, x, 8, 3, 2, 1, 0.5, 16, None, 0, False, 0)
    with pytest.raises(RuntimeError, match='must be a CUDA tensor'):
        _shencoder.sh_encode_forward(x, x, 8, 3, 4, None)
    with pytest.raises(RuntimeError, match='must be a CUDA tensor'):
        _raymarching.near_far_from_aabb(x, x, torch.zeros(6), 8, 0.2, torch.zeros(8), torch.zeros(8))
    with pytest.raises(RuntimeError, match='must be a CUDA tensor'):
        _ffmlp.ffmlp_forward(x, x, 8, 32, 16, 64, 2, 0, 6, x, x)
    with pytest.raises(TypeError):     # pybind: wrong arity, as with the reference's modules
        _gridencoder.grid_encode_forward(x, x)
    _ffmlp.allocate_splitk(4)          # callable no-ops (ffmlp.py:126 calls it from every FFMLP constructor)
    _ffmlp.free_splitk()
