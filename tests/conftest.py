"""pytest configuration: registers the `gpu` marker, puts the repo root (for `oracle`, `bench`) and the
product directory `torch-ngp_amd/` (for `gridencoder`, `shencoder`, `raymarching`, `ffmlp`, `encoding`,
`activation`, `nerf`, exactly the import names the reference uses) on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'torch-ngp_amd')
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')


def pytest_collection_modifyitems(config, items):
    # GPU tests must never silently pass on a box without a GPU
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
