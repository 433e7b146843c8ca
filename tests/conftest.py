import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` through gpurun)')


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests must fail loudly (not skip) when there is no GPU or no HIP library."""
    return


@pytest.fixture(scope='session')
def oracle_libs():
    """Build the C oracle (and, when /root/reference is present, oracle/_ref) once."""
    import subprocess
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    return True


@pytest.fixture
def inject_depth(monkeypatch):
    """TEST-ONLY replacement of the on-device depth chain: `inject_depth(pts_list_on_device, dense (B,V,H,W))` makes the
    product `BEVWarp` un-project through the given completed depth maps (keyed by the samples' point tensors), so a
    kernel-parity test does not hinge on float round-off inside scatter / completion.  The product path itself has no
    such hook (the chain has its own parity tests, and the shape-R tests run it un-injected)."""
    import torch
    from deepinteraction_amd.mmdet3d_plugin.models.utils.encoder_utils import BEVWarp
    table = {}
    orig = BEVWarp.dense_depth

    def patched(geom, pts, I_H, I_W):
        d = table.get(pts.data_ptr())
        if d is None:
            return orig(geom, pts, I_H, I_W)
        if getattr(geom, 'dense_depth', None) is None:
            geom.dense_depth = d.to(pts.device, torch.float32).contiguous()
        return geom.dense_depth

    monkeypatch.setattr(BEVWarp, 'dense_depth', staticmethod(patched))

    def register(pts_list, dense):
        for b, p in enumerate(pts_list):
            table[p.data_ptr()] = dense[b]
        register.keep = getattr(register, 'keep', []) + [pts_list]      # addresses stay unique while registered
    return register
