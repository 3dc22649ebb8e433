import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; builds oracle/libpbr_oracle.so on first use)."""
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
    return dict(np.load(path, allow_pickle=False))


@pytest.fixture(scope="session")
def golden2():
    """Full-size IBL fixtures (tests/golden/make_golden_v2.py): LUT 256^2 / 512^2 CRC + samples, prefiltered env
    512^2 x 5 sampled texels, SH9 of the 512^2 bench sky."""
    path = os.path.join(ROOT, "tests", "golden", "golden_v2.npz")
    return dict(np.load(path, allow_pickle=False))


@pytest.fixture(scope="session")
def ibl(orc):
    import common
    return common.small_ibl(orc)


@pytest.fixture(scope="session")
def ctx():
    """HIP context on cuda:0 through the C ABI — fails loudly when the extension is missing."""
    import torch
    assert torch.cuda.is_available(), "GPU test selected but no GPU visible"
    from direct12pbrrenderer_amd.api import PbrContext
    c = PbrContext(0)
    yield c
    c.sync()
    c.close()
