import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from znippy_amd import hip
    ctx = hip.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def gpu_ctx_roles():
    """A context whose tables ALWAYS go through the role-split persistent kernel first (ZNIPPY_ROLES_MIN=1; by default
    only tables of >= 2048 small tiles do), so small test tables exercise it.  Switches are read at context creation."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from znippy_amd import hip
    old = os.environ.get("ZNIPPY_ROLES_MIN")
    os.environ["ZNIPPY_ROLES_MIN"] = "1"
    try:
        ctx = hip.Context(0)
    finally:
        if old is None:
            del os.environ["ZNIPPY_ROLES_MIN"]
        else:
            os.environ["ZNIPPY_ROLES_MIN"] = old
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def gpu_ctx_fz_only():
    """A context with NO serial fallback behind the two-phase foreign-frame path (ZNIPPY_FZ_ONLY=1): a frame that path
    does not finish shows up as a corrupt row, so a clean result proves the two-phase kernels decoded it."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from znippy_amd import hip
    old = os.environ.get("ZNIPPY_FZ_ONLY")
    os.environ["ZNIPPY_FZ_ONLY"] = "1"
    try:
        ctx = hip.Context(0)
    finally:
        if old is None:
            del os.environ["ZNIPPY_FZ_ONLY"]
        else:
            os.environ["ZNIPPY_FZ_ONLY"] = old
    yield ctx
    ctx.close()
