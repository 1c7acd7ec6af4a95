import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def hipb():
    """The HIP backend, exactly as shipped (no skip: on a GPU box a missing GPU/library must fail loudly).  In particular the
    planes gate (nn._PCONV_MIN_GFLOP: which conv passes take the pre-split-operand kernels of vf_pgemm.hip) is the shipped one;
    tests that must exercise the planes path on the suite's small nets ask for the `planes_gate` fixture."""
    import video_filler_amd  # noqa: F401
    from video_filler_amd.backend import get_backend
    return get_backend()


PLANES_GATES = ["planes", "shipped"]


@pytest.fixture(params=PLANES_GATES)
def planes_gate(request, hipb):
    """Both routings of the 4x4 stride-2 conv passes, per test:
      planes   gate dropped to 0: every pass with >= 1024 GEMM rows takes the planes kernels (k_pconv_dma / k_pwgrad_group), their
               BatchNorm-written operand planes and the weight-plane refresh — what configs[1] runs at batchSize 64, here on the
               suite's small nets;
      shipped  the shipped threshold (3 GFLOP per pass): on small nets / batches every pass stays on k_igemm with the BatchNorm
               statistics from its epilogue — what configs[2] and configs[4] run at their batch sizes."""
    from video_filler_amd import nn
    old = nn._PCONV_MIN_GFLOP
    nn._PCONV_MIN_GFLOP = 0.0 if request.param == "planes" else nn.PCONV_MIN_GFLOP_SHIPPED
    try:
        yield request.param
    finally:
        nn._PCONV_MIN_GFLOP = old


HOSTS = ["mirror", "cabi"]


@pytest.fixture(params=HOSTS)
def host(request, hipb):
    """who drives netG / netD in the trainers: the module-by-module Python mirror (nn.py) or the library's own net object
    (vf_net_*, cnet.CNet) — the boundary a Lua host binds (VERDICT r2 missing #1).  Same kernels, same plan, same bars."""
    from video_filler_amd import trainers
    old = trainers.DEFAULT_HOST
    trainers.DEFAULT_HOST = request.param
    try:
        yield request.param
    finally:
        trainers.DEFAULT_HOST = old
