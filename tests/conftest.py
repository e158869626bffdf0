import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


@pytest.fixture
def extend_shape():
    """Force the extend-attention workgroup shape for one test (sgl_amd_debug_extend_attention_shape: "82" = 8 waves, 256
    rows per workgroup, "42" / "41" = the 4-wave forms, "auto"; flags 1 = general single-image kernel, 2 = the ping-pong
    kernel, 0 = the 32x32 two-score-set kernel where it applies); the override is cleared afterwards."""
    from sglang_amd import native

    def force(shape, flags=0):
        native.call("sgl_amd_debug_extend_attention_shape", 0 if shape in (None, "auto") else int(shape), int(flags))

    yield force
    native.call("sgl_amd_debug_extend_attention_shape", 0, 0)
