import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle is PyTorch-CPU: on the GPU box's 256 hardware threads the default thread count oversubscribes badly (the
    # f64 autograd checks of the -m gpu suite took 560 of its 607 s); 32 threads is the fastest setting measured there
    try:
        import torch
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
    except Exception:  # noqa: BLE001
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False
