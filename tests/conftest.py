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


@pytest.fixture
def knobs():
    """Tuning knobs of libams_hip.so are environment variables the library reads ONCE; `knobs(AMS_XDS_FORCE="4,1,1", AMS_PWX_NO_TAIL=None)`
    sets (or, with None, removes) them and makes the library read them again; the environment and the library's copy are restored afterwards."""
    from ams_amd import hip
    saved = {}

    def set_knobs(**kv):
        for k, v in kv.items():
            assert k.startswith("AMS_")
            saved.setdefault(k, os.environ.get(k))
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        hip.check(hip.lib().ams_debug_reload_knobs())

    yield set_knobs
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    hip.check(hip.lib().ams_debug_reload_knobs())


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False
