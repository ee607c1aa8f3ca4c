import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; torch.cuda.is_available() is False")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True, scope="module")
def _release_gpu_memory_between_modules():
    """Some GPU tests launch further processes on the same device (bench.py with 2 / 4 / 8 ranks on one GPU): what this process keeps from
    earlier modules - torch's cached blocks (a 41 GB stage-4 cache ...), the library's workspace arenas (25 GB per context that encoded a
    whole window) - must not depend on the order the modules ran in."""
    yield
    if "torch" not in sys.modules:
        return
    import gc
    import torch
    if not torch.cuda.is_available():
        return
    gc.collect()
    try:
        from orca_amd import engine, sv_drivers
        sv_drivers.clear_encoding_cache()          # the drivers' per-thread stores (segments, chromosome encodings, stage-4 caches)
        for ctx in getattr(engine._tls, "ctxs", {}).values():
            ctx.release_workspace()
        for pool in engine._thread_pools().values():
            pool.release_workspaces()
    except Exception:
        pass
    torch.cuda.empty_cache()
