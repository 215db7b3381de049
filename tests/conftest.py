import os
import sys

import pytest

try:  # tests that hand torch tensors to the library need ONE HIP runtime in the process: torch's copy must load first
    import torch  # noqa: F401  (zune-jpeg_amd/host.py: _share_torch_hip_runtime)
except Exception:  # noqa: BLE001 -- torch is optional for the CPU suite
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def synth():
    import importlib
    return importlib.import_module("zune-jpeg_amd.synth")
