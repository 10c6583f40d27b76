import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def data_dir():
    return os.path.join(ROOT, "tests", "data")


@pytest.fixture(scope="session")
def engine():
    """The HIP engine on cuda:0.  No fallback: a missing library or device is an error."""
    os.environ["TELR_DEBUG"] = "1"      # keep stage-level captures for the parity tests
    # torch first: its wheel carries its own HIP runtime, and a process can initialise only one -- loaded first, libtelrhip.so
    # binds to the same one, and tensors (SeqSet.packed, torch.distributed) and the engine share the device
    import torch  # noqa: F401
    from telr_amd.aligner import Engine
    return Engine(0)
