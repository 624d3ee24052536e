import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def free_port():
    """A TCP port nobody listens on right now (the multi-process tests rendezvous on 127.0.0.1): fixed port numbers
    collide with a previous test's sockets still in TIME_WAIT, or with another job on a shared box."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]
