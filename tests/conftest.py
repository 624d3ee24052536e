import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


ORACLE_THREADS = 8


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle's index kernels (index_add_, index_select on a few thousand rows) are 10x SLOWER with the 128
    # OpenMP threads torch picks on the GPU box's 256-core host than with 8: test_config3_three_step_loss_trajectory
    # 105 s -> 10 s (round 6, OMP_NUM_THREADS = 4 / 8 / 16: 13 / 10 / 11 s).  Tests that want another count set it
    # themselves and restore this one.
    try:
        import torch
        if torch.get_num_threads() > ORACLE_THREADS:
            torch.set_num_threads(ORACLE_THREADS)
    except Exception:       # noqa: BLE001 — torch missing: nothing to bound
        pass


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
