import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def nccl_group():
    """ONE 1-rank RCCL process group for the whole session (tests of the
    collective step on a single GPU): creating and destroying groups test by
    test re-initialises RCCL inside one process, which has aborted the
    interpreter on occasion."""
    import socket
    import torch
    import torch.distributed as dist
    created = False
    if not dist.is_initialized():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}",
                                rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        created = True
    yield dist
    if created:
        dist.destroy_process_group()
