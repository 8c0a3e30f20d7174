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


def pytest_sessionfinish(session, exitstatus):
    """What the gate screen kept / drew and where imposed convolution gates differed from the
    oracle's own, for the record (profiles/rNN/gate_screen.txt): a screen that
    quietly rejects more and more would be choosing the batches."""
    try:
        from tests import gate_screen as gs
    except Exception:
        return
    if not (gs.ALL_STATS or gs.LAST_DISAGREEMENTS):
        return
    out = os.environ.get("SCAE_GATE_SCREEN_LOG",
                         os.path.join(ROOT, "gpurun_out", "gate_screen.txt"))
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        with open(out, "w") as f:
            f.write(f"SAFETY {gs.SAFETY}  FLOOR {gs.FLOOR}  MIN_KEPT_RATIO {gs.MIN_KEPT_RATIO}  "
                    f"DISAGREE_BAR {gs.DISAGREE_BAR}\n")
            f.write("# gate screen: what, kept, drawn, ratio, skipped leading ReLU calls\n")
            for what, kept, drawn, skip in gs.ALL_STATS:
                f.write(f"screen  {what!r}  {kept}  {drawn}  {kept / max(1, drawn):.3f}  skip={skip}\n")
            f.write("# imposed convolution gates: what, layer, units decided differently, units, "
                    "worst |pre| / layer max\n")
            for what, k, n_diff, n, worst in gs.LAST_DISAGREEMENTS:
                f.write(f"impose  {what!r}  layer {k}  {n_diff}  {n}  {worst:.2e}\n")
    except OSError:
        pass
