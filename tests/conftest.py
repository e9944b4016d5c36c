import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the full sweep (further tone streams per size); deselected unless "
                                       "RPF_RUN_SLOW=1 -- tools/gpu_r06.sh final runs it, the driver's -m gpu does not")


def pytest_collection_modifyitems(config, items):
    """`pytest -m gpu` must stay well inside the driver's step limit: the tests marked `slow` --
    always `gpu` tests too -- leave the run unless RPF_RUN_SLOW=1."""
    if os.environ.get("RPF_RUN_SLOW") == "1":
        return
    keep, drop = [], []
    for item in items:
        (drop if item.get_closest_marker("slow") else keep).append(item)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def _have_sources():
    return os.path.isdir(os.path.join(ROOT, "oracle"))


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the checker libraries once if they are missing (the engine library
    itself is built by __graft_entry__.build(); tests never rebuild it on a GPU
    box, they use the .so that travelled with the repo)."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "librpf_oracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    if not os.path.exists(os.path.join(ROOT, "tests", "emul", "librpf_emul.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "emul")], check=True)
    if not os.path.exists(os.path.join(ROOT, "rtl-power-fftw_amd", "librpf_engine.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "rtl-power-fftw_amd", "csrc")], check=True)
    yield
