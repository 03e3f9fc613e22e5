import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _library_options_are_per_test():
    """Kernel-form options (dvg_set_option) never leak from one test into the next: every test starts from the defaults,
    plus whatever DVG_TEST_OPTIONS ("name=value,...": how a test re-runs others under an option in a child process) asks."""
    if not _has_gpu():
        yield
        return
    from image_generation_amd import _lib

    _lib.check(_lib.lib().dvg_reset_options(), "dvg_reset_options")
    for item in filter(None, os.environ.get("DVG_TEST_OPTIONS", "").split(",")):
        name, value = item.split("=")
        _lib.set_option(name.strip(), int(value))
    yield
    _lib.check(_lib.lib().dvg_reset_options(), "dvg_reset_options")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
