import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: tens of seconds of CPU oracle work beside the GPU run")


# Collection order of the GPU suite (the driver runs `pytest -m gpu -x`): the oracle parity tests first, whole runs next, everything that
# starts several processes on the one GPU last — a failure in a multi-process functional test must never stand in front of the parity suite.
# Inside a file, tests marked `slow` go last.  Files not named here keep their alphabetical place between the parity block and the
# multi-process block.
_GPU_ORDER_FIRST = ["test_golden.py", "test_gpu_parity.py", "test_gpu_fullsize_oracle.py", "test_gpu_conv1_exact.py", "test_gpu_resnet.py", "test_gpu_resnet_hidden.py",
                    "test_gpu_e2e.py", "test_gpu_fullsize.py", "test_gpu_overlap.py"]
_GPU_ORDER_LAST = ["test_gpu_split.py", "test_gpu_native_comm.py", "test_gpu_bench_launcher.py"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        if name in _GPU_ORDER_FIRST:
            block = (0, _GPU_ORDER_FIRST.index(name))
        elif name in _GPU_ORDER_LAST:
            block = (2, _GPU_ORDER_LAST.index(name))
        else:
            block = (1, 0)
        return block

    # stable: keeps the definition order inside (file, slow-or-not)
    keyed = [((rank(it), os.path.basename(str(it.fspath)), 1 if it.get_closest_marker("slow") else 0), i, it) for i, it in enumerate(items)]
    keyed.sort(key=lambda k: (k[0], k[1]))
    items[:] = [k[2] for k in keyed]


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.build()
    return o


@pytest.fixture(autouse=True)
def _oracle_conv1_chain_unless_a_test_asks(request):
    """oracle.set_conv1_exact is process-wide (like set_threads) and tests/oracle_engine.py switches it to follow an engine's configuration: every test starts
    from the fmaf chain again, so that a test which compares against the chain oracle never inherits another test's setting."""
    o = sys.modules.get("oracle")
    if o is not None and getattr(o, "_lib", None) is not None:
        o.set_conv1_exact(False)
    yield
