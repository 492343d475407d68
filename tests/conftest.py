import os
import sys

# the oracle issues tens of thousands of tiny CPU ops: spinning OpenMP workers burn the cores between them
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
os.environ.setdefault("KMP_BLOCKTIME", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # the oracle's many tiny CPU ops thrash with one OpenMP thread per core; cap the pool at 8
    try:
        import torch
        torch.set_num_threads(min(8, torch.get_num_threads()))
    except Exception:
        pass
