import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
# every uniform batch send of the test session also runs the one-thread K1 sort and fails unless the threaded one agrees
# (opus_amd/csrc/host/engine.cpp, merge_uniform)
os.environ.setdefault("OPUSGPU_CHECK_SORT", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
