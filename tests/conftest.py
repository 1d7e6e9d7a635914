import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files(prefix="s1_"):
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.spdg")))


def golden_ids(prefix="s1_"):
    return [os.path.basename(f)[:-5] for f in golden_files(prefix)]
