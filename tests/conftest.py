"""Test configuration.

* ``gpu`` marker: tests that need a real MI355X (run with ``-m gpu`` on the GPU box).
* Everything else runs on CPU.  The CPU oracle (``oracle/``) is test infrastructure and is
  imported here, never from ``bourse_amd/``.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (HIP path through the C ABI)")


@pytest.fixture(scope="session")
def oracle():
    import pyoracle

    pyoracle.lib()
    return pyoracle
