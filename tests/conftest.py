import os
import sys

import pytest

# exercise the fused E+M Lloyd pass on the small parity shapes too (it is
# normally reserved for batches of >= 256 chunks)
os.environ.setdefault('HSGK_FUSED_MIN_CHUNKS', '1')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def oracle():
  from oracle import oracle as orc
  orc.lib()
  return orc
