import pathlib
import sys

import pytest

REPO = pathlib.Path(__file__).resolve().parent.parent
PACKAGE = REPO / 'vision-transform-codes_amd'

# The package is laid out like the reference: its root goes on sys.path and the
# plugins are imported as analysis_transforms.* / dict_update_rules.*
for p in (str(PACKAGE), str(REPO / 'oracle'), str(REPO / 'tests')):
  if p not in sys.path:
    sys.path.insert(0, p)


def pytest_configure(config):
  config.addinivalue_line(
      'markers', 'gpu: needs a real MI355X (run with `pytest -m gpu`)')


@pytest.fixture(scope='session')
def device():
  import torch
  if not torch.cuda.is_available():
    pytest.skip('no HIP device')
  return torch.device('cuda:0')
