import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
  """Three worker processes, whole test FILES per worker, when pytest-xdist is there (it is part
  of the image) and the command line does not say otherwise: the minute-long items of the GPU
  suite are CPU oracles of full-size configurations, which then run beside the other files'
  kernel tests -- 360 s -> 203 s on the GPU box, same tests. EPOS_TEST_WORKERS=<n> overrides
  (0 = one process); an explicit -n / -p no:xdist is respected, and so are --pdb / --trace
  and -s (capture=no), which xdist cannot serve: those runs stay in one process."""
  opt = config.option
  if hasattr(config, 'workerinput') or not hasattr(opt, 'numprocesses'):
    return None                       # inside a worker, or xdist not loaded
  if opt.numprocesses is not None or getattr(opt, 'collectonly', False):
    return None                       # the command line decided
  if getattr(opt, 'usepdb', False) or getattr(opt, 'trace', False) or \
      getattr(opt, 'capture', None) == 'no':
    return None                       # interactive debugging / live output: one process
  n = int(os.environ.get('EPOS_TEST_WORKERS', '3'))
  if n > 0:
    opt.numprocesses = n
    opt.dist = 'loadfile'
  return None


def pytest_configure(config):
  config.addinivalue_line(
      'markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
  return GOLDEN
