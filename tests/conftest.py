import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
  # a fresh checkout has no librip_hip.so (it is built in-tree and git-ignored): build it once before collection so
  # that the suite does not depend on `__graft_entry__.build()` having been run by hand (hipcc cross-compiles without a
  # GPU; with the objects cached this is a hash check).  The PRODUCT still fails loudly without the library.
  lib = os.path.join(ROOT, "oatomobile_amd", "librip_hip.so")
  if not os.path.exists(lib):
    try:
      import __graft_entry__
      __graft_entry__.build()
    except Exception as e:  # no hipcc here / a compiler that rejects a flag: the oracle-only tests must still run
      config._rip_build_error = "%s: %s" % (type(e).__name__, e)
      sys.stderr.write("conftest: building librip_hip.so failed (%s); tests that load the library will FAIL with this "
                       "message, the oracle / golden tests still run\n" % config._rip_build_error)


def pytest_runtest_setup(item):
  """A test that needs librip_hip.so fails AT ITS START with the recorded build error (a compile error on the GPU box
  used to surface as dozens of load errors far from the cause).  Tests that only touch the oracle / golden files run."""
  err = getattr(item.config, "_rip_build_error", None)
  if err is None or os.path.exists(os.path.join(ROOT, "oatomobile_amd", "librip_hip.so")):
    return
  if item.get_closest_marker("gpu") is not None or item.fspath.basename in ("test_host_cpu.py", "test_distributed_cpu.py"):
    pytest.fail("librip_hip.so is missing because __graft_entry__.build() failed: %s" % err, pytrace=False)


@pytest.fixture(scope="session")
def golden():
  import numpy as np

  def load(name):
    return np.load(os.path.join(GOLDEN, name))

  return load
