"""The checker pinned where it checks (What's weak 1 ii).  tests/conftest.py rebuilds oracle/libqpsk_oracle.so with the GPU
box's own gcc, and that build then judges every `-m gpu` parity test; tests/test_oracle_golden.py -- the oracle against the committed
reference-generated fixtures tests/golden/*.npz -- carries no gpu marker, so on the box the checker itself went unchecked.  This module
collects the SAME test functions under the gpu marker: nothing of the reference travels (the fixtures are data, already committed), no
GPU is touched; it only makes sure the oracle that the box built is the oracle the reference pinned."""
import pytest

import test_oracle_golden as _golden

pytestmark = pytest.mark.gpu

for _name in dir(_golden):
    if _name.startswith("test_"):
        globals()[_name] = getattr(_golden, _name)
del _name
