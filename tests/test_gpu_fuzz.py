"""A fixed handful of differential-fuzz cases (tests/fuzzcases.py) in the GPU suite: random configurations through the library's own
kernel choice against the oracle, bit for bit.  tools/fuzz.py runs the same generator for as long as one likes."""
import pytest

import fuzzcases

pytestmark = pytest.mark.gpu


def _modem(**kw):
    import qpsk_amd
    return qpsk_amd.Modem(**kw)


@pytest.mark.parametrize("first", [1000, 1012, 1024])
def test_fuzz_batch(oracle, first):
    ran = 0
    for seed in range(first, first + 12):
        desc, bad = fuzzcases.batch_case(oracle, _modem, seed, max_samples=2_500_000)
        if bad is None:
            continue
        ran += 1
        assert not bad, "%s: %s differ" % (desc, bad)
    assert ran >= 8


@pytest.mark.parametrize("first", [2000, 2008])
def test_fuzz_streams(oracle, first):
    ran = 0
    for seed in range(first, first + 8):
        desc, bad = fuzzcases.streams_case(oracle, _modem, seed)
        if bad is None:
            continue
        ran += 1
        assert not bad, "%s: %s" % (desc, bad)
    assert ran >= 5


def test_fuzz_stages(oracle):
    ran = 0
    for seed in range(3000, 3024):
        desc, bad = fuzzcases.stages_case(oracle, _modem, seed)
        if bad is None:
            continue
        ran += 1
        assert not bad, "%s: %s" % (desc, bad)
    assert ran >= 16
