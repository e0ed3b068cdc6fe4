"""-m gpu: a fixed set of seeds of the differential tester (tests/fuzz_plans.py): random plans x tables x batchings
through the C ABI against the oracle.  tools/fuzz_device.py runs any other range of seeds."""
import pytest

from fuzz_plans import run_seed

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(48))
def test_seed(seed):
    run_seed(seed)
