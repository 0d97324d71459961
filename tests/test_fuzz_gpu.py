"""A fixed slice of the differential campaign of tests/fuzz_dem.py in the GPU suite: 24 random small systems (pair
style, packing, polydispersity, walls or periodic faces, frozen layers in both fix orders, cohesion, friction and
damping switches, skin, speeds), each through several neighbour rebuilds, HIP against the oracle."""
import pytest

from tests import fuzz_dem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(1000, 1024))
def test_random_small_system_matches_the_oracle(seed):
    n, pair, rebuilds, worst = fuzz_dem.run_case(seed)
    assert n > 0 and rebuilds >= 1
