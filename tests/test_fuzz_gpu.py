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


@pytest.mark.parametrize("seed", range(100, 112))
def test_random_coupled_step_matches_the_oracle(seed):
    """tests/fuzz_cloud.py: random force switches, drag model, sub-cycling, mesh and smoothing parameters through the
    coupled step (drag assembly, DEM sub-steps, cell owner, scatter, smoothing, calcTcFields) against the oracle."""
    from tests import fuzz_cloud
    n, drag, mesh_n, on, smoothed = fuzz_cloud.run_case(seed)
    assert n > 0
