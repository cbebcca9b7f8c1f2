"""The hand-derived known-answer cases of the two stateful map roles (tests/kat_cases.py: sepclusters, raycast update sweep)
on the HIP library: the same numbers that pin the oracle, without the oracle in between.  With an oracle that cannot be pinned
against the reference itself these cases are the independent pin of k_col_* / k_counted_range / k_sep_erase and of
k_raycast / k_ray_sweep - and of the host control code around them (driver_aux.h)."""
import pytest

import kat_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_its", [0, 1, 3])
def test_sepclusters_two_islands_by_hand(hip, n_its):
    kat_cases.sepclusters_case(hip, n_its)


@pytest.mark.parametrize("new_rule", [True, False])
@pytest.mark.parametrize("n_its", [1, 3])
def test_raycast_update_three_rays_by_hand(hip, new_rule, n_its):
    kat_cases.raycast_case(hip, new_rule, n_its)
