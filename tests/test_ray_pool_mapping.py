"""The ray pool's chunk order (tinybvh_amd/csrc/ray_pool.h: acquire), restated on the host: stripe s hands out, as its k-th chunk, chunk
k * P + ((s + 5 k) mod P) of the batch, and stops at the first chunk that begins beyond the batch.  Whatever the batch size and the
number of stripes, every ray is handed out exactly once, and a stripe's chunks wander over all positions of a row (the point of the
rotation: for rays in image order a fixed position would be a set of pixel columns)."""
import numpy as np
import pytest

CHUNK = 64


def handed_out(n_rays: int, parts_log2: int):
    P = 1 << parts_log2
    seen = np.zeros(n_rays, np.int32)
    positions = [set() for _ in range(P)]
    for s in range(P):
        k = 0
        while True:
            pos = (s + 5 * k) & (P - 1)
            first = (k * P + pos) * CHUNK
            if first >= n_rays:
                break                      # "exhausted": every later row lies beyond the batch altogether
            seen[first:min(first + CHUNK, n_rays)] += 1
            positions[s].add(pos)
            k += 1
    return seen, positions


@pytest.mark.parametrize("parts_log2", [0, 3, 5, 6])
@pytest.mark.parametrize("n_rays", [1, 63, 64, 65, 4096, 100_003, 1 << 20])
def test_every_ray_is_handed_out_exactly_once(n_rays, parts_log2):
    seen, _ = handed_out(n_rays, parts_log2)
    assert seen.min() == 1 and seen.max() == 1


def test_a_stripe_visits_every_position_of_a_row():
    _, positions = handed_out(1 << 20, 5)
    assert all(len(p) == 32 for p in positions)
