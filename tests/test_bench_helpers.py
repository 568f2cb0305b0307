"""Host logic of bench.py that has bitten before: grouping the traversal dispatches of a rocprofv3 --pmc child run into queries (a probed query is two
dispatches, an unprobed one is one; the first kernel of a pair has two forms while the coherent-schedule tuner measures)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

COH = "void tbvh::(anonymous namespace)::k_cwbvh<false, 8, 16, 8, true, false, 0, 5, 3, 0, 8>(...)"
COH_STRICT = "void tbvh::(anonymous namespace)::k_cwbvh<false, 8, 16, 1, false, false, 0, 5, 4, 0, 8>(...)"
COH_SPLIT = "void tbvh::(anonymous namespace)::k_cwbvh<false, 6, 16, 8, true, false, 0, 5, 3, 16, 7>(...)"
INC = "void tbvh::(anonymous namespace)::k_cwbvh<false, 8, 16, 1, false, false, 0, 13, 2, 0, 8>(...)"
INC_SPLIT = "void tbvh::(anonymous namespace)::k_cwbvh<false, 8, 16, 1, false, false, 0, 13, 2, 16, 6>(...)"
PLAIN = "void tbvh::(anonymous namespace)::k_cwbvh<false, 8, 16, 1, false, false, 0, 5, 0, 16, 8>(...)"
PLAIN_BIG = "void tbvh::(anonymous namespace)::k_cwbvh<false, 8, 16, 1, false, false, 0, 8, 0, 0, 8>(...)"


def group(seq):
    ids = list(range(10, 10 + len(seq)))
    return [[i - 10 for i in q] for q in bench.group_dispatches_into_queries(ids, dict(zip(ids, seq)))]


def test_probed_scene_every_query_is_a_pair():
    seq = [COH, INC, COH_SPLIT, INC_SPLIT, COH_SPLIT, INC_SPLIT] + [COH, INC] * 6
    assert group(seq) == [[2 * k, 2 * k + 1] for k in range(9)]


def test_pairs_while_the_tuner_alternates_the_first_kernel():
    seq = [COH, INC, COH_STRICT, INC, COH, INC]
    assert group(seq) == [[0, 1], [2, 3], [4, 5]]


def test_small_preparation_batches_are_single_dispatches():
    # 4.2 M-ray child on a probed scene: the third preparation batch (1.4 M rays) is below the probe's threshold
    seq = [COH_SPLIT, INC_SPLIT, COH_SPLIT, INC_SPLIT, PLAIN] + [COH_SPLIT, INC_SPLIT] * 6
    g = group(seq)
    assert len(g) == 9 and g[2] == [4] and g[-1] == [15, 16]


def test_unprobed_scene_is_one_dispatch_per_query():
    assert group([PLAIN_BIG] * 9) == [[k] for k in range(9)]
