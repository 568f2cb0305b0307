"""Host logic of bench.py that has bitten before: grouping the traversal dispatches of a rocprofv3 --pmc child run into queries (a probed query is two
dispatches, an unprobed one is one; the first kernel of a pair has two forms while the coherent-schedule tuner measures)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

COH = "void tbvh::(anonymous namespace)::k_cwbvh<false, 8, 16, 8, true, false, 0, 5, 3, 0, 8, 0>(...)"
COH_STRICT = "void tbvh::(anonymous namespace)::k_cwbvh<false, 8, 16, 1, false, false, 0, 5, 4, 0, 8, 0>(...)"
COH_SPLIT = "void tbvh::(anonymous namespace)::k_cwbvh<false, 6, 16, 8, true, false, 0, 5, 3, 16, 7, 0>(...)"
INC = "void tbvh::(anonymous namespace)::k_cwbvh<false, 8, 16, 1, false, false, 0, 13, 2, 0, 8, 0>(...)"
INC_SPLIT = "void tbvh::(anonymous namespace)::k_cwbvh<false, 8, 16, 1, false, false, 0, 13, 2, 16, 6, 0>(...)"
PLAIN = "void tbvh::(anonymous namespace)::k_cwbvh<false, 8, 16, 1, false, false, 0, 5, 0, 16, 8, 0>(...)"
PLAIN_BIG = "void tbvh::(anonymous namespace)::k_cwbvh<false, 8, 16, 1, false, false, 0, 8, 0, 0, 8, 0>(...)"


PACKET = "void tbvh::(anonymous namespace)::k_cwbvh_packet<false, false>(...)"


def group(seq):
    ids = list(range(10, 10 + len(seq)))
    return [[i - 10 for i in q] for q in bench.group_dispatches_into_queries(ids, dict(zip(ids, seq)))]


def test_probed_scene_every_query_is_a_pair():
    seq = [COH, INC, COH_SPLIT, INC_SPLIT, COH_SPLIT, INC_SPLIT] + [COH, INC] * 6
    assert group(seq) == [[2 * k, 2 * k + 1] for k in range(9)]


def test_pairs_while_the_tuner_alternates_the_first_kernel():
    seq = [COH, INC, COH_STRICT, INC, COH, INC]
    assert group(seq) == [[0, 1], [2, 3], [4, 5]]


def test_pairs_when_the_first_kernel_is_the_packet_traversal():
    seq = [PACKET, INC, COH, INC, COH_STRICT, INC, PACKET, INC]
    assert group(seq) == [[0, 1], [2, 3], [4, 5], [6, 7]]


def test_small_preparation_batches_are_single_dispatches():
    # 4.2 M-ray child on a probed scene: the third preparation batch (1.4 M rays) is below the probe's threshold
    seq = [COH_SPLIT, INC_SPLIT, COH_SPLIT, INC_SPLIT, PLAIN] + [COH_SPLIT, INC_SPLIT] * 6
    g = group(seq)
    assert len(g) == 9 and g[2] == [4] and g[-1] == [15, 16]


def test_unprobed_scene_is_one_dispatch_per_query():
    assert group([PLAIN_BIG] * 9) == [[k] for k in range(9)]


# ---- `python bench.py --gpus N` without torchrun: which devices one process drives (the reference has no multi-device at all, tiny_ocl.h:362-364) ----
import json  # noqa: E402
import subprocess  # noqa: E402

import pytest  # noqa: E402


def test_resolve_devices_plain():
    assert bench.resolve_devices(1, 1) == [0]
    assert bench.resolve_devices(8, 8) == list(range(8))
    assert bench.resolve_devices(2, 8) == [0, 1]


def test_resolve_devices_refuses_a_1_gpu_number_under_an_n_gpu_flag():
    with pytest.raises(ValueError, match="--gpus 8 but 1 HIP device"):
        bench.resolve_devices(8, 1)
    with pytest.raises(ValueError):
        bench.resolve_devices(2, 0)
    with pytest.raises(ValueError):
        bench.resolve_devices(0, 4)


def test_resolve_devices_map():
    assert bench.resolve_devices(2, 1, "0,0") == [0, 0]          # two contexts on one device: how a 1-GPU box exercises the N-context path
    assert bench.resolve_devices(3, 2, "1, 0,1") == [1, 0, 1]
    with pytest.raises(ValueError, match="lists 2 devices for --gpus 3"):
        bench.resolve_devices(3, 4, "0,1")
    with pytest.raises(ValueError, match="names device"):
        bench.resolve_devices(2, 1, "0,1")
    with pytest.raises(ValueError, match="comma-separated"):
        bench.resolve_devices(2, 2, "a,b")


def test_plain_gpus_n_without_devices_says_so_and_exits_nonzero():
    """No GPU here: `python bench.py --gpus 2` (no torchrun) must not fall back to anything — one JSON line naming the problem, exit code 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TBVH_BENCH_DEVICE_MAP", "TBVH_BENCH_FORCE_DIST")}
    import tinybvh_amd as tb
    if tb.device_count() >= 2:
        pytest.skip("two devices visible: the run would start")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    line = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] is None and "HIP device" in line["error"]
