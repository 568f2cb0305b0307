"""pytest configuration: the `gpu` marker, and shared fixtures for the checkers."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        import tinybvh_amd as tb
        return tb.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The restated oracle under the LIBRARY's tie rule (at exactly equal t the smaller prim, then the smaller instance, wins;
    oracle/tbvh_oracle.c: orc_set_tie_rule; tinybvh_amd/csrc/device_common.h: hit_wins): against it the GPU kernels must report
    the exact prim, ties included.  Everything else is BVH::Intersect restated."""
    from oracle_lib import Oracle
    return Oracle(tie_rule=1)


@pytest.fixture(scope="session")
def oracle_ties(oracle):
    return oracle


@pytest.fixture(scope="session")
def oracle_ref():
    """The restated oracle under the REFERENCE's tie rule (the later test wins, tiny_bvh.h:1656): what the bit-for-bit comparisons
    with oracle/_ref (the real tiny_bvh.h) and with the golden vectors generated from it need."""
    from oracle_lib import Oracle
    return Oracle(tie_rule=0)


@pytest.fixture(scope="session")
def reference():
    from oracle_lib import Reference, have_reference
    if not have_reference():
        pytest.skip("oracle/_ref/libtinybvh_ref.so not built (needs the reference checkout at build time)")
    return Reference()


@pytest.fixture(scope="session")
def ctx():
    import tinybvh_amd as tb
    c = tb.Context(0)
    yield c
    c.close()
