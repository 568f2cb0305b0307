"""The C++ hosts above the C ABI run on the GPU box: the minimal example (library's own host
builder) and — when it was built, i.e. the reference header was present at build time — the
reference's speedtest GPU section re-hosted on the engine with the REAL tiny_bvh.h building the
layouts (the drop-in situation end to end, in the reference's own language)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "examples", "_build")


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [5, 8, 10])
def test_minimal_gpu_example(layout):
    exe = os.path.join(BUILD, "minimal_gpu")
    if not os.path.exists(exe):
        pytest.skip("examples/_build/minimal_gpu not built (run __graft_entry__.build())")
    out = subprocess.run([exe, str(layout)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "of 1024 rays hit" in out.stdout


@pytest.mark.gpu
def test_speedtest_gpu_section_with_real_tinybvh():
    exe = os.path.join(BUILD, "speedtest_gpu_section")
    if not os.path.exists(exe):
        pytest.skip("needs the reference header at build time")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all layouts agree with BVH::Intersect" in out.stdout


@pytest.mark.gpu
def test_wavefront_demos_with_real_tinybvh():
    """tiny_bvh_gpu.cpp:128-158 and tiny_bvh_gpu2.cpp:187-198 re-hosted on tbvh_wavefront_* with real tinybvh objects (BVH8_CWBVH::Build /
    BuildHQ, BVH_GPU TLAS over BLASInstance records rebuilt per frame); the program checks convergence and agreement with the CPU library."""
    exe = os.path.join(BUILD, "wavefront_demos")
    if not os.path.exists(exe):
        pytest.skip("needs the reference header at build time")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "wavefront demos ok" in out.stdout


@pytest.mark.gpu
def test_reference_minimal_gpu_main_unmodified():
    """/root/reference/tiny_bvh_minimal_gpu.cpp compiled with ZERO edits against include/shim/tiny_ocl.h (tinyocl::Buffer / Kernel("traverse.cl",
    "batch_ailalaine") / SetArguments / Run over the C ABI; __graft_entry__.build()) prints, for its 1024 rays, what the real
    tinybvh::BVH::Intersect finds on the CPU for the same rand() sequence (examples/ref_minimal_check.cpp, examples/fixed_rand.c): its own output."""
    exe, chk = os.path.join(BUILD, "ref_minimal_gpu"), os.path.join(BUILD, "ref_minimal_check")
    if not (os.path.exists(exe) and os.path.exists(chk)):
        pytest.skip("needs the reference checkout at build time")
    got = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert got.returncode == 0, got.stdout[-500:] + got.stderr[-500:]
    want = subprocess.run([chk], capture_output=True, text=True, timeout=120)
    assert want.returncode == 0
    g = [l for l in got.stdout.split("\n") if l.startswith("ray ")]
    w = [l for l in want.stdout.split("\n") if l.startswith("ray ")]
    assert len(g) == 1024 and len(w) == 1024
    # Two PROGRAMS build the rays (the reference's main and the checker): the compiler may round `Ray( O, D )`'s normalisation differently in the two
    # translation units, so a printed distance may differ in its last digit (seen: 54 lines in 1024, all by one ulp; traced from ONE process the engine's records are
    # bit-identical to BVH::Intersect on this very scene — tools/debug/shim_bits.cpp, and tests/test_speedtest_blocks_in_tinyocl_names below).  Bar:
    # every distance within BASELINE.json's 1e-5 relative (observed: 1e-6, one ulp), and the text of at least 90 % of the lines identical (observed: 95 %).
    tg = [float(l.rsplit(" ", 1)[1]) for l in g]
    tw = [float(l.rsplit(" ", 1)[1]) for l in w]
    worst = max(abs(a - b) / max(abs(b), 1e-30) for a, b in zip(tg, tw))
    same = sum(a == b for a, b in zip(g, w))
    assert worst <= 1e-5 and same >= 920, (worst, same, [(a, b) for a, b in zip(g, w) if a != b][:5])


@pytest.mark.gpu
def test_speedtest_blocks_in_tinyocl_names():
    """The three GPU blocks of tiny_bvh_speedtest.cpp:1092-1241 in tinyocl's own names and statement order (Buffer, CopyToDevice, Kernel, SetArguments,
    Run with a cl_event, clGetEventProfilingInfo, CopyFromDevice) on include/shim/tiny_ocl.h, layouts from the real BuildHQ; t bit-equal to BVH::Intersect."""
    exe = os.path.join(BUILD, "shim_speedtest_blocks")
    if not os.path.exists(exe):
        pytest.skip("needs the reference header at build time")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shim speedtest blocks ok" in out.stdout
