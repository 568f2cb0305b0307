"""bench_detail.py — the side legs of bench.py (`python bench.py --detail`): everything that is NOT the contract line's own measurement.

Each leg runs outside the timed region, on rank 0 of a one-GPU run, and lands in the sidecar file the line names (`detail_file`), never on the
line itself: BASELINE configs 1, 2, 5, the 64 M-ray batch from one process over several contexts, the reference's own OpenCL kernels on the same
GPU, blobs and rays, the same kernels on scenes inside / beyond the Infinity Cache, the rotated scene, the other two layouts, host-side rays,
whole wavefront frames and the device-side maintenance operations.  The oracle (tests/oracle_lib.py) appears here as the checker and as the
counter of node visits / triangle tests only."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
BENCH = os.path.join(ROOT, "bench.py")


def _bench():
    import bench
    return bench


def strong_one_process(tb, R, sc0, verts, eye, view, side4, k, log, devices=None):
    """Config 4's batch (side4 x side4 camera rays, bounced to depths 1-3 in thirds) from ONE process over k contexts — context i on device
    i mod (visible devices) — through tbvh_intersect_sharded_device: the BVH uploaded once per context, every shard generated, traced and kept
    on its device, one host thread enqueueing all launches.  Reports the batch rate, per-device kernel ms and the host dispatch gap."""
    from tinybvh_amd.sharding import shard_range
    n_dev = tb.device_count()
    n4 = side4 * side4
    cam4 = R.camera(eye, view, side4, side4, 1, 1)
    ctxs = [tb.Context(devices[i] if devices else i % n_dev) for i in range(k)]
    try:
        h = sc0.host
        reps = [tb.BVH8_CWBVH(c).Upload(h.blob(0, np.uint32, 4), h.blob(1, np.uint32, 4)) for c in ctxs]
        d_rays, counts = [], []
        for i, (c, r) in enumerate(zip(ctxs, reps)):
            b, e = shard_range(n4, i, k)
            m = e - b
            dv = c.malloc(verts.nbytes); c.to_device(dv, verts)
            d_a, d_b = c.malloc(max(m, 1) * 64), c.malloc(max(m, 1) * 64)
            c.generate_primary(cam4, d_a, b, m)
            r.intersect_device(d_a, m)
            t3 = m // 3
            c.generate_bounce(dv, d_a, d_b, m, 4001)
            r.intersect_device(d_b + t3 * 64, m - t3)
            c.generate_bounce(dv, d_b + t3 * 64, d_b + t3 * 64, m - t3, 4002)
            r.intersect_device(d_b + 2 * t3 * 64, m - 2 * t3)
            c.generate_bounce(dv, d_b + 2 * t3 * 64, d_b + 2 * t3 * 64, m - 2 * t3, 4003)
            c.synchronize()
            c.free(d_a); c.free(dv)
            d_rays.append(d_b); counts.append(m)
        tb.intersect_sharded_device(reps, d_rays, counts, fresh=True)     # warm-up
        wall, kms, dms = [], [], []
        for _ in range(3):
            t0 = time.perf_counter()
            km, dm = tb.intersect_sharded_device(reps, d_rays, counts, fresh=True)
            wall.append(time.perf_counter() - t0); kms.append(km); dms.append(dm)
        w = float(np.mean(wall))
        return {"workload": f"one {n4}-ray diffuse batch (depth 1-3) from ONE process over {k} contexts on {min(k, n_dev)} device(s), BVH replicated, no collective",
                "entry_point": "tbvh_intersect_sharded_device", "contexts": k, "devices": min(k, n_dev), "rays": n4, "ms_per_batch": w * 1e3, "mrays": n4 / w / 1e6,
                "kernel_ms_per_device": [float(x) for x in np.mean(np.array(kms), 0)], "host_dispatch_ms_per_device": [float(x) for x in np.mean(np.array(dms), 0)]}
    finally:
        for c in ctxs:
            c.close()



def configs_1_and_2(tb, ctx, R, scenes):
    """BASELINE.json configs[0] and [1] on the Sponza stand-in with the speedtest's 1 M camera rays (tiny_bvh_speedtest.cpp:1092-1141):
    config1  BVH::Build seconds and BVH::Intersect MRays/s on the host — the real tiny_bvh.h through oracle/_ref where that library travelled
             with the repo ("reference"), else the library's own builder and the C restatement ("port");
    config2  BVH_GPU (Aila-Laine) on this GPU: the HIP kernel, and the reference's own batch_ailalaine (traverse_bvh2.cl:209-219) on the SAME
             blobs and rays through ROCm OpenCL when oracle/_ref/libtinybvh_refocl.so loads, with the agreement of the two hit sets."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle, Reference, ReferenceOpenCL, compare_hits, have_reference
    verts, label = scenes.get("sponza")
    side = 1024
    n = side * side
    cam = R.camera(*scenes.SPONZA_CAMERAS[0], side, side, 1, 1)
    d = ctx.malloc(n * 64)
    ctx.generate_primary(cam, d, 0, n)
    rays = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(rays, d)
    out = {}
    cores = _bench().usable_cores()
    # config 1
    if have_reference():
        ref = Reference()
        t0 = time.time(); rs = ref.build(verts, hq=False, threaded=False); build_s = time.time() - t0
        sec_mt, hits = rs.time_mt(1, rays, threads=cores)
        sec_1, _ = rs.time_mt(1, rays[: n // 8], threads=1)
        out["config1"] = {"kind": "reference", "scene": label, "rays": n, "bvh_build_s": build_s, "bvh_intersect_mrays": n / sec_mt / 1e6, "cores": cores,
                          "bvh_intersect_mrays_1_thread": (n // 8) / sec_1 / 1e6, "hits": int(hits)}
    else:
        orc = Oracle()
        t0 = time.time(); h = tb.HostBVH(verts, tb.LAYOUT_BVH2_WALD); build_s = time.time() - t0
        k = 100_000
        t0 = time.time(); orc.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, rays[:k]); sec = time.time() - t0
        out["config1"] = {"kind": "port", "scene": label, "rays": k, "bvh_build_s": build_s, "bvh_intersect_mrays": k / sec / 1e6, "cores": 1}
    # config 2
    sc = tb.BVH_GPU(ctx).Build(verts)
    ms = []
    for p_ in range(24):                 # (the first 16: the 8-wide copy is made, its tuner settles)
        sc.intersect_device_fresh(d, n, 1e30)
        t = ctx.time_last_ms()
        if p_ >= 16:
            ms.append(t)
    mine = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(mine, d)
    c2 = {"scene": label, "rays": n, "layout": "BVH_GPU", "bvh_gpu_mrays": n / (float(np.median(ms)) * 1e-3) / 1e6, "ref_opencl_mrays": "n/a", "ratio": "n/a", "hitmiss_diff": "n/a"}
    try:
        ocl = ReferenceOpenCL()
        h = sc.host
        theirs, ref_ms = ocl.run(5, [h.blob(0, np.uint32, 16), h.blob(1, np.uint32, 1), verts], rays, passes=5)
        cmp_ = compare_hits(mine[: theirs.shape[0]], theirs, rtol=1e-4)   # the .cl kernels use native_recip and strict comparisons: t to 1e-4
        c2.update({"ref_opencl_mrays": theirs.shape[0] / (ref_ms * 1e-3) / 1e6, "ref_kernel": "batch_ailalaine (traverse_bvh2.cl) through ROCm OpenCL, same blobs, same rays",
                   "hitmiss_diff": cmp_["hitmiss"], "prim_diff": cmp_["prim_mismatch"], "opencl_device": ocl.device})
        c2["ratio"] = c2["bvh_gpu_mrays"] / c2["ref_opencl_mrays"]
    except Exception as e:
        c2["ref_opencl_error"] = repr(e)[:300]
    out["config2"] = c2
    sc.free(); ctx.free(d)
    return out


def reference_opencl_headline(tb, ctx, sc, d_prim, d_diff, n, kern_ms, timed_got, par_stride, ns_par):
    """The reference's OWN kernel for this path — batch_cwbvh (traverse_cwbvh.cl:554-570), compiled by ROCm OpenCL from the source text embedded
    in oracle/_ref/libtinybvh_refocl.so — on the SAME GPU, the SAME BVH8_CWBVH blobs and the SAME 16.7 M-ray primary and diffuse batches the
    metric is quoted on, next to the HIP kernels' timed launches; plus the agreement of the two hit sets (the .cl kernel derives rD with
    native_recip and compares strictly: t to 1e-4)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import ReferenceOpenCL, compare_hits
    ocl = ReferenceOpenCL()
    h = sc.host
    blobs = [h.blob(0, np.uint32, 4), h.blob(1, np.uint32, 4)]
    out = {"ref_kernel": "batch_cwbvh (traverse_cwbvh.cl) through ROCm OpenCL, same blobs, same rays, same GPU", "opencl_device": ocl.device, "rays_per_batch": n}
    tot_ref = tot_hip = 0.0
    for kind, d in (("primary", d_prim), ("diffuse", d_diff)):
        # the HIP side of the comparison = the strided sample of the records the TIMED launches left (this scene has been refitted to moved
        # vertices by the device_side_ops leg since; the blobs on the host, which the OpenCL kernel gets, and the rays have not changed)
        rays = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(rays, d)
        rays["t"] = 1e30; rays["u"] = 0; rays["v"] = 0; rays["prim"] = 0
        theirs, ref_ms = ocl.run(10, blobs, rays, passes=3)
        del rays
        cmp_ = compare_hits(timed_got[kind], theirs[::par_stride][:ns_par], rtol=1e-4)
        hip_ms = float(np.mean(kern_ms[kind]))
        out[kind] = {"hip_mrays": n / (hip_ms * 1e-3) / 1e6, "ref_opencl_mrays": theirs.shape[0] / (ref_ms * 1e-3) / 1e6, "ratio": ref_ms / hip_ms,
                     "hitmiss_diff": cmp_["hitmiss"], "prim_diff": cmp_["prim_mismatch"], "rays_compared": cmp_["n"]}
        tot_ref += ref_ms; tot_hip += hip_ms
        del theirs
    out["primary_plus_diffuse"] = {"hip_mrays": 2 * n / (tot_hip * 1e-3) / 1e6, "ref_opencl_mrays": 2 * n / (tot_ref * 1e-3) / 1e6, "ratio": tot_ref / tot_hip}
    return out


def hbm_regime(a, log):
    """north_star asks for ">= 50 % HBM roofline on the node-fetch loop"; the bench's own scene (2.83 M triangles: 0.2 GB of tree, 0.44 GB with
    the incoherent-batch copies) is served by the L2s and the 256 MB Infinity Cache to a degree the TCC counters cannot state (they count L2
    misses, whoever serves them).  So the SAME kernels, builder (the library's host SAH builder — the product's default), launches and counters run
    on two more sizes of the same street generator that bracket it: 1 M triangles (0.07 GB, 0.16 GB with the copies: everything beyond the L2s comes from the Infinity
    Cache) and 12 M triangles (0.9 GB: mostly from HBM), 4.19 M camera rays and bounce rays (depth 1-3) per launch.  Per scene a child of this
    script times the launches and counts node visits S / triangle tests T per ray with the oracle's mirror; children under `rocprofv3 --pmc`
    give bytes beyond the L2s and the mean latency of an L2 miss (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ).  The latency separates the two
    regimes; `frac_of_hbm_peak` of the 12 M scene is the figure north_star asks for, on the default builder."""
    import copy
    import subprocess
    import tempfile
    res = {"scenes": {}, "peak_tb_per_s": 8.0, "builder": "library host builder (binned SAH + SAH-optimal wide collapse): the default"}
    tmpdir = tempfile.mkdtemp(prefix="tbvh_hbm_", dir="/tmp")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TBVH_BENCH_FORCE_DIST"):
        env.pop(k, None)
    try:
        for tag, scene in (("fits_infinity_cache", "street1m"), ("beyond_infinity_cache", "street12m")):
            b = copy.copy(a)
            b.scene, b.side, b.device_build, b.layout, b.variant = scene, 2048, False, 10, 0
            b.blob_cache = os.path.join(tmpdir, scene + ".cwbvh")
            b.coh_pin = None         # (the timing child lets the scene's tuner settle; the counter children are pinned to what it chose)
            env.pop("TBVH_COHERENT_TUNER", None)
            cmd = [sys.executable, BENCH, "--hbm-child", "--scene", b.scene, "--side", str(b.side), "--layout", "10", "--blob-cache", b.blob_cache]
            try:
                r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=400, check=True)
                out = json.loads([l for l in r.stdout.decode().split("\n") if l.startswith("{")][-1])
            except Exception as e:
                log(f"[bench] hbm_regime child ({scene}) failed: {e!r}")
                res["scenes"][tag] = {"error": repr(e)}
                continue
            n = out["rays_per_launch"]
            b.coh_pin = {1: "0", 2: "2", 3: "3"}.get(out.get("coherent_schedule"), "0")
            pm = _bench().live_counters(b, log, passes=("FETCH_SIZE", "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum")) if not a.no_pmc else None
            row = {"scene": out["scene"], "triangles": out["triangles"], "bvh_mb": out["bvh_mb"], "rays_per_launch": n, "tree": out["tree"],
                   "coherent_schedule": {1: "deferred + gated", 2: "strict", 3: "one traversal per wave"}.get(out.get("coherent_schedule"), "n/a (no per-launch probe on a scene of this size: the strict schedule)")}
            for kind in ("primary", "diffuse"):
                alg = 80.0 * out.get(kind + "_S", 0.0) + 48.0 * out.get(kind + "_T", 0.0) + 80.0
                sec = out[kind + "_ms"] * 1e-3
                k_ = {"mrays": out[kind + "_mrays"], "node_visits_per_ray": out.get(kind + "_S"), "triangle_tests_per_ray": out.get(kind + "_T"),
                      "algorithmic_bytes_per_ray": alg, "algorithmic_tb_per_s": alg * n / sec / 1e12}
                c = (pm or {}).get(kind, {})
                if "FETCH_SIZE" in c:
                    tr = c["FETCH_SIZE"] * 2048.0
                    k_.update(fetched_bytes_per_ray=tr / n, fetched_tb_per_s=tr / sec / 1e12, frac_of_hbm_peak=tr / sec / 8e12)
                if c.get("TCC_EA0_RDREQ_sum"):
                    k_["mean_l2_miss_latency_cycles"] = c["TCC_EA0_RDREQ_LEVEL_sum"] / c["TCC_EA0_RDREQ_sum"]
                row[kind] = k_
            res["scenes"][tag] = row
        res["traffic_source"] = "live: rocprofv3 --pmc FETCH_SIZE (x 2, guide correction; read side only) and TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ child runs per scene"
    finally:
        import shutil
        shutil.rmtree(tmpdir, ignore_errors=True)
    return res


LAYOUT_BYTES = {10: (80, 48), 8: (64, 48), 5: (64, 52)}   # node bytes, bytes per triangle test (SURVEY par. 8(d))


def scene_leg(a, log, scene, side, layout, ref_ocl, valu_ceiling, note="", env_extra=None):
    """One more (scene, layout) measured like the headline: a child of this script times `side`^2 camera and bounce rays (HIP events) and counts
    S / T with the oracle's mirror (and, ref_ocl, times the reference's own OpenCL kernel of the layout on the same blobs and rays); two more
    children under `rocprofv3 --pmc` give the bytes beyond the L2s and the VALU counters of the same launches."""
    import copy
    import subprocess
    import tempfile
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TBVH_BENCH_FORCE_DIST", "TBVH_COHERENT_TUNER"):
        env.pop(k, None)
    env.update(env_extra or {})
    tmpdir = tempfile.mkdtemp(prefix="tbvh_leg_", dir="/tmp")
    b = copy.copy(a)
    b.env_extra = env_extra
    b.scene, b.side, b.device_build, b.layout, b.variant, b.coh_pin = scene, side, False, layout, 0, None
    b.blob_cache = os.path.join(tmpdir, scene + ".cwbvh") if layout == 10 else ""
    cmd = [sys.executable, BENCH, "--hbm-child", "--scene", scene, "--side", str(side), "--layout", str(layout)] + \
          (["--blob-cache", b.blob_cache] if b.blob_cache else []) + (["--ref-ocl"] if ref_ocl else [])
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, check=True)
        out = json.loads([l for l in r.stdout.decode().split("\n") if l.startswith("{")][-1])
        n = out["rays_per_launch"]
        b.coh_pin = {1: "0", 2: "2", 3: "3"}.get(out.get("coherent_schedule"), "0")      # the counter children run the schedule the timing child's tuner chose
        pm = _bench().live_counters(b, log, passes=("FETCH_SIZE", "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES")) if not a.no_pmc else None
        nb, tbytes = LAYOUT_BYTES[layout]
        row = {"scene": out["scene"], "triangles": out["triangles"], "layout": {5: "BVH_GPU", 8: "BVH4_GPU", 10: "BVH8_CWBVH"}[layout], "bvh_mb": out["bvh_mb"], "rays_per_launch": n,
               "coherent_schedule": {1: "deferred + gated", 2: "strict", 3: "one traversal per wave"}.get(out.get("coherent_schedule"), "undecided") if layout == 10 else None, "note": note}
        if "opencl_device" in out:
            row["opencl_device"] = out["opencl_device"]
        if "ref_opencl_error" in out:
            row["ref_opencl_error"] = out["ref_opencl_error"]
        for kind in ("primary", "diffuse"):
            S, T = out.get(kind + "_S", 0.0), out.get(kind + "_T", 0.0)
            alg = 80.0 + nb * S + tbytes * T
            sec = out[kind + "_ms"] * 1e-3
            k_ = {"mrays": out[kind + "_mrays"], "launch_ms": out[kind + "_ms"], "node_visits_per_ray": S, "triangle_tests_per_ray": T,
                  "algorithmic_bytes_per_ray": alg, "algorithmic_tb_per_s": alg * n / sec / 1e12}
            if kind + "_ref_opencl_mrays" in out:
                k_.update(ref_opencl_mrays=out[kind + "_ref_opencl_mrays"], ratio=out[kind + "_ratio"], hitmiss_diff=out[kind + "_hitmiss_diff"], prim_diff=out[kind + "_prim_diff"])
            c = (pm or {}).get(kind, {})
            if "FETCH_SIZE" in c:
                tr = c["FETCH_SIZE"] * 2048.0
                k_["fabric"] = {"fetched_bytes_per_ray": tr / n, "achieved_gbps": tr / sec / 1e9, "peak_gbps": 8000.0, "frac": tr / sec / 8e12}
            if c.get("SQ_INSTS_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
                rate = c["SQ_INSTS_VALU"] / sec / 1e9
                lane = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
                k_["valu"] = {"insts_valu_per_ray": c["SQ_INSTS_VALU"] * 64.0 / n, "issue_frac": rate / valu_ceiling if valu_ceiling else None, "lane_utilisation": lane}
            row[kind] = k_
        if pm:
            row["counters_source"] = pm.get("source")
        return row
    except Exception as e:
        log(f"[bench] scene leg ({scene}, layout {layout}) failed: {e!r}")
        return {"error": repr(e)[:300]}
    finally:
        import shutil
        shutil.rmtree(tmpdir, ignore_errors=True)


def config5_setup(tb, R, scenes, ctx, blas_layout):
    """BASELINE config 5's scene: 1000 instances of the Dragon stand-in (10 x 10 x 10 grid, scale 0.7, seeded rotation), 3840 x 2160 camera rays."""
    dv, dlabel = scenes.get("dragon")
    blas = tb.LAYOUT_CLASSES[blas_layout](ctx).Build(dv)
    side, scale = 10, 0.7
    g = np.stack(np.meshgrid(np.arange(side), np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    ang = (np.arange(g.shape[0]) * 0.37).astype(np.float32)
    c_, s_ = np.cos(ang), np.sin(ang)
    T = np.zeros((g.shape[0], 4, 4), np.float32)
    T[:, 0, 0] = c_ * scale; T[:, 0, 2] = s_ * scale; T[:, 1, 1] = scale; T[:, 2, 0] = -s_ * scale; T[:, 2, 2] = c_ * scale; T[:, 3, 3] = 1
    T[:, :3, 3] = g * 2.0
    inst = tb.make_instances(T, np.zeros(g.shape[0], np.uint32))
    W_, H_ = 3840, 2160
    ext = 2.0 * side
    cam = R.camera((-0.6 * ext, 0.8 * ext, -0.9 * ext), (0.62, -0.38, 0.68), W_, H_, 1, 1)
    tlas = tb.TLAS(ctx).Build(inst, [blas])
    return dlabel, blas, tlas, cam, W_ * H_


def tlas_child(a, tb, R, scenes):
    """Config 5 in a process of its own: camera rays through the TLAS over BVH4_GPU BLASes (k_tlas4) — under rocprofv3 --pmc (--pmc-child) only
    the launches; otherwise also the TLAS over BVH8_CWBVH BLASes (entered through their 4-wide copies: k_tlas4 as well) next to the reference's traverse_tlas (traverse_tlas.cl:13-107, through
    wavefront2.cl's Extend as tiny_bvh_gpu2.cpp:191 launches it) on the same TLAS nodes, instance records, BLAS blobs and rays."""
    ctx = tb.Context(0)
    dlabel, blas, tlas, cam, nt = config5_setup(tb, R, scenes, ctx, 8)
    d = ctx.malloc(nt * 64)
    ctx.generate_primary(cam, d, 0, nt)
    ms = []
    for f in range(4):
        tlas.intersect_device_fresh(d, nt, 1e30); ctx.synchronize()
        if f:
            ms.append(ctx.time_last_ms())
    if a.pmc_child:
        ctx.close()
        return
    out = {"blas": dlabel, "camera_rays": nt, "k_tlas4_ms": float(np.mean(ms)), "k_tlas4_mrays": nt / float(np.mean(ms)) / 1e3}
    tlas.free(); blas.free()
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_lib import ReferenceOpenCL
        ocl = ReferenceOpenCL()
        _, blas8, tlas8, _, _ = config5_setup(tb, R, scenes, ctx, 10)
        ms8 = []
        for f in range(4):
            tlas8.intersect_device_fresh(d, nt, 1e30); ctx.synchronize()
            if f:
                ms8.append(ctx.time_last_ms())
        mine = np.zeros(nt, tb.RAY_DTYPE); ctx.from_device(mine, d)
        rays = mine.copy(); rays["t"] = 1e30; rays["u"] = 0; rays["v"] = 0; rays["prim"] = 0
        nodes, idx, irec = tlas8.Download()
        h = blas8.host
        ref, ref_ms = ocl.tlas_extend(nodes, idx, irec, h.blob(0, np.uint32, 4), h.blob(1, np.uint32, 4), rays, passes=3)
        mh, rh = mine["t"][: ref.shape[0]] < 1e30, ref[:, 0] < 1e30
        out.update({"cwbvh_blas_ms": float(np.mean(ms8)), "cwbvh_blas_mrays": nt / float(np.mean(ms8)) / 1e3, "ref_opencl_traverse_tlas_ms": ref_ms,
                    "ref_opencl_traverse_tlas_mrays": ref.shape[0] / ref_ms / 1e3, "ratio_cwbvh_blas": ref_ms / float(np.mean(ms8)),
                    "cwbvh_blas_note": "closest-hit queries enter BVH8_CWBVH BLASes through their 4-wide copies (k_tlas4; capi_scene.hip: blasView) since round 6",
                    "ratio_k_tlas4_bvh4_blas": ref_ms / float(np.mean(ms)), "hitmiss_diff": int((mh != rh).sum()), "opencl_device": ocl.device,
                    "ref_kernel": "traverse_tlas (traverse_tlas.cl:13-107) via wavefront2.cl Extend, BVH8_CWBVH BLAS (the configuration of tiny_bvh_gpu2.cpp), same TLAS / instances / rays"})
    except Exception as e:
        out["ref_opencl_error"] = repr(e)[:300]
    try:
        out["mixed_layouts"] = tlas_mixed_layouts(tb, R, scenes, ctx)
    except Exception as e:
        out["mixed_layouts"] = {"error": repr(e)[:300]}
    print(json.dumps(out), flush=True)
    ctx.close()


def tlas_mixed_layouts(tb, R, scenes, ctx):
    """BLASes of three layouts and sizes under ONE TLAS (BVH_GPU 100 k, BVH8_CWBVH 20 k, BVH4_GPU 5 k triangles; 1000 instances, the reference's traverse_tlas.cl:50-72
    allows two BLAS types): with the library's copies every BLAS is entered through a 4-wide form by closest-hit queries and an 8-wide one by any-hit queries —
    one kernel class per kind of query —; with TBVH_WIDE_COPY_MIN=0 (no copies) the TLAS runs the flat three-state loop."""
    def unit(m):
        m = m.copy(); m[:, :3] -= 0.5 * (m[:, :3].min(0) + m[:, :3].max(0)); m[:, :3] *= np.float32(1.6 / float((m[:, :3].max(0) - m[:, :3].min(0)).max()))
        return np.ascontiguousarray(m)
    meshes = [unit(scenes.blob(100_000, seed=3)), unit(scenes.blob(20_000, seed=4)), unit(scenes.blob(5_000, seed=5))]
    side = 10
    rng = np.random.default_rng(3)
    T = np.zeros((side ** 3, 4, 4), np.float32)
    g = np.stack(np.meshgrid(np.arange(side), np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    ang = rng.random(side ** 3).astype(np.float32) * 6.28
    c_, s_ = np.cos(ang), np.sin(ang)
    T[:, 0, 0] = c_ * 0.5; T[:, 0, 2] = s_ * 0.5; T[:, 1, 1] = 0.5; T[:, 2, 0] = -s_ * 0.5; T[:, 2, 2] = c_ * 0.5; T[:, 3, 3] = 1
    T[:, :3, 3] = g * 2.0
    inst = tb.make_instances(T, (np.arange(side ** 3) % 3).astype(np.uint32))
    W_, H_ = 2560, 1600
    nt = W_ * H_
    cam = R.camera((-12.0, 16.0, -18.0), (0.62, -0.38, 0.68), W_, H_, 1, 1)
    d = ctx.malloc(nt * 64); d_sh = ctx.malloc(nt * 64); d_occ = ctx.malloc(nt)
    rr = R.random_rays(1 << 22, (-1.0, -1.0, -1.0), (20.0, 20.0, 20.0), seed=9)
    d_r = ctx.malloc(rr.shape[0] * 64)
    res = {"blas": "BVH_GPU 100 k + BVH8_CWBVH 20 k + BVH4_GPU 5 k triangles (procedural blobs), 1000 instances", "camera_rays": nt, "random_rays": int(rr.shape[0])}
    old = os.environ.get("TBVH_WIDE_COPY_MIN")
    try:
        for label, env in (("with_copies", None), ("without_copies", "0")):
            if env is None:
                os.environ.pop("TBVH_WIDE_COPY_MIN", None)
            else:
                os.environ["TBVH_WIDE_COPY_MIN"] = env
            blas = [tb.BVH_GPU(ctx).Build(meshes[0]), tb.BVH8_CWBVH(ctx).Build(meshes[1]), tb.BVH4_GPU(ctx).Build(meshes[2])]
            tlas = tb.TLAS(ctx).Build(inst, blas)

            def timed(fn, passes=6):
                ms = []
                for p in range(passes):
                    fn(); ctx.synchronize()
                    if p:
                        ms.append(ctx.time_last_ms())
                return float(np.median(ms))
            ctx.generate_primary(cam, d, 0, nt)
            ms_cam = timed(lambda: tlas.intersect_device_fresh(d, nt, 1e30))
            rec = np.zeros(nt, tb.RAY_DTYPE); ctx.from_device(rec, d)
            ctx.generate_shadow(d, d_sh, nt, (10.0, 40.0, 10.0), 1e-4)
            ms_sh = timed(lambda: tlas.occluded_device(d_sh, nt, d_occ))
            ctx.to_device(d_r, rr)
            ms_r = timed(lambda: tlas.intersect_device_fresh(d_r, rr.shape[0], 1e30))
            res[label] = {"camera_mrays": nt / ms_cam / 1e3, "shadow_mrays": nt / ms_sh / 1e3, "random_mrays": rr.shape[0] / ms_r / 1e3,
                          "hits": int((rec["t"] < 1e30).sum()), "prim_checksum": int(rec["prim"][rec["t"] < 1e30].astype(np.uint64).sum())}
            tlas.free()
            for b in blas:
                b.free()
    finally:
        if old is None:
            os.environ.pop("TBVH_WIDE_COPY_MIN", None)
        else:
            os.environ["TBVH_WIDE_COPY_MIN"] = old
        for p_ in (d, d_sh, d_occ, d_r):
            ctx.free(p_)
    return res


def tlas_leg(a, log, valu_ceiling):
    """detail.tlas_1000_instances: the reference's traverse_tlas beside k_tlas8 / k_tlas4, and k_tlas4's counters (FETCH_SIZE; the SQ VALU group)."""
    import shutil
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TBVH_BENCH_FORCE_DIST"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH, "--tlas-child"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, check=True)
    out = json.loads([l for l in r.stdout.decode().split("\n") if l.startswith("{")][-1])
    res = {"vs_reference_opencl": out}
    if not a.no_pmc and shutil.which("rocprofv3"):
        cnt = {}
        for pass_ in ("FETCH_SIZE", "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES"):
            try:
                rows = _bench().pmc_dispatches(["--tlas-child", "--pmc-child"], pass_.split(), lambda kn: "k_tlas4" in kn)
                for cn in pass_.split():
                    vals = [r_.get(cn, 0.0) for r_ in rows][-3:]
                    cnt[cn] = float(np.mean(vals))
            except Exception as e:
                log(f"[bench] rocprofv3 --pmc {pass_!r} TLAS child failed: {e!r}")
        nt, sec = out["camera_rays"], out["k_tlas4_ms"] * 1e-3
        k_ = {"kernel": "k_tlas4 (BVH4_GPU BLASes), 3840 x 2160 camera rays", "launch_ms": out["k_tlas4_ms"]}
        if "FETCH_SIZE" in cnt:
            tr = cnt["FETCH_SIZE"] * 2048.0
            k_["fabric"] = {"fetched_bytes_per_ray": tr / nt, "achieved_gbps": tr / sec / 1e9, "peak_gbps": 8000.0, "frac": tr / sec / 8e12}
        if cnt.get("SQ_INSTS_VALU") and cnt.get("SQ_ACTIVE_INST_VALU"):
            rate = cnt["SQ_INSTS_VALU"] / sec / 1e9
            k_["valu"] = {"insts_valu_per_ray": cnt["SQ_INSTS_VALU"] * 64.0 / nt, "issue_frac": rate / valu_ceiling if valu_ceiling else None,
                          "lane_utilisation": cnt["SQ_THREAD_CYCLES_VALU"] / (64.0 * cnt["SQ_ACTIVE_INST_VALU"])}
        res["counters"] = k_
    return res


def host_rays_leg(tb, ctx, sc, d_prim, n):
    """The speedtest's literal call (tiny_bvh_speedtest.cpp:1110-1137): a HOST tinybvh::Ray[] (128-byte records) traced in place through
    tbvh_intersect, and the packed 64-byte form; from pageable memory (host threads pack into the library's pinned ring) and, packed, from
    page-locked memory of the library's (tbvh_pinned_malloc: the tinyocl::Buffer of this boundary; DMA straight from it).  84 bytes cross the link per ray (64 up, 20
    down); `frac_of_link` = that traffic over the call's wall time against the link's measured pinned hipMemcpyAsync rates."""
    up, down = ctx.link_bandwidth_gbps(1 << 28, 3)
    rays64 = np.zeros(n, dtype=tb.RAY_DTYPE); ctx.from_device(rays64, d_prim)
    rays64["t"] = 1e30; rays64["u"] = 0; rays64["v"] = 0; rays64["prim"] = 0
    rays128 = np.zeros((n, 32), np.uint32)
    rays128[:, :16] = rays64.view(np.uint32).reshape(n, 16)
    ideal_s = n * 64 / (up * 1e9) + n * 20 / (down * 1e9)
    out = {"rays": n, "link_h2d_gbps": up, "link_d2h_gbps": down, "bytes_per_ray_on_the_link": 84, "mrays_at_link_rate": n / ideal_s / 1e6,
           "call": "tbvh_intersect(scene, host Ray[], n, stride): replaces the memcpy loop + CopyToDevice + Kernel::Run + CopyFromDevice of tiny_bvh_speedtest.cpp:1110-1137"}
    first_hits = None
    pinned64 = ctx.pinned_array((n, 16), np.uint32)       # the packed array in page-locked memory of the library's (tbvh_pinned_malloc)
    pinned64[:] = rays64.view(np.uint32).reshape(n, 16)
    for tag, arr in (("stride_128", rays128), ("stride_64", rays64.view(np.uint32).reshape(n, 16)), ("stride_64_pinned", pinned64)):
        try:
            wall, kern = [], []
            for p_ in range(4):
                arr[:, 12] = np.float32(1e30).view(np.uint32); arr[:, 13:16] = 0
                t0 = time.perf_counter()
                sc.Intersect(arr)
                dt = time.perf_counter() - t0
                if p_:
                    wall.append(dt); kern.append(ctx.time_last_ms())
            w = float(np.median(wall))
            hits = int((arr[:, 12].view(np.float32) < 1e30).sum())
            if first_hits is None:
                first_hits = arr[:, 11:16].copy()
                same = True
            else:
                same = bool(np.array_equal(arr[:, 11:16], first_hits))
            out[tag] = {"mrays": n / w / 1e6, "ms_per_call": w * 1e3, "kernel_ms": float(np.median(kern)), "frac_of_link": ideal_s / w, "hits": hits, "records_equal_first_variant": same}
        except Exception as e:
            out[tag] = {"error": repr(e)[:300]}
    ctx.pinned_free(pinned64)
    return out


def wavefront_leg(tb, ctx, sc, d_verts, cam, light, side, seed):
    """Whole wavefront path-traced frames (Generate, {Extend, Shade} x 3, Connect; all queues on the device): config 4's pipeline end to end
    (wavefront.cl:93-274)."""
    wf = tb.Wavefront(ctx, side, side)
    try:
        st = None
        for f in range(3):
            st = wf.render(sc, d_verts, cam, light, (3000.0, 3000.0, 3000.0), max_depth=3, seed=seed + f)
        total = sum(st["extend_rays"]) + sum(st["shadow_rays"])
        return {"extend_rays": st["extend_rays"], "shadow_rays": st["shadow_rays"], "frame_ms": st["frame_ms"], "mrays_all_stages": total / st["frame_ms"] / 1e3}
    finally:
        wf.close()


def device_ops_leg(tb, ctx, sc, verts, d_verts):
    """The device-side maintenance operations on the bench scene: refit of the uploaded blob to displaced vertices (BVH::Refit, tiny_bvh.h:3055-3093,
    on the device) and a full LBVH rebuild.  NOTE: leaves `sc` refitted to the moved vertices — run it after everything that traces `sc`."""
    moved = verts.copy()
    moved[:, 1] += np.float32(1e-3) * np.sin(verts[:, 0]).astype(np.float32)
    ctx.to_device(d_verts, moved)
    n_tris = verts.shape[0] // 3
    sc.Refit((d_verts, n_tris), on_device=True); sc.Refit((d_verts, n_tris), on_device=True)
    ms_refit = ctx.time_last_ms()
    built = tb.BVH8_CWBVH(ctx).BuildOnDevice(moved); built.free()
    built = tb.BVH8_CWBVH(ctx).BuildOnDevice(moved)
    ms_build = ctx.time_last_ms()
    built.free()
    return {"refit_ms": ms_refit, "device_build_ms": ms_build, "triangles": n_tris}


def tlas_frames_leg(tb, ctx, R, scenes, frames=4):
    """BASELINE config 5: 1000 instances of the Dragon stand-in (10 x 10 x 10 grid, scale 0.7, seeded rotation, tiny_bvh_gpu2.cpp:113-119),
    3840 x 2160 camera rays per frame through the TLAS; every frame the BLAS vertices move and the BLAS is refitted on the device, the TLAS is
    rebuilt on the device from the frame's transforms, then the frame's rays are traced; plus 4 M incoherent rays."""
    dv, dlabel = scenes.get("dragon")
    blas = tb.BVH4_GPU(ctx).Build(dv)
    side, scale = 10, 0.7

    def frame_instances(t):
        g = np.stack(np.meshgrid(np.arange(side), np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
        ang = (t * 0.5 + np.arange(g.shape[0]) * 0.37).astype(np.float32)
        c_, s_ = np.cos(ang), np.sin(ang)
        T = np.zeros((g.shape[0], 4, 4), np.float32)
        T[:, 0, 0] = c_ * scale; T[:, 0, 2] = s_ * scale; T[:, 1, 1] = scale; T[:, 2, 0] = -s_ * scale; T[:, 2, 2] = c_ * scale; T[:, 3, 3] = 1
        T[:, :3, 3] = g * 2.0
        return tb.make_instances(T, np.zeros(g.shape[0], np.uint32))
    W_, H_ = 3840, 2160
    nt = W_ * H_
    ext = 2.0 * side
    tcam = R.camera((-0.6 * ext, 0.8 * ext, -0.9 * ext), (0.62, -0.38, 0.68), W_, H_, 1, 1)
    d_tr = ctx.malloc(nt * 64)
    ctx.generate_primary(tcam, d_tr, 0, nt)
    tlas = tb.TLAS(ctx).Build(frame_instances(0.0), [blas])
    d_dv = ctx.malloc(dv.nbytes)
    ms_trace, ms_rebuild, ms_refit = [], [], []
    for f in range(frames):
        moved = dv.copy(); moved[:, 1] += np.float32(2e-3 * (f + 1)) * np.sin(dv[:, 0] * 3.0).astype(np.float32)
        ctx.to_device(d_dv, moved)
        blas.Refit((d_dv, dv.shape[0] // 3), on_device=True)
        rf = ctx.time_last_ms()
        blas._bounds = np.concatenate([moved[:, :3].min(0), moved[:, :3].max(0)]).astype(np.float32)   # the BLAS's new root box
        tlas._bounds_sent = False                                                                        # goes along with the transforms
        tlas.RebuildOnDevice(np.ascontiguousarray(frame_instances(float(f))["transform"]))
        rb = ctx.time_last_ms()
        tlas.intersect_device_fresh(d_tr, nt, 1e30)
        if f:
            ms_refit.append(rf); ms_rebuild.append(rb); ms_trace.append(ctx.time_last_ms())
    ctx.free(d_dv)
    rr = R.random_rays(1 << 22, (-1.0, -1.0, -1.0), (ext, ext, ext), seed=9)
    ctx.to_device(d_tr, rr)
    ms_inc = []
    for f in range(3):
        tlas.intersect_device_fresh(d_tr, rr.shape[0], 1e30)
        if f:
            ms_inc.append(ctx.time_last_ms())
    out = {"instances": side ** 3, "blas": dlabel, "blas_layout": "BVH4_GPU", "camera_rays": nt,
           "camera_mrays": nt / float(np.mean(ms_trace)) / 1e3, "trace_ms": float(np.mean(ms_trace)),
           "device_tlas_rebuild_ms": float(np.mean(ms_rebuild)), "device_blas_refit_ms": float(np.mean(ms_refit)),
           "incoherent_rays": int(rr.shape[0]), "incoherent_mrays": rr.shape[0] / float(np.mean(ms_inc)) / 1e3}
    ctx.free(d_tr); tlas.free(); blas.free()
    return out


def config2_quick(tb, ctx, R, scenes):
    """BASELINE configs[1] without the comparisons: BVH_GPU (Aila-Laine) on the Sponza stand-in, the speedtest's 1 M camera rays
    (tiny_bvh_speedtest.cpp:1092-1141), 8 launches, median of the last 6 HIP-event times."""
    verts, label = scenes.get("sponza")
    side = 1024
    n = side * side
    cam = R.camera(*scenes.SPONZA_CAMERAS[0], side, side, 1, 1)
    d = ctx.malloc(n * 64)
    ctx.generate_primary(cam, d, 0, n)
    sc = tb.BVH_GPU(ctx).Build(verts)
    for p_ in range(16):                 # untimed: the scene's 8-wide copy is made by the first query, its tuner tries its schedules on the next ones
        sc.intersect_device_fresh(d, n, 1e30); ctx.synchronize()
    for p_ in range(8):
        sc.intersect_device_fresh(d, n, 1e30)
    ctx.synchronize()
    ms = ctx.time_history(6)
    sc.free(); ctx.free(d)
    return {"scene": label, "rays": n, "layout": "BVH_GPU", "bvh_gpu_mrays": n / (float(np.median(ms)) * 1e-3) / 1e6, "launch_ms": float(np.median(ms))}
