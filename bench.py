#!/usr/bin/env python
"""bench.py — the contract benchmark (one JSON line on rank 0).

Metric (BASELINE.json): MRays/s (primary + diffuse) on Bistro CWBVH; achieved HBM GB/s.

Workload (config.workload): BASELINE.json configs[2]+[3] on one GPU — Bistro-exterior
(real `bistro_ext_part{1,2}.bin` if present, else the labelled 2.83 M-triangle procedural
stand-in), BVH8_CWBVH layout, per GPU and per step:
    16 M primary rays   Intersect   (4096 x 4096 pinhole, speedtest tile order)
    16 M diffuse rays   Intersect   (incoherent: bounce depths 1, 2 and 3 in equal thirds)
    16 M shadow rays    IsOccluded  (from the primary hit points toward a point light)
`value` = (primary + diffuse rays of ALL ranks) / wall time of the K timed steps, where a step
runs the two Intersect passes through tbvh_intersect_device_fresh (every ray starts from
tmax = 1e30 and every hit record is written, so each step does the full work of a new frame);
the shadow pass is timed separately (HIP events) and reported in `detail`.  Rays are generated on the device before the timed region and
are resident in HBM.  N > 1: the BVH is replicated, every rank traces its own batch
(same camera, its own RNG seed for the bounce rays), no data-path collective: weak scaling.
Beside it, `detail.config4_strong` is BASELINE.json configs[3] as specified: ONE 64 M-ray diffuse batch (8192 x 8192
camera, bounce depths 1-3) cut into N contiguous wave-aligned shards (tinybvh_amd.sharding.shard_range), one per rank,
timed with the same barrier / max-over-ranks rule: strong scaling.

`roofline` carries the contract's algorithmic-HBM line for the dominant kernel (diffuse batch) and for the primary batch,
the measured device copy bandwidth as a second denominator, the fabric-side traffic measured LIVE by a rocprofv3 --pmc
child run of this same script (FETCH_SIZE and WRITE_SIZE in separate passes, MI355X_MICROARCH.md corrections), and the
VALU-issue roofline of both kernels (what actually bounds them: DESIGN.md §5).  `detail.hbm_regime` prices the same kernels against HBM where
they really fetch from it: the street generator at 1 M and at 12 M triangles (0.9 GB of tree), S / T per ray from the oracle's mirror, bytes
from two more rocprofv3 --pmc children per scene (hbm_regime below).

Beside the headline (all outside the timed region, rank 0, one-GPU runs):
    detail.rotated_scene    the SAME triangles rotated off the coordinate axes (scenes.street_rot), same camera carried along: MRays/s, S / T per ray,
                            live counters — the other end of the range real scenes lie in (review of round 4, item 1)
    detail.other_layouts    k_bvh2 / k_bvh4 on the headline batches with the reference's batch_ailalaine / batch_gpu4way (ROCm OpenCL) beside them and
                            one FETCH_SIZE + one SQ VALU-group counter child each
    detail.tlas_1000_instances   config 5; k_tlas8 / k_tlas4 beside the reference's traverse_tlas, counters of k_tlas4
    detail.host_rays        the speedtest's literal call: tbvh_intersect on a HOST tinybvh::Ray[] (stride 128) and packed (stride 64), against the
                            box's measured link rate

N GPUs, two ways: one process per GPU, launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(torch is used for the barrier / max-reduce over ranks only), or — plain `python bench.py --gpus N`, no torchrun — ONE process that drives N
devices itself: one context, BVH replica and set of batches per device, one host thread enqueueing every device's launches; `n_gpus` = N and
detail.per_gpu has N rows either way.  Fewer than N devices visible: an error line and exit code 2, never a 1-GPU number under an N-GPU flag
(TBVH_BENCH_DEVICE_MAP="0,0" maps N contexts onto listed devices: how a 1-GPU box exercises the path).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="bistro")
    ap.add_argument("--side", type=int, default=4096, help="primary rays per GPU = side^2")
    ap.add_argument("--layout", type=int, default=10, help="5 BVH_GPU, 8 BVH4_GPU, 10 BVH8_CWBVH (BVHBase::BVHType)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child runs (roofline.traffic from profiles/)")
    ap.add_argument("--no-strong", action="store_true", help="skip the 64 M-ray strong-scaling batch of config 4")
    ap.add_argument("--no-configs", action="store_true", help="skip detail.config1 / config2 / reference_blob")
    ap.add_argument("--one-process-devices", type=int, default=0, help="also trace config 4's 64 M-ray batch from THIS process over K contexts (device i mod the visible devices) through tbvh_intersect_sharded_device")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity check of the timed kernels (the line then says parity_checked: false; without this flag a check that could not run is a failure)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--device-build", action="store_true", help="build the layout on the device (tbvh_build_device: LBVH) instead of the host builder")
    ap.add_argument("--no-hbm-regime", action="store_true", help="skip detail.hbm_regime (the same kernels on a 30 M-triangle scene, beyond the Infinity Cache)")
    ap.add_argument("--hbm-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--ref-ocl", action="store_true", help=argparse.SUPPRESS)   # (scene child: also time the reference's own OpenCL kernel of the layout on the same blobs and rays)
    ap.add_argument("--tlas-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-rotated", action="store_true", help="skip detail.rotated_scene (the same triangles off the coordinate axes)")
    ap.add_argument("--no-other-layouts", action="store_true", help="skip detail.other_layouts (BVH_GPU and BVH4_GPU on the headline batches, with the reference's OpenCL kernels and counters)")
    ap.add_argument("--no-host-rays", action="store_true", help="skip detail.host_rays (tbvh_intersect on a host Ray[])")
    ap.add_argument("--blob-cache", default="", help="BVH8_CWBVH blob file (BVH8_CWBVH::Save format): read if it exists, else written after the host build (child runs of one bench share one build)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    use_dist = world > 1 or bool(os.environ.get("TBVH_BENCH_FORCE_DIST"))  # the env knob exercises the RCCL path on one GPU
    # `python bench.py --gpus N` WITHOUT torchrun (the driver's plain command line): this one process drives N devices itself — one context,
    # one BVH replica and one set of batches per device, one host thread enqueueing every device's launches (resolve_devices below)
    inproc_devices = None
    if not use_dist and "WORLD_SIZE" not in os.environ and a.gpus > 1 and not (a.pmc_child or a.hbm_child or a.tlas_child):
        import tinybvh_amd as tb_
        try:
            inproc_devices = resolve_devices(a.gpus, tb_.device_count(), os.environ.get("TBVH_BENCH_DEVICE_MAP"))
        except ValueError as e:
            print(json.dumps({"metric": "MRays/s (primary + diffuse) on Bistro CWBVH", "value": None, "unit": "MRays/s", "n_gpus": a.gpus, "error": str(e),
                              "visible_devices": tb_.device_count()}), flush=True)
            log(f"[bench] {e}")
            sys.exit(2)
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import tinybvh_amd as tb
    from tinybvh_amd import rays as R
    from tinybvh_amd import scenes

    if a.tlas_child:
        tlas_child(a, tb, R, scenes)
        return

    def flush_c_stdio():
        # RCCL prints a banner ("Hostname", "Librccl path") through C stdio when the communicator comes up; a pipe holds
        # it back until exit, i.e. until after the JSON line.  Every rank pushes it out at the barriers instead, so that
        # rank 0's JSON line is the last line of the job's stdout.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass

    others_ref = []   # (filled once the other devices of a one-process run exist)

    def sync_all():
        if use_dist:
            import torch
            dist.barrier()
            torch.cuda.synchronize()
            flush_c_stdio()
        ctx.synchronize()
        for o_ in others_ref:
            o_[0].synchronize()

    # ---- scene + layout (host build, untimed) ------------------------------------------------
    t0 = time.time()
    verts, label = scenes.get(a.scene)
    n_tris = verts.shape[0] // 3
    ctx = tb.Context(local_rank)
    # N ranks build the same BVH at the same time on one host: give each its share of the cores (the build is deterministic
    # whatever the thread count)
    build_threads = max(1, usable_cores() // world) if world > 1 else 0
    replication = None
    if use_dist and a.layout == tb.LAYOUT_CWBVH and not a.device_build:
        # N processes on one node: ONE host build (rank 0, all cores), the blobs travel as a BVH8_CWBVH::Save-compatible file, ranks 1..N-1 load
        # it — instead of N concurrent builds on the node's cores (tinybvh_amd/sharding.py; SURVEY par. 8(e): the BVH is replicated per GPU)
        from tinybvh_amd.sharding import build_once_load_everywhere
        blob_path = os.path.join("/tmp", f"tbvh_bench_{os.environ.get('MASTER_PORT', '0')}_{n_tris}.cwbvh")
        host, rep_s = build_once_load_everywhere(verts, rank, world, dist, blob_path)
        if rank == 0:
            try:
                os.remove(blob_path)
            except OSError:
                pass
        sc = tb.BVH8_CWBVH(ctx).Upload(host.blob(0, np.uint32, 4), host.blob(1, np.uint32, 4))
        sc.host = host
        replication = {"how": "rank 0 builds and writes the blob file (tbvh_cwbvh_file_write), the other ranks read it (tbvh_cwbvh_file_read)", "rank0_seconds": rep_s}
    elif a.blob_cache and a.layout == tb.LAYOUT_CWBVH and not a.device_build:
        # the blob cache of SURVEY par. 8(f4): the children of one bench run (timing + rocprofv3 --pmc passes on the same scene) share ONE host build
        if os.path.exists(a.blob_cache):
            host = tb.HostBVH.from_cwbvh_file(a.blob_cache, n_tris)
        else:
            host = tb.HostBVH(verts, tb.LAYOUT_CWBVH)
            host.save_cwbvh(a.blob_cache + ".tmp"); os.replace(a.blob_cache + ".tmp", a.blob_cache)
        sc = tb.BVH8_CWBVH(ctx).Upload(host.blob(0, np.uint32, 4), host.blob(1, np.uint32, 4))
        sc.host = host
    else:
        sc = tb.LAYOUT_CLASSES[a.layout](ctx).BuildOnDevice(verts) if a.device_build else tb.LAYOUT_CLASSES[a.layout](ctx).Build(verts, threads=build_threads)
    if a.variant:
        sc.set_variant(a.variant)
    if rank == 0:
        log(f"[bench] scene: {label}; {n_tris} tris; layout {a.layout}; host build+upload {time.time() - t0:.1f}s; device bytes {sc.device_bytes / 1e6:.0f} MB")

    # ---- ray batches on the device (untimed) -----------------------------------------------------
    n = a.side * a.side
    cams = scenes.cameras(a.scene)
    eye, view = cams[0]   # the same camera on every rank: equal work per GPU, so the N-GPU aggregate measures scaling, not workload differences
    cam = R.camera(eye, view, a.side, a.side, 1, 1)
    ext = float((verts[:, :3].max(0) - verts[:, :3].min(0)).max())
    light = (0.0, 0.9 * float(verts[:, 1].max()), 0.0)

    def make_batches(ctx_, sc_, seed_):
        """camera rays, bounce rays (thirds of depth 1 / 2 / 3: wavefront.cl's 3-bounce loop, wavefront.cl:225) and shadow rays of one device"""
        d_verts_ = ctx_.malloc(verts.nbytes); ctx_.to_device(d_verts_, verts)
        d_prim_, d_diff_, d_shad_, d_tmp_ = (ctx_.malloc(n * 64) for _ in range(4))
        third = n // 3
        ctx_.generate_primary(cam, d_prim_, 0, n)
        sc_.intersect_device(d_prim_, n)
        ctx_.generate_shadow(d_prim_, d_shad_, n, light, ext * 5e-7)
        ctx_.generate_bounce(d_verts_, d_prim_, d_tmp_, n, seed_ + 1)          # depth 1 for all
        # the first third stays at depth 1; the rest is traced and bounced again, in place
        sc_.intersect_device(d_tmp_ + third * 64, n - third)
        ctx_.generate_bounce(d_verts_, d_tmp_ + third * 64, d_tmp_ + third * 64, n - third, seed_ + 2)   # depth 2
        sc_.intersect_device(d_tmp_ + 2 * third * 64, n - 2 * third)
        ctx_.generate_bounce(d_verts_, d_tmp_ + 2 * third * 64, d_tmp_ + 2 * third * 64, n - 2 * third, seed_ + 3)  # depth 3
        d_diff_, d_tmp_ = d_tmp_, d_diff_
        ctx_.reset_hits(d_prim_, n)
        ctx_.synchronize()
        return d_verts_, d_prim_, d_diff_, d_shad_, d_tmp_

    seed = 1000 * (rank + 1)
    d_verts, d_prim, d_diff, d_shad, d_tmp = make_batches(ctx, sc, seed)
    d_occ = ctx.malloc(n)
    # the other devices of a one-process N-GPU run: (context, replica, camera batch, bounce batch) each; the blobs are uploaded from the ONE host build
    others = []
    if inproc_devices:
        for k_, dev in enumerate(inproc_devices[1:], start=1):
            c_ = tb.Context(dev)
            r_ = tb.BVH8_CWBVH(c_).Upload(sc.host.blob(0, np.uint32, 4), sc.host.blob(1, np.uint32, 4)) if a.layout == tb.LAYOUT_CWBVH else tb.LAYOUT_CLASSES[a.layout](c_).Build(verts)
            dv_, dp_, dd_, ds_, dt_ = make_batches(c_, r_, 1000 * (k_ + 1))
            for x_ in (dv_, ds_, dt_):
                c_.free(x_)
            others.append((c_, r_, dp_, dd_))
        others_ref.extend(others)
        log(f"[bench] one process, {len(inproc_devices)} contexts on devices {inproc_devices}")

    if a.pmc_child:   # under rocprofv3 --pmc: 3 preparation launches above, then (primary, diffuse) x 3; nothing else
        for _ in range(3):
            sc.intersect_device_fresh(d_prim, n, 1e30)
            sc.intersect_device_fresh(d_diff, n, 1e30)
        ctx.synchronize()
        ctx.close()
        return

    if a.hbm_child:   # detail.hbm_regime: the timed kernels on a scene beyond the Infinity Cache; one JSON line, nothing else
        import ctypes as C
        out = {"scene": label, "triangles": n_tris, "bvh_mb": sc.device_bytes / 1e6, "rays_per_launch": n, "tree": "device LBVH" if a.device_build else "host SAH"}
        if a.layout == 10 and not os.environ.get("TBVH_COHERENT_TUNER"):
            for _ in range(14):                # let the scene's tuner try its three schedules for coherent batches and settle (untimed)
                if sc.coherent_schedule(False)[0]:
                    break
                sc.intersect_device_fresh(d_prim, n, 1e30); ctx.synchronize()
            out["coherent_schedule"] = int(sc.coherent_schedule(False)[0])
        for kind, d in (("primary", d_prim), ("diffuse", d_diff)):
            ms = []
            for p_ in range(4):
                sc.intersect_device_fresh(d, n, 1e30)
                if p_:
                    ms.append(ctx.time_last_ms())
            out[kind + "_ms"] = float(np.mean(ms)); out[kind + "_mrays"] = n / (out[kind + "_ms"] * 1e-3) / 1e6
        # node visits S and triangle tests T per ray: the oracle's mirror of the layout (tiny_bvh.h:7046-7154 / 5252-5343 / 4657-4712 restated) on a
        # strided 16 k sample, over the blobs as they are on the device (the tree may have been built there)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_lib import Oracle
        orc = Oracle()
        host = getattr(sc, "host", None)
        if a.layout == 10:
            blobs = [host.blob(0, np.uint32, 4), host.blob(1, np.uint32, 4)] if host is not None else list(sc.download_blobs())
        elif a.layout == 8:
            blobs = [host.blob(0, np.uint32, 4)]
        else:
            blobs = [host.blob(0, np.uint32, 16), host.blob(1, np.uint32, 1), verts]
        for kind, d in (("primary", d_prim), ("diffuse", d_diff)):
            full = np.zeros(n, dtype=tb.RAY_DTYPE); ctx.from_device(full, d)
            sample = full[:: max(n // 16384, 1)][:16384].copy()
            sample["t"] = 1e30
            if a.layout == 10:
                _, cnt = orc.cwbvh_intersect(blobs[0], blobs[1], sample, counts=True)
            elif a.layout == 8:
                _, cnt = orc.bvh4_intersect(blobs[0], sample, counts=True)
            else:
                _, cnt = orc.bvhgpu_intersect(blobs[0], blobs[1], verts, sample, counts=True)
            out[kind + "_S"] = float(cnt[0]) / sample.shape[0]; out[kind + "_T"] = float(cnt[1]) / sample.shape[0]
            if a.ref_ocl:
                # the reference's own kernel of this layout (batch_ailalaine traverse_bvh2.cl:209-219 / batch_gpu4way traverse_bvh4.cl:277-286 /
                # batch_cwbvh traverse_cwbvh.cl:554-570) through ROCm OpenCL: same GPU, same blobs, same batch
                try:
                    from oracle_lib import ReferenceOpenCL, compare_hits
                    ocl = ReferenceOpenCL()
                    mine = full[:: max(n // 65536, 1)][:65536].copy()
                    full["t"] = 1e30; full["u"] = 0; full["v"] = 0; full["prim"] = 0
                    theirs, ref_ms = ocl.run(a.layout, blobs, full, passes=3)
                    cmp_ = compare_hits(mine, theirs[:: max(n // 65536, 1)][:65536], rtol=1e-4)   # (the .cl kernels use native_recip and strict comparisons: t to 1e-4)
                    out[kind + "_ref_opencl_mrays"] = theirs.shape[0] / (ref_ms * 1e-3) / 1e6
                    out[kind + "_ratio"] = ref_ms / out[kind + "_ms"]
                    out[kind + "_hitmiss_diff"] = int(cmp_["hitmiss"]); out[kind + "_prim_diff"] = int(cmp_["prim_mismatch"])
                    out["opencl_device"] = ocl.device
                    del theirs
                except Exception as e:
                    out["ref_opencl_error"] = repr(e)[:300]
            del full
        print(json.dumps(out), flush=True)
        ctx.close()
        return

    kern_ms = {"primary": [], "diffuse": [], "shadow": []}

    def sync_all_local():
        ctx.synchronize()
        for o_ in others:
            o_[0].synchronize()

    def step():
        # "fresh" = re-arm (hit = {1e30,0,0,0}) fused into the traversal kernel: every step traces
        # every ray from scratch and writes every hit record, like a new frame would.  Nothing waits between the launches: the
        # per-launch HIP-event durations are read ONCE after the loop (tbvh_time_history), as a renderer would enqueue them.
        sc.intersect_device_fresh(d_prim, n, 1e30)
        sc.intersect_device_fresh(d_diff, n, 1e30)
        for c_, r_, dp_, dd_ in others:     # (asynchronous launches on each context's own stream: one host thread keeps N devices busy)
            r_.intersect_device_fresh(dp_, n, 1e30)
            r_.intersect_device_fresh(dd_, n, 1e30)

    # the any-hit pass (config "16 M IsOccluded shadow rays") is reported in `detail`; it is not
    # part of the metric's step (primary + diffuse), so it is timed by HIP events only
    for i in range(max(a.warmup, 6)):       # (also lets the scene's coherent-schedule tuner settle for any-hit launches: it alternates schedules while it measures)
        sc.occluded_device(d_shad, n, d_occ)
        if i % 2:
            ctx.synchronize()
    if a.layout == tb.LAYOUT_CWBVH:
        for _ in range(12):                 # untimed: the tuner tries three schedules three times each before it decides
            if sc.coherent_schedule(True)[0]:
                break
            sc.occluded_device(d_shad, n, d_occ); ctx.synchronize()
    for i in range(a.steps):
        sc.occluded_device(d_shad, n, d_occ)
    ctx.synchronize()
    kern_ms["shadow"] = ctx.time_history(min(a.steps, 128))
    # strided sample of the shadow batch and its occlusion flags, for the parity check below (the buffers are freed before it)
    ns_par = 65536
    par_stride = max(n // ns_par, 1)
    shadow_sample = shadow_occ = None
    if rank == 0:
        full = np.zeros(n, dtype=tb.RAY_DTYPE); ctx.from_device(full, d_shad)
        shadow_sample = full[::par_stride][:ns_par].copy(); del full
        occ_all = np.zeros(n, np.uint8); ctx.from_device(occ_all, d_occ)
        shadow_occ = occ_all[::par_stride][:ns_par].copy(); del occ_all
    for i in range(max(a.warmup, 1)):
        step()
        if i % 2:
            sync_all_local()                # (finished launches are what the coherent-schedule tuner learns from; a renderer's frames end likewise)
    if a.layout == tb.LAYOUT_CWBVH:
        for _ in range(12):                 # untimed: make sure the tuner has decided before the timed region, whatever --warmup was
            if sc.coherent_schedule(False)[0]:
                break
            sc.intersect_device_fresh(d_prim, n, 1e30); ctx.synchronize()
        for c_, r_, dp_, dd_ in others:
            for _ in range(12):
                if r_.coherent_schedule(False)[0]:
                    break
                r_.intersect_device_fresh(dp_, n, 1e30); c_.synchronize()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    hist = ctx.time_history(2 * min(a.steps, 128))      # (primary, diffuse) x steps, oldest first
    kern_ms["primary"], kern_ms["diffuse"] = hist[0::2], hist[1::2]
    # the records the TIMED launches left in HBM, sampled now — before anything else traces into these buffers — for the parity checks below
    timed_got = {}
    if rank == 0:
        for kind, dptr in (("diffuse", d_diff), ("primary", d_prim)):
            full = np.zeros(n, dtype=tb.RAY_DTYPE)
            ctx.from_device(full, dptr)
            timed_got[kind] = full[::par_stride][:ns_par].copy()
            del full
    if use_dist:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- config 4 as BASELINE.json words it: ONE 64 M-ray diffuse batch, sharded over the ranks (strong scaling) ------
    strong = None
    km4_local, dm4_local, el4_local = [], [], 0.0
    n_gpus = len(inproc_devices) if inproc_devices else world
    if not a.no_strong and not inproc_devices:   # (a one-process N-GPU run shards the batch over its contexts instead: strong_one_process below)
        try:
            from tinybvh_amd.sharding import shard_range
            side4 = 8192 if a.side >= 4096 else 2 * a.side
            n4 = side4 * side4
            b4, e4 = shard_range(n4, rank, world)
            m4 = e4 - b4
            cam4 = R.camera(eye, view, side4, side4, 1, 1)
            d_a, d_b = ctx.malloc(max(m4, 1) * 64), ctx.malloc(max(m4, 1) * 64)
            if m4:
                # this rank's slice [b4, e4) of the global batch: camera rays of those pixels, bounced 1-3 times (thirds)
                ctx.generate_primary(cam4, d_a, b4, m4)
                sc.intersect_device(d_a, m4)
                t3 = m4 // 3
                ctx.generate_bounce(d_verts, d_a, d_b, m4, 4001)
                sc.intersect_device(d_b + t3 * 64, m4 - t3)
                ctx.generate_bounce(d_verts, d_b + t3 * 64, d_b + t3 * 64, m4 - t3, 4002)
                sc.intersect_device(d_b + 2 * t3 * 64, m4 - 2 * t3)
                ctx.generate_bounce(d_verts, d_b + 2 * t3 * 64, d_b + 2 * t3 * 64, m4 - 2 * t3, 4003)
                tb.intersect_sharded_device([sc], [d_b], [m4], fresh=True, tmax=1e30)      # warm-up
            sync_all()
            t0 = time.perf_counter()
            reps4 = 3
            km4, dm4 = [], []
            for _ in range(reps4):
                if m4:   # through the C ABI's device-resident multi-device entry point (this process owns one device: a 1-device call)
                    km, dm = tb.intersect_sharded_device([sc], [d_b], [m4], fresh=True, tmax=1e30)
                    km4.append(km[0]); dm4.append(dm[0])
            sync_all()
            el4 = time.perf_counter() - t0
            km4_local, dm4_local, el4_local = km4, dm4, el4 / reps4
            if use_dist:
                import torch
                t = torch.tensor([el4], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el4 = float(t.item())
            strong = {"workload": f"one {n4}-ray diffuse batch (depth 1-3), {world} contiguous wave-aligned shard(s), BVH replicated, no collective",
                      "rays": n4, "ms_per_batch": el4 / reps4 * 1e3, "mrays": n4 / (el4 / reps4) / 1e6, "scaling": "strong",
                      "rank0_shard": [b4, e4], "entry_point": "tbvh_intersect_sharded_device (one device per process)",
                      "rank0_kernel_ms": float(np.mean(km4)) if km4 else None, "rank0_host_dispatch_ms": float(np.mean(dm4)) if dm4 else None}
            ctx.free(d_a); ctx.free(d_b)
        except Exception as e:
            log(f"[bench] config 4 strong-scaling batch failed: {e!r}")
    # what every GPU did, side by side (SURVEY par. 8(e): per-GPU kernel ms and dispatch gap next to the max-over-ranks wall)
    per_gpu = None
    mine = [float(np.mean(kern_ms["primary"])), float(np.mean(kern_ms["diffuse"])), elapsed_local / a.steps * 1e3,
            float(np.mean(km4_local)) if km4_local else -1.0, float(np.mean(dm4_local)) if dm4_local else -1.0, el4_local * 1e3]
    if use_dist:
        import torch
        t = torch.tensor(mine, dtype=torch.float64, device="cuda")
        rows = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(rows, t)
        rows = [[float(x) for x in r.cpu()] for r in rows]
    else:
        rows = [mine]
        for c_, r_, dp_, dd_ in others:     # the other devices of a one-process run: their own event times of the same timed steps
            h_ = c_.time_history(2 * min(a.steps, 128))
            rows.append([float(np.mean(h_[0::2])), float(np.mean(h_[1::2])), elapsed_local / a.steps * 1e3, -1.0, -1.0, 0.0])
    per_gpu = [{"rank": i, "device": (inproc_devices[i] if inproc_devices else None), "primary_kernel_ms": r[0], "diffuse_kernel_ms": r[1], "step_wall_ms": r[2], "step_dispatch_gap_ms": r[2] - r[0] - r[1],
                "config4_shard_kernel_ms": r[3] if r[3] >= 0 else None, "config4_host_dispatch_ms": r[4] if r[4] >= 0 else None, "config4_shard_wall_ms": r[5]} for i, r in enumerate(rows)]

    # the same batch from ONE process over K devices through the C ABI (tbvh_intersect_sharded_device): K = --one-process-devices, or every
    # visible device when this is a single-process run that sees more than one
    one_proc = None
    kdev = a.one_process_devices if a.one_process_devices else (len(inproc_devices) if inproc_devices else 0)
    if rank == 0 and world == 1 and kdev >= 2 and not a.no_strong:
        try:
            one_proc = strong_one_process(tb, R, sc, verts, eye, view, 8192 if a.side >= 4096 else 2 * a.side, kdev, log, devices=inproc_devices)
        except Exception as e:
            log(f"[bench] one-process multi-device batch failed: {e!r}")

    # whole wavefront path-traced frames (Generate, {Extend, Shade} x 3, Connect; all queues on the
    # device) — config 4's pipeline end to end, reported in `detail` (outside the timed steps)
    wf_detail = None
    try:
        for p_ in (d_tmp, d_shad):
            ctx.free(p_)
        wf = tb.Wavefront(ctx, a.side, a.side)
        frames = []
        for f in range(3):
            frames.append(wf.render(sc, d_verts, cam, light, (3000.0, 3000.0, 3000.0), max_depth=3, seed=seed + f))
        st = frames[-1]
        total = sum(st["extend_rays"]) + sum(st["shadow_rays"])
        wf_detail = {"extend_rays": st["extend_rays"], "shadow_rays": st["shadow_rays"], "frame_ms": st["frame_ms"],
                     "mrays_all_stages": total / st["frame_ms"] / 1e3}
        wf.close()
    except Exception as e:
        log(f"[bench] wavefront frame failed: {e!r}")

    # the device-side maintenance operations on the same scene (outside the timed steps, rank 0 only): refit of the
    # uploaded blob to displaced vertices and a full LBVH rebuild, reported in `detail`
    dev_ops = None
    if rank == 0:
        try:
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
            dev_ops = {"refit_ms": ms_refit, "device_build_ms": ms_build, "triangles": n_tris}
        except Exception as e:
            log(f"[bench] device refit / build failed: {e!r}")

    # BASELINE config 5 next to the headline number (outside the timed steps, rank 0 only): 1000 instances of the Dragon
    # stand-in (10 x 10 x 10 grid, scale 0.07, seeded rotation), 3840 x 2160 camera rays per frame through the TLAS, the
    # TLAS rebuilt on the device from new transforms every frame; plus the same number of incoherent rays
    tlas_detail = None
    if rank == 0:
        try:
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
            for f in range(4):
                # "animated refit each frame": the BLAS vertices move a little, the BLAS is refitted on the device, then the
                # TLAS is rebuilt on the device from the frame's transforms, then the frame's rays are traced
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
            tlas_detail = {"instances": side ** 3, "blas": dlabel, "blas_layout": "BVH4_GPU", "camera_rays": nt,
                           "camera_mrays": nt / float(np.mean(ms_trace)) / 1e3, "trace_ms": float(np.mean(ms_trace)),
                           "device_tlas_rebuild_ms": float(np.mean(ms_rebuild)), "device_blas_refit_ms": float(np.mean(ms_refit)),
                           "incoherent_rays": int(rr.shape[0]), "incoherent_mrays": rr.shape[0] / float(np.mean(ms_inc)) / 1e3}
            ctx.free(d_tr); tlas.free(); blas.free()
        except Exception as e:
            log(f"[bench] TLAS configuration failed: {e!r}")

    # BASELINE configs 1 and 2 and the drop-in case, next to the headline number (outside the timed steps, rank 0 only)
    cfg12 = None
    ref_blob = None
    ref_ocl = None
    if rank == 0 and not a.no_configs:
        try:
            cfg12 = configs_1_and_2(tb, ctx, R, scenes)
        except Exception as e:
            log(f"[bench] configs 1 / 2 failed: {e!r}")
        try:
            ref_blob = reference_blob_step(tb, ctx, verts, d_prim, d_diff, n, timed_got, par_stride, ns_par)
        except Exception as e:
            log(f"[bench] reference-blob step failed: {e!r}")
        if a.layout == tb.LAYOUT_CWBVH:
            try:
                ref_ocl = reference_opencl_headline(tb, ctx, sc, d_prim, d_diff, n, kern_ms, timed_got, par_stride, ns_par)
            except Exception as e:
                log(f"[bench] reference OpenCL kernel on the headline batches failed: {e!r}")
                ref_ocl = {"error": repr(e)[:300]}

    # ---- results (rank 0) ---------------------------------------------------------------------------
    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        rays_per_step = 2 * n * n_gpus  # primary + diffuse (the metric); shadow reported in detail
        value = rays_per_step / (elapsed / a.steps) / 1e6
        mean = {k: float(np.mean(v)) for k, v in kern_ms.items()}
        detail = {k + "_mrays": n / (mean[k] * 1e-3) / 1e6 for k in mean}
        detail["kernel_ms"] = mean
        detail["primary_plus_diffuse_kernel_mrays"] = 2 * n / ((mean["primary"] + mean["diffuse"]) * 1e-3) / 1e6
        # what a step costs beyond its two queries' own HIP-event time (launch latency the stream could not hide, the barrier): the
        # round-3 driver run had 0.42 ms here, from a synchronisation after every launch and a three-launch probed query
        detail["dispatch_gap_ms"] = ms_per_step - (mean["primary"] + mean["diffuse"])
        if a.layout == tb.LAYOUT_CWBVH:
            names = {0: "still measuring", 1: "deferred triangles + gated triangle phase on 32 waves per CU", 2: "strict", 3: "one traversal per wave of 64 consecutive rays (kernels_cwbvh_packet.hip)"}
            detail["coherent_schedule"] = {kind: {"decision": names[t_[0]], "samples": [t_[1], t_[2]], "strict_over_deferred_time_per_ray": t_[3] / 1000.0}
                                           for kind, t_ in (("closest_hit", sc.coherent_schedule(False)), ("any_hit", sc.coherent_schedule(True)))}
            detail["coherent_schedule"]["how"] = "measured per scene by the library during the first launches (CohTuner, tinybvh_amd/csrc/capi_internal.h); TBVH_COHERENT_TUNER pins it"
        detail["wavefront_frame_3_bounces"] = wf_detail
        detail["device_side_ops"] = dev_ops
        detail["tlas_1000_instances"] = tlas_detail
        detail["config4_strong"] = strong if strong is not None else one_proc
        if inproc_devices:
            detail["launch"] = {"how": "one process, one context per device, one host thread enqueues every device's launches (no torchrun, no collective)", "devices": inproc_devices,
                                "visible_devices": tb.device_count(), "device_map_env": os.environ.get("TBVH_BENCH_DEVICE_MAP")}
        detail["per_gpu"] = per_gpu
        if replication:
            detail["bvh_replication"] = replication
        if one_proc:
            detail["config4_strong_one_process"] = one_proc
        if cfg12:
            detail["config1"] = cfg12.get("config1")
            detail["config2"] = cfg12.get("config2")
        detail["reference_blob"] = ref_blob
        detail["ref_opencl_cwbvh"] = ref_ocl
        if world == 1 and not a.no_hbm_regime and a.layout == 10:
            detail["hbm_regime"] = hbm_regime(a, log)
        valu_ceiling = None
        try:
            valu_ceiling = ctx.valu_issue_ginstr(3)
        except Exception as e:
            log(f"[bench] VALU ceiling measurement failed: {e!r}")
        if world == 1 and not a.no_rotated and a.layout == 10 and a.scene == "bistro":
            # the other end of the range real scenes lie in: the SAME triangles with every wall off the coordinate axes (review of round 4, item 1)
            detail["rotated_scene"] = scene_leg(a, log, "street_rot", a.side, 10, False, valu_ceiling,
                                                note="the bench scene rotated by irrational angles about two axes, same camera carried along; node visits S / triangle tests T per ray from the oracle's mirror: "
                                                     "the gap to the headline is S and T (boxes of off-axis geometry are mostly empty), for every builder incl. the reference's BuildHQ: profiles/r05_rotated.txt")
        if world == 1 and not a.no_other_layouts and a.scene == "bistro":
            detail["other_layouts"] = {name: scene_leg(a, log, a.scene, a.side, lay, True, valu_ceiling, note=f"{kern} on the headline batches next to the reference's {refk} (ROCm OpenCL, same GPU, blobs and rays)")
                                       for name, lay, kern, refk in (("BVH_GPU", 5, "k_bvh2", "batch_ailalaine (traverse_bvh2.cl:209-219)"), ("BVH4_GPU", 8, "k_bvh4", "batch_gpu4way (traverse_bvh4.cl:277-286)"))}
        if world == 1 and tlas_detail is not None and not a.no_other_layouts:
            try:
                tlas_detail.update(tlas_leg(a, log, valu_ceiling))
            except Exception as e:
                log(f"[bench] TLAS leg failed: {e!r}")
        if rank == 0 and world == 1 and not a.no_host_rays:
            try:
                detail["host_rays"] = host_rays_leg(tb, ctx, sc, d_prim, n)
            except Exception as e:
                log(f"[bench] host-rays leg failed: {e!r}")
                detail["host_rays"] = {"error": repr(e)[:300]}

        # ---- parity of the timed kernels, in this run (outside the timed region; the oracle is the checker, never the thing measured) -------
        # a strided 65 k sample of the primary and the diffuse batch: the GPU records the timed launches left in HBM against BVH::Intersect
        # restated (oracle/tbvh_oracle.c, library tie rule) on the BVH2 the layout was encoded from, and against the oracle's mirror of this
        # layout (which also counts node visits S and triangle tests T per ray for the roofline lines); the shadow batch's occlusion flags
        # against BVH::IsOccluded restated.  A real mismatch makes this process exit non-zero after the JSON line.
        roof = None
        parity = {"n": 0, "ok": False}
        S_T = {}
        try:
            if a.no_parity:
                raise RuntimeError("--no-parity")
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from oracle_lib import Oracle, compare_hits
            orc = Oracle()
            h = sc.host
            parity = {"n": ns_par, "rule": "exact prim (library tie rule: smaller prim at equal t), t / u / v bit-identical", "hitmiss": 0, "prim_real": 0, "t_bad": 0, "uv_bad": 0,
                      "tie": 0, "onsurf": 0, "not_bit_identical": 0, "shadow_flags_differ": 0}
            for kind in ("diffuse", "primary"):
                got = timed_got[kind]
                sample = got.copy()
                sample["t"] = 1e30; sample["u"] = 0; sample["v"] = 0; sample["prim"] = 0
                if a.layout == tb.LAYOUT_CWBVH:
                    mirror, cnt = orc.cwbvh_intersect(h.blob(0, np.uint32, 4), h.blob(1, np.uint32, 4), sample, counts=True)
                elif a.layout == tb.LAYOUT_BVH4_GPU:
                    mirror, cnt = orc.bvh4_intersect(h.blob(0, np.uint32, 4), sample, counts=True)
                else:
                    mirror, cnt = orc.bvhgpu_intersect(h.blob(0, np.uint32, 16), h.blob(1, np.uint32, 1), verts, sample, counts=True)
                S_T[kind] = (float(cnt[0]) / sample.shape[0], float(cnt[1]) / sample.shape[0])
                want = orc.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, sample)
                for ref_records in (want, mirror):
                    cmp_ = compare_hits(got, ref_records)
                    for k in ("hitmiss", "prim_real", "t_bad", "uv_bad", "tie", "onsurf"):
                        parity[k] += cmp_[k]
                    parity["not_bit_identical"] += cmp_["same_prim"] - cmp_["bit_identical"]
                parity[kind + "_hits"] = int((got["t"] < 1e30).sum())
            if shadow_sample is not None:
                want_occ = orc.bvh2_occluded(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, shadow_sample)
                parity["shadow_flags_differ"] = int((want_occ != shadow_occ).sum())
                parity["shadow_occluded"] = int(want_occ.sum())
            # ... and against the REAL reference under ITS OWN tie rule (oracle/_ref: BVH::Intersect of tiny_bvh.h on its own BuildHQ tree), the
            # library's two deliberate deviations counted, not tolerated away (tests/oracle_lib.py: compare_with_real_reference)
            if ref_blob and ref_blob.get("timed_launches_vs_real_reference"):
                parity["vs_real_reference"] = ref_blob.pop("timed_launches_vs_real_reference")
            parity["ok"] = (parity["hitmiss"] == 0 and parity["prim_real"] == 0 and parity["t_bad"] == 0 and parity["uv_bad"] == 0 and parity["tie"] == 0 and
                            parity["not_bit_identical"] == 0 and parity["onsurf"] <= 16 and parity["shadow_flags_differ"] <= 2)
            vr = parity.get("vs_real_reference")
            if vr and "error" not in vr:
                parity["ok"] = parity["ok"] and all(vr[k]["hitmiss"] == 0 and vr[k]["prim_real"] == 0 and vr[k]["t_bad"] == 0 and vr[k]["uv_differs"] == 0 and
                                                    vr[k]["farther_by_ulps"] == 0 for k in ("primary", "diffuse"))
        except Exception as e:
            log(f"[bench] parity sample failed: {e!r}")
            parity["error"] = repr(e)
        detail["parity_sample"] = parity

        # ---- roofline -------------------------------------------------------------------------------------------------------------------
        # Everything here comes from the guide's peaks and from counters read live in child runs of this script under `rocprofv3 --pmc`
        # (separate passes, --kernel-trace only): no hand-counted instruction constants.
        #   roofline.{achieved, peak, frac, traffic}   the dominant kernel (incoherent flavor, diffuse batch): bytes its launch moved beyond the L2s
        #            (FETCH_SIZE x 2 + WRITE_SIZE: the guide's gfx950 correction) over its mean HIP-event duration, against the guide's 8 TB/s
        #            HBM3E peak.  FETCH_SIZE counts L2 misses, INCLUDING those the Infinity Cache serves: an upper bound on HBM bytes.
        #            `frac_of_measured_read` = the same against this GPU's measured streaming-read bandwidth (tbvh_measure_read_bandwidth).
        #   roofline.infinity_cache   what separates HBM from Infinity-Cache traffic: the TCC has no counter for it (the cache sits memory-side),
        #            but the mean latency of an L2 miss does (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ); detail.hbm_regime brackets this scene
        #            between a host-SAH scene that fits the Infinity Cache and one far beyond it, same kernels, same counters.
        #   roofline.valu   SQ_INSTS_VALU per launch over the launch time against the measured issue ceiling of the node-test instruction mix
        #            (tbvh_measure_valu_issue) = how busy the issue ports are; SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU) = how many lanes
        #            of an issued instruction do work; their product = useful fraction of the VALU ceiling.
        #   roofline.algorithmic_hbm   the contract's line (64 + 16 + node_bytes x S + tri_bytes x T bytes per ray, S / T from the oracle's
        #            mirror on the parity sample, over the launch time, against 8 TB/s): counts every visit as an HBM fetch while the tree
        #            lives in the L2s and the Infinity Cache, so it can exceed 1 — a model of the work (SURVEY par. 8(d) says so itself).
        try:
            nb, tbytes = {tb.LAYOUT_CWBVH: (80, 48), tb.LAYOUT_BVH4_GPU: (64, 48), tb.LAYOUT_BVH_GPU: (64, 52)}[a.layout]
            copy_gbps = read_gbps = valu_ginstr = None
            try:
                copy_gbps = ctx.copy_bandwidth_gbps(1 << 30, 5)
                read_gbps = ctx.read_bandwidth_gbps(1 << 30, 5)
                valu_ginstr = ctx.valu_issue_ginstr(3)
            except Exception as e:
                log(f"[bench] ceiling measurement failed: {e!r}")
            # (the children run the schedule this process's tuner settled on for coherent batches, pinned: a child is too short to decide for itself)
            a.coh_pin = {1: "0", 2: "2", 3: "3"}.get(sc.coherent_schedule(False)[0] if a.layout == tb.LAYOUT_CWBVH else 0)
            pm = live_counters(a, log) if (world == 1 and not a.no_pmc) else None
            traffic_src = pm.get("source") if pm else None
            lines = {}
            for kind in ("diffuse", "primary"):
                if kind not in S_T:
                    continue
                S, T = S_T[kind]
                sec = mean[kind] * 1e-3
                bpr = 64 + 16 + nb * S + tbytes * T
                alg = bpr * n / sec / 1e9
                c = (pm or {}).get(kind, {})
                tr = (c["FETCH_SIZE"] * 2048.0 + c.get("WRITE_SIZE", 0.0) * 1024.0) if "FETCH_SIZE" in c else None
                fabric = (tr / sec / 1e9) if tr else None
                valu = None
                if "SQ_INSTS_VALU" in c:
                    rate = c["SQ_INSTS_VALU"] / sec / 1e9                        # G wave-instructions / s, whole chip
                    lane = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]) if c.get("SQ_ACTIVE_INST_VALU") else None
                    issue = rate / valu_ginstr if valu_ginstr else None
                    valu = {"insts_valu_per_launch": c["SQ_INSTS_VALU"], "insts_valu_per_ray": c["SQ_INSTS_VALU"] * 64.0 / n, "issue_rate_ginstr_per_s": rate,
                            "issue_ceiling_ginstr_per_s": valu_ginstr, "issue_frac": issue, "lane_utilisation": lane,
                            "useful_frac": (issue * lane) if (issue is not None and lane is not None) else None,
                            "source": "live SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU, SQ_ACTIVE_INST_VALU (rocprofv3 --pmc child); ceiling: tbvh_measure_valu_issue (the node-test instruction mix at 8 waves per SIMD)"}
                lat = None
                if c.get("TCC_EA0_RDREQ_sum"):
                    lat = {"mean_l2_miss_latency_cycles": c["TCC_EA0_RDREQ_LEVEL_sum"] / c["TCC_EA0_RDREQ_sum"], "l2_miss_requests_per_ray": c["TCC_EA0_RDREQ_sum"] / n}
                lines[kind] = {"avg_launch_ms": mean[kind], "nodes_per_ray": S, "tris_per_ray": T,
                               "fabric": {"achieved": fabric, "peak": 8000.0, "unit": "GB/s", "frac": fabric / 8000.0 if fabric else None,
                                          "frac_of_measured_read": (fabric / read_gbps) if (fabric and read_gbps) else None, "traffic_bytes_per_launch": tr,
                                          "bytes_per_ray": tr / n if tr else None},
                               "valu": valu, "l2_miss_latency": lat,
                               "algorithmic_hbm": {"achieved": alg, "peak": 8000.0, "unit": "GB/s", "frac": alg / 8000.0, "bytes_per_ray": bpr}}
            kname = {tb.LAYOUT_CWBVH: "k_cwbvh<false, ..., NSTRIDE = kNodeHybrid, PROBED = 2> (incoherent flavor; diffuse batch)", tb.LAYOUT_BVH4_GPU: "k_bvh4_w8<false> (diffuse batch)", tb.LAYOUT_BVH_GPU: "k_bvh2<false> (diffuse batch)"}[a.layout]
            d_ = lines.get("diffuse")
            if d_:
                f_ = d_["fabric"]
                roof = {"bound": "hbm", "kernel": kname, "achieved": f_["achieved"], "peak": 8000.0, "unit": "GB/s", "frac": f_["frac"],
                        "frac_is": "bytes the launch moved beyond the L2s (FETCH_SIZE x 2 + WRITE_SIZE, live rocprofv3 --pmc child; Infinity-Cache hits included: an upper bound on HBM bytes) / mean HIP-event launch time, over the guide's 8 TB/s HBM3E peak",
                        "frac_of_measured_read": f_["frac_of_measured_read"], "traffic": f_["traffic_bytes_per_launch"], "traffic_source": traffic_src,
                        "measured_copy_gbps": copy_gbps, "measured_read_gbps": read_gbps, "measured_valu_ginstr_per_s": valu_ginstr,
                        "fabric": f_, "valu": d_["valu"], "l2_miss_latency": d_["l2_miss_latency"], "algorithmic_hbm": d_["algorithmic_hbm"],
                        "nodes_per_ray": d_["nodes_per_ray"], "tris_per_ray": d_["tris_per_ray"], "avg_launch_ms": d_["avg_launch_ms"],
                        "limiter": "incoherent rays: no single wall - VALU issue (valu.issue_frac), the L1 lookup rate and the latency of ~9 L2 misses per ray sit within a quarter of each other; taking 10 % of the bytes, 7 % of the L1 lookups or 6 % of the instructions away moved the launch by < 1 % each, removing one DEPENDENT load (the triangle record's third, sunk behind a branch by the compiler) by 7 % (DESIGN.md par. 5 Round 4, par. 9); camera rays: roofline.primary, on the schedule the scene's tuner settled on (detail.coherent_schedule)",
                        "primary": lines.get("primary")}
        except Exception as e:  # the checker is optional for the number itself
            log(f"[bench] roofline failed: {e!r}")

        cpu = None
        if not a.no_cpu_baseline:
            try:
                cpu = cpu_baseline(tb, ctx, verts, d_prim, d_diff, n)
            except Exception as e:
                log(f"[bench] cpu baseline failed: {e!r}")

        out = {
            "metric": "MRays/s (primary + diffuse) on Bistro CWBVH", "value": value, "unit": "MRays/s",
            "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{label}; BVH8_CWBVH; per GPU per step {n} primary + {n} diffuse (depth 1-3) Intersect; + {n} shadow IsOccluded timed separately",
                       "scene_tris": n_tris, "layout": {5: "BVH_GPU", 8: "BVH4_GPU", 10: "BVH8_CWBVH"}[a.layout],
                       "rays_per_gpu_per_step": 2 * n, "shadow_rays_per_gpu": n, "sharding": f"value: weak — every GPU its own {2 * n}-ray step, BVH replicated, no collective; detail.config4_strong: one 64 M-ray batch in {n_gpus} contiguous shard(s)"},
            "parity_checked": bool(parity.get("n")) and "error" not in parity, "parity_ok": bool(parity.get("ok", False)),
            "detail": detail, "roofline": roof, "cpu_baseline": cpu,
        }
        flush_c_stdio()
        print(json.dumps(out), flush=True)
    sync_all()
    if use_dist:
        dist.destroy_process_group()
    for c_, r_, dp_, dd_ in others:
        c_.close()
    ctx.close()
    if rank == 0 and not a.no_parity:
        if "error" in parity:      # the timed kernels were never checked: not a result either
            log(f"[bench] the parity check of the timed kernels could not run: {parity['error']}")
            sys.exit(4)
        if not parity.get("ok", False):
            log(f"[bench] PARITY MISMATCH on the timed kernels: {parity}")
            sys.exit(3)


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
    cores = usable_cores()
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
    for p_ in range(8):
        sc.intersect_device_fresh(d, n, 1e30)
        t = ctx.time_last_ms()
        if p_ >= 2:
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


def reference_blob_step(tb, ctx, verts, d_prim, d_diff, n, timed_got, par_stride, ns_par):
    """The drop-in case in the driver's own run: the SAME timed step (primary + diffuse, fresh) on blobs encoded by the real tiny_bvh.h —
    BVH8_CWBVH::BuildHQ through oracle/_ref (tiny_bvh_speedtest.cpp:1196-1204) — uploaded verbatim through tbvh_upload_cwbvh; and the
    headline's parity against the REAL reference: a 65 k strided sample of the primary and of the diffuse batch traced by the real
    BVH::Intersect (tiny_bvh.h:3222-3304, its own BuildHQ tree, its own tie rule), compared with (a) the records the TIMED launches
    left (the library's own tree) and (b) the records the GPU produces on the reference-built CWBVH blob."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Reference, compare_with_real_reference, have_reference
    if not have_reference():
        return {"kind": "n/a", "why": "oracle/_ref/libtinybvh_ref.so did not travel with the repo"}
    ref = Reference()
    t0 = time.time()
    rs = ref.build(verts, hq=True, threaded=True)
    nodes, tris = rs.blob(10, 0, np.uint32, 4), rs.blob(10, 1, np.uint32, 4)
    build_s = time.time() - t0
    sc = tb.BVH8_CWBVH(ctx).Upload(nodes, tris)
    ms = {"primary": [], "diffuse": []}
    for p_ in range(5):
        for kind, d in (("primary", d_prim), ("diffuse", d_diff)):
            sc.intersect_device_fresh(d, n, 1e30)
    ctx.synchronize()
    hist = ctx.time_history(10)
    ms["primary"], ms["diffuse"] = hist[4::2], hist[5::2]
    mp, md = float(np.median(ms["primary"])), float(np.median(ms["diffuse"]))
    out = {"kind": "reference", "builder": "tinybvh BVH8_CWBVH::BuildHQ (oracle/_ref), blobs uploaded verbatim", "host_build_s": build_s, "node_blocks": int(nodes.shape[0]), "tri_blocks": int(tris.shape[0]),
           "primary_mrays": n / (mp * 1e-3) / 1e6, "diffuse_mrays": n / (md * 1e-3) / 1e6, "primary_plus_diffuse_mrays": 2 * n / ((mp + md) * 1e-3) / 1e6}
    try:
        vs_blob, vs_timed = {}, {}
        for kind, d in (("primary", d_prim), ("diffuse", d_diff)):
            full = np.zeros(n, dtype=tb.RAY_DTYPE)
            ctx.from_device(full, d)
            got_blob = full[::par_stride][:ns_par].copy()
            del full
            sample = got_blob.copy()
            sample["t"] = 1e30; sample["u"] = 0; sample["v"] = 0; sample["prim"] = 0
            want = rs.intersect(1, sample)          # the real BVH::Intersect
            vs_blob[kind] = compare_with_real_reference(got_blob, want)
            if timed_got and kind in timed_got:
                vs_timed[kind] = compare_with_real_reference(timed_got[kind], want)
        rule = "real tinybvh BVH::Intersect on its BuildHQ tree, reference tie rule; classes: tests/oracle_lib.py compare_with_real_reference"
        out["vs_real_reference"] = dict(vs_blob, rule=rule, rays_sampled=2 * ns_par)
        if vs_timed:
            out["timed_launches_vs_real_reference"] = dict(vs_timed, rule=rule, rays_sampled=2 * ns_par,
                                                           differ_from_reference=sum(v["differ_from_reference"] for v in vs_timed.values()))
    except Exception as e:
        out["vs_real_reference"] = {"error": repr(e)[:300]}
    sc.free()
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


def resolve_devices(gpus, visible, device_map=None):
    """The HIP devices a ONE-PROCESS `--gpus N` run drives (no torchrun): devices 0 .. N-1, or — TBVH_BENCH_DEVICE_MAP="0,0,1,..." — the N listed
    ones (several contexts may share a device: how a 1-GPU box exercises the N-context path).  Raises ValueError when fewer than N devices are
    visible and no map covers for it: a 1-GPU number must never be reported under an N-GPU flag."""
    if gpus < 1:
        raise ValueError(f"--gpus {gpus}: at least one GPU")
    if device_map:
        try:
            devs = [int(x) for x in device_map.split(",") if x.strip() != ""]
        except ValueError:
            raise ValueError(f"TBVH_BENCH_DEVICE_MAP={device_map!r}: a comma-separated list of device indices")
        if len(devs) != gpus:
            raise ValueError(f"TBVH_BENCH_DEVICE_MAP lists {len(devs)} devices for --gpus {gpus}")
        bad = [d for d in devs if d < 0 or d >= visible]
        if bad:
            raise ValueError(f"TBVH_BENCH_DEVICE_MAP names device(s) {bad}, {visible} visible")
        return devs
    if visible < gpus:
        raise ValueError(f"--gpus {gpus} but {visible} HIP device(s) visible (launch through torch.distributed.run, or map contexts onto devices with TBVH_BENCH_DEVICE_MAP)")
    return list(range(gpus))


def usable_cores():
    """Host threads this process can really run at once: the affinity mask, cut by the cgroup CPU quota if there is one
    (os.cpu_count() reports the whole machine even inside a container limited to a few cores)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


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
            cmd = [sys.executable, os.path.abspath(__file__), "--hbm-child", "--scene", b.scene, "--side", str(b.side), "--layout", "10", "--blob-cache", b.blob_cache]
            try:
                r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=400, check=True)
                out = json.loads([l for l in r.stdout.decode().split("\n") if l.startswith("{")][-1])
            except Exception as e:
                log(f"[bench] hbm_regime child ({scene}) failed: {e!r}")
                res["scenes"][tag] = {"error": repr(e)}
                continue
            n = out["rays_per_launch"]
            b.coh_pin = {1: "0", 2: "2", 3: "3"}.get(out.get("coherent_schedule"), "0")
            pm = live_counters(b, log, passes=("FETCH_SIZE", "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum")) if not a.no_pmc else None
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


def scene_leg(a, log, scene, side, layout, ref_ocl, valu_ceiling, note=""):
    """One more (scene, layout) measured like the headline: a child of this script times `side`^2 camera and bounce rays (HIP events) and counts
    S / T with the oracle's mirror (and, ref_ocl, times the reference's own OpenCL kernel of the layout on the same blobs and rays); two more
    children under `rocprofv3 --pmc` give the bytes beyond the L2s and the VALU counters of the same launches."""
    import copy
    import subprocess
    import tempfile
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TBVH_BENCH_FORCE_DIST", "TBVH_COHERENT_TUNER"):
        env.pop(k, None)
    tmpdir = tempfile.mkdtemp(prefix="tbvh_leg_", dir="/tmp")
    b = copy.copy(a)
    b.scene, b.side, b.device_build, b.layout, b.variant, b.coh_pin = scene, side, False, layout, 0, None
    b.blob_cache = os.path.join(tmpdir, scene + ".cwbvh") if layout == 10 else ""
    cmd = [sys.executable, os.path.abspath(__file__), "--hbm-child", "--scene", scene, "--side", str(side), "--layout", str(layout)] + \
          (["--blob-cache", b.blob_cache] if b.blob_cache else []) + (["--ref-ocl"] if ref_ocl else [])
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, check=True)
        out = json.loads([l for l in r.stdout.decode().split("\n") if l.startswith("{")][-1])
        n = out["rays_per_launch"]
        b.coh_pin = {1: "0", 2: "2", 3: "3"}.get(out.get("coherent_schedule"), "0")      # the counter children run the schedule the timing child's tuner chose
        pm = live_counters(b, log, passes=("FETCH_SIZE", "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES")) if not a.no_pmc else None
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
    the launches; otherwise also the TLAS over BVH8_CWBVH BLASes (k_tlas8) next to the reference's traverse_tlas (traverse_tlas.cl:13-107, through
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
        out.update({"k_tlas8_ms": float(np.mean(ms8)), "k_tlas8_mrays": nt / float(np.mean(ms8)) / 1e3, "ref_opencl_traverse_tlas_ms": ref_ms,
                    "ref_opencl_traverse_tlas_mrays": ref.shape[0] / ref_ms / 1e3, "ratio_k_tlas8_cwbvh_blas": ref_ms / float(np.mean(ms8)),
                    "ratio_k_tlas4_bvh4_blas": ref_ms / float(np.mean(ms)), "hitmiss_diff": int((mh != rh).sum()), "opencl_device": ocl.device,
                    "ref_kernel": "traverse_tlas (traverse_tlas.cl:13-107) via wavefront2.cl Extend, BVH8_CWBVH BLAS (the configuration of tiny_bvh_gpu2.cpp), same TLAS / instances / rays"})
    except Exception as e:
        out["ref_opencl_error"] = repr(e)[:300]
    print(json.dumps(out), flush=True)
    ctx.close()


def pmc_dispatches(child_args, counters, match, env_extra=None, timeout=300):
    """Runs `python bench.py <child_args>` under `rocprofv3 --pmc <counters> --kernel-trace` (counters in their own run, as the pool requires) and
    returns, in launch order, {counter: value summed over the dispatch's rows} for every dispatch whose kernel name satisfies `match`."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    d = tempfile.mkdtemp(prefix="tbvh_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", **(env_extra or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TBVH_BENCH_FORCE_DIST"):
        env.pop(k, None)
    try:
        cmd = ["rocprofv3", "--output-format", "csv", "--pmc"] + counters + ["--kernel-trace", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__)] + child_args
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
        per = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] in counters and match(r["Kernel_Name"]):
                    row = per.setdefault(int(r["Dispatch_Id"]), {})
                    row[r["Counter_Name"]] = row.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        return [per[k] for k in sorted(per)]
    finally:
        shutil.rmtree(d, ignore_errors=True)


def tlas_leg(a, log, valu_ceiling):
    """detail.tlas_1000_instances: the reference's traverse_tlas beside k_tlas8 / k_tlas4, and k_tlas4's counters (FETCH_SIZE; the SQ VALU group)."""
    import shutil
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TBVH_BENCH_FORCE_DIST"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--tlas-child"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, check=True)
    out = json.loads([l for l in r.stdout.decode().split("\n") if l.startswith("{")][-1])
    res = {"vs_reference_opencl": out}
    if not a.no_pmc and shutil.which("rocprofv3"):
        cnt = {}
        for pass_ in ("FETCH_SIZE", "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES"):
            try:
                rows = pmc_dispatches(["--tlas-child", "--pmc-child"], pass_.split(), lambda kn: "k_tlas4" in kn)
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


def group_dispatches_into_queries(ids, names):
    """Traversal dispatches (ids in launch order, names[id] = kernel name) -> queries.  A probed query on a scene with the incoherent-batch copies is
    TWO dispatches back to back: the first kernel — the coherent flavor, `k_cwbvh<..., NSTRIDE 5, PROBED 3, ...>`, or its strict form `PROBED 4` while
    the scene's coherent-schedule tuner is still measuring — then the incoherent flavor (`NSTRIDE 13 = kNodeHybrid, PROBED 2`); every other query
    is one dispatch."""
    queries, i = [], 0
    while i < len(ids):
        pair = i + 1 < len(ids) and (", 5, 3, " in names[ids[i]] or ", 5, 4, " in names[ids[i]] or "k_cwbvh_packet<" in names[ids[i]]) and ", 13, 2, " in names[ids[i + 1]]
        queries.append(ids[i:i + 2] if pair else ids[i:i + 1])
        i += 2 if pair else 1
    return queries


def live_counters(a, log, passes=("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES", "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum")):
    """Hardware counters of the two timed kernels, measured now: this script is run again as a short child (--pmc-child: same scene, same
    batches, three (primary, diffuse) launch pairs) under `rocprofv3 --pmc <pass>` once per pass (TCC counters do not fit one pass; --kernel-trace
    only, as the pool requires).  Returns {"primary": {counter: per-launch value}, "diffuse": {...}, "source": ...}, the mean of the last two
    pairs (the first warms the caches), summed over the dispatches of one query (a probed query is two traversal dispatches).
    FETCH_SIZE is in KB and tallies 64 of every 128 bytes on gfx950 (MI355X_MICROARCH.md): callers multiply by 2048; WRITE_SIZE x 1024.
    A pass that fails (a counter this box does not have) is skipped with a note; None if nothing could be collected."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    out = {"primary": {}, "diffuse": {}}
    got_any = False
    for pass_ in passes:
        counters = pass_.split()
        d = tempfile.mkdtemp(prefix="tbvh_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--output-format", "csv", "--pmc"] + counters + ["--kernel-trace", "-d", d, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--scene", a.scene, "--side", str(a.side), "--layout", str(a.layout),
               "--variant", str(a.variant)] + (["--device-build"] if getattr(a, "device_build", False) else []) + \
              (["--blob-cache", a.blob_cache] if getattr(a, "blob_cache", "") else [])
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TBVH_BENCH_FORCE_DIST"):
            env.pop(k, None)
        if getattr(a, "coh_pin", None) is not None:
            env["TBVH_COHERENT_TUNER"] = a.coh_pin
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=200, check=True)
            per_disp, names = {}, {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    kn = r["Kernel_Name"]
                    if r["Counter_Name"] in counters and ("k_cwbvh<false" in kn or "k_cwbvh_packet<false" in kn or "k_bvh4_w8<false" in kn or "k_bvh4<false" in kn or "k_bvh2<false" in kn):
                        row = per_disp.setdefault(int(r["Dispatch_Id"]), {})
                        row[r["Counter_Name"]] = row.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                        names[int(r["Dispatch_Id"])] = kn
            ids = sorted(per_disp)
            # the child makes 9 queries (3 preparing the batches — smaller ones among them —, then (primary, diffuse) x 3); a probed query on a scene
            # with the incoherent-batch copies is TWO traversal dispatches back to back — the coherent flavor (PROBED = 3), then the incoherent one
            # (NSTRIDE = kNodeHybrid = 13, PROBED = 2); the one the probe's verdict is not for leaves at once —: group them into queries (PROBED = 4: the
            # strict form of the first kernel, while the scene's coherent-schedule tuner is measuring)
            queries = group_dispatches_into_queries(ids, names)
            if len(queries) != 9:
                raise RuntimeError(f"{len(ids)} traversal dispatches in {len(queries)} queries in the {pass_!r} pass, expected 9 queries")
            for cn in counters:
                vals = [sum(per_disp[j].get(cn, 0.0) for j in q) for q in queries][-6:]
                out["primary"][cn] = (vals[2] + vals[4]) / 2
                out["diffuse"][cn] = (vals[3] + vals[5]) / 2
            got_any = True
        except Exception as e:
            log(f"[bench] rocprofv3 --pmc {pass_!r} child failed: {e!r}")
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if not got_any:
        return None
    out["source"] = "live: rocprofv3 --pmc child runs of this command, one per counter group (" + "; ".join(passes) + "); FETCH_SIZE x 2 (guide correction for gfx950), Infinity-Cache hits included"
    return out


def cpu_baseline(tb, ctx, verts, d_prim, d_diff, n):
    """Reference BVH8_CPU (AVX2) on all host cores over a bounded sample of the same rays (oracle/_ref, kind 'reference'),
    plus — SURVEY.md §8(d) — the same on ONE thread and BVH::Intersect (the oracle's own traversal) on all cores and on
    one; falls back to the single-threaded C restatement ('port') where oracle/_ref is absent."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle, Reference, have_reference
    ns = min(1 << 23, n)  # 8 M primary + 8 M diffuse: a bounded sample, seconds of CPU work
    buf = np.zeros(ns, dtype=tb.RAY_DTYPE)
    batches = []
    for d in (d_prim, d_diff):
        ctx.from_device(buf, d + ((n - ns) // 2) * 64)
        b = buf.copy(); b["t"] = 1e30
        batches.append(b)
    cores = usable_cores()
    if have_reference():
        ref = Reference()
        t0 = time.time()
        rs = ref.build(verts, hq=False, threaded=True)
        rs.time_mt(11, batches[0][:1024], threads=1)  # builds BVH8_CPU
        log(f"[bench] reference BVH + BVH8_CPU build {time.time() - t0:.1f}s")
        sec = sum(rs.time_mt(11, b, threads=cores)[0] for b in batches)

        def rate(layout, threads, k):   # k rays of each batch, strided over the sample
            sub = [np.ascontiguousarray(b[:: max(ns // k, 1)][:k]) for b in batches]
            return sum(x.shape[0] for x in sub) / sum(rs.time_mt(layout, x, threads=threads)[0] for x in sub) / 1e6, sub[0].shape[0]
        r8_1, k8 = rate(11, 1, 1 << 20)
        r1_mt, k1m = rate(1, cores, 1 << 21)
        r1_1, k11 = rate(1, 1, 1 << 18)
        return {"value": 2 * ns / sec / 1e6, "unit": "MRays/s", "cores": cores, "kind": "reference",
                "sample": f"tinybvh BVH8_CPU::Intersect (AVX2), {cores} threads, {ns} primary + {ns} diffuse rays of the GPU batches",
                "threads_1": {"value": r8_1, "unit": "MRays/s", "cores": 1, "sample": f"BVH8_CPU::Intersect, 1 thread, {k8} + {k8} rays"},
                "bvh_intersect": {"value": r1_mt, "unit": "MRays/s", "cores": cores, "sample": f"BVH::Intersect (the parity oracle's traversal), {cores} threads, {k1m} + {k1m} rays",
                                  "threads_1": {"value": r1_1, "unit": "MRays/s", "cores": 1, "sample": f"BVH::Intersect, 1 thread, {k11} + {k11} rays"}}}
    orc = Oracle()
    h = tb.HostBVH(verts, tb.LAYOUT_BVH2_WALD)
    ns2 = 100_000
    t0 = time.time()
    for b in batches:
        orc.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, b[:ns2])
    sec = time.time() - t0
    return {"value": 2 * ns2 / sec / 1e6, "unit": "MRays/s", "cores": 1, "kind": "port",
            "sample": f"C restatement of BVH::Intersect, 1 thread, {ns2} primary + {ns2} diffuse rays"}


if __name__ == "__main__":
    main()
