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
they really fetch from it: the street generator at 30 M triangles (3.3 GB of tree), S / T per ray from the instrumented kernel, bytes from two
more rocprofv3 --pmc children (hbm_regime below).

One process per GPU; launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
torch is used for the barrier / max-reduce over ranks only.
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
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    use_dist = world > 1 or bool(os.environ.get("TBVH_BENCH_FORCE_DIST"))  # the env knob exercises the RCCL path on one GPU
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import tinybvh_amd as tb
    from tinybvh_amd import rays as R
    from tinybvh_amd import scenes

    def flush_c_stdio():
        # RCCL prints a banner ("Hostname", "Librccl path") through C stdio when the communicator comes up; a pipe holds
        # it back until exit, i.e. until after the JSON line.  Every rank pushes it out at the barriers instead, so that
        # rank 0's JSON line is the last line of the job's stdout.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass

    def sync_all():
        if use_dist:
            import torch
            dist.barrier()
            torch.cuda.synchronize()
            flush_c_stdio()
        ctx.synchronize()

    # ---- scene + layout (host build, untimed) ------------------------------------------------
    t0 = time.time()
    verts, label = scenes.get(a.scene)
    n_tris = verts.shape[0] // 3
    ctx = tb.Context(local_rank)
    # N ranks build the same BVH at the same time on one host: give each its share of the cores (the build is deterministic
    # whatever the thread count)
    build_threads = max(1, usable_cores() // world) if world > 1 else 0
    sc = tb.LAYOUT_CLASSES[a.layout](ctx).BuildOnDevice(verts) if a.device_build else tb.LAYOUT_CLASSES[a.layout](ctx).Build(verts, threads=build_threads)
    if a.variant:
        sc.set_variant(a.variant)
    if rank == 0:
        log(f"[bench] scene: {label}; {n_tris} tris; layout {a.layout}; host build+upload {time.time() - t0:.1f}s; device bytes {sc.device_bytes / 1e6:.0f} MB")

    # ---- ray batches on the device (untimed) -----------------------------------------------------
    n = a.side * a.side
    cams = scenes.STREET_CAMERAS if (a.scene == "bistro" or a.scene.startswith("street")) else scenes.SPONZA_CAMERAS
    eye, view = cams[0]   # the same camera on every rank: equal work per GPU, so the N-GPU aggregate measures scaling, not workload differences
    cam = R.camera(eye, view, a.side, a.side, 1, 1)
    d_verts = ctx.malloc(verts.nbytes); ctx.to_device(d_verts, verts)
    d_prim, d_diff, d_shad, d_tmp = (ctx.malloc(n * 64) for _ in range(4))
    d_occ = ctx.malloc(n)
    ext = float((verts[:, :3].max(0) - verts[:, :3].min(0)).max())
    light = (0.0, 0.9 * float(verts[:, 1].max()), 0.0)
    third = n // 3
    ctx.generate_primary(cam, d_prim, 0, n)
    sc.intersect_device(d_prim, n)
    ctx.generate_shadow(d_prim, d_shad, n, light, ext * 5e-7)
    # diffuse batch: thirds of depth 1 / 2 / 3 (wavefront.cl's 3-bounce loop, wavefront.cl:225)
    seed = 1000 * (rank + 1)
    ctx.generate_bounce(d_verts, d_prim, d_tmp, n, seed + 1)          # depth 1 for all
    # the first third stays at depth 1; the rest is traced and bounced again, in place
    sc.intersect_device(d_tmp + third * 64, n - third)
    ctx.generate_bounce(d_verts, d_tmp + third * 64, d_tmp + third * 64, n - third, seed + 2)   # depth 2
    sc.intersect_device(d_tmp + 2 * third * 64, n - 2 * third)
    ctx.generate_bounce(d_verts, d_tmp + 2 * third * 64, d_tmp + 2 * third * 64, n - 2 * third, seed + 3)  # depth 3
    d_diff, d_tmp = d_tmp, d_diff
    ctx.reset_hits(d_prim, n)
    ctx.synchronize()

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
        for kind, d in (("primary", d_prim), ("diffuse", d_diff)):
            ms = []
            for p_ in range(4):
                sc.intersect_device_fresh(d, n, 1e30)
                if p_:
                    ms.append(ctx.time_last_ms())
            out[kind + "_ms"] = float(np.mean(ms)); out[kind + "_mrays"] = n / (out[kind + "_ms"] * 1e-3) / 1e6
        if a.layout == 10:
            sc.set_variant(59)   # the instrumented strict kernel: node visits and triangle tests per ray
            for kind, d in (("primary", d_prim), ("diffuse", d_diff)):
                st = (C.c_uint64 * 8)()
                tb.lib.tbvh_debug_stats(ctx._h, st, 1)
                sc.intersect_device_fresh(d, n, 1e30)
                tb.lib.tbvh_debug_stats(ctx._h, st, 1)
                out[kind + "_S"] = int(st[2]) / n; out[kind + "_T"] = int(st[4]) / n
        print(json.dumps(out), flush=True)
        ctx.close()
        return

    kern_ms = {"primary": [], "diffuse": [], "shadow": []}

    def step():
        # "fresh" = re-arm (hit = {1e30,0,0,0}) fused into the traversal kernel: every step traces
        # every ray from scratch and writes every hit record, like a new frame would.  Nothing waits between the launches: the
        # per-launch HIP-event durations are read ONCE after the loop (tbvh_time_history), as a renderer would enqueue them.
        sc.intersect_device_fresh(d_prim, n, 1e30)
        sc.intersect_device_fresh(d_diff, n, 1e30)

    # the any-hit pass (config "16 M IsOccluded shadow rays") is reported in `detail`; it is not
    # part of the metric's step (primary + diffuse), so it is timed by HIP events only
    for i in range(a.warmup + a.steps):
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
    for _ in range(a.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
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
    if not a.no_strong:
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
    # the same batch from ONE process over K devices through the C ABI (tbvh_intersect_sharded_device): K = --one-process-devices, or every
    # visible device when this is a single-process run that sees more than one
    one_proc = None
    kdev = a.one_process_devices if a.one_process_devices else (tb.device_count() if (world == 1 and tb.device_count() > 1 and a.gpus > 1) else 0)
    if rank == 0 and world == 1 and kdev >= 2 and not a.no_strong:
        try:
            one_proc = strong_one_process(tb, R, sc, verts, eye, view, 8192 if a.side >= 4096 else 2 * a.side, kdev, log)
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
    if rank == 0 and not a.no_configs:
        try:
            cfg12 = configs_1_and_2(tb, ctx, R, scenes)
        except Exception as e:
            log(f"[bench] configs 1 / 2 failed: {e!r}")
        try:
            ref_blob = reference_blob_step(tb, ctx, verts, d_prim, d_diff, n, timed_got, par_stride, ns_par)
        except Exception as e:
            log(f"[bench] reference-blob step failed: {e!r}")

    # ---- results (rank 0) ---------------------------------------------------------------------------
    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        rays_per_step = 2 * n * world  # primary + diffuse (the metric); shadow reported in detail
        value = rays_per_step / (elapsed / a.steps) / 1e6
        mean = {k: float(np.mean(v)) for k, v in kern_ms.items()}
        detail = {k + "_mrays": n / (mean[k] * 1e-3) / 1e6 for k in mean}
        detail["kernel_ms"] = mean
        detail["primary_plus_diffuse_kernel_mrays"] = 2 * n / ((mean["primary"] + mean["diffuse"]) * 1e-3) / 1e6
        # what a step costs beyond its two queries' own HIP-event time (launch latency the stream could not hide, the barrier): the
        # round-3 driver run had 0.42 ms here, from a synchronisation after every launch and a three-launch probed query
        detail["dispatch_gap_ms"] = ms_per_step - (mean["primary"] + mean["diffuse"])
        detail["wavefront_frame_3_bounces"] = wf_detail
        detail["device_side_ops"] = dev_ops
        detail["tlas_1000_instances"] = tlas_detail
        detail["config4_strong"] = strong
        if one_proc:
            detail["config4_strong_one_process"] = one_proc
        if cfg12:
            detail["config1"] = cfg12.get("config1")
            detail["config2"] = cfg12.get("config2")
        detail["reference_blob"] = ref_blob
        if world == 1 and not a.no_hbm_regime and a.layout == 10:
            detail["hbm_regime"] = hbm_regime(a, log)

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
        # ONE headline fraction per kernel that cannot exceed 1: the larger of
        #   fabric   bytes the kernel really moved beyond the L2s per launch (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, live child run:
        #            Infinity-Cache hits included, so an upper bound on HBM bytes) / launch time, over the MEASURED streaming-read bandwidth of
        #            this GPU (tbvh_measure_read_bandwidth; the data-sheet 8 TB/s is never reached: profiles/r03_copy_rate.txt);
        #   valu     useful lane-operations per second — S x (VALU of one node visit) + T x (of one triangle test) + the per-ray part, counted
        #            in the gfx950 ISA of the shipped kernel — over the MEASURED issue ceiling for that instruction mix (tbvh_measure_valu_issue
        #            x 64 lanes; profiles/r03_valu_issue.txt).
        # The contract's algorithmic-HBM line (64 + 16 + node_bytes x S + tri_bytes x T bytes per ray over the launch time against 8 TB/s) is
        # kept as `algorithmic_hbm`: the tree lives in the L2s and the Infinity Cache, so that figure counts bytes that never reach HBM and can
        # exceed 1 — a model of the work, not of a memory system.
        try:
            nb, tbytes = {tb.LAYOUT_CWBVH: (80, 48), tb.LAYOUT_BVH4_GPU: (64, 48), tb.LAYOUT_BVH_GPU: (64, 52)}[a.layout]
            valu_node, valu_tri, valu_ray = {tb.LAYOUT_CWBVH: (235, 65, 70), tb.LAYOUT_BVH4_GPU: (150, 65, 70), tb.LAYOUT_BVH_GPU: (60, 65, 70)}[a.layout]
            copy_gbps = read_gbps = valu_ginstr = None
            try:
                copy_gbps = ctx.copy_bandwidth_gbps(1 << 30, 5)
                read_gbps = ctx.read_bandwidth_gbps(1 << 30, 5)
                valu_ginstr = ctx.valu_issue_ginstr(3)
            except Exception as e:
                log(f"[bench] ceiling measurement failed: {e!r}")
            traffic, traffic_src = live_pmc_traffic(a, log) if (world == 1 and not a.no_pmc) else (None, None)
            if traffic is None:
                pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
                if os.path.exists(pmc):
                    try:
                        j = json.load(open(pmc))
                        traffic = {"diffuse": j.get("diffuse_kernel_hbm_bytes_per_launch"), "primary": j.get("primary_kernel_hbm_bytes_per_launch")}
                        traffic_src = "profiles/pmc_traffic.json (committed rocprofv3 --pmc run of this command)"
                    except Exception:
                        traffic = None
            lines = {}
            for kind in ("diffuse", "primary"):
                if kind not in S_T:
                    continue
                S, T = S_T[kind]
                sec = mean[kind] * 1e-3
                bpr = 64 + 16 + nb * S + tbytes * T
                alg = bpr * n / sec / 1e9
                lane_ops = (S * valu_node + T * valu_tri + valu_ray) * n / sec / 1e9            # G lane-ops/s
                valu_peak = valu_ginstr * 64 if valu_ginstr else None
                tr = (traffic or {}).get(kind)
                fabric = (tr / sec / 1e9) if tr else None
                f_fabric = (fabric / read_gbps) if (fabric and read_gbps) else None
                f_valu = (lane_ops / valu_peak) if valu_peak else None
                cands = [(f, nm) for f, nm in ((f_fabric, "fabric"), (f_valu, "valu")) if f is not None]
                head = max(cands) if cands else (None, None)
                lines[kind] = {"frac": head[0], "frac_is": head[1], "avg_launch_ms": mean[kind], "nodes_per_ray": S, "tris_per_ray": T,
                               "fabric": {"achieved": fabric, "peak": read_gbps, "unit": "GB/s", "frac": f_fabric, "traffic_bytes_per_launch": tr},
                               "valu_issue": {"achieved": lane_ops, "peak": valu_peak, "unit": "G lane-ops/s", "frac": f_valu},
                               "algorithmic_hbm": {"achieved": alg, "peak": 8000.0, "unit": "GB/s", "frac": alg / 8000.0, "bytes_per_ray": bpr}}
            kname = {tb.LAYOUT_CWBVH: "k_cwbvh<false, ..., PROBED = 2> (incoherent flavor; diffuse batch)", tb.LAYOUT_BVH4_GPU: "k_bvh4_w8<false> (diffuse batch)", tb.LAYOUT_BVH_GPU: "k_bvh2<false> (diffuse batch)"}[a.layout]
            d_ = lines.get("diffuse")
            if d_:
                use_fabric = d_["frac_is"] == "fabric"
                top = d_["fabric"] if use_fabric else d_["valu_issue"]
                roof = {"bound": "hbm" if use_fabric else "valu-issue", "kernel": kname, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": d_["frac"],
                        "frac_is": ("fabric-side traffic (FETCH_SIZE x 2 + WRITE_SIZE) over the measured read bandwidth" if use_fabric else
                                    "useful VALU lane-operations over the measured issue ceiling of the node-test mix") + "; = max(fabric.frac, valu_issue.frac), <= 1 by construction",
                        "traffic": d_["fabric"]["traffic_bytes_per_launch"], "traffic_source": traffic_src,
                        "measured_copy_gbps": copy_gbps, "measured_read_gbps": read_gbps, "measured_valu_ginstr_per_s": valu_ginstr,
                        "fabric": d_["fabric"], "valu_issue": d_["valu_issue"], "algorithmic_hbm": d_["algorithmic_hbm"],
                        "nodes_per_ray": d_["nodes_per_ray"], "tris_per_ray": d_["tris_per_ray"], "avg_launch_ms": d_["avg_launch_ms"],
                        "valu_issue_model": {"lane_ops_per_node_visit": valu_node, "lane_ops_per_triangle_test": valu_tri, "lane_ops_per_ray": valu_ray,
                                             "peak": "tbvh_measure_valu_issue (k_valu_mix: 32-instruction block in the proportions of cw_test_node, 8 waves per SIMD) x 64 lanes; profiles/r03_valu_issue.txt"},
                        "limiter": "incoherent rays: the L2-miss path (lines from the Infinity Cache) with VALU issue close behind; camera rays: VALU issue (DESIGN.md §5)",
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
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{label}; BVH8_CWBVH; per GPU per step {n} primary + {n} diffuse (depth 1-3) Intersect; + {n} shadow IsOccluded timed separately",
                       "scene_tris": n_tris, "layout": {5: "BVH_GPU", 8: "BVH4_GPU", 10: "BVH8_CWBVH"}[a.layout],
                       "rays_per_gpu_per_step": 2 * n, "shadow_rays_per_gpu": n, "sharding": f"value: weak — every rank its own {2 * n}-ray step, BVH replicated, no collective; detail.config4_strong: one 64 M-ray batch in {world} contiguous shard(s)"},
            "parity_checked": bool(parity.get("n")) and "error" not in parity, "parity_ok": bool(parity.get("ok", False)),
            "detail": detail, "roofline": roof, "cpu_baseline": cpu,
        }
        flush_c_stdio()
        print(json.dumps(out), flush=True)
    sync_all()
    if use_dist:
        dist.destroy_process_group()
    ctx.close()
    if rank == 0 and not a.no_parity:
        if "error" in parity:      # the timed kernels were never checked: not a result either
            log(f"[bench] the parity check of the timed kernels could not run: {parity['error']}")
            sys.exit(4)
        if not parity.get("ok", False):
            log(f"[bench] PARITY MISMATCH on the timed kernels: {parity}")
            sys.exit(3)


def strong_one_process(tb, R, sc0, verts, eye, view, side4, k, log):
    """Config 4's batch (side4 x side4 camera rays, bounced to depths 1-3 in thirds) from ONE process over k contexts — context i on device
    i mod (visible devices) — through tbvh_intersect_sharded_device: the BVH uploaded once per context, every shard generated, traced and kept
    on its device, one host thread enqueueing all launches.  Reports the batch rate, per-device kernel ms and the host dispatch gap."""
    from tinybvh_amd.sharding import shard_range
    n_dev = tb.device_count()
    n4 = side4 * side4
    cam4 = R.camera(eye, view, side4, side4, 1, 1)
    ctxs = [tb.Context(i % n_dev) for i in range(k)]
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
    """north_star asks for ">= 50 % HBM roofline on the node-fetch loop"; the bench's own scene (184 MB of tree) lives in the L2s and the Infinity
    Cache, so that bar is measured where the kernel really fetches from HBM: the same street generator at 30 M triangles (3.3 GB of nodes and
    triangles, built on the device), 4.19 M camera rays and bounce rays (depth 1-3) per launch, the shipped kernels.  A child of this script
    times the launches and counts node visits S / triangle tests T per ray with the instrumented kernel; two more children under
    `rocprofv3 --pmc` give the bytes fetched and written per launch.  Reported per ray kind: MRays/s, algorithmic bytes per ray
    (80 S + 48 T + 64 + 16: SURVEY.md par. 8(d)), fetched bytes per ray, and both as TB/s against the 8 TB/s HBM peak."""
    import copy
    import subprocess
    b = copy.copy(a)
    b.scene, b.side, b.device_build, b.layout, b.variant = "street30m", 2048, True, 10, 0
    cmd = [sys.executable, os.path.abspath(__file__), "--hbm-child", "--device-build", "--scene", b.scene, "--side", str(b.side), "--layout", "10"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TBVH_BENCH_FORCE_DIST"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=400, check=True)
        out = json.loads([l for l in r.stdout.decode().split("\n") if l.startswith("{")][-1])
    except Exception as e:
        log(f"[bench] hbm_regime child failed: {e!r}")
        return {"error": repr(e)}
    n = out["rays_per_launch"]
    traffic, src = (live_pmc_traffic(b, log) if not a.no_pmc else (None, None))
    for kind in ("primary", "diffuse"):
        alg = 80.0 * out.get(kind + "_S", 0.0) + 48.0 * out.get(kind + "_T", 0.0) + 80.0
        sec = out[kind + "_ms"] * 1e-3
        row = {"mrays": out[kind + "_mrays"], "node_visits_per_ray": out.get(kind + "_S"), "triangle_tests_per_ray": out.get(kind + "_T"),
               "algorithmic_bytes_per_ray": alg, "algorithmic_tb_per_s": alg * n / sec / 1e12}
        if traffic:
            row["fabric_bytes_per_ray"] = traffic[kind] / n
            row["fabric_tb_per_s"] = traffic[kind] / sec / 1e12
            row["frac_of_hbm_peak"] = traffic[kind] / sec / 8e12
        out[kind] = row
    out["traffic_source"] = src
    out["peak_tb_per_s"] = 8.0
    for k in [k for k in list(out) if k.endswith(("_ms", "_mrays", "_S", "_T")) and "_" in k and k.split("_")[0] in ("primary", "diffuse")]:
        out.pop(k)
    return out


def live_pmc_traffic(a, log):
    """Fabric-side bytes per launch of the two timed kernels, measured now: this script is run again as a short child
    (--pmc-child: same scene, same batches, three (primary, diffuse) launch pairs) under `rocprofv3 --pmc FETCH_SIZE` and,
    separately, `--pmc WRITE_SIZE` (TCC counters do not fit one pass; --kernel-trace only).  FETCH_SIZE is in KB and
    tallies 64 of every 128 bytes on gfx950 (MI355X_MICROARCH.md), hence x 1024 x 2; WRITE_SIZE x 1024 taken as is.
    Infinity-Cache hits are counted, so this is an upper bound on HBM bytes.  Returns ({"primary": bytes, "diffuse": bytes}, source)
    or (None, None)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, None
    out = {"primary": 0.0, "diffuse": 0.0}
    for counter, scale in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
        d = tempfile.mkdtemp(prefix="tbvh_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--output-format", "csv", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--scene", a.scene, "--side", str(a.side), "--layout", str(a.layout),
               "--variant", str(a.variant)] + (["--device-build"] if getattr(a, "device_build", False) else [])
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TBVH_BENCH_FORCE_DIST"):
            env.pop(k, None)
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    kn = r["Kernel_Name"]
                    if r["Counter_Name"] == counter and ("k_cwbvh<false" in kn or "k_bvh4_w8<false" in kn or "k_bvh4<false" in kn or "k_bvh2<false" in kn):
                        rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            rows.sort()
            # the child makes 9 queries (3 preparing the batches, then (primary, diffuse) x 3); a probed query on a scene with the incoherent-batch
            # copies is TWO traversal dispatches (the flavor the probe's verdict is not for leaves at once): sum per query
            per_query = len(rows) // 9
            if per_query not in (1, 2) or len(rows) != 9 * per_query:
                raise RuntimeError(f"{len(rows)} traversal dispatches in the {counter} pass, expected 9 or 18")
            allv = [v for _, v in rows]
            vals = [sum(allv[i * per_query:(i + 1) * per_query]) for i in range(9)][-6:]   # (primary, diffuse) x 3; the first pair warms the caches
            out["primary"] += (vals[2] + vals[4]) / 2 * scale
            out["diffuse"] += (vals[3] + vals[5]) / 2 * scale
        except Exception as e:
            log(f"[bench] rocprofv3 --pmc {counter} child failed: {e!r}")
            return None, None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out, "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this command (FETCH_SIZE x 2, guide correction); includes Infinity-Cache hits"


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
