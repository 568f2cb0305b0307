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
    sc = tb.LAYOUT_CLASSES[a.layout](ctx).Build(verts, threads=build_threads)
    if a.variant:
        sc.set_variant(a.variant)
    if rank == 0:
        log(f"[bench] scene: {label}; {n_tris} tris; layout {a.layout}; host build+upload {time.time() - t0:.1f}s; device bytes {sc.device_bytes / 1e6:.0f} MB")

    # ---- ray batches on the device (untimed) -----------------------------------------------------
    n = a.side * a.side
    cams = scenes.STREET_CAMERAS if a.scene == "bistro" else scenes.SPONZA_CAMERAS
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

    kern_ms = {"primary": [], "diffuse": [], "shadow": []}

    def step(record: bool):
        # "fresh" = re-arm (hit = {1e30,0,0,0}) fused into the traversal kernel: every step traces
        # every ray from scratch and writes every hit record, like a new frame would
        sc.intersect_device_fresh(d_prim, n, 1e30)
        if record:
            kern_ms["primary"].append(ctx.time_last_ms())
        sc.intersect_device_fresh(d_diff, n, 1e30)
        if record:
            kern_ms["diffuse"].append(ctx.time_last_ms())

    # the any-hit pass (config "16 M IsOccluded shadow rays") is reported in `detail`; it is not
    # part of the metric's step (primary + diffuse), so it is timed by HIP events only
    for i in range(a.warmup + a.steps):
        sc.occluded_device(d_shad, n, d_occ)
        if i >= a.warmup:
            kern_ms["shadow"].append(ctx.time_last_ms())
    for _ in range(a.warmup):
        step(False)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

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

    # ---- results (rank 0) ---------------------------------------------------------------------------
    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        rays_per_step = 2 * n * world  # primary + diffuse (the metric); shadow reported in detail
        value = rays_per_step / (elapsed / a.steps) / 1e6
        mean = {k: float(np.mean(v)) for k, v in kern_ms.items()}
        detail = {k + "_mrays": n / (mean[k] * 1e-3) / 1e6 for k in mean}
        detail["kernel_ms"] = mean
        detail["primary_plus_diffuse_kernel_mrays"] = 2 * n / ((mean["primary"] + mean["diffuse"]) * 1e-3) / 1e6
        detail["wavefront_frame_3_bounces"] = wf_detail
        detail["device_side_ops"] = dev_ops
        detail["tlas_1000_instances"] = tlas_detail

        # roofline of the dominant kernel (CWBVH Intersect on the diffuse batch): algorithmic
        # bytes per ray = 64 (ray in) + 16 (hit out) + node_bytes*S + tri_bytes*T, SURVEY.md
        # §8(d), with S,T counted by the oracle's mirror of this layout on a sample of the same rays.
        roof = None
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from oracle_lib import Oracle
            orc = Oracle()
            ns = 65536
            sample = np.zeros(ns, dtype=tb.RAY_DTYPE)
            stride = max(n // ns, 1)
            # strided sample of the diffuse batch, re-armed
            full = np.zeros(n, dtype=tb.RAY_DTYPE) if n * 64 <= (2 << 30) else None
            if full is not None:
                ctx.from_device(full, d_diff)
                sample = full[::stride][:ns].copy()
                del full
            sample["t"] = 1e30
            h = sc.host
            if a.layout == tb.LAYOUT_CWBVH:
                _, cnt = orc.cwbvh_intersect(h.blob(0, np.uint32, 4), h.blob(1, np.uint32, 4), sample, counts=True)
                nb, tbytes = 80, 48
            elif a.layout == tb.LAYOUT_BVH4_GPU:
                _, cnt = orc.bvh4_intersect(h.blob(0, np.uint32, 4), sample, counts=True)
                nb, tbytes = 64, 48
            else:
                _, cnt = orc.bvhgpu_intersect(h.blob(0, np.uint32, 16), h.blob(1, np.uint32, 1), verts, sample, counts=True)
                nb, tbytes = 64, 52
            S, T = float(cnt[0]) / sample.shape[0], float(cnt[1]) / sample.shape[0]
            bytes_per_ray = 64 + 16 + nb * S + tbytes * T
            achieved = bytes_per_ray * n / (mean["diffuse"] * 1e-3) / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get("diffuse_kernel_hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "kernel": "k_cwbvh<false> (diffuse batch)" if a.layout == tb.LAYOUT_CWBVH else "intersect (diffuse batch)",
                    "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                    "traffic": traffic, "algorithmic_bytes_per_ray": bytes_per_ray, "nodes_per_ray": S, "tris_per_ray": T,
                    "avg_launch_ms": mean["diffuse"]}
        except Exception as e:  # the checker is optional for the number itself
            log(f"[bench] roofline sample failed: {e!r}")

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
                       "rays_per_gpu_per_step": 2 * n, "shadow_rays_per_gpu": n, "sharding": f"rays x{world}, BVH replicated, no collective"},
            "detail": detail, "roofline": roof, "cpu_baseline": cpu,
        }
        flush_c_stdio()
        print(json.dumps(out), flush=True)
    sync_all()
    if use_dist:
        dist.destroy_process_group()
    ctx.close()


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


def cpu_baseline(tb, ctx, verts, d_prim, d_diff, n):
    """Reference BVH8_CPU (AVX2) on all host cores over a bounded sample of the same rays
    (oracle/_ref, kind 'reference'); falls back to the single-threaded C oracle ('port')."""
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
        rs.time_mt(8, batches[0][:1024], threads=1)  # builds BVH8_CPU
        log(f"[bench] reference BVH + BVH8_CPU build {time.time() - t0:.1f}s")
        sec = sum(rs.time_mt(8, b, threads=cores)[0] for b in batches)
        return {"value": 2 * ns / sec / 1e6, "unit": "MRays/s", "cores": cores, "kind": "reference",
                "sample": f"tinybvh BVH8_CPU::Intersect (AVX2), {cores} threads, {ns} primary + {ns} diffuse rays of the GPU batches"}
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
