#!/usr/bin/env python
"""bench.py — the contract benchmark: ONE compact JSON line (< 4 KB) as the LAST line of rank 0's stdout.

Metric (BASELINE.json): MRays/s (primary + diffuse) on Bistro CWBVH; achieved HBM GB/s.

Workload (config.workload): BASELINE.json configs[2]+[3] on one GPU — Bistro-exterior (real `bistro_ext_part{1,2}.bin` if present under
TBVH_SCENE_DIR, else the labelled 2.83 M-triangle procedural stand-in), BVH8_CWBVH layout, per GPU and per step:
    16 M primary rays   Intersect   (4096 x 4096 pinhole, speedtest tile order)
    16 M diffuse rays   Intersect   (incoherent: bounce depths 1, 2 and 3 in equal thirds)
`value` = (primary + diffuse rays of ALL ranks) / wall time of the K timed steps (barrier + synchronize on both sides, MAX over ranks); a
step runs the two passes through tbvh_intersect_device_fresh (every ray starts from tmax = 1e30 and every hit record is written: the full
work of a new frame).  Rays are generated on the device before the timed region and are resident in HBM.  16 M shadow rays (IsOccluded) are
timed separately by HIP events (`shadow_mrays`).  N > 1: the BVH is replicated, every rank traces its own batch, no data-path collective
(weak scaling); `config4_strong_mrays` is configs[3] as worded: ONE 64 M-ray diffuse batch cut into N contiguous wave-aligned shards.

The line carries (keys fixed, tests/test_bench_line.py):
    value, ms_per_step, config{workload, scene_tris, layout, rays_per_gpu_per_step}
    roofline      dominant kernel (diffuse batch): `achieved` = ALGORITHMIC bytes per launch (SURVEY par. 8(d): 64 + 16 + 80 S + 48 T per
                  ray, S / T counted by the oracle's mirror on the parity sample) / mean HIP-event launch time; `traffic` = bytes beyond the
                  L2s per launch from live `rocprofv3 --pmc` children (FETCH_SIZE x 2 + WRITE_SIZE, the guide's gfx950 correction) and
                  `traffic_frac` = that / launch time / 8 TB/s; VALU issue and lane utilisation from the SQ counters; `primary` = the same, short
    cpu_baseline  the reference's BVH8_CPU::Intersect (oracle/_ref) on the host cores over a bounded sample of the same rays
    parity        the records the TIMED launches left, a 65 k strided sample per batch, against the oracle and the REAL BVH::Intersect
    reference_blob  the same timed step on blobs encoded by the real tinybvh BVH8_CWBVH::BuildHQ, uploaded verbatim (the drop-in case),
                  coherent-schedule tuner settled first, as many steps as the headline
Everything else (per-GPU rows, tuner state, counters in full, the legs of `--detail`) goes to the sidecar file named by `detail_file`.

`--detail` adds the side legs of bench_detail.py (configs 1 / 2 / 5, reference OpenCL kernels on the same GPU, HBM-regime scenes, rotated
scene, other layouts, host rays, wavefront frames, device refit / build); `--ceilings` re-measures the box's ceilings (streaming read / copy
bandwidth, VALU issue rate of the node-test mix) instead of reading profiles/ceilings.json.

N GPUs, two ways: one process per GPU, launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(torch is used for the barrier / max-reduce over ranks only), or — plain `python bench.py --gpus N`, no torchrun — ONE process that drives N
devices itself.  Fewer than N devices visible: an error line and exit code 2, never a 1-GPU number under an N-GPU flag
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

METRIC = "MRays/s (primary + diffuse) on Bistro CWBVH"
LINE_LIMIT = 4096                      # bytes: the driver's reader gave up on round 5's 21 KB line
LAYOUT_NAMES = {5: "BVH_GPU", 8: "BVH4_GPU", 10: "BVH8_CWBVH"}
LAYOUT_BYTES = {10: (80, 48), 8: (64, 48), 5: (64, 52)}   # node bytes, bytes per triangle test (SURVEY par. 8(d))
SCHEDULE_NAMES = {0: "undecided", 1: "deferred+gated", 2: "strict", 3: "wave packet"}
CEILINGS_FILE = os.path.join(ROOT, "profiles", "ceilings.json")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class Legs:
    """wall seconds per leg of the run (stderr + sidecar): what the driver's clock is spent on"""
    def __init__(self):
        self.t, self.rows = time.time(), []

    def mark(self, name):
        now = time.time()
        self.rows.append((name, round(now - self.t, 2)))
        log(f"[bench] leg {name}: {now - self.t:.1f} s")
        self.t = now


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="bistro")
    ap.add_argument("--side", type=int, default=4096, help="primary rays per GPU = side^2")
    ap.add_argument("--layout", type=int, default=10, help="5 BVH_GPU, 8 BVH4_GPU, 10 BVH8_CWBVH (BVHBase::BVHType)")
    ap.add_argument("--detail", action="store_true", help="also run the side legs of bench_detail.py (sidecar file only; the line stays compact)")
    ap.add_argument("--detail-out", default="", help="sidecar path (default: gpurun_out/bench_detail.json if gpurun_out/ exists, else ./bench_detail.json)")
    ap.add_argument("--ceilings", action="store_true", help="re-measure the box's ceilings (read / copy bandwidth, VALU issue rate) instead of reading profiles/ceilings.json")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child runs (roofline.traffic is then null)")
    ap.add_argument("--no-strong", action="store_true", help="skip the 64 M-ray strong-scaling batch of config 4")
    ap.add_argument("--no-reference-blob", action="store_true", help="skip the drop-in leg (the timed step on blobs of the real tinybvh BuildHQ)")
    ap.add_argument("--one-process-devices", type=int, default=0, help="also trace config 4's 64 M-ray batch from THIS process over K contexts (device i mod the visible devices) through tbvh_intersect_sharded_device")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity check of the timed kernels (the line then says parity.checked: false; without this flag a check that could not run is a failure)")
    ap.add_argument("--device-build", action="store_true", help="build the layout on the device (tbvh_build_device: LBVH) instead of the host builder")
    ap.add_argument("--blob-cache", default="", help="BVH8_CWBVH blob file (BVH8_CWBVH::Save format): read if it exists, else written after the host build (child runs of one bench share one build)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--hbm-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--ref-ocl", action="store_true", help=argparse.SUPPRESS)   # (scene child: also time the reference's own OpenCL kernel of the layout on the same blobs and rays)
    ap.add_argument("--tlas-child", action="store_true", help=argparse.SUPPRESS)
    # flags of earlier rounds, accepted and ignored (their legs now live behind --detail)
    for old in ("--no-configs", "--no-hbm-regime", "--no-rotated", "--no-other-layouts", "--no-host-rays"):
        ap.add_argument(old, action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    legs = Legs()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    use_dist = world > 1 or bool(os.environ.get("TBVH_BENCH_FORCE_DIST"))  # the env knob exercises the RCCL path on one GPU
    child = a.pmc_child or a.hbm_child or a.tlas_child
    # `python bench.py --gpus N` WITHOUT torchrun (the driver's plain command line): this one process drives N devices itself — one context,
    # one BVH replica and one set of batches per device, one host thread enqueueing every device's launches (resolve_devices below)
    inproc_devices = None
    if not use_dist and "WORLD_SIZE" not in os.environ and a.gpus > 1 and not child:
        import tinybvh_amd as tb_
        try:
            inproc_devices = resolve_devices(a.gpus, tb_.device_count(), os.environ.get("TBVH_BENCH_DEVICE_MAP"))
        except ValueError as e:
            print(json.dumps({"metric": METRIC, "value": None, "unit": "MRays/s", "n_gpus": a.gpus, "error": str(e), "visible_devices": tb_.device_count()}), flush=True)
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
        import bench_detail
        bench_detail.tlas_child(a, tb, R, scenes)
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

    others = []   # the other devices of a one-process N-GPU run: (context, replica, camera batch, bounce batch)

    def sync_all():
        if use_dist:
            import torch
            dist.barrier()
            torch.cuda.synchronize()
            flush_c_stdio()
        ctx.synchronize()
        for o_ in others:
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
    own_cache = None
    if not child and not use_dist and not a.blob_cache and a.layout == tb.LAYOUT_CWBVH and not a.device_build and not a.no_pmc:
        # the rocprofv3 --pmc children of this run load the blobs this process builds (the blob cache of SURVEY par. 8(f4)) instead of building again
        import tempfile
        own_cache = tempfile.mkdtemp(prefix="tbvh_bench_", dir="/tmp")
        a.blob_cache = os.path.join(own_cache, f"{a.scene}.cwbvh")
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
    legs.mark("scene_and_build")

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
    if inproc_devices:
        for k_, dev in enumerate(inproc_devices[1:], start=1):
            c_ = tb.Context(dev)
            r_ = tb.BVH8_CWBVH(c_).Upload(sc.host.blob(0, np.uint32, 4), sc.host.blob(1, np.uint32, 4)) if a.layout == tb.LAYOUT_CWBVH else tb.LAYOUT_CLASSES[a.layout](c_).Build(verts)
            dv_, dp_, dd_, ds_, dt_ = make_batches(c_, r_, 1000 * (k_ + 1))
            for x_ in (dv_, ds_, dt_):
                c_.free(x_)
            others.append((c_, r_, dp_, dd_))
        log(f"[bench] one process, {len(inproc_devices)} contexts on devices {inproc_devices}")

    if a.pmc_child:   # under rocprofv3 --pmc: 3 preparation launches above, then (primary, diffuse) x 3; nothing else
        for _ in range(3):
            sc.intersect_device_fresh(d_prim, n, 1e30)
            sc.intersect_device_fresh(d_diff, n, 1e30)
        ctx.synchronize()
        ctx.close()
        return

    if a.hbm_child:   # a (scene, layout) of bench_detail.py's legs: timed launches, S / T, optionally the reference's OpenCL kernel; one JSON line
        hbm_child(a, tb, ctx, sc, verts, label, n_tris, n, d_prim, d_diff)
        return
    legs.mark("ray_batches")

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

    def settle(scene_, ctx_, d_, anyhit=False, tries=14):
        """untimed launches until the scene's coherent-schedule tuner has decided for this class of query (it tries three schedules three times each)"""
        if a.layout != tb.LAYOUT_CWBVH:
            return 0
        for _ in range(tries):
            if scene_.coherent_schedule(anyhit)[0]:
                break
            if anyhit:
                scene_.occluded_device(d_, n, d_occ)
            else:
                scene_.intersect_device_fresh(d_, n, 1e30)
            ctx_.synchronize()
        return int(scene_.coherent_schedule(anyhit)[0])

    # the any-hit pass (config "16 M IsOccluded shadow rays") is not part of the metric's step (primary + diffuse): HIP events only
    for i in range(2):
        sc.occluded_device(d_shad, n, d_occ)
    ctx.synchronize()
    settle(sc, ctx, d_shad, anyhit=True)
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
    settle(sc, ctx, d_prim)                 # the tuner has decided before the timed region, whatever --warmup was
    for c_, r_, dp_, dd_ in others:
        settle(r_, c_, dp_)
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
    legs.mark("shadow_warmup_timed_steps")

    # ---- config 4 as BASELINE.json words it: ONE 64 M-ray diffuse batch, sharded over the ranks (strong scaling) ------
    strong = None
    km4_local, dm4_local, el4_local = [], [], 0.0
    n_gpus = len(inproc_devices) if inproc_devices else world
    for p_ in (d_tmp, d_shad, d_occ):
        ctx.free(p_)
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
            import bench_detail
            one_proc = bench_detail.strong_one_process(tb, R, sc, verts, eye, view, 8192 if a.side >= 4096 else 2 * a.side, kdev, log, devices=inproc_devices)
        except Exception as e:
            log(f"[bench] one-process multi-device batch failed: {e!r}")
    legs.mark("config4_strong")

    # ---- the drop-in case (rank 0, outside the timed region): the same step on blobs of the real tinybvh BuildHQ ------------------------
    ref_blob = None
    if rank == 0 and not a.no_reference_blob and a.layout == tb.LAYOUT_CWBVH:
        try:
            ref_blob = reference_blob_step(tb, ctx, verts, d_prim, d_diff, n, timed_got, par_stride, ns_par, a.steps)
        except Exception as e:
            log(f"[bench] reference-blob step failed: {e!r}")
            ref_blob = {"kind": "error", "error": repr(e)[:200]}
        legs.mark("reference_blob")

    full = None
    parity = {"n": 0, "ok": False}
    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        rays_per_step = 2 * n * n_gpus  # primary + diffuse (the metric); shadow reported beside it
        value = rays_per_step / (elapsed / a.steps) / 1e6
        mean = {k: float(np.mean(v)) for k, v in kern_ms.items()}
        detail = {k + "_mrays": n / (mean[k] * 1e-3) / 1e6 for k in mean}
        detail["kernel_ms"] = mean
        detail["primary_plus_diffuse_kernel_mrays"] = 2 * n / ((mean["primary"] + mean["diffuse"]) * 1e-3) / 1e6
        # what a step costs beyond its two queries' own HIP-event time (launch latency the stream could not hide, the barrier)
        detail["dispatch_gap_ms"] = ms_per_step - (mean["primary"] + mean["diffuse"])
        sched = {}
        if a.layout == tb.LAYOUT_CWBVH:
            for kind, anyhit in (("closest_hit", False), ("any_hit", True)):
                t_ = sc.coherent_schedule(anyhit)
                sched[kind] = {"decision": SCHEDULE_NAMES[t_[0]], "samples": [t_[1], t_[2]]}
            sched["how"] = "measured per scene by the library during the first launches (CohTuner, tinybvh_amd/csrc/capi_internal.h); TBVH_COHERENT_TUNER pins it"
        detail["coherent_schedule"] = sched
        detail["config4_strong"] = strong if strong is not None else one_proc
        if inproc_devices:
            detail["launch"] = {"how": "one process, one context per device, one host thread enqueues every device's launches (no torchrun, no collective)", "devices": inproc_devices,
                                "visible_devices": tb.device_count(), "device_map_env": os.environ.get("TBVH_BENCH_DEVICE_MAP")}
        detail["per_gpu"] = per_gpu
        if replication:
            detail["bvh_replication"] = replication
        if one_proc:
            detail["config4_strong_one_process"] = one_proc
        detail["reference_blob"] = ref_blob

        # ---- parity of the timed kernels, in this run (outside the timed region; the oracle is the checker, never the thing measured) -------
        # a strided 65 k sample of the primary and the diffuse batch: the GPU records the timed launches left in HBM against BVH::Intersect
        # restated (oracle/tbvh_oracle.c, library tie rule) on the BVH2 the layout was encoded from, and against the oracle's mirror of this
        # layout (which also counts node visits S and triangle tests T per ray for the roofline lines); the shadow batch's occlusion flags
        # against BVH::IsOccluded restated.  A real mismatch makes this process exit non-zero after the JSON line.
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
        legs.mark("parity")

        # ---- roofline (contract: `achieved` = ALGORITHMIC bytes per launch / mean launch time; `traffic` = PMC bytes per launch) ------------------
        roof = None
        try:
            ceil = ceilings(ctx, a.ceilings, log)
            # (the children run the schedule this process's tuner settled on for coherent batches, pinned: a child is too short to decide for itself)
            a.coh_pin = {1: "0", 2: "2", 3: "3"}.get(sc.coherent_schedule(False)[0] if a.layout == tb.LAYOUT_CWBVH else 0)
            passes = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES") + \
                     (("TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum",) if a.detail else ())
            pm = live_counters(a, log, passes=passes) if (world == 1 and not a.no_pmc) else None
            roof = roofline_lines(a.layout, n, mean, S_T, pm, ceil)
        except Exception as e:  # the checker is optional for the number itself
            log(f"[bench] roofline failed: {e!r}")
        legs.mark("roofline_pmc_children")

        cpu = None
        if not a.no_cpu_baseline:
            try:
                cpu = cpu_baseline(tb, ctx, verts, d_prim, d_diff, n, more=a.detail)
            except Exception as e:
                log(f"[bench] cpu baseline failed: {e!r}")
            legs.mark("cpu_baseline")

        # BASELINE configs 2 and 5 beside the headline (cheap: ~1 s each; their comparisons with the reference's OpenCL kernels live behind --detail)
        if world == 1 and not inproc_devices:
            import bench_detail as bd
            for name, fn in (("config2", lambda: bd.config2_quick(tb, ctx, R, scenes)), ("tlas_1000_instances", lambda: bd.tlas_frames_leg(tb, ctx, R, scenes))):
                try:
                    detail[name] = fn()
                except Exception as e:
                    log(f"[bench] {name} failed: {e!r}")
            legs.mark("config2_config5")

        if a.detail and world == 1:
            side_legs(a, tb, ctx, R, scenes, sc, verts, d_verts, d_prim, d_diff, n, cam, light, seed, kern_ms, timed_got, par_stride, ns_par, detail, legs,
                      (roof or {}).get("ceilings", {}).get("valu_ginstr_per_s"))

        full = {
            "metric": METRIC, "value": value, "unit": "MRays/s",
            "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{label}; {LAYOUT_NAMES[a.layout]}; per GPU per step {n} primary + {n} diffuse (depth 1-3) Intersect",
                       "scene_tris": n_tris, "layout": LAYOUT_NAMES[a.layout], "rays_per_gpu_per_step": 2 * n, "shadow_rays_per_gpu": n,
                       # the real bistro_ext_part{1,2}.bin are stripped from the reference checkout: a labelled procedural stand-in of the same triangle count unless
                       # TBVH_SCENE_DIR holds the real files; its node visits / triangle tests per ray (roofline.nodes_per_ray, tris_per_ray) are NOT real Bistro's
                       "scene_is_stand_in": "stand-in" in label,
                       "sharding": f"value: weak — every GPU its own {2 * n}-ray step, BVH replicated, no collective; config4_strong: one 64 M-ray batch in {n_gpus} contiguous shard(s)"},
            "parity_checked": bool(parity.get("n")) and "error" not in parity, "parity_ok": bool(parity.get("ok", False)),
            "detail": detail, "roofline": roof, "cpu_baseline": cpu, "legs_s": legs.rows,
        }
        path = a.detail_out or (os.path.join("gpurun_out", "bench_detail.json") if os.path.isdir("gpurun_out") else "bench_detail.json")
        try:
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
        except OSError as e:
            log(f"[bench] could not write {path}: {e!r}")
            path = None
        line = compact_line(full, path)
        flush_c_stdio()
        print(json.dumps(line, separators=(",", ":")), flush=True)
    sync_all()
    if use_dist:
        dist.destroy_process_group()
    for c_, r_, dp_, dd_ in others:
        c_.close()
    ctx.close()
    if own_cache:
        import shutil
        shutil.rmtree(own_cache, ignore_errors=True)
    if rank == 0 and not a.no_parity:
        if "error" in parity:      # the timed kernels were never checked: not a result either
            log(f"[bench] the parity check of the timed kernels could not run: {parity['error']}")
            sys.exit(4)
        if not parity.get("ok", False):
            log(f"[bench] PARITY MISMATCH on the timed kernels: {parity}")
            sys.exit(3)


def side_legs(a, tb, ctx, R, scenes, sc, verts, d_verts, d_prim, d_diff, n, cam, light, seed, kern_ms, timed_got, par_stride, ns_par, detail, legs, valu_ceiling):
    """`--detail`: the legs of bench_detail.py, each guarded (a failing leg is a note in the sidecar, never a lost headline)"""
    import bench_detail as bd

    def leg(name, fn):
        try:
            detail[name] = fn()
        except Exception as e:
            log(f"[bench] leg {name} failed: {e!r}")
            detail[name] = {"error": repr(e)[:300]}
        legs.mark(name)
    cfg = {}
    leg("_cfg12", lambda: cfg.update(bd.configs_1_and_2(tb, ctx, R, scenes)) or None)
    detail.pop("_cfg12", None)
    detail["config1"], detail["config2"] = cfg.get("config1"), cfg.get("config2")
    if a.layout == tb.LAYOUT_CWBVH:
        leg("ref_opencl_cwbvh", lambda: bd.reference_opencl_headline(tb, ctx, sc, d_prim, d_diff, n, kern_ms, timed_got, par_stride, ns_par))
    leg("host_rays", lambda: bd.host_rays_leg(tb, ctx, sc, d_prim, n))
    leg("wavefront_frame_3_bounces", lambda: bd.wavefront_leg(tb, ctx, sc, d_verts, cam, light, a.side, seed))
    if isinstance(detail.get("tlas_1000_instances"), dict) and "error" not in detail["tlas_1000_instances"]:
        try:
            detail["tlas_1000_instances"].update(bd.tlas_leg(a, log, valu_ceiling))
        except Exception as e:
            log(f"[bench] TLAS leg failed: {e!r}")
        legs.mark("tlas_vs_reference")
    if a.layout == tb.LAYOUT_CWBVH:
        leg("hbm_regime", lambda: bd.hbm_regime(a, log))
        if a.scene == "bistro":
            leg("rotated_scene", lambda: bd.scene_leg(a, log, "street_rot", a.side, 10, False, valu_ceiling,
                                                      note="the bench scene rotated by irrational angles about two axes, same camera carried along; S / T per ray from the oracle's mirror"))
    if a.scene == "bistro":
        detail["other_layouts"] = {}
        for name, lay, kern, refk in (("BVH_GPU", 5, "k_bvh2", "batch_ailalaine (traverse_bvh2.cl:209-219)"), ("BVH4_GPU", 8, "k_bvh4", "batch_gpu4way (traverse_bvh4.cl:277-286)")):
            # as shipped (the scene's 8-wide copy serves the queries: DESIGN.md par. 3.4) with the reference's OpenCL kernel beside it, then the layout's OWN kernel
            # (TBVH_WIDE_COPY_MIN=0: no copy) with its counters
            for tag, env_extra, ocl, what in ((name, None, True, f"the scene's 8-wide copy (as shipped) on the headline batches next to the reference's {refk} (ROCm OpenCL, same GPU, blobs and rays)"),
                                              (name + "_native", {"TBVH_WIDE_COPY_MIN": "0"}, False, f"{kern} on the uploaded nodes (TBVH_WIDE_COPY_MIN=0)")):
                try:
                    detail["other_layouts"][tag] = bd.scene_leg(a, log, a.scene, a.side, lay, ocl, valu_ceiling, note=what, env_extra=env_extra)
                except Exception as e:
                    detail["other_layouts"][tag] = {"error": repr(e)[:300]}
                legs.mark("other_layout_" + tag)
    leg("device_side_ops", lambda: bd.device_ops_leg(tb, ctx, sc, verts, d_verts))      # LAST: refits `sc` to moved vertices


def hbm_child(a, tb, ctx, sc, verts, label, n_tris, n, d_prim, d_diff):
    """`--hbm-child`: the timed kernels on one (scene, layout) for bench_detail.py's legs — HIP-event launch times, S / T per ray from the oracle's
    mirror on a strided 16 k sample and, --ref-ocl, the reference's own OpenCL kernel of the layout on the same blobs and rays; one JSON line."""
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


def ceilings(ctx, measure, log_):
    """The box's ceilings the roofline lines divide by besides the guide's 8 TB/s: streaming read / copy bandwidth and the VALU issue rate of the
    node-test instruction mix (tbvh_measure_*).  Cached in profiles/ceilings.json (they are properties of the part, stable to ~1 %); `--ceilings`
    measures them again (about 0.2 s of GPU time, 8 launches of a 7 ms microbenchmark among them) and reports the live values."""
    cached = None
    try:
        cached = json.load(open(CEILINGS_FILE))
    except (OSError, ValueError):
        pass
    if cached and not measure:
        return dict(cached, source="profiles/ceilings.json")
    c = {"read_gbps": ctx.read_bandwidth_gbps(1 << 30, 5), "copy_gbps": ctx.copy_bandwidth_gbps(1 << 30, 5), "valu_ginstr_per_s": ctx.valu_issue_ginstr(3)}
    try:
        up, down = ctx.link_bandwidth_gbps(1 << 28, 3)
        c.update(link_h2d_gbps=up, link_d2h_gbps=down)
    except Exception as e:
        log_(f"[bench] link rate measurement failed: {e!r}")
    c["source"] = "measured in this run (tbvh_measure_read_bandwidth / _copy_bandwidth / _valu_issue / _link_bandwidth)"
    return c


def roofline_lines(layout, n, mean, S_T, pm, ceil):
    """roofline of the dominant kernel (diffuse batch) + the primary batch, from: mean HIP-event launch ms (`mean`), node visits S / triangle tests T
    per ray (`S_T`, the oracle's mirror on the parity sample), live counters per launch (`pm`, may be None) and the box's ceilings.
        achieved      ALGORITHMIC bytes per launch (SURVEY par. 8(d): 64 + 16 + node_bytes S + tri_bytes T per ray, x rays) / launch time — counts every
                      visit as an HBM fetch while most are served by L1 / L2 / Infinity Cache, so frac = achieved / 8 TB/s may exceed 1
        traffic       bytes the launch moved beyond the L2s: FETCH_SIZE x 2 KB (the guide's gfx950 correction: the counter tallies 64 of every 128 bytes)
                      + WRITE_SIZE x 1 KB; includes Infinity-Cache hits — an upper bound on HBM bytes; traffic_frac = traffic / time / 8 TB/s
        valu          SQ_INSTS_VALU / time against the measured issue ceiling; lanes = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)"""
    nb, tbytes = LAYOUT_BYTES[layout]
    valu_ginstr, read_gbps = ceil.get("valu_ginstr_per_s"), ceil.get("read_gbps")
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
        row = {"avg_launch_ms": mean[kind], "rays_per_launch": n, "nodes_per_ray": S, "tris_per_ray": T, "algorithmic_bytes_per_ray": bpr,
               "achieved": alg, "peak": 8000.0, "unit": "GB/s", "frac": alg / 8000.0, "traffic": tr,
               "traffic_gbps": tr / sec / 1e9 if tr else None, "traffic_frac": tr / sec / 8e12 if tr else None,
               "traffic_frac_of_measured_read": (tr / sec / 1e9 / read_gbps) if (tr and read_gbps) else None,
               "traffic_over_algorithmic": tr / (bpr * n) if tr else None}
        if c.get("SQ_INSTS_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
            rate = c["SQ_INSTS_VALU"] / sec / 1e9                        # G wave-instructions / s, whole chip
            lane = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
            issue = rate / valu_ginstr if valu_ginstr else None
            row.update(valu_insts_per_ray=c["SQ_INSTS_VALU"] * 64.0 / n, valu_issue_frac=issue, lane_utilisation=lane,
                       valu_useful_frac=(issue * lane) if issue is not None else None)
        if c.get("TCC_EA0_RDREQ_sum"):
            row.update(l2_miss_latency_cycles=c["TCC_EA0_RDREQ_LEVEL_sum"] / c["TCC_EA0_RDREQ_sum"], l2_miss_requests_per_ray=c["TCC_EA0_RDREQ_sum"] / n)
        lines[kind] = row
    d_ = lines.get("diffuse")
    if not d_:
        return None
    kname = {10: "k_cwbvh<incoherent flavor> (diffuse batch)", 8: "k_bvh4_w8 (diffuse batch)", 5: "k_bvh2 (diffuse batch)"}[layout]
    roof = {"bound": "hbm", "kernel": kname}
    roof.update(d_)
    roof["traffic_source"] = (pm or {}).get("source")
    roof["ceilings"] = ceil
    roof["note"] = ("achieved = algorithmic bytes / launch time (the contract's line; > peak is possible: the tree is served mostly by L1 / L2 / Infinity Cache); "
                    "traffic = live FETCH_SIZE x 2 KB + WRITE_SIZE x 1 KB per launch, Infinity-Cache hits included")
    roof["primary"] = lines.get("primary")
    return roof


def reference_blob_step(tb, ctx, verts, d_prim, d_diff, n, timed_got, par_stride, ns_par, steps):
    """The drop-in case in the driver's own run: the SAME timed step (primary + diffuse, fresh) on blobs encoded by the real tiny_bvh.h —
    BVH8_CWBVH::BuildHQ through oracle/_ref (tiny_bvh_speedtest.cpp:1196-1204) — uploaded verbatim through tbvh_upload_cwbvh.  The scene's
    coherent-schedule tuner settles first (untimed launches until tbvh_scene_get_coherent_schedule reports a decision), then `steps` steps are
    enqueued back to back like the headline's and their HIP-event times read afterwards (tiny_bvh_speedtest.cpp:1117-1137's scheme).
    Also the headline's parity against the REAL reference: a 65 k strided sample of the primary and of the diffuse batch traced by the real
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
    tuner_launches = 0
    for _ in range(16):                 # untimed: the tuner tries its three schedules three times each before it decides
        if sc.coherent_schedule(False)[0]:
            break
        sc.intersect_device_fresh(d_prim, n, 1e30); ctx.synchronize()
        tuner_launches += 1
    decided = int(sc.coherent_schedule(False)[0])
    k = min(max(steps, 1), 64)
    for p_ in range(2):                 # warm-up of the settled schedule
        sc.intersect_device_fresh(d_prim, n, 1e30); sc.intersect_device_fresh(d_diff, n, 1e30)
    ctx.synchronize()
    t0 = time.perf_counter()
    for p_ in range(k):
        sc.intersect_device_fresh(d_prim, n, 1e30)
        sc.intersect_device_fresh(d_diff, n, 1e30)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / k
    hist = ctx.time_history(2 * k)
    mp, md = float(np.mean(hist[0::2])), float(np.mean(hist[1::2]))
    out = {"kind": "reference", "builder": "tinybvh BVH8_CWBVH::BuildHQ (oracle/_ref), blobs uploaded verbatim", "host_build_s": build_s, "node_blocks": int(nodes.shape[0]), "tri_blocks": int(tris.shape[0]),
           "steps": k, "coherent_schedule": SCHEDULE_NAMES.get(decided, str(decided)), "tuner_launches_before_decision": tuner_launches,
           "primary_ms": mp, "diffuse_ms": md, "step_wall_ms": wall * 1e3,
           "primary_mrays": n / (mp * 1e-3) / 1e6, "diffuse_mrays": n / (md * 1e-3) / 1e6, "primary_plus_diffuse_mrays": 2 * n / ((mp + md) * 1e-3) / 1e6,
           "primary_plus_diffuse_wall_mrays": 2 * n / wall / 1e6}
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
        out["vs_real_reference"] = dict(vs_blob, rule=rule, rays_sampled=2 * ns_par, differ_from_reference=sum(v["differ_from_reference"] for v in vs_blob.values()))
        if vs_timed:
            out["timed_launches_vs_real_reference"] = dict(vs_timed, rule=rule, rays_sampled=2 * ns_par,
                                                           differ_from_reference=sum(v["differ_from_reference"] for v in vs_timed.values()))
    except Exception as e:
        out["vs_real_reference"] = {"error": repr(e)[:300]}
    sc.free()
    return out


def _r(x, nd=4):
    """numbers of the line: 4 significant decimals are what the measurement carries"""
    if isinstance(x, float):
        return round(x, nd)
    return x


def compact_line(full, detail_file=None):
    """The contract line from the run's full record: fixed keys, short strings, < LINE_LIMIT bytes (tests/test_bench_line.py).  Anything that does
    not fit is dropped from the END of `optional` — never a contract key."""
    d = full.get("detail") or {}
    line = {k: _r(full.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = full.get("config") or {}
    line["config"] = {"workload": str(cfg.get("workload", ""))[:220], "scene_tris": cfg.get("scene_tris"), "layout": cfg.get("layout"), "rays_per_gpu_per_step": cfg.get("rays_per_gpu_per_step"),
                      "scene_is_stand_in": cfg.get("scene_is_stand_in")}
    roof = full.get("roofline")
    if roof:
        short = ("avg_launch_ms", "nodes_per_ray", "tris_per_ray", "algorithmic_bytes_per_ray", "achieved", "frac", "traffic", "traffic_frac", "valu_issue_frac", "lane_utilisation")
        line["roofline"] = {"kernel": roof.get("kernel"), "bound": roof.get("bound"), "achieved": _r(roof.get("achieved"), 1), "peak": roof.get("peak"), "unit": roof.get("unit"),
                            "frac": _r(roof.get("frac")), "traffic": roof.get("traffic"), "traffic_gbps": _r(roof.get("traffic_gbps"), 1), "traffic_frac": _r(roof.get("traffic_frac")),
                            "avg_launch_ms": _r(roof.get("avg_launch_ms")), "rays_per_launch": roof.get("rays_per_launch"),
                            "algorithmic_bytes_per_ray": _r(roof.get("algorithmic_bytes_per_ray"), 1), "nodes_per_ray": _r(roof.get("nodes_per_ray"), 2), "tris_per_ray": _r(roof.get("tris_per_ray"), 2),
                            "valu_issue_frac": _r(roof.get("valu_issue_frac")), "lane_utilisation": _r(roof.get("lane_utilisation")),
                            "traffic_source": "live rocprofv3 --pmc children: FETCH_SIZE x2KB + WRITE_SIZE x1KB per launch (incl. Infinity-Cache hits)" if roof.get("traffic") else None,
                            "note": "achieved = algorithmic bytes (64+16+80S+48T per ray) / launch time; may exceed peak: tree served mostly on-chip",
                            "primary": {k: _r((roof.get("primary") or {}).get(k)) for k in short} if roof.get("primary") else None}
    else:
        line["roofline"] = None
    cpu = full.get("cpu_baseline")
    line["cpu_baseline"] = ({"value": _r(cpu.get("value")), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"), "sample": str(cpu.get("sample", ""))[:160],
                             "threads_1": _r((cpu.get("threads_1") or {}).get("value"))} if cpu else None)
    par = d.get("parity_sample") or {}
    vr = par.get("vs_real_reference") or {}
    line["parity"] = {"checked": bool(full.get("parity_checked")), "ok": bool(full.get("parity_ok")), "rays_sampled": 2 * par.get("n", 0),
                      "differ_from_oracle": (par.get("hitmiss", 0) + par.get("prim_real", 0) + par.get("t_bad", 0) + par.get("uv_bad", 0) + par.get("tie", 0) + par.get("not_bit_identical", 0)) if par.get("n") else None,
                      "differ_from_reference": vr.get("differ_from_reference"), "shadow_flags_differ": par.get("shadow_flags_differ")}
    rb = d.get("reference_blob") or {}
    line["reference_blob"] = ({"primary": _r(rb.get("primary_mrays"), 1), "diffuse": _r(rb.get("diffuse_mrays"), 1), "combined": _r(rb.get("primary_plus_diffuse_mrays"), 1),
                               "steps": rb.get("steps"), "schedule": rb.get("coherent_schedule"), "differ_from_reference": (rb.get("vs_real_reference") or {}).get("differ_from_reference"),
                               "builder": "tinybvh BVH8_CWBVH::BuildHQ, blobs verbatim"} if rb.get("kind") == "reference" else {"kind": rb.get("kind", "n/a")})
    line["kernel_mrays"] = {"primary": _r(d.get("primary_mrays"), 1), "diffuse": _r(d.get("diffuse_mrays"), 1), "combined": _r(d.get("primary_plus_diffuse_kernel_mrays"), 1), "shadow": _r(d.get("shadow_mrays"), 1)}
    sch = d.get("coherent_schedule") or {}
    line["schedule"] = {k: (sch.get(k) or {}).get("decision") for k in ("closest_hit", "any_hit")} if sch else None
    c4 = d.get("config4_strong") or {}
    line["config4_strong"] = {"rays": c4.get("rays"), "mrays": _r(c4.get("mrays"), 1), "shards": full.get("n_gpus")} if c4 else None
    optional = []
    c2 = d.get("config2")
    if c2:
        optional.append(("config2", {"layout": "BVH_GPU", "rays": c2.get("rays"), "mrays": _r(c2.get("bvh_gpu_mrays"), 1), "ref_opencl_mrays": _r(c2.get("ref_opencl_mrays"), 1) if isinstance(c2.get("ref_opencl_mrays"), float) else None}))
    t5 = d.get("tlas_1000_instances")
    if t5 and "error" not in t5:
        optional.append(("config5", {"blas_layout": t5.get("blas_layout"), "camera_mrays": _r(t5.get("camera_mrays"), 1), "tlas_rebuild_ms": _r(t5.get("device_tlas_rebuild_ms")), "blas_refit_ms": _r(t5.get("device_blas_refit_ms"))}))
    if full.get("legs_s"):
        optional.append(("legs_s", {k: v for k, v in full["legs_s"]}))
    line["detail_file"] = detail_file
    for k, v in optional:
        line[k] = v
    while len(json.dumps(line, separators=(",", ":"))) >= LINE_LIMIT and optional:
        line.pop(optional.pop()[0], None)
    return line


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
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--no-pmc", "--scene", a.scene, "--side", str(a.side), "--layout", str(a.layout),
               "--variant", str(a.variant)] + (["--device-build"] if getattr(a, "device_build", False) else []) + \
              (["--blob-cache", a.blob_cache] if getattr(a, "blob_cache", "") else [])
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TBVH_BENCH_FORCE_DIST"):
            env.pop(k, None)
        if getattr(a, "coh_pin", None) is not None:
            env["TBVH_COHERENT_TUNER"] = a.coh_pin
        env.update(getattr(a, "env_extra", None) or {})
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


def cpu_baseline(tb, ctx, verts, d_prim, d_diff, n, more=True):
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
