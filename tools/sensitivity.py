#!/usr/bin/env python
"""Are the launch heuristics overfit to the two bench scenes?  (round 4, review item 6)

Every default-on heuristic of the BVH8_CWBVH path, switched off one at a time, on structurally different scenes — none of them the axis-aligned
street / atrium the thresholds were tuned on as they stand — with camera, bounce and shadow batches of 1 / 4.2 / 16.7 M rays.  Interleaved: every
round runs every configuration once per cell, the report is the median over the rounds, as MRays/s of the default and the change when the
heuristic is OFF (positive = the default loses there).

    scenes      street_rot   the Bistro stand-in rotated by irrational angles about two axes (no axis-aligned wall left)
                foliage      2 M thin, randomly oriented triangles in clusters (no large occluder at all)
                bunnies      bunny.bin (the reference's mesh) x 30, each rotated and scaled, flattened to 2.1 M triangles
                soup         2 M uniformly random triangles (tiny_bvh_minimal_gpu.cpp's scene shape, scaled up)
                atrium1m     the Sponza stand-in generator at 1 M triangles (the `small` / probed boundary class by size: ~70 MB)
                street12m    the street generator at 12 M triangles (0.9 GB: beyond the probed size class, padded nodes)
    heuristics  copies       the incoherent-batch copies + two-flavor launch (TBVH_INCOHERENT_COPIES=0 turns them off)
                probe        the per-launch coherence probe and what hangs on it: two flavors, the coherent schedule (off: debug flag 64, every launch unprobed)
                coh. schedule  which schedule serves coherent batches: the library measures per scene (CohTuner) — shown: pinned to the deferred + gated one
                             (TBVH_COHERENT_TUNER=0) and to the strict one (=2), and what the tuner decided
                split        split rays at the end of a launch below 12 M rays (TBVH_SPLIT_RAYS=0)
                waves28      28 instead of 24 waves per CU for the incoherent flavor (off: 24)
                hybridK      first 8192 nodes packed (alternatives: 0 = all padded, all = all packed)
                embed        one triangle in every padded node's line (off: flag 8)
usage: tools/sensitivity.py [--scenes a,b] [--rounds 5] [--sizes 1024,2048,4096] > profiles/r04_sensitivity.txt"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R, scenes  # noqa: E402


def rot(verts, ax, ang):
    c, s = np.float32(np.cos(ang)), np.float32(np.sin(ang))
    v = verts.copy(); i, j = [(1, 2), (2, 0), (0, 1)][ax]
    v[:, i], v[:, j] = c * verts[:, i] - s * verts[:, j], s * verts[:, i] + c * verts[:, j]
    return v


def rotv(p, ax, ang):
    return rot(np.asarray([list(p) + [0]], np.float32), ax, ang)[0, :3]


def make_scene(name):
    rng = np.random.default_rng(11)
    if name == "street_rot":
        return scenes.street_rot(), scenes.street_rot_camera(0)
    if name == "foliage":
        k = 500
        centers = np.stack([rng.uniform(-40, 40, k), rng.uniform(0, 25, k), rng.uniform(-40, 40, k)], -1).astype(np.float32)
        v = scenes._pack([scenes._leaves(rng, centers, (3.0, 2.5, 3.0), 4000, 0.12)])
        return v, ((-55.0, 12.0, -50.0), (0.7, -0.05, 0.7))
    if name == "bunnies":
        b = scenes.load_bin(scenes.find_real("bunny.bin"))
        lo, hi = b[:, :3].min(0), b[:, :3].max(0)
        b0 = (b[:, :3] - (lo + hi) / 2) / np.float32((hi - lo).max())
        out = []
        for i in range(30):
            ang = rng.uniform(0, 6.28, 3)
            p = np.zeros((b0.shape[0], 4), np.float32); p[:, :3] = b0 * np.float32(rng.uniform(2.0, 4.5))
            p = rot(rot(rot(p, 0, ang[0]), 1, ang[1]), 2, ang[2])
            p[:, :3] += np.array([(i % 6) * 5.0, (i // 6 % 2) * 1.5, (i // 6) * 5.0], np.float32)
            out.append(p)
        v = np.concatenate(out)
        return v, ((-8.0, 9.0, -8.0), (0.62, -0.35, 0.7))
    if name == "soup":
        return scenes.soup(2_000_000, seed=7, extent=10.0, size=0.06), ((-4.0, 6.0, -5.0), (0.6, -0.15, 0.78))
    if name == "atrium1m":
        return scenes.atrium(1_000_000, seed=1), scenes.SPONZA_CAMERAS[0]
    if name == "street12m":
        return scenes.get("street12m")[0], scenes.STREET_CAMERAS[0]
    raise KeyError(name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="street_rot,foliage,bunnies,soup,atrium1m,street12m")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--sizes", default="1024,2048,4096")
    a = ap.parse_args()
    sizes = [int(x) for x in a.sizes.split(",")]
    envs = {"base": {}, "copies_off": {"TBVH_INCOHERENT_COPIES": "0"}, "split_off": {"TBVH_SPLIT_RAYS": "0"}, "coh_deferred": {"TBVH_COHERENT_TUNER": "0"}, "coh_strict": {"TBVH_COHERENT_TUNER": "2"}}
    knobs = ("TBVH_INCOHERENT_COPIES", "TBVH_SPLIT_RAYS", "TBVH_COHERENT_TUNER")
    for sname in a.scenes.split(","):
        t0 = time.time()
        verts, (eye, view) = make_scene(sname)
        host = tb.HostBVH(verts, tb.LAYOUT_CWBVH)
        nodes, tris = host.blob(0, np.uint32, 4), host.blob(1, np.uint32, 4)
        mb = (nodes.nbytes + tris.nbytes) / 1e6
        print(f"== {sname}: {verts.shape[0] // 3} triangles, {mb:.0f} MB of CWBVH blobs (generated + built in {time.time() - t0:.1f}s)", flush=True)
        ctxs, scs = {}, {}
        for en, ev in envs.items():
            for k in knobs:
                os.environ.pop(k, None)
            os.environ.update(ev)
            ctxs[en] = tb.Context(0)
            scs[en] = tb.BVH8_CWBVH(ctxs[en]).Upload(nodes, tris)
        for k in knobs:
            os.environ.pop(k, None)
        c0 = ctxs["base"]
        in_size_class = 48e6 <= mb <= 384e6 / 1.0 if False else (48 <= mb <= 384)
        # more scene objects on the base context: other placements of the hybrid copy (only where the size class has one)
        alt = {}
        if in_size_class:
            for nm, K, fl in (("K0", 0, 0), ("Kall", 10**9, 0), ("embed_off", 8192, 8)):
                s_ = tb.BVH8_CWBVH(c0).Upload(nodes, tris)
                c0.set_debug_flags(fl); s_.set_hybrid(K); c0.set_debug_flags(0)
                alt[nm] = s_
        d_verts = c0.malloc(verts.nbytes); c0.to_device(d_verts, verts)
        ext = float((verts[:, :3].max(0) - verts[:, :3].min(0)).max())
        ctr = (verts[:, :3].max(0) + verts[:, :3].min(0)) / 2
        light = (float(ctr[0]), float(verts[:, 1].max() + 0.2 * ext), float(ctr[2]))
        for side in sizes:
            n = side * side
            cam = R.camera(eye, view, side, side, 1, 1)
            d_p, d_b, d_s = (c0.malloc(n * 64) for _ in range(3))
            d_occ = c0.malloc(n)
            c0.generate_primary(cam, d_p, 0, n)
            scs["base"].intersect_device(d_p, n)
            c0.generate_shadow(d_p, d_s, n, light, ext * 5e-7)
            c0.generate_bounce(d_verts, d_p, d_b, n, 77)
            c0.synchronize()
            chk = np.zeros(min(n, 1 << 18), tb.RAY_DTYPE); c0.from_device(chk, d_p)
            hit_frac = float((chk["t"] < 1e30).mean())
            # configurations: (name, context key, scene object, variant, flags)
            if n >= (1 << 21) and in_size_class and scs["base"].coherent_schedule(False)[0] == 0:
                for _ in range(6):          # let the tuner of the default scene settle before anything is timed (it alternates schedules while it measures)
                    scs["base"].intersect_device_fresh(d_p, n, 1e30); scs["base"].occluded_device(d_s, n, d_occ)
                c0.synchronize()
                for _ in range(2):
                    scs["base"].intersect_device_fresh(d_p, n, 1e30); scs["base"].occluded_device(d_s, n, d_occ)
                c0.synchronize()
                print(f"        [coherent-schedule tuner on this scene: closest-hit {scs['base'].coherent_schedule(False)}, any-hit {scs['base'].coherent_schedule(True)}  (decision 1 = deferred + gated, 2 = strict; samples; 1000 x strict / deferred)]", flush=True)
            cfgs = [("default", "base", scs["base"], 0, 0), ("copies off", "copies_off", scs["copies_off"], 0, 0), ("probe off", "base", scs["base"], 0, 64),
                    ("split off", "split_off", scs["split_off"], 0, 0), ("24 waves", "base", scs["base"], 0, 24 << 8)]
            if in_size_class:
                cfgs += [("coh deferred", "coh_deferred", scs["coh_deferred"], 0, 0), ("coh strict", "coh_strict", scs["coh_strict"], 0, 0)]
            cfgs += [(nm.replace("_", " "), "base", s_, 0, 0) for nm, s_ in alt.items()]
            ms = {c[0]: {"camera": [], "bounce": [], "shadow": []} for c in cfgs}
            for r in range(a.rounds + 1):
                for nm, ck, sc, var, fl in cfgs[r % len(cfgs):] + cfgs[:r % len(cfgs)]:      # (the order rotates: whoever runs first in a round sees another clock)
                    cx = ctxs[ck]
                    cx.set_debug_flags(fl); sc.set_variant(var)
                    for kind, fn in (("camera", lambda: sc.intersect_device_fresh(d_p, n, 1e30)), ("bounce", lambda: sc.intersect_device_fresh(d_b, n, 1e30)),
                                     ("shadow", lambda: sc.occluded_device(d_s, n, d_occ))):
                        fn()
                        t = cx.time_last_ms()
                        if r:
                            ms[nm][kind].append(t)
                    cx.set_debug_flags(0); sc.set_variant(0)
            base = {k: float(np.median(v)) for k, v in ms["default"].items()}
            print(f"  {n / 1e6:5.1f} M rays ({hit_frac * 100:.0f} % of camera rays hit): default  camera {n / base['camera'] / 1e3:7.0f}  bounce {n / base['bounce'] / 1e3:7.0f}  shadow {n / base['shadow'] / 1e3:7.0f} MRays/s", flush=True)
            for nm, *_ in cfgs[1:]:
                med = {k: float(np.median(v)) for k, v in ms[nm].items()}
                d = {k: base[k] / med[k] - 1 for k in med}        # rate(off) / rate(default) - 1
                flag = "   <-- the default loses > 3 %" if max(d.values()) > 0.03 else ""
                print(f"        {nm:12s} camera {d['camera']:+6.1%}  bounce {d['bounce']:+6.1%}  shadow {d['shadow']:+6.1%}{flag}", flush=True)
            for p_ in (d_p, d_b, d_s, d_occ):
                c0.free(p_)
        c0.free(d_verts)
        for s_ in list(alt.values()) + list(scs.values()):
            s_.free()
        for cx in ctxs.values():
            cx.close()


if __name__ == "__main__":
    main()
