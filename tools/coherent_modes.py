"""Coherent batches under the three schedules the scene tuner chooses between (TBVH_COHERENT_TUNER = 0 deferred + gated, 2 strict, 3 one traversal per wave),
one process per pin (the knob is read when a context is made): 16.7 M camera and shadow rays, median of 5, and the records byte-compared with pin 0's."""
import os, sys, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import tinybvh_amd as tb
    from tinybvh_amd import rays as R, scenes
    import zlib
    name, side = sys.argv[2], int(sys.argv[3])
    verts, label = scenes.get(name)
    ctx = tb.Context(0)
    sc = tb.BVH8_CWBVH(ctx).Build(verts)
    n = side * side
    d_p, d_s, d_occ = ctx.malloc(n * 64), ctx.malloc(n * 64), ctx.malloc(n)
    ctx.generate_primary(R.camera(*scenes.cameras(name)[0], side, side, 1, 1), d_p, 0, n)
    sc.intersect_device_fresh(d_p, n, 1e30)
    ext = float((verts[:, :3].max(0) - verts[:, :3].min(0)).max())
    ctx.generate_shadow(d_p, d_s, n, (0.0, 0.9 * float(verts[:, 1].max()), 0.0), ext * 5e-7)
    out = {}
    for kind, fn in (("camera", lambda: sc.intersect_device_fresh(d_p, n, 1e30)), ("shadow", lambda: sc.occluded_device(d_s, n, d_occ))):
        ms = []
        for p_ in range(7):
            fn(); ctx.synchronize()
            if p_ >= 2:
                ms.append(ctx.time_last_ms())
        out[kind] = n / (float(np.median(ms)) * 1e-3) / 1e6
    got = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(got, d_p); occ = np.zeros(n, np.uint8); ctx.from_device(occ, d_occ)
    out["crc_camera"] = zlib.crc32(got.view(np.uint8)[:, None].reshape(n, 64)[:, 48:].tobytes()); out["crc_shadow"] = zlib.crc32(occ.tobytes()); out["label"] = label
    print(json.dumps(out)); ctx.close(); sys.exit(0)
for name, side in (("bistro", 4096), ("street_rot", 4096), ("sponza", 4096), ("bistro", 2048)):
    base = None
    for pin in ("0", "2", "3"):
        env = dict(os.environ, TBVH_COHERENT_TUNER=pin)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, str(side)], env=env, capture_output=True, text=True, timeout=600)
        try:
            o = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][-1])
        except Exception:
            print(name, side, "pin", pin, "FAILED", r.stderr[-400:]); continue
        if base is None:
            base = o; print(o["label"][:60], f"{side * side} rays")
        print(f"   pin {pin}: camera {o['camera']:7.0f} ({o['camera'] / base['camera'] - 1:+.1%})  shadow {o['shadow']:7.0f} ({o['shadow'] / base['shadow'] - 1:+.1%})   records {'same' if o['crc_camera'] == base['crc_camera'] else 'DIFFER'} / flags {'same' if o['crc_shadow'] == base['crc_shadow'] else 'DIFFER'}", flush=True)
