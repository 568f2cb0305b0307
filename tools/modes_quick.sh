export TMPDIR=/tmp
python - <<'PY'
import os, sys, subprocess, json
sys.path.insert(0, os.getcwd())
for name, side in (("bistro", 4096), ("street_rot", 4096)):
    for pin in ("0", "3"):
        env = dict(os.environ, TBVH_COHERENT_TUNER=pin)
        r = subprocess.run([sys.executable, "tools/coherent_modes.py", "--child", name, str(side)], env=env, capture_output=True, text=True, timeout=600)
        try:
            o = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][-1]); print(name, side, "pin", pin, round(o["camera"]), round(o["shadow"]), o["crc_camera"], o["crc_shadow"], flush=True)
        except Exception: print("FAILED", r.stderr[-500:])
PY
