#!/usr/bin/env python3
"""Tuning: bench value vs the deflate_ways option (concurrent slice ranges of the device DEFLATE)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for w in (sys.argv[1:] or ["1", "2", "4", "8"]):
    env = dict(os.environ, CCT_DEFLATE_WAYS=w)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(w, d["value"], d["ms_per_step"], d["stages"]["ms"], d["verified"], flush=True)
    except Exception as e:  # noqa: BLE001
        print(w, "failed", e, out.stderr[-500:], flush=True)
