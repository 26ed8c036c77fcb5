#!/usr/bin/env python3
"""print the essentials of a bench.py JSON line: value, ms, roofline bound / frac, parity flags"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def show(name, x):
    if x is None or "error" in x:
        print(name, "ERROR", None if x is None else x["error"]); return
    r = x.get("roofline") or {}
    p = x.get("parity") or {}
    print(f"{name:14s} {x['value']:12.5g} {x['unit']:10s} ms {x.get('ms_per_step', x.get('ms_per_frame', 0)):9.4f}  kernel {r.get('kernel')} "
          f"{r.get('avg_launch_ms', 0):.4f} ms bound {r.get('bound')} frac {r.get('frac', 0):.3f} {r.get('fractions')}  parity {p.get('parity_ok', p.get('parity_sample_ok'))}")
show("headline", d)
for k, v in (d.get("secondary") or {}).items():
    show(k, v)
