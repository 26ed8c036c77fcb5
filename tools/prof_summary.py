#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats result (rocpd sqlite) into a small text summary for profiles/."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print(f"# rocprofv3 --kernel-trace --stats -- {' '.join(sys.argv[2:])}")
print("# kernel, calls, total_us, average_us, percent")
for n, c, t, a, p in rows:
    print(f"{n[:90]}, {c}, {t:.0f}, {a:.0f}, {p:.3f}")
