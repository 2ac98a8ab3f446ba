#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / min / max (ms), like --stats."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, (end - start) from kernels").fetchall()
agg = {}
for n, d in rows:
    a = agg.setdefault(n, [0, 0, 1 << 62, 0]); a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print('%-70s %6s %12s %12s %12s %12s %7s' % ('kernel', 'calls', 'total_ms', 'avg_ms', 'min_ms', 'max_ms', 'pct'))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-70s %6d %12.3f %12.3f %12.3f %12.3f %6.2f%%' % (n[:70], a[0], a[1] / 1e6, a[1] / a[0] / 1e6, a[2] / 1e6, a[3] / 1e6, 100.0 * a[1] / tot))
