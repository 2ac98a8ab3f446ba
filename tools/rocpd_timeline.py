#!/usr/bin/env python3
"""Kernel timeline (start/end in ms since the first kernel, per stream/queue) from a rocprofv3 rocpd sqlite trace."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
q = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else 'tid')
rows = cur.execute("select name, start, end, %s from kernels order by start" % q).fetchall()
t0 = rows[0][1]
for n, s, e, qid in rows:
    if (e - s) < 200000: continue            # skip sub-0.2 ms kernels
    print('%9.2f -> %9.2f  (%7.2f ms)  q=%s  %s' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, qid, n[:40]))
