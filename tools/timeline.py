#!/usr/bin/env python
"""Kernel timeline (start offset, duration, stream/queue) of the last `n` dispatches in a rocprofv3 rocpd
database -- to see what overlaps what in a pipelined bench pass.  usage: timeline.py <results.db> [n [first]]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
t = "kernels" if "kernels" in tabs else kd[0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
name = "name" if "name" in cols else ("kernel_name" if "kernel_name" in cols else cols[0])
first = int(sys.argv[3]) if len(sys.argv) > 3 else -1      # >= 0: n dispatches from the first-th on (by start time) instead of the last n
if first >= 0:
    q = "select %s, start, end, %s from %s order by start asc limit %d offset %d" % (name, "queue_id" if "queue_id" in cols else "0", t, n, first)
    rows = list(cur.execute(q))
else:
    q = "select %s, start, end, %s from %s order by start desc limit %d" % (name, "queue_id" if "queue_id" in cols else "0", t, n)
    rows = list(cur.execute(q))[::-1]
t0 = rows[0][1]
for nm, s, e, qid in rows:
    print("%10.3f ms  +%8.3f ms  q%-4s %s" % ((s - t0) / 1e6, (e - s) / 1e6, qid, nm[:60]))
