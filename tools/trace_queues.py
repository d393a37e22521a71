#!/usr/bin/env python
"""Which HIP stream ran on which hardware queue, and the cadence of frame starts, from a rocprofv3 --kernel-trace database.
usage: tools/trace_queues.py <results.db>"""
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, queue_id, stream_id from kernels order by start"))
m = {}
for n, s, e, q, st in rows:
    m.setdefault((st, q), [0, set()])
    m[(st, q)][0] += 1
    m[(st, q)][1].add(n.split("(")[0].replace("void ", "")[:28])
for (st, q), (c, names) in sorted(m.items()):
    print("stream %3s -> queue %s : %5d dispatches  %s" % (st, q, c, sorted(names)[:4]))
gp = np.array([r[1] for r in rows if "k_gen_path" in r[0]]) / 1e6
print("frame starts (ms between k_gen_path launches):", np.round(np.diff(gp), 2).tolist())
