#!/usr/bin/env python3
"""Compact copy of a rocprofv3 kernel trace for offline analysis: npz of (start, end, stream, name id) + the name table.
    python tools/timeline_dump.py <results.db> <out.npz>"""
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
lane = next((c for c in ("stream_id", "queue_id") if c in cols), None)
rows = db.execute("select start, end, name, %s from kernels order by start" % (lane or "0")).fetchall()
names = sorted({r[2].split("(")[0].replace("void ", "") for r in rows})
nid = {n: i for i, n in enumerate(names)}
np.savez_compressed(sys.argv[2], start=np.array([r[0] for r in rows], np.int64), end=np.array([r[1] for r in rows], np.int64),
                    stream=np.array([r[3] for r in rows], np.int64),
                    name=np.array([nid[r[2].split("(")[0].replace("void ", "")] for r in rows], np.int32), names=np.array(names))
