#!/usr/bin/env python3
"""Run __graft_entry__.smoke() N times in one process; print every outcome (flakiness hunt)."""
import os, sys, io, contextlib, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bad = 0
for i in range(n):
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            entry.smoke()
        print("run %d ok   %s" % (i, " | ".join(l for l in buf.getvalue().splitlines() if l.startswith("smoke:"))[-260:]), flush=True)
    except Exception as e:
        bad += 1
        print("run %d FAIL %s: %s\n   %s" % (i, type(e).__name__, str(e)[:300], buf.getvalue().strip()[-400:]), flush=True)
print("failures %d / %d" % (bad, n))
