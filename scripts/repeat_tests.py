#!/usr/bin/env python3
"""Run a pytest selection N times in ONE process and print every failure's assertion line (flakiness hunt)."""
import sys, io, contextlib
import pytest

n = int(sys.argv[1])
args = sys.argv[2:]
fails = 0
for i in range(n):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = pytest.main(["-q", "-x", "-p", "no:cacheprovider"] + args)
    out = buf.getvalue()
    if rc != 0:
        fails += 1
        lines = [l for l in out.splitlines() if l.startswith(("E  ", "FAILED"))]
        print("run %d FAILED:\n  %s" % (i, "\n  ".join(lines[:8])), flush=True)
    else:
        print("run %d ok (%s)" % (i, out.strip().splitlines()[-1]), flush=True)
print("failures: %d / %d" % (fails, n))
