#!/usr/bin/env python3
"""Timeline of ONE HIP-graph-replayed D-step from a rocprofv3 --kernel-trace CSV: start offset, duration, queue and grid of every
dispatch, idle gaps of the device and the time each kernel name holds.  usage: step_timeline.py kernel_trace.csv [marker-substring]
(the marker is a kernel that runs once per step; default: multi-tensor Adam)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "adam"
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?")))
ev.sort()
marks = [i for i, e in enumerate(ev) if marker.lower() in e[2].lower()]
if len(marks) < 3:
    sys.exit("marker %r seen %d times" % (marker, len(marks)))
a, b = marks[-3] + 1, marks[-2] + 1          # the second-to-last complete step
step = ev[a:b]
t0 = step[0][0]
print("# step: %d dispatches, %.3f ms from first start to last end" % (len(step), (max(e[1] for e in step) - t0) / 1e6))
busy_until = t0
idle = 0
for s, e, name, q, g, w in step:
    if s > busy_until:
        idle += s - busy_until
    busy_until = max(busy_until, e)
    short = name.replace("void ", "").replace("cslgan::", "")[:70]
    print("%9.1f us  +%8.1f us  q%-3s grid %-8s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, g, short))
print("# device idle (no kernel running) inside the step: %.3f ms" % (idle / 1e6))
by = collections.defaultdict(float)
for s, e, name, q, g, w in step:
    by[name.replace("void ", "").replace("cslgan::", "")[:60]] += (e - s) / 1e3
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:25]:
    print("# %9.1f us  %s" % (v, k))
