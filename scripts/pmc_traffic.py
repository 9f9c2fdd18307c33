#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs into profiles/*pmc_traffic.json.
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps_incl_warmup> <out.json>"""
import collections, csv, json, sys


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void cslgan::", "").replace("cslgan::", "")
        d[name][0] += 1
        d[name][1] += float(r["Counter_Value"])
    return d


f, w, steps, out = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3]), sys.argv[4]
d = {}
for k in set(f) | set(w):
    n = f.get(k, [0, 0])[0] or w.get(k, [0, 0])[0]
    fb, wb = f.get(k, [0, 0])[1] * 1024 * 2, w.get(k, [0, 0])[1] * 1024      # KB -> B; gfx950 FETCH_SIZE reads half of a wide stream
    d[k] = {"launches_per_step": n / steps, "fetch_MB_per_launch": fb / max(n, 1) / 1e6, "write_MB_per_launch": wb / max(n, 1) / 1e6,
            "MB_per_step": (fb + wb) / steps / 1e6}
d = dict(sorted(d.items(), key=lambda kv: -kv[1]["MB_per_step"])[:24])
kc = {k: v for k, v in d.items() if k.startswith(("igemm_kc", "igemm_halo", "igemm_skinny", "igemm_x3h"))}     # the conv2d_fwd / dgrad family
launches, mb = sum(v["launches_per_step"] for v in kc.values()), sum(v["MB_per_step"] for v in kc.values())
res = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) around `python3 bench.py --steps 3 --warmup 1 "
               "--no-cpu-baseline` on 1x MI355X; KB counters x1024; FETCH_SIZE doubled (gfx950 reports half of a wide coalesced stream, "
               "MI355X_MICROARCH.md HBM section); Infinity-Cache hits are counted, so this is an upper bound on HBM bytes",
       "step_total_MB_top24_kernels": sum(v["MB_per_step"] for v in d.values()),
       "conv_family": {"kernels": "igemm_kc + igemm_halo + igemm_x3h + igemm_skinny, all instantiations (conv2d_fwd / conv2d_dgrad entries)",
                       "launches_per_step": launches, "MB_per_step": mb, "MB_per_launch": mb / launches},
       "per_kernel": d}
json.dump(res, open(out, "w"), indent=1)
print("conv family: %.1f launches/step, %.0f MB/step, %.1f MB/launch; step total %.0f MB" % (launches, mb, mb / launches, res["step_total_MB_top24_kernels"]))
for k, v in list(d.items())[:6]:
    print("  %-50s fetch %.1f write %.1f MB/launch" % (k, v["fetch_MB_per_launch"], v["write_MB_per_launch"]))
