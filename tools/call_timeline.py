#!/usr/bin/env python3
"""Device timeline of ONE drop-in call from a rocprofv3 run with --kernel-trace --memory-copy-trace (csv):
usage: call_timeline.py DIR [index of the call from the end, default 3].  A call = everything the device did after the previous
call's last kernel (k_root_gain*) up to and including this call's; times relative to the call's first device operation."""
import csv, glob, sys
d = sys.argv[1]; back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = []
for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:60]))
for fn in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy"))))
ev.sort()
ends = [i for i, e in enumerate(ev) if "k_root_gain" in e[2] or "k_gain_stream" in e[2]]
i1 = ends[-back]
i0 = ends[-back - 1] + 1
t0 = ev[i0][0]
print("# start  end  (duration) us  operation      [gap to the previous call's last kernel: %.1f us]" % ((ev[i0][0] - ev[i0 - 1][1]) / 1e3))
for s, e, n in ev[i0:i1 + 1]:
    print("%8.1f %8.1f  (%6.1f)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
