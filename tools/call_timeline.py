#!/usr/bin/env python3
"""Device timeline of ONE drop-in call from a rocprofv3 run with --kernel-trace --memory-copy-trace (csv):
usage: call_timeline.py DIR [index of the call from the end, default 3].  Prints every copy and kernel of that call with
start / end relative to the call's first device operation."""
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
# a call ends with the device-to-host copies: split at k_gather launches
starts = [i for i, e in enumerate(ev) if "k_gather" in e[2]]
i0 = starts[-back]
# walk back to the first copy of this call (the H2D copies in front of k_gather)
j = i0
while j > 0 and ev[j - 1][2].startswith("C") and "DEVICE_TO_HOST" not in ev[j - 1][2].upper() and "DtoH" not in ev[j - 1][2]: j -= 1
end = starts[-back + 1] if back > 1 else len(ev)
k = end
while k > j and ev[k - 1][2].startswith("C") and not ("DEVICE_TO_HOST" in ev[k - 1][2].upper() or "DtoH" in ev[k - 1][2]): k -= 1
t0 = ev[j][0]
for s, e, n in ev[j:k]:
    print("%8.1f %8.1f  (%6.1f)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
