"""usage: tools/timeline.py <rocprofv3 output dir> [marker-substring]  -- per-kernel timeline (start, gap to the
previous kernel, duration) of the LAST step in a kernel trace; a step starts at the last-but-(n-1) kernel whose name
contains the marker (default: scan_kernel)."""
import csv
import glob
import sys

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "scan_kernel<"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
f = sorted(glob.glob(d + "/*/*_kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a = idx[-back]
t0 = int(rows[a]["Start_Timestamp"])
prev = t0
busy = 0
for r in rows[a:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("tgx::", "")[:60]
    print("%9.3f  gap %7.3f  dur %7.3f  %s" % ((s - t0) / 1e6, (s - prev) / 1e6, (e - s) / 1e6, name))
    prev = e
    busy += e - s
print("span %.3f ms, busy %.3f ms" % ((prev - t0) / 1e6, busy / 1e6))
