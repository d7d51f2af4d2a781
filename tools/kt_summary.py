"""Print a rocprofv3 kernel_stats.csv with short kernel names: calls, total ms (divided by argv[2] runs), average us."""
import csv
import re
import sys

div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    m = re.search(r"(k_[A-Za-z0-9_]+(<[^>(]*>)?|__amd_rocclr_[A-Za-z]+|rocprim[A-Za-z_:0-9]*)", n)
    short = m.group(1) if m else n[:50]
    ms = int(r["TotalDurationNs"]) / 1e6 / div
    tot += ms
    print("%-44s %7d calls %9.3f ms  avg %9.1f us" % (short[:44], int(r["Calls"]), ms, float(r["AverageNs"]) / 1e3))
print("sum of kernel time: %.3f ms" % tot)
