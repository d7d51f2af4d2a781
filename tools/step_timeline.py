"""Print the kernel timeline of the last complete bench step from a rocprofv3 kernel trace CSV
(rocprofv3 --kernel-trace --output-format csv ... -- python bench.py): start offset, gap to the previous
kernel's end, duration, grid, name."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_flood_clear" in r["Kernel_Name"]]
i0, i1 = idx[-2], idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
prev = t0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f +%5.1f dur %6.1f  grid %8s  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3,
                                                  r.get("Grid_Size_X", r.get("Grid_Size", "")), r["Kernel_Name"][:60]))
    prev = e
