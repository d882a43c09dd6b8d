"""Print the kernel timeline of a few steps from a rocprofv3 --kernel-trace CSV.  Usage: python tools/timeline.py trace.csv ANCHOR [occurrence=10] [steps=2]
(ANCHOR = substring of the kernel that starts a step, e.g. k_zstd_encode<11 or k_fused_roles)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2]
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 10
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
if len(idx) <= occ + steps:
    occ = max(0, len(idx) - steps - 1)
i0, i1 = idx[occ], idx[occ + steps]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = None
for r in rows[max(0, i0 - 3):i1 + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:10.1f} {e/1e3:10.1f} {(e-s)/1e3:9.1f} us  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'].split('(')[0][-52:]}")
