#!/usr/bin/env python3
"""Dev tool: from a rocprofv3 --kernel-trace CSV, the timeline of the LAST update epoch of the 512-wide nets (kernel, start offset,
duration, gap to the kernel before) -- where an epoch's wall time goes besides its kernels.
usage: python tools/epoch_gaps.py <kernel_trace.csv> [name-substring of the epoch's first kernel]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "resmlp_fwd<16>"
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
# the last complete epoch: from the second-to-last `first` kernel that is followed by a reduce
starts = [i for i in idx if any("resmlp_reduce" in rows[j]["Kernel_Name"] for j in range(i, min(i + 8, len(rows))))]
i0 = starts[-2]
i1 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][:70]:70s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f} us")
    prev_end = e
print(f"epoch wall (first start -> next epoch's first start): {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us")
