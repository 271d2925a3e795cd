#!/usr/bin/env python3
"""tools/trace_gaps.py <kernel_trace.csv> [tail_count]: for the last `tail_count` dispatches of a rocprofv3 kernel trace (default:
all), the number of dispatches, the sum of the kernel durations, the wall span from the first start to the last end, and the idle
share -- what a launch-bound sequence of small kernels spends between kernels rather than in them."""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
if len(sys.argv) > 2:
    rows = rows[-int(sys.argv[2]):]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = sorted(rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1))
print(f"{len(rows)} dispatches: kernels {busy / 1e3:.1f} us, span {span / 1e3:.1f} us, idle {100 * (1 - busy / span):.1f} %, "
      f"median gap {gaps[len(gaps) // 2] / 1e3:.2f} us, median kernel {sorted(e - s for s, e, _ in rows)[len(rows) // 2] / 1e3:.2f} us")
by = {}
for s, e, k in rows:
    k = k.split("(")[0].replace("void tfk::", "")[:70]
    by.setdefault(k, [0, 0])
    by[k][0] += 1
    by[k][1] += e - s
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {c:5d} x {t / c / 1e3:8.2f} us  {k}")
