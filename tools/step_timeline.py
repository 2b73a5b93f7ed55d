#!/usr/bin/env python3
"""Timeline of ONE step of the default bench line from a rocprofv3 kernel trace: per dispatch start / end relative to the step's first
kernel, the gap to the previous END on the whole device, stream id.   python tools/step_timeline.py <kernel_trace.csv> [step index from the end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps start at the first level-1 resize after a k_search_init / k_bf2nn / k_assemble tail: split on the first k_resize_rows following a matcher kernel
starts = [i for i, r in enumerate(rows) if "k_resize_rows" in r["Kernel_Name"] and i > 0 and ("k_search_init" in rows[i - 1]["Kernel_Name"] or "k_bf2nn" in rows[i - 1]["Kernel_Name"] or "k_prev_matched" in rows[i - 1]["Kernel_Name"])]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = starts[-k - 1], starts[-k]
t0 = int(rows[a]["Start_Timestamp"]); last_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
    print("%-28s q%-3s start %8.1f us  dur %7.1f us  gap-after-prev-end %6.1f us" % (name, r.get("Queue_Id", "?"), (s - t0) / 1e3, (e - s) / 1e3, (s - last_end) / 1e3))
    last_end = max(last_end, e)
print("step: %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
