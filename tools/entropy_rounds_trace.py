#!/usr/bin/env python3
"""Prints the durations of the sync kernels of the last file in a rocprofv3 kernel trace directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
sync = [i for i, r in enumerate(rows) if "sync" in r["Kernel_Name"]]
starts = [i for i in sync if i == 0 or "sync" not in rows[i - 1]["Kernel_Name"]]
last = [r for r in rows[starts[-1]:] if "sync" in r["Kernel_Name"] or "write" in r["Kernel_Name"]]
print(" ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f}" for r in last))
