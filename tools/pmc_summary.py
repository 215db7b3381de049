#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter CSVs (FETCH_SIZE / WRITE_SIZE passes) for the fused kernel into
profiles/pmc_latest.json: HBM bytes per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md
prescribes (FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced
streaming read, so the read side is doubled).

    python tools/pmc_summary.py <dir-with-counter-csvs> [--kernel zj_fused_kernel] [--out profiles/pmc_latest.json]
"""
import argparse
import csv
import glob
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--kernel", default="zj_fused_kernel")
    ap.add_argument("--out", default=None)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    vals = {}
    files = glob.glob(os.path.join(a.dir, "**", "*counter_collection*.csv"), recursive=True)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if a.kernel not in row.get("Kernel_Name", ""):
                    continue
                name, v = row.get("Counter_Name"), row.get("Counter_Value")
                if name is None or v is None:
                    continue
                vals.setdefault(name, []).append(float(v))
    dbs = glob.glob(os.path.join(a.dir, "**", "*.db"), recursive=True)  # rocpd (sqlite) output format
    for f in dbs:
        import sqlite3
        try:
            cur = sqlite3.connect(f).cursor()
            for kname, cname, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
                if a.kernel in (kname or ""):
                    vals.setdefault(cname, []).append(float(v))
        except Exception as e:  # noqa: BLE001
            print("skip", f, e)
    files += dbs
    out = {"source": f"rocprofv3 --pmc, {len(files)} csv file(s) {a.tag}".strip(), "kernel": a.kernel, "counters": {}}
    for k, v in vals.items():
        out["counters"][k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
    f = out["counters"].get("FETCH_SIZE", {}).get("mean")
    w = out["counters"].get("WRITE_SIZE", {}).get("mean")
    if f is not None and w is not None:
        out["fetch_bytes_raw"] = f * 1024
        out["fetch_bytes_corrected"] = 2 * f * 1024  # gfx950: x2 for 16 B/lane streaming reads
        out["write_bytes"] = w * 1024
        out["hbm_bytes_per_launch"] = int(2 * f * 1024 + w * 1024)
        out["note"] = "FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section); WRITE_SIZE uncalibrated"
    out["workload"], out["frames_per_launch"] = "420-rgb", 16  # what bench.py launches by default
    sq = out["counters"].get("SQ_INSTS_VALU", {}).get("mean")
    if sq is not None:
        out["sq_insts_valu_per_launch"] = int(sq)
        out["sq_source"] = f"rocprofv3 --pmc SQ_INSTS_VALU {a.tag}".strip()
    print(json.dumps(out, indent=1))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
