"""Condense the rocprofv3 outputs of tools/profile_round.sh into two small CSVs for profiles/:
  <tag>_kernel_stats.csv   per-kernel calls / total / average duration (copy of rocprofv3's --stats table, gsr kernels first)
  <tag>_pmc.csv            per-kernel averages per launch of every collected counter; FETCH_SIZE / WRITE_SIZE are in KiB in
                           rocprofv3's units and are converted to bytes; `fetch_bytes_x2` applies the gfx950 correction of
                           MI355X_MICROARCH.md (FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("gsr::", "")
    return name.split("(")[0]


def main(tag, src="gpurun_out", dst="profiles"):
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, f"{tag}_stats", "*", "*_kernel_stats.csv"))
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
        with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], f"{float(r['TotalDurationNs']) / 1e3:.1f}",
                            f"{float(r['AverageNs']) / 1e3:.2f}", r["Percentage"]])
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    for part in ("fetch", "write", "sq", "lds"):
        for path in glob.glob(os.path.join(src, f"{tag}_pmc_{part}", "*", "*_counter_collection.csv")):
            seen = set()
            for r in csv.DictReader(open(path)):
                k = short(r["Kernel_Name"])
                a = acc[k][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
                if part == "sq" and r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    d = dur[k]
                    d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                    d[1] += 1
    counters = sorted({c for k in acc for c in acc[k]})
    with open(os.path.join(dst, f"{tag}_pmc.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "avg_us_under_pmc"] + [c + "_per_launch" for c in counters] +
                   ["fetch_bytes", "fetch_bytes_x2", "write_bytes"])
        for k in sorted(acc, key=lambda k: -dur[k][0]):
            per = {c: (acc[k][c][0] / acc[k][c][1] if acc[k][c][1] else 0.0) for c in counters}
            fb, wb = per.get("FETCH_SIZE", 0.0) * 1024.0, per.get("WRITE_SIZE", 0.0) * 1024.0
            n = max(v[1] for v in acc[k].values())
            w.writerow([k, n, f"{dur[k][0] / max(dur[k][1], 1) / 1e3:.2f}"] + [f"{per[c]:.1f}" for c in counters] +
                       [f"{fb:.0f}", f"{2 * fb:.0f}", f"{wb:.0f}"])
    print("wrote", dst)


if __name__ == "__main__":
    main(*sys.argv[1:])
