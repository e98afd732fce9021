"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs: python tools/pmc_kernel.py <dir> [<dir> ...] [--match blend]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = "blend"
    for a in sys.argv[1:]:
        if a.startswith("--match="):
            match = a.split("=", 1)[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    for d in args:
        for path in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            seen = set()
            for r in csv.DictReader(open(path)):
                k = r["Kernel_Name"].replace("void ", "").replace("gsr::", "").split("(")[0]
                if match not in k:
                    continue
                a = acc[k][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
                key = (path, r["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    dur[k][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                    dur[k][1] += 1
    for k in sorted(acc):
        print(f"== {k}: avg {dur[k][0] / max(dur[k][1], 1) / 1e3:.1f} us under PMC")
        for c in sorted(acc[k]):
            v, n = acc[k][c]
            print(f"   {c:28s} {v / max(n, 1):16.1f}")


if __name__ == "__main__":
    main()
