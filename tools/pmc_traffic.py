"""Per-kernel memory-side traffic from separate rocprofv3 FETCH_SIZE / WRITE_SIZE passes: for each prefix <p> given, reads
<p>_fetch/*/*_counter_collection.csv and <p>_write/...; prints average bytes per launch of the gsr kernels (FETCH_SIZE doubled: the
gfx950 correction of MI355X_MICROARCH.md; both counters are in KiB)."""
import csv
import glob
import sys
from collections import defaultdict


def short(name):
    return name.replace("void ", "").replace("gsr::", "").split("(")[0]


def per_kernel(prefix, part, counter):
    acc = defaultdict(lambda: [0.0, 0, 0.0])
    for path in glob.glob(f"{prefix}_{part}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"]) * 1024.0
            a[1] += 1
            a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return {k: (v[0] / v[1], v[2] / v[1] / 1e3) for k, v in acc.items() if v[1]}


def main(prefixes):
    for p in prefixes:
        f, w = per_kernel(p, "fetch", "FETCH_SIZE"), per_kernel(p, "write", "WRITE_SIZE")
        print(f"== {p}")
        for k in sorted(f, key=lambda k: -f[k][0]):
            if not any(s in k for s in ("blend", "preprocess", "bucket", "sort", "hist")):
                continue
            wb = w.get(k, (0.0, 0.0))[0]
            print(f"  {k:60s} 2xFETCH {2 * f[k][0] / 1e6:8.1f} MB   WRITE {wb / 1e6:8.1f} MB   avg {f[k][1]:7.1f} us (under pmc)")


if __name__ == "__main__":
    main(sys.argv[1:])
