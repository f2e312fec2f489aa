"""Condense a gpurun_out/<dir> profile collection (scripts/collect_profiles.sh) into the small
files committed under profiles/: per-kernel rocprofv3 stats, per-kernel HBM traffic from the
FETCH_SIZE / WRITE_SIZE PMC passes (gfx950 correction: FETCH_SIZE counts 128-B requests as
64 B, so read bytes = 2 x FETCH_SIZE; both counters are in KiB -- MI355X_MICROARCH.md, HBM),
and the bench JSON lines."""
import collections
import csv
import glob
import json
import os
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)

ks = glob.glob(os.path.join(src, "kt", "*", "*kernel_stats.csv"))
if ks:
    open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w").write(open(ks[0]).read())


def pmc(name):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, name, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            out[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return out


fetch, write = pmc("pmc_fetch"), pmc("pmc_write")
kernels = sorted({k for k, _ in fetch} | {k for k, _ in write})
rows = []
for k in kernels:
    if "pysdr" not in k:
        continue
    f = fetch.get((k, "FETCH_SIZE"), [])
    w = write.get((k, "WRITE_SIZE"), [])
    fm = sum(f) / len(f) if f else 0.0
    wm = sum(w) / len(w) if w else 0.0
    import re
    m = re.search(r"(\w+_kernel)", k)
    rows.append(dict(kernel=m.group(1) if m else k, launches_sampled=len(f),
                     FETCH_SIZE_KiB=fm, WRITE_SIZE_KiB=wm,
                     read_bytes=2 * fm * 1024, write_bytes=wm * 1024,
                     hbm_bytes_per_launch=2 * fm * 1024 + wm * 1024))
json.dump(rows, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
for name in ("bench_default.json", "bench_kt.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        open(os.path.join(dst, f"{tag}_{name}"), "w").write(open(p).read())
print(json.dumps(rows, indent=1))
