#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE (KB, rocprofv3 --pmc, separate passes) per launch of every q4_gemv_kernel instantiation -> corrected HBM bytes per launch
(gfx950: FETCH_SIZE reports half of the bytes of wide coalesced streaming reads, MI355X_MICROARCH.md HBM section); last line = JSON for
profiles/pmc_traffic.json.   python tools/traffic_summary.py <fetch_dir> <write_dir>"""
import collections, csv, glob, json, os, sys
def load(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and ("gemv" in r.get("Kernel_Name", "") or "decode_engine" in r.get("Kernel_Name", "")):
                k = r["Kernel_Name"].replace("void vox::", "").replace("(anonymous namespace)::", ""); k = k[:k.index("(")] if "(" in k else k
                a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}
fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(fe):
    f, n = fe[k]; w = wr.get(k, (0.0, 0))[0]
    print(f"{k}: launches {n}  FETCH_SIZE avg {f:.1f} KB  WRITE_SIZE avg {w:.1f} KB  -> HBM bytes per launch (2 x fetch + write) {int(2 * f * 1024 + w * 1024)}")
    out[k] = int(2 * f * 1024 + w * 1024)
print(json.dumps(out))
