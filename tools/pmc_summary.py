#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes (one directory per pass): python tools/pmc_summary.py dir1 dir2 ...
Prints, for the kernels that dominate, every collected counter averaged per dispatch, plus derived figures."""
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "?"); c = r.get("Counter_Name"); v = float(r.get("Counter_Value", 0) or 0)
            a = acc[k][c]; a[0] += v; a[1] += 1
def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void vox::", "").replace("vox::", "")
    return k[:k.index("(")] if "(" in k else k
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_BUSY_CYCLES", [0, 1]))[0])
for k, cs in rows[:int(os.environ.get("VOX_PMC_TOP", "22"))]:
    m = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
    n = max(v[1] for v in cs.values())
    print(f"== {short(k)}   dispatches {n}")
    print("   " + "  ".join(f"{c}={m[c]:.4g}" for c in sorted(m)))
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        extra = []
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
            if c in m: extra.append(f"{c}/WAVE_CYCLES={m[c] / wc:.2f}")
        if "SQ_WAVES" in m: extra.append(f"wave_cycles_per_wave={4 * wc / m['SQ_WAVES']:.0f} (x4: quad-cycles)")
        if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m: extra.append(f"valu_per_wave={m['SQ_INSTS_VALU'] / m['SQ_WAVES']:.0f} mfma_per_wave={m.get('SQ_INSTS_MFMA', 0) / m['SQ_WAVES']:.0f}")
        print("   " + "  ".join(extra))
    if "TCC_HIT_sum" in m:
        print(f"   L2 hit rate {m['TCC_HIT_sum'] / max(m['TCC_HIT_sum'] + m.get('TCC_MISS_sum', 0), 1):.3f}")
    if "WRITE_SIZE" in m:
        print(f"   WRITE_SIZE = {m['WRITE_SIZE'] / 1e3:.3f} MB per dispatch (KB units, uncalibrated on gfx950)")
    if "FETCH_SIZE" in m:
        print(f"   FETCH_SIZE x2 (gfx950 correction) = {2 * m['FETCH_SIZE'] / 1e3:.3f} MB per dispatch (FETCH_SIZE is in KB)")
    if "TCP_TCC_READ_REQ_LATENCY_sum" in m and m.get("TCP_TCC_READ_REQ_sum"):
        print(f"   mean L1->L2 read latency {m['TCP_TCC_READ_REQ_LATENCY_sum'] / m['TCP_TCC_READ_REQ_sum']:.0f} cycles")
