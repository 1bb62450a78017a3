"""Summarises a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass into per-kernel MFMA utilisation.
usage: pmc_mfma_busy.py <counter_collection.csv> <out.json>"""
import collections, csv, json, re, sys

busy = collections.defaultdict(float)
act = collections.defaultdict(float)
n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void ", "")
    m = re.search(r"(?:\(anonymous namespace\)::)?(\w+)(?:<[^(]*>)?\(", name)
    k = m.group(1) if m else name[:60]
    if "gemm" not in k and "conv1" not in k:
        continue
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[k] += float(r["Counter_Value"])
        n[k] += 1
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        act[k] += float(r["Counter_Value"])
out = {k: {"launches": n[k], "mfma_busy_cycles": busy[k], "gui_active_cycles": act[k],
           "busy_per_simd": busy[k] / (act[k] * 128.0) if act[k] else None} for k in sorted(busy)}
json.dump({"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (own pass), " + (sys.argv[3] if len(sys.argv) > 3 else "bench.py --batch 1024") + "; "
                   "busy_per_simd = sum(busy over SIMDs) / (sum(active over XCDs) * 128 SIMDs per XCD)",
           "kernels": out}, open(sys.argv[2], "w"), indent=1, sort_keys=True)
for k, v in out.items():
    print(k, v)
