"""Writes the sidecar of a rocprofv3 kernel_stats.csv: <csv minus .csv>.sources.json = the hashes of the kernel source files at profiling
time and which file defines which kernel, so that bench.py quotes the profile's durations (`frac_rocprof`) only while a kernel's source
is unchanged.  usage: stats_sources.py <kernel_stats.csv>"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hashlib

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speech_recognition_amd", "csrc")


def source_of():
    ns = {}
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.py")).read()
    start = src.index("SOURCE_OF = {")
    end = src.index("}", start) + 1
    exec(src[start:end], ns)
    return ns["SOURCE_OF"]


so = source_of()
sources = {}
for f in sorted(set(so.values())):
    path = os.path.join(CSRC, f)
    if os.path.exists(path):
        sources[f] = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
out = sys.argv[1][:-4] + ".sources.json"
json.dump({"sources": sources, "source_of": so}, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out)
