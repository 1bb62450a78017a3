"""Summarises the SQ counter passes of scripts/pmc_stft.sh (the shipped stft4_kernel alone, batch 1024 / 80 bands / 60
coefficients) into one JSON: counters per launch, instructions per frame quad by class, and the pipe shares the DESIGN.md
section-5 discussion quotes.  Units per /opt/skills/guides/MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
count quad-cycles (x 4 = shader cycles), SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_INSTS_* count wave instructions.
usage: pmc_stft.py <combined counter_collection rows> <out.json> [quads per launch = 25600]"""
import collections, csv, hashlib, json, os, sys

rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    rows[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in rows.items()}          # per launch (summed over XCDs / SEs by rocprofv3)
n = {k: len(v) for k, v in rows.items()}
quads = float(sys.argv[3]) if len(sys.argv) > 3 else 1024 * 25.0
g = lambda k: c.get(k, float("nan"))
wave_cyc = 4.0 * g("SQ_WAVE_CYCLES")
gui = g("GRBM_GUI_ACTIVE") / 8.0                             # per-XCD active cycles of the launch
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speech_recognition_amd", "csrc", "stft4.hip")
out = {
    "note": "rocprofv3 --pmc passes of `python3 scripts/bench_stft.py 1024,80,60` (stft4_kernel alone), averages per launch; "
            "quad-cycle counters x 4; shares are of the launch's wave-cycles unless stated",
    "source_sha256_16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16],
    "launches_seen": n, "counters_per_launch": c,
    "per_quad": {"quads_per_launch": quads,
                 "valu": g("SQ_INSTS_VALU") / quads, "mfma": g("SQ_INSTS_MFMA") / quads, "lds": g("SQ_INSTS_LDS") / quads,
                 "salu": g("SQ_INSTS_SALU") / quads, "smem": g("SQ_INSTS_SMEM") / quads,
                 "vmem_rd": g("SQ_INSTS_VMEM_RD") / quads, "vmem_wr": g("SQ_INSTS_VMEM_WR") / quads,
                 "wave_cycles": wave_cyc / quads},
    "shares_of_wave_cycles": {"issuing_any": 4.0 * g("SQ_ACTIVE_INST_ANY") / wave_cyc,
                              "waiting_to_issue (SQ_WAIT_INST_ANY)": 4.0 * g("SQ_WAIT_INST_ANY") / wave_cyc,
                              "parked (SQ_WAIT_ANY: s_waitcnt / barrier)": 4.0 * g("SQ_WAIT_ANY") / wave_cyc,
                              "lds_issue_stall (SQ_WAIT_INST_LDS)": 4.0 * g("SQ_WAIT_INST_LDS") / wave_cyc},
    "pipe_busy_of_simd_cycles": {
        # per SIMD: 1024 SIMDs x per-XCD active cycles
        "valu (SQ_ACTIVE_INST_VALU x 4 / (1024 x cycles))": 4.0 * g("SQ_ACTIVE_INST_VALU") / (1024.0 * gui),
        "mfma (SQ_VALU_MFMA_BUSY_CYCLES / (1024 x cycles))": g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * gui),
        "lds (SQ_LDS_IDX_ACTIVE / (256 CUs x cycles))": g("SQ_LDS_IDX_ACTIVE") / (256.0 * gui),
        "lds_bank_conflict_share_of_lds_cycles": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")},
    "launch_cycles_per_xcd": gui,
}
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print(json.dumps(out["per_quad"], indent=1))
print(json.dumps(out["shares_of_wave_cycles"], indent=1))
print(json.dumps(out["pipe_busy_of_simd_cycles"], indent=1))
