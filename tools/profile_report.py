#!/usr/bin/env python3
"""Condense a tools/make_profiles.sh output directory into REPORT.md + pmc_traffic.json."""
import csv, glob, json, sys, collections
d = sys.argv[1]
print(f"# rocprofv3 evidence ({d})\n")
print("Command profiled: `python3 bench.py --steps 10 --warmup 3 --cpu-baseline 0 --configs 0 --handback 0 --two-part 0` (PIHNA, K(119), 1 GPU).\n")
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
for f in glob.glob(d + "/stats/**/*kernel_stats.csv", recursive=True):
    print("## kernel-trace --stats\n\n| kernel | calls | total ns | average ns | % |\n|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        print(f"| `{r['Name'][:90]}` | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {r['Percentage']} |")
    print()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/pmc*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = per[(r["Dispatch_Id"], r["Kernel_Name"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for (did, name), c in per.items():
        for k, v in c.items():
            agg[name][k].append(v)
print("## counters (mean per dispatch)\n")
traffic = None
for name, c in agg.items():
    if "pack" in name or "fill" in name.lower():
        continue
    print(f"`{name[:100]}`\n")
    print("| counter | mean per launch |\n|---|---|")
    m = {k: sum(v) / len(v) for k, v in c.items()}
    for k in sorted(m):
        print(f"| {k} | {m[k]:.6g} |")
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m and ("rg" in name or "tet4" in name):
        # guide: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane)
        # coalesced reads -> doubled.  Reads here are 16-B-per-lane loads (pair records, node records).
        traffic = {"fetch_kib_raw": m["FETCH_SIZE"], "write_kib": m["WRITE_SIZE"],
                   "hbm_bytes_per_launch": (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0,
                   "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 FETCH_SIZE half-count correction applied",
                   "kernel": name[:120]}
        if "SQ_INSTS_VALU_FMA_F64" in m:
            # wave-instructions x 64 lanes; FMA = 2 flop; transcendental (v_rcp_f64, v_sqrt_f64) counted as 1
            traffic["fp64_flop_per_launch"] = 64.0 * (m.get("SQ_INSTS_VALU_ADD_F64", 0.0) + m.get("SQ_INSTS_VALU_MUL_F64", 0.0) +
                                                      2.0 * m["SQ_INSTS_VALU_FMA_F64"] + m.get("SQ_INSTS_VALU_TRANS_F64", 0.0))
            traffic["fp64_wave_instructions_per_launch"] = (m.get("SQ_INSTS_VALU_ADD_F64", 0.0) + m.get("SQ_INSTS_VALU_MUL_F64", 0.0) +
                                                            m["SQ_INSTS_VALU_FMA_F64"] + m.get("SQ_INSTS_VALU_TRANS_F64", 0.0))
    print()
try:
    b = json.loads(open(d + "/bench.json").read().strip().splitlines()[-1])
    print("## un-profiled bench line\n\n```json\n" + json.dumps(b) + "\n```\n")
    if traffic:
        from rdcfes_amd import build as B
        traffic.update({"workload": "K(119)", "n_gpus": 1, "source_hash": B.source_hash(), "kernel_ms_avg_unprofiled": b["roofline"]["kernel_ms_avg"],
                        "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"]})
except Exception as e:
    print(f"(no bench.json: {e})")
if traffic:
    json.dump(traffic, open(d + "/pmc_traffic.json", "w"), indent=1)
    print("## HBM traffic per launch\n\n```json\n" + json.dumps(traffic, indent=1) + "\n```")
