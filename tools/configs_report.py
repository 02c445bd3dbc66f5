#!/usr/bin/env python3
"""Condense a tools/make_profiles_configs.sh output directory into REPORT.md (stdout) + pmc_configs.json: per entry of
bench.py's "configs" array the kernel(s) it runs, their rocprofv3 average duration, HBM bytes and FP64 flop per launch."""
import collections, csv, glob, json, sys
from pathlib import Path
d = sys.argv[1]
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
# bench.py configs[] workload prefix -> substrings identifying its kernels (a config's launch = the sum over them)
CONFIGS = [
    # k_tet4_ev<EXP_MODE, MINW, ABL, GEN>: GEN = false is the shipped parameter pattern (16 moments), true = every term (22 moments)
    # (k_tet4_evq<EXP_MODE, TL, GEN>: the resident form of large launches)
    ("cfg2: PIHNA TET4 K(55)", ["k_tet4_ev<3, 3, 0, false>(rdc::HostPrepEv", "k_tet4_ev<0, 3, 0, false>(rdc::HostPrepEv"], 998250),
    ("PIHNA TET4 K(119), all transport terms", ["k_tet4_ev<3, 2, 0, true>(rdc::HostPrepEv", "k_tet4_evq<3, false, true>(rdc::HostPrepEv", "k_tet4_rg5<rdc::Pihna,"], 10110954),
    ("cfg3: RIPF TET4 K(94), params run/RIPF133/input.dat (shipped)", ["k_tet4_rg5<rdc::RipfReduced", "k_tet4_evc<rdc::RipfReduced"], 4983504),
    ("cfg3: RIPF TET4 K(94), params run/RIPF133/input.dat (full)", ["k_tet4_evc<rdc::Ripf,", "k_tet4_rg5<rdc::Ripf,"], 4983504),
    ("cfg5 (RD half): HCC HEX8 H(126)", ["k_hex8_cl<rdc::Hcc,", "k_hex8_clp<rdc::Hcc,"], 2000376),
    ("cfg5 (RD half, shipped)", ["k_hex8_cl<rdc::HccMassOnly"], 2000376),
    ("cfg5 (solid half)", ["k_solid_cl<", "k_solid_sides"], 2000376),
]
stats = {}
for f in glob.glob(d + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        stats[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/pmc*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        key = (r["Dispatch_Id"], r["Kernel_Name"])
        per[key][r["Counter_Name"]] = per[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for (did, name), c in per.items():
        for k, v in c.items():
            agg[name][k].append(v)
try:
    unprof = {c["workload"]: c for c in json.loads(open(d + "/configs.json").read().strip().splitlines()[-1])["configs"]}
except Exception as e:
    unprof = {}
from rdcfes_amd import build as B
out = {"source_hash": B.source_hash(),
       "note": "per launch, from tools/make_profiles_configs.sh: hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE half-count "
               "correction), fp64_flop = 64*(2*FMA + MUL + ADD + TRANS) wave instructions; a configuration = the sum over its kernels",
       "kernels": {}}
print(f"# rocprofv3 evidence for bench.py's `configs` array ({d})\n")
print("Command profiled: `python3 bench.py --configs-only 1` (3 warm-ups + 6-10 timed launches per configuration).\n")
print("| configuration | kernel(s) | calls | rocprof avg ms | HIP-event ms (un-profiled run) | HBM GB / launch (counters) | algorithmic GB | FP64 GFLOP / launch | LDS instr / launch | LDS bank-conflict cycles |")
print("|---|---|---|---|---|---|---|---|---|---|")
for key, subs, ne in CONFIGS:
    names = [n for n in stats if any(s in n for s in subs)]
    if key.startswith("cfg2"):   # the K(55) launches of k_tet4_ev only
        pass
    if not names:
        continue
    avg_ms = sum(stats[n][1] for n in names) / 1e6
    calls = min(stats[n][0] for n in names)
    m = collections.defaultdict(float)
    for n in names:
        for k, v in agg.get(n, {}).items():
            m[k] += sum(v) / len(v)
    hbm = (2.0 * m.get("FETCH_SIZE", 0.0) + m.get("WRITE_SIZE", 0.0)) * 1024.0 if "FETCH_SIZE" in m and "WRITE_SIZE" in m else None
    flop = 64.0 * (m.get("SQ_INSTS_VALU_ADD_F64", 0.0) + m.get("SQ_INSTS_VALU_MUL_F64", 0.0) + 2.0 * m.get("SQ_INSTS_VALU_FMA_F64", 0.0) +
                   m.get("SQ_INSTS_VALU_TRANS_F64", 0.0)) if "SQ_INSTS_VALU_FMA_F64" in m else None
    u = next((c for w, c in unprof.items() if w.startswith(key)), None)
    short = " + ".join(n.split("(")[0].replace("void rdc::", "")[:60] for n in names)
    print(f"| {key} | `{short}` | {calls} | {avg_ms:.3f} | {u['kernel_ms']:.3f} |" if u else f"| {key} | `{short}` | {calls} | {avg_ms:.3f} | - |", end="")
    print(f" {hbm / 1e9:.3f} |" if hbm else " - |", end="")
    print(f" {u['algorithmic_bytes_per_launch'] / 1e9:.3f} |" if u else " - |", end="")
    print(f" {flop / 1e9:.1f} |" if flop else " - |", end="")
    print(f" {m.get('SQ_INSTS_LDS', 0):.4g} | {m.get('SQ_LDS_BANK_CONFLICT', 0):.4g} |")
    if hbm and flop:
        out["kernels"][key] = {"kernel": short, "hbm_bytes_per_launch": hbm, "fp64_flop_per_launch": flop, "rocprof_avg_ms": avg_ms,
                               "lds_instructions_per_launch": m.get("SQ_INSTS_LDS"), "lds_bank_conflict_cycles": m.get("SQ_LDS_BANK_CONFLICT"),
                               "lds_idx_active_cycles": m.get("SQ_LDS_IDX_ACTIVE")}
print("\nNOTE cfg2 and the headline share the kernel `k_tet4_ev`: this command only launches it on K(55).\n")
print("## kernel-trace --stats\n\n| kernel | calls | average ns |\n|---|---|---|")
for n, (c, a, t) in sorted(stats.items(), key=lambda kv: -kv[1][2])[:16]:
    print(f"| `{n[:110]}` | {c} | {a:.0f} |")
json.dump(out, open(d + "/pmc_configs.json", "w"), indent=1)
print("\n## pmc_configs.json\n\n```json\n" + json.dumps(out, indent=1) + "\n```")
