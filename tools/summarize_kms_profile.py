"""Summarise gpurun_out/<tag>_* (tools/profile_kms.sh: rocprofv3 passes of tools/kms_bench.py) into profiles/<tag>_summary.md: kernel table of a
mk_gate_nand_new batch and, for kms_tlev_rotate_kernel and pm_mac_kernel, the fractions of the FP64 issue rate, the LDS array and HBM that
their FULL-BATCH launches reach (per-launch counters of the largest launches / their kernel-trace duration).
    python tools/summarize_kms_profile.py <tag>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from summarize_profile import CLK_HZ, CUS, FP64_PEAK_GINST, HBM_PEAK_BPS, SIMDS_PER_CU  # noqa: E402

tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out")


def newest(pattern):
    r = glob.glob(os.path.join(G, pattern), recursive=True)
    return max(r, key=os.path.getmtime) if r else None


lines = [f"# rocprofv3 summary `{tag}` -- `python3 tools/kms_bench.py` (KMS scheme, mk_gate_nand_new, 1x MI355X)\n"]
tj = os.path.join(G, f"{tag}_trace.json")
if os.path.exists(tj):
    lines.append("bench line of the traced run: `" + open(tj).read().strip().splitlines()[-1] + "`\n")
ks = newest(f"{tag}_trace/**/*kernel_stats.csv")
if ks:
    lines.append("## kernel trace (`rocprofv3 --kernel-trace --stats`; warm-up, one-party probe, the gate batch and the fast_boot batch together)\n")
    lines.append("| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|")
    for row in csv.DictReader(open(ks)):
        lines.append(f"| `{row['Name'][:80]}` | {row['Calls']} | {float(row['AverageNs'])/1e6:.4f} | {float(row['TotalDurationNs'])/1e6:.2f} | {float(row['Percentage']):.2f} |")
    lines.append("")
KEYS = ("kms_tlev_rotate_pair_kernel", "kms_tlev_rotate_kernel", "pm_mac_kernel")   # two jobs per workgroup (launches above 256 jobs) / one job / products
# per-dispatch durations of the kernels of interest: the full-batch launches are the longest ones
dur = defaultdict(list)
kt = newest(f"{tag}_trace/**/*kernel_trace.csv")
if kt:
    for row in csv.DictReader(open(kt)):
        for key in KEYS:
            if key in row["Kernel_Name"]:
                dur[key].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-9)
ctr = defaultdict(lambda: defaultdict(list))
for sub in ("fetch", "sq", "lds", "f64"):
    f = newest(f"{tag}_{sub}/**/*counter_collection.csv")
    if not f:
        continue
    for row in csv.DictReader(open(f)):
        for key in KEYS:
            if key in row["Kernel_Name"]:
                ctr[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                if "Scratch_Size" in row and row["Scratch_Size"] != "":
                    ctr[key]["_scratch"].append(float(row["Scratch_Size"]))
                if "VGPR_Count" in row and row["VGPR_Count"] != "":
                    ctr[key]["_vgpr"].append(float(row["VGPR_Count"]))


def top(vals, k):
    v = sorted(vals)[-k:]
    return sum(v) / len(v)


out = {}
for key, k in (("kms_tlev_rotate_pair_kernel", 2), ("kms_tlev_rotate_kernel", 2), ("pm_mac_kernel", 4)):
    if not dur[key] or not ctr[key]:
        continue
    t = top(dur[key], k)
    c = {n: top(v, k) for n, v in ctr[key].items()}
    f64 = c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_FMA_F64", 0)
    d = dict(launch_ms=t * 1e3, fp64_issue_frac=f64 / t / 1e9 / FP64_PEAK_GINST,
             valu_busy=c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (CUS * SIMDS_PER_CU * CLK_HZ * t),
             lds_busy=c.get("SQ_LDS_IDX_ACTIVE", 0) / (CUS * CLK_HZ * t),
             hbm_frac=2 * c.get("FETCH_SIZE", 0) * 1024 / t / HBM_PEAK_BPS, hbm_gb_per_launch=2 * c.get("FETCH_SIZE", 0) * 1024 / 1e9,
             wait_any_frac=c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
             vector_insts=c.get("SQ_INSTS_VALU"), fp64_insts=f64, lds_insts=c.get("SQ_INSTS_LDS"), bank_conflicts=c.get("SQ_LDS_BANK_CONFLICT"),
             scratch_bytes_per_lane=c.get("_scratch"), vgprs=c.get("_vgpr"), launches_averaged=k)
    out[key] = d
    lines.append(f"## `{key}`: the {k} largest launches{' (one per party)' if 'tlev' in key else ''}\n")
    for n, v in d.items():
        if v is not None:
            lines.append(f"* {n} = {v:.6g}")
    lines.append("")
lines.append("FETCH_SIZE doubled per MI355X_MICROARCH.md (section HBM); FP64 issue peak = 614.4 G wave-instructions / s (256 CUs x 4 SIMDs x 2.4 GHz / 4).\n")
P = os.path.join(ROOT, "profiles")
open(os.path.join(P, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join(P, f"{tag}_counters.json"), "w"), indent=1)
print("\n".join(lines))
