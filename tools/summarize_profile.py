"""Summarise gpurun_out/<tag>_* rocprofv3 outputs into profiles/<tag>_summary.md (+ copies of the small CSVs)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
workload, batch, alg_mb = "4096 NAND / step, SK-128", 4096, 61.9
try:   # the bench line of the traced run names the workload
    _j = json.loads(open(os.path.join(G, f"{tag}_trace.json")).read().strip().splitlines()[-1])
    batch = _j["config"]["gates_per_gpu_per_step"]
    workload = f"{batch} NAND / step, {_j['config']['param_set']}"
    alg_mb = _j["roofline"]["algorithmic_bytes_per_launch"] / batch / 1e6
except Exception:
    pass
lines = [f"# rocprofv3 summary `{tag}` — `python3 bench.py --no-cpu-baseline ...` ({workload}, 1x MI355X)\n"]


def find(pattern):
    r = glob.glob(os.path.join(G, pattern), recursive=True)
    return r[0] if r else None


ks = find(f"{tag}_trace/**/*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(P, f"{tag}_kernel_stats.csv"))
    lines.append("## kernel trace (`rocprofv3 --kernel-trace --stats`)\n")
    lines.append("| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|")
    for row in csv.DictReader(open(ks)):
        lines.append(f"| `{row['Name'][:90]}` | {row['Calls']} | {float(row['AverageNs'])/1e6:.4f} | {float(row['TotalDurationNs'])/1e6:.2f} | {float(row['Percentage']):.2f} |")
    lines.append("")
tj = os.path.join(G, f"{tag}_trace.json")
if os.path.exists(tj):
    try:
        j = json.loads(open(tj).read().strip().splitlines()[-1])
        lines.append(f"bench line of the traced run: value = {j['value']:.0f} gates/s, blind-rotate avg launch {j['roofline']['avg_launch_ms']:.3f} ms (HIP events), "
                     f"roofline.frac = {j['roofline']['frac']:.3f}\n")
    except Exception as e:
        lines.append(f"(could not parse {tj}: {e})\n")


def pmc(sub):
    f = find(f"{tag}_{sub}/**/*counter_collection.csv")
    if not f:
        return {}
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


for sub, title in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("sq", "SQ counters"), ("lds", "LDS / VMEM counters"), ("tcc", "L2 hit/miss")):
    acc = pmc(sub)
    if not acc:
        continue
    lines.append(f"## PMC pass: {title}\n")
    lines.append("| kernel | counter | per-launch mean (largest launches) | launches |\n|---|---|---|---|")
    for k, ctrs in acc.items():
        if "blind_rotate" not in k and "keyswitch" not in k:
            continue
        for c, vals in ctrs.items():
            big = sorted(vals)[-max(1, len(vals) // 2):]   # the full-batch launches (warm-up + timed), not tiny ones
            lines.append(f"| `{k[:60]}` | {c} | {sum(big)/len(big):.6g} | {len(vals)} |")
    lines.append("")
fetch, write = pmc("fetch"), pmc("write")
for k in fetch:
    if "blind_rotate" in k:
        fv = sorted(fetch[k]["FETCH_SIZE"])[-3:]
        wv = sorted(write.get(k, {}).get("WRITE_SIZE", [0]))[-3:]
        f_kb, w_kb = sum(fv) / len(fv), sum(wv) / len(wv)
        traffic = (2 * f_kb + w_kb) * 1024
        lines.append(f"**HBM traffic of one blind-rotate launch ({batch} gates)**: FETCH_SIZE = {f_kb:.0f} KB (x2 gfx950 correction for 16-B/lane "
                     f"coalesced reads -> {2*f_kb*1024/1e6:.1f} MB), WRITE_SIZE = {w_kb:.0f} KB -> **{traffic/1e6:.1f} MB per launch** "
                     f"= {traffic/batch/1e3:.1f} KB per gate (algorithmic: {alg_mb:.1f} MB per gate; the key stays in L2 / Infinity Cache).\n")
open(os.path.join(P, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
