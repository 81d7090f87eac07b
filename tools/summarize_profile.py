"""Summarise gpurun_out/<tag>_* rocprofv3 outputs (tools/profile_bench.sh) into
    <tag>_summary.md      human-readable tables
    <tag>_counters.json   per-launch counters of the dominant (blind-rotate) kernel + the sha256 of the kernel sources they were
                          measured on -- the file bench.py reads for roofline.traffic / hbm_measured_frac / fp64_issue_frac / lds_busy
    <tag>_kernel_stats.csv copy of rocprofv3's kernel-trace statistics

    python tools/summarize_profile.py <tag>          # in the build container: gpurun_out/ -> profiles/ (the judged copies)
    python tools/summarize_profile.py <tag> --here   # on the GPU box: write next to the raw data in gpurun_out/
Every number in the bench line's `roofline` object is derived from <tag>_counters.json and the live launch time by
`derive()` below (bench.py imports it), so it can be re-derived from profiles/ by this script alone.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_hash import kernel_source_hash  # noqa: E402

# MI355X constants (/opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters)
CUS, SIMDS_PER_CU, CLK_HZ, HBM_PEAK_BPS = 256, 4, 2.4e9, 8.0e12
FP64_CYCLES_PER_WAVE_INST = 4          # 16 FP64 lanes per SIMD per clock: a wave64 FP64 instruction holds its SIMD for 4 cycles
FP64_PEAK_TFLOPS = CUS * SIMDS_PER_CU * 16 * 2 * CLK_HZ / 1e12   # 78.6 (vector FP64 FMA peak)
FP64_PEAK_GINST = CUS * SIMDS_PER_CU * CLK_HZ / FP64_CYCLES_PER_WAVE_INST / 1e9   # 614.4 G wave-instructions / s


def derive(counters, launch_seconds):
    """Roofline fractions of one launch of the dominant kernel from its per-launch counters and its duration.
    Returns a dict; entries are None where the counter pass is missing."""
    c = counters["per_launch"]
    out = {}
    hbm = c.get("hbm_bytes")
    out["traffic"] = hbm
    out["hbm_measured_frac"] = hbm / launch_seconds / HBM_PEAK_BPS if hbm is not None else None
    f64 = c.get("fp64_wave_insts")
    if f64 is not None:
        out["fp64_wave_insts_per_launch"] = f64
        out["fp64_ginst_per_s"] = f64 / launch_seconds / 1e9
        out["fp64_issue_frac"] = f64 / launch_seconds / 1e9 / FP64_PEAK_GINST
        fl = c.get("fp64_flops")
        out["fp64_tflops"] = fl / launch_seconds / 1e12 if fl is not None else None
        out["fp64_flop_frac"] = out["fp64_tflops"] / FP64_PEAK_TFLOPS if fl is not None else None
    else:
        out["fp64_issue_frac"] = None
    lds = c.get("SQ_LDS_IDX_ACTIVE")
    out["lds_busy"] = lds / (CUS * CLK_HZ * launch_seconds) if lds is not None else None   # LDS-array cycles, one array per CU
    valu = c.get("SQ_INSTS_VALU")
    out["valu_wave_insts_per_launch"] = valu
    # all vector instructions, FP64 or not: SQ_ACTIVE_INST_VALU counts quad-cycles in which a wave has a VALU instruction in flight,
    # summed over the chip; divided by the SIMD-cycles of the launch it is the share of the time the vector pipes are occupied
    act = c.get("SQ_ACTIVE_INST_VALU")
    out["valu_busy"] = act * 4 / (CUS * SIMDS_PER_CU * CLK_HZ * launch_seconds) if act is not None else None
    busy = c.get("SQ_BUSY_CYCLES")
    out["effective_clock_ghz"] = busy / 32 / launch_seconds / 1e9 if busy is not None else None   # counter is summed over the 32 shader engines
    return out


def main():
    tag = sys.argv[1]
    here = "--here" in sys.argv[2:]
    G = os.path.join(ROOT, "gpurun_out")
    P = G if here else os.path.join(ROOT, "profiles")
    os.makedirs(P, exist_ok=True)

    def find(pattern):
        # gpurun merges a run's files INTO the local gpurun_out/, next to those of earlier runs with the same tag (rocprofv3 names its
        # csv files by process id): take the newest match, never an older run's
        r = glob.glob(os.path.join(G, pattern), recursive=True)
        return max(r, key=os.path.getmtime) if r else None

    bench_line = None
    tj = os.path.join(G, f"{tag}_trace.json")
    if os.path.exists(tj):
        try:
            bench_line = json.loads(open(tj).read().strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            print(f"(could not parse {tj}: {e})", file=sys.stderr)
    batch = bench_line["config"]["gates_per_gpu_per_step"] if bench_line else 4096
    pset = bench_line["config"]["param_set"] if bench_line else "SK-128"
    kernel = bench_line["roofline"]["kernel"] if bench_line else "blind_rotate"
    kbase = kernel.split("<")[0]
    lines = [f"# rocprofv3 summary `{tag}` — `python3 bench.py --no-cpu-baseline ...` ({batch} NAND / step, {pset}, 1x MI355X)\n"]
    hfile = os.path.join(G, f"{tag}_kernel_hash.txt")
    khash = open(hfile).read().strip() if os.path.exists(hfile) else kernel_source_hash()
    lines.append(f"kernel sources sha256 (tools/kernel_hash.py): `{khash}`\n")

    per_launch = {}
    ks = find(f"{tag}_trace/**/*kernel_stats.csv")
    if ks:
        if not here:
            shutil.copy(ks, os.path.join(P, f"{tag}_kernel_stats.csv"))
        lines.append("## kernel trace (`rocprofv3 --kernel-trace --stats`)\n")
        lines.append("| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|")
        for row in csv.DictReader(open(ks)):
            lines.append(f"| `{row['Name'][:90]}` | {row['Calls']} | {float(row['AverageNs'])/1e6:.4f} | {float(row['TotalDurationNs'])/1e6:.2f} | {float(row['Percentage']):.2f} |")
            if kbase in row["Name"] and "avg_launch_ns_kernel_trace" not in per_launch:
                per_launch["avg_launch_ns_kernel_trace"] = float(row["AverageNs"])
        lines.append("")
    if bench_line:
        r = bench_line["roofline"]
        lines.append(f"bench line of the traced run: value = {bench_line['value']:.0f} gates/s, dominant kernel `{kernel}` avg launch "
                     f"{r['avg_launch_ms']:.3f} ms (HIP events in bench.py)\n")

    def pmc(sub):
        f = find(f"{tag}_{sub}/**/*counter_collection.csv")
        if not f:
            return {}
        acc = defaultdict(lambda: defaultdict(list))
        extra = {}
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if kbase in row["Kernel_Name"]:
                for col in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Workgroup_Size", "Grid_Size"):
                    if col in row and row[col] != "":
                        extra[col] = float(row[col])
        if extra:
            per_launch.setdefault("dispatch", {}).update(extra)
        return acc

    def full_launch_mean(vals):
        big = sorted(vals)[-max(1, len(vals) // 2):]   # the full-batch launches (warm-up + timed), not set-up dispatches
        return sum(big) / len(big)

    for sub, title in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("sq", "SQ counters"), ("lds", "LDS / VMEM counters"),
                       ("f64", "VALU instruction mix"), ("tcc", "L2 hit/miss, GRBM")):
        acc = pmc(sub)
        if not acc:
            continue
        lines.append(f"## PMC pass: {title}\n")
        lines.append("| kernel | counter | per-launch mean (full-batch launches) | launches |\n|---|---|---|---|")
        for k, ctrs in acc.items():
            if "blind_rotate" not in k and "keyswitch" not in k:
                continue
            for c, vals in ctrs.items():
                m = full_launch_mean(vals)
                lines.append(f"| `{k[:60]}` | {c} | {m:.6g} | {len(vals)} |")
                if kbase in k:
                    per_launch[c] = m
        lines.append("")
    if "FETCH_SIZE" in per_launch and "WRITE_SIZE" in per_launch:
        # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 tallies the 128-B requests of 16-B/lane coalesced reads at 64 B: double FETCH_SIZE
        # (MI355X_MICROARCH.md, section HBM); WRITE_SIZE is exact for 16-B/lane streaming stores
        per_launch["hbm_bytes"] = (2 * per_launch["FETCH_SIZE"] + per_launch["WRITE_SIZE"]) * 1024
        lines.append(f"**HBM traffic of one `{kernel}` launch ({batch} gates)**: FETCH_SIZE = {per_launch['FETCH_SIZE']:.0f} KB (x2 gfx950 correction "
                     f"-> {2*per_launch['FETCH_SIZE']*1024/1e6:.1f} MB), WRITE_SIZE = {per_launch['WRITE_SIZE']:.0f} KB -> "
                     f"**{per_launch['hbm_bytes']/1e6:.1f} MB per launch** = {per_launch['hbm_bytes']/batch/1e3:.1f} KB per gate.\n")
    f64_names = ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")
    if all(n in per_launch for n in f64_names[:3]):
        add, mul, fma = (per_launch[n] for n in f64_names[:3])
        trans = per_launch.get(f64_names[3], 0.0)
        per_launch["fp64_wave_insts"] = add + mul + fma + trans
        per_launch["fp64_flops"] = 64.0 * (add + mul + trans + 2 * fma)
        per_launch["fp64_source"] = "PMC SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 (wave-level instruction counts)"
    counters = dict(tag=tag, kernel=kernel, param_set=pset, gates_per_launch=batch, kernel_source_sha256=khash, per_launch=per_launch,
                    method="rocprofv3 --kernel-trace --stats, then one --pmc pass per counter group (tools/profile_bench.sh); per-launch = mean of the "
                           "full-batch launches; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM")
    if "avg_launch_ns_kernel_trace" in per_launch:
        d = derive(counters, per_launch["avg_launch_ns_kernel_trace"] * 1e-9)
        counters["derived_at_kernel_trace_duration"] = d
        lines.append("## derived (tools/summarize_profile.py: derive(), launch duration = kernel-trace average)\n")
        for k, v in d.items():
            if v is not None:
                lines.append(f"* {k} = {v:.6g}")
        lines.append("")
    json.dump(counters, open(os.path.join(P, f"{tag}_counters.json"), "w"), indent=1)
    open(os.path.join(P, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
