#!/usr/bin/env python3
"""SQ counters of a whole vx_prove (rocprofv3 --pmc ... -- python3 bench.py --steps 1 --warmup 0 ...): one line per kernel.

    python3 tools/summarize_pmc_prove.py <counter_collection.csv> "<title>" "<command>" > profiles/r03_pmc_sq_prove.md

clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration; cycles per VALU instruction per SIMD = GRBM cycles/XCD x 1024 SIMDs /
SQ_INSTS_VALU; wait shares = SQ_WAIT_INST_ANY (issue stalls) and SQ_WAIT_INST_LDS over SQ_WAVE_CYCLES (all in quad-cycles
summed over waves, so the shares are per-wave averages)."""
import csv
import sys
from collections import defaultdict


def main():
    path, title, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    vals = defaultdict(lambda: defaultdict(float))
    ns = defaultdict(dict)
    meta = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        vals[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ns[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        meta[k] = (r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("SGPR_Count", ""), r.get("Scratch_Size", r.get("Private_Segment_Size", "")), r.get("LDS_Block_Size", ""))
    print(f"# {title}\n\n    {cmd}\n")
    print("| kernel | launches | ms | clock GHz | SQ_INSTS_VALU (1e6) | cycles / VALU instr / SIMD | ACTIVE_INST_VALU / INSTS_VALU | WAIT_INST_ANY / WAVE_CYCLES | WAIT_INST_LDS / WAVE_CYCLES | SALU / VALU | VGPR | scratch B/lane | LDS B |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    rows = []
    for k, v in vals.items():
        launches, total_ns = len(ns[k]), sum(ns[k].values())
        if total_ns < 50_000:
            continue
        valu = v.get("SQ_INSTS_VALU", 0.0)
        grbm = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        ghz = grbm / total_ns if total_ns else 0
        cpi = grbm * 1024 / valu if valu else float("nan")
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        rows.append((total_ns, f"| `{k[:90]}` | {launches} | {total_ns / 1e6:.3f} | {ghz:.3f} | {valu / 1e6:.1f} | {cpi:.2f} | "
                     f"{(v.get('SQ_ACTIVE_INST_VALU', 0) / valu if valu else 0):.3f} | {(v.get('SQ_WAIT_INST_ANY', 0) / wc if wc else 0):.3f} | "
                     f"{(v.get('SQ_WAIT_INST_LDS', 0) / wc if wc else 0):.3f} | {(v.get('SQ_INSTS_SALU', 0) / valu if valu else 0):.3f} | "
                     f"{meta[k][0]} | {meta[k][3]} | {meta[k][4]} |"))
    for _, line in sorted(rows, reverse=True):
        print(line)


if __name__ == "__main__":
    main()
