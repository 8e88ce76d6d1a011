#!/usr/bin/env python3
"""Turn a `rocprofv3 --kernel-trace --stats --output-format csv` run into the markdown summary kept under
profiles/ (gpurun_out/ is scratch).  Also cross-checks bench.py's HIP-event numbers: for the coset-LDE launches
(the ntt_pass_kernel dispatches with Grid_Size_Z == 8 workgroups deep, i.e. 8 cosets) it prints the per-launch
duration (both passes) next to bench.py's `roofline.ms_per_launch`.

usage: tools/summarize_rocprof.py <dir-with-*_kernel_stats.csv> <bench.json> <out.md> [title]
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    d, bench_json, out = sys.argv[1], sys.argv[2], sys.argv[3]
    title = sys.argv[4] if len(sys.argv) > 4 else "rocprofv3 kernel summary"
    stats = glob.glob(d + "/*kernel_stats.csv")[0]
    trace = glob.glob(d + "/*kernel_trace.csv")[0]
    rows = list(csv.DictReader(open(stats)))
    bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
    L = [f"# {title}", "",
         f"Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps {bench['steps']} --warmup {bench['warmup']} --no-cpu-baseline --no-host-witness-leg`",
         f"(workload: {bench['config']['workload'][:160]}…)", "",
         f"bench.py line of the same run: value = {bench['value']:.4f} {bench['unit']}, ms_per_step = {bench['ms_per_step']:.2f}", "",
         "## Per-kernel totals (all dispatches of the run: circuit build + warmup + timed steps)", "",
         "| kernel | calls | total ms | avg µs | % |", "|---|---:|---:|---:|---:|"]
    for r in rows[:25]:
        name = r["Name"].split("(")[0].replace("void ", "")
        L.append(f"| `{name}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} |")
    # coset-LDE launches: z-dimension 8 cosets (Grid_Size_Z = 8 workgroups * 1 thread)
    tr = list(csv.DictReader(open(trace)))
    lde = defaultdict(list)
    for r in tr:
        # coset-LDE launches: second pass has 8 cosets in grid.z; the first pass (the PRE-scaled instantiation) folds the
        # cosets into grid.x (XCD-aware ordering), so it is recognised by its template arguments instead
        folded_first_pass = "true, true, true, false>" in r["Kernel_Name"]
        if "pass_kernel" in r["Kernel_Name"] and "ntt" in r["Kernel_Name"] and (int(r["Grid_Size_Z"]) == 8 or folded_first_pass):
            cols = int(r["Grid_Size_Y"])
            lde[(r["Kernel_Name"].split("(")[0].replace("void ", ""), cols)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    L += ["", "## Coset-LDE dispatches (8 cosets per column), by kernel instantiation and column count", "",
          "| kernel | columns | dispatches | avg ms |", "|---|---:|---:|---:|"]
    per_cols = defaultdict(float)
    counts = {}
    for (k, cols), v in sorted(lde.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        avg = sum(v) / len(v) / 1e6
        L.append(f"| `{k}` | {cols} | {len(v)} | {avg:.3f} |")
        per_cols[cols] += avg
        counts[cols] = len(v)
    if per_cols:
        # bench.py averages over the launches of a proof: wires(135) + zs(20) + quotient(16) [+ fri(4) is labelled fri_lde]
        proof_cols = [c for c in per_cols if c in (135, 20, 16)]
        if proof_cols:
            avg_launch = sum(per_cols[c] for c in proof_cols) / len(proof_cols)
            L += ["", f"Average coset-LDE launch (both passes; mean over the {sorted(proof_cols)}-column launches of a proof): "
                      f"**{avg_launch:.3f} ms** by rocprof vs **{bench['roofline']['ms_per_launch']:.3f} ms** by bench.py's HIP events "
                      f"(`roofline.ms_per_launch`; algorithmic bytes per launch {bench['roofline']['alg_bytes_per_launch'] / 1e9:.2f} GB ⇒ "
                      f"{bench['roofline']['achieved']} GB/s, frac {bench['roofline']['frac']})."]
    L += ["", "## bench.py stage times (HIP events on the library's stream), ms per proof", "",
          "| stage | ms | algorithmic GB/s |", "|---|---:|---:|"]
    for k, v in bench["stage_ms_per_step"].items():
        L.append(f"| {k} | {v} | {bench.get('stage_alg_GBps', {}).get(k, '')} |")
    open(out, "w").write("\n".join(L) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
