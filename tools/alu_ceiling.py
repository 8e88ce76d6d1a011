#!/usr/bin/env python3
"""VALU-issue ceiling of hash_leaves_colmajor_kernel (the kernel that takes >55 % of a proof), from kept evidence.

The kernel is integer-ALU bound: neither the HBM nor the MFMA roofline applies.  Its ceiling is the VALU issue rate
for ITS instruction mix:
    cycles per permutation per SIMD  =  sum over instruction classes of  (dynamic count per permutation) x (issue cycles)
with
  * the issue cycles of each class MEASURED by tools/ubench_int.hip (profiles/r02_ubench_int.md: wall-clock rates with
    the shader clock read from s_memtime / s_memrealtime in the same run), and
  * the dynamic per-class counts taken from the gfx950 assembly of the kernel (`hipcc -S`), every basic block weighted
    by the trip counts of the loops around it (sponge loop: ceil(ncols / 8) permutations; inside one permutation:
    3 + 4 full rounds in loops, 4 blocks of four partial rounds in a loop, the rest straight-line), cross-checked against the PMC count SQ_INSTS_VALU.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/vxprover.s vectorx_amd/csrc/vxprover.hip
    python3 tools/alu_ceiling.py /tmp/vxprover.s profiles/r02_ubench_int.md [SQ_INSTS_VALU per permutation] [PMC clock GHz] [PMC cycles per VALU instruction] > profiles/r02_alu_ceiling.json

bench.py reads profiles/r02_alu_ceiling.json and reports, from the run's own hash_leaves time,
    achieved  = permutations/s x valu_insts_per_perm / 64      (wavefront-instructions per second)
    ceiling   = 1024 SIMDs x clock / mix_cycles_per_inst
    frac      = achieved / ceiling
"""
import json
import re
import sys
from collections import Counter

KERNEL = "_Z27hash_leaves_colmajor_kernel"   # mangled-name prefix
NCOLS = 135
PERMS = (NCOLS + 7) // 8  # sponge permutations per row
INNER_TRIPS = (3, 4, 4)   # loops inside one permutation since round 4's hashing schedule: full rounds 0..2, [round 3's S-boxes and the block its dense layer opens: straight-line], blocks 1..4 of 4 partial rounds, [block of 3: straight-line], last 4 full rounds

# plain 32-bit VOP1/VOP2 ops without carry-out: the only class that issues faster than 4.4 cycles
FAST = {"v_mov_b32_e32", "v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32",
        "v_lshrrev_b32_e32", "v_lshlrev_b32_e32", "v_ashrrev_i32_e32", "v_not_b32_e32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"}


def kernel_text(asm, name):
    m = re.search(r"^" + re.escape(name) + r"\w*:", asm, re.M)
    a = m.start()
    return asm[a:asm.index(".Lfunc_end", a)]


def weighted_histogram(text):
    """Counter of opcode -> dynamic executions per ROW, and the loop table used."""
    # loop headers and their depth-1 parent, from the compiler's comments
    loops = {}
    for m in re.finditer(r"^(\.LBB\d+_(\d+)):\s*;\s*(=>This Loop Header|Parent Loop BB\d+_(\d+)|=>This Inner Loop Header)", text, re.M):
        loops[m.group(2)] = m.group(4)  # header id -> parent id (None for top-level)
    # trip counts: the depth-1 loop with children = the sponge loop; its children in order = first full rounds (4), partial
    # blocks (7), last full rounds (4); any other top-level loop (the ncols <= 4 path) is not executed for 135 columns
    sponge = next(h for h, p in loops.items() if p is None and any(q == h for q in loops.values()))
    children = [h for h, p in loops.items() if p == sponge]
    trips = {sponge: PERMS}
    for h, t in zip(children, INNER_TRIPS):
        trips[h] = t
    hist = Counter()
    cur_weight = 1
    cur_loop = None
    for line in text.split("\n"):
        t = line.strip()
        m = re.match(r"^\.LBB\d+_(\d+):(.*)", t)
        if m:
            hid, rest = m.group(1), m.group(2)
            if hid in trips:
                cur_loop = hid
            else:
                mm = re.search(r"in Loop: Header=BB\d+_(\d+)", rest)
                cur_loop = mm.group(1) if mm else None
            continue
        mm = re.match(r"^; %bb\.\d+:\s*;\s*in Loop: Header=BB\d+_(\d+)", t)
        if mm:
            cur_loop = mm.group(1)
            continue
        if re.match(r"^; %bb\.\d+:", t):
            cur_loop = None
            continue
        if not t or t[0] in ";." or t.startswith("s_endpgm"):
            continue
        w = 1
        h = cur_loop
        if h is not None and h not in trips:
            w = 0  # a loop that is not on the 135-column path
        while h is not None and h in trips:
            w *= trips[h]
            h = loops.get(h)
        hist[t.split()[0]] += w
    return hist, {"sponge_loop": sponge, "children": children, "trips": trips}


def ubench_cycles(md):
    """cycles = last column of the single-opcode table"""
    cyc = {}
    for line in md.split("\n"):
        if not line.startswith("| `"):
            continue
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        cyc[cells[0].strip("`")] = float(cells[-1])
    return cyc


def main():
    asm = open(sys.argv[1]).read()
    md = open(sys.argv[2]).read()
    pmc_valu_per_perm = float(sys.argv[3]) if len(sys.argv) > 3 else None
    pmc_clock_ghz = float(sys.argv[4]) if len(sys.argv) > 4 else None   # GRBM_GUI_ACTIVE / 8 XCDs / duration of the kernel in the PMC run
    hist, loops = weighted_histogram(kernel_text(asm, KERNEL))
    valu = {k: v for k, v in hist.items() if k.startswith("v_")}
    n_valu = sum(valu.values())
    n_fast = sum(v for k, v in valu.items() if k in FAST)
    cyc = ubench_cycles(md)
    slow_ops = ["v_mad_u64_u32 v,s[..],a,b,acc", "v_add_co_u32_e64 (sgpr-pair carry out)", "v_subb_co_u32_e64 (sgpr in, sgpr out)",
                "v_cndmask_b32_e64 v,0,-1,s", "v_lshl_add_u64", "v_lshlrev_b64", "v_cmp_lt_u64_e32"]
    c_slow = sum(cyc[o] for o in slow_ops) / len(slow_ops)
    c_fast_run = cyc["v_mov_b32_e32"]
    # a fast op between slow ones does not reach its stand-alone rate: price it from the measured mixed stream
    # (v_mad ; v_add ; v_and = 3 x avg  ->  fast = (3 avg - slow) / 2)
    c_fast_mixed = (3 * cyc["v_mad_u64_u32 ; v_add_u32 ; v_and_b32 (per VALU)"] - c_slow) / 2
    mix_lo = (c_slow * (n_valu - n_fast) + c_fast_run * n_fast) / n_valu     # every fast op at its best rate
    mix_hi = (c_slow * (n_valu - n_fast) + c_fast_mixed * n_fast) / n_valu   # fast ops as measured inside slow streams
    per_perm = n_valu / PERMS
    out = {
        "kernel": "hash_leaves_colmajor_kernel",
        "method": "dynamic per-class VALU counts from the gfx950 assembly (loop-weighted) x issue cycles measured by tools/ubench_int.hip",
        "ncols": NCOLS, "perms_per_row": PERMS, "loops": loops,
        "valu_insts_per_row_static": n_valu, "valu_insts_per_perm_static": round(per_perm, 1),
        "valu_insts_per_perm_pmc": pmc_valu_per_perm,
        "pmc_run": {"clock_ghz": pmc_clock_ghz, "cycles_per_valu_inst": float(sys.argv[5]) if len(sys.argv) > 5 else None, "source": "profiles/r02_pmc_sq.md (commit workload)"},
        "s_nop_per_perm": round(hist.get("s_nop", 0) / PERMS, 1),
        "salu_per_perm": round(sum(v for k, v in hist.items() if k.startswith("s_") and k != "s_nop") / PERMS, 1),
        "class_counts_per_perm": {"slow (4.4-cycle class: v_mad_u64_u32, carry adds/subs, v_cndmask_e64, 64-bit ops)": round((n_valu - n_fast) / PERMS, 1),
                                  "fast (plain 32-bit VOP1/VOP2)": round(n_fast / PERMS, 1)},
        "top_opcodes_per_perm": {k: round(v / PERMS, 1) for k, v in Counter(valu).most_common(14)},
        "cycles": {"slow": round(c_slow, 3), "fast_in_runs": c_fast_run, "fast_between_slow": round(c_fast_mixed, 3)},
        "mix_cycles_per_inst": round(mix_hi, 3),
        "mix_cycles_per_inst_optimistic": round(mix_lo, 3),
        "simds": 1024,
        "source_ubench": sys.argv[2],
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
