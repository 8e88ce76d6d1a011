#!/usr/bin/env python3
"""Register / scratch budget of every kernel of libvxprover.so, read from the gfx950 code object's metadata.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/vxprover.s vectorx_amd/csrc/vxprover.hip
    python3 tools/kernel_resources.py /tmp/vxprover.s > profiles/r03_kernel_resources.md

(no GPU needed: hipcc cross-compiles).  `private_segment_fixed_size` = scratch bytes per lane; `vgpr_spill_count` /
`sgpr_spill_count` = values the register allocator had to park (SGPR spills go to VGPR lanes, VGPR spills to scratch)."""
import re
import sys


def kernels(path):
    out, cur = [], None
    for ln in open(path):
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)$", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" or (k == "args" and cur is None):
            pass
        if ln.lstrip().startswith("- .") and k in ("agpr_count", "args"):   # first key of a kernel's metadata map
            cur = {}
            out.append(cur)
        if cur is not None and k in ("name", "vgpr_count", "sgpr_count", "agpr_count", "private_segment_fixed_size", "vgpr_spill_count",
                                     "sgpr_spill_count", "group_segment_fixed_size", "max_flat_workgroup_size"):
            cur[k] = v
    return [k for k in out if "name" in k]


def short(name):
    m = re.match(r"_Z\d+(\w+?)(?:I|P|\d|v|$)", name)
    return name


def main():
    ks = kernels(sys.argv[1])
    import subprocess
    print("# Kernel register / scratch budget (gfx950 code-object metadata; tools/kernel_resources.py)\n")
    print("| kernel | VGPRs | AGPRs | SGPRs | VGPR spills | SGPR spills | scratch B/lane | LDS B | max block |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
    for k in ks:
        try:
            nm = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
        except Exception:
            nm = k["name"]
        print(f"| `{nm}` | {k.get('vgpr_count')} | {k.get('agpr_count', 0)} | {k.get('sgpr_count')} | {k.get('vgpr_spill_count')} | "
              f"{k.get('sgpr_spill_count')} | {k.get('private_segment_fixed_size')} | {k.get('group_segment_fixed_size')} | {k.get('max_flat_workgroup_size')} |")


if __name__ == "__main__":
    main()
