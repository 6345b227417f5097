#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output: per-kernel duration stats (--kernel-trace --stats) and per-kernel mean
counter values (--pmc passes).  Usage: tools/pmc_summary.py <dir> [<dir> ...]  -> text on stdout."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def foreign(name: str) -> bool:
    """torch's own elementwise / reduction kernels and the runtime's copy / fill kernels: not part of the path, left out of the record"""
    return name.startswith(("at::", "__amd_rocclr", "void at::"))


def main():
    for d in sys.argv[1:]:
        print(f"### {d.split('gpurun_out/')[-1]}")
        for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
            print(f"# kernel stats: {os.path.relpath(f, d)}")
            with open(f) as fh:
                rows = list(csv.DictReader(fh))
            for r in [r for r in rows if not foreign(short(r['Name']))][:25]:
                print(f"  {float(r['Percentage']):6.2f}%  calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:10.2f} us  "
                      f"min {float(r['MinNs'])/1e3:9.2f}  max {float(r['MaxNs'])/1e3:9.2f}  {short(r['Name'])}")
        acc = defaultdict(lambda: [0.0, 0])
        meta = {}
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = (short(r["Kernel_Name"]), r["Counter_Name"])
                    acc[k][0] += float(r["Counter_Value"])
                    acc[k][1] += 1
                    meta[short(r["Kernel_Name"])] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
                                                     r.get("Grid_Size"), r.get("Workgroup_Size"))
        kernels = sorted({k[0] for k in acc if not foreign(k[0])})
        for kn in kernels:
            m = meta[kn]
            print(f"# counters (mean per dispatch): {kn}\n    vgpr {m[0]} agpr {m[1]} sgpr {m[2]} lds {m[3]} grid {m[4]} wg {m[5]}")
            for (k2, c), (tot, n) in sorted(acc.items()):
                if k2 == kn:
                    print(f"    {c:28s} {tot / n:18.1f}   (dispatches {n})")


if __name__ == "__main__":
    main()
