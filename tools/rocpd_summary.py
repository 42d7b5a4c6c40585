#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace --stats run) as a small text table.
Usage: tools/rocpd_summary.py <results.db> > profiles/<name>.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "rocprim::", name)
    m = re.search(r"rocprim::detail::(\w+)<", name.split("trampoline_kernel<", 1)[-1]) if "trampoline_kernel" in name else None
    if m:
        tail = "partition" if "partition_config_selector" in name else ""
        return "rocprim::%s %s" % (m.group(1), tail)
    return name.split("(")[0][:90]


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("# rocprofv3 --kernel-trace --stats (rocpd top_kernels view); durations in microseconds")
    print("%-72s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in rows:
        print("%-72s %8d %14.3f %12.3f %8.2f" % (short(name), calls, total, avg, pct))
    print("\n# per-dispatch resources of the scoring kernel")
    for r in cur.execute("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count, "
                         "min(duration), avg(duration), max(duration), count(*) from kernels where name like '%score_%' group by name"):
        print("%s grid=%d wg=%d lds=%d scratch=%d vgpr=%d agpr=%d sgpr=%d dur_ns min/avg/max=%d/%d/%d n=%d"
              % ((short(r[0]),) + tuple(r[1:8]) + (r[8], r[9], r[10], r[11])))


if __name__ == "__main__":
    main(sys.argv[1])
