"""Turns a rocprofv3 (rocpd sqlite) result into the kernel-stats CSV we commit.
usage: python profiles/summarize.py gpurun_out/prof/x_results.db profiles/r01_name.csv"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)            # drop the argument list
    name = name.replace("void ", "").replace("cloudaae::", "")
    if "at::native" in name:
        m = re.search(r"(FillFunctor|CUDAFunctor_add|MulFunctor|normal_kernel|[A-Za-z_]+Functor)", name)
        name = "torch:" + (m.group(1) if m else name[:40])
    return name[:110]


def main(db, out):
    """One row per (kernel, grid): the same template instance serves several layers (e.g. the
    128x128 GEMM runs the 64-channel edge GEMMs and the dgcnn_agg GEMM), and only the grid tells
    them apart.  grid = workgroups (rocprofv3 reports work-items; divided by the workgroup size)."""
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute(
        "select name, grid_x / workgroup_x, grid_y / workgroup_y, grid_z / workgroup_z, count(*), "
        "sum(end - start) / 1000.0, avg(end - start) / 1000.0 from kernels "
        "group by name, grid_x, grid_y, grid_z, workgroup_x order by 6 desc"))
    total = sum(r[5] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_workgroups", "calls", "total_us", "avg_us", "percent"])
        for n, gx, gy, gz, c, t, a in rows:
            w.writerow([short(n), "%dx%dx%d" % (gx, gy, gz), c, round(t, 1), round(a, 3), round(100.0 * t / total, 2)])
        w.writerow(["TOTAL", "", sum(r[4] for r in rows), round(total, 1), "", 100.0])
    print("wrote", out, "rows:", len(rows), "total_us:", round(total, 1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
