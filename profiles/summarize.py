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
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    total = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for n, c, t, a, p in rows:
            w.writerow([short(n), c, round(t, 1), round(a, 3), round(p, 2)])
        w.writerow(["TOTAL", sum(r[1] for r in rows), round(total, 1), "", 100.0])
    print("wrote", out, "kernels:", len(rows), "total_us:", round(total, 1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
