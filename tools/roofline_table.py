"""Per-kernel roofline table of one profiled step: python tools/roofline_table.py r06_trainstep_b32_n1024 [steps]
Joins profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats) with profiles/<tag>_pmc_hbm_traffic.csv (separate
FETCH_SIZE / WRITE_SIZE passes, FETCH x 2 on gfx950) and prints a markdown table: share of the step, average duration, HBM
bytes per launch as the counters saw them, the HBM rate that is and its fraction of 8 TB/s."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, steps=63):
    stats = [r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv")))
             if r["avg_us"] and r["kernel"].upper() != "TOTAL"]
    traffic = {}
    for r in csv.DictReader(open(os.path.join(ROOT, "profiles", tag + "_pmc_hbm_traffic.csv"))):
        traffic[(r["kernel"], r["grid_workgroups"])] = r
    total = sum(float(r["total_us"]) for r in stats if r["total_us"])
    print("| kernel | grid | launches / step | avg µs | % of step | HBM MB / launch (PMC) | GB/s | of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|")
    for r in stats:
        if not r["total_us"] or float(r["total_us"]) / total < 0.004:
            continue
        t = traffic.get((r["kernel"], r["grid_workgroups"]))
        mb = float(t["hbm_bytes_per_launch"]) / 1e6 if t else None
        avg = float(r["avg_us"])
        gbs = mb * 1e6 / (avg * 1e-6) / 1e9 if mb else None
        print("| `%s` | %s | %.1f | %.1f | %.1f | %s | %s | %s |" % (
            r["kernel"][:58], r["grid_workgroups"], int(float(r["calls"])) / float(steps), avg, 100.0 * float(r["total_us"]) / total,
            "%.1f" % mb if mb is not None else "-", "%.0f" % gbs if gbs else "-", "%.2f" % (gbs / 8000.0) if gbs else "-"))
    print("\ntrace total per step: %.1f µs" % (total / steps))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 63)
