#!/bin/bash
# SQ counter passes over one program: bash tools/pmc_kernel.sh TAG KERNEL_SUBSTRING -- python3 /abs/path/prog.py args...
# (the passes run from /tmp: give the program by absolute path, e.g. $GRAFT_REPO_ROOT/tools/dev/run_f32_agg.py 32 fwd;
#  PMC_EXTRA="COUNTER ..." adds a third pass, e.g. TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; PAT=$2; shift 3
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
         ${PMC_EXTRA:+"$PMC_EXTRA"}; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d "$OUT/p$i" -o p -- "$@" > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" "$PAT" <<'PY'
import sqlite3, glob, sys, json
out, pat = sys.argv[1], sys.argv[2]
res = {}
for db in glob.glob(out + "/p*/**/*_results.db", recursive=True):
    cur = sqlite3.connect(db).cursor()
    for name, counter, cnt, avg in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like ? group by kernel_name, counter_name", ("%" + pat + "%",)):
        res.setdefault(name.split("(")[0][-60:], {})[counter] = round(avg, 1)
    for name, cnt, avg in cur.execute("select name, count(*), avg(end-start)/1000.0 from kernels where name like ? group by name", ("%" + pat + "%",)):
        res.setdefault(name.split("(")[0][-60:], {})["avg_us(pmc run)"] = round(avg, 2)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf "$OUT"/p1 "$OUT"/p2 "$OUT"/p3
