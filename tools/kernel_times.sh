#!/bin/bash
# Average duration of the kernels whose name contains PATTERN in a short profiled run of the step bench:
#   bash tools/kernel_times.sh PATTERN [bench.py args...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PAT=$1; shift
OUT=$(mktemp -d /tmp/kt.XXXXXX)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$OUT" -o t -- python3 "$ROOT/bench.py" --steps 30 --warmup 5 --step-only "$@" > "$OUT/log" 2>&1
python3 - "$OUT" "$PAT" <<'PY'
import sqlite3, glob, sys
for db in glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True):
    cur = sqlite3.connect(db).cursor()
    for name, n, avg in cur.execute("select name, count(*), avg(end-start)/1000.0 from kernels where name like ? group by name order by 3 desc", ("%" + sys.argv[2] + "%",)):
        print("%-90s %5d %8.2f us" % (name.split("(")[0][-90:], n, avg))
PY
rm -rf "$OUT"
