#!/bin/bash
# per-kernel averages of one kNN shape: bash tools/dev/knn_kernels.sh B N C LD K
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$(mktemp -d /tmp/kk.XXXXXX)
cd /tmp && export TMPDIR=/tmp
WARM_S=0.3 timeout 120 rocprofv3 --kernel-trace -d "$OUT" -o t -- python3 "$ROOT/tools/bench_knn1.py" "$@" 20 > "$OUT/log" 2>&1
python3 - "$OUT" <<'PY'
import sqlite3, glob, sys
for db in glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True):
    cur = sqlite3.connect(db).cursor()
    for name, n, avg in cur.execute("select name, count(*), avg(end-start)/1000.0 from kernels where name like '%knn%' or name like '%fill%' group by name order by 3 desc"):
        print("%-80s %6d %9.2f us" % (name.split("(")[0][-80:], n, avg))
PY
rm -rf "$OUT"
