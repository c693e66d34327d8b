#!/bin/bash
# The kernels of ONE replayed step in start order, with stream, start offset and duration (us):
#   bash tools/kernel_sequence.sh [bench.py args...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(mktemp -d /tmp/ks.XXXXXX)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$OUT" -o t -- python3 "$ROOT/bench.py" --steps 12 --warmup 4 --step-only "$@" > "$OUT/log" 2>&1
python3 - "$OUT" <<'PY'
import sqlite3, glob, sys
for db in glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = list(cur.execute("select name, start, end, %s from kernels order by start" % q))
    # the last step: from the last input_assemble kernel on
    idx = [i for i, r in enumerate(rows) if "input_assemble" in r[0]]
    if len(idx) < 2:
        continue
    a, b = idx[-2], idx[-1]
    t0 = rows[a][1]
    for name, s, e, qid in rows[a:b]:
        print("%8.1f %7.1f q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, qid, name.split("(")[0][-70:]))
PY
rm -rf "$OUT"
