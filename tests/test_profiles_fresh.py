"""The evidence bench.py prices its kernels with must belong to the sources that are running: profiles/roofline_traffic.json
records the git blob hash of every source file a priced kernel lives in, and bench.py (measured_traffic) reports
`roofline.traffic: null` when one of them changed after the PMC passes.  Round 5 lost the number that way (a comment-only
commit after the collection); this test makes the staleness visible HERE, before the driver's run."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_roofline_traffic_was_collected_on_these_sources():
    import sys
    sys.path.insert(0, ROOT)
    import bench
    path = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if not os.path.exists(path):
        pytest.skip("no profiles/roofline_traffic.json")
    rec = json.load(open(path))
    assert "knn64" in rec, "the dominant kernel (bench.py's `roofline`) has no traffic record"
    stale = []
    for site, r in rec.items():
        for rel, sha in r.get("source_blobs", {}).items():
            f = os.path.join(ROOT, rel)
            if not os.path.exists(f) or bench.git_blob_sha(f) != sha:
                stale.append((site, rel))
    assert not stale, "re-run tools/collect_round.sh + tools/collect_copy.sh: sources changed after the PMC passes: %r" % stale
    assert bench.measured_traffic(32, 1024, "knn64") is not None
