"""bench.py's N > 1 branch on the one-GPU box: two and three ranks that share the GPU,
partitions through host memory (gloo) -- the pack calls, the windowed exchange that runs
while the next call packs, both exchange protocols and the end-to-end check of what
arrived (counts and checksums per (source, destination) pair, destinations, emit order)
all execute; only xGMI is missing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ranks,extra", [(2, []), (3, ["--split", "1"]), (2, ["--exchange", "rounds"])])
def test_bench_multirank_rehearsal_verifies_its_exchange(ranks, extra):
    env = dict(os.environ, MGS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--records", "3e6", "--steps", "1",
           "--warmup", "1", "--comm-size", "8M", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == ranks and out["config"]["rounds"] >= 3
    ex = out["exchange"]
    if "--exchange" in extra:       # round by round: the receive side is one image per round, not checked here
        assert ex["exchanges"] >= 3
        return
    assert ex["verified"] is True and ex["records_checked"] > 0
    assert ex["batch_ms"] > 0 and ex["rounds_ms"] > 0
    if "--split" not in extra:
        assert ex["split"] == 4 and ex["windows_equal_batch"] is True
