"""`bench.py --gpus 2` on the GPU box: the N > 1 code path end to end -- the parent starts its own ranks (never an exec), each rank runs ITS
stream on its device (both ranks share device 0 on a 1-GPU box: gloo for the barrier / MAX), checks it against the oracle, the verdicts
are reduced over the gloo group and rank 0's ONE JSON line carries every rank's record; the exit code is the parity verdict.  This is the
path the driver's 8-GPU run takes for cfg4 (BASELINE.json configs[3]: one 61.44 MS/s stream per GPU; the reference's loop per stream,
src/sdr_pmr446.c:788-908) -- here at a block size that keeps the test under a minute."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.nopoison]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_check_their_own_streams_and_report_in_one_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PMR_DEBUG_POISON")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--workload", "cfg3", "--also", "none",
                        "--log2-block", "24", "--steps", "6", "--warmup", "2", "--regions", "2", "--no-cpu-baseline", "--no-one-open"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and [l for l in r.stdout.splitlines() if l.strip() and not l.startswith("{")] == [], r.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["dist"]["backend_used"] == "gloo"
    assert "cfg4" in d["config"]["workload"] and d["value"] > 1e4
    p = d["parity_checked"]
    assert p["ranks"] == 2 and p["all_ok"] and p["worst_lsb"] <= 1
    assert [e["rank"] for e in p["per_rank"]] == [0, 1] and all(e["ok"] and e["frames_checked"] > 1000 and e["channels_checked"] == 224 for e in p["per_rank"])
