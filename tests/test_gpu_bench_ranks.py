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


def _run_bench(extra, timeout=1200):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PMR_DEBUG_POISON")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dist-backend", "gloo", "--workload", "cfg3", "--also", "none",
                           "--no-cpu-baseline", "--no-one-open", "--no-kernel-events"] + extra, env=env, capture_output=True, text=True, timeout=timeout)


def test_eight_ranks_first_contact_rehearsal():
    """The shape of the driver's 8-GPU run (cfg4: eight 61.44 MS/s streams, BASELINE.json configs[3]) on ONE device: eight processes,
    eight handles on device 0, eight streams (stream id = rank), eight parity records reduced into ONE JSON line, exit code 0 --
    everything of the N = 8 path but the other seven device ordinals and RCCL (which cannot exist on a 1-GPU box).  2^22-sample blocks."""
    r = _run_bench(["--gpus", "8", "--log2-block", "22", "--steps", "4", "--warmup", "1", "--regions", "1"])
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and [l for l in r.stdout.splitlines() if l.strip() and not l.startswith("{")] == [], r.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and "cfg4" in d["config"]["workload"] and d["value"] > 1e3
    assert d["dist"]["backend_used"] == "gloo" and d["dist"]["devices_visible"] >= 1
    assert d["library"]["experiment_build"] is False and not d["library"]["selected_by_PMR_LIBRARY"] and len(d["library"]["sha256"]) == 64
    p = d["parity_checked"]
    assert p["ranks"] == 8 and p["all_ok"] and p["worst_lsb"] <= 1
    assert [e["rank"] for e in p["per_rank"]] == list(range(8))
    assert all(e["ok"] and e["device"] == 0 and e["frames_checked"] > 200 and e["channels_checked"] == 224 for e in p["per_rank"])


def test_a_failing_rank_fails_the_job_after_rank_0s_line():
    """One rank's check made to fail (--test-fail-rank: that rank falsifies the PCM it hands to its own oracle check): rank 0's line still
    comes out, says which rank failed, and the job's exit code is non-zero -- what an 8-GPU run with one bad device would look like."""
    r = _run_bench(["--gpus", "2", "--log2-block", "22", "--steps", "4", "--warmup", "1", "--regions", "1", "--test-fail-rank", "1"])
    assert r.returncode != 0, r.stdout[-300:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    p = json.loads(lines[0])["parity_checked"]
    assert p["ranks"] == 2 and not p["all_ok"] and [e["ok"] for e in p["per_rank"]] == [True, False]
    assert "PARITY FAILED" in r.stderr
