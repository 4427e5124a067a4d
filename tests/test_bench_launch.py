"""`python bench.py --gpus N` must start its own N ranks (the driver's N = 1 command line has no launcher around it,
and neither will its N > 1 one).  Exercised here without a GPU: --stub-scorer keeps the whole N-rank protocol of the
real run -- self-launch under torch.distributed.run on 127.0.0.1, rendezvous, rank 0 quantizes + packs, blob broadcast
(gloo here, RCCL on GPUs) and per-rank sha256, fences around exactly K steps, max-over-ranks timing, one JSON line from
rank 0 -- and replaces only the scorer.  The reference's concurrency model this stands for: independent callers over
one immutable model (MultiThreadedStressTest.java:48-61)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(extra, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub-scorer", "--steps", "5", "--warmup", "2"] + extra,
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # ONE line, from rank 0 only
    return json.loads(lines[0])


def test_bench_starts_its_own_two_ranks():
    res = _run(["--gpus", "2"])
    assert res["n_gpus"] == 2 and res["steps"] == 5 and res["warmup"] == 2 and res["scaling"] == "weak"
    m = res["multi_gpu"]
    assert m["ranks"] == 2 and m["collective_backend"] == "gloo" and m["rccl_ranks"] == 0
    assert m["blob_sha256_all_equal"] and len(m["blob_sha256"]) == 64
    assert len(m["per_rank_frames_per_s"]["all"]) == 2
    assert m["per_rank_frames_per_s"]["min"] <= m["per_rank_frames_per_s"]["max"]
    # whole-job value: both ranks' frames over the slowest rank's time
    assert res["config"]["global_frames"] == 2 * res["config"]["frames_per_gpu"]
    assert res["value"] <= 2 * m["per_rank_frames_per_s"]["max"] * 1.001
    # the host-fed legs of the N-rank line (round 6): per-rank caller threads and the one-process group, beside `value`
    hf = res["host_fed"]
    assert len(hf["per_rank_utterances_per_s"]) == 2 and hf["caller_threads_per_rank"] == 4 and hf["frames_per_utterance"] == 100
    assert hf["utterances_per_s_whole_node"] <= sum(hf["per_rank_utterances_per_s"]) * 1.001
    assert hf["frames_per_s_whole_node"] == pytest.approx(100 * hf["utterances_per_s_whole_node"], rel=1e-3)
    assert res["one_process_group"]["devices"] == 2 and res["one_process_group"]["frames_per_s_whole_node"] > 0


def test_bench_single_rank_needs_no_launcher():
    res = _run(["--gpus", "1"])
    assert res["n_gpus"] == 1 and res["multi_gpu"]["ranks"] == 1 and res["multi_gpu"]["blob_sha256_all_equal"]


def test_bench_refuses_a_mismatched_launcher():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub-scorer", "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "launcher started 3" in (p.stderr + p.stdout)
