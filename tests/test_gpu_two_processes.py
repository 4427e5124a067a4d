"""Two PROCESSES scoring on one GPU without anybody setting FDNN_FUSE_NORM.  The fused soft-max's
workgroups wait for their frame tile's siblings; two processes' fused kernels can hold each other's CUs.  The first process
on a device owns its marker (/dev/shm/fdnn-gpu-<bus id>); a second one finds it taken and runs the unfused output path by
itself.  Concurrency model of the reference: MultiThreadedStressTest.java:48-69 (threads; processes are this library's
own concern)."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from fast_dnn_amd import api, formats as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from fast_dnn_amd import api, formats as F
dnn = api.QuantizedDnn.loadFromFile(sys.argv[2])
n = 10000
x = torch.from_numpy(F.synth_features(n, 432, seed=41)).cuda()
out = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s); torch.cuda.synchronize()
print("READY", flush=True)
sys.stdin.readline()
times = []
for _ in range(20):
    t0 = time.perf_counter()
    dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s); torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
print(json.dumps({"shared": api.device_shared(0), "giveups": dnn.fuseGiveups(), "times": times,
                  "rowsum": float(out[:64].sum(1).mean().item()), "sha": __import__("hashlib").sha256(out[:256].cpu().numpy().tobytes()).hexdigest()}), flush=True)
"""


def test_second_process_on_the_device_runs_unfused_by_itself(net_model_path):
    import hashlib

    import torch

    env = {k: v for k, v in os.environ.items() if k != "FDNN_FUSE_NORM"}
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)  # this process owns (or already owned) the marker
    assert not api.device_shared(0)
    n = 10000
    x = torch.from_numpy(F.synth_features(n, 432, seed=41)).cuda()
    out = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    want = hashlib.sha256(out[:256].cpu().numpy().tobytes()).hexdigest()
    child = subprocess.Popen([sys.executable, "-c", CHILD, ROOT, net_model_path], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             text=True, env=env)
    line = child.stdout.readline()
    assert line.strip() == "READY", (line, child.stderr.read()[-2000:])
    child.stdin.write("go\n")
    child.stdin.flush()
    times = []
    for _ in range(20):
        t0 = time.perf_counter()
        dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    res = json.loads(child.stdout.readline())
    err = child.stderr.read()
    child.wait(timeout=60)
    assert res["shared"] is True and "unfused soft-max" in err          # the second process noticed, and said so once
    assert res["giveups"] == 0 and dnn.fuseGiveups() == 0               # nobody sat out a bounded wait
    assert abs(res["rowsum"] - 1.0) < 1e-3
    assert res["sha"] == want                                           # fused and unfused paths: the same bits (one tree order)
    assert hashlib.sha256(out[:256].cpu().numpy().tobytes()).hexdigest() == want
    for t in (times, res["times"]):
        t = sorted(t)
        # no cliff: the second slowest of the 20 within 3 x the median (one host hiccup is not a cliff), and the median itself
        # nowhere near the tens of milliseconds a sat-out wait costs
        assert t[-2] < 3.0 * t[len(t) // 2] + 2e-3 and t[len(t) // 2] < 20e-3, t
    dnn.delete()
