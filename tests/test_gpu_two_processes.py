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
    if api.device_shared(0):  # (advisor, round 5) machine state, not a defect: a leftover or parallel scorer holds the marker
        dnn.delete()
        pytest.skip("another process already holds the advisory marker of GPU 0: this test needs to be the first scorer on it")
    n = 10000
    x = torch.from_numpy(F.synth_features(n, 432, seed=41)).cuda()
    out = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    want = hashlib.sha256(out[:256].cpu().numpy().tobytes()).hexdigest()
    alone = []  # this box's single-process pass, the yardstick of the bounds below
    for _ in range(9):
        t0 = time.perf_counter()
        dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
        torch.cuda.synchronize()
        alone.append(time.perf_counter() - t0)
    base = sorted(alone)[len(alone) // 2]
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
        # a small multiple of this box's pass alone (two processes take turns on the device) -- a sat-out wait is tens of
        # milliseconds, i.e. tens of passes.  (Bounds relative to the run's own baseline: advisor, round 5.)
        assert t[-2] < 3.0 * t[len(t) // 2] + 4.0 * base and t[len(t) // 2] < 12.0 * base, (base, t)
    dnn.delete()


def test_a_give_up_flips_the_model_to_the_scale_pass_for_good(net_model_path, capfd):
    """the device side of the detection.  A fused soft-max workgroup that sits out its bounded wait
    raises a word in host memory (fdnn_gemm.hip / fdnn_ppo.hip); from the next call on the model runs GEMM + scale pass --
    the same bits -- and says so once.  The word is written here as the kernel would (fdnn_debug_raise_fuse_fault)."""
    import torch

    if os.environ.get("FDNN_FUSE_NORM"):
        pytest.skip("FDNN_FUSE_NORM forces the path")
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    if api.device_shared(0):
        dnn.delete()
        pytest.skip("another process holds the advisory marker of GPU 0: this process does not fuse to begin with")
    n = 4096
    x = torch.from_numpy(F.synth_features(n, 432, seed=43)).cuda()
    out = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream

    def one_pass():
        dnn.profileBegin()
        dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
        torch.cuda.synchronize()
        prof = dnn.profileEnd()
        return out.cpu().numpy().copy(), {k for k, v in prof.items() if v["launches"]}

    fused_rows, fused_classes = one_pass()
    assert "normalize" not in fused_classes
    capfd.readouterr()
    dnn.raiseFuseFault(1)
    a, ca = one_pass()
    b, cb = one_pass()
    err = capfd.readouterr().err
    assert "normalize" in ca and "normalize" in cb
    assert err.count("sat out its bounded wait") == 1
    assert np.array_equal(a.view(np.uint32), fused_rows.view(np.uint32)) and np.array_equal(b.view(np.uint32), fused_rows.view(np.uint32))
    dnn.raiseFuseFault(0)
    c, cc = one_pass()
    assert "normalize" not in cc and np.array_equal(c.view(np.uint32), fused_rows.view(np.uint32))
    dnn.delete()
