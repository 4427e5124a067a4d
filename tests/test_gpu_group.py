"""One process, several devices (fdnn_group_*): the host-C++ counterpart of dist.py.  The GPU box
has ONE device, so replicas are placed on it twice or three times -- the whole protocol runs
(leader quantizes, blob copied device-to-device, peers adopt it, frames sharded over host
threads, results gathered in place) and must equal the single-device result bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from fast_dnn_amd import api, formats as F
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_group_equals_single_device(mid_model_path, devices, monkeypatch):
    # (thresholds are read when the group is created: make a 1001-frame call a "large" one, so that it is sharded)
    monkeypatch.setenv("FDNN_GROUP_SPLIT_MIN", "64")
    monkeypatch.setenv("FDNN_GROUP_SHARD_MIN", "16")
    x = F.synth_features(1001, 432, seed=88)   # ragged shards
    one = api.QuantizedDnn.loadFromFile(mid_model_path, device=0)
    want = one.calculate(x)
    grp = api.DeviceGroup(mid_model_path, devices)
    assert grp.size() == len(devices) and grp.weightTransport() == "peer-copy"
    got = grp.calculate(x)
    assert np.array_equal(got, want)
    assert np.abs(got[:50] - Oracle(mid_model_path).calculate(x[:50])).max() <= 2e-6
    # every replica alone gives the same bits (bit-identical weights on all of them)
    for r in range(grp.size()):
        assert np.array_equal(grp.model(r).calculate(x[:200]), want[:200])
    # fewer frames than devices, one frame, and the argument errors of QuantizedDnn.java:154-161
    assert np.array_equal(grp.calculate(x[:1]), want[:1])
    assert np.array_equal(grp.calculate(x[:2]), want[:2])
    assert grp.calculate(np.zeros((0, 432), np.float32)).shape == (0, 0)
    with pytest.raises(ValueError):
        grp.calculate(np.zeros((3, 429), np.float32))
    # fused layer-0 flavour is a group-wide setting
    grp.model(0).setInputLayerFma(True)
    one.setInputLayerFma(True)
    assert np.array_equal(grp.calculate(x), one.calculate(x))
    grp.delete()
    one.delete()


def test_group_routes_small_calls_whole_and_large_calls_over_persistent_workers(mid_model_path):
    """Default thresholds: a call of fewer than 4096 frames stays whole on ONE replica (round robin; the JNI serving
    shape: many Java threads with 100-frame utterances), a larger one is cut into shards of at least 1024 frames that
    run on the devices' persistent, NUMA-pinned host threads.  Results are the single device's bits either way,
    also from eight concurrent caller threads."""
    import threading

    one = api.QuantizedDnn.loadFromFile(mid_model_path, device=0)
    xs = [F.synth_features(100 + 7 * i, 432, seed=200 + i) for i in range(8)]
    wants = [one.calculate(x) for x in xs]
    big = F.synth_features(5000, 432, seed=99)
    want_big = one.calculate(big)
    grp = api.DeviceGroup(mid_model_path, [0, 0, 0])
    assert grp.workerCpus(1) == ""              # no worker threads before the first large call
    got = [None] * 8

    def caller(i):
        for _ in range(5):
            got[i] = grp.calculate(xs[i])

    th = [threading.Thread(target=caller, args=(i,)) for i in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(8):
        assert np.array_equal(got[i], wants[i])
    assert np.array_equal(grp.calculate(big), want_big)      # 5000 frames: three shards (1667 / 1667 / 1666) on the workers
    assert np.array_equal(grp.calculate(big[:4500]), want_big[:4500])
    cpus = [grp.workerCpus(r) for r in range(3)]
    assert all(isinstance(c, str) for c in cpus)             # "" where /sys has no local_cpulist for the device
    grp.delete()
    one.delete()


def test_env_var_makes_the_plain_load_a_group(mid_model_path, tmp_path):
    """FDNN_DEVICES: fdnn_model_load (what the JNI initialize() calls) returns the leader of an
    attached group; calculate() on it shards, delete() frees every replica.  Also the one-rank
    ncclBroadcast plumbing (FDNN_GROUP_BCAST=rccl on a one-device group)."""
    code = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
from fast_dnn_amd import api, formats as F
x = F.synth_features(333, 432, seed=5)
dnn = api.QuantizedDnn.loadFromFile({mid_model_path!r})
np.save({str(tmp_path / 'got.npy')!r}, dnn.calculate(x))
ctx = dnn.getNewLazyContext(4); ctx.calculateUntilOutput(x[:4]); ctx.delete()   # contexts live on the leader
dnn.delete()
g = api.DeviceGroup({mid_model_path!r}, [0])
print("transport", g.weightTransport())
g.delete()
"""
    env = dict(os.environ, FDNN_DEVICES="0,0", FDNN_GROUP_BCAST="rccl")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "transport rccl" in r.stdout, r.stdout + r.stderr[-2000:]
    one = api.QuantizedDnn.loadFromFile(mid_model_path, device=0)
    want = one.calculate(F.synth_features(333, 432, seed=5))
    one.delete()
    assert np.array_equal(np.load(str(tmp_path / "got.npy")), want)
