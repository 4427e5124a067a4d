"""Sustained throughput of the multi-stream scoring loop (fdnn_server_*).
  leg 1: device-resident 10 000-frame batches, in-flight depth 1..4, against back-to-back
         fdnn_calculate_device on one stream;
  leg 2: the serving shape -- T host threads each scoring 100-frame utterances (1 s of speech)
         through coalesced host submissions, against per-call fdnn_calculate from the same threads."""
import json, os, sys, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fast_dnn_amd import api, formats as F

p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "fdnn_net_seed1_gauss.bin")
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
O = dnn.outputDimension()
res = {}
n = int(os.environ.get("FRAMES", "10000")); K = int(os.environ.get("STEPS", "200"))
x = torch.from_numpy(F.synth_features(n, 432, seed=1000)).cuda()
outs = [torch.empty((n, O), dtype=torch.float32, device="cuda") for _ in range(4)]
s = torch.cuda.current_stream().cuda_stream
for _ in range(300): dnn.calculate_device(x.data_ptr(), n, outs[0].data_ptr(), s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K): dnn.calculate_device(x.data_ptr(), n, outs[0].data_ptr(), s)
torch.cuda.synchronize()
res["single_stream_frames_per_s"] = round(n * K / (time.perf_counter() - t0), 1)
for depth in (1, 2, 3, 4):
    srv = api.ScoringServer(dnn, n, depth)
    for i in range(50): srv.submit_device(x.data_ptr(), n, outs[i % depth].data_ptr())
    srv.drain()
    t0 = time.perf_counter()
    for i in range(K): srv.submit_device(x.data_ptr(), n, outs[i % depth].data_ptr())
    srv.drain()
    res[f"server_depth{depth}_frames_per_s"] = round(n * K / (time.perf_counter() - t0), 1)
    srv.close()
print(json.dumps(res), flush=True)

# serving shape
utt = F.synth_features(100, 432, seed=5)
for T in (1, 8, 32, 64):
    per = int(os.environ.get("UTTS", "40"))
    def run(fn):
        th = [threading.Thread(target=fn) for _ in range(T)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        return T * per / (time.perf_counter() - t0)
    def percall():
        for _ in range(per): dnn.calculate(utt)
    run(percall)
    a = run(percall)
    srv = api.ScoringServer(dnn, 6400, 3, int(os.environ.get("LINGER", "150")))
    def viaserver():
        for _ in range(per):
            t, out = srv.submit(utt); srv.wait(t)
    run(viaserver)
    b = run(viaserver)
    st = srv.stats()
    srv.close()
    print(json.dumps({"threads": T, "per_call_utts_per_s": round(a, 1), "server_utts_per_s": round(b, 1),
                      "server_streams_at_realtime": round(b, 1), "batches": st["batches"], "requests": st["requests"]}), flush=True)
dnn.delete()
