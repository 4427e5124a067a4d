"""A/B of library builds on the bench net, per kernel class and per call.

  python tools/ab_libs.py [--frames 10000] [--mode gauss] [--reps 2] [--chain 0|1] NAME=path/to/lib.so ...

Every (library, repetition) runs in its own process (FDNN_LIB), the libraries alternate inside a repetition, so that a
box's clock drift falls on all of them alike.  Printed per run: the time of one pass as back-to-back calls on one
stream (no HIP events inside), the per-class HIP-event times of the profile hooks (layer 0 / hidden / output; each
bracketed launch carries ~4 us of event overhead), and a checksum of the probabilities (a variant that changes a bit
shows up here before it shows up in a test).
"""
import argparse, json, os, subprocess, sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r"""
import os, sys, json, time, hashlib
sys.path.insert(0, os.environ["AB_ROOT"])
import numpy as np, torch
from fast_dnn_amd import api, formats as F
n, mode = int(os.environ["AB_FRAMES"]), os.environ["AB_MODE"]
p = "/tmp/fdnn_ab_%s.bin" % mode
if not os.path.exists(p):
    F.write_model_bin(p + ".tmp", F.synth_net(F.NET_TOPOLOGY, seed=1, mode=mode)); os.replace(p + ".tmp", p)
dnn = api.QuantizedDnn.loadFromFile(p)
if os.environ.get("AB_CHAIN_FORCE"):  # 1: the chained hidden layers wherever the shape allows, 0: never
    api.set_chain(int(os.environ["AB_CHAIN_FORCE"]), 1)
x = torch.from_numpy(F.synth_features(n, 432, seed=5)).cuda()
out = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
t_end = time.time() + 0.5
while time.time() < t_end:
    for _ in range(20): dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
K = int(os.environ.get("AB_STEPS", "200"))
best = 1e9
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / K)
dnn.profileBegin()
for _ in range(50): dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
torch.cuda.synchronize()
prof = dnn.profileEnd()
o = out.cpu().numpy()
print(json.dumps({"step_us": round(best * 1e6, 1), **{k: round(v["ms"] / 50 * 1e3, 1) for k, v in prof.items() if v["launches"]},
                  "sha": hashlib.sha256(o.tobytes()).hexdigest()[:12]}))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=10000)
    ap.add_argument("--mode", default="gauss")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--chain", default=None, help="FDNN_CHAIN for the runs (0: a launch per hidden layer, 1: the default rule)")
    ap.add_argument("--chain-force", default=None, help="comma list, e.g. 0,1: every library is run once per value (api.set_chain)")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    libs = [l.split("=", 1) if "=" in l else (os.path.basename(l), l) for l in a.libs]
    if a.chain_force:
        libs = [(f"{name}/chain{v}", path + "|" + v) for name, path in libs for v in a.chain_force.split(",")]
    res = {name: [] for name, _ in libs}
    for rep in range(a.reps):
        for name, path in libs:
            force = None
            if "|" in path:
                path, force = path.split("|")
            env = dict(os.environ, FDNN_LIB=os.path.abspath(path), AB_ROOT=ROOT, AB_FRAMES=str(a.frames), AB_MODE=a.mode)
            if force is not None:
                env["AB_CHAIN_FORCE"] = force
            if a.chain is not None:
                env["FDNN_CHAIN"] = a.chain
            r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
            try:
                d = json.loads(line)
            except Exception:
                d = {"error": (r.stderr or r.stdout)[-400:]}
            res[name].append(d)
            print(f"rep {rep} {name:>14s} {a.mode} n={a.frames}: {d}", flush=True)
    print("--- best step_us per library")
    for name, runs in res.items():
        ok = [r for r in runs if "step_us" in r]
        if ok:
            b = min(ok, key=lambda r: r["step_us"])
            print(f"{name:>14s}: {b}")


if __name__ == "__main__":
    main()
