#!/usr/bin/env python3
"""bench.py -- acoustic frames/s of the quantized scorer on 1..N MI355X.

A "step" is one pass of the whole hot path (QuantizedDnn.calculate: shift/scale, fp32 layer 0,
six int8 2048x2048 layers, int8 8000x2048 output layer, soft-max) over one 10 000-frame batch per
GPU of the synthetic 7x2048 -> 8000 net (BASELINE.json configs[2]; configs[1]'s "model" file is a
feature batch).  Inputs and outputs are device resident; the weights are
quantized once on rank 0 and broadcast over RCCL at load time only; there is no collective in the
timed region (frames are independent: weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` without a launcher (WORLD_SIZE unset) starts its own N ranks: it re-executes itself under
torch.distributed.run on 127.0.0.1 with a free port and passes the exit code on.  Every N runs the SAME 10 000 frames per
GPU per step; at N > 1 a second value is reported for 125 000 frames per GPU (BASELINE configs[4]: 1 M frames over 8 GPUs).
`--share-device` puts every rank on device 0 (a one-GPU box: the launch, rendezvous, broadcast and timing protocol with a
real scorer; the blob then travels over gloo, RCCL refuses two ranks on one device); `--stub-scorer` runs the same
protocol without any GPU (gloo, host buffers, a scorer that only fills its output) -- the CPU test of the launch logic.

The K timed steps are submitted to the scoring loop (fdnn_server_*, in-flight depth 2: the
HBM-bound soft-max scale of step i runs under the VALU-bound layer 0 of step i+1, every step a
complete pass with its own output buffer) and the clock stops when the last result is complete.
The same K steps back to back on one stream (fdnn_calculate_device) are reported as `single_stream`.

Rank 0 prints ONE JSON line (contract in the task statement) with
  roofline             the kernel class with the largest share of the step, live HIP-event times
  roofline_kernels     all four classes (layer 0, hidden int8 GEMM, output int8 GEMM, soft-max scale)
  roofline_int8_gemm   the hidden-layer int8 GEMM by name (the MFMA kernel the net is made of)
  end_to_end           value against the int8-MFMA ceiling of the whole net (60 M frames/s)
  lazy_40pct           BASELINE configs[3]: LazyContext contract, 40 % mask with 3 % churn, same batch
  small_batch          one 100-frame utterance / one 8-frame block per call, device resident: us per call, weight-stream
                       GB/s against the 8 TB/s HBM figure (the regime of the reference's own callers)
  serving              16 caller threads x 100-frame utterances host-to-host: dense per call / through the scoring loop, the
                       lazy contract per call and through the loop (fdnn_server_submit_lazy_bits), from Python threads and
                       (native_harness) from tools/serve_bench.cpp
  cpu_baseline         the compiled reference itself (oracle/_ref; the oracle's SSE4.1 port as fallback) on this host's cores, N=1 only
Setup (model load, 0.5 s of untimed passes that bring a cold device to its sustained clocks,
reported as `setup.clock_ramp_steps`) comes before the W warm-up steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FRAMES_PER_GPU = 10000
INT8_PEAK_TOPS = 5000.0  # dense int8 MFMA, 2x the 2.5 PF bf16 dense peak (MI355X_MICROARCH.md)
INT8_MEASURED_TOPS = 3944.0  # what a bare int8 MFMA microbenchmark reaches (MI355X_MICROARCH.md, matrix-core table)
FP32_NOFMA_TFLOPS = 78.65  # fp32 vector peak 157.3 counts an fma as two; multiply and add rounded separately -> half
HBM_PEAK_GBS = 8000.0
WEIGHT_BYTES_PER_PASS = 41_549_824 + 3_538_944 + 89_344 + 3_456  # int8 layers + fp32 layer 0 + biases + shift/scale = 45.2 MB
INT8_OPS_PER_FRAME = 83_099_648  # 6*2048^2 + 8000*2048 MAC, x2
INT8_OPS_HIDDEN_LAYERS = 2 * 6 * 2048 * 2048
ROOFLINE_FRAMES_PER_S = INT8_PEAK_TOPS * 1e12 / INT8_OPS_PER_FRAME  # 60.2 M frames/s


def host_info() -> dict:
    """CPU model, physical cores, logical CPUs this process may use, cgroup CPU quota."""
    info = {"logical_cpus": len(os.sched_getaffinity(0))}
    try:
        model, cores = None, set()
        phys = core = None
        allowed = os.sched_getaffinity(0)
        cpu = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                cpu = int(v)
            elif k == "model name" and model is None:
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and cpu is not None:
                if cpu in allowed and phys is not None:
                    cores.add((phys, core))
                phys = core = cpu = None
        info["cpu_model"] = model
        info["physical_cores"] = len(cores) or None
    except OSError:
        pass
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota = None if txt[0] == "max" else float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                quota = None if q < 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            info["cgroup_cpu_quota"] = quota
            break
        except (OSError, ValueError, IndexError):
            continue
    return info


def cpu_baseline(model_path: str, frames: int = 100) -> dict:
    """The reference on this host's cores, next to the GPU number.  Where oracle/_ref is present (the reference's own
    dnn.cc / float_dnn.cc compiled by oracle/Makefile; it travels with the repo as a .so) the figure reported as
    `value` is THE REFERENCE ITSELF (kind "reference"): a native pthread harness (oracle/ref_tap.cpp:
    ref_bench_threads), one thread per usable core, every thread scoring independent 100-frame utterances with a
    fresh CalculationContext per call (jni_dnn.cc:49-51; concurrency model of MultiThreadedStressTest.java:48-61;
    timed region = context construction + Calculate as in the reference CLI, dnn.cc:64-71).  The oracle's SSE4.1
    port (oracle/fdnn_oracle.c: orc_bench_threads, frame block 8) is timed the same way beside it and is the
    fallback (kind "port") where the compiled reference is absent.  No interpreter in any timed region."""
    import ctypes as C

    from fast_dnn_amd import formats as F
    from oracle.oracle import Oracle

    orc = Oracle(model_path)
    L = Oracle.lib()
    L.orc_bench_threads.restype = C.c_double
    L.orc_bench_threads.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    x = F.synth_features(frames, seed=900)
    xp = x.ctypes.data_as(C.POINTER(C.c_float))
    dim = int(x.shape[1])

    def run_port(threads, utts):
        per = (C.c_double * threads)()
        wall = L.orc_bench_threads(orc.h, xp, frames, 8, 1, threads, utts, per)
        if wall <= 0:
            raise RuntimeError("oracle thread harness failed")
        return wall, list(per)

    host = host_info()
    usable = host.get("physical_cores") or host["logical_cpus"]
    if host.get("cgroup_cpu_quota"):
        usable = max(1, min(usable, int(host["cgroup_cpu_quota"])))
    usable = min(usable, host["logical_cpus"])
    utts = 2

    def measure(run):
        run(1, 1)  # warm-up
        one = sorted(frames / run(1, 1)[0] for _ in range(3))
        # the host is shared (cgroup quota, other tenants): three passes, min / median / max all in the line
        passes = sorted((run(usable, utts) for _ in range(3)), key=lambda r: r[0])
        rates = sorted(usable * utts * frames / w for w, _ in passes)
        per = passes[1][1]
        return {"one": one, "rates": rates, "slow": max(per), "fast": min(per)}

    port = measure(run_port)
    ref = ref_diff = None
    try:
        from oracle.oracle import RefLib

        lib = RefLib()
        lib.L.ref_bench_threads.restype = C.c_double
        lib.L.ref_bench_threads.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        rm = lib.load(model_path)

        def run_ref(threads, utts_):
            per = (C.c_double * threads)()
            wall = lib.L.ref_bench_threads(rm.h, xp, frames, dim, 8, threads, utts_, per)
            if wall <= 0:
                raise RuntimeError("reference thread harness failed")
            return wall, list(per)

        ref = measure(run_ref)
        ref_diff = float(np.abs(rm.calculate(x, 8) - orc.calculate(x, 8)).max())  # the two agree on this very sample
        rm.close()
    except (OSError, FileNotFoundError, AttributeError):
        ref = None
    main_ = ref if ref is not None else port
    r3 = lambda v: round(float(v), 1)
    out = {
        "value": r3(main_["rates"][1]), "unit": "frames/s", "cores": usable, "kind": "reference" if ref is not None else "port",
        "value_min_median_max": [r3(v) for v in main_["rates"]],
        "value_1thread": r3(main_["one"][1]), "value_1thread_min_median_max": [r3(v) for v in main_["one"]],
        "scaling_vs_1thread": round(main_["rates"][1] / main_["one"][1], 2),
        "slowest_thread_s": round(main_["slow"], 3), "fastest_thread_s": round(main_["fast"], 3),
        "port_value": r3(port["rates"][1]), "port_value_min_median_max": [r3(v) for v in port["rates"]],
        "port_value_1thread": r3(port["one"][1]),
        "reference_value": None if ref is None else r3(ref["rates"][1]),
        "reference_value_1thread": None if ref is None else r3(ref["one"][1]),
        "port_vs_reference_1thread": None if ref is None else round(port["one"][1] / ref["one"][1], 3),
        "reference_vs_port_max_abs_diff": ref_diff,
        "host": host,
        "sample": f"{'the compiled reference (oracle/_ref: dnn.cc + float_dnn.cc, -O2 -msse4 -ffp-contract=off)' if ref is not None else 'oracle SSE4.1 port (pmaddubsw, frame-block 8)'}"
                  f" on the same net, native pthread harness: 3 x one {frames}-frame utterance on 1 thread, then 3 passes of {usable * utts} "
                  f"utterances on {usable} threads (one per physical core within the cgroup quota; min / median / max given, median = value), "
                  f"a fresh context per call; every thread streams the 45 MB of weights once per 8-frame block, so the multi-thread "
                  f"figure is bound by shared cache / memory bandwidth, not by core count",
    }
    return out


def self_launch(n_ranks: int) -> int:
    """`python bench.py --gpus N` with no launcher around it: re-execute under torch.distributed.run, one rank per GPU on
    127.0.0.1 (the container hostname may not resolve) with a free port; returns the launcher's exit code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def timed_steps(submit, drain, fence, warmup: int, steps: int) -> float:
    """The contract's timed region: W untimed steps, then EXACTLY K steps bracketed by fence() (barrier + device sync)."""
    for i in range(warmup):
        submit(i)
    drain()
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        submit(i)
    drain()
    fence()
    return time.perf_counter() - t0


def max_over_ranks(dist, world: int, elapsed: float, dev) -> float:
    if world == 1:
        return elapsed
    import torch

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def rank_report(dist, world: int, rank: int, digest: str, own_frames_per_s: float, device_name: str, device_key: str = "", shared: bool = False,
                peers=None) -> dict:
    """What the driver can check about an N-rank run: how many ranks the process group really has and on which backend,
    that every rank holds the same weight blob (sha256 of what the rank's model exports), each rank's own rate, that the
    ranks sit on DIFFERENT devices (unless --share-device), and which of the other devices rank 0's can reach peer to peer
    (hipDeviceCanAccessPeer: the xGMI links the one-process group path copies the blob over)."""
    grouped = dist is not None and dist.is_available() and dist.is_initialized()
    if world == 1:
        backend = dist.get_backend() if grouped else None
        return {"ranks": 1, "collective_backend": backend, "rccl_ranks": dist.get_world_size() if backend == "nccl" else 0,
                "collective_exercised": bool(grouped), "blob_sha256": digest, "blob_sha256_all_equal": True,
                "per_rank_frames_per_s": {"min": round(own_frames_per_s, 1), "max": round(own_frames_per_s, 1)}, "devices": [device_name],
                "peer_access_from_rank0": peers}
    parts = [None] * world
    dist.all_gather_object(parts, (rank, digest, own_frames_per_s, device_name, device_key))
    backend = dist.get_backend()
    rates = [p[2] for p in parts]
    keys = [p[4] for p in parts]
    if not shared:
        assert len(set(keys)) == world, f"{world} ranks on {len(set(keys))} distinct device(s) {keys}: one rank per GPU, or --share-device"
    return {"ranks": dist.get_world_size(), "collective_backend": backend,
            "rccl_ranks": dist.get_world_size() if backend == "nccl" else 0, "collective_exercised": True,
            "blob_sha256": parts[0][1], "blob_sha256_all_equal": len({p[1] for p in parts}) == 1,
            "per_rank_frames_per_s": {"min": round(min(rates), 1), "max": round(max(rates), 1), "all": [round(r, 1) for r in rates]},
            "devices": [p[3] for p in parts], "distinct_devices": len(set(keys)), "peer_access_from_rank0": peers}


def numa_cpus_of_device(local: int):
    """CPUs next to GPU `local` (sysfs local_cpulist of its PCI function), or None: the host-fed callers of a rank are kept on
    the socket its GPU hangs off (per-rank pinned buffers + NUMA-local callers: SURVEY 8(e))."""
    try:
        import torch

        bdf = torch.cuda.get_device_properties(local).pci_bus_id if hasattr(torch.cuda.get_device_properties(local), "pci_bus_id") else None
        if not bdf:
            return None
        txt = open(f"/sys/bus/pci/devices/{str(bdf).lower()}/local_cpulist").read().strip()
        cpus = set()
        for part in txt.split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0)
        return sorted(cpus & allowed) or None
    except Exception:  # noqa: BLE001
        return None


def host_fed_leg(make_caller, threads: int, per: int, frames_per_utt: int, fence, dist, world: int, red_dev, cpus=None) -> dict:
    """N > 1's host-fed shape (the reference's concurrency model, MultiThreadedStressTest.java:48-69: independent caller
    threads over one immutable model, the result rows copied out per call, jni_dnn.cc:54-57): on EVERY rank `threads` caller
    threads score `per` utterances of `frames_per_utt` frames each, host frames in, host soft-max rows out; one untimed
    round, then one round bracketed by fence() on both sides, max over ranks.  make_caller(t) -> the thread's body."""
    import threading

    def one_round():
        th = [threading.Thread(target=make_caller(t)) for t in range(threads)]
        for h in th:
            h.start()
        for h in th:
            h.join()

    old_aff = None
    if cpus:
        try:
            old_aff = os.sched_getaffinity(0)
            os.sched_setaffinity(0, cpus)  # (threads started from here inherit it)
        except Exception:  # noqa: BLE001
            old_aff = None
    one_round()
    fence()
    t0 = time.perf_counter()
    one_round()
    fence()
    own = time.perf_counter() - t0
    if old_aff is not None:
        os.sched_setaffinity(0, old_aff)
    el = max_over_ranks(dist, world, own, red_dev)
    own_rate = threads * per / own
    rates = [own_rate]
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, own_rate)
        rates = parts
    return {"utterances_per_s_whole_node": round(world * threads * per / el, 1),
            "frames_per_s_whole_node": round(world * threads * per * frames_per_utt / el, 1),
            "per_rank_utterances_per_s": [round(r, 1) for r in rates],
            "caller_threads_per_rank": threads, "utterances_per_thread": per, "frames_per_utterance": frames_per_utt,
            "numa_local_cpus": None if not cpus else len(cpus)}


def stub_main(args, rank: int, world: int) -> None:
    """--stub-scorer: the N-rank protocol of main() without a GPU.  gloo, host tensors; rank 0 quantizes and packs the tiny
    model with the library's host half (no device needed), the blob is broadcast and validated on every rank, the timed
    region runs a scorer that only fills its output; same fences, same max-over-ranks reduction, same multi-rank keys."""
    import hashlib

    import torch
    import torch.distributed as dist

    from fast_dnn_amd import api, dist as fd, formats as F

    dev = torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    tmp = os.environ.get("TMPDIR", "/tmp")
    model_path = os.path.join(tmp, "fdnn_stub_tiny.bin")
    if rank == 0:
        F.ensure_model_file(model_path, [432, 64, 64, 64, 100], seed=3, mode="gauss")
    blob = torch.from_numpy(api.HostModel(model_path).blob()) if rank == 0 else None
    blob = fd.broadcast_blob(blob, rank, world, dev)
    info = api.host_blob_check(blob.numpy())
    digest = hashlib.sha256(blob.numpy().tobytes()).hexdigest()
    n = args.frames if args.frames > 0 else 256
    O = info["output_dim"]
    out = np.empty((n, O), dtype=np.float32)

    def submit(_i):
        out.fill(1.0 / O)

    def fence():
        if world > 1:
            dist.barrier()

    own = timed_steps(submit, lambda: None, fence, args.warmup, args.steps)
    elapsed = max_over_ranks(dist, world, own, dev)
    rep = rank_report(dist, world, rank, digest, n * args.steps / own, "cpu (stub)", device_key="cpu", shared=True)
    # the two host-fed legs of the N-rank line, with the stub scorer: per-rank caller threads, and one process over N "devices"
    bufs = [np.empty((100, O), dtype=np.float32) for _ in range(4)]

    def make_caller(t):
        def body():
            for _ in range(10):
                bufs[t].fill(1.0 / O)
        return body

    host_fed = host_fed_leg(make_caller, 4, 10, 100, fence, dist, world, dev)
    one_process = None
    if rank == 0:
        t0 = time.perf_counter()
        for _ in range(3):
            for _d in range(world):
                out.fill(1.0 / O)
        one_process = {"devices": world, "frames_per_s_whole_node": round(3 * world * n / (time.perf_counter() - t0), 1), "stub": True}
    fence()
    if rank == 0:
        print(json.dumps({
            "metric": "acoustic frames/sec (whole node), 7x2048->8000 nnet", "value": round(world * n * args.steps / elapsed, 1),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "none (stub scorer)", "data": "synthetic",
            "config": {"workload": "STUB: launch / rendezvous / broadcast / timing protocol only, no scorer, no GPU", "frames_per_gpu": n,
                       "global_frames": world * n, "parallelism": f"frame-sharded x{world}, replicated weights"},
            "multi_gpu": rep, "host_fed": host_fed, "one_process_group": one_process, "stub": True}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: ~0.2 s of GPU time per pass; short runs (20 steps) read low without the clock ramp below
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=0,
                    help="frames per GPU per step; default 10 000 at every N (BASELINE configs[2]), so that the 1 -> N ratio "
                         "compares like with like; at N > 1 the 125 000-frame shard of configs[4] is timed as a second value "
                         "(`config4_125k_per_gpu`, skip with --no-config4)")
    ap.add_argument("--no-config4", action="store_true", help="N > 1: skip the 125 000-frames-per-GPU leg")
    ap.add_argument("--share-device", action="store_true",
                    help="every rank on device 0 (one-GPU box): weights broadcast over gloo, unfused soft-max (two processes on "
                         "one GPU must not spin on each other's workgroups, INTEGRATION.md)")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1: initialise the nccl (= RCCL) process group with ONE rank and take the weights through the real broadcast + import "
                         "path of the N > 1 runs (multi_gpu.rccl_ranks = 1, collective_exercised): proves librccl loads and device-tensor "
                         "collectives work on this box")
    ap.add_argument("--stub-scorer", action="store_true",
                    help="no GPU: gloo, host buffers and a scorer that only fills its output -- exercises launch, rendezvous, "
                         "blob broadcast + hash, barriers, max-over-ranks timing and the JSON line (tests/test_bench_launch.py)")
    ap.add_argument("--mode", default="gauss", choices=["gauss", "nosat"], help="synthetic weight distribution")
    ap.add_argument("--in-flight", type=int, default=2, help="steps in flight in the scoring loop (1 = no overlap between steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lazy", action="store_true", help="skip the configs[3] leg")
    ap.add_argument("--no-nosat", action="store_true", help="skip the pair-free-net extra")
    ap.add_argument("--no-small", action="store_true", help="skip the small-batch (100- and 8-frame call) leg")
    ap.add_argument("--no-serving", action="store_true", help="skip the 100-frame-utterance serving leg")
    ap.add_argument("--host-fed", action="store_true", help="N = 1: also run the N > 1 host-fed legs (per-rank caller threads, one-process group)")
    ap.add_argument("--clock-ramp-s", type=float, default=0.5, help="seconds of untimed load before the W warm-up steps (setup)")
    ap.add_argument("--single-stream-only", action="store_true",
                    help="profiling runs (tools/profile_round.sh): only the back-to-back single-stream steps, so that rocprofv3's "
                         "per-kernel averages are not mixed with the overlapped legs")
    ap.add_argument("--l0-fma", action="store_true",
                    help="layer 0 with the fused multiply-add numerics of a -march=native reference build (fp32 MFMA)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))  # start the N ranks ourselves
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.stub_scorer:
        stub_main(args, rank, world)
        return
    if args.share_device:
        os.environ["FDNN_FUSE_NORM"] = "0"  # before the library is loaded: several processes on one GPU take the scale pass

    import hashlib

    import torch
    import torch.distributed as dist

    from fast_dnn_amd import api, formats as F

    if not torch.cuda.is_available() or api.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the scorer has no CPU path")
    if args.share_device:
        local = 0
    elif local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: no device {local} on this box ({torch.cuda.device_count()} visible); --share-device puts every rank on device 0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_coll = bool(args.force_collective) and world == 1
    if world > 1 or force_coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_coll and "MASTER_PORT" not in os.environ:
            import socket

            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.share_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from fast_dnn_amd.dist import load_replicated

    tmp = os.environ.get("TMPDIR", "/tmp")
    model_path = os.path.join(tmp, f"fdnn_net_seed1_{args.mode}.bin")
    if rank == 0:
        F.ensure_model_file(model_path, F.NET_TOPOLOGY, seed=1, mode=args.mode)
    dnn = load_replicated(model_path, local, rank, world, host_broadcast=args.share_device, force_collective=force_coll)
    # what THIS rank's model holds after the broadcast, hashed on the host (compared across ranks below)
    nb = dnn.blobSize()
    held = torch.empty(nb, dtype=torch.uint8, device=dev)
    dnn.exportBlob(held.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    blob_digest = hashlib.sha256(held.cpu().numpy().tobytes()).hexdigest()
    del held
    O = dnn.outputDimension()
    n = args.frames if args.frames > 0 else FRAMES_PER_GPU
    depth = max(1, args.in_flight)
    if args.l0_fma:
        dnn.setInputLayerFma(True)

    x = torch.from_numpy(F.synth_features(n, 432, seed=1000 + rank)).to(dev)
    outs = [torch.empty((n, O), dtype=torch.float32, device=dev) for _ in range(depth)]
    stream = torch.cuda.current_stream()
    srv = api.ScoringServer(dnn, n, depth)

    def step_single():
        dnn.calculate_device(x.data_ptr(), n, outs[0].data_ptr(), stream.cuda_stream)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Setup, before the contract's W warm-up steps: bring the device to its sustained clocks.  A cold
    # MI355X needs some tens of milliseconds of load to ramp up; without this a short run (K = 5..20)
    # reads 10-20 % below the steady state that K = 200 measures.  Reported as setup.clock_ramp_steps.
    ramp_steps = 0
    ramp_t0 = time.perf_counter()
    while time.perf_counter() - ramp_t0 < args.clock_ramp_s:
        for _ in range(10):
            step_single()
        torch.cuda.synchronize()
        ramp_steps += 10

    if args.single_stream_only:
        for _ in range(args.warmup):
            step_single()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_single()
        fence()
        el = time.perf_counter() - t0
        if rank == 0:
            print(json.dumps({"mode": "single-stream-only", "frames_per_s": round(world * n * args.steps / el, 1),
                              "ms_per_step": round(el / args.steps * 1e3, 4), "steps": args.steps, "frames_per_gpu": n}), flush=True)
        srv.close()
        dnn.delete()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- the timed region: K complete passes through the scoring loop
    own_elapsed = timed_steps(lambda i: srv.submit_device(x.data_ptr(), n, outs[i % depth].data_ptr()), srv.drain, fence, args.warmup, args.steps)
    red_dev = torch.device("cpu") if args.share_device else dev  # gloo reduces host tensors
    elapsed = max_over_ranks(dist, world, own_elapsed, red_dev)
    props = torch.cuda.get_device_properties(local)
    dev_key = f"{local}:{getattr(props, 'uuid', '')}:{getattr(props, 'pci_bus_id', '')}"
    peers = None
    if rank == 0:
        peers = {str(j): bool(torch.cuda.can_device_access_peer(local, j)) for j in range(torch.cuda.device_count()) if j != local}
    multi = rank_report(dist, world, rank, blob_digest, n * args.steps / own_elapsed,
                        f"{torch.cuda.get_device_name(local)} #{local}", device_key=dev_key, shared=bool(args.share_device), peers=peers)
    multi["shared_device"] = bool(args.share_device)
    if world > 1 and not args.share_device:
        assert multi["rccl_ranks"] == world, f"RCCL process group has {multi['rccl_ranks']} ranks, {world} expected"
    assert multi["blob_sha256_all_equal"], "ranks hold different weight blobs"

    # sanity on the last outputs: soft-max rows sum to one, every in-flight buffer holds the same result
    row_sum = float(outs[0][:64].sum(1).mean().item())
    if not os.environ.get("FDNN_BENCH_NOCHECK"):  # (set only for kernel-ablation timing builds)
        assert abs(row_sum - 1.0) < 1e-3, row_sum
        for o in outs[1:]:
            assert torch.equal(o, outs[0]), "in-flight steps disagree"

    # ---- the same K steps back to back on one stream
    for _ in range(args.warmup):
        step_single()
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step_single()
    fence()
    single_elapsed = time.perf_counter() - t1

    # ---- N > 1: BASELINE configs[4]'s shard (1 M frames over 8 GPUs = 125 000 per GPU) as a second value
    config4 = None
    if world > 1 and not args.no_config4 and not args.share_device:
        n4 = 125000
        k4 = max(5, args.steps // 10)
        x4 = torch.from_numpy(F.synth_features(n4, 432, seed=2000 + rank)).to(dev)
        o4 = torch.empty((n4, O), dtype=torch.float32, device=dev)
        own4 = timed_steps(lambda i: dnn.calculate_device(x4.data_ptr(), n4, o4.data_ptr(), stream.cuda_stream), lambda: None, fence, 2, k4)
        el4 = max_over_ranks(dist, world, own4, red_dev)
        assert abs(float(o4[:64].sum(1).mean().item()) - 1.0) < 1e-3
        config4 = {"workload": f"BASELINE configs[4]: {world * n4} frames per step sharded over {world} GPUs ({n4} per GPU), "
                               "fdnn_calculate_device back to back on one stream per rank",
                   "frames_per_s": round(world * n4 * k4 / el4, 1), "ms_per_step": round(el4 / k4 * 1e3, 4), "steps": k4,
                   "frames_per_gpu": n4}
        del x4, o4

    # ---- N > 1: the HOST-FED shapes, beside the device-resident figure (which scales by construction).  SURVEY 8(e): what
    # threatens ">= 6x at 8 GPUs" is feeding -- every GPU's 32 KB per frame back to one host's memory.  (a) per rank: caller
    # threads over 100-frame utterances through the rank's scoring loop (fdnn_server_submit + wait), host rows in and out, the
    # callers kept on the GPU's socket; (b) ONE process over all N devices (fdnn_group_calculate, FDNN_DEVICES): rank 0 only,
    # the other ranks idle at a barrier.  Neither is `value`.
    host_fed, one_process = None, None
    if (world > 1 or args.host_fed) and not args.no_serving:
        T_hf, per_hf, uf_hf = 8, 40, 100
        utt_hf = F.synth_features(uf_hf, 432, seed=5 + rank)
        bufs_hf = [np.zeros((uf_hf, O), dtype=np.float32) for _ in range(T_hf)]
        hsrv = api.ScoringServer(dnn, 6400, 3, 100)

        def make_caller(t):
            def body():
                for _ in range(per_hf):
                    tk, _o = hsrv.submit(utt_hf, out=bufs_hf[t])
                    hsrv.wait(tk)
            return body

        host_fed = host_fed_leg(make_caller, T_hf, per_hf, uf_hf, fence, dist, world, red_dev, cpus=numa_cpus_of_device(local))
        hsrv.close()
        assert abs(float(bufs_hf[0].sum(1).mean()) - 1.0) < 1e-3
        host_fed["device_to_host_GB_per_s_whole_node"] = round(host_fed["frames_per_s_whole_node"] * O * 4 / 1e9, 1)
        host_fed["note"] = ("per rank: fdnn_server_submit + wait from caller threads pinned to the GPU's socket, 100-frame utterances, host frames in / "
                            "host soft-max rows out (32 KB per frame); whole node = all ranks' utterances over the slowest rank's time")
        if not args.share_device:
            n_dev = torch.cuda.device_count() if world > 1 else 1
            if rank == 0:
                try:
                    grp = api.DeviceGroup(model_path, list(range(min(n_dev, world))))
                    ng = grp.size() * 2000  # 2 000 frames per device per call: 64 MB of rows back per device
                    xg = F.synth_features(ng, 432, seed=77)
                    grp.calculate(xg[: grp.size() * 200])
                    tg = time.perf_counter()
                    reps_g = 3
                    for _ in range(reps_g):
                        pg = grp.calculate(xg)
                    eg = time.perf_counter() - tg
                    assert abs(float(pg[:8].sum(1).mean()) - 1.0) < 1e-3
                    one_process = {"devices": grp.size(), "frames_per_call": ng, "frames_per_s_whole_node": round(reps_g * ng / eg, 1),
                                   "weight_transport": grp.weightTransport(),
                                   "note": "api.DeviceGroup.calculate = fdnn_group_calculate: one process, one worker thread per device, "
                                           "contiguous frame shards, host frames in / host rows out"}
                    grp.delete()
                except Exception as e_:  # noqa: BLE001
                    one_process = {"error": str(e_)[:300]}
            fence()

    # ---- per-kernel HIP events over the same K steps (single stream; rank 0 reports)
    dnn.profileBegin()
    for _ in range(args.steps):
        step_single()
    torch.cuda.synchronize()
    prof = dnn.profileEnd()

    # ---- the other layer-0 flavour (rank 0, N=1 only, informational)
    alt = None
    if world == 1:
        dnn.setInputLayerFma(not args.l0_fma)
        for _ in range(args.warmup):
            step_single()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            step_single()
        torch.cuda.synchronize()
        alt = n * args.steps / (time.perf_counter() - t2)
        dnn.setInputLayerFma(args.l0_fma)

    # ---- the same topology WITHOUT saturating weight pairs (a trained, heavy-tailed net has few: SURVEY 7 hard part 2): what
    # the Gaussian bench net's pair screens cost shows as the difference to the headline's single-stream figure
    nosat = None
    if world == 1 and args.mode == "gauss" and not args.no_nosat:
        try:
            np_path = os.path.join(tmp, "fdnn_net_seed1_nosat.bin")
            F.ensure_model_file(np_path, F.NET_TOPOLOGY, seed=1, mode="nosat")
            dn = api.QuantizedDnn.loadFromFile(np_path)
            for _ in range(max(5, args.warmup)):
                dn.calculate_device(x.data_ptr(), n, outs[0].data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            tn = time.perf_counter()
            for _ in range(args.steps):
                dn.calculate_device(x.data_ptr(), n, outs[0].data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            en = time.perf_counter() - tn
            nosat = {"frames_per_s_single_stream": round(n * args.steps / en, 1), "ms_per_step": round(en / args.steps * 1e3, 4),
                     "vs_headline_net_single_stream": round((single_elapsed / args.steps) / (en / args.steps), 4),
                     "note": "same 432 -> 7x2048 -> 8000 topology and seed, weights quantised so that no adjacent pair can leave int16 "
                             "(|w_q| <= 64): the layers run the instances without the pair-saturation walk"}
            # ... and, on this net, the round-6 role-split fused output kernel (fdnn_ppo.hip; selectable, bit-identical, not the
            # default): the same pass with it switched on, per-class device time of the output layer from the profiling scopes
            try:
                def _out_us(ppo_mode):
                    api.set_ppo(ppo_mode)
                    for _ in range(5):
                        dn.calculate_device(x.data_ptr(), n, outs[0].data_ptr(), stream.cuda_stream)
                    torch.cuda.synchronize()
                    dn.profileBegin()
                    t_ = time.perf_counter()
                    for _ in range(20):
                        dn.calculate_device(x.data_ptr(), n, outs[0].data_ptr(), stream.cuda_stream)
                    torch.cuda.synchronize()
                    e_ = time.perf_counter() - t_
                    pr_ = dn.profileEnd()
                    return round(pr_["output_gemm"]["ms"] / 20 * 1e3, 1), round(e_ / 20 * 1e3, 4)

                in_phase = _out_us(0)
                split = _out_us(1)
                nosat["role_split_output_kernel"] = {"output_layer_us": split[0], "in_phase_output_layer_us": in_phase[0], "ms_per_step": split[1],
                                                     "in_phase_ms_per_step": in_phase[1], "give_ups": int(dn.fuseGiveups()),
                                                     "note": "fdnn_debug_set_ppo(1) vs (0), 20 passes each under the library's profiling scopes (which add a few us per pass)"}
            except Exception as e2_:  # noqa: BLE001
                nosat["role_split_output_kernel"] = {"error": str(e2_)[:200]}
            finally:
                api.set_ppo(-1)
            dn.delete()
        except Exception as e_:  # noqa: BLE001
            nosat = {"error": str(e_)[:200]}

    # ---- BASELINE configs[3]: the lazy contract, 40 % of the output nodes active, 3 % churn per frame
    lazy = None
    steps_for_ratio = args.steps
    if world == 1 and not args.no_lazy:
        masks = F.generate_masks_fast(n, O, 0.40, 0.03, seed=11)
        active = float(masks.mean())
        md = torch.from_numpy(masks).to(dev)
        k_lazy = args.steps  # the same K as the dense leg
        for i in range(args.warmup):
            srv.submit_device(x.data_ptr(), n, outs[i % depth].data_ptr(), md.data_ptr())
        srv.drain()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for i in range(k_lazy):
            srv.submit_device(x.data_ptr(), n, outs[i % depth].data_ptr(), md.data_ptr())
        srv.drain()
        torch.cuda.synchronize()
        lazy_s = (time.perf_counter() - t3) / k_lazy
        off = md[:64] == 0
        o64 = outs[(k_lazy - 1) % depth][:64]
        lo = torch.where(off, o64, torch.full_like(o64, float("inf"))).min(1).values
        hi = torch.where(off, o64, torch.full_like(o64, float("-inf"))).max(1).values
        assert bool((lo == hi).all()), "masked-out nodes must all read 1/total"
        masked_ops = INT8_OPS_HIDDEN_LAYERS + 2 * 2048 * O * active  # count masked work only
        lazy = {
            "workload": f"BASELINE configs[3]: same net and batch, LazyContext contract, masks with {active:.3f} of the {O} output "
                        f"nodes active and 3 % churn per frame (FuncTest.java:121-133 statistics), masks device resident",
            "frames_per_s": round(n / lazy_s, 1), "ms_per_step": round(lazy_s * 1e3, 4), "steps": k_lazy,
            "int8_ops_per_frame_masked_work": int(masked_ops),
            "int8_tops_masked_work": round(masked_ops * n / lazy_s / 1e12, 1),
            "frac_of_int8_peak_masked_work": round(masked_ops * n / lazy_s / 1e12 / INT8_PEAK_TOPS, 4),
            "note": "the output GEMM runs dense and masks in its epilogue: over a 320-frame tile the union of the per-frame "
                    "masks covers ~all nodes (DESIGN.md, mask-union density), so row compaction has nothing to drop; the caller's "
                    "byte masks (80 MB per step) are packed to bits by one HBM pass (mask_pack_kernel) and the GEMM reads one "
                    "64-bit word per frame row and 64 nodes",
        }
        # the same contract with the masks as bits (fdnn_ctx_lazy_output_batch_bits_device): no 80 MB of mask bytes, no pack pass
        bd = torch.from_numpy(F.pack_mask_bits(masks).view(np.int64)).to(dev)
        ctx = dnn.getNewLazyContext(n)
        lz = outs[0]

        def lazy_bits_step():
            ctx.calculateUntilOutputDevice(x.data_ptr(), stream.cuda_stream)
            ctx.calculateForOutputNodesBatchBitsDevice(bd.data_ptr(), lz.data_ptr(), 0, n, stream.cuda_stream)

        for _ in range(args.warmup):
            lazy_bits_step()
        torch.cuda.synchronize()
        t3b = time.perf_counter()
        for _ in range(k_lazy):
            lazy_bits_step()
        torch.cuda.synchronize()
        bits_s = (time.perf_counter() - t3b) / k_lazy
        md_ref = torch.empty_like(lz)
        ctx.calculateForOutputNodesBatchDevice(md.data_ptr(), md_ref.data_ptr(), 0, n, stream.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(md_ref, lz), "bit-mask and byte-mask results differ"
        ctx.delete()
        lazy["bit_masks"] = {"frames_per_s": round(n / bits_s, 1), "ms_per_step": round(bits_s * 1e3, 4), "steps": k_lazy,
                             "vs_dense_single_stream": round((single_elapsed / steps_for_ratio) / bits_s, 4),
                             "note": "LazyContext on one stream: calculateUntilOutput + fdnn_ctx_lazy_output_batch_bits_device, masks as "
                                     "uint64 [n][125] (10 MB), device resident; compared with the dense single-stream step"}
        del md, masks, bd, md_ref

    # ---- small batches, device resident: one utterance (100 frames = 1 s of speech, the reference's own call shape) and
    # a decoder-sized block (8 frames) per call, calls back to back on one stream.  The regime is bound by streaming the
    # 45.2 MB of weights once per call: reported against the 8 TB/s HBM figure.
    small = None
    if world == 1 and not args.no_small:
        small = {"weights_bytes_per_call": WEIGHT_BYTES_PER_PASS, "hbm_peak_GB_per_s": HBM_PEAK_GBS, "calls": []}
        for sn in (100, 8):
            reps = 1500
            for _ in range(100):
                dnn.calculate_device(x.data_ptr(), sn, outs[0].data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for _ in range(reps):
                dnn.calculate_device(x.data_ptr(), sn, outs[0].data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t4) / reps * 1e6
            gbs = WEIGHT_BYTES_PER_PASS / (us * 1e-6) / 1e9
            small["calls"].append({"frames": sn, "us_per_call": round(us, 2), "frames_per_s": round(sn / us * 1e6, 1),
                                   "weight_stream_GB_per_s": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                                   "x_realtime_one_stream": round(sn / us * 1e6 / 100.0, 1)})
        assert abs(float(outs[0][:8].sum(1).mean().item()) - 1.0) < 1e-3
        # An honest bound for this regime (round 5 reported "0.09 of the HBM peak": the weights do not come from HBM -- 45 MB sit
        # in the 256 MB Infinity Cache between calls -- and one workgroup's operand stream, not the chip's, is what a launch waits
        # for).  Per launch: the bytes ONE workgroup pulls through its CU's L2 -> LDS path at the 42 B/clk/CU that path delivers
        # (tools/ubench_dma_waves.hip), plus the stream's kernel boundary (1.45 us between trivial dependent kernels,
        # MI355X_MICROARCH.md price list).  100 frames: hidden layer = 64 KB of weights + 64 KB of activation rows per 32 x 32
        # tile; output layer = two 64-node tiles per CU (128 KB weights + 64 KB rows each); layer 0 = 110 KB + 27 KB per tile.
        clk = 2.1e9
        per_cu = 42.0 * clk
        stream_us = (6 * 131072 + 2 * 196608 + 141312) / per_cu * 1e6
        chain_us = 9 * 1.45
        c100 = small["calls"][0]["us_per_call"]
        rs = {"call": "100 frames, device resident", "measured_us": c100,
              "bound": "per-workgroup operand stream through one CU's L2->LDS path (42 B/clk/CU) + the launch chain (9 dependent launches)",
              "operand_stream_us": round(stream_us, 2), "launch_chain_us": round(chain_us, 2), "floor_us": round(stream_us + chain_us, 2),
              "frac": round((stream_us + chain_us) / c100, 4),
              "note": "a model, not a counter: bytes per workgroup from fdnn_small.hip's tile shapes; what the rest is -- first-stage "
                      "latency behind every boundary, the in-workgroup split-K reduction, the scale pass as a ninth launch -- is in "
                      "profiles/r06_small_100_kernel_stats.csv / r06_small_100_pmc.json (rocprofv3, replayed below when committed)"}
        for nn_ in (100, 1000):
            try:
                import csv

                rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", f"r06_small_{nn_}_kernel_stats.csv"))))
                def _short(k):  # kernel name without namespaces and without its argument list (the first '(' outside template brackets)
                    k = k.replace("void ", "").replace("fdnn::(anonymous namespace)::", "").replace("fdnn::", "")
                    depth, out_ = 0, []
                    for ch in k:
                        if ch == "<":
                            depth += 1
                        elif ch == ">":
                            depth -= 1
                        elif ch == "(" and depth == 0:
                            break
                        out_.append(ch)
                    return "".join(out_).strip()[:80]

                ks = {_short(r_["Name"]): round(float(r_["AverageNs"]) / 1e3, 2) for r_ in rows if "fdnn" in r_["Name"] and int(r_["Calls"]) >= 100}
                rs[f"kernel_avg_us_{nn_}_frames_replayed"] = ks
                rs["replayed_from"] = "profiles/r06_small_*_kernel_stats.csv"
            except Exception:  # noqa: BLE001
                pass
        small["roofline_small"] = rs
        small["note"] = ("fdnn_calculate_device, device-resident frames in and soft-max rows out, nine launches per call (layer 0, six "
                         "hidden layers, output layer, soft-max scale) on the small-batch kernels (fdnn_small.hip, l0_small_kernel); "
                         "round 2 took 119 / 115 us for these two calls")

    # ---- BASELINE configs[1]'s batch size: 1 000 frames per call, full soft-max, device resident (the reference's own
    # 1000-frame recipe is FuncTest.java:31-38); mid-size calls are one partly filled round of workgroups per layer
    config1 = None
    if world == 1 and not args.no_small:
        mid = []
        for sn in (1000, 2000, 4000):
            if sn > n:
                continue
            reps = 400
            for _ in range(40):
                dnn.calculate_device(x.data_ptr(), sn, outs[0].data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            for _ in range(reps):
                dnn.calculate_device(x.data_ptr(), sn, outs[0].data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t5) / reps * 1e6
            tops = INT8_OPS_PER_FRAME * sn / (us * 1e-6) / 1e12
            mid.append({"frames": sn, "us_per_call": round(us, 2), "ns_per_frame": round(us * 1e3 / sn, 1), "frames_per_s": round(sn / us * 1e6, 1),
                        "int8_tops": round(tops, 1), "frac_of_int8_peak": round(tops / INT8_PEAK_TOPS, 4)})
        assert abs(float(outs[0][:8].sum(1).mean().item()) - 1.0) < 1e-3
        config1 = {"workload": "BASELINE configs[1] batch size: 1000 frames per fdnn_calculate_device call on the 432 -> 7x2048 -> 8000 net "
                               "(the shipped data/16khz.bin is a FEATURE batch, SURVEY section 0; its frames tiled to 1000 are a parity test), "
                               "calls back to back on one stream; 2000 and 4000 frames beside it",
                   "calls": mid, **(mid[0] if mid else {})}

    # ---- the serving shape: 16 caller threads, 100-frame utterances (1 s of speech each), host buffers
    serving = None
    if world == 1 and not args.no_serving:
        import threading

        T, per, uf = 16, 60, 100
        utt = F.synth_features(uf, 432, seed=5)
        bufs = [np.zeros((uf, O), dtype=np.float32) for _ in range(T)]  # resident, reused (a JVM float[] would be)
        ssrv = api.ScoringServer(dnn, 6400, 3, 100)

        def via_server(t):
            for _ in range(per):
                tk, _o = ssrv.submit(utt, out=bufs[t])
                ssrv.wait(tk)

        def per_call(t):
            for _ in range(per):
                dnn.calculate(utt)

        def run(fn):
            th = [threading.Thread(target=fn, args=(t,)) for t in range(T)]
            t0_ = time.perf_counter()
            for h in th:
                h.start()
            for h in th:
                h.join()
            return T * per / (time.perf_counter() - t0_)

        run(via_server)
        s_rate = run(via_server)
        st = ssrv.stats()
        ssrv.close()
        run(per_call)
        p_rate = run(per_call)
        assert abs(float(bufs[0].sum(1).mean()) - 1.0) < 1e-3
        # the lazy contract in the same shape: one LazyContext per caller thread (QuantizedDnn.java:72-98), 40 % masks as bits,
        # rows back compacted (active probabilities + 1 / total per frame over PCIe, rebuilt in the caller's array)
        ubits = F.pack_mask_bits(F.generate_masks_fast(uf, O, 0.40, 0.03, seed=12))
        ctxs = [dnn.getNewLazyContext(uf) for _ in range(T)]

        def lazy_call(t):
            for _ in range(per):
                ctxs[t].calculateUntilOutput(utt)
                ctxs[t].calculateForOutputNodesBatchBits(ubits, 0, out=bufs[t])

        run(lazy_call)
        l2_rate = run(lazy_call)
        for cx in ctxs:
            cx.delete()

        # round 5: the same contract as ONE call per utterance (fdnn_calculate_lazy_bits: hidden layers + masked output +
        # compacted return, one stream synchronisation)
        def lazy_one_call(t):
            for _ in range(per):
                dnn.calculateLazy(utt, bits=ubits, out=bufs[t])

        run(lazy_one_call)
        l_rate = run(lazy_one_call)
        assert abs(float(bufs[0].sum(1).mean()) - 1.0) < 1e-3

        # ... and through the scoring loop (fdnn_server_submit_lazy_bits): the callers' utterances coalesced into one batch,
        # the rows back compacted, each caller's thread rebuilding its own
        lsrv = api.ScoringServer(dnn, 6400, 3, 100)

        def lazy_via_server(t):
            for _ in range(per):
                tk, _o = lsrv.submitLazy(utt, ubits, out=bufs[t])
                lsrv.wait(tk)

        run(lazy_via_server)
        ls_rate = run(lazy_via_server)
        lst = lsrv.stats()
        lsrv.close()
        assert abs(float(bufs[0].sum(1).mean()) - 1.0) < 1e-3
        # the same shape without an interpreter in the loop (tools/serve_bench.cpp, built with the library): Python caller
        # threads serialise their own bookkeeping on the GIL, ~60 us per utterance of a ~1 ms round trip
        native = None
        sb = os.path.join(ROOT, "fast-dnn_amd", "lib", "serve_bench")
        if os.path.exists(sb) and os.path.exists(model_path):
            import subprocess

            native = {}
            for mode_, extra in (("percall", []), ("server", ["6400", "3", "100"]), ("lazy", []), ("lazyserver", ["6400", "3", "100"])):
                try:
                    r_ = subprocess.run([sb, model_path, str(T), "150", str(uf), mode_] + extra, capture_output=True, text=True, timeout=120)
                    native[mode_] = json.loads(r_.stdout.strip().splitlines()[-1])["utts_per_s"]
                except Exception as e_:  # noqa: BLE001
                    native[mode_] = None
                    native["error"] = str(e_)[:200]
            if native.get("percall") and native.get("lazyserver"):
                native["lazy_loop_vs_dense_per_call"] = round(native["lazyserver"] / native["percall"], 3)
                native["lazy_loop_vs_dense_best"] = round(native["lazyserver"] / max(native["percall"], native.get("server") or 0), 3)
            native["note"] = (f"tools/serve_bench.cpp: {T} native caller threads x 150 utterances of {uf} frames each, its own process and model "
                              "handle; percall = fdnn_calculate, server = fdnn_server_submit + wait, lazy = fdnn_calculate_lazy_bits, "
                              "lazyserver = fdnn_server_submit_lazy_bits + wait (40 % active nodes, 3 % churn)")
        serving = {
            "native_harness": native,
            "lazy_40pct_utterances_per_s": round(max(l_rate, ls_rate), 1),
            "lazy_vs_dense_per_call": round(max(l_rate, ls_rate) / p_rate, 3),
            "lazy_vs_dense_best": round(max(l_rate, ls_rate) / max(s_rate, p_rate), 3),
            "lazy_40pct_through_the_scoring_loop_utterances_per_s": round(ls_rate, 1),
            "lazy_40pct_one_call_utterances_per_s": round(l_rate, 1),
            "lazy_loop_batches": lst["batches"],
            "lazy_40pct_two_call_protocol_utterances_per_s": round(l2_rate, 1),
            "lazy_note": "through_the_scoring_loop = fdnn_server_submit_lazy_bits + wait (bit-mask utterances of all callers coalesced, "
                         "one masked pass per batch, rows back compacted: 40 % of the floats + one value per frame over PCIe, each caller "
                         "rebuilds its rows); one_call = fdnn_calculate_lazy_bits per utterance; two_call_protocol = one LazyContext per "
                         "caller thread, calculateUntilOutput + fdnn_ctx_lazy_output_batch_bits per utterance (round 4's figure); "
                         "lazy_40pct_utterances_per_s = the better of the first two",
            "workload": f"{T} caller threads x {per} utterances of {uf} frames (1 s of speech), host frames in, host soft-max rows out "
                        f"({uf * O * 4 / 1e6:.1f} MB per utterance)",
            "utterances_per_s_through_the_scoring_loop": round(s_rate, 1),
            "utterances_per_s_per_call_fdnn_calculate": round(p_rate, 1),
            "streams_at_real_time": round(max(s_rate, p_rate), 1),
            "device_to_host_GB_per_s": round(max(s_rate, p_rate) * uf * O * 4 / 1e9, 1),
            "batches": st["batches"], "requests": st["requests"],
            "note": "Python caller threads (ctypes releases the GIL inside the library); tools/serve_bench.cpp is the native "
                    "harness (profiles/r02_serve_bench_raw.log); bound by PCIe + host memcpy of the 32 KB per frame, not by the GPU",
        }

    srv.close()
    if rank == 0:
        steps = args.steps
        # HBM-side bytes per launch come from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs, FETCH_SIZE doubled as the gfx950 guide prescribes); bench.py
        # itself cannot run under rocprof.  Newest round first.
        pmc, pmc_file = {}, None
        for cand in ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json", "r01_pmc_summary.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", cand)))
                pmc_file = cand
                break
            except Exception:
                continue

        def traffic_of(prefixes):
            """HBM bytes per launch from the committed PMC summary; several prefixes = the kernels of one launch group, summed."""
            if n != FRAMES_PER_GPU:
                return None
            total, found = 0, False
            for prefix in ([prefixes] if isinstance(prefixes, str) else prefixes):
                for name, c in pmc.items():
                    if name.startswith(prefix) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                        total += int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1000)
                        found = True
                        break
            return total if found else None

        step_ms = elapsed / steps * 1e3
        kinds = []

        def add(key, kernel, bound, work_per_launch, peak, unit, scale, alg_bytes, pmc_prefix):
            pr = prof[key]
            if not pr["launches"]:
                return
            per_launch_ms = pr["ms"] / pr["launches"]
            per_step_ms = pr["ms"] / steps
            achieved = work_per_launch / (per_launch_ms * 1e-3) / scale
            traffic = traffic_of(pmc_prefix)
            kinds.append({
                "kernel": kernel, "bound": bound, "achieved": round(achieved, 1), "peak": peak, "unit": unit,
                "frac": round(achieved / peak, 4),
                # against what a bare MFMA microbenchmark reaches on this chip (MI355X_MICROARCH.md: >= 3 944 TOP/s int8)
                "frac_of_measured_ceiling": round(achieved / INT8_MEASURED_TOPS, 4) if unit == "TOP/s" else None,
                "traffic": traffic,
                # NOT measured in this run: the HBM-side bytes per launch of the committed rocprofv3 --pmc passes
                "traffic_replayed_from": f"profiles/{pmc_file}" if traffic is not None else None,
                "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(per_launch_ms, 4),
                "launches_per_step": pr["launches"] // steps, "ms_per_step": round(per_step_ms, 4),
                "share_of_single_stream_step": round(per_step_ms / (single_elapsed / steps * 1e3), 4),
            })

        # layer 0, canonical flavour: pre-pass (frames -> 24-bit integers as three int8 digit planes), int8 screening kernel
        # (six exact digit products on the int8 MFMA, sampled chain sums, rigorous bound, flags), exact recomputation of the
        # ~0.35 % of outputs whose table byte the approximation could change.  Priced against the int8 MFMA peak with the
        # digit products actually issued (6 x 2 x 512 x 2048 ops per frame: the layer's own 2 x 432 x 2048 fp32 flop per frame
        # would read 19 TFLOP/s); the fused flavour still runs on the fp32 MFMA.
        if args.l0_fma:
            add("l0", "layer 0: l0_mfma_kernel (fused flavour, fp32 MFMA)", "mfma", 2.0 * 432 * 2048 * n, 157.3, "TFLOP/s", 1e12,
                4 * (432 * n + 432 * 2048) + 2048 * n, "l0_mfma_kernel")
        else:
            add("l0", "layer 0: l0_digits_kernel + l0_split_kernel + l0_fix_list_kernel (canonical numerics: exact int8 digit products on the "
                "int8 MFMA, rigorous error bound, exact unfused recomputation of ~0.35 % of the outputs)",
                "mfma", 6 * 2.0 * 512 * 2048 * n, INT8_PEAK_TOPS, "TOP/s", 1e12,
                4 * (432 * n + 432 * 2048) + 2048 * n, ("l0_digits_kernel", "l0_split_kernel", "l0_fix_list_kernel"))
        n_hidden = 6  # int8 hidden layers of the 432 -> 7x2048 -> 8000 net
        hid_launches = max(1, prof["hidden_gemm"]["launches"] // steps)
        layers_per_launch = n_hidden // hid_launches if n_hidden % hid_launches == 0 else 1
        if layers_per_launch > 1:  # round 5: ONE persistent launch for all hidden layers (fdnn_chain.hip)
            add("hidden_gemm", f"qchain_kernel (the {layers_per_launch} int8 hidden layers in one persistent launch: int8 MFMA 32x32x32, 2048x2048 layers + "
                "dequant/bias/sigmoid-table epilogues, tasks from per-XCD queues, frame-tile-local hand-off)",
                "mfma", layers_per_launch * 2.0 * 2048 * 2048 * n, INT8_PEAK_TOPS, "TOP/s", 1e12,
                layers_per_launch * (2048 * 2048 + 2 * n * 2048), "qchain_kernel")
        else:
            add("hidden_gemm", "qgemm_kernel<hidden> (int8 MFMA 32x32x32, 2048x2048 layer + dequant/bias/sigmoid-table epilogue)",
                "mfma", 2.0 * 2048 * 2048 * n, INT8_PEAK_TOPS, "TOP/s", 1e12, 2048 * 2048 + 2 * n * 2048, "qgemm_kernel hidden")
        fused_out = not prof["normalize"]["launches"]
        ppo_out = fused_out and any(name.startswith("qppo_kernel") for name in pmc)  # (round 6: the role-split fused kernel is the default at this size when the committed profile shows it)
        if ppo_out:
            add("output_gemm", "qppo_kernel (int8 MFMA, 8000x2048 layer, two groups of four waves alternating k-loop / epilogue; FUSED soft-max: exp in place in the "
                "accumulation registers, row sums exchanged between the 32 node tiles of a frame half, probabilities written as whole row segments, 32 KB per frame out)",
                "mfma", 2.0 * 2048 * O * n, INT8_PEAK_TOPS, "TOP/s", 1e12, 2048 * O + n * 2048 + 4 * n * O, "qppo_kernel output")
        else:
            add("output_gemm", "qgemm_kernel<output> (int8 MFMA, 8000x2048 layer + dequant/bias/exp epilogue" +
                (" + FUSED soft-max: row sums exchanged between the 256-node tiles of a frame tile, probabilities written directly, "
                 "32 KB per frame out)" if fused_out else ", 32 KB of exp(z) per frame out)"),
                "mfma", 2.0 * 2048 * O * n, INT8_PEAK_TOPS, "TOP/s", 1e12, 2048 * O + n * 2048 + 4 * n * O, "qgemm_kernel output")
        add("normalize", "normalize_kernel (soft-max scale: read + write [n][8000] fp32)", "hbm", 2.0 * O * 4 * n, HBM_PEAK_GBS,
            "GB/s", 1e9, 2 * O * 4 * n, "normalize_kernel")
        dominant = max(kinds, key=lambda k: k["ms_per_step"])
        gemm = next(k for k in kinds if k["kernel"].startswith("qgemm_kernel<hidden>") or k["kernel"].startswith("qchain_kernel"))
        rocprof = None
        for cand in ("r06_roofline.json", "r05_roofline.json", "r04_roofline.json"):  # the same fractions from the committed rocprofv3 averages (tools/profile_round.sh)
            try:
                rocprof = dict(json.load(open(os.path.join(ROOT, "profiles", cand))), replayed_from=f"profiles/{cand}",
                               replay_note="committed rocprofv3 averages of an earlier run of this command, NOT quantities of this run")
                break
            except Exception:  # noqa: BLE001
                continue
        value = world * n * steps / elapsed
        res = {
            "metric": "acoustic frames/sec (whole node), 7x2048->8000 nnet",
            "value": round(value, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "multi_gpu": multi,
            "config4_125k_per_gpu": config4,
            "host_fed": host_fed,
            "one_process_group": one_process,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(step_ms, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int8 (u8 activations x s8 weights -> int32; fp32 layer 0 and soft-max)",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[2]{'' if world == 1 else ' on every GPU (configs[4] as config4_125k_per_gpu)'}: synthetic Kaldi nnet 432 -> 7x2048 -> 8000 "
                            f"({args.mode} weights, seed 1), {n}-frame batch per GPU, full soft-max, device-resident in/out",
                "frames_per_gpu": n, "global_frames": world * n, "parallelism": f"frame-sharded x{world}, replicated weights",
                "layer0_numerics": "fused (reference built -march=native)" if args.l0_fma else "unfused (reference built -msse4, canonical)",
                "steps_in_flight": depth,
                "submission": "fdnn_server_submit_device: every step a complete pass into its own output buffer (dense and batched-lazy "
                              "steps alike scale their soft-max inside the output kernel)",
            },
            "x_realtime_per_gpu": round(value / world / 100.0, 1),
            "int8_tops_end_to_end": round(INT8_OPS_PER_FRAME * value / world / 1e12, 1),
            "single_stream": {"frames_per_s": round(world * n * steps / single_elapsed, 1), "ms_per_step": round(single_elapsed / steps * 1e3, 4),
                              "note": "the same K steps as back-to-back fdnn_calculate_device calls on one stream (no overlap between steps)"},
            "roofline": dict(dominant, note="largest share of the step; times from HIP events on the launch stream, which add ~4 us per "
                                            "bracketed launch; `traffic` is replayed from the committed PMC passes (traffic_replayed_from), frac / avg_launch_ms are live"),
            "roofline_int8_gemm": gemm,
            "roofline_kernels": kinds,
            "end_to_end": {"bound": "mfma", "achieved": round(value / world, 1), "peak": round(ROOFLINE_FRAMES_PER_S, 1), "unit": "frames/s per GPU",
                           "frac": round(value / world / ROOFLINE_FRAMES_PER_S, 4),
                           "note": "5 POP/s int8 / 83.1 M int8 ops per frame (layers 1..7); layer 0 (2 % of the MACs, fp32 in the reference) runs as "
                                   "six int8 digit products + an exact fix and takes about a fifth of the step; the 32 KB per frame of "
                                   "probabilities leave from inside the output kernel"},
            "traffic_source": pmc_file,
            "rocprof": rocprof,
            "setup": {"clock_ramp_steps": ramp_steps, "clock_ramp_s": args.clock_ramp_s,
                      "note": "untimed forward passes before the W warm-up steps, so that short runs see sustained clocks"},
            "kernel_ms_per_step": {k: round(v["ms"] / steps, 4) for k, v in prof.items()},
            "other_layer0_flavour": None if alt is None else {
                "layer0_numerics": "unfused (canonical)" if args.l0_fma else "fused (reference built -march=native), fp32 MFMA",
                "frames_per_s_single_stream": round(alt, 1)},
            "nosat_pair_free_net": nosat,
            "lazy_40pct": lazy,
            "config1_1000": config1,
            "small_batch": small,
            "serving": serving,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(model_path)
        print(json.dumps(res), flush=True)
    dnn.delete()
    if world > 1 or force_coll:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
