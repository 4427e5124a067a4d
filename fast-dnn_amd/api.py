"""Python mirror of ``suskun.nn.QuantizedDnn`` over the C-ABI of libfast-dnn.so.

Same names, argument meaning and error behaviour as the Java facade
(src/java/suskun/nn/QuantizedDnn.java), so tests read like the reference's own
FuncTest / MultiThreadedStressTest.  This module is a binding only: every
computation happens in the HIP library; there is no fallback, and importing
works without a GPU only so that host-side helpers and symbol checks can run.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FDNN_LIB") or os.path.join(_HERE, "lib", "libfast-dnn.so")  # FDNN_LIB: experiment builds
CSRC = os.path.join(_HERE, "csrc")

FDNN_OK, FDNN_E_ARG, FDNN_E_IO, FDNN_E_FORMAT, FDNN_E_DEVICE, FDNN_E_NOMEM, FDNN_E_STATE = 0, -1, -2, -3, -4, -5, -6


class FdnnError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"fdnn error {code}: {msg}")
        self.code = code


def build(verbose: bool = False) -> str:
    """Compile libfast-dnn.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    jobs = str(max(1, min(8, os.cpu_count() or 1)))  # the three kernel files take a minute each: compile them side by side
    subprocess.check_call(["make", "-C", CSRC, "-j", jobs, "all"] + ([] if verbose else ["-s"]))
    return LIB_PATH


_lib = None

_c_f32p = C.POINTER(C.c_float)
_c_i8p = C.POINTER(C.c_int8)
_c_u8p = C.POINTER(C.c_uint8)
_c_i32p = C.POINTER(C.c_int32)

# name -> (restype, argtypes); also the list of exported C-ABI symbols that
# include/fdnn.h declares (tests check both directions)
SIGNATURES = {
    "fdnn_last_error": (C.c_char_p, []),
    "fdnn_version": (C.c_char_p, []),
    "fdnn_device_count": (C.c_int, []),
    "fdnn_model_load": (C.c_int, [C.c_char_p, C.c_float, C.POINTER(C.c_void_p)]),
    "fdnn_model_load_on": (C.c_int, [C.c_char_p, C.c_float, C.c_int, C.POINTER(C.c_void_p)]),
    "fdnn_model_free": (None, [C.c_void_p]),
    "fdnn_model_input_dim": (C.c_int, [C.c_void_p]),
    "fdnn_model_output_dim": (C.c_int, [C.c_void_p]),
    "fdnn_model_hidden_dim": (C.c_int, [C.c_void_p]),
    "fdnn_model_layer_count": (C.c_int, [C.c_void_p]),
    "fdnn_model_layer_dim": (C.c_int, [C.c_void_p, C.c_int]),
    "fdnn_model_device": (C.c_int, [C.c_void_p]),
    "fdnn_model_set_l0_fma": (C.c_int, [C.c_void_p, C.c_int]),
    "fdnn_debug_set_l0_kernel": (C.c_int, [C.c_void_p, C.c_int]),
    "fdnn_debug_set_chain": (C.c_int, [C.c_int, C.c_int]),
    "fdnn_model_chain_faults": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong)]),
    "fdnn_debug_set_pp": (C.c_int, [C.c_int, C.c_int]),
    "fdnn_debug_set_ppo": (C.c_int, [C.c_int]),
    "fdnn_debug_raise_fuse_fault": (C.c_int, [C.c_void_p, C.c_int]),
    "fdnn_device_shared": (C.c_int, [C.c_int]),
    "fdnn_debug_set_fuse": (C.c_int, [C.c_int]),
    "fdnn_debug_set_l0_list_cap": (C.c_int, [C.c_void_p, C.c_int]),
    "fdnn_debug_chain_clocks": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]),
    "fdnn_calculate": (C.c_int, [C.c_void_p, _c_f32p, C.c_int, C.c_int, C.c_int, _c_f32p]),
    "fdnn_calculate_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "fdnn_calculate_lazy": (C.c_int, [C.c_void_p, _c_f32p, C.c_int, C.c_int, _c_i8p, _c_f32p]),
    "fdnn_calculate_lazy_bits": (C.c_int, [C.c_void_p, _c_f32p, C.c_int, C.c_int, C.POINTER(C.c_uint64), _c_f32p]),
    "fdnn_calculate_lazy_bits_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fdnn_ctx_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "fdnn_ctx_free": (None, [C.c_void_p]),
    "fdnn_ctx_frame_count": (C.c_int, [C.c_void_p]),
    "fdnn_ctx_output_dim": (C.c_int, [C.c_void_p]),
    "fdnn_ctx_forward_hidden": (C.c_int, [C.c_void_p, _c_f32p]),
    "fdnn_ctx_forward_hidden_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "fdnn_ctx_lazy_output": (C.c_int, [C.c_void_p, C.c_int, _c_i8p, _c_f32p]),
    "fdnn_ctx_lazy_output_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _c_i8p, _c_f32p]),
    "fdnn_ctx_lazy_output_batch_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fdnn_ctx_output": (C.c_int, [C.c_void_p, _c_f32p]),
    "fdnn_ctx_output_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "fdnn_ctx_read_hidden": (C.c_int, [C.c_void_p, _c_u8p]),
    "fdnn_server_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "fdnn_server_free": (None, [C.c_void_p]),
    "fdnn_server_set_linger_us": (C.c_int, [C.c_void_p, C.c_int]),
    "fdnn_server_submit_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]),
    "fdnn_server_submit": (C.c_int, [C.c_void_p, _c_f32p, C.c_int, _c_i8p, _c_f32p, C.POINTER(C.c_uint64)]),
    "fdnn_server_submit_lazy_bits": (C.c_int, [C.c_void_p, _c_f32p, C.c_int, C.c_void_p, _c_f32p, C.POINTER(C.c_uint64)]),
    "fdnn_debug_lazy_expand": (C.c_int, [_c_f32p, _c_f32p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "fdnn_server_wait": (C.c_int, [C.c_void_p, C.c_uint64]),
    "fdnn_server_drain": (C.c_int, [C.c_void_p]),
    "fdnn_server_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "fdnn_model_enable_batcher": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "fdnn_group_load": (C.c_int, [C.c_char_p, C.c_float, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "fdnn_group_free": (None, [C.c_void_p]),
    "fdnn_group_size": (C.c_int, [C.c_void_p]),
    "fdnn_group_model": (C.c_void_p, [C.c_void_p, C.c_int]),
    "fdnn_group_weight_transport": (C.c_char_p, [C.c_void_p]),
    "fdnn_group_worker_cpus": (C.c_char_p, [C.c_void_p, C.c_int]),
    "fdnn_group_calculate": (C.c_int, [C.c_void_p, _c_f32p, C.c_int, C.c_int, C.c_int, _c_f32p]),
    "fdnn_group_shard": (None, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "fdnn_group_attach": (C.c_int, [C.c_void_p]),
    "fdnn_model_blob_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "fdnn_model_export_blob": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "fdnn_model_import_blob": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "fdnn_debug_frame_chunks": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_int]),
    "fdnn_debug_frame_chunks_for": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "fdnn_debug_production_acc_out": (C.c_int, [C.c_void_p, _c_f32p, C.c_int, C.c_int, _c_i8p, _c_i32p, _c_f32p]),
    "fdnn_debug_forward_taps": (C.c_int, [C.c_void_p, _c_f32p, C.c_int, _c_i8p, _c_f32p, _c_u8p, _c_i32p, _c_i32p, _c_f32p, _c_f32p]),
    "fdnn_debug_layer0": (C.c_int, [C.c_void_p, _c_f32p, C.c_int, _c_u8p, C.POINTER(C.c_ulonglong)]),
    "fdnn_debug_layer0_screen": (C.c_int, [C.c_void_p, _c_f32p, C.c_int, _c_u8p, _c_f32p, _c_f32p, C.POINTER(C.c_ulonglong)]),
    "fdnn_ctx_lazy_output_batch_bits": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "fdnn_ctx_lazy_output_batch_bits_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fdnn_model_fuse_giveups": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong)]),
    "fdnn_debug_device_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]),
    "fdnn_profile_begin": (C.c_int, [C.c_void_p]),
    "fdnn_profile_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "fdnn_host_model_load": (C.c_int, [C.c_char_p, C.c_float, C.POINTER(C.c_void_p)]),
    "fdnn_host_model_free": (None, [C.c_void_p]),
    "fdnn_host_model_layers": (C.c_int, [C.c_void_p]),
    "fdnn_host_model_layer_in": (C.c_int, [C.c_void_p, C.c_int]),
    "fdnn_host_model_layer_out": (C.c_int, [C.c_void_p, C.c_int]),
    "fdnn_host_model_multiplier": (C.c_float, [C.c_void_p, C.c_int]),
    "fdnn_host_model_weights_q": (C.c_int, [C.c_void_p, C.c_int, _c_i8p]),
    "fdnn_host_model_bias": (C.c_int, [C.c_void_p, C.c_int, _c_f32p]),
    "fdnn_host_model_wsum128": (C.c_int, [C.c_void_p, C.c_int, _c_i32p]),
    "fdnn_host_model_risky_pairs": (C.c_longlong, [C.c_void_p, C.c_int]),
    "fdnn_host_model_blob_size": (C.c_size_t, [C.c_void_p]),
    "fdnn_host_model_blob": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "fdnn_host_blob_check": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.POINTER(C.c_int)]),
    "fdnn_host_sigmoid_lut": (C.c_int, [_c_u8p]),
    "fdnn_host_quantize": (C.c_int, [_c_f32p, C.c_int, C.c_int, C.c_float, _c_i8p, _c_f32p]),
}

JNI_SYMBOLS = [
    "Java_suskun_nn_QuantizedDnn_" + n
    for n in ("initialize", "inputDimension", "outputDimension", "calculate", "getContext", "calculateUntilOutput",
              "calculateLazy", "deleteLazyContext", "delete", "layerDimension", "layerCount",
              "calculateLazyBatch")  # (the last one: an extension)
]


def _one_hip_runtime() -> None:
    """A process must hold ONE HIP runtime.  PyTorch wheels bundle their own ``libamdhip64.so``
    (same SONAME as ROCm's); if libfast-dnn.so pulls in /opt/rocm's copy first and torch is
    imported afterwards, torch loads a second runtime, which finds no device.  So when torch is
    installed but not imported yet, its copy is loaded first and both sides share it (the order
    bench.py uses anyway: torch first)."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(p):
        try:
            C.CDLL(p, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib() -> C.CDLL:
    """Load libfast-dnn.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the scorer)")
        _one_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc: int) -> None:
    if rc != FDNN_OK:
        raise FdnnError(rc, lib().fdnn_last_error().decode(errors="replace"))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def device_count() -> int:
    return int(lib().fdnn_device_count())


def device_shared(device: int = 0) -> bool:
    """True when another process held the GPU's marker first: this process runs the unfused soft-max (fdnn_device_shared)."""
    r = int(lib().fdnn_device_shared(int(device)))
    if r < 0:
        _check(r)
    return r == 1


def set_fuse(mode: int) -> None:
    """Soft-max scale of large batches (process-wide, tests): 0 = separate pass, 1 = inside the output kernel, -1 = default."""
    _check(lib().fdnn_debug_set_fuse(int(mode)))


def set_chain(mode: int, min_frames: int = 0) -> None:
    """How the int8 hidden layers run (process-wide; results are bit-identical): 1 = one persistent launch for batches of
    at least min_frames frames, 0 = one launch per layer, -1 = the default."""
    _check(lib().fdnn_debug_set_chain(int(mode), int(min_frames)))


def set_pp(mode: int, min_frames: int = 0) -> None:
    """How a large batch's int8 hidden layers run when launched layer by layer (process-wide; results are bit-identical):
    1 = the role-split kernel (fdnn_pp.hip) for batches of at least min_frames frames, 0 = the in-phase tiles, -1 = default."""
    _check(lib().fdnn_debug_set_pp(int(mode), int(min_frames)))


def set_ppo(mode: int) -> None:
    """How a large dense batch's output layer runs when its soft-max is fused (process-wide; results are bit-identical):
    1 = the role-split kernel (fdnn_ppo.hip) whenever the shape allows, 0 = the in-phase fused tiles, -1 = default."""
    _check(lib().fdnn_debug_set_ppo(int(mode)))


class LazyContext:
    """``QuantizedDnn.LazyContext`` (QuantizedDnn.java:72-98)."""

    def __init__(self, dnn: "QuantizedDnn", handle: int, input_vector_count: int):
        self.dnn = dnn
        self.handle = handle
        self.inputVectorCount = input_vector_count
        self.currentVectorIndex = 0

    def calculateUntilOutput(self, input) -> None:
        x = _f32(input)
        if x.shape != (self.inputVectorCount, self.dnn.inputDimension()):
            raise ValueError(f"expected {self.inputVectorCount}x{self.dnn.inputDimension()} frames, got {x.shape}")
        _check(lib().fdnn_ctx_forward_hidden(self.handle, x.ctypes.data_as(_c_f32p)))

    def calculateForOutputNodes(self, activeNodesMask) -> np.ndarray:
        mask = np.ascontiguousarray(activeNodesMask, dtype=np.int8)
        if mask.shape != (self.dnn.outputDimension(),):
            raise ValueError("mask length must equal the output dimension")
        out = np.empty(self.dnn.outputDimension(), dtype=np.float32)
        _check(lib().fdnn_ctx_lazy_output(self.handle, self.currentVectorIndex, mask.ctypes.data_as(_c_i8p),
                                          out.ctypes.data_as(_c_f32p)))
        self.currentVectorIndex += 1
        return out

    # batched form of the same contract
    def calculateForOutputNodesBatch(self, masks, first: int = 0) -> np.ndarray:
        masks = np.ascontiguousarray(masks, dtype=np.int8)
        if masks.ndim != 2 or masks.shape[1] != self.dnn.outputDimension():  # the C side reads count x output_dim bytes
            raise ValueError(f"masks must be count x {self.dnn.outputDimension()}, got {masks.shape}")
        count = masks.shape[0]
        if first < 0 or first + count > self.inputVectorCount:
            raise ValueError(f"frames [{first}, {first + count}) outside the context's {self.inputVectorCount} frames")
        out = np.empty((count, self.dnn.outputDimension()), dtype=np.float32)
        _check(lib().fdnn_ctx_lazy_output_batch(self.handle, first, count, masks.ctypes.data_as(_c_i8p),
                                                out.ctypes.data_as(_c_f32p)))
        return out

    # device-resident forms (raw device pointers, e.g. ``tensor.data_ptr()``; enqueued on ``stream``)
    def calculateForOutputNodesBatchBits(self, bits, first: int = 0, out=None) -> np.ndarray:
        """Batched lazy output with the masks as bits: uint64 [count][ceil(O / 64)] (formats.pack_mask_bits).
        ``out``: a resident float32 [count][O] array to fill (a JVM caller's float[] would be)."""
        b = np.ascontiguousarray(bits, dtype=np.uint64)
        O = self.dnn.outputDimension()
        if b.ndim != 2 or b.shape[1] != (O + 63) // 64:
            raise ValueError("bits must be [count][ceil(outputDimension / 64)] uint64")
        if out is None:
            out = np.empty((b.shape[0], O), dtype=np.float32)
        elif out.dtype != np.float32 or out.shape != (b.shape[0], O) or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous float32 [count][outputDimension] array")
        _check(lib().fdnn_ctx_lazy_output_batch_bits(self.handle, first, b.shape[0], b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out

    def calculateForOutputNodesBatchBitsDevice(self, d_bits: int, d_out: int, first: int, count: int, stream: int = 0) -> None:
        _check(lib().fdnn_ctx_lazy_output_batch_bits_device(self.handle, first, count, C.c_void_p(d_bits), C.c_void_p(d_out), C.c_void_p(stream)))

    def calculateUntilOutputDevice(self, d_input: int, stream: int = 0) -> None:
        _check(lib().fdnn_ctx_forward_hidden_device(self.handle, C.c_void_p(d_input), C.c_void_p(stream)))

    def calculateForOutputNodesBatchDevice(self, d_masks: int, d_out: int, first: int, count: int, stream: int = 0) -> None:
        _check(lib().fdnn_ctx_lazy_output_batch_device(self.handle, first, count, C.c_void_p(d_masks), C.c_void_p(d_out),
                                                       C.c_void_p(stream)))

    def chainClocks(self, cap_tasks: int, fetch: bool = False):
        """Measurement builds (-DFDNN_CHAIN_CLK=1): arm (fetch=False) / read the chained kernel's per-task phase clocks."""
        if not fetch:
            _check(lib().fdnn_debug_chain_clocks(self.handle, None, int(cap_tasks)))
            return None
        buf = np.zeros(8 + 10 * int(cap_tasks), dtype=np.int64)
        _check(lib().fdnn_debug_chain_clocks(self.handle, buf.ctypes.data_as(C.POINTER(C.c_longlong)), int(cap_tasks)))
        k = int(buf[0] & 0xffffffff)
        return buf[8:8 + 10 * min(k, int(cap_tasks))].reshape(-1, 10)

    def hiddenActivations(self) -> np.ndarray:
        out = np.empty((self.inputVectorCount, self.dnn.hiddenDimension()), dtype=np.uint8)
        _check(lib().fdnn_ctx_read_hidden(self.handle, out.ctypes.data_as(_c_u8p)))
        return out

    def delete(self) -> None:
        if self.handle:
            lib().fdnn_ctx_free(self.handle)
            self.handle = None


def group_shard(n: int, world: int, rank: int):
    """[start, stop) of one device's contiguous frame shard (``fdnn_group_shard``)."""
    a, b = C.c_int(), C.c_int()
    lib().fdnn_group_shard(n, world, rank, C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


class DeviceGroup:
    """One process, several devices of one node (``fdnn_group_*``): weights quantized once and sent
    device-to-device at load, the frames of each ``calculate`` call sharded over the replicas."""

    def __init__(self, dnnFile: str, devices: Sequence[int], weightCutOffValue: float = 3.0):
        if weightCutOffValue <= 0:
            raise ValueError(f"Weight cut off value must be positive. But it is {weightCutOffValue}")
        h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        _check(lib().fdnn_group_load(os.path.abspath(dnnFile).encode(), weightCutOffValue, arr, len(devices), C.byref(h)))
        self.handle = h.value

    def size(self) -> int:
        return lib().fdnn_group_size(self.handle)

    def model(self, index: int) -> "QuantizedDnn":
        """A view of replica ``index`` (owned by the group: do not delete it)."""
        return QuantizedDnn(lib().fdnn_group_model(self.handle, index))

    def weightTransport(self) -> str:
        return lib().fdnn_group_weight_transport(self.handle).decode()

    def workerCpus(self, index: int) -> str:
        """CPU list replica `index`'s persistent host thread pinned itself to ("" before the first large call)."""
        return lib().fdnn_group_worker_cpus(self.handle, int(index)).decode()

    def calculate(self, input, batchSize: int = 10) -> np.ndarray:
        x = np.asarray(input, dtype=np.float32)
        if x.shape[0] == 0:
            return np.zeros((0, 0), dtype=np.float32)
        m = self.model(0)
        if x.ndim != 2 or x.shape[1] != m.inputDimension():
            raise ValueError(f"Input vector size {x.shape[-1]} must be equal with network input size {m.inputDimension()}")
        x = np.ascontiguousarray(x)
        out = np.empty((x.shape[0], m.outputDimension()), dtype=np.float32)
        _check(lib().fdnn_group_calculate(self.handle, x.ctypes.data_as(_c_f32p), x.shape[0], x.shape[1], batchSize,
                                          out.ctypes.data_as(_c_f32p)))
        return out

    def delete(self) -> None:
        if self.handle:
            lib().fdnn_group_free(self.handle)
            self.handle = None


class ScoringServer:
    """Multi-stream scoring loop over one model (``fdnn_server_*``): up to ``depth`` batches in
    flight, the soft-max scale of one batch under layer 0 of the next, host submissions from any
    number of threads coalesced into full batches.  Generalises ``QuantizedDnn.LazyContext`` /
    the caller-thread model of MultiThreadedStressTest.java to many utterances per GPU."""

    def __init__(self, dnn: "QuantizedDnn", max_frames: int, depth: int = 2, linger_us: int = 0):
        h = C.c_void_p()
        _check(lib().fdnn_server_create(dnn.nativeDnnHandle, max_frames, depth, C.byref(h)))
        self.handle = h.value
        self.dnn = dnn
        self._keep = {}  # ticket -> arrays that must outlive the submission
        if linger_us:
            _check(lib().fdnn_server_set_linger_us(self.handle, linger_us))

    def submit_device(self, d_x: int, n: int, d_out: int, d_masks: int = 0) -> int:
        t = C.c_uint64()
        _check(lib().fdnn_server_submit_device(self.handle, C.c_void_p(d_x), n, C.c_void_p(d_masks) if d_masks else None,
                                               C.c_void_p(d_out), C.byref(t)))
        return int(t.value)

    def submit(self, x, masks=None, out=None):
        """Host frames -> (ticket, out array).  ``out`` is valid after ``wait(ticket)``; pass a
        C-contiguous float32 [n][output_dim] array to have the result written there (reused buffers)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.dnn.inputDimension():
            raise ValueError(f"Input vector size {x.shape[-1]} must be equal with network input size {self.dnn.inputDimension()}")
        O = self.dnn.outputDimension()
        m = None
        if masks is not None:
            m = np.ascontiguousarray(masks, dtype=np.int8)
            if m.shape != (x.shape[0], O):
                raise ValueError(f"masks must be {x.shape[0]} x {O}, got {m.shape}")
        if out is None:
            out = np.empty((x.shape[0], O), dtype=np.float32)
        elif out.dtype != np.float32 or out.shape != (x.shape[0], O) or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous float32 array of shape {(x.shape[0], O)}")
        t = C.c_uint64()
        _check(lib().fdnn_server_submit(self.handle, x.ctypes.data_as(_c_f32p), x.shape[0],
                                        m.ctypes.data_as(_c_i8p) if m is not None else None, out.ctypes.data_as(_c_f32p), C.byref(t)))
        self._keep[int(t.value)] = (x, m, out)
        return int(t.value), out

    def submitLazy(self, x, bits, out=None):
        """The lazy contract through the loop, masks as bits ([n][ceil(output_dim / 64)] uint64, ``formats.pack_mask_bits``):
        coalesced with the other callers' bit-mask submissions, rows back compacted and rebuilt inside ``out`` by the
        thread that waits for the ticket -> (ticket, out array)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.dnn.inputDimension():
            raise ValueError(f"Input vector size {x.shape[-1]} must be equal with network input size {self.dnn.inputDimension()}")
        O = self.dnn.outputDimension()
        b = np.ascontiguousarray(bits, dtype=np.uint64)
        if b.shape != (x.shape[0], (O + 63) // 64):
            raise ValueError(f"bits must be {x.shape[0]} x {(O + 63) // 64} uint64, got {b.shape}")
        if out is None:
            out = np.empty((x.shape[0], O), dtype=np.float32)
        elif out.dtype != np.float32 or out.shape != (x.shape[0], O) or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous float32 array of shape {(x.shape[0], O)}")
        t = C.c_uint64()
        _check(lib().fdnn_server_submit_lazy_bits(self.handle, x.ctypes.data_as(_c_f32p), x.shape[0], C.c_void_p(b.ctypes.data),
                                                  out.ctypes.data_as(_c_f32p), C.byref(t)))
        self._keep[int(t.value)] = (x, b, out)
        return int(t.value), out

    def wait(self, ticket: int) -> None:
        try:
            _check(lib().fdnn_server_wait(self.handle, ticket))
        finally:
            self._keep.pop(ticket, None)

    def drain(self) -> None:
        _check(lib().fdnn_server_drain(self.handle))
        self._keep.clear()

    def stats(self) -> dict:
        v = [C.c_uint64() for _ in range(4)]
        _check(lib().fdnn_server_stats(self.handle, *[C.byref(a) for a in v]))
        return dict(zip(("batches", "frames", "requests", "coalesced_requests"), (int(a.value) for a in v)))

    def close(self) -> None:
        if self.handle:
            lib().fdnn_server_free(self.handle)
            self.handle = None


class QuantizedDnn:
    """``suskun.nn.QuantizedDnn`` (QuantizedDnn.java) on one MI355X."""

    def __init__(self, handle: int):
        self.nativeDnnHandle = handle

    # -- construction ---------------------------------------------------
    @staticmethod
    def loadFromFile(dnnFile: str, weightCutOffValue: float = 3.0, device: Optional[int] = None) -> "QuantizedDnn":
        if weightCutOffValue <= 0:  # QuantizedDnn.java:55-57
            raise ValueError(f"Weight cut off value must be positive. But it is {weightCutOffValue}")
        h = C.c_void_p()
        path = os.path.abspath(dnnFile).encode()
        if device is None:
            _check(lib().fdnn_model_load(path, weightCutOffValue, C.byref(h)))
        else:
            _check(lib().fdnn_model_load_on(path, weightCutOffValue, device, C.byref(h)))
        return QuantizedDnn(h.value)

    @staticmethod
    def fromDeviceBlob(d_ptr: int, nbytes: int, device: int) -> "QuantizedDnn":
        h = C.c_void_p()
        _check(lib().fdnn_model_import_blob(C.c_void_p(d_ptr), nbytes, device, C.byref(h)))
        return QuantizedDnn(h.value)

    def delete(self) -> None:
        if self.nativeDnnHandle:
            lib().fdnn_model_free(self.nativeDnnHandle)
            self.nativeDnnHandle = None

    # -- queries ----------------------------------------------------------
    def inputDimension(self) -> int:
        return lib().fdnn_model_input_dim(self.nativeDnnHandle)

    def outputDimension(self) -> int:
        return lib().fdnn_model_output_dim(self.nativeDnnHandle)

    def hiddenDimension(self) -> int:
        return lib().fdnn_model_hidden_dim(self.nativeDnnHandle)

    def layerDimension(self, layerIndex: int) -> int:
        return lib().fdnn_model_layer_dim(self.nativeDnnHandle, layerIndex)

    def layerCount(self) -> int:
        return lib().fdnn_model_layer_count(self.nativeDnnHandle)

    def enableBatcher(self, max_frames: int, depth: int = 2, linger_us: int = 0) -> None:
        """Route ``calculate`` through an internal coalescing server (also: FDNN_BATCHER env at load)."""
        _check(lib().fdnn_model_enable_batcher(self.nativeDnnHandle, max_frames, depth, linger_us))

    def setInputLayerFma(self, on: bool) -> None:
        _check(lib().fdnn_model_set_l0_fma(self.nativeDnnHandle, int(on)))

    def setInputLayerListCap(self, cap: int) -> None:
        """Tests: cap the flagged-output list of the int8-screened input layer (contexts created afterwards)."""
        _check(lib().fdnn_debug_set_l0_list_cap(self.nativeDnnHandle, int(cap)))

    def setInputLayerKernel(self, kind: int) -> None:
        """0 = chosen by batch size, 1 = chain-pass kernel, 2 = 64 x 64-tile kernel, 3 = screened path where available (same bits)."""
        _check(lib().fdnn_debug_set_l0_kernel(self.nativeDnnHandle, int(kind)))

    # -- dense path ---------------------------------------------------------
    def calculate(self, input, batchSize: int = 10) -> np.ndarray:
        """float[][] calculate(float[][] input, int batchSize) -- QuantizedDnn.java:149-167."""
        x = np.asarray(input, dtype=np.float32)
        if x.shape[0] == 0:  # QuantizedDnn.java:154-156
            return np.zeros((0, 0), dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.inputDimension():
            raise ValueError(f"Input vector size {x.shape[-1]} must be equal with network input size {self.inputDimension()}")
        x = np.ascontiguousarray(x)
        out = np.empty((x.shape[0], self.outputDimension()), dtype=np.float32)
        _check(lib().fdnn_calculate(self.nativeDnnHandle, x.ctypes.data_as(_c_f32p), x.shape[0], x.shape[1], batchSize,
                                    out.ctypes.data_as(_c_f32p)))
        return out

    def calculate_device(self, d_x: int, n: int, d_out: int, stream: int = 0) -> None:
        """Device-resident form: raw device pointers (e.g. ``tensor.data_ptr()``), enqueued on ``stream``."""
        _check(lib().fdnn_calculate_device(self.nativeDnnHandle, C.c_void_p(d_x), n, C.c_void_p(d_out), C.c_void_p(stream)))

    def calculateLazy(self, input, masks=None, bits=None, out=None) -> np.ndarray:
        """One-call lazy scoring: hidden layers + masked output layer for every frame of ``input`` (LazyContext's
        calculateUntilOutput + calculateForOutputNodes per frame, QuantizedDnn.java:72-107, in one native call).
        ``masks`` [n][outputDimension] bytes (non-zero = active) or ``bits`` [n][ceil(O / 64)] uint64."""
        x = _f32(input)
        n, O = x.shape[0], self.outputDimension()
        if n and x.shape[1] != self.inputDimension():
            raise ValueError(f"input vector size {x.shape[1]} must be equal with network input size {self.inputDimension()}")
        if out is None:
            out = np.empty((n, O), dtype=np.float32)
        if bits is not None:
            b = np.ascontiguousarray(bits, dtype=np.uint64)
            if b.shape != (n, (O + 63) // 64):
                raise ValueError("bits must be [frames][ceil(outputDimension / 64)] uint64")
            _check(lib().fdnn_calculate_lazy_bits(self.nativeDnnHandle, x.ctypes.data_as(_c_f32p), n, x.shape[1] if n else self.inputDimension(),
                                                  b.ctypes.data_as(C.POINTER(C.c_uint64)), out.ctypes.data_as(_c_f32p)))
        else:
            m = np.ascontiguousarray(masks, dtype=np.int8)
            if m.shape != (n, O):
                raise ValueError("masks must be [frames][outputDimension]")
            _check(lib().fdnn_calculate_lazy(self.nativeDnnHandle, x.ctypes.data_as(_c_f32p), n, x.shape[1] if n else self.inputDimension(),
                                             m.ctypes.data_as(_c_i8p), out.ctypes.data_as(_c_f32p)))
        return out

    def calculate_lazy_bits_device(self, d_x: int, n: int, d_bits: int, d_out: int, stream: int = 0) -> None:
        _check(lib().fdnn_calculate_lazy_bits_device(self.nativeDnnHandle, C.c_void_p(d_x), n, C.c_void_p(d_bits), C.c_void_p(d_out), C.c_void_p(stream)))

    # -- lazy path ----------------------------------------------------------
    def getNewLazyContext(self, inputVectorCount: int, batchSize: int = 8) -> LazyContext:
        h = C.c_void_p()
        _check(lib().fdnn_ctx_create(self.nativeDnnHandle, inputVectorCount, batchSize, C.byref(h)))
        return LazyContext(self, h.value, inputVectorCount)

    # -- weight blob (multi-GPU) -------------------------------------------
    def blobSize(self) -> int:
        n = C.c_size_t()
        _check(lib().fdnn_model_blob_size(self.nativeDnnHandle, C.byref(n)))
        return int(n.value)

    def exportBlob(self, d_dst: int, capacity: int, stream: int = 0) -> None:
        _check(lib().fdnn_model_export_blob(self.nativeDnnHandle, C.c_void_p(d_dst), capacity, C.c_void_p(stream)))

    # -- per-kernel HIP-event timing ------------------------------------------
    PROF_KINDS = ("l0", "fix", "hidden_gemm", "output_gemm", "normalize")

    def raiseFuseFault(self, value: int = 1) -> None:
        """Tests: as if a fused soft-max launch of this model had just given up its bounded wait (1) / forget it (0)."""
        _check(lib().fdnn_debug_raise_fuse_fault(self.nativeDnnHandle, int(value)))

    def fuseGiveups(self) -> int:
        """Tiles of the fused soft-max that had to be finished by the clean-up kernel since load (fdnn_model_fuse_giveups)."""
        v = C.c_ulonglong(0)
        _check(lib().fdnn_model_fuse_giveups(self.nativeDnnHandle, C.byref(v)))
        return int(v.value)

    def chainFaults(self) -> int:
        """Waits of the chained hidden-layer kernel that ran into their bound since load (0 in a healthy setup)."""
        v = C.c_ulonglong()
        _check(lib().fdnn_model_chain_faults(self.nativeDnnHandle, C.byref(v)))
        return int(v.value)

    def deviceCounters(self, n: int = 32):
        a = (C.c_ulonglong * n)()
        _check(lib().fdnn_debug_device_counters(self.nativeDnnHandle, a, n))
        return list(a)

    def profileBegin(self) -> None:
        _check(lib().fdnn_profile_begin(self.nativeDnnHandle))

    def profileEnd(self) -> dict:
        ms = (C.c_double * 5)()
        cnt = (C.c_int * 5)()
        _check(lib().fdnn_profile_end(self.nativeDnnHandle, ms, cnt))
        return {k: {"ms": float(ms[i]), "launches": int(cnt[i])} for i, k in enumerate(self.PROF_KINDS)}

    def layer0(self, input):
        """Layer 0 alone through the production kernels: (u8 [n][hidden], outputs recomputed exactly)."""
        x = _f32(input)
        out = np.empty((x.shape[0], self.hiddenDimension()), dtype=np.uint8)
        rec = C.c_ulonglong()
        _check(lib().fdnn_debug_layer0(self.nativeDnnHandle, x.ctypes.data_as(_c_f32p), x.shape[0], out.ctypes.data_as(_c_u8p), C.byref(rec)))
        return out, int(rec.value)

    # -- parity taps ----------------------------------------------------------
    def layer0Screen(self, input):
        """(u8 [n][H], t~ [n][H], Dd [n][H], recomputed): the int8-screened input layer with what the screen saw per output."""
        x = _f32(input)
        H = self.hiddenDimension()
        out = np.empty((x.shape[0], H), dtype=np.uint8)
        t = np.empty((x.shape[0], H), dtype=np.float32)
        dd = np.empty((x.shape[0], H), dtype=np.float32)
        rec = C.c_ulonglong(0)
        _check(lib().fdnn_debug_layer0_screen(self.nativeDnnHandle, x.ctypes.data_as(_c_f32p), x.shape[0], out.ctypes.data_as(_c_u8p),
                                              t.ctypes.data_as(_c_f32p), dd.ctypes.data_as(_c_f32p), C.byref(rec)))
        return out, t, dd, int(rec.value)

    def productionOutputAcc(self, input, stride: int, masks=None, probs: bool = False):
        """int32 accumulators of the output layer's PRODUCTION kernel instance for every stride-th frame
        ([ceil(n/stride)][O]); with probs=True also the call's probabilities."""
        x = _f32(input)
        n, O = x.shape[0], self.outputDimension()
        acc = np.empty(((n + stride - 1) // stride, O), dtype=np.int32)
        pr = np.empty((n, O), dtype=np.float32) if probs else None
        m = None
        if masks is not None:
            m = np.ascontiguousarray(masks, dtype=np.int8)
            assert m.shape == (n, O)
        _check(lib().fdnn_debug_production_acc_out(
            self.nativeDnnHandle, x.ctypes.data_as(_c_f32p), n, stride, m.ctypes.data_as(_c_i8p) if m is not None else None,
            acc.ctypes.data_as(_c_i32p), pr.ctypes.data_as(_c_f32p) if pr is not None else None))
        return (acc, pr) if probs else acc

    def forwardTaps(self, input, masks=None) -> dict:
        x = _f32(input)
        n, H, O = x.shape[0], self.hiddenDimension(), self.outputDimension()
        nh = self.layerCount() - 1
        t = dict(
            l0_lin=np.empty((n, H), dtype=np.float32),
            u8_acts=np.empty((nh, n, H), dtype=np.uint8),
            acc_hid=np.empty((nh - 1, n, H), dtype=np.int32),
            acc_out=np.empty((n, O), dtype=np.int32),
            logits=np.empty((n, O), dtype=np.float32),
            probs=np.empty((n, O), dtype=np.float32),
        )
        m = None
        if masks is not None:
            m = np.ascontiguousarray(masks, dtype=np.int8)
            assert m.shape == (n, O)
        _check(lib().fdnn_debug_forward_taps(
            self.nativeDnnHandle, x.ctypes.data_as(_c_f32p), n, m.ctypes.data_as(_c_i8p) if m is not None else None,
            t["l0_lin"].ctypes.data_as(_c_f32p), t["u8_acts"].ctypes.data_as(_c_u8p), t["acc_hid"].ctypes.data_as(_c_i32p),
            t["acc_out"].ctypes.data_as(_c_i32p), t["logits"].ctypes.data_as(_c_f32p), t["probs"].ctypes.data_as(_c_f32p)))
        return t


class HostModel:
    """Load-time half only (no device): .bin parse + quantizer + packed sections."""

    def __init__(self, path: str, cutoff: float = 3.0):
        h = C.c_void_p()
        _check(lib().fdnn_host_model_load(os.path.abspath(path).encode(), cutoff, C.byref(h)))
        self.h = h.value
        L = lib()
        self.n_layers = L.fdnn_host_model_layers(self.h)

    def close(self):
        if self.h:
            lib().fdnn_host_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def layer_in(self, j):
        return lib().fdnn_host_model_layer_in(self.h, j)

    def layer_out(self, j):
        return lib().fdnn_host_model_layer_out(self.h, j)

    def multiplier(self, j) -> float:
        return float(lib().fdnn_host_model_multiplier(self.h, j))

    def weights_q(self, j) -> np.ndarray:
        out = np.empty((self.layer_out(j), self.layer_in(j)), dtype=np.int8)
        _check(lib().fdnn_host_model_weights_q(self.h, j, out.ctypes.data_as(_c_i8p)))
        return out

    def bias(self, j) -> np.ndarray:
        out = np.empty(self.layer_out(j), dtype=np.float32)
        _check(lib().fdnn_host_model_bias(self.h, j, out.ctypes.data_as(_c_f32p)))
        return out

    def wsum128(self, j) -> np.ndarray:
        out = np.empty(self.layer_out(j), dtype=np.int32)
        _check(lib().fdnn_host_model_wsum128(self.h, j, out.ctypes.data_as(_c_i32p)))
        return out

    def risky_pairs(self, j) -> int:
        return int(lib().fdnn_host_model_risky_pairs(self.h, j))

    def blob_size(self) -> int:
        return int(lib().fdnn_host_model_blob_size(self.h))

    def blob(self) -> np.ndarray:
        out = np.empty(self.blob_size(), dtype=np.uint8)
        _check(lib().fdnn_host_model_blob(self.h, out.ctypes.data_as(C.c_void_p), out.size))
        return out


def host_blob_check(blob: np.ndarray) -> dict:
    """Receiver-side validation of a broadcast weight blob (no device needed)."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    d = [C.c_int() for _ in range(4)]
    _check(lib().fdnn_host_blob_check(blob.ctypes.data_as(C.c_void_p), blob.size, *[C.byref(v) for v in d]))
    return dict(zip(("input_dim", "hidden_dim", "output_dim", "n_affine"), (int(v.value) for v in d)))


def frame_chunks(n: int, chained: bool = True):
    """(first frame, frame count) of every chunk a pass over ``n`` frames runs as (host logic, no device needed); ``chained``:
    whether the batch's hidden layers run as one chained launch (if not, a small tail past a whole round is split off)."""
    buf = (C.c_int * 4096)()
    k = lib().fdnn_debug_frame_chunks_for(int(n), 1 if chained else 0, buf, 2048)
    if k < 0:
        raise ValueError("fdnn_debug_frame_chunks")
    return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(k)]


def host_sigmoid_lut() -> np.ndarray:
    out = np.empty(1280, dtype=np.uint8)
    _check(lib().fdnn_host_sigmoid_lut(out.ctypes.data_as(_c_u8p)))
    return out


def host_quantize(w, cutoff: float = 3.0):
    w = _f32(w)
    out = np.empty(w.shape, dtype=np.int8)
    mult = C.c_float()
    _check(lib().fdnn_host_quantize(w.ctypes.data_as(_c_f32p), w.shape[0], w.shape[1], cutoff, out.ctypes.data_as(_c_i8p),
                                    C.byref(mult)))
    return out, float(mult.value)
