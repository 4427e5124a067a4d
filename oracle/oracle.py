"""ctypes front-ends for the parity checker.  TEST INFRASTRUCTURE ONLY.

``Oracle``  -- oracle/libfdnn_oracle.so, the C restatement (travels everywhere).
``RefLib``  -- oracle/_ref/libfastdnn_ref*.so, the reference itself compiled from
               /root/reference (present only where it was built).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under fast-dnn_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libfdnn_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")


def build(ref: bool = True) -> None:
    """Compile the checker (and the reference where /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"] + (["ref"] if ref else []))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: Optional[np.ndarray], ty):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


class _Taps(C.Structure):
    _fields_ = [
        ("u8_acts", C.POINTER(C.c_uint8)),
        ("acc_hid", C.POINTER(C.c_int32)),
        ("acc_out", C.POINTER(C.c_int32)),
        ("logits", C.POINTER(C.c_float)),
        ("l0_lin", C.POINTER(C.c_float)),
        ("sat_events", C.c_longlong),
    ]


class Oracle:
    """CPU restatement of QuantizedDnn / CalculationContext (oracle/fdnn_oracle.c)."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            if not os.path.exists(ORACLE_SO):
                build(ref=False)
            L = C.CDLL(ORACLE_SO)
            L.orc_model_load.restype = C.c_void_p
            L.orc_model_load.argtypes = [C.c_char_p, C.c_float]
            L.orc_model_free.argtypes = [C.c_void_p]
            for name in ("orc_n_layers",):
                getattr(L, name).argtypes = [C.c_void_p]
            for name in ("orc_layer_in", "orc_layer_out"):
                getattr(L, name).argtypes = [C.c_void_p, C.c_int]
            L.orc_layer_mult.argtypes = [C.c_void_p, C.c_int]
            L.orc_layer_mult.restype = C.c_float
            L.orc_layer_wq.argtypes = [C.c_void_p, C.c_int]
            L.orc_layer_wq.restype = C.POINTER(C.c_int8)
            L.orc_layer_bias.argtypes = [C.c_void_p, C.c_int]
            L.orc_layer_bias.restype = C.POINTER(C.c_float)
            L.orc_layer_w0.argtypes = [C.c_void_p]
            L.orc_layer_w0.restype = C.POINTER(C.c_float)
            L.orc_shift.argtypes = [C.c_void_p]
            L.orc_shift.restype = C.POINTER(C.c_float)
            L.orc_scale.argtypes = [C.c_void_p]
            L.orc_scale.restype = C.POINTER(C.c_float)
            L.orc_lut.argtypes = [C.POINTER(C.c_uint8)]
            L.orc_sigmoid_q.argtypes = [C.c_float]
            L.orc_sigmoid_q.restype = C.c_uint8
            L.orc_quantize.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int8), C.POINTER(C.c_float)]
            L.orc_hidden.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.POINTER(_Taps)]
            L.orc_output.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(_Taps)]
            L.orc_calculate.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(_Taps)]
            L.orc_lazy_output.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_int8), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
            L.orc_lazy_batch.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_int8), C.c_int, C.POINTER(C.c_float)]
            L.orc_risky_pairs.argtypes = [C.c_void_p, C.c_int]
            L.orc_risky_pairs.restype = C.c_longlong
            L.orc_set_l0_fma.argtypes = [C.c_int]
            cls._lib = L
        return cls._lib

    # ---- static helpers
    @classmethod
    def lut(cls) -> np.ndarray:
        out = np.zeros(1280, dtype=np.uint8)
        cls.lib().orc_lut(_ptr(out, C.c_uint8))
        return out

    @classmethod
    def sigmoid_q(cls, x: float) -> int:
        return int(cls.lib().orc_sigmoid_q(C.c_float(x)))

    @classmethod
    def quantize(cls, w: np.ndarray, cutoff: float = 3.0):
        w = _f32(w)
        out = np.zeros(w.shape, dtype=np.int8)
        mult = C.c_float()
        cls.lib().orc_quantize(_ptr(w, C.c_float), w.shape[0], w.shape[1], cutoff, _ptr(out, C.c_int8), C.byref(mult))
        return out, float(mult.value)

    @classmethod
    def set_l0_fma(cls, on: bool) -> None:
        cls.lib().orc_set_l0_fma(1 if on else 0)

    # ---- model
    def __init__(self, path: str, cutoff: float = 3.0):
        L = self.lib()
        self.h = L.orc_model_load(path.encode(), cutoff)
        if not self.h:
            raise IOError(f"oracle: cannot load {path}")
        self.n_layers = L.orc_n_layers(self.h)
        self.in_dim = L.orc_layer_in(self.h, 0)
        self.hidden = L.orc_layer_out(self.h, 0)
        self.out_dim = L.orc_layer_out(self.h, self.n_layers - 1)

    def close(self):
        if self.h:
            self.lib().orc_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def layer_wq(self, j: int) -> np.ndarray:
        L = self.lib()
        o, i = L.orc_layer_out(self.h, j), L.orc_layer_in(self.h, j)
        return np.ctypeslib.as_array(L.orc_layer_wq(self.h, j), shape=(o, i)).copy()

    def layer_mult(self, j: int) -> float:
        return float(self.lib().orc_layer_mult(self.h, j))

    def layer_bias(self, j: int) -> np.ndarray:
        L = self.lib()
        return np.ctypeslib.as_array(L.orc_layer_bias(self.h, j), shape=(L.orc_layer_out(self.h, j),)).copy()

    def risky_pairs(self, j: int) -> int:
        return int(self.lib().orc_risky_pairs(self.h, j))

    def calculate(self, x, batch: int = 10, sse: bool = True, taps: bool = False):
        """QuantizedDnn.calculate.  With taps=True also returns a dict of per-layer taps."""
        x = _f32(x)
        n = x.shape[0]
        assert x.shape[1] == self.in_dim, (x.shape, self.in_dim)
        out = np.zeros((n, self.out_dim), dtype=np.float32)
        t = None
        keep = {}
        if taps:
            nh = self.n_layers - 1
            keep = dict(
                u8_acts=np.zeros((nh, n, self.hidden), dtype=np.uint8),
                acc_hid=np.zeros((max(nh - 1, 0), n, self.hidden), dtype=np.int32),
                acc_out=np.zeros((n, self.out_dim), dtype=np.int32),
                logits=np.zeros((n, self.out_dim), dtype=np.float32),
                l0_lin=np.zeros((n, self.hidden), dtype=np.float32),
            )
            t = _Taps(_ptr(keep["u8_acts"], C.c_uint8), _ptr(keep["acc_hid"], C.c_int32), _ptr(keep["acc_out"], C.c_int32),
                      _ptr(keep["logits"], C.c_float), _ptr(keep["l0_lin"], C.c_float), 0)
        rc = self.lib().orc_calculate(self.h, _ptr(x, C.c_float), n, batch, int(sse), _ptr(out, C.c_float), C.byref(t) if t else None)
        if rc:
            raise RuntimeError(f"oracle rc={rc}")
        if taps:
            keep["sat_events"] = int(t.sat_events)
            return out, keep
        return out

    def hidden_acts(self, x, batch: int = 8, sse: bool = True) -> np.ndarray:
        x = _f32(x)
        n = x.shape[0]
        act = np.zeros((n, self.hidden), dtype=np.uint8)
        rc = self.lib().orc_hidden(self.h, _ptr(x, C.c_float), n, batch, int(sse), _ptr(act, C.c_uint8), None)
        if rc:
            raise RuntimeError(f"oracle rc={rc}")
        return act

    # ---- the same calls from several threads (ctypes drops the GIL; the model is read-only): every ROW of a
    # production-size batch against the restatement in about a second per 10 000 frames on 16 cores
    @staticmethod
    def _chunks(n: int, threads: int):
        threads = max(1, min(threads, n))
        step = -(-n // threads)
        return [(lo, min(n, lo + step)) for lo in range(0, n, step)]

    @staticmethod
    def default_threads() -> int:
        try:
            return max(1, min(32, len(os.sched_getaffinity(0))))
        except AttributeError:
            return max(1, min(32, os.cpu_count() or 1))

    def hidden_acts_mt(self, x, threads: Optional[int] = None) -> np.ndarray:
        """hidden_acts over frame ranges in parallel (frames are independent: dnn.cc:402-424 has no cross-frame state)."""
        from concurrent.futures import ThreadPoolExecutor

        x = _f32(x)
        n = x.shape[0]
        act = np.zeros((n, self.hidden), dtype=np.uint8)
        L = self.lib()

        def run(r):
            lo, hi = r
            rc = L.orc_hidden(self.h, _ptr(x[lo:hi], C.c_float), hi - lo, 8, 1, _ptr(act[lo:hi], C.c_uint8), None)
            if rc:
                raise RuntimeError(f"oracle rc={rc}")

        ch = self._chunks(n, threads or self.default_threads())
        with ThreadPoolExecutor(len(ch)) as ex:
            list(ex.map(run, ch))
        return act

    def output_mt(self, act, masks=None, want_acc: bool = False, threads: Optional[int] = None):
        """CalculateOutput (dnn.cc:428-454) -- or, with masks, LazyOutputActivations per frame (dnn.cc:355-392) -- over the
        given last-hidden-layer activations, frame ranges in parallel.  Returns probs, or (probs, acc_out int32)."""
        from concurrent.futures import ThreadPoolExecutor

        act = np.ascontiguousarray(act, dtype=np.uint8)
        n = act.shape[0]
        out = np.zeros((n, self.out_dim), dtype=np.float32)
        acc = np.zeros((n, self.out_dim), dtype=np.int32) if want_acc else None
        if masks is not None:
            masks = np.ascontiguousarray(masks, dtype=np.int8)
        L = self.lib()

        def run(r):
            lo, hi = r
            if masks is not None:
                rc = L.orc_lazy_batch(self.h, _ptr(act[lo:hi], C.c_uint8), hi - lo, _ptr(masks[lo:hi], C.c_int8), 1, _ptr(out[lo:hi], C.c_float))
            else:
                t = _Taps(None, None, _ptr(acc[lo:hi], C.c_int32) if want_acc else None, None, None, 0)
                rc = L.orc_output(self.h, _ptr(act[lo:hi], C.c_uint8), hi - lo, 8, 1, _ptr(out[lo:hi], C.c_float), C.byref(t) if want_acc else None)
            if rc:
                raise RuntimeError(f"oracle rc={rc}")

        ch = self._chunks(n, threads or self.default_threads())
        with ThreadPoolExecutor(len(ch)) as ex:
            list(ex.map(run, ch))
        return (out, acc) if want_acc else out

    def lazy(self, x, masks, batch: int = 8, sse: bool = True) -> np.ndarray:
        """LazyContext: calculateUntilOutput(x) then calculateForOutputNodes(mask) per frame."""
        act = self.hidden_acts(x, batch, sse)
        masks = np.ascontiguousarray(masks, dtype=np.int8)
        n = act.shape[0]
        out = np.zeros((n, self.out_dim), dtype=np.float32)
        rc = self.lib().orc_lazy_batch(self.h, _ptr(act, C.c_uint8), n, _ptr(masks, C.c_int8), int(sse), _ptr(out, C.c_float))
        if rc:
            raise RuntimeError(f"oracle rc={rc}")
        return out


class RefLib:
    """The reference itself (oracle/_ref/libfastdnn_ref.so, built from /root/reference)."""

    def __init__(self, fma: bool = False):
        so = os.path.join(REF_DIR, "libfastdnn_ref_fma.so" if fma else "libfastdnn_ref.so")
        if not os.path.exists(so):
            raise FileNotFoundError(so)
        L = C.CDLL(so)
        L.ref_model_load.restype = C.c_void_p
        L.ref_model_load.argtypes = [C.c_char_p, C.c_float]
        L.ref_model_free.argtypes = [C.c_void_p]
        for n_ in ("ref_input_dim", "ref_output_dim", "ref_q_layer_count", "ref_hidden_dim"):
            getattr(L, n_).argtypes = [C.c_void_p]
        for n_ in ("ref_q_out", "ref_q_in"):
            getattr(L, n_).argtypes = [C.c_void_p, C.c_int]
        L.ref_q_mult.argtypes = [C.c_void_p, C.c_int]
        L.ref_q_mult.restype = C.c_float
        L.ref_q_weights.argtypes = [C.c_void_p, C.c_int]
        L.ref_q_weights.restype = C.POINTER(C.c_int8)
        L.ref_lut.argtypes = [C.POINTER(C.c_uint8)]
        L.ref_sigmoid_get.argtypes = [C.c_float]
        L.ref_sigmoid_get.restype = C.c_uint8
        L.ref_quantize.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int8), C.POINTER(C.c_float)]
        L.ref_calculate.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.ref_hidden.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
        L.ref_lazy.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int8), C.POINTER(C.c_float)]
        L.ref_forward_taps.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                       C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                       C.POINTER(C.c_float)]
        self.L = L

    def lut(self) -> np.ndarray:
        out = np.zeros(1280, dtype=np.uint8)
        self.L.ref_lut(_ptr(out, C.c_uint8))
        return out

    def sigmoid_get(self, x: float) -> int:
        return int(self.L.ref_sigmoid_get(C.c_float(x)))

    def quantize(self, w, cutoff: float = 3.0):
        w = _f32(w)
        out = np.zeros(w.shape, dtype=np.int8)
        mult = C.c_float()
        self.L.ref_quantize(_ptr(w, C.c_float), w.shape[0], w.shape[1], cutoff, _ptr(out, C.c_int8), C.byref(mult))
        return out, float(mult.value)

    def load(self, path: str, cutoff: float = 3.0) -> "RefModel":
        return RefModel(self, path, cutoff)


class RefModel:
    def __init__(self, lib: RefLib, path: str, cutoff: float):
        self.lib = lib
        self.L = lib.L
        self.h = self.L.ref_model_load(path.encode(), cutoff)
        self.in_dim = self.L.ref_input_dim(self.h)
        self.out_dim = self.L.ref_output_dim(self.h)
        self.hidden = self.L.ref_hidden_dim(self.h)
        self.n_q = self.L.ref_q_layer_count(self.h)

    def close(self):
        if self.h:
            self.L.ref_model_free(self.h)
            self.h = None

    def q_weights(self, j: int) -> np.ndarray:
        o, i = self.L.ref_q_out(self.h, j), self.L.ref_q_in(self.h, j)
        return np.ctypeslib.as_array(self.L.ref_q_weights(self.h, j), shape=(o, i)).copy()

    def q_mult(self, j: int) -> float:
        return float(self.L.ref_q_mult(self.h, j))

    def calculate(self, x, batch: int = 10) -> np.ndarray:
        x = _f32(x)
        n = x.shape[0]
        out = np.zeros((n, self.out_dim), dtype=np.float32)
        self.L.ref_calculate(self.h, _ptr(x, C.c_float), n, x.shape[1], batch, _ptr(out, C.c_float))
        return out

    def hidden_acts(self, x, batch: int = 8) -> np.ndarray:
        x = _f32(x)
        n = x.shape[0]
        act = np.zeros((n, self.hidden), dtype=np.uint8)
        self.L.ref_hidden(self.h, _ptr(x, C.c_float), n, x.shape[1], batch, _ptr(act, C.c_uint8))
        return act

    def lazy(self, x, masks, batch: int = 8) -> np.ndarray:
        x = _f32(x)
        masks = np.ascontiguousarray(masks, dtype=np.int8)
        n = x.shape[0]
        out = np.zeros((n, self.out_dim), dtype=np.float32)
        self.L.ref_lazy(self.h, _ptr(x, C.c_float), n, x.shape[1], batch, _ptr(masks, C.c_int8), _ptr(out, C.c_float))
        return out

    def forward_taps(self, x, batch: int = 10) -> dict:
        x = _f32(x)
        n = x.shape[0]
        nh = self.n_q  # hidden layers = fp32 layer + (n_q - 1) int8 hidden layers
        t = dict(
            l0_lin=np.zeros((n, self.hidden), dtype=np.float32),
            u8_acts=np.zeros((nh, n, self.hidden), dtype=np.uint8),
            acc_hid=np.zeros((nh - 1, n, self.hidden), dtype=np.float32),
            acc_out=np.zeros((n, self.out_dim), dtype=np.float32),
            logits=np.zeros((n, self.out_dim), dtype=np.float32),
            probs=np.zeros((n, self.out_dim), dtype=np.float32),
        )
        self.L.ref_forward_taps(self.h, _ptr(x, C.c_float), n, x.shape[1], batch, _ptr(t["l0_lin"], C.c_float),
                                _ptr(t["u8_acts"], C.c_uint8), _ptr(t["acc_hid"], C.c_float), _ptr(t["acc_out"], C.c_float),
                                _ptr(t["logits"], C.c_float), _ptr(t["probs"], C.c_float))
        return t
