"""A mock JVM for the JNI shim: a 235-slot JNINativeInterface_ table built with
ctypes, filling the slots the shim uses, plus a Python re-statement of the Java
facade (src/java/suskun/nn/QuantizedDnn.java) that calls the exported
``Java_suskun_nn_QuantizedDnn_*`` symbols exactly as a JVM would.

No JDK exists in this image, so this is how the drop-in boundary is exercised.
Array elements are handed out as COPIES and released with the mode the shim
passes, so a shim that wrote into its input (as the reference does,
dnn.cc:175-192) or forgot a release shows up in ``MockJvm.stats``.
"""
import ctypes as C

import numpy as np

TABLE_SIZE = 235
JNI_ABORT = 2


class MockJvm:
    def __init__(self):
        self.objects = {}
        self.next_id = 0x1000
        self.pending_exception = None
        self.stats = {"get_float": 0, "release_float": 0, "get_byte": 0, "release_byte": 0, "get_str": 0, "release_str": 0,
                      "release_modes": []}
        self._live = {}  # element pointer -> (array id, ctypes buffer)
        self._cbs = []
        table = (C.c_void_p * TABLE_SIZE)()

        def put(index, restype, argtypes, fn):
            cb = C.CFUNCTYPE(restype, *argtypes)(fn)
            self._cbs.append(cb)
            table[index] = C.cast(cb, C.c_void_p).value

        vp = C.c_void_p
        put(6, vp, [vp, C.c_char_p], self._find_class)
        put(14, C.c_int, [vp, vp, C.c_char_p], self._throw_new)
        put(169, vp, [vp, vp, vp], self._get_string_utf)
        put(170, None, [vp, vp, vp], self._release_string_utf)
        put(171, C.c_int, [vp, vp], self._get_array_length)
        put(181, vp, [vp, C.c_int], self._new_float_array)
        put(184, vp, [vp, vp, vp], self._get_byte_elems)
        put(189, vp, [vp, vp, vp], self._get_float_elems)
        put(192, None, [vp, vp, vp, C.c_int], self._release_byte_elems)
        put(197, None, [vp, vp, vp, C.c_int], self._release_float_elems)
        put(213, None, [vp, vp, C.c_int, C.c_int, vp], self._set_float_region)
        self._table = table
        self._env_struct = C.c_void_p(C.addressof(table))  # JNIEnv = { functions* }
        self.env = C.addressof(self._env_struct)

    # ---- object registry
    def new_object(self, value):
        self.next_id += 8
        self.objects[self.next_id] = value
        return self.next_id

    def get(self, handle):
        return self.objects[handle]

    def take_exception(self):
        e, self.pending_exception = self.pending_exception, None
        return e

    # ---- table entries
    def _find_class(self, env, name):
        return self.new_object(("class", name.decode()))

    def _throw_new(self, env, cls, msg):
        self.pending_exception = (self.objects[cls][1], msg.decode(errors="replace"))
        return 0

    def _get_string_utf(self, env, jstr, is_copy):
        buf = C.create_string_buffer(self.objects[jstr].encode())
        addr = C.addressof(buf)
        self._live[addr] = (jstr, buf)
        self.stats["get_str"] += 1
        return addr

    def _release_string_utf(self, env, jstr, chars):
        self._live.pop(chars)
        self.stats["release_str"] += 1

    def _get_array_length(self, env, arr):
        return int(self.objects[arr].shape[0])

    def _new_float_array(self, env, n):
        return self.new_object(np.zeros(n, dtype=np.float32))

    def _get_elems(self, arr, key):
        a = self.objects[arr]
        buf = (C.c_char * max(a.nbytes, 1)).from_buffer_copy(a.tobytes() or b"\0")
        addr = C.addressof(buf)
        self._live[addr] = (arr, buf)
        self.stats[key] += 1
        return addr

    def _release_elems(self, arr, elems, mode, key):
        _, buf = self._live.pop(elems)
        self.stats[key] += 1
        self.stats["release_modes"].append(mode)
        if mode != JNI_ABORT:  # copy back, as a JVM that handed out a copy would
            a = self.objects[arr]
            a[...] = np.frombuffer(buf, dtype=a.dtype, count=a.size).reshape(a.shape)

    def _get_float_elems(self, env, arr, is_copy):
        return self._get_elems(arr, "get_float")

    def _get_byte_elems(self, env, arr, is_copy):
        return self._get_elems(arr, "get_byte")

    def _release_float_elems(self, env, arr, elems, mode):
        self._release_elems(arr, elems, mode, "release_float")

    def _release_byte_elems(self, env, arr, elems, mode):
        self._release_elems(arr, elems, mode, "release_byte")

    def _set_float_region(self, env, arr, start, n, src):
        a = self.objects[arr]
        a[start:start + n] = np.ctypeslib.as_array(C.cast(src, C.POINTER(C.c_float)), shape=(n,))

    def leaks(self):
        return len(self._live)


class JavaException(Exception):
    def __init__(self, cls, msg):
        super().__init__(f"{cls}: {msg}")
        self.cls = cls


class JavaQuantizedDnn:
    """QuantizedDnn.java re-stated over the mock JVM (same checks, defaults, flattening)."""

    def __init__(self, lib, jvm: MockJvm):
        self.L = lib
        self.jvm = jvm
        self.handle = 0
        P = "Java_suskun_nn_QuantizedDnn_"
        vp, ci, cl, cf = C.c_void_p, C.c_int32, C.c_int64, C.c_float
        self.f = {}
        for name, res, args in [
            ("initialize", cl, [vp, vp, vp, cf]), ("inputDimension", ci, [vp, vp, cl]), ("outputDimension", ci, [vp, vp, cl]),
            ("calculate", vp, [vp, vp, cl, vp, ci, ci, ci]), ("getContext", cl, [vp, vp, cl, ci, ci]),
            ("calculateUntilOutput", None, [vp, vp, cl, vp]), ("calculateLazy", vp, [vp, vp, cl, ci, vp]),
            ("deleteLazyContext", None, [vp, vp, cl]), ("delete", None, [vp, vp, cl]),
            ("layerDimension", ci, [vp, vp, cl, ci]), ("layerCount", ci, [vp, vp, cl]),
            ("calculateLazyBatch", vp, [vp, vp, cl, vp, ci, ci, vp]),  # extension (include/fdnn_jni.h)
        ]:
            fn = getattr(lib, P + name)
            fn.restype, fn.argtypes = res, args
            self.f[name] = fn
        self.this = jvm.new_object("this")

    def _call(self, name, *args):
        r = self.f[name](self.jvm.env, self.this, *args)
        exc = self.jvm.take_exception()
        if exc:
            raise JavaException(*exc)
        return r

    @classmethod
    def loadFromFile(cls, lib, jvm, path, weightCutOffValue=3.0):
        if weightCutOffValue <= 0:  # QuantizedDnn.java:55-57
            raise ValueError("Weight cut off value must be positive. But it is %s" % weightCutOffValue)
        d = cls(lib, jvm)
        d.handle = d._call("initialize", jvm.new_object(path), weightCutOffValue)
        d.inputDim = d.inputDimension()
        d.outputDim = d.outputDimension()
        return d

    def inputDimension(self):
        return self._call("inputDimension", self.handle)

    def outputDimension(self):
        return self._call("outputDimension", self.handle)

    def layerDimension(self, k):
        return self._call("layerDimension", self.handle, k)

    def layerCount(self):
        return self._call("layerCount", self.handle)

    def delete(self):
        self._call("delete", self.handle)

    def calculate(self, input2d, batchSize=10):
        if len(input2d) == 0:  # QuantizedDnn.java:154-156
            return np.zeros((0, 0), dtype=np.float32)
        if len(input2d[0]) != self.inputDim:  # :157-161
            raise ValueError("Input vector size %d must be equal with network input size %d" % (len(input2d[0]), self.inputDim))
        flat = np.ascontiguousarray(input2d, dtype=np.float32).reshape(-1).copy()  # toVector :170-178
        jarr = self.jvm.new_object(flat)
        res = self._call("calculate", self.handle, jarr, len(input2d), len(input2d[0]), batchSize)
        return self.jvm.get(res).reshape(len(input2d), self.outputDim), flat  # toMatrix :180-186

    def calculateLazyBatch(self, input2d, masks2d):
        """The extension's Java side: flatten frames and masks row-major, one native call, unflatten."""
        if len(input2d) == 0:
            return np.zeros((0, 0), dtype=np.float32)
        if len(input2d[0]) != self.inputDim:
            raise ValueError("Input vector size %d must be equal with network input size %d" % (len(input2d[0]), self.inputDim))
        flat = np.ascontiguousarray(input2d, dtype=np.float32).reshape(-1).copy()
        fm = np.ascontiguousarray(masks2d, dtype=np.int8).reshape(-1).copy()
        res = self._call("calculateLazyBatch", self.handle, self.jvm.new_object(flat), len(input2d), len(input2d[0]), self.jvm.new_object(fm))
        return self.jvm.get(res).reshape(len(input2d), self.outputDim)

    class LazyContext:
        def __init__(self, dnn, handle, n):
            self.dnn, self.handle, self.n, self.currentVectorIndex = dnn, handle, n, 0

        def calculateUntilOutput(self, input2d):
            flat = np.ascontiguousarray(input2d, dtype=np.float32).reshape(-1).copy()
            self.dnn._call("calculateUntilOutput", self.handle, self.dnn.jvm.new_object(flat))

        def calculateForOutputNodes(self, mask):
            jm = self.dnn.jvm.new_object(np.ascontiguousarray(mask, dtype=np.int8).copy())
            res = self.dnn._call("calculateLazy", self.handle, self.currentVectorIndex, jm)
            self.currentVectorIndex += 1
            return self.dnn.jvm.get(res)

        def delete(self):
            self.dnn._call("deleteLazyContext", self.handle)

    def getNewLazyContext(self, n, batchSize=8):
        return JavaQuantizedDnn.LazyContext(self, self._call("getContext", self.handle, n, batchSize), n)
