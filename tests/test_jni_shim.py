"""The JNI surface, driven by a mock JVM (tests/jni_mock.py): the unmodified Java
facade's call sequence against Java_suskun_nn_QuantizedDnn_*."""
import numpy as np
import pytest

from conftest import golden
from fast_dnn_amd import api, formats as F
from jni_mock import JNI_ABORT, JavaException, JavaQuantizedDnn, MockJvm


def test_mock_env_table_and_error_path_without_gpu(tmp_path):
    """Runs everywhere: a bad path must surface as a Java exception, not a crash
    (the reference exits the JVM, float_dnn.cc:171,:185-188)."""
    jvm = MockJvm()
    with pytest.raises(JavaException) as e:
        JavaQuantizedDnn.loadFromFile(api.lib(), jvm, str(tmp_path / "nope.bin"))
    assert e.value.cls == "java/lang/RuntimeException" and "cannot open" in str(e.value)
    assert jvm.stats["get_str"] == jvm.stats["release_str"] == 1 and jvm.leaks() == 0
    with pytest.raises(ValueError):
        JavaQuantizedDnn.loadFromFile(api.lib(), jvm, "x", weightCutOffValue=-1)


@pytest.mark.gpu
def test_java_facade_sequence(tiny_model_path):
    g = golden("tiny.npz")
    jvm = MockJvm()
    dnn = JavaQuantizedDnn.loadFromFile(api.lib(), jvm, tiny_model_path)
    assert (dnn.inputDim, dnn.outputDim, dnn.layerCount()) == (432, 100, 4)
    assert dnn.layerDimension(0) == 64 and dnn.layerDimension(2) == 100 and dnn.layerDimension(7) == -1
    probs, flat = dnn.calculate(g["x16"], 10)
    assert np.abs(probs - g["probs"]).max() <= 2e-6
    assert (flat == g["x16"].reshape(-1)).all()  # input array untouched
    assert dnn.calculate([])[0:0].shape[0] == 0
    with pytest.raises(ValueError):
        dnn.calculate(np.zeros((2, 429), np.float32))
    # a result that cannot be a Java float[] (jsize is 32 bits: 2^31 - 1 elements) is refused before anything is touched
    with pytest.raises(JavaException) as e:
        dnn._call("calculate", dnn.handle, jvm.new_object(np.zeros(432, np.float32)), 21_474_837, 432, 10)  # x 100 outputs > 2^31 - 1
    assert e.value.cls == "java/lang/IllegalArgumentException" and "Java float[]" in str(e.value)
    # lazy protocol, FuncTest.java:104-112
    masks = F.generate_masks(100, 100, 0.4, 0.03, seed=5)
    ctx = dnn.getNewLazyContext(100)
    ctx.calculateUntilOutput(g["x16"])
    rows = np.stack([ctx.calculateForOutputNodes(masks[i]) for i in range(100)])
    from oracle.oracle import Oracle

    want = Oracle(tiny_model_path).lazy(g["x16"], masks)
    assert np.abs(rows - want).max() <= 2e-6
    with pytest.raises(JavaException) as e:
        ctx.calculateForOutputNodes(masks[0])  # frame 100 of 100
    assert e.value.cls == "java/lang/IllegalArgumentException"
    ctx.delete()
    # the one-call extension returns the same rows as the per-frame protocol (bit for bit: same kernels per frame)
    batch = dnn.calculateLazyBatch(g["x16"], masks)
    assert np.abs(batch - want).max() <= 2e-6 and np.array_equal(batch, rows)
    with pytest.raises(JavaException) as e:
        dnn._call("calculateLazyBatch", dnn.handle, jvm.new_object(g["x16"].reshape(-1).copy()), 100, 432, jvm.new_object(np.zeros(5, np.int8)))
    assert e.value.cls == "java/lang/IllegalArgumentException"
    dnn.delete()
    assert jvm.leaks() == 0
    assert jvm.stats["get_float"] == jvm.stats["release_float"]
    assert jvm.stats["get_byte"] == jvm.stats["release_byte"]
    assert set(jvm.stats["release_modes"]) == {JNI_ABORT}  # jni_dnn.cc:58,:93,:115
