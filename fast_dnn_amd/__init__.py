"""Import alias: the package itself lives in ``fast-dnn_amd/`` (a directory name
Python cannot import directly); this shim points ``__path__`` there."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "fast-dnn_amd")]
