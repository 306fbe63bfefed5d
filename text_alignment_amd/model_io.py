"""Loader for ocropy line models (`*.pyrnn.gz`, e.g. the reference's
salzinnes_model-00054500.pyrnn.gz, reference alignToOCR.py:390-405; absent from the tree,
.MISSING_LARGE_BLOBS:1-2).  Row N1 of SURVEY.md section 8f.

The file is a gzip'd Python-2 protocol-2 pickle of ocrolib.lstm.SeqRecognizer (SURVEY.md
Appendix B.6).  Unpickling arbitrary globals would execute arbitrary code, so the unpickler here
resolves ONLY the handful of names that object graph uses, mapping the ocrolib classes to inert
attribute bags and the numpy reconstruction helpers to numpy's own; anything else raises.
Local paths only -- nothing is ever fetched.
"""
import gzip
import io
import pickle

import numpy as np

from .ocr import LineModel


class _Bag(object):
    """Inert stand-in for an ocrolib class: keeps whatever state the pickle sets."""

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):
            state = dict(state[0] or {}, **state[1])
        self.__dict__.update(state)


_OCROLIB_CLASSES = ("SeqRecognizer", "Stacked", "Parallel", "Reversed", "LSTM", "Softmax", "Codec",
                    "Logreg", "MLP", "Network", "CenterNormalizer", "MeanNormalizer")
_OCROLIB_MODULES = ("ocrolib.lstm", "lstm.lstm", "lstm", "ocrolib.lineest", "lineest", "ocrolib.common")


def _numpy_global(module, name):
    if module in ("numpy.core.multiarray", "numpy._core.multiarray") and name in ("_reconstruct", "scalar"):
        import numpy.core.multiarray as ma
        return getattr(ma, name)
    if module == "numpy" and name in ("ndarray", "dtype"):
        return getattr(np, name)
    return None


class RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module in _OCROLIB_MODULES and name in _OCROLIB_CLASSES:
            return type(name, (_Bag,), {})
        g = _numpy_global(module, name)
        if g is not None:
            return g
        if (module, name) in (("copy_reg", "_reconstructor"), ("copyreg", "_reconstructor")):
            import copyreg
            return copyreg._reconstructor
        if (module, name) in (("__builtin__", "object"), ("builtins", "object")):
            return object
        if (module, name) == ("_codecs", "encode"):      # how Python 3 writes bytes at protocol 2
            import _codecs
            return _codecs.encode
        raise pickle.UnpicklingError("global %s.%s is not allowed in a line-model file" % (module, name))


def _find(obj, name):
    v = getattr(obj, name, None)
    if v is None:
        raise ValueError("line model has no attribute %r" % name)
    return v


def model_from_graph(rec):
    """SeqRecognizer -> Stacked.nets -> [Parallel.nets -> [LSTM, Reversed.net -> LSTM], Softmax]."""
    stacked = _find(rec, "lstm")
    parallel, softmax = _find(stacked, "nets")[0], _find(stacked, "nets")[1]
    fwd_net, rev_wrap = _find(parallel, "nets")[0], _find(parallel, "nets")[1]
    rev_net = _find(rev_wrap, "net")
    keys = ("WGI", "WGF", "WGO", "WCI", "WIP", "WFP", "WOP")
    fwd = {k: np.asarray(_find(fwd_net, k), dtype=np.float64) for k in keys}
    rev = {k: np.asarray(_find(rev_net, k), dtype=np.float64) for k in keys}
    W2 = np.asarray(_find(softmax, "W2"), dtype=np.float64)
    no = W2.shape[0]
    code2char = _find(_find(rec, "codec"), "code2char")
    codec = [code2char.get(k, "~") for k in range(no)]     # unknown codes decode to '~'
    return LineModel(fwd, rev, W2, codec)


def load_pyrnn(path):
    """Read a local .pyrnn.gz (or uncompressed .pyrnn) line model."""
    with open(path, "rb") as f:
        head = f.read(2)
    opener = gzip.open if head == b"\x1f\x8b" else open
    with opener(path, "rb") as f:
        data = f.read()
    rec = RestrictedUnpickler(io.BytesIO(data), encoding="latin1").load()
    return model_from_graph(rec)
