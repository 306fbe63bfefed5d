"""oracle/ocr_ref_f64.py -- float64 numpy restatement of the `ocropus-rpred` line recogniser.
TEST/BENCH INFRASTRUCTURE ONLY (never imported by text_alignment_amd/).

PARITY UNPINNED.  The arithmetic the reference runs at alignToOCR.py:142-147 lives in the
third-party package ocropy==1.3.3 (reference requirements.txt:2), which is not vendored under
/root/reference, not installed here, and whose model files are absent
(.MISSING_LARGE_BLOBS:1-2).  The reference has no test or golden vector at this boundary.
This file therefore restates ocropy 1.3.3's published algorithm (ocrolib/lstm.py `LSTM.forward`,
`Softmax.forward`, `translate_back`; ocropus-rpred's llocs writer) as recorded in SURVEY.md
Appendix B, and it -- not ocropy -- is the oracle of record for kernels K3/K4.  Anchors in the
reference itself: the command line (alignToOCR.py:142-143), the `.llocs` format its parser
expects (alignToOCR.py:157-170: "char<TAB>x" per line, x a float in strip pixels) and the
filtering of '~' and '' classes (alignToOCR.py:175).

Shapes: ni = 48 input rows (line height), ns = 100 states per direction,
na = 1 + ni + ns = 149, No = number of classes (class 0 = blank "", 1 = " ", 2 = "~").
"""
import numpy as np


class LineModel(object):
    """Weights of Stacked([Parallel(LSTM(ni,ns), Reversed(LSTM(ni,ns))), Softmax(2*ns, No)])."""

    def __init__(self, ni, ns, no, fwd, rev, W2, codec):
        self.ni, self.ns, self.no = ni, ns, no
        self.fwd, self.rev = fwd, rev          # dicts: WGI WGF WGO WCI (ns x na), WIP WFP WOP (ns)
        self.W2 = W2                           # (no, 1 + 2*ns)
        self.codec = codec                     # list of No strings


def synthetic_model(seed, ni=48, ns=100, no=96):
    """Seeded random weights in the shapes of an ocropy line model."""
    rng = np.random.default_rng(seed)
    na = 1 + ni + ns

    def lstm():
        d = {}
        for k in ("WGI", "WGF", "WGO", "WCI"):
            d[k] = rng.uniform(-0.5, 0.5, size=(ns, na))
        for k in ("WIP", "WFP", "WOP"):
            d[k] = rng.uniform(-0.5, 0.5, size=(ns,))
        return d
    fwd, rev = lstm(), lstm()
    W2 = rng.uniform(-1.0, 1.0, size=(no, 1 + 2 * ns))
    chars = list("abcdefghijklmnopqrstuvwxyz.,;:-^0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ'\"()[]!?/&*+=<>#%@$")
    codec = ["", " ", "~"] + chars
    codec = (codec + ["c%d" % k for k in range(no)])[:no]
    return LineModel(ni, ns, no, fwd, rev, W2, codec)


def synthetic_line(seed, width=None, ni=48):
    """A normalised text-line strip after prepare_line: (T, ni) float64, ink = 1, 16 zero
    columns of padding on each side (Appendix B.2).  T = width + 32."""
    rng = np.random.default_rng(seed)
    if width is None:
        width = int(rng.integers(800, 2001))
    ink = (rng.random((width, ni)) < 0.15).astype(np.float64)
    # cheap blur along both axes so values are not only 0/1
    img = ink.copy()
    img[1:] += 0.5 * ink[:-1]
    img[:-1] += 0.5 * ink[1:]
    img[:, 1:] += 0.5 * ink[:, :-1]
    img[:, :-1] += 0.5 * ink[:, 1:]
    img = np.clip(img, 0.0, 1.0)
    xs = np.zeros((width + 32, ni), dtype=np.float64)
    xs[16:16 + width] = img
    return xs


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(np.clip(-x, -20, 20)))).astype(x.dtype)


def lstm_forward(w, xs, h0=None, c0=None, t0=0, return_cell=False, dtype=np.float64):
    """One direction, Appendix B.3 (`forward_py`): returns (T, ns) outputs.

    h0, c0, t0: continue a sequence whose first t0 steps were run elsewhere (the peephole rules
    "[t > 0]" refer to the position in the whole sequence).  return_cell: also the (T, ns) cell
    states.  dtype=np.float32 runs the same loop in single precision (tests use it to show how fast
    two float32 implementations of THIS model drift apart; the oracle of record is float64)."""
    T = xs.shape[0]
    ns = w["WGI"].shape[0]
    W = {k: np.asarray(v, dtype=dtype) for k, v in w.items()}
    xs = np.asarray(xs, dtype=dtype)
    h = np.zeros(ns, dtype=dtype) if h0 is None else np.asarray(h0, dtype=dtype)
    c = np.zeros(ns, dtype=dtype) if c0 is None else np.asarray(c0, dtype=dtype)
    out = np.zeros((T, ns), dtype=dtype)
    cells = np.zeros((T, ns), dtype=dtype)
    one = np.ones(1, dtype=dtype)
    for t in range(T):
        src = np.concatenate((one, xs[t], h))
        gi = W["WGI"].dot(src)
        gf = W["WGF"].dot(src)
        go = W["WGO"].dot(src)
        ci = np.tanh(W["WCI"].dot(src))
        if t + t0 > 0:
            gi = gi + W["WIP"] * c
            gf = gf + W["WFP"] * c
        gi = _sigmoid(gi)
        gf = _sigmoid(gf)
        c_new = ci * gi
        if t + t0 > 0:
            c_new = c_new + gf * c
            go = go + W["WOP"] * c_new          # output peephole skipped at t = 0
        go = _sigmoid(go)
        c = c_new
        h = np.tanh(c) * go
        out[t] = h
        cells[t] = c
    return (out, cells) if return_cell else out


def bilstm_states(model, xs):
    """(T, 2*ns): forward outputs followed by the reversed LSTM's outputs flipped back."""
    f = lstm_forward(model.fwd, xs)
    b = lstm_forward(model.rev, xs[::-1])[::-1]
    return np.concatenate([f, b], axis=1)


def softmax_layer(model, ys):
    """Appendix B.4: returns (logits, probabilities), both (T, No)."""
    T = ys.shape[0]
    src = np.concatenate([np.ones((T, 1)), ys], axis=1)
    z = src.dot(model.W2.T)
    p = np.exp(np.clip(z, -100, 100))
    p = p / p.sum(axis=1, keepdims=True)
    return z, p


def translate_back(outputs, threshold=0.7):
    """Appendix B.5: maximal runs of t with outputs[t,0] < threshold; per run the (t, class) of
    the maximum over the run x all classes (first in C order on ties)."""
    T = outputs.shape[0]
    res = []
    t = 0
    while t < T:
        if outputs[t, 0] < threshold:
            s = t
            while t < T and outputs[t, 0] < threshold:
                t += 1
            seg = outputs[s:t]
            k = int(np.argmax(seg))             # flat index, first maximum in C order
            res.append((s + k // seg.shape[1], k % seg.shape[1]))
        else:
            t += 1
    return res


def recognise(model, xs, raw_width=None):
    """Full line: returns dict(logits, probs, decoded=[(t, class)], llocs=[(char, x)])."""
    ys = bilstm_states(model, xs)
    z, p = softmax_layer(model, ys)
    dec = translate_back(p)
    T = xs.shape[0]
    if raw_width is None:
        raw_width = T - 32
    scale = float(raw_width) / (T - 32)
    llocs = [(model.codec[c], (t - 16) * scale) for (t, c) in dec]
    return dict(states=ys, logits=z, probs=p, decoded=dec, llocs=llocs)


def llocs_text(llocs):
    """The `.llocs` wire format parsed at alignToOCR.py:157-170."""
    return "".join("%s\t%.1f\n" % (ch, x) for ch, x in llocs)
