"""Line recogniser on MI355X -- the in-process replacement of the `ocropus-rpred` subprocess
that the reference shells out to (reference alignToOCR.py:142-147).

The network is ocropy 1.3.3's line model (third-party):
Stacked([Parallel(LSTM(48,100), Reversed(LSTM(48,100))), Softmax(200, No)]) followed by
translate_back(threshold 0.7).  `LineRecognizer` packs a model's weights into the MFMA fragment
layout of csrc/ta_lstm.hip once, then recognises any number of prepared text lines per call:
lines are sorted by length, grouped 16 to a workgroup, and run through K3 (BiLSTM), K4
(output layer + softmax) and K5 (decode) behind the C ABI (`ta_lstm_forward`,
`ta_lstm_output` / `ta_lstm_output_split`, `ta_decode`).
"""
import numpy as np
import torch

from . import _native

NI = 48
NS = 100
MAX_T = 5000            # ocropy preallocates 5000 timesteps ("input too large for LSTM model")
THRESHOLD = 0.7         # translate_back default
PAD = 16                # prepare_line pad


class RecognitionError(Exception):
    pass


MAX_CLASSES = 128        # csrc/ta_lstm.hip: kMaxCT = 8 class tiles of 16
PRECISIONS = ("f32", "split", "f64")
# The default is the reference's arithmetic type: ocropy computes in float64 numpy, and "f64" is
# the mode in which the 1e-3 logit tolerance holds free-running on every line.  "f32" (1.7x faster on the recogniser
# alone, ~1.1x on whole pages) and "split" are opt-in fast modes with a measured agreement (bench.py: ocr.agreement).
DEFAULT_PRECISION = "f64"


class LineModel(object):
    """Weights of one line-recognition model.

    fwd / rev: dicts with WGI, WGF, WGO, WCI (ns x (1+ni+ns)) and WIP, WFP, WOP (ns);
    W2: (no, 1 + 2*ns); codec: list of `no` strings (class 0 = "", 1 = " ", 2 = "~").
    """

    def __init__(self, fwd, rev, W2, codec, ni=NI, ns=NS):
        self.ni, self.ns = ni, ns
        self.fwd, self.rev = fwd, rev
        self.W2 = np.asarray(W2)
        self.no = int(self.W2.shape[0])
        self.codec = list(codec)
        if ni != NI or ns != NS:
            raise ValueError("the HIP kernels are built for ni=48, ns=100 line models")
        if self.W2.shape[1] != 1 + 2 * ns or len(self.codec) != self.no:
            raise ValueError("inconsistent output layer / codec sizes")
        if not 2 <= self.no <= MAX_CLASSES:
            raise ValueError("a line model needs 2..%d output classes (lstm_output_kernel holds the whole "
                             "softmax row of a timestep in one accumulator tile); this one has %d"
                             % (MAX_CLASSES, self.no))

    @classmethod
    def random(cls, seed, no=96):
        """Random-init weights of the architecture (for benchmarks: the trained model files of
        the reference are absent, .MISSING_LARGE_BLOBS:1-2)."""
        rng = np.random.default_rng(seed)
        na = 1 + NI + NS

        def lstm():
            d = {}
            for k in ("WGI", "WGF", "WGO", "WCI"):
                d[k] = rng.uniform(-0.5, 0.5, size=(NS, na))
            for k in ("WIP", "WFP", "WOP"):
                d[k] = rng.uniform(-0.5, 0.5, size=(NS,))
            return d
        fwd, rev = lstm(), lstm()
        W2 = rng.uniform(-1.0, 1.0, size=(no, 1 + 2 * NS))
        codec = ["", " ", "~"] + [chr(ord('a') + (k % 26)) for k in range(no - 3)]
        return cls(fwd, rev, W2, codec)


def _bf16_bits(x):
    """float32 array -> bf16 bit patterns (round to nearest even), and the value they stand for."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32)
    back = (r << 16).astype(np.uint32).view(np.float32)
    return r.astype(np.uint16), back


def _pack_lstm_split(model):
    """B fragments of csrc/ta_lstm.hip's split-operand kernel: W = W_hi (bf16) + W_r (fp16 of the
    rest), [dir 2][wave 7][plane 2: hi, r][gate 4][k-step 5][lane 64][8] 16-bit patterns."""
    out = np.zeros((2, 7, 2, 4, 5, 64, 8), dtype=np.uint16)
    lane = np.arange(64)
    for d, w in enumerate((model.fwd, model.rev)):
        for g, name in enumerate(("WGI", "WGF", "WGO", "WCI")):
            W = np.asarray(w[name], dtype=np.float64)
            Wp = np.zeros((112, 160), dtype=np.float32)
            Wp[:NS, 0:1 + NI] = W[:, 0:1 + NI]
            Wp[:NS, 52:152] = W[:, 1 + NI:]
            hi_bits, hi_val = _bf16_bits(Wp)
            rest_bits = (Wp - hi_val).astype(np.float16).view(np.uint16)
            for wv in range(7):
                rows = 16 * wv + (lane & 15)
                for ks in range(5):
                    for j in range(8):
                        cols = 32 * ks + 8 * (lane >> 4) + j
                        out[d, wv, 0, g, ks, :, j] = hi_bits[rows, cols]
                        out[d, wv, 1, g, ks, :, j] = rest_bits[rows, cols]
    assert out.size * 2 == _native.lib.ta_lstm_packed_weight_floats(1) * 4
    return out


def _pack_output_split(model):
    """B fragments of the split-operand output layer: W2 = W_hi (bf16) + W_r (fp16 of the rest),
    [plane 2: hi, r][class tile][k-step 7][lane 64][8] 16-bit patterns of
    W2[16 tile + lane % 16][1 + 32 kstep + 4 (lane // 16) + 16 (j // 4) + j % 4] (inputs beyond 200
    zero; the kernel reads a row of hout in this order), and the bias
    column W2[:, 0] padded to whole class tiles."""
    nct = (model.no + 15) // 16
    W2 = np.asarray(model.W2, dtype=np.float64)
    Wp = np.zeros((nct * 16, 7 * 32), dtype=np.float32)
    Wp[:model.no, :2 * NS] = W2[:, 1:]
    hi_bits, hi_val = _bf16_bits(Wp)
    rest_bits = (Wp - hi_val).astype(np.float16).view(np.uint16)
    out = np.zeros((2, nct, 7, 64, 8), dtype=np.uint16)
    lane = np.arange(64)
    for ct in range(nct):
        rows = 16 * ct + (lane & 15)
        for ks in range(7):
            for j in range(8):
                cols = 32 * ks + 4 * (lane >> 4) + 16 * (j // 4) + j % 4
                out[0, ct, ks, :, j] = hi_bits[rows, cols]
                out[1, ct, ks, :, j] = rest_bits[rows, cols]
    assert out.size * 2 == _native.lib.ta_lstm_output_split_weight_bytes(model.no)
    bias = np.zeros(nct * 16, dtype=np.float32)
    bias[:model.no] = W2[:, 0]
    return out, bias


def _pack_lstm(model):
    nfl = _native.lib.ta_lstm_packed_weight_floats(0)
    wp = np.zeros((2, 7, 4, 38, 64), dtype=np.float32)
    assert wp.size == nfl
    peep = np.zeros((2, 3, 112), dtype=np.float32)
    lane = np.arange(64)
    for d, w in enumerate((model.fwd, model.rev)):
        for g, name in enumerate(("WGI", "WGF", "WGO", "WCI")):
            W = np.asarray(w[name], dtype=np.float64)
            Wp = np.zeros((112, 152), dtype=np.float64)
            Wp[:NS, 0:1 + NI] = W[:, 0:1 + NI]            # bias + x
            Wp[:NS, 52:152] = W[:, 1 + NI:]               # h
            for wv in range(7):
                for kk in range(38):
                    wp[d, wv, g, kk, :] = Wp[16 * wv + (lane & 15), 4 * kk + (lane >> 4)]
        for q, name in enumerate(("WIP", "WFP", "WOP")):
            peep[d, q, :NS] = np.asarray(w[name], dtype=np.float64)
    nct = (model.no + 15) // 16
    W2 = np.asarray(model.W2, dtype=np.float64)
    w2p = np.zeros((201, nct * 16), dtype=np.float32)
    w2p[0, :model.no] = W2[:, 0]                                   # bias seeds the accumulators
    for kk in range(50):
        for kq in range(4):
            w2p[1 + 4 * kk + kq, :model.no] = W2[:, 1 + 50 * kq + kk]
    return wp, peep, w2p


def _pack_lstm4(model):
    """weights of the four-lines-per-workgroup recurrence kernel (ta_lstm_forward mode 2): one k per MFMA,
    [dir 2][wave 7][k 152][lane 64] = W_gate(lane % 4)[unit 16 * wave + lane // 4][k], k as in mode 0"""
    wp = np.zeros((2, 7, 152, 64), dtype=np.float32)
    assert wp.size == _native.lib.ta_lstm_packed_weight_floats(2)
    lane = np.arange(64)
    for d, w in enumerate((model.fwd, model.rev)):
        Wp = np.zeros((4, 112, 152), dtype=np.float64)
        for g, name in enumerate(("WGI", "WGF", "WGO", "WCI")):
            W = np.asarray(w[name], dtype=np.float64)
            Wp[g, :NS, 0:1 + NI] = W[:, 0:1 + NI]            # bias + x
            Wp[g, :NS, 52:152] = W[:, 1 + NI:]               # h
        for wv in range(7):
            wp[d, wv] = Wp[lane % 4, 16 * wv + lane // 4, :].T
    return wp


def _pack_lstm_f64(model):
    """weights of the float64 recurrence (csrc/ta_lstm_f64.hip): A fragments of v_mfma_f64_16x16x4_f64.  The 400
    pre-activations of a step are tiled as 25 tiles of 16 rows = (4 gates) x (4 units), row i = 4 gate + unit-in-tile;
    a wave owns tiles 6 wave .. 6 wave + 5 (slots 0..5); slot 6 is tile 24, which the waves split along k.
    wh [dir 2][wave 4][slot 7][k-step 25][lane 64] = W_gate(i // 4)[unit 4 tile(wave, slot) + i % 4][49 + 4 kstep + lane // 16]
    wx [dir 2][tile 25][slot 13][lane 64]          = W_gate(2 (i // 8) + i % 2)[unit 4 tile + (i % 8) // 2][1 + 4 slot + lane // 16]
                                                     for slots 0..11 (B fragments over x alone, column i of a tile = position i of
                                                     the tile in a row of Gx); slot 12 = the column's bias W_gate(..)[unit ..][0] in
                                                     every lane of the column: the projection's accumulators start from it
    with i = lane % 16; peep [dir 2][WIP, WFP, WOP][100]."""
    wh = np.zeros((2, 4, 7, 25, 64), dtype=np.float64)
    wx = np.zeros((2, 25, 13, 64), dtype=np.float64)
    peep = np.zeros((2, 3, NS), dtype=np.float64)
    lane = np.arange(64)
    i, kq = lane % 16, lane // 16
    for d, w in enumerate((model.fwd, model.rev)):
        Wg = np.stack([np.asarray(w[name], dtype=np.float64) for name in ("WGI", "WGF", "WGO", "WCI")])   # [gate][unit][149]
        Wh = Wg[:, :, 1 + NI:]                                                                                  # [gate][unit][100]
        for tile in range(25):                      # B fragments: column j = lane % 16 is the (unit, gate) at in-row index 16 tile + j
            units, gates = 4 * tile + (i % 8) // 2, 2 * (i // 8) + i % 2
            for kk in range(12):
                wx[d, tile, kk] = Wg[gates, units, 1 + 4 * kk + kq]
            wx[d, tile, 12] = Wg[gates, units, 0]
        for wv in range(4):
            for s in range(7):
                units = 4 * (6 * wv + s if s < 6 else 24) + i % 4
                for kk in range(25):
                    wh[d, wv, s, kk] = Wh[i // 4, units, 4 * kk + kq]
        for q, name in enumerate(("WIP", "WFP", "WOP")):
            peep[d, q] = np.asarray(w[name], dtype=np.float64)
    # the four-line kernel's A fragments (v_mfma_f64_4x4x4_4b_f64: A[i][k] of block b in lane i + 4 b + 16 k; block =
    # unit-in-tile, row = gate): wh4 [dir 2][tile 25][k-step 25][lane 64] = W_gate(lane % 4)[unit 4 tile + (lane // 4) % 4][49 + 4 kstep + lane // 16]
    wh4 = np.zeros((2, 25, 25, 64), dtype=np.float64)
    for d, w in enumerate((model.fwd, model.rev)):
        Wh = np.stack([np.asarray(w[name], dtype=np.float64) for name in ("WGI", "WGF", "WGO", "WCI")])[:, :, 1 + NI:]
        for tile in range(25):
            for kk in range(25):
                wh4[d, tile, kk] = Wh[lane % 4, 4 * tile + (lane // 4) % 4, 4 * kk + lane // 16]
    lib = _native.lib
    assert (wh.size, wx.size, peep.size, wh4.size) == tuple(lib.ta_lstm_f64_weight_doubles(k) for k in range(4))
    return wh, wx, peep, wh4


_pool = None
COPY_THREADS = 4        # pool threads that copy prepared lines into the pinned staging buffer (8 until round 5: with the
                        # copies of a chunk running under the previous chunk's host stage, four keep up)


def _copy_pool():
    global _pool
    if _pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _pool = ThreadPoolExecutor(COPY_THREADS)
    return _pool


# Overrides for tests and timing tools: plain module attributes (this module reads no environment variables; the tools
# under tools/ translate their TA_* variables into these, tools/switches.py).  None = the product's own choice.
FORCE_GROUP = None              # 4 | 16: lines per workgroup of the recurrence (f32 / f64 modes)
FORCE_CLASS_SPLIT = None        # True | False: split mode, K3 + K4 per length class on side streams
CLASS_SPLIT_MIN_LINES = 384     # below this (a few pages) the recurrence's tail has nothing worth hiding
GROUP4_MAX_LINES = 2048         # exact-f32 mode: batches up to this size run in groups of 4 lines (see prepare)
F64_GROUP4_MAX_LINES = 1 << 30  # float64 mode: groups of 4 lines (lstm_seq4_f64_kernel) up to this batch size
F64_GX_MAX_ROWS = 3200000       # float64 mode: rows whose hoisted input projection (6 400 B per row) is held at once: 20 GB
F64_GX_KEEP_BYTES = 6 << 30     # ... and a scratch buffer larger than this is not kept by the recogniser between batches
# float64 mode: runs of this many groups (16 lines each) or more are pipelined per length class (forward_f64); cuts as
# shares of the run's groups.  Measured (tools/f64_time.py, lines of 800 .. 2000 columns): two halves 17.3 ms per 1 920
# lines against 19.0 one after the other (384 / 960 / 3 840 / 5 760 lines: 14.1 / 15.6 / 33.3 / 54.3 against 14.4 / 16.2 /
# 34.4 / 55.7); three or four classes are SLOWER (20 .. 21 ms: the projection of the last class is left a third of the
# CUs), and so are unequal halves (0.45 or 0.55: 19.3).
F64_CLASS_PIPELINE = True
F64_CLASS_MIN_GROUPS = 16
F64_CLASS_CUTS = (0.5,)
# class-split state: the side streams per device, and the verdict of the one-off timing check per (device, mode)
# (a recogniser on another GPU or in another mode is timed for itself); guarded by a lock -- page threads share it
_split_state = {"streams": {}, "ok": {}, "times_ms": {}}
_split_lock = __import__("threading").Lock()


def _device_key(device):
    return (device.type, device.index if device.index is not None else torch.cuda.current_device())


def _class_streams(device):
    """Three HIGH-priority side streams per device.  HIP spreads the streams of one priority over a few
    hardware queues in creation order, and two class streams that land on one queue run their classes one
    after the other (20 ms per 1 920 lines instead of 10.7); torch's high-priority pool is a queue set of its
    own, whose first three streams are neighbours -- measured independent of whatever normal-priority streams
    the process has made (tools/ocr_overlap_probe.py)."""
    key = _device_key(device)
    with _split_lock:
        if key not in _split_state["streams"]:
            _split_state["streams"][key] = [torch.cuda.Stream(device=device, priority=-1) for _ in range(3)]
        return _split_state["streams"][key]


def _class_split_wanted(rec, st):
    """FORCE_CLASS_SPLIT decides; otherwise batches of CLASS_SPLIT_MIN_LINES lines or more take the
    class split unless the one-off check below found it SLOWER in this process (another user of the
    high-priority queues): the first eligible batch is run both ways once, timed with events."""
    if FORCE_CLASS_SPLIT is not None:
        return FORCE_CLASS_SPLIT
    if st["n"] < CLASS_SPLIT_MIN_LINES or st["ngroups"] < 3 or st.get("continuation"):
        return False
    if rec.mode != 1:
        # exact-f32 mode: up to GROUP4_MAX_LINES its groups of four lines pack onto the CUs and leave no idle tail
        # to hide anything under; above that the output layer is a small share of the pass and the split
        # measured no gain (5 760 lines: 25.6 ms one launch each, 25.9 per class)
        return False
    key = _device_key(rec.device) + (rec.mode,)
    with _split_lock:
        verdict = _split_state["ok"].get(key)
    if verdict is None:
        times = {}
        for split in (True, False, True, False):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rec.run(st, decode=False, class_split=split)
            e1.record()
            e1.synchronize()
            times[split] = min(times.get(split, 1e30), e0.elapsed_time(e1))
        verdict = times[True] <= 1.02 * times[False]
        with _split_lock:
            _split_state["ok"][key] = verdict
            _split_state["times_ms"][key] = times
    return verdict


def _is_raw_strip(ln):
    """a raw greyscale strip: uint8, numpy on the host or torch on the device"""
    dt = getattr(ln, "dtype", None)
    return dt is torch.uint8 or (dt is not None and not isinstance(dt, torch.dtype) and dt == np.uint8)


class LineRecognizer(object):
    """precision:
    "f64" (the default since round 5) accumulates, carries state and evaluates the gate functions in float64 on the
    f64 matrix cores (csrc/ta_lstm_f64.hip) -- the arithmetic type of the reference's recogniser; the 1e-3 logit
    tolerance holds FREE-RUNNING on every line of the (chaotic) spec model (worst 1.5e-5 over 384 lines).
    "f32" (opt-in) runs the recurrence as an exact f32-input MFMA chain -- bit for bit a k-ordered float32 fmaf
    chain, i.e. what any float32 implementation of ocropy's loop computes.
    "split" (opt-in, fastest) runs it on the 16-bit matrix cores with split operands (weights
    bf16 + fp16 = 19 significant bits, activations three bf16 terms + one fp16, f32 accumulation,
    four products per k-step); its pre-activation error is about three times the f32 mode's.
    The two float32 modes hold the 1e-3 logit parity of the spec model per 128-step segment; FREE-RUNNING on that
    model at 800 .. 2000 columns, against the float64 restatement (96 lines per model,
    tools/ocr_mode_agreement.py, profiles/r05_ocr_mode_agreement.json): f32 median logit error
    9.1e-5 / 2.7e-5 (models 7001 / 7002), 88 / 94 of 96 lines within 1e-3; split 3.3e-4 / 3.5e-5,
    78 / 91 of 96; decoded characters: f32 identical on all 19 614 of the sample, split one different.
    ("bf16x3", the name of the split mode's first form, is accepted.)"""

    def __init__(self, model, device="cuda", precision=DEFAULT_PRECISION):
        if not torch.cuda.is_available():
            raise RuntimeError("text_alignment_amd needs an AMD GPU (MI355X); there is no CPU fallback")
        if precision not in ("split", "bf16x3", "f32", "f64"):
            raise ValueError("precision must be 'f32', 'split' or 'f64'")
        self.model = model
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            # pin the index: "cuda" alone means the CALLING THREAD's current device, and the pool threads that issue the
            # rows' transfer start on device 0 whatever device this rank's main thread has selected
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.mode = {"f32": 0, "f64": 3}.get(precision, 1)
        wp, peep, w2p = _pack_lstm(model)
        if self.mode == 1:
            wp = _pack_lstm_split(model)
        self.wp = torch.from_numpy(wp).to(self.device)
        self.wp4 = torch.from_numpy(_pack_lstm4(model)).to(self.device) if self.mode == 0 else None
        self.peep = torch.from_numpy(peep).to(self.device)
        self.w2p = torch.from_numpy(w2p).to(self.device)
        if self.mode == 1:
            w2s, bias = _pack_output_split(model)
            self.w2s = torch.from_numpy(w2s).to(self.device)
            self.w2bias = torch.from_numpy(bias).to(self.device)
        if self.mode == 3:
            self.wh64, self.wx64, self.peep64, self.wh64g4 = (torch.from_numpy(w).to(self.device) for w in _pack_lstm_f64(model))
            self._gx = {}           # Gx scratch per compute stream (process_batch runs consecutive chunks on two streams)

    # ---- host -> device ------------------------------------------------------------------
    def _stage_rows_begin(self, lines, row_start, rows):
        """Start copying all prepared lines into the pinned staging buffer (line k at rows row_start[k] ...).  The lines
        are copied (and, if need be, converted) by a few pool threads -- numpy releases the GIL for these copies, and at
        ~275 KB per line a page is 8 MB, so one thread's memcpy rate would bound the whole pipeline.  Returns what
        _stage_rows_end needs; the caller may do other host work in between (alignToOCR.process_batch finishes the
        previous chunk of pages there)."""
        if rows == 0:
            return None
        # two staging buffers in turn: the copies of a batch may start while the previous batch's transfer is in flight
        slots = getattr(self, "_stage_slots", None)
        if slots is None:
            slots = self._stage_slots = [{"buf": None, "done": None}, {"buf": None, "done": None}]
            self._stage_next = 0
        slot = slots[self._stage_next]
        self._stage_next ^= 1
        if slot["done"] is not None:
            slot["done"].synchronize()              # the transfer before last still reads this buffer
        if slot["buf"] is None or slot["buf"].shape[0] < rows:
            slot["buf"] = None
            slot["buf"] = torch.empty((int(rows * 1.25) + 1024, NI), dtype=torch.float32, pin_memory=True)
        view = slot["buf"].numpy()
        # float32 rows as they are: one native call per thread and no interpreter lock (ta_host_copy_pieces); anything that
        # needs converting (float64 rows, non-contiguous views) goes through numpy's assignment, line by line
        plain = all(ln.dtype == np.float32 and ln.flags.c_contiguous for ln in lines)

        def copy(span):
            a, b = span
            if plain:
                _native.host_copy_pieces(view, lines[a:b], 4 * NI * np.asarray(row_start[a:b], dtype=np.int64))
                return
            for k in range(a, b):
                view[row_start[k]:row_start[k] + lines[k].shape[0]] = lines[k]
        nthreads = min(COPY_THREADS, max(1, len(lines) // 64))
        if nthreads == 1:
            copy((0, len(lines)))
            return (rows, [], slot)
        done_rows = np.cumsum([ln.shape[0] for ln in lines])
        cuts = np.searchsorted(done_rows, np.linspace(0, rows, nthreads + 1)).tolist()
        cuts[0], cuts[-1] = 0, len(lines)
        self._upload_stream()                       # (created here, on the caller's thread and device)
        copies = [_copy_pool().submit(copy, (cuts[i], cuts[i + 1])) for i in range(nthreads)]
        # the transfer itself is issued by the pool as soon as the last copy is done (queued behind the copies: it cannot
        # starve them), not by the caller when it next looks -- in the page pipeline that was 2.4 ms of PCIe time per
        # chunk on the critical path between two chunks' kernels
        return (rows, [_copy_pool().submit(self._issue_upload, copies, rows, slot)], slot)

    def _issue_upload(self, copies, rows, slot):
        for f in copies:
            f.result()
        up = self._upload_stream()
        with torch.cuda.stream(up):
            x_dev = torch.empty((rows, NI), dtype=torch.float32, device=self.device)
            x_dev.copy_(slot["buf"][:rows], non_blocking=True)
            # the staging buffer is reused by the batch after next: the transfer has to be over before then
            slot["done"] = torch.cuda.Event()
            slot["done"].record(up)
        return x_dev

    def _upload_stream(self):
        up = getattr(self, "_up_stream", None)
        if up is None:
            up = self._up_stream = torch.cuda.Stream(device=self.device)
        return up

    def _stage_rows_end(self, pending):
        """Wait for the staging copies and send the rows over PCIe in one asynchronous transfer ON THE UPLOAD STREAM: the
        copy engine then works beside the kernels of whatever batch the compute stream is busy with, and the compute
        stream only waits for the event.  The device tensor comes from the upload stream's pool and is recorded as
        used by the compute stream (the allocator must not hand it out again before that stream is done with it)."""
        if pending is None:
            return torch.zeros((1, NI), dtype=torch.float32, device=self.device)
        rows, futures, slot = pending
        main = torch.cuda.current_stream(self.device)
        x_dev = futures[0].result() if futures else self._issue_upload([], rows, slot)
        main.wait_event(slot["done"])
        x_dev.record_stream(main)
        return x_dev

    def _upload_rows(self, lines, row_start, rows):
        """All prepared lines as one [rows, 48] float32 device tensor (line k at rows row_start[k] ...)."""
        return self._stage_rows_end(self._stage_rows_begin(lines, row_start, rows))

    SPAN_MERGE_GAP = 256        # rows: two lines of a page-locked block this close go over in one transfer (<= 48 KB idle)

    def _span_rows_begin(self, lines, rows):
        """Lines that are RowSpans (page.RowBlock: rows in page-locked or device memory): nothing is copied on the host.
        Every run of neighbouring spans of a page-locked block crosses PCIe as it lies, in one asynchronous transfer
        issued HERE on the upload stream (a chunk of the page pipeline: under the previous chunk's kernels); spans of a
        device block are read where they are.  Returns the per-line source ADDRESSES for ta_rows_gather and what has to
        stay alive until the gather has run."""
        if rows == 0:
            return None
        dev = self.device
        src = np.zeros(len(lines), dtype=np.int64)
        by_block = {}
        for k, ln in enumerate(lines):
            by_block.setdefault(id(ln.block), (ln.block, []))[1].append(k)
        keep, done = [], None
        up = self._upload_stream()
        for block, ks in by_block.values():
            t = block.tensor
            if block.kind == "device":
                if t.device != dev:
                    raise ValueError("a device RowBlock must live on the recogniser's device (%s, not %s)" % (dev, t.device))
                base = t.data_ptr()
                for k in ks:
                    src[k] = base + 4 * NI * lines[k].start
                keep.append(t)
                continue
            ks = sorted(ks, key=lambda k: lines[k].start)
            runs, a, b, members = [], lines[ks[0]].start, lines[ks[0]].stop, [ks[0]]
            for k in ks[1:]:
                if lines[k].start - b <= self.SPAN_MERGE_GAP:
                    b = max(b, lines[k].stop)
                    members.append(k)
                else:
                    runs.append((a, b, members))
                    a, b, members = lines[k].start, lines[k].stop, [k]
            runs.append((a, b, members))
            with torch.cuda.stream(up):
                for a, b, members in runs:
                    d = torch.empty((b - a, NI), dtype=torch.float32, device=dev)
                    d.copy_(t[a:b], non_blocking=True)
                    keep.append(d)
                    base = d.data_ptr()
                    for k in members:
                        src[k] = base + 4 * NI * (lines[k].start - a)
                done = torch.cuda.Event()
                done.record(up)
        return {"src": src, "keep": keep, "done": done, "rows": rows}

    # ---- batched device pass -------------------------------------------------------------
    def prepare(self, lines, defer=False, _measured=None):
        """Upload lines and allocate outputs.  A line is either a prepared (T, 48) float array
        (ink = 1, padded) or a raw 2-D uint8 strip (white background; a host array or a tensor already
        on the device), which is normalised on the device (lineest_gpu, csrc/ta_lineest.hip) without a
        host round trip.  defer = True: what can be STARTED is started and the call returns; `complete(st)` -- or
        `run(st)` -- waits for it and does the rest.  Host rows: the staging copies are started.  Raw strips (all lines
        raw): the normaliser's measuring pass is enqueued -- its output widths are data-dependent, so the batch's layout
        itself waits for it; a caller with other host work (alignToOCR.process_batch: the second stages of older chunks)
        does it before complete(), instead of waiting here for kernels queued behind the previous chunk's recogniser."""
        from .page import RowSpan
        nspans = sum(1 for ln in lines if isinstance(ln, RowSpan))
        if 0 < nspans < len(lines):         # a mixed batch: the spans go the way of host arrays (a device span is downloaded)
            lines = [ln.numpy() if isinstance(ln, RowSpan) else ln for ln in lines]
            nspans = 0
        raw = [] if nspans else [k for k, ln in enumerate(lines) if _is_raw_strip(ln)]
        n = len(lines)
        T = np.zeros(n, dtype=np.int64)

        def layout(Tl):
            """first row of every line: lines sorted by length (longest first, the order the groups of 16 take
            them in), so that the rows of any run of groups are one contiguous range of x / hout / summary"""
            order_ = np.argsort(-Tl, kind="stable")
            start = np.empty(len(Tl), dtype=np.int64)
            start[order_] = np.cumsum(Tl[order_]) - Tl[order_]
            return start
        x_raw, T_raw = None, None
        if raw:
            from . import lineest_gpu
            if len(raw) == n and (defer or _measured is not None):
                if _measured is None:
                    return {"_raw": (lines, lineest_gpu.measure_strips_begin(lines, device=self.device))}
                lineest_gpu.measure_strips_end(_measured)                       # the wait; sizes known from here on
                x_raw, T_raw = lineest_gpu.resample_strips(_measured, 0, n, layout=layout)
            else:
                x_raw, T_raw, _ = lineest_gpu.normalize_strips([lines[k] for k in raw], device=self.device,
                                                               layout=layout if len(raw) == n else None)
            T[raw] = T_raw
        if nspans:                                   # spans of RowBlocks: (T, 48) by construction
            T[:] = [ln.stop - ln.start for ln in lines]
        else:
            raw_set = set(raw)
            for k, ln in enumerate(lines):
                if k in raw_set:
                    continue
                if ln.ndim != 2 or ln.shape[1] != NI:
                    raise ValueError("a prepared line must have shape (T, 48)")
                T[k] = ln.shape[0]
        if n and T.max() > MAX_T:
            raise RecognitionError("input too large for LSTM model")
        order = np.argsort(-T, kind="stable")
        # Lines per workgroup of the recurrence.  Exact-f32 mode has two kernels that compute the same bits:
        # groups of 16 (v_mfma_f32_16x16x4) and groups of 4 (v_mfma_f32_4x4x1, a quarter of the cost per step).
        # A step costs the same however few of a group's rows are lines and a batch takes as long as its
        # longest group, so small and medium batches take the small groups (a page: 3.1 instead of 10 ms;
        # 1 920 lines: 960 short workgroups pack onto the CUs instead of 240 long ones idling for the
        # longest, 8.7 instead of 10.0 ms); large batches keep the 16-line kernel, whose step has less
        # overhead per line (2 560 lines: 12.7 against 13.5 ms for the whole pass).
        # Float64 mode has the same pair (v_mfma_f64_16x16x4 / v_mfma_f64_4x4x4_4b, equal to the bit as well); there
        # the small groups win at every batch size measured (F64_GROUP4_MAX_LINES).
        G = 4 if ((self.mode == 0 and n <= GROUP4_MAX_LINES) or (self.mode == 3 and n <= F64_GROUP4_MAX_LINES)) else 16
        if FORCE_GROUP is not None and self.mode in (0, 3):
            G = FORCE_GROUP
        ngroups = (n + G - 1) // G
        group_lines = np.full((max(ngroups, 1), G), -1, dtype=np.int32)
        group_lines.reshape(-1)[:n] = order
        row_start = layout(T) if n else np.zeros(0, dtype=np.int64)
        rows = int(T.sum())
        raw_set = set(raw)
        host = [k for k in range(n) if k not in raw_set]
        pending = None
        if nspans:
            pending = self._span_rows_begin(lines, rows)
            x_dev = None
        elif not raw:
            pending = self._stage_rows_begin(lines, row_start, rows)
            x_dev = None
        elif not host:
            x_dev = x_raw                           # the normaliser wrote its rows in this layout
        else:                                   # mixed batch: stitch the two sources together
            x_dev = torch.empty((rows, NI), dtype=torch.float32, device=self.device)
            xh = torch.from_numpy(np.ascontiguousarray(
                np.concatenate([lines[k] for k in host], axis=0).astype(np.float32))).to(self.device)
            ph = pr = 0
            for k in range(n):
                t = int(T[k])
                if k in raw_set:
                    x_dev[row_start[k]:row_start[k] + t] = x_raw[pr:pr + t]; pr += t
                else:
                    x_dev[row_start[k]:row_start[k] + t] = xh[ph:ph + t]; ph += t
        # first row of every group (+ the end): group g owns rows group_row[g] .. group_row[g + 1]
        group_row = np.zeros(max(ngroups, 1) + 1, dtype=np.int64)
        if n:
            group_row[:ngroups] = row_start[order[::G]]
        group_row[ngroups:] = rows
        st = {"n": n, "rows": rows, "T_host": T, "row_start_host": row_start, "ngroups": ngroups,
              "group_row_host": group_row, "group_size": G,
              "_pending": (pending, x_dev, group_lines, len(lines))}
        if not (defer and not raw):
            self.complete(st)
        return st

    def complete(self, st):
        """the device part of prepare(): the rows' transfer, the batch's metadata, the output buffers"""
        if "_raw" in st:                        # raw strips whose measuring pass was enqueued by prepare(defer=True)
            lines, ms = st.pop("_raw")
            st.update(self.prepare(lines, defer=False, _measured=ms))
            return st
        if "_pending" not in st:
            return st
        pending, x_dev, group_lines, nlines = st.pop("_pending")
        n, rows, T, row_start = st["n"], st["rows"], st["T_host"], st["row_start_host"]
        lines = [None] * nlines
        spans = pending if isinstance(pending, dict) else None
        if x_dev is None and spans is None:
            x_dev = self._stage_rows_end(pending)
        dev = self.device
        meta = [row_start if n else np.zeros(1, np.int64), T.astype(np.int32) if len(lines) else np.zeros(1, np.int32),
                group_lines]
        if spans is not None:
            meta.append(spans["src"])
        up = _native.upload_packed(meta, dev)
        st["row_off"], st["T"], st["group_lines"] = up[:3]
        if spans is not None:
            # the permutation into the recogniser's row order, on the compute stream, behind the blocks' transfers
            main = torch.cuda.current_stream(dev)
            if spans["done"] is not None:
                main.wait_event(spans["done"])
            x_dev = torch.empty((rows, NI), dtype=torch.float32, device=dev)
            _native.check(_native.lib.ta_rows_gather(up[3].data_ptr(), st["row_off"].data_ptr(), st["T"].data_ptr(), n,
                                                     int(T.max()), x_dev.data_ptr(), main.cuda_stream), "ta_rows_gather")
            for t in spans["keep"]:
                t.record_stream(main)           # (allocated on the upload stream, or the caller's block: read by this one)
        st["x"] = x_dev
        st["hout"] = torch.empty((max(rows, 1), 2 * NS), dtype=torch.float32, device=dev)
        st["probs"] = None            # full probabilities only on request (tests, inspection)
        st["logits"] = None
        st["summary"] = torch.empty((max(rows, 1), 4), dtype=torch.float32, device=dev)
        st["dec_t"] = torch.zeros(max(rows, 1), dtype=torch.int32, device=dev)
        st["dec_c"] = torch.zeros(max(rows, 1), dtype=torch.int32, device=dev)
        # dec_n[k] = decoded characters of line k; ONE more word at the end is the batch's device status (zero; the float64
        # recurrence ORs TA_LSTM_F64_PARTS_LATE into it): it comes back with the counts and check_status() raises on it
        st["dec_n"] = torch.zeros(max(len(lines), 1) + 1, dtype=torch.int32, device=dev)
        return st

    def run(self, st, want_logits=False, lstm=True, output=True, decode=True, from_probs=False, class_split=None):
        """Enqueue K3, K4, K5 behind torch's current stream.  By default K4 emits only the 16-byte
        per-timestep summaries and K5 decodes from them; want_logits / from_probs also
        materialise the (rows, No) probabilities (and logits) and decode from those.

        In split-operand mode a batch of at least CLASS_SPLIT_MIN_LINES lines runs K3 and K4 per length CLASS (the longest tenth
        of the groups, the next fifth, the rest; rows are laid out in group order, so a class is one
        row range) on three side streams: the recurrence is a chain of T dependent steps per group, the
        longest group sets its time and the CUs of the short groups idle towards the end -- the output
        layer of the classes that are done runs there instead of behind the whole recurrence (split mode:
        5.85 -> 5.2 ms per 1 920 lines, against 4.7 for the recurrence alone).  Same kernels on the same rows:
        results are bit for bit those of the single launches.  class_split = True / False forces the choice
        (tests, timing); None leaves it to _class_split_wanted."""
        self.complete(st)
        if st["n"] == 0:
            return
        lib = _native.lib
        cs = torch.cuda.current_stream(self.device)
        stream = cs.cuda_stream
        cont = st.get("continuation")          # (h0, c0, tstart) device tensors, or None: fresh lines
        full = want_logits or from_probs
        if full and st["probs"] is None:
            shape = (max(st["rows"], 1), self.model.no)
            st["probs"] = torch.empty(shape, dtype=torch.float32, device=self.device)
            st["logits"] = torch.empty(shape, dtype=torch.float32, device=self.device)

        ng, G = st["ngroups"], st["group_size"]

        def forward_f64(g0, g1, stream_):
            """float64 mode: the input projection of every row of a run of groups in one GEMM (Gx, 6 400 bytes per row,
            held for at most F64_GX_MAX_ROWS rows at a time -- runs of groups are contiguous row ranges), then the
            recurrence over those groups.

            A run of F64_CLASS_MIN_GROUPS groups or more is cut into length classes (groups are in order of falling length;
            F64_CLASS_CUTS: the longer half and the shorter half), each with its own piece of the Gx buffer: the
            projections are enqueued on stream_ class by class, and the recurrence of every class but the last goes
            to a side stream as soon as ITS projection is done.  The longest lines set the recurrence's time (a chain
            of T dependent steps), so they start after half of the projection instead of all of it, and the
            projection of the shorter class runs on the CUs that recurrence has not claimed.  Same kernels on the
            same rows: results are bit for bit those of the one-after-the-other order."""
            grow = st["group_row_host"]
            gx_buf = None
            # (groups of four lines leave the projection no idle CUs to hide under: measured no gain, so no side streams)
            piped = F64_CLASS_PIPELINE and stream_ == stream and G != 4
            a = g0
            while a < g1:
                b = a + 1
                while b < g1 and int(grow[b + 1] - grow[a]) <= F64_GX_MAX_ROWS:
                    b += 1
                r0, r1 = int(grow[a]), int(grow[b])
                need = lib.ta_lstm_f64_gx_bytes(r1 - r0)
                gx_buf = self._gx.get(stream)
                if gx_buf is None or gx_buf.numel() * 8 < need:
                    gx_buf = None
                    self._gx.pop(stream, None)                       # (free the old one first)
                    gx_buf = self._gx[stream] = torch.empty(max(need // 8, 1), dtype=torch.float64, device=self.device)
                cuts = [a, b]
                if piped and (b - a) * G >= 16 * F64_CLASS_MIN_GROUPS:
                    cuts = sorted(set([a, b] + [a + int(round(f * (b - a))) for f in F64_CLASS_CUTS]))
                side = _class_streams(self.device) if len(cuts) > 2 else []
                used, off = [], 0
                for k in range(len(cuts) - 1):
                    ca, cb = cuts[k], cuts[k + 1]
                    c0, c1 = int(grow[ca]), int(grow[cb])
                    gx = gx_buf.data_ptr() + off
                    off += lib.ta_lstm_f64_gx_bytes(c1 - c0)
                    _native.check(lib.ta_lstm_xproj_f64(st["x"].data_ptr() + 4 * NI * c0, c1 - c0, self.wx64.data_ptr(),
                                                        gx, stream_), "ta_lstm_xproj_f64")
                    seq_stream = stream_
                    if k < len(cuts) - 2:                            # every class but the last: its own stream
                        done = torch.cuda.Event()
                        done.record(cs)
                        side[k].wait_event(done)
                        seq_stream = side[k].cuda_stream
                        used.append(side[k])
                    seq = lib.ta_lstm_forward_f64_g4 if G == 4 else lib.ta_lstm_forward_f64
                    _native.check(seq(
                        gx, c0, c1 - c0, st["row_off"].data_ptr(), st["T"].data_ptr(),
                        st["group_lines"].data_ptr() + 4 * G * ca, cb - ca, (self.wh64g4 if G == 4 else self.wh64).data_ptr(), self.peep64.data_ptr(),
                        st["hout"].data_ptr(), cont[0].data_ptr() if cont else None, cont[1].data_ptr() if cont else None,
                        cont[2].data_ptr() if cont else None, st["dec_n"].data_ptr() + 4 * (st["dec_n"].numel() - 1),
                        seq_stream), "ta_lstm_forward_f64")
                for sd in used:                                      # (also orders the reuse of the buffer)
                    cs.wait_stream(sd)
                a = b
            # a large batch's scratch goes back to torch's caching allocator (it stays cached there: the next large batch
            # gets it back at once, and meanwhile other tensors may use the memory); small ones are kept by the recogniser
            if gx_buf is not None and gx_buf.numel() * 8 > F64_GX_KEEP_BYTES:
                gx_buf.record_stream(cs)
                self._gx.pop(stream, None)

        # (h0 / c0 / tstart of a continuation are indexed by LINE id, not by position in the launch: a launch over
        # groups g0 .. g1 passes the whole arrays, un-offset, whichever run of groups it covers)
        def forward(g0, g1, stream_):
            if self.mode == 3:
                return forward_f64(g0, g1, stream_)
            _native.check(lib.ta_lstm_forward(
                st["x"].data_ptr(), st["row_off"].data_ptr(), st["T"].data_ptr(),
                st["group_lines"].data_ptr() + 4 * G * g0, g1 - g0, (self.wp4 if G == 4 else self.wp).data_ptr(),
                self.peep.data_ptr(), st["hout"].data_ptr(), 2 if G == 4 else self.mode,
                cont[0].data_ptr() if cont else None, cont[1].data_ptr() if cont else None,
                cont[2].data_ptr() if cont else None, stream_), "ta_lstm_forward")

        def outputs(r0, r1, stream_):
            no = self.model.no
            probs = st["probs"].data_ptr() + 4 * no * r0 if full else None
            logits = st["logits"].data_ptr() + 4 * no * r0 if full else None
            hout, summary = st["hout"].data_ptr() + 4 * 2 * NS * r0, st["summary"].data_ptr() + 16 * r0
            if self.mode == 1:
                _native.check(lib.ta_lstm_output_split(hout, r1 - r0, self.w2s.data_ptr(), self.w2bias.data_ptr(), no,
                                                       probs, logits, summary, stream_), "ta_lstm_output_split")
            else:
                _native.check(lib.ta_lstm_output(hout, r1 - r0, self.w2p.data_ptr(), no, probs, logits, summary,
                                                 stream_), "ta_lstm_output")
        if class_split is None:
            class_split = lstm and output and _class_split_wanted(self, st)
        if self.mode == 3:
            # float64 mode pipelines its own two kernels over length classes (forward_f64) with pieces of ONE Gx buffer;
            # three concurrent per-class launches of it would share that buffer -- the class split is not for this mode
            class_split = False
        if class_split and lstm and output and ng >= 3:
            cuts = [0, max(1, int(round(0.1 * ng))), max(2, int(round(0.3 * ng))), ng]
            side = _class_streams(self.device)
            for c in range(3):
                side[c].wait_stream(cs)
                forward(cuts[c], cuts[c + 1], side[c].cuda_stream)
                r0, r1 = int(st["group_row_host"][cuts[c]]), int(st["group_row_host"][cuts[c + 1]])
                if r1 > r0:
                    outputs(r0, r1, side[c].cuda_stream)
            for c in range(3):
                cs.wait_stream(side[c])
        else:
            if lstm:
                forward(0, ng, stream)
            if output:
                outputs(0, st["rows"], stream)
        if decode and from_probs:
            _native.check(lib.ta_decode(
                st["probs"].data_ptr(), st["row_off"].data_ptr(), st["T"].data_ptr(), st["n"],
                self.model.no, THRESHOLD, st["dec_t"].data_ptr(), st["dec_c"].data_ptr(),
                st["dec_n"].data_ptr(), st["row_off"].data_ptr(), stream), "ta_decode")
        elif decode:
            _native.check(lib.ta_decode_summary(
                st["summary"].data_ptr(), st["row_off"].data_ptr(), st["T"].data_ptr(), st["n"],
                THRESHOLD, st["dec_t"].data_ptr(), st["dec_c"].data_ptr(),
                st["dec_n"].data_ptr(), st["row_off"].data_ptr(), stream), "ta_decode_summary")

    @staticmethod
    def check_status(dec_n_host):
        """dec_n as it came back from the device (counts + the status word, see complete()): a batch whose recurrence
        reported trouble has no usable result -- its outputs are NaN from the step it happened -- and says so HERE, instead
        of handing back whatever the decoder made of them."""
        word = int(dec_n_host[-1])
        if word != 0:
            raise RuntimeError("line recogniser: the device reported status %#x for this batch (bit 0: a workgroup of the "
                               "float64 recurrence gave up waiting for the partial sums its waves exchange; its outputs "
                               "are NaN from that step on) -- the batch's characters are not usable" % word)

    def decoded(self, st):
        """Host lists [(t, class), ...] per line (translate_back order)."""
        if st["n"] == 0:
            return []
        dt = st["dec_t"].cpu().numpy()
        dc = st["dec_c"].cpu().numpy()
        dn = st["dec_n"].cpu().numpy()
        self.check_status(dn)
        out = []
        for b in range(st["n"]):
            o = int(st["row_start_host"][b])
            k = int(dn[b])
            out.append(list(zip(dt[o:o + k].tolist(), dc[o:o + k].tolist())))
        return out

    def recognise(self, lines, want_probs=False, from_probs=False):
        st = self.prepare(lines)
        self.last_T = st["T_host"]              # timesteps per line (raw strips: known only now)
        self.run(st, want_logits=want_probs, from_probs=from_probs)
        dec = self.decoded(st)
        if not want_probs:
            return dec
        probs = st["probs"].cpu().numpy()
        logits = st["logits"].cpu().numpy()
        states = st["hout"].cpu().numpy()
        sl = [slice(int(st["row_start_host"][b]), int(st["row_start_host"][b] + st["T_host"][b])) for b in range(st["n"])]
        return dec, [probs[s] for s in sl], [logits[s] for s in sl], [states[s] for s in sl]

    # ---- wire format ------------------------------------------------------------------------
    def llocs(self, decoded, T, raw_width):
        """(char, x) pairs of ocropus-rpred's --llocs output for one line: x in raw strip pixels
        from the strip's left edge (parsed at reference alignToOCR.py:157-170)."""
        scale = float(raw_width) / (T - 2 * PAD)
        codec = self.model.codec
        return [(codec[c], (t - PAD) * scale) for (t, c) in decoded]


def llocs_text(llocs):
    return "".join("%s\t%.1f\n" % (ch, x) for ch, x in llocs)
