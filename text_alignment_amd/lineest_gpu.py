"""Device side of the line normaliser (csrc/ta_lineest.hip): raw greyscale strips in, the
recogniser's input rows out, without a host round trip: ocropy 1.3.3's CenterNormalizer +
prepare_line (parity unpinned).  The GPU tests check it against the
float64 scipy restatement oracle/lineest_ref.py.  Strips are 2-D uint8 greyscale images, as the PNG
files the reference hands to ocropus-rpred (alignToOCR.py:131-132); there is no host path.
"""
import numpy as np
import torch

from . import _native
from . import page as page_mod

TARGET_HEIGHT = 48          # CenterNormalizer target_height
PAD = 16                    # prepare_line pad (Appendix B.2)

_kernels = {}
_stage = {}                 # pinned staging buffer for host strips + the event of its last transfer
_stage_lock = None
_pool = None


def _upload_host_strips(strips, pix_off, dev):
    """host strips -> one packed uint8 device tensor: copied into a pinned staging buffer by a few threads
    (numpy releases the GIL for the copies; 64 pages of strips are 160 MB, a single memcpy stream and a
    pageable transfer were a third of such a batch) and sent in one asynchronous transfer"""
    global _stage_lock, _pool
    import threading
    if _stage_lock is None:
        _stage_lock = threading.Lock()
    total = int(pix_off[-1])
    with _stage_lock:
        st = _stage.get("buf")
        if _stage.get("done") is not None:
            _stage["done"].synchronize()                 # the previous batch's transfer still reads the buffer
        if st is None or st.numel() < total:
            st = _stage["buf"] = torch.empty(int(total * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)
        view = st.numpy()

        plain = all(isinstance(s_, np.ndarray) and s_.flags.c_contiguous for s_ in strips)

        def copy(span):
            a, b = span
            if plain:                                    # one native call, no interpreter lock (ta_host_copy_pieces)
                _native.host_copy_pieces(view, strips[a:b], pix_off[a:b])
                return
            for k in range(a, b):
                view[pix_off[k]:pix_off[k + 1]] = np.asarray(strips[k]).reshape(-1)
        n = len(strips)
        nthreads = min(8, max(1, total >> 22))           # ~4 MB per thread at least
        if nthreads == 1:
            copy((0, n))
        else:
            if _pool is None:
                from concurrent.futures import ThreadPoolExecutor
                _pool = ThreadPoolExecutor(8, thread_name_prefix="ta-strips")
            cuts = np.searchsorted(pix_off, np.linspace(0, total, nthreads + 1)).tolist()
            cuts[0], cuts[-1] = 0, n
            list(_pool.map(copy, [(cuts[i], cuts[i + 1]) for i in range(nthreads)]))
        d_pix = torch.empty(total, dtype=torch.uint8, device=dev)
        d_pix.copy_(st[:total], non_blocking=True)
        _stage["done"] = torch.cuda.Event()
        _stage["done"].record()
    return d_pix


def _gauss_weights(sigma):
    """scipy.ndimage's 1-D gaussian kernel (truncate = 4.0), bit for bit"""
    radius = int(4.0 * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum(), radius


def _line_kernels(h):
    if h not in _kernels:
        _kernels[h] = [_gauss_weights(s) for s in (h * 0.5, h * 1.0, h * 0.3)]      # rows, columns, centre line
    return _kernels[h]


class MeasuredStrips(object):
    """State between the two passes of the normaliser: the strips' pixels and the measuring pass's results on the device,
    the output widths on the host (`wo`; T = wo + 32 timesteps).  resample_strips() turns any run of strips [a, b) of it
    into recogniser rows without another wait for the device."""
    __slots__ = ("n", "dev", "d_pix", "d_pix_off", "d_hh", "d_ww", "d_col_off", "center", "minmax", "r", "wout",
                 "arg", "col_off", "wo", "T", "_sizes_host", "_sizes_ready")


def measure_strips(strips, device="cuda"):
    """First pass (ta_linenorm_measure) over `strips` -- 2-D uint8 images (white background), each a host array or a
    tensor already on the device: centre line, band height and OUTPUT WIDTH of every strip.  The widths are
    data-dependent and the caller sizes its buffers with them, so this pass ends with the one wait for the device the
    normaliser needs.  measure_strips_begin + measure_strips_end: a caller with other host work puts it in between."""
    return measure_strips_end(measure_strips_begin(strips, device))


def measure_strips_begin(strips, device="cuda"):
    """the measuring pass ENQUEUED on torch's current stream -- upload, kernels, the read-back of the sizes into
    page-locked memory behind an event -- and nothing waited for: measure_strips_end(ms) does that"""
    dev = torch.device(device)
    lib = _native.lib
    n = len(strips)
    ms = MeasuredStrips()
    ms.n, ms.dev = n, dev
    ms._sizes_host = ms._sizes_ready = None
    if n == 0:
        ms.wo = ms.T = np.zeros(0, np.int64)
        return ms
    hh = np.zeros(n, np.int32); ww = np.zeros(n, np.int32)
    spans = [isinstance(s, page_mod.DeviceStrip) for s in strips]
    on_device = [sp or isinstance(s, torch.Tensor) for sp, s in zip(spans, strips)]
    want_index = torch.cuda.current_device() if dev.index is None else dev.index
    checked = set()
    for k, s in enumerate(strips):
        if spans[k]:
            buf = s.buffer
            if id(buf) not in checked:                # (a batch's strips share a few buffers: checked once each)
                if buf.dim() != 1 or buf.dtype != torch.uint8 or buf.device.type != "cuda" or buf.device.index != want_index:
                    raise TypeError("a device strip lies in a 1-D uint8 buffer on %s" % dev)
                checked.add(id(buf))
            if s.shape[0] * s.shape[1] == 0:
                raise ValueError("empty or constant text-line image")
            if s.start < 0 or s.start + s.shape[0] * s.shape[1] > buf.numel():
                raise ValueError("a device strip lies outside its buffer")
        elif on_device[k]:
            if (s.dim() != 2 or s.dtype != torch.uint8 or not s.is_contiguous() or s.device.type != "cuda" or
                    s.device.index != want_index):
                raise TypeError("a device strip is a contiguous 2-D uint8 tensor on %s" % dev)
            if s.numel() == 0:
                raise ValueError("empty or constant text-line image")
        else:
            s = np.asarray(s)
            if s.ndim != 2 or s.dtype != np.uint8:
                raise TypeError("the device normaliser takes 2-D uint8 strips")
            if s.size == 0:
                raise ValueError("empty or constant text-line image")
        hh[k], ww[k] = s.shape
    pix_off = np.zeros(n + 1, np.int64); np.cumsum(hh.astype(np.int64) * ww, out=pix_off[1:])
    col_off = np.zeros(n + 1, np.int64); np.cumsum(ww, out=col_off[1:])
    # gaussian kernels: one set per distinct strip height, offsets point at the centre taps
    gw_parts, gw_off, gr, where, pos = [], np.zeros((n, 3), np.int64), np.zeros((n, 3), np.int32), {}, 0
    for k in range(n):
        h = int(hh[k])
        if h not in where:
            offs = []
            for wts, rad in _line_kernels(h):
                gw_parts.append(wts); offs.append((pos + rad, rad)); pos += len(wts)
            where[h] = offs
        for q, (o, rad) in enumerate(where[h]):
            gw_off[k, q], gr[k, q] = o, rad
    gw = np.concatenate(gw_parts)

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    if not any(on_device):
        d_pix = _upload_host_strips(strips, pix_off, dev)
    else:
        # neighbours that follow each other in the same buffer are ONE slice of it (the preprocessing cuts a batch of
        # pages' strips into one packed buffer, in order): a chunk's ~500 strips are two or three pieces
        parts, k = [], 0
        while k < n:
            s = strips[k]
            if spans[k]:
                end, j = s.start + int(hh[k]) * int(ww[k]), k + 1
                while j < n and spans[j] and strips[j].buffer is s.buffer and strips[j].start == end:
                    end += int(hh[j]) * int(ww[j])
                    j += 1
                parts.append(s.buffer[s.start:end])
                k = j
            else:
                parts.append(s.reshape(-1) if on_device[k] else up(np.asarray(s).ravel()))
                k += 1
        d_pix = parts[0] if len(parts) == 1 else torch.cat(parts)
    # the batch's metadata in one transfer (nine small arrays: a `.to(device)` from pageable memory each was 2 ms of host time)
    (ms.d_pix_off, ms.d_hh, ms.d_ww, d_gw, d_gw_off, d_gr, ms.d_col_off, d_ws_off) = _native.upload_packed(
        [pix_off[:-1].copy(), hh, ww, gw, gw_off, gr, col_off[:-1].copy(), 3 * pix_off[:-1]], dev)
    ws = torch.empty(3 * int(pix_off[-1]), dtype=torch.float64, device=dev)
    ms.arg = torch.empty(int(col_off[-1]), dtype=torch.int32, device=dev)
    ms.center = torch.empty_like(ms.arg)
    sizes = torch.empty(3 * n, dtype=torch.int32, device=dev)          # wout [n] | minmax [2 n]: read back in one piece
    ms.wout, ms.minmax = sizes[:n], sizes[n:]
    ms.r = torch.empty(n, dtype=torch.int32, device=dev)
    ms.d_pix, ms.col_off = d_pix, col_off
    stream = torch.cuda.current_stream(dev).cuda_stream
    _native.check(lib.ta_linenorm_measure(
        d_pix.data_ptr(), ms.d_pix_off.data_ptr(), ms.d_hh.data_ptr(), ms.d_ww.data_ptr(), n,
        d_gw.data_ptr(), d_gw_off.data_ptr(), d_gr.data_ptr(), ws.data_ptr(), d_ws_off.data_ptr(),
        ms.arg.data_ptr(), ms.center.data_ptr(), ms.d_col_off.data_ptr(), ms.minmax.data_ptr(),
        ms.r.data_ptr(), ms.wout.data_ptr(), stream), "ta_linenorm_measure")
    # output sizes are data-dependent: they come back through page-locked memory behind an event (a plain .cpu() would
    # make the host wait HERE, for kernels that may be queued behind another batch's recogniser)
    ms._sizes_host = torch.empty(3 * n, dtype=torch.int32, pin_memory=True)
    ms._sizes_host.copy_(sizes, non_blocking=True)
    ms._sizes_ready = torch.cuda.Event()
    ms._sizes_ready.record()
    ws.record_stream(torch.cuda.current_stream(dev))       # (freed below while the kernels that use it may still be queued)
    del ws
    ms.wo = ms.T = None
    return ms


def measure_strips_end(ms):
    """THE wait of the normaliser: the measured sizes on the host (ms.wo, ms.T); ValueError for a constant strip"""
    if ms.wo is not None:
        return ms
    ms._sizes_ready.synchronize()
    n = ms.n
    sized = ms._sizes_host.numpy()
    ms.wo = sized[:n].astype(np.int64)
    if bool((sized[n::2] == sized[n + 1::2]).any()):      # the measuring pass found a strip's minimum = its maximum
        raise ValueError("empty or constant text-line image")
    ms.T = ms.wo + 2 * PAD
    ms._sizes_host = ms._sizes_ready = None
    return ms


def resample_strips(ms, a=0, b=None, layout=None):
    """Second pass (ta_linenorm_resample) over strips [a, b) of a measured batch: (x, T) with x = float32 device tensor
    [sum T, 48] -- strip a + k owns rows sum(T[:k]) .. + T[k], or, with `layout`, rows layout(T)[k] .. + T[k] (a
    permutation of those ranges, for a caller that wants the lines in an order of its own).  Enqueues on torch's current
    stream and waits for nothing: every size is known from the measuring pass."""
    b = ms.n if b is None else b
    n = b - a
    dev = ms.dev
    if n <= 0:
        return torch.zeros((0, TARGET_HEIGHT), dtype=torch.float32, device=dev), np.zeros(0, np.int64)
    wo, T = ms.wo[a:b], ms.T[a:b]
    row_off = np.zeros(n + 1, np.int64); np.cumsum(T, out=row_off[1:])
    row_start = row_off[:-1].copy() if layout is None else np.ascontiguousarray(layout(T), dtype=np.int64)
    tmp_off = np.zeros(n + 1, np.int64); np.cumsum(wo * TARGET_HEIGHT, out=tmp_off[1:])
    tmp = torch.empty(max(int(tmp_off[-1]), 1), dtype=torch.float32, device=dev)
    omax = torch.empty(n, dtype=torch.int32, device=dev)
    x = torch.empty((int(row_off[-1]), TARGET_HEIGHT), dtype=torch.float32, device=dev)
    d_tmp_off, d_row_off = _native.upload_packed([tmp_off[:-1].copy(), row_start], dev)     # named: they must outlive the launch
    stream = torch.cuda.current_stream(dev).cuda_stream
    # the per-strip arrays of the measured batch, from strip `a` on (their offsets into pix / center are absolute)
    _native.check(_native.lib.ta_linenorm_resample(
        ms.d_pix.data_ptr(), ms.d_pix_off.data_ptr() + 8 * a, ms.d_hh.data_ptr() + 4 * a, ms.d_ww.data_ptr() + 4 * a, n,
        ms.center.data_ptr(), ms.d_col_off.data_ptr() + 8 * a, ms.minmax.data_ptr() + 8 * a, ms.r.data_ptr() + 4 * a,
        ms.wout.data_ptr() + 4 * a, tmp.data_ptr(), d_tmp_off.data_ptr(), omax.data_ptr(), x.data_ptr(),
        d_row_off.data_ptr(), stream), "ta_linenorm_resample")
    # The measuring pass may have run on ANOTHER stream than this one (process_batch: a chunk's first stage measures on
    # the caller's stream, its launch resamples on the chunk's compute stream): what the resampling reads must not go back
    # to the other stream's allocator -- and from there into the next chunk's measuring pass -- before this stream has
    # read it, however early the last reference to `ms` is dropped.
    here = torch.cuda.current_stream(dev)
    for t in (ms.d_pix, ms.d_pix_off, ms.d_hh, ms.d_ww, ms.center, ms.d_col_off, ms.minmax, ms.r, ms.wout):
        t.record_stream(here)
    return x, T


def normalize_strips(strips, device="cuda", want_debug=False, layout=None):
    """strips: list of 2-D uint8 images (white background), each a host array or a tensor already on
    the device.  Returns (x, T, debug): x = float32
    device tensor [sum T, 48] (line b owns rows sum(T[:b]) .. + T[b]; or, with `layout`, rows
    layout(T)[b] .. + T[b] -- a permutation of those ranges, for a caller that wants the lines in an order of
    its own: the output widths are only known in here), T = int64 array of timesteps (normalised width + 32).
    measure_strips + resample_strips over the whole list."""
    dev = torch.device(device)
    n = len(strips)
    if n == 0:
        return torch.zeros((0, TARGET_HEIGHT), dtype=torch.float32, device=dev), np.zeros(0, np.int64), {}
    ms = measure_strips(strips, dev)
    x, T = resample_strips(ms, 0, n, layout)
    debug = {}
    if want_debug:
        c = ms.center.cpu().numpy()
        col_off = ms.col_off
        debug = {"center": [c[col_off[k]:col_off[k + 1]] for k in range(n)], "r": ms.r.cpu().numpy(),
                 "arg": [ms.arg.cpu().numpy()[col_off[k]:col_off[k + 1]] for k in range(n)]}
    return x, T, debug
