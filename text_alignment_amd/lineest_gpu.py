"""Device side of the line normaliser (csrc/ta_lineest.hip): raw greyscale strips in, the
recogniser's input rows out, without a host round trip: ocropy 1.3.3's CenterNormalizer +
prepare_line (parity unpinned).  The GPU tests check it against the
float64 scipy restatement oracle/lineest_ref.py.  Strips are 2-D uint8 greyscale images, as the PNG
files the reference hands to ocropus-rpred (alignToOCR.py:131-132); there is no host path.
"""
import numpy as np
import torch

from . import _native

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

        def copy(span):
            for k in range(*span):
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


def normalize_strips(strips, device="cuda", want_debug=False, layout=None):
    """strips: list of 2-D uint8 images (white background), each a host array or a tensor already on
    the device.  Returns (x, T, debug): x = float32
    device tensor [sum T, 48] (line b owns rows sum(T[:b]) .. + T[b]; or, with `layout`, rows
    layout(T)[b] .. + T[b] -- a permutation of those ranges, for a caller that wants the lines in an order of
    its own: the output widths are only known in here), T = int64 array of timesteps (normalised width + 32)."""
    dev = torch.device(device)
    lib = _native.lib
    n = len(strips)
    if n == 0:
        return torch.zeros((0, TARGET_HEIGHT), dtype=torch.float32, device=dev), np.zeros(0, np.int64), {}
    hh = np.zeros(n, np.int32); ww = np.zeros(n, np.int32)
    on_device = [isinstance(s, torch.Tensor) for s in strips]
    for k, s in enumerate(strips):
        if on_device[k]:
            if (s.dim() != 2 or s.dtype != torch.uint8 or not s.is_contiguous() or s.device.type != "cuda" or
                    s.device.index != (torch.cuda.current_device() if dev.index is None else dev.index)):
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
        d_pix = torch.cat([s.reshape(-1) if on_device[k] else up(np.asarray(s).ravel()) for k, s in enumerate(strips)])
    d_pix_off, d_hh, d_ww = up(pix_off[:-1].copy()), up(hh), up(ww)
    d_gw, d_gw_off, d_gr, d_col_off = up(gw), up(gw_off), up(gr), up(col_off[:-1].copy())
    d_ws_off = up(3 * pix_off[:-1])
    ws = torch.empty(3 * int(pix_off[-1]), dtype=torch.float64, device=dev)
    arg = torch.empty(int(col_off[-1]), dtype=torch.int32, device=dev)
    center = torch.empty_like(arg)
    minmax = torch.empty(2 * n, dtype=torch.int32, device=dev)
    r = torch.empty(n, dtype=torch.int32, device=dev)
    wout = torch.empty(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _native.check(lib.ta_linenorm_measure(
        d_pix.data_ptr(), d_pix_off.data_ptr(), d_hh.data_ptr(), d_ww.data_ptr(), n,
        d_gw.data_ptr(), d_gw_off.data_ptr(), d_gr.data_ptr(), ws.data_ptr(), d_ws_off.data_ptr(),
        arg.data_ptr(), center.data_ptr(), d_col_off.data_ptr(), minmax.data_ptr(),
        r.data_ptr(), wout.data_ptr(), stream), "ta_linenorm_measure")
    sized = torch.cat([wout, minmax]).cpu().numpy()       # output sizes are data-dependent: one small sync
    wo = sized[:n].astype(np.int64)
    if bool((sized[n::2] == sized[n + 1::2]).any()):      # the measuring pass found a strip's minimum = its maximum
        raise ValueError("empty or constant text-line image")
    del ws
    T = wo + 2 * PAD
    row_off = np.zeros(n + 1, np.int64); np.cumsum(T, out=row_off[1:])
    row_start = row_off[:-1].copy() if layout is None else np.ascontiguousarray(layout(T), dtype=np.int64)
    tmp_off = np.zeros(n + 1, np.int64); np.cumsum(wo * TARGET_HEIGHT, out=tmp_off[1:])
    tmp = torch.empty(max(int(tmp_off[-1]), 1), dtype=torch.float32, device=dev)
    omax = torch.empty(n, dtype=torch.int32, device=dev)
    x = torch.empty((int(row_off[-1]), TARGET_HEIGHT), dtype=torch.float32, device=dev)
    d_tmp_off, d_row_off = up(tmp_off[:-1].copy()), up(row_start)     # named: they must outlive the launch
    _native.check(lib.ta_linenorm_resample(
        d_pix.data_ptr(), d_pix_off.data_ptr(), d_hh.data_ptr(), d_ww.data_ptr(), n,
        center.data_ptr(), d_col_off.data_ptr(), minmax.data_ptr(), r.data_ptr(), wout.data_ptr(),
        tmp.data_ptr(), d_tmp_off.data_ptr(), omax.data_ptr(), x.data_ptr(),
        d_row_off.data_ptr(), stream), "ta_linenorm_resample")
    debug = {}
    if want_debug:
        c = center.cpu().numpy()
        debug = {"center": [c[col_off[k]:col_off[k + 1]] for k in range(n)], "r": r.cpu().numpy(),
                 "arg": [arg.cpu().numpy()[col_off[k]:col_off[k + 1]] for k in range(n)]}
    return x, T, debug
