"""Device side of the page preprocessing (csrc/ta_preproc.hip): the image operations of
`textAlignPreprocessing.preprocess_images` / `identify_text_lines` (the Gamera-free restatement of
reference textAlignPreprocessing.py:160-285) on the GPU.  Control flow, the projection / peak
numerics and the selection of components are Python, as in the reference; every full-page pass is a
kernel: Otsu histogram, thresholding, connected components (despeckle, hole filling, tall-component
removal, line components), the skew search, the rotation, the run filters and the row projection.
uint8 greyscale pages (textAlignPreprocessing.to_grey_u8 reduces anything else).  Checked against the
scipy restatement oracle/preproc_ref.py (tests/test_preproc_gpu.py).

Batched by STAGE: a page's preprocessing is ~50 small kernels with six data-dependent host decisions in
between (Otsu threshold, the two skew sweeps, peaks, component selection, strip sizes).  Each decision
costs a wait for the device, and one page at a time those waits -- not the kernels -- were what a page
cost.  The functions below take a LIST of pages and wait once per stage for all of them (`find_lines` of
one page is a batch of one): per-page results are unchanged.  Since round 6 a stage is ONE call into the
library for the whole batch (`ta_pp_*_batch`: host arrays of device pointers in, nothing waited for) and
the arithmetic between two stages runs in the library's host loops (`ta_host_*`, each pinned to the numpy
expression it replaces; the numpy forms stay here / in textAlignPreprocessing as cross-checks): a page
thread holds the interpreter lock for ~0.3 ms per page instead of ~1 ms, which is what lets several of
them feed one GPU (textAlignPreprocessing.find_lines_many).  STAGE_CLOCK / _mark: optional clocks at the
stage boundaries (tools/pages_img_stages.py).
"""
import ctypes

import numpy as np
import torch
from scipy import special

from . import _native
from . import page as page_mod
from . import textAlignPreprocessing as host      # parameters + the pinned projection / peak numerics


class DeviceBinImage(page_mod.Image):
    """A binarised page that lives on the device: `dim` / `ncols` / `nrows` as `process` and its
    callers read them; `.ink` (bool array, True = ink) is downloaded on first use."""

    def __init__(self, plane, dev=None):
        page_mod.Image.__init__(self, int(plane.shape[1]), int(plane.shape[0]))
        self.plane = plane
        self.dev = dev
        self._ink = None

    @property
    def ink(self):
        if self._ink is None:
            self._ink = self.plane.cpu().numpy().astype(bool)
        return self._ink


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


LABEL_FLAGS = 0          # flags of the two stage calls that label: 0 = components over runs; _native.TA_PP_LABEL_PIXELS = a label per
                         # pixel (rounds 3-5; the tests run both and compare)
STAGE_CLOCK = None       # tools/pages_img_stages.py puts a list here: (thread, checkpoint name, perf_counter) per checkpoint


def _mark(name):
    if STAGE_CLOCK is not None:
        import threading
        import time
        STAGE_CLOCK.append((threading.get_ident(), name, time.perf_counter()))


def _carve(dev, counts, dtype):
    """ONE device buffer with room for counts[k] elements of `dtype` per page, each page's piece 256-byte aligned:
    (buffer, element offsets).  A stage's planes of a whole batch are one allocation instead of one per page."""
    item = torch.empty(0, dtype=dtype).element_size()
    unit = 256 // item
    offs, total = [], 0
    for c in counts:
        offs.append(total)
        total += (int(c) + unit - 1) // unit * unit
    return torch.empty(max(total, 1), dtype=dtype, device=dev), offs


def _addr(buf, offs):
    """device addresses of the pieces of a carved buffer, as the library's stage functions take them (a host array)"""
    base, item = buf.data_ptr(), buf.element_size()
    return np.array([base + int(o) * item for o in offs], dtype=np.uint64)


def _i32(values):
    return np.array([int(v) for v in values], dtype=np.int32)


class _Dev(object):
    def __init__(self, device="cuda"):
        self.dev = torch.device(device)
        if self.dev.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("text_alignment_amd needs an AMD GPU (MI355X): page preprocessing runs as HIP "
                               "kernels and there is no CPU fallback")
        self.lib = _native.lib
        self.flag = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.count = torch.zeros(1, dtype=torch.int32, device=self.dev)
        # the stream every launch of this handle goes to: torch's current one WHERE THE HANDLE IS MADE (a
        # handle lives for one batch, inside the caller's stream context; asking torch per launch was 0.15 ms
        # per page)
        self.stream = torch.cuda.current_stream(self.dev).cuda_stream
        self.row_sums = None         # preprocess_images_batch leaves the eroded planes' row projections here (buffer, offsets)

    # ---- one image (kept for tests and tools) ------------------------------------------------
    def label(self, ink):
        """(lab, stats) of a uint8 ink plane: labels and the five per-root statistics planes"""
        return self.label_many([ink])[0]

    def components(self, lab, stats, cap=1 << 12):
        """host array [ncomp][6] = {root, area, x0, y0, x1, y1}, sorted by root (raster order)"""
        return self.components_many([(lab, stats)], cap)[0]

    # ---- many images, one wait per stage -----------------------------------------------------
    def label_many(self, inks):
        """[(lab, stats)] of uint8 ink planes; the labelling rounds of all images share their waits"""
        n = len(inks)
        if n == 0:
            return []
        labs = [torch.empty(k.shape, dtype=torch.int32, device=self.dev) for k in inks]
        stats = [torch.empty((5,) + tuple(k.shape), dtype=torch.int32, device=self.dev) for k in inks]
        flags = torch.zeros(n, dtype=torch.int32, device=self.dev)
        hh = (ctypes.c_int32 * n)(*[int(k.shape[0]) for k in inks])
        ww = (ctypes.c_int32 * n)(*[int(k.shape[1]) for k in inks])
        _native.check(self.lib.ta_pp_label_batch(n, _ptr_array(inks), hh, ww, _ptr_array(labs), _ptr_array(stats),
                                                 flags.data_ptr(), self.stream), "ta_pp_label_batch")
        return list(zip(labs, stats))

    def components_many(self, labelled, cap=1 << 12):
        """component tables of [(lab, stats)]: host arrays [ncomp][6] = {root, area, x0, y0, x1, y1} sorted
        by root (raster order); one download for all images (an image with more than `cap` components is
        collected again with room for them)"""
        n = len(labelled)
        if n == 0:
            return []
        table, recs, counts = self.component_buffer(n, cap)
        for k, (lab, stats) in enumerate(labelled):
            h, w = lab.shape
            _native.check(self.lib.ta_pp_components(lab.data_ptr(), stats.data_ptr(), h, w, recs[k].data_ptr(),
                                                    cap, counts[k:].data_ptr(), self.stream), "ta_pp_components")
        return self.component_tables(n, labelled.__getitem__, table, cap)

    def component_buffer(self, n, cap):
        """(table, recs, counts): ONE int32 device buffer holding the component records [n][cap][6] and, behind them,
        the n counts -- so that both come back in one download (a page's table is 96 KB; a second wait costs more)"""
        table = torch.empty(n * cap * 6 + n, dtype=torch.int32, device=self.dev)
        return table, table[:n * cap * 6].view(n, cap, 6), table[n * cap * 6:]

    def component_tables(self, n, labelled, table, cap):
        """the download half of components_many: a component_buffer filled on the device -> sorted host tables
        (labelled(k): the (lab, stats) planes of image k, asked for only when its table has to be collected again)"""
        host = table.cpu().numpy()
        cnt = host[n * cap * 6:]
        host_recs = host[:n * cap * 6].reshape(n, cap, 6)
        out = []
        for k in range(n):
            c = int(cnt[k])
            if c > cap:                                    # rare: a page with thousands of components
                big = torch.empty((c, 6), dtype=torch.int32, device=self.dev)
                lab, stats = labelled(k)
                h, w = lab.shape
                _native.check(self.lib.ta_pp_components(lab.data_ptr(), stats.data_ptr(), h, w, big.data_ptr(),
                                                        c, self.count.data_ptr(), self.stream), "ta_pp_components")
                r = big.cpu().numpy()
            else:
                r = host_recs[k, :c]
            out.append(r[np.argsort(r[:, 0], kind="stable")])
        return out


def otsu_from_histogram(hist):
    """Otsu's threshold from a 256-bin histogram (first maximum of the between-class variance)"""
    hist = hist.astype(np.float64)
    total = hist.sum()
    cum = np.cumsum(hist)
    mean_cum = np.cumsum(hist * np.arange(256))
    mean_all = mean_cum[-1]
    with np.errstate(divide='ignore', invalid='ignore'):
        between = (mean_all * cum - mean_cum * total) ** 2 / (cum * (total - cum))
    between[~np.isfinite(between)] = 0
    return int(np.argmax(between))


def otsu_thresholds(hists):
    """otsu_from_histogram of every row of an int32 [n][256] array, in the library's host loop (ta_host_otsu_batch)"""
    hists = np.ascontiguousarray(hists, dtype=np.int32).reshape(-1, 256)
    thr = np.zeros(len(hists), np.int32)
    _native.check(_native.lib.ta_host_otsu_batch(hists.ctypes.data, len(hists), thr.ctypes.data), "ta_host_otsu_batch")
    return thr


def rotation_angles_device(d, inks, lo=-6.0, hi=6.0, coarse=0.25, fine=0.05):
    """the skew angle of every page: the angle in [lo, hi] degrees whose rotation makes the row projection
    sharpest (largest variance), coarse sweep then a fine sweep around the best, with the per-angle
    row histograms built on the device from the ink coordinates (pixel (y, x) lands on row
    cy + (y - cy) cos a - (x - cx) sin a of the page decimated to <= 1200 rows / columns); one download
    per sweep for all pages.  A page without ink reports 0."""
    n = len(inks)
    steps, hs, ws = _decimation([k.shape for k in inks])
    # the ink pixels of every decimated page, listed once for all angles of both sweeps (ta_pp_ink_points)
    pts, offs = _carve(d.dev, [a * b for a, b in zip(hs, ws)], torch.int32)
    counts = torch.zeros(max(n, 1), dtype=torch.int32, device=d.dev)
    for k, ink in enumerate(inks):
        h, w = ink.shape
        _native.check(d.lib.ta_pp_ink_points(ink.data_ptr(), h, w, steps[k], pts[offs[k]:].data_ptr(), counts[k:].data_ptr(),
                                             d.stream), "ta_pp_ink_points")
    return _skew_search(d, _addr(pts, offs), counts, hs, ws, lo, hi, coarse, fine, keep=pts)


def _decimation(shapes):
    steps = [max(1, int(max(sh) / 1200)) for sh in shapes]
    hs = [(int(sh[0]) + st - 1) // st for sh, st in zip(shapes, steps)]
    ws = [(int(sh[1]) + st - 1) // st for sh, st in zip(shapes, steps)]
    return steps, hs, ws


def _skew_search(d, points, counts, hs, ws, lo=-6.0, hi=6.0, coarse=0.25, fine=0.05, keep=None):
    """rotation_angles_device from the pages' point lists (device addresses `points`, device counts)"""
    n = len(hs)
    d_hs, d_ws = _i32(hs), _i32(ws)

    def sharpest(grids):
        """per page the index of the angle whose row histogram has the largest variance, and whether the page has ink"""
        # the angles' cosines / sines of all pages in ONE asynchronous upload, the histograms of all pages in one buffer,
        # one stage call and one download; the variances and their argmax in the library's host loop (numpy's np.var to
        # the last bit -- _sharpest_rows_numpy is the expression it replaces -- without the interpreter lock)
        tables = []
        for g in grids:
            cs = np.empty(2 * len(g), np.float64)
            rad = np.deg2rad(g)
            cs[0::2], cs[1::2] = np.cos(rad), np.sin(rad)
            tables.append(cs)
        d_cs = _native.upload_packed(tables, d.dev)
        nang = _i32([len(g) for g in grids])
        sizes = [int(nang[k]) * hs[k] for k in range(n)]
        hist, hoffs = _carve(d.dev, sizes, torch.int32)
        cs_ptr = np.array([t.data_ptr() for t in d_cs], dtype=np.uint64)
        hist_ptr = _addr(hist, hoffs)                                            # (host arrays: alive until the call returns)
        _native.check(d.lib.ta_pp_angle_histograms_points_batch(
            n, points.ctypes.data, counts.data_ptr(), d_hs.ctypes.data, d_ws.ctypes.data, cs_ptr.ctypes.data,
            nang.ctypes.data, hist_ptr.ctypes.data, d.stream), "ta_pp_angle_histograms_points_batch")
        _mark("sweep enqueued")
        flat = hist.cpu().numpy()
        _mark("sweep back")
        return sharpest_rows(flat, hoffs, nang, d_hs)
    grid = np.arange(lo, hi + 1e-9, coarse)
    coarse_best, has_ink = sharpest([grid] * n)
    fine_grids = [np.arange(grid[int(b_)] - coarse, grid[int(b_)] + coarse + 1e-9, fine) for b_ in coarse_best]
    fine_best, _ = sharpest(fine_grids)
    return [float(np.round(g[int(b_)], 3)) if ink else 0.0 for ink, g, b_ in zip(has_ink, fine_grids, fine_best)]


def sharpest_rows(flat, offs, nang, hs):
    """(best, any) per page: best[k] = int(np.argmax(np.var(H_k, axis=1))) of page k's histograms H_k = nang[k] rows of
    hs[k] int32 counts at flat[offs[k]:], any[k] = H_k.any() -- ta_host_sharpest_rows"""
    n = len(offs)
    flat = np.ascontiguousarray(flat, dtype=np.int32)
    off = np.array([int(o) for o in offs], dtype=np.int64)
    nang, hs = np.ascontiguousarray(nang, dtype=np.int32), np.ascontiguousarray(hs, dtype=np.int32)
    if n and int((off + nang.astype(np.int64) * hs).max()) > flat.size:
        raise ValueError("a page's histograms lie outside the buffer")
    best, some = np.zeros(n, np.int32), np.zeros(n, np.uint8)
    _native.check(_native.lib.ta_host_sharpest_rows(flat.ctypes.data, off.ctypes.data, nang.ctypes.data, hs.ctypes.data, n,
                                                    best.ctypes.data, some.ctypes.data, None), "ta_host_sharpest_rows")
    return best, some.astype(bool)


def _sharpest_rows_numpy(flat, offs, nang, hs):
    """the numpy expressions sharpest_rows replaces (np.var per row, bit for bit): the cross-check of the tests"""
    best, some = [], []
    for o, a, h in zip(offs, nang, hs):
        H = np.asarray(flat[int(o):int(o) + int(a) * int(h)]).reshape(int(a), int(h))
        best.append(int(np.argmax(np.var(H, axis=1))))
        some.append(bool(H.any()))
    return np.array(best, np.int32), np.array(some, bool)


def _rotation_geometry(h, w, angle):
    """scipy.ndimage.rotate's own geometry (ndimage/_interpolation.py rotate, reshape=True): the output shape and the
    six numbers [matrix, offset] of the backward mapping"""
    c, s = special.cosdg(angle), special.sindg(angle)
    rot = np.array([[c, s], [-s, c]])
    out_bounds = rot @ [[0, 0, h, h], [0, w, 0, w]]
    out_shape = (np.ptp(out_bounds, axis=1) + 0.5).astype(int)
    out_center = rot @ ((out_shape - 1) / 2)
    in_center = (np.asarray([h, w]) - 1) / 2
    offset = in_center - out_center
    return (int(out_shape[0]), int(out_shape[1]),
            np.array([rot[0, 0], rot[0, 1], rot[1, 0], rot[1, 1], offset[0], offset[1]], np.float64))


def _upload_pages(d, pages_px):
    """the pages of a batch on the device.  Pageable numpy pages are gathered into ONE page-locked buffer by the
    library's host copy loop (no interpreter lock held) and sent in one asynchronous transfer, the planes being views of
    one device buffer; a page that is a torch tensor is taken where it lies -- on the device as it is, from page-locked
    host memory by an asynchronous transfer of its own (no staging copy)."""
    out = [None] * len(pages_px)
    host_ix = []
    for k, px in enumerate(pages_px):
        if isinstance(px, torch.Tensor):
            if px.device.type == "cuda":
                if px.device.index != (torch.cuda.current_device() if d.dev.index is None else d.dev.index):
                    raise ValueError("a device page lies on another GPU than the one that preprocesses it")
                out[k] = px
            else:
                out[k] = px.to(d.dev, non_blocking=px.is_pinned())
        else:
            host_ix.append(k)
    if host_ix:
        host_px = [pages_px[k] for k in host_ix]
        sizes = [int(px.size) for px in host_px]
        offs = np.concatenate(([0], np.cumsum([(sz + 255) // 256 * 256 for sz in sizes]))).astype(np.int64)
        stage = torch.empty(max(int(offs[-1]), 1), dtype=torch.uint8, pin_memory=True)
        _native.host_copy_pieces(stage.numpy(), host_px, offs[:-1])
        dev = stage.to(d.dev, non_blocking=True)
        for k, o, sz, px in zip(host_ix, offs[:-1], sizes, host_px):
            out[k] = dev[int(o):int(o) + sz].view(px.shape)
    return out


def _page_plane(pg):
    """a page's uint8 greyscale plane: a C-contiguous numpy array, or a contiguous torch tensor (device or host)"""
    px = getattr(pg, "pixels", pg)
    if isinstance(px, torch.Tensor):
        if px.dtype != torch.uint8 or px.dim() != 2:
            raise TypeError("the device preprocessing takes 2-D uint8 pages")
        return px.contiguous()
    px = np.asarray(px)
    if px.dtype != np.uint8 or px.ndim != 2:
        raise TypeError("the device preprocessing takes 2-D uint8 pages")
    return np.ascontiguousarray(px)


def preprocess_images_batch(pages, despeckle_amt=host.despeckle_amt, filter_runs=1, filter_runs_amt=2,
                            correct_rotation=True, device="cuda"):
    """[(ink, eroded, angle)] as uint8 device planes + the device handle: reference
    textAlignPreprocessing.py:160-195 for a list of uint8 greyscale pages"""
    d = _Dev(device)
    host_px = [_page_plane(pg) for pg in pages]
    n = len(host_px)
    d.row_sums = None
    if n == 0:
        return d, []
    lib, dev, st = d.lib, d.dev, d.stream
    _mark("start")
    imgs = _upload_pages(d, host_px)
    _mark("pages staged")
    hh, ww = _i32([px.shape[0] for px in host_px]), _i32([px.shape[1] for px in host_px])
    npix = hh.astype(np.int64) * ww
    img_ptr = np.array([t.data_ptr() for t in imgs], dtype=np.uint64)
    # stage 1: grey-level histograms -> Otsu thresholds (host, as in the reference)
    hist = torch.empty((n, 256), dtype=torch.int32, device=dev)
    _native.check(lib.ta_pp_histogram_batch(n, img_ptr.ctypes.data, npix.ctypes.data, hist.data_ptr(), st), "ta_pp_histogram_batch")
    _mark("histograms enqueued")
    hist_host = hist.cpu().numpy()
    _mark("histograms back")
    thrs = otsu_thresholds(hist_host)
    # stage 2: threshold, despeckle ink and background, drop tall components, list the decimated pages' ink points
    inks, ink_off = _carve(dev, npix, torch.uint8)
    lab, lab_off = _carve(dev, npix, torch.int32)
    stats, stats_off = _carve(dev, 5 * npix, torch.int32)
    steps, hs, ws = _decimation([px.shape for px in host_px])
    pts, pts_off = _carve(dev, [a_ * b_ for a_, b_ in zip(hs, ws)], torch.int32)
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    ink_ptr, lab_ptr, stats_ptr, pts_ptr = _addr(inks, ink_off), _addr(lab, lab_off), _addr(stats, stats_off), _addr(pts, pts_off)
    d_steps = _i32(steps)
    _native.check(lib.ta_pp_binarise_batch(n, img_ptr.ctypes.data, hh.ctypes.data, ww.ctypes.data, thrs.ctypes.data,
                                           int(despeckle_amt), int(host.sat_area_thresh), ink_ptr.ctypes.data,
                                           lab_ptr.ctypes.data, stats_ptr.ctypes.data, d_steps.ctypes.data,
                                           pts_ptr.ctypes.data, counts.data_ptr(), LABEL_FLAGS, st), "ta_pp_binarise_batch")
    _mark("binarise enqueued")
    # stage 3: the skew search (two sweeps, a download each; variances and the choice on the host)
    skews = _skew_search(d, pts_ptr, counts, hs, ws, -6, 6)
    _mark("skews chosen")
    # stage 4: rotation, run filters, row projection
    geo = [_rotation_geometry(int(h), int(w), a_) if (correct_rotation and a_ != 0) else None
           for h, w, a_ in zip(hh, ww, skews)]
    oh = _i32([g[0] if g else h for g, h in zip(geo, hh)])
    ow = _i32([g[1] if g else w for g, w in zip(geo, ww)])
    turned = [g[2] for g in geo if g]
    maps = iter(_native.upload_packed(turned, dev)) if turned else iter(())
    keep_maps = [next(maps) if g else None for g in geo]
    mo_ptr = np.array([t.data_ptr() if t is not None else 0 for t in keep_maps], dtype=np.uint64)
    onp = oh.astype(np.int64) * ow
    outs, out_off = _carve(dev, onp, torch.uint8)
    tmp, tmp_off = _carve(dev, onp, torch.uint8)
    eroded, er_off = _carve(dev, onp, torch.uint8)
    sums, sums_off = _carve(dev, oh, torch.int32)
    out_ptr, tmp_ptr, er_ptr, sums_ptr = _addr(outs, out_off), _addr(tmp, tmp_off), _addr(eroded, er_off), _addr(sums, sums_off)
    _native.check(lib.ta_pp_deskew_batch(n, ink_ptr.ctypes.data, hh.ctypes.data, ww.ctypes.data, mo_ptr.ctypes.data,
                                         out_ptr.ctypes.data, oh.ctypes.data, ow.ctypes.data, tmp_ptr.ctypes.data,
                                         er_ptr.ctypes.data, int(filter_runs_amt), int(filter_runs), sums_ptr.ctypes.data, st),
                  "ta_pp_deskew_batch")
    _mark("deskew enqueued")
    d.row_sums = (sums, sums_off)                                 # identify_text_lines_batch takes them from here
    out = []
    for k in range(n):
        m = int(onp[k])
        shape = (int(oh[k]), int(ow[k]))
        out.append((outs[out_off[k]:out_off[k] + m].view(shape), eroded[er_off[k]:er_off[k] + m].view(shape),
                    host.reported_angle(skews[k])))              # sign: see host.reported_angle
    return d, out


def preprocess_images(input_image, despeckle_amt=host.despeckle_amt, filter_runs=1, filter_runs_amt=2,
                      correct_rotation=True, device="cuda"):
    """(d, ink, eroded, angle) of one uint8 greyscale page"""
    d, out = preprocess_images_batch([input_image], despeckle_amt, filter_runs, filter_runs_amt,
                                     correct_rotation, device)
    return (d,) + out[0]


def identify_text_lines_batch(d, planes, row_sums=None):
    """text lines of preprocessed pages (reference textAlignPreprocessing.py:198-285) from their
    (ink, eroded) device planes: [(line strips, peak locations, smoothed projection)].  row_sums: the carved buffer
    (tensor, offsets) of the eroded planes' row projections when the deskew stage has made them already."""
    n = len(planes)
    if n == 0:
        return []
    lib, dev, st = d.lib, d.dev, d.stream
    hh, ww = _i32([e.shape[0] for _, e in planes]), _i32([e.shape[1] for _, e in planes])
    npix = hh.astype(np.int64) * ww
    er_ptr = np.array([e.data_ptr() for _, e in planes], dtype=np.uint64)
    ink_ptr = np.array([i.data_ptr() for i, _ in planes], dtype=np.uint64)
    if row_sums is None:
        sums, sums_off = _carve(dev, hh, torch.int32)
        for k, (ink, eroded) in enumerate(planes):
            _native.check(lib.ta_pp_row_sums(eroded.data_ptr(), int(hh[k]), int(ww[k]), sums[sums_off[k]:].data_ptr(), st),
                          "ta_pp_row_sums")
    else:
        sums, sums_off = row_sums
    _mark("lines: start")
    flat = sums.cpu().numpy().astype(np.int64)
    _mark("projections back")
    # peaks of the smoothed projections and the white lines between neighbouring text lines: host, as in the reference
    found = host.peaks_of_projections(flat, sums_off, hh)
    smoothed_all, peaks_all, row_lists = [f[0] for f in found], [f[1] for f in found], [f[2] for f in found]
    _mark("peaks found")
    # stage 5: clear those rows in a copy of the eroded plane, label it, collect the component tables
    d_rows = _native.upload_packed(row_lists, dev)
    rows_ptr = np.array([t.data_ptr() if len(r) else 0 for t, r in zip(d_rows, row_lists)], dtype=np.uint64)
    nrows = _i32([len(r) for r in row_lists])
    work, work_off = _carve(dev, npix, torch.uint8)
    lab, lab_off = _carve(dev, npix, torch.int32)
    stats, stats_off = _carve(dev, 5 * npix, torch.int32)
    cap = 1 << 12
    table, recs, counts = d.component_buffer(n, cap)
    work_ptr, lab_ptr, stats_ptr = _addr(work, work_off), _addr(lab, lab_off), _addr(stats, stats_off)
    _native.check(lib.ta_pp_line_components_batch(n, er_ptr.ctypes.data, hh.ctypes.data, ww.ctypes.data, rows_ptr.ctypes.data,
                                                  nrows.ctypes.data, work_ptr.ctypes.data, lab_ptr.ctypes.data,
                                                  stats_ptr.ctypes.data, recs.data_ptr(), cap, counts.data_ptr(), LABEL_FLAGS, st),
                  "ta_pp_line_components_batch")
    def labelled(k):
        # (a page with more components than the table holds: labelled again per pixel, whose planes ta_pp_components reads)
        return d.label(work[work_off[k]:work_off[k] + int(npix[k])].view(int(hh[k]), int(ww[k])))
    _mark("components enqueued")
    recs_all = d.component_tables(n, labelled, table, cap)
    _mark("components back")
    boxes_all, total = [], 0
    for peaks, recs_k in zip(peaks_all, recs_all):
        boxes = []
        big = recs_k[recs_k[:, 1] > host.noise_area_thresh]
        if len(big):
            comps = big[:, 2:6].astype(np.int64)                  # ulx, uly, lrx, lry
            heights = comps[:, 3] - comps[:, 1] + 1
            med = np.median(heights)
            comps = comps[heights < med * host.remove_capitals_scale]
            cc_median_height = np.median(comps[:, 3] - comps[:, 1] + 1)
            for ulx, uly, lrx, lry in host.line_boxes(peaks, comps, cc_median_height):
                boxes.append((ulx, uly, lrx, lry, total))
                total += (lry - uly + 1) * (lrx - ulx + 1)
        boxes_all.append(boxes)
    _mark("boxes chosen")
    # stage 6: cut the strips on the device (ink black on white, as the reference saves them) into one packed buffer
    # and leave them there: the recogniser's normaliser reads them where they are, `strip.pixels` downloads
    packed = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
    d_boxes = _native.upload_packed([np.array(boxes, dtype=np.int64).reshape(-1, 5) for boxes in boxes_all], dev)
    box_ptr = np.array([t.data_ptr() if boxes else 0 for t, boxes in zip(d_boxes, boxes_all)], dtype=np.uint64)
    nboxes = _i32([len(boxes) for boxes in boxes_all])
    _native.check(lib.ta_pp_cut_strips_batch(n, ink_ptr.ctypes.data, hh.ctypes.data, ww.ctypes.data, box_ptr.ctypes.data,
                                             nboxes.ctypes.data, packed.data_ptr(), st), "ta_pp_cut_strips_batch")
    out = []
    for boxes, peaks, smoothed in zip(boxes_all, peaks_all, smoothed_all):
        strips = []
        for ulx, uly, lrx, lry, off in boxes:
            hh_, ww_ = lry - uly + 1, lrx - ulx + 1
            strips.append(page_mod.Strip(ulx, uly, hh_, width=ww_, device_pixels=page_mod.DeviceStrip(packed, off, hh_, ww_)))
        out.append((strips, peaks, smoothed))
    _mark("strips made")
    return out


def identify_text_lines(image_bin, image_eroded):
    """(line strips, peak locations, smoothed projection) of one preprocessed page"""
    d = _Dev(image_bin.plane.device)          # a handle of this call: it launches on the stream current HERE
    return identify_text_lines_batch(d, [(image_bin.plane, image_eroded.plane)])[0]


def find_lines_batch(pages, device="cuda"):
    """[(image_bin, image_eroded, angle, strips, peak locations)] of uint8 greyscale pages"""
    if not pages:
        return []
    d, pre = preprocess_images_batch(pages, device=device)
    lines = identify_text_lines_batch(d, [(ink, eroded) for ink, eroded, _ in pre], row_sums=d.row_sums)
    return [(DeviceBinImage(ink, d), DeviceBinImage(eroded, d), angle, strips, peaks)
            for (ink, eroded, angle), (strips, peaks, _) in zip(pre, lines)]


def find_lines(input_image, device="cuda"):
    """(image_bin, image_eroded, angle, strips, peak locations) of one uint8 greyscale page"""
    return find_lines_batch([input_image], device=device)[0]
