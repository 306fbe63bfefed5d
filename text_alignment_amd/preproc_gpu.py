"""Device side of the page preprocessing (csrc/ta_preproc.hip): the image operations of
`textAlignPreprocessing.preprocess_images` / `identify_text_lines` (the Gamera-free restatement of
reference textAlignPreprocessing.py:160-285) on the GPU.  Control flow, the projection / peak
numerics and the selection of components are Python, as in the reference; every full-page pass is a
kernel: Otsu histogram, thresholding, connected components (despeckle, hole filling, tall-component
removal, line components), the skew search, the rotation, the run filters and the row projection.
uint8 greyscale pages (textAlignPreprocessing.to_grey_u8 reduces anything else).  Checked against the
scipy restatement oracle/preproc_ref.py (tests/test_preproc_gpu.py).

Batched by STAGE: a page's preprocessing is ~40 small kernels with a dozen data-dependent host
decisions in between (Otsu threshold, labelling rounds, the two skew sweeps, peaks, component
selection, strip sizes).  Each decision costs a wait for the device, and one page at a time those
waits -- not the kernels -- were what a page cost.  The functions below take a LIST of pages and wait
once per stage for all of them (`find_lines` of one page is a batch of one): per-page results are
unchanged, the waits per page fall from ~16 to ~16 / batch.
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

    # ---- one image (kept for tests and tools) ------------------------------------------------
    def label(self, ink):
        """(lab, stats) of a uint8 ink plane: labels and the five per-root statistics planes"""
        return self.label_many([ink])[0]

    def filter(self, ink, lab, stats, min_area=0, max_height=2 ** 30):
        h, w = ink.shape
        _native.check(self.lib.ta_pp_filter_components(ink.data_ptr(), lab.data_ptr(), stats.data_ptr(), h, w,
                                                       int(min_area), int(max_height), self.stream),
                      "ta_pp_filter_components")

    def components(self, lab, stats, cap=1 << 12):
        """host array [ncomp][6] = {root, area, x0, y0, x1, y1}, sorted by root (raster order)"""
        return self.components_many([(lab, stats)], cap)[0]

    def despeckle(self, ink, size):
        self.despeckle_many([ink], size)

    def invert(self, ink):
        _native.check(self.lib.ta_pp_invert(ink.data_ptr(), ink.numel(), self.stream), "ta_pp_invert")

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

    def despeckle_many(self, inks, size):
        for ink, (lab, stats) in zip(inks, self.label_many(inks)):
            self.filter(ink, lab, stats, min_area=size)

    def components_many(self, labelled, cap=1 << 12):
        """component tables of [(lab, stats)]: host arrays [ncomp][6] = {root, area, x0, y0, x1, y1} sorted
        by root (raster order); one download for all images (an image with more than `cap` components is
        collected again with room for them)"""
        n = len(labelled)
        if n == 0:
            return []
        counts = torch.zeros(n, dtype=torch.int32, device=self.dev)
        recs = torch.empty((n, cap, 6), dtype=torch.int32, device=self.dev)
        for k, (lab, stats) in enumerate(labelled):
            h, w = lab.shape
            _native.check(self.lib.ta_pp_components(lab.data_ptr(), stats.data_ptr(), h, w, recs[k].data_ptr(),
                                                    cap, counts[k:].data_ptr(), self.stream), "ta_pp_components")
        cnt = counts.cpu().numpy()
        small = int(min(cap, max(int(cnt.max()), 1)))
        host_recs = recs[:, :small].cpu().numpy()
        out = []
        for k, (lab, stats) in enumerate(labelled):
            c = int(cnt[k])
            if c > cap:                                    # rare: a page with thousands of components
                big = torch.empty((c, 6), dtype=torch.int32, device=self.dev)
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


def otsu_threshold_device(d, img):
    return otsu_thresholds_device(d, [img])[0]


def otsu_thresholds_device(d, imgs):
    hist = torch.empty((len(imgs), 256), dtype=torch.int32, device=d.dev)
    for k, img in enumerate(imgs):
        _native.check(d.lib.ta_pp_histogram(img.data_ptr(), img.numel(), hist[k].data_ptr(), d.stream), "ta_pp_histogram")
    hh = hist.cpu().numpy()
    return [otsu_from_histogram(hh[k]) for k in range(len(imgs))]


def rotation_angles_device(d, inks, lo=-6.0, hi=6.0, coarse=0.25, fine=0.05):
    """the skew angle of every page: the angle in [lo, hi] degrees whose rotation makes the row projection
    sharpest (largest variance), coarse sweep then a fine sweep around the best, with the per-angle
    row histograms built on the device from the ink coordinates (pixel (y, x) lands on row
    cy + (y - cy) cos a - (x - cx) sin a of the page decimated to <= 1200 rows / columns); one download
    per sweep for all pages.  A page without ink reports 0."""
    n = len(inks)
    steps = [max(1, int(max(k.shape) / 1200)) for k in inks]
    hs = [(int(k.shape[0]) + st - 1) // st for k, st in zip(inks, steps)]
    ws = [(int(k.shape[1]) + st - 1) // st for k, st in zip(inks, steps)]
    # the ink pixels of every decimated page, listed once for all angles of both sweeps (ta_pp_ink_points)
    points, counts = [], torch.zeros(max(n, 1), dtype=torch.int32, device=d.dev)
    for k, ink in enumerate(inks):
        pts = torch.empty(max(hs[k] * ws[k], 1), dtype=torch.int32, device=d.dev)
        h, w = ink.shape
        _native.check(d.lib.ta_pp_ink_points(ink.data_ptr(), h, w, steps[k], pts.data_ptr(), counts[k:].data_ptr(),
                                             d.stream), "ta_pp_ink_points")
        points.append(pts)

    def sweep(grids):
        # the angles' cosines / sines of all pages in ONE asynchronous upload, the histograms of all pages in one buffer
        # and one download (a small `.to(device)` from pageable memory waits for everything the stream holds: per page
        # it kept the host from ever running ahead of the device)
        tables = []
        for g in grids:
            cs = np.empty(2 * len(g), np.float64)
            rad = np.deg2rad(g)
            cs[0::2], cs[1::2] = np.cos(rad), np.sin(rad)
            tables.append(cs)
        d_cs = _native.upload_packed(tables, d.dev)
        sizes = [len(grids[k]) * hs[k] for k in range(n)]
        hist = torch.empty(max(sum(sizes), 1), dtype=torch.int32, device=d.dev)
        pos = 0
        for k in range(n):
            _native.check(d.lib.ta_pp_angle_histograms_points(points[k].data_ptr(), counts[k:].data_ptr(), hs[k], ws[k],
                                                              d_cs[k].data_ptr(), len(grids[k]), hist[pos:].data_ptr(),
                                                              d.stream), "ta_pp_angle_histograms_points")
            pos += sizes[k]
        flat = hist.cpu().numpy()
        out, pos = [], 0
        for k in range(n):
            out.append(flat[pos:pos + sizes[k]].reshape(len(grids[k]), hs[k]))
            pos += sizes[k]
        return out
    grid = np.arange(lo, hi + 1e-9, coarse)
    hh = sweep([grid] * n)
    empty = [not bool(h.any()) for h in hh]
    best = [grid[int(np.argmax(np.var(h, axis=1)))] for h in hh]          # np.var per row, bit for bit
    fine_grids = [np.arange(b - coarse, b + coarse + 1e-9, fine) for b in best]
    hh = sweep(fine_grids)
    return [0.0 if e else float(np.round(g[int(np.argmax(np.var(h, axis=1)))], 3))
            for e, g, h in zip(empty, fine_grids, hh)]


def rotation_angle_device(d, ink, lo=-6.0, hi=6.0, coarse=0.25, fine=0.05):
    return rotation_angles_device(d, [ink], lo, hi, coarse, fine)[0]


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


def rotate_many(d, inks, angles):
    """scipy.ndimage.rotate(float32(ink), angle, reshape=True, order=1) > 0.5 of every page, the resampling on the
    device; the pages' mappings go up in one asynchronous transfer"""
    geo = [None if a == 0 else _rotation_geometry(int(k.shape[0]), int(k.shape[1]), a) for k, a in zip(inks, angles)]
    turned = [g for g in geo if g is not None]
    maps = iter(_native.upload_packed([g[2] for g in turned], d.dev)) if turned else iter(())
    out = []
    for ink, g in zip(inks, geo):
        if g is None:
            out.append(ink.clone())
            continue
        h, w = ink.shape
        oh, ow, _ = g
        res = torch.empty((oh, ow), dtype=torch.uint8, device=d.dev)
        _native.check(d.lib.ta_pp_rotate(ink.data_ptr(), h, w, res.data_ptr(), oh, ow, next(maps).data_ptr(), d.stream),
                      "ta_pp_rotate")
        out.append(res)
    return out


def rotate_device(d, ink, angle):
    return rotate_many(d, [ink], [angle])[0]


def open_runs_device(d, ink, length, axis):
    if length <= 1:
        return ink
    h, w = ink.shape
    out = torch.empty_like(ink)
    _native.check(d.lib.ta_pp_open_runs(ink.data_ptr(), out.data_ptr(), h, w, int(length), int(axis), d.stream),
                  "ta_pp_open_runs")
    return out


def _upload_pages(d, host_px):
    """the pages of a batch on the device: gathered into ONE page-locked buffer by the library's host copy loop (no
    interpreter lock held) and sent in one asynchronous transfer; the planes are views of one device buffer"""
    sizes = [int(px.size) for px in host_px]
    offs = np.concatenate(([0], np.cumsum([(sz + 255) // 256 * 256 for sz in sizes]))).astype(np.int64)
    stage = torch.empty(max(int(offs[-1]), 1), dtype=torch.uint8, pin_memory=True)
    _native.host_copy_pieces(stage.numpy(), host_px, offs[:-1])
    dev = stage.to(d.dev, non_blocking=True)
    return [dev[int(o):int(o) + sz].view(px.shape) for o, sz, px in zip(offs[:-1], sizes, host_px)]


def preprocess_images_batch(pages, despeckle_amt=host.despeckle_amt, filter_runs=1, filter_runs_amt=2,
                            correct_rotation=True, device="cuda"):
    """[(ink, eroded, angle)] as uint8 device planes + the device handle: reference
    textAlignPreprocessing.py:160-195 for a list of uint8 greyscale pages"""
    d = _Dev(device)
    host_px = []
    for pg in pages:
        px = np.asarray(getattr(pg, "pixels", pg))
        if px.dtype != np.uint8 or px.ndim != 2:
            raise TypeError("the device preprocessing takes 2-D uint8 pages")
        host_px.append(np.ascontiguousarray(px))
    imgs = _upload_pages(d, host_px)
    thrs = otsu_thresholds_device(d, imgs)
    inks = []
    for img, thr in zip(imgs, thrs):
        ink = torch.empty_like(img)
        _native.check(d.lib.ta_pp_threshold(img.data_ptr(), img.numel(), thr, 0, ink.data_ptr(), d.stream),
                      "ta_pp_threshold")
        inks.append(ink)
    d.despeckle_many(inks, despeckle_amt)
    for ink in inks:
        d.invert(ink)                                        # fill small holes: despeckle the background
    d.despeckle_many(inks, despeckle_amt)
    for ink in inks:
        d.invert(ink)
    for ink, (lab, stats) in zip(inks, d.label_many(inks)):  # drop components taller than the threshold
        d.filter(ink, lab, stats, max_height=host.sat_area_thresh)
    skews = rotation_angles_device(d, inks, -6, 6)
    out = []
    if correct_rotation:
        inks = rotate_many(d, inks, skews)
    for ink, skew in zip(inks, skews):
        eroded = ink
        for _ in range(filter_runs):
            eroded = open_runs_device(d, eroded, filter_runs_amt, 0)
            eroded = open_runs_device(d, eroded, filter_runs_amt, 1)
        if eroded is ink:
            eroded = ink.clone()
        out.append((ink, eroded, host.reported_angle(skew)))     # sign: see host.reported_angle
    return d, out


def preprocess_images(input_image, despeckle_amt=host.despeckle_amt, filter_runs=1, filter_runs_amt=2,
                      correct_rotation=True, device="cuda"):
    """(d, ink, eroded, angle) of one uint8 greyscale page"""
    d, out = preprocess_images_batch([input_image], despeckle_amt, filter_runs, filter_runs_amt,
                                     correct_rotation, device)
    return (d,) + out[0]


def identify_text_lines_batch(d, planes):
    """text lines of preprocessed pages (reference textAlignPreprocessing.py:198-285) from their
    (ink, eroded) device planes: [(line strips, peak locations, smoothed projection)]"""
    n = len(planes)
    sums = []
    for ink, eroded in planes:
        h, w = eroded.shape
        s = torch.empty(h, dtype=torch.int32, device=d.dev)
        _native.check(d.lib.ta_pp_row_sums(eroded.data_ptr(), h, w, s.data_ptr(), d.stream), "ta_pp_row_sums")
        sums.append(s)
    flat = torch.cat(sums).cpu().numpy().astype(np.int64) if n else np.zeros(0, np.int64)
    smoothed_all, peaks_all, works, pos, row_lists = [], [], [], 0, []
    for ink, eroded in planes:
        h, w = eroded.shape
        project = flat[pos:pos + h]
        pos += h
        smoothed = host.moving_avg_filter(project, host.filter_size)
        peaks = host.find_peak_locations(smoothed)
        rows = []
        for a, b in zip(peaks[:-1], peaks[1:]):
            idx = int(np.argmin(smoothed[a:b])) + a
            rows.extend(range(max(idx - 1, 0), idx + 1))          # 2-pixel white line
        row_lists.append(np.array(sorted(set(rows)), dtype=np.int32))
        smoothed_all.append(smoothed); peaks_all.append(peaks)
    d_rows = _native.upload_packed(row_lists, d.dev) if n else []
    for (ink, eroded), rows, dr in zip(planes, row_lists, d_rows):
        work = eroded.clone()
        if len(rows):
            _native.check(d.lib.ta_pp_clear_rows(work.data_ptr(), eroded.shape[1], dr.data_ptr(), len(rows), d.stream),
                          "ta_pp_clear_rows")
        works.append(work)
    recs_all = d.components_many(d.label_many(works))
    boxes_all, total = [], 0
    for (ink, eroded), peaks, recs in zip(planes, peaks_all, recs_all):
        boxes = []
        big = recs[recs[:, 1] > host.noise_area_thresh]
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
    # cut the strips on the device (ink black on white, as the reference saves them) into one packed buffer
    # and leave them there: the recogniser's normaliser reads them where they are, `strip.pixels` downloads
    packed = torch.empty(max(total, 1), dtype=torch.uint8, device=d.dev)
    d_boxes = _native.upload_packed([np.array(boxes, dtype=np.int64).reshape(-1, 5) for boxes in boxes_all], d.dev) if n else []
    for (ink, eroded), boxes, db in zip(planes, boxes_all, d_boxes):
        if boxes:
            h, w = ink.shape
            _native.check(d.lib.ta_pp_cut_strips(ink.data_ptr(), h, w, db.data_ptr(), len(boxes),
                                                 packed.data_ptr(), d.stream), "ta_pp_cut_strips")
    out = []
    for boxes, peaks, smoothed in zip(boxes_all, peaks_all, smoothed_all):
        strips = []
        for ulx, uly, lrx, lry, off in boxes:
            hh, ww = lry - uly + 1, lrx - ulx + 1
            strips.append(page_mod.Strip(ulx, uly, hh, width=ww,
                                         device_pixels=packed[off:off + hh * ww].view(hh, ww)))
        out.append((strips, peaks, smoothed))
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
    lines = identify_text_lines_batch(d, [(ink, eroded) for ink, eroded, _ in pre])
    return [(DeviceBinImage(ink, d), DeviceBinImage(eroded, d), angle, strips, peaks)
            for (ink, eroded, angle), (strips, peaks, _) in zip(pre, lines)]


def find_lines(input_image, device="cuda"):
    """(image_bin, image_eroded, angle, strips, peak locations) of one uint8 greyscale page"""
    return find_lines_batch([input_image], device=device)[0]
