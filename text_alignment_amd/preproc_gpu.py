"""Device side of the page preprocessing (csrc/ta_preproc.hip): the image operations of
`textAlignPreprocessing.preprocess_images` / `identify_text_lines` (the Gamera-free restatement of
reference textAlignPreprocessing.py:160-285) on the GPU, one page at a time.  Control flow, the
projection / peak numerics and the selection of components are Python, as in the reference; every
full-page pass is a kernel: Otsu histogram, thresholding, connected components (despeckle, hole
filling, tall-component removal, line components), the skew search, the rotation, the run filters
and the row projection.  uint8 greyscale pages (textAlignPreprocessing.to_grey_u8 reduces anything
else).  Checked against the scipy restatement oracle/preproc_ref.py (tests/test_preproc_gpu.py).
"""
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


class _Dev(object):
    def __init__(self, device="cuda"):
        self.dev = torch.device(device)
        if self.dev.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("text_alignment_amd needs an AMD GPU (MI355X): page preprocessing runs as HIP "
                               "kernels and there is no CPU fallback")
        self.lib = _native.lib
        self.flag = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.count = torch.zeros(1, dtype=torch.int32, device=self.dev)

    @property
    def stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def label(self, ink):
        """(lab, stats) of a uint8 ink plane: labels and the five per-root statistics planes"""
        h, w = ink.shape
        lab = torch.empty((h, w), dtype=torch.int32, device=self.dev)
        stats = torch.empty((5, h, w), dtype=torch.int32, device=self.dev)
        _native.check(self.lib.ta_pp_label(ink.data_ptr(), h, w, lab.data_ptr(), stats.data_ptr(),
                                           self.flag.data_ptr(), self.stream), "ta_pp_label")
        return lab, stats

    def filter(self, ink, lab, stats, min_area=0, max_height=2 ** 30):
        h, w = ink.shape
        _native.check(self.lib.ta_pp_filter_components(ink.data_ptr(), lab.data_ptr(), stats.data_ptr(), h, w,
                                                       int(min_area), int(max_height), self.stream),
                      "ta_pp_filter_components")

    def components(self, lab, stats, cap=1 << 16):
        """host array [ncomp][6] = {root, area, x0, y0, x1, y1}, sorted by root (raster order)"""
        h, w = lab.shape
        while True:
            recs = torch.empty((cap, 6), dtype=torch.int32, device=self.dev)
            _native.check(self.lib.ta_pp_components(lab.data_ptr(), stats.data_ptr(), h, w, recs.data_ptr(),
                                                    cap, self.count.data_ptr(), self.stream), "ta_pp_components")
            n = int(self.count.item())
            if n <= cap:
                out = recs[:n].cpu().numpy()
                return out[np.argsort(out[:, 0], kind="stable")]
            cap = n

    def despeckle(self, ink, size):
        lab, stats = self.label(ink)
        self.filter(ink, lab, stats, min_area=size)

    def invert(self, ink):
        _native.check(self.lib.ta_pp_invert(ink.data_ptr(), ink.numel(), self.stream), "ta_pp_invert")


def otsu_threshold_device(d, img):
    hist = torch.empty(256, dtype=torch.int32, device=d.dev)
    _native.check(d.lib.ta_pp_histogram(img.data_ptr(), img.numel(), hist.data_ptr(), d.stream), "ta_pp_histogram")
    hist = hist.cpu().numpy().astype(np.float64)
    total = hist.sum()
    cum = np.cumsum(hist)
    mean_cum = np.cumsum(hist * np.arange(256))
    mean_all = mean_cum[-1]
    with np.errstate(divide='ignore', invalid='ignore'):
        between = (mean_all * cum - mean_cum * total) ** 2 / (cum * (total - cum))
    between[~np.isfinite(between)] = 0
    return int(np.argmax(between))


def rotation_angle_device(d, ink, lo=-6.0, hi=6.0, coarse=0.25, fine=0.05):
    """host.rotation_angle_projections with the per-angle row histograms built on the device"""
    h, w = ink.shape
    step = max(1, int(max(h, w) / 1200))
    hs = (h + step - 1) // step

    def scores(grid):
        cs = np.empty(2 * len(grid), np.float64)
        rad = np.deg2rad(grid)
        cs[0::2], cs[1::2] = np.cos(rad), np.sin(rad)
        d_cs = torch.from_numpy(cs).to(d.dev)
        hist = torch.empty((len(grid), hs), dtype=torch.int32, device=d.dev)
        _native.check(d.lib.ta_pp_angle_histograms(ink.data_ptr(), h, w, step, d_cs.data_ptr(), len(grid),
                                                   hist.data_ptr(), d.stream), "ta_pp_angle_histograms")
        hh = hist.cpu().numpy()
        return [float(np.var(hh[k])) for k in range(len(grid))]
    if not bool(ink.any()):
        return 0.0
    grid = np.arange(lo, hi + 1e-9, coarse)
    best = grid[int(np.argmax(scores(grid)))]
    grid = np.arange(best - coarse, best + coarse + 1e-9, fine)
    best = grid[int(np.argmax(scores(grid)))]
    return float(np.round(best, 3))


def rotate_device(d, ink, angle):
    """host.rotate: scipy.ndimage.rotate(float32(ink), angle, reshape=True, order=1) > 0.5, with
    scipy's own geometry (ndimage/_interpolation.py rotate) computed here and the resampling on
    the device"""
    if angle == 0:
        return ink.clone()
    h, w = ink.shape
    c, s = special.cosdg(angle), special.sindg(angle)
    rot = np.array([[c, s], [-s, c]])
    out_bounds = rot @ [[0, 0, h, h], [0, w, 0, w]]
    out_shape = (np.ptp(out_bounds, axis=1) + 0.5).astype(int)
    out_center = rot @ ((out_shape - 1) / 2)
    in_center = (np.asarray([h, w]) - 1) / 2
    offset = in_center - out_center
    mo = torch.from_numpy(np.array([rot[0, 0], rot[0, 1], rot[1, 0], rot[1, 1], offset[0], offset[1]],
                                   np.float64)).to(d.dev)
    oh, ow = int(out_shape[0]), int(out_shape[1])
    out = torch.empty((oh, ow), dtype=torch.uint8, device=d.dev)
    _native.check(d.lib.ta_pp_rotate(ink.data_ptr(), h, w, out.data_ptr(), oh, ow, mo.data_ptr(), d.stream),
                  "ta_pp_rotate")
    return out


def open_runs_device(d, ink, length, axis):
    if length <= 1:
        return ink
    h, w = ink.shape
    out = torch.empty_like(ink)
    _native.check(d.lib.ta_pp_open_runs(ink.data_ptr(), out.data_ptr(), h, w, int(length), int(axis), d.stream),
                  "ta_pp_open_runs")
    return out


def preprocess_images(input_image, despeckle_amt=host.despeckle_amt, filter_runs=1, filter_runs_amt=2,
                      correct_rotation=True, device="cuda"):
    """(ink, eroded, angle) as uint8 device planes: the device counterpart of
    textAlignPreprocessing.preprocess_images for a uint8 greyscale page"""
    px = np.asarray(getattr(input_image, "pixels", input_image))
    if px.dtype != np.uint8 or px.ndim != 2:
        raise TypeError("the device preprocessing takes 2-D uint8 pages")
    d = _Dev(device)
    img = torch.from_numpy(np.ascontiguousarray(px)).to(d.dev)
    thr = otsu_threshold_device(d, img)
    ink = torch.empty_like(img)
    _native.check(d.lib.ta_pp_threshold(img.data_ptr(), img.numel(), thr, 0, ink.data_ptr(), d.stream),
                  "ta_pp_threshold")
    d.despeckle(ink, despeckle_amt)
    d.invert(ink)                                            # fill small holes: despeckle the background
    d.despeckle(ink, despeckle_amt)
    d.invert(ink)
    lab, stats = d.label(ink)                                # drop components taller than the threshold
    d.filter(ink, lab, stats, max_height=host.sat_area_thresh)
    skew = rotation_angle_device(d, ink, -6, 6)
    if correct_rotation:
        ink = rotate_device(d, ink, skew)
    eroded = ink
    for _ in range(filter_runs):
        eroded = open_runs_device(d, eroded, filter_runs_amt, 0)
        eroded = open_runs_device(d, eroded, filter_runs_amt, 1)
    if eroded is ink:
        eroded = ink.clone()
    return d, ink, eroded, host.reported_angle(skew)     # sign: see host.reported_angle


def find_lines(input_image, device="cuda"):
    """(image_bin, image_eroded, angle, strips, peak locations) of one uint8 greyscale page"""
    d, ink, eroded, angle = preprocess_images(input_image, device=device)
    image_bin, image_eroded = DeviceBinImage(ink, d), DeviceBinImage(eroded, d)
    strips, peaks, _ = identify_text_lines(image_bin, image_eroded)
    return image_bin, image_eroded, angle, strips, peaks


def identify_text_lines(image_bin, image_eroded):
    """text lines of a preprocessed page (reference textAlignPreprocessing.py:198-285) from the two
    device planes: (line strips, peak locations, smoothed projection)"""
    ink, eroded = image_bin.plane, image_eroded.plane
    d = image_bin.dev or _Dev(ink.device)
    h, w = eroded.shape
    sums = torch.empty(h, dtype=torch.int32, device=d.dev)
    _native.check(d.lib.ta_pp_row_sums(eroded.data_ptr(), h, w, sums.data_ptr(), d.stream), "ta_pp_row_sums")
    project = sums.cpu().numpy().astype(np.int64)
    smoothed = host.moving_avg_filter(project, host.filter_size)
    peaks = host.find_peak_locations(smoothed)
    rows = []
    for a, b in zip(peaks[:-1], peaks[1:]):
        idx = int(np.argmin(smoothed[a:b])) + a
        rows.extend(range(max(idx - 1, 0), idx + 1))          # 2-pixel white line
    work = eroded.clone()
    if rows:
        d_rows = torch.tensor(sorted(set(rows)), dtype=torch.int32, device=d.dev)
        _native.check(d.lib.ta_pp_clear_rows(work.data_ptr(), w, d_rows.data_ptr(), d_rows.numel(), d.stream),
                      "ta_pp_clear_rows")
    lab, stats = d.label(work)
    recs = d.components(lab, stats)
    comps = [(int(r[2]), int(r[3]), int(r[4]), int(r[5])) for r in recs if r[1] > host.noise_area_thresh]
    if not comps:
        return [], peaks, smoothed
    heights = [c[3] - c[1] + 1 for c in comps]
    med = np.median(heights)
    comps = [c for c, hgt in zip(comps, heights) if hgt < med * host.remove_capitals_scale]
    cc_median_height = np.median([c[3] - c[1] + 1 for c in comps])
    boxes = []
    box = np.asarray(comps, dtype=np.int64)
    for loc in peaks:
        hit = box[host.coincide_mask(loc, box[:, 1], box[:, 3] - box[:, 1] + 1, cc_median_height)]
        if not len(hit):
            continue
        boxes.append((int(hit[:, 0].min()), int(hit[:, 1].min()), int(hit[:, 2].max()), int(hit[:, 3].max())))
    # cut the strips on the device (ink black on white, as the reference saves them) and bring
    # only those over: the page itself stays where it is
    flat = [((1 - ink[uly:lry + 1, ulx:lrx + 1]) * 255).reshape(-1) for ulx, uly, lrx, lry in boxes]
    packed = torch.cat(flat).cpu().numpy() if flat else np.zeros(0, np.uint8)
    strips, pos = [], 0
    for ulx, uly, lrx, lry in boxes:
        hh, ww = lry - uly + 1, lrx - ulx + 1
        pixels = packed[pos:pos + hh * ww].reshape(hh, ww)
        pos += hh * ww
        strips.append(page_mod.Strip(ulx, uly, hh, width=ww, pixels=pixels))
    return strips, peaks, smoothed
