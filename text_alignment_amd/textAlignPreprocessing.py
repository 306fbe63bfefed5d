"""Gamera-free preprocessing and text-line finding.  Counterpart of
the reference module of the same name (reference textAlignPreprocessing.py:38-285), so that
`alignToOCR.process` can start from a raw text-layer image (a numpy array) instead of a
`page.PreparedPage`.

* The projection / peak-finding numerics (`moving_avg_filter` :147, `calculate_peak_prominence`
  :59, `find_peak_locations` :113, `vertically_coincide` :38 of the reference) are plain numpy in
  the reference too; they are restated here, on the host as in the reference, and PINNED to golden
  vectors captured from the imported reference (tests/golden/preproc.json).
* The image operations the reference delegates to the Gamera C++ toolkit (`to_onebit`,
  `despeckle`, `cc_analysis`, `rotation_angle_projections`, `rotate`, `filter_short_runs`,
  `filter_narrow_runs`, `projection_rows`, `draw_line`, `subimage`; reference :167-195, :212-253)
  run as HIP kernels (csrc/ta_preproc.hip, driven by preproc_gpu.py).  There is no host version of
  them in the package: `preprocess_images` / `identify_text_lines` / `find_lines` below are the
  reference's entry points over the device path, and the restatement the kernels are checked
  against lives with the other checkers (oracle/preproc_ref.py).  Gamera is not installed here, so
  both are PARITY UNPINNED: same pipeline, same parameters, not bit-checked against Gamera.

Pages are numpy arrays (greyscale or colour of any numeric type, or bool with True = ink).
"""
import numpy as np

from . import page as page_mod

# PARAMETERS FOR PREPROCESSING (reference textAlignPreprocessing.py:12-16)
saturation_thresh = 0.9
sat_area_thresh = 150
despeckle_amt = 100
noise_area_thresh = 100

# PARAMETERS FOR TEXT LINE SEGMENTATION (reference :18-22)
filter_size = 30
prominence_tolerance = 0.70
collision_strip_scale = 1
remove_capitals_scale = 10000



# --------------------------------------------------------------------------- pinned numerics
def vertically_coincide(hline_position, comp_offset, comp_nrows, collision,
                        collision_scale=collision_strip_scale):
    """True if any part of a component (rows comp_offset .. comp_offset + comp_nrows) lies within
    the horizontal strip of height `collision` centred on hline_position (reference :38-56)."""
    collision *= collision_strip_scale
    top, bottom = comp_offset, comp_offset + comp_nrows
    strip_top = hline_position - int(collision / 2)
    strip_bottom = hline_position + int(collision / 2)
    above = top < strip_top and bottom < strip_top
    below = top > strip_bottom and bottom > strip_bottom
    return (not above and not below)


def coincide_mask(hline_position, comp_offsets, comp_nrows, collision):
    """vertically_coincide for many components at once (arrays of offsets and heights)"""
    collision = collision * collision_strip_scale
    top = np.asarray(comp_offsets)
    bottom = top + np.asarray(comp_nrows)
    strip_top = hline_position - int(collision / 2)
    strip_bottom = hline_position + int(collision / 2)
    above = (top < strip_top) & (bottom < strip_top)
    below = (top > strip_bottom) & (bottom > strip_bottom)
    return ~above & ~below


def line_boxes(peak_locs, comps, collision):
    """the bounding box of every text line of a page: for each peak location the union (ulx, uly, lrx, lry) of the
    components [ulx, uly, lrx, lry] that vertically_coincide with it, lines without a component left out (reference
    :253-276, one line at a time there).  Integer comparisons in the library's host loop (ta_host_line_boxes; a page has
    ~30 lines and up to a few thousand components); line_boxes_numpy is the same as one numpy table."""
    comps = np.ascontiguousarray(comps, dtype=np.int64).reshape(-1, 4)
    loc = np.ascontiguousarray(peak_locs, dtype=np.int64).reshape(-1)
    if not len(loc) or not len(comps):
        return []
    from . import _native
    half = int(collision * collision_strip_scale / 2)
    boxes, hit = np.empty((len(loc), 4), np.int64), np.empty(len(loc), np.uint8)
    _native.check(_native.lib.ta_host_line_boxes(comps.ctypes.data, len(comps), loc.ctypes.data, len(loc), half,
                                                 boxes.ctypes.data, hit.ctypes.data), "ta_host_line_boxes")
    return boxes[hit.astype(bool)].tolist()


def line_boxes_numpy(peak_locs, comps, collision):
    """line_boxes as one (peaks x components) numpy table: the cross-check of the native loop"""
    comps = np.asarray(comps, dtype=np.int64).reshape(-1, 4)
    loc = np.asarray(peak_locs, dtype=np.int64).reshape(-1, 1)
    if not len(loc) or not len(comps):
        return []
    half = int(collision * collision_strip_scale / 2)
    top = comps[:, 1][None, :]
    bottom = top + (comps[:, 3] - comps[:, 1] + 1)[None, :]
    strip_top, strip_bottom = loc - half, loc + half
    hit = ~((top < strip_top) & (bottom < strip_top)) & ~((top > strip_bottom) & (bottom > strip_bottom))
    big = np.iinfo(np.int64).max
    ulx = np.where(hit, comps[:, 0][None, :], big).min(axis=1)
    uly = np.where(hit, comps[:, 1][None, :], big).min(axis=1)
    lrx = np.where(hit, comps[:, 2][None, :], -big).max(axis=1)
    lry = np.where(hit, comps[:, 3][None, :], -big).max(axis=1)
    keep = hit.any(axis=1)
    return np.stack([ulx, uly, lrx, lry], axis=1)[keep].tolist()


def calculate_peak_prominence(data, index, data_max=None):
    '''log of the prominence of the peak at `index`: isolated peaks score high, peaks in the
    foothills of larger ones low (reference :59-110).  `data_max` may carry max(data) when many
    indices of the same array are scored.'''
    here = data[index]
    if (index == 0 or index == len(data) - 1 or data[index - 1] > here or data[index + 1] > here or
            (data[index - 1] == here and data[index + 1] == here)):
        return 0
    if here == (max(data) if data_max is None else data_max):
        return np.log(here)
    higher = np.nonzero(np.asarray(data) > here)[0]                 # indices of everything above this peak
    cut = int(np.searchsorted(higher, index))
    nearest_right = higher[cut] if cut < len(higher) else np.inf
    nearest_left = higher[cut - 1] if cut > 0 else -np.inf
    nearest = nearest_left if (nearest_right - index) > (index - nearest_left) else nearest_right
    lo, hi = min(nearest, index), max(nearest, index)
    key_col = np.min(data[int(lo):int(hi)]) if isinstance(data, np.ndarray) else min(data[int(lo):int(hi)])
    return np.log(data[index] - key_col + 1)


def find_peak_locations(data, tol=prominence_tolerance, ranked=False):
    '''indices of the prominent peaks of a row projection (reference :113-144)'''
    d = np.asarray(data)
    if len(d) == 0 or tol < 0:
        return _find_peak_locations_all_rows(data, tol, ranked)
    data_max = d.max()
    # only local maxima can score: find them in one array pass (the same test
    # calculate_peak_prominence starts with); every other row has prominence 0, which never
    # exceeds a tolerance >= 0, so only the candidates are scored and normalised
    cand = np.zeros(len(d), dtype=bool)
    if len(d) > 2:
        mid, left, right = d[1:-1], d[:-2], d[2:]
        cand[1:-1] = ~((left > mid) | (right > mid) | ((left == mid) & (right == mid)))
    idx = np.flatnonzero(cand).tolist()
    if isinstance(data, np.ndarray) and d.ndim == 1:
        vals = (_candidate_prominences_native if d.dtype == np.float64 else _candidate_prominences)(d, idx, data_max)
    else:
        vals = [calculate_peak_prominence(d, i, data_max) for i in idx]
    top = max(vals + [0])             # rows that are no candidates score 0 (the first and the last row always do)
    if top == 0:
        return []
    peaks = [(i, v / top) for i, v in zip(idx, vals) if v / top > tol]
    # both corners of a flat-topped peak are prominent: drop the first of two equal neighbours
    dupes = [peaks[i] for i in range(len(peaks) - 2) if peaks[i][1] == peaks[i + 1][1]]
    for dup in dupes:
        peaks.remove(dup)
    if ranked:
        peaks.sort(key=lambda p: p[1] * -1)
        return peaks
    return [p[0] for p in peaks]


def _candidate_prominences_native(d, idx, data_max):
    """the same for a float64 projection with the scans in the library's host code
    (ta_pp_peak_prominence_args: plain float64 loops, no device work) -- a page's ~60 candidates over its
    ~4 000 rows are a quarter-million comparisons either way, but there they neither build tables nor hold
    the interpreter lock; the logarithms stay here, one numpy scalar at a time"""
    if not idx:
        return []
    from . import _native
    dd = np.ascontiguousarray(d)
    ix = np.asarray(idx, dtype=np.int32)
    arg = np.empty(len(ix), dtype=np.float64)
    _native.check(_native.lib.ta_pp_peak_prominence_args(dd.ctypes.data, len(dd), ix.ctypes.data, len(ix),
                                                         float(data_max), arg.ctypes.data),
                  "ta_pp_peak_prominence_args")
    return [np.log(v) for v in arg]


def _candidate_prominences(d, idx, data_max):
    """[calculate_peak_prominence(d, i, data_max) for i in idx] for rows that passed its local-maximum
    test, all at once: the nearest higher sample to either side from one comparison table, the key
    column from one pass of minimum.reduceat; the logarithms are still taken one scalar at a time, as
    the per-row form takes them"""
    if not idx:
        return []
    n, ix = len(d), np.asarray(idx)
    here = d[ix]
    higher = d[None, :] > here[:, None]
    cols = np.arange(n)[None, :]
    right = higher & (cols > ix[:, None])
    left = higher & (cols < ix[:, None])
    has_r, has_l = right.any(axis=1), left.any(axis=1)
    nr = right.argmax(axis=1)                                   # first higher sample to the right
    nl = n - 1 - left[:, ::-1].argmax(axis=1)                   # last one to the left
    dist_r = np.where(has_r, nr - ix, np.inf)
    dist_l = np.where(has_l, ix - nl, np.inf)
    go_left = dist_r > dist_l
    lo = np.where(go_left, nl, ix)
    hi = np.where(go_left, ix, nr)
    top = here == data_max                                      # nothing is higher: scored by its height
    lo[top], hi[top] = 0, 1
    bounds = np.empty(2 * len(ix), dtype=np.intp)
    bounds[0::2], bounds[1::2] = lo, hi
    key = np.minimum.reduceat(d, bounds)[0::2]
    arg = np.where(top, here, here - key + 1)
    return [np.log(v) for v in arg]


def peaks_of_projections(flat, offs, lens, tol=prominence_tolerance, size=filter_size):
    """[(smoothed, peaks, white rows)] of a batch of pages' row projections (page k: flat[offs[k] : offs[k] + lens[k]],
    integer sums): moving_avg_filter, find_peak_locations and the rows `identify_text_lines` clears between
    neighbouring peaks (reference :147-157, :113-144, :222-232) -- the library's host loops
    (ta_host_peak_candidates / ta_host_peak_select) around ONE np.log of all pages' candidates, instead of a dozen
    numpy calls and a Python loop per page under the interpreter lock.  peaks_of_projections_numpy is the per-page form."""
    n = len(offs)
    lens32 = np.ascontiguousarray(lens, dtype=np.int32)
    if n == 0:
        return []
    if tol < 0 or int(lens32.min()) == 0:
        return peaks_of_projections_numpy(flat, offs, lens, tol, size)       # (rows of prominence 0 as peaks; empty data raises)
    from . import _native
    lib = _native.lib
    flat = np.ascontiguousarray(flat, dtype=np.int64)
    off = np.ascontiguousarray(offs, dtype=np.int64)
    if int((off + lens32).max()) > flat.size or int(off.min()) < 0:
        raise ValueError("a page's projection lies outside the buffer")
    smoothed = np.empty(flat.size, np.float64)
    cand_idx, cand_arg, cand_n = np.empty(flat.size, np.int32), np.ones(flat.size, np.float64), np.zeros(n, np.int32)
    _native.check(lib.ta_host_peak_candidates(flat.ctypes.data, off.ctypes.data, lens32.ctypes.data, n, int(size),
                                              smoothed.ctypes.data, cand_idx.ctypes.data, cand_arg.ctypes.data,
                                              cand_n.ctypes.data), "ta_host_peak_candidates")
    # numpy's logarithm of every page's candidates in one call (element for element what np.log gives one at a time)
    total = int(cand_n.sum())
    first = np.cumsum(cand_n) - cand_n
    pos = np.repeat(off - first, cand_n) + np.arange(total)
    cand_log = np.zeros(flat.size, np.float64)
    cand_log[pos] = np.log(cand_arg[pos])
    peaks, npeaks = np.empty(flat.size, np.int32), np.zeros(n, np.int32)
    rows, nrows = np.empty(2 * flat.size, np.int32), np.zeros(n, np.int32)
    _native.check(lib.ta_host_peak_select(smoothed.ctypes.data, off.ctypes.data, lens32.ctypes.data, n, cand_idx.ctypes.data,
                                          cand_log.ctypes.data, cand_n.ctypes.data, float(tol), peaks.ctypes.data,
                                          npeaks.ctypes.data, rows.ctypes.data, nrows.ctypes.data), "ta_host_peak_select")
    out = []
    for k in range(n):
        o = int(off[k])
        out.append((smoothed[o:o + int(lens32[k])], peaks[o:o + int(npeaks[k])].tolist(),
                    rows[2 * o:2 * o + int(nrows[k])].copy()))
    return out


def peaks_of_projections_numpy(flat, offs, lens, tol=prominence_tolerance, size=filter_size):
    """peaks_of_projections page by page with this module's numpy functions"""
    out = []
    for o, m in zip(offs, lens):
        project = np.asarray(flat[int(o):int(o) + int(m)]).astype(np.int64)
        smoothed = moving_avg_filter(project, size)
        peaks = find_peak_locations(smoothed, tol)
        rows = []
        for a, b in zip(peaks[:-1], peaks[1:]):
            idx = int(np.argmin(smoothed[a:b])) + a
            rows.extend(range(max(idx - 1, 0), idx + 1))          # 2-pixel white line
        out.append((smoothed, peaks, np.array(sorted(set(rows)), dtype=np.int32)))
    return out


def _find_peak_locations_all_rows(data, tol, ranked):
    """the reference's loop over every row, for the cases the candidate form does not cover (a
    negative tolerance makes rows of prominence 0 peaks; empty data raises as the reference does)"""
    data_max = max(data) if len(data) else None
    proms = [(i, calculate_peak_prominence(data, i, data_max)) for i in range(len(data))]
    top = max([p[1] for p in proms])
    if top == 0 or len(proms) == 0:
        return []
    proms = [(i, v / top) for i, v in proms]
    peaks = [p for p in proms if p[1] > tol]
    dupes = [peaks[i] for i in range(len(peaks) - 2) if peaks[i][1] == peaks[i + 1][1]]
    for dup in dupes:
        peaks.remove(dup)
    if ranked:
        peaks.sort(key=lambda p: p[1] * -1)
        return peaks
    return [p[0] for p in peaks]


def moving_avg_filter(data, filter_size=filter_size):
    '''moving average over filter_size samples to either side; the ends stay zero (reference :147-157)'''
    smoothed = np.zeros(len(data))
    data = np.asarray(data)
    n = len(data)
    if n > 2 * filter_size and data.dtype.kind in "iub":
        # integer samples (a row projection): window sums are exact in any order, so a running
        # sum gives bit for bit what np.mean gives per window
        c = np.concatenate([[0], np.cumsum(data.astype(np.int64))])
        win = c[2 * filter_size + 1:] - c[:n - 2 * filter_size]
        smoothed[filter_size:n - filter_size] = win / float(2 * filter_size + 1)
        return smoothed
    for k in range(filter_size, n - filter_size):
        smoothed[k] = np.mean(data[k - filter_size: k + filter_size + 1])
    return smoothed


# --------------------------------------------------------------------------- page images
def to_grey_u8(image):
    """RGB / greyscale of any numeric type, or a bool onebit image (True = ink) -> 2-D uint8 greyscale,
    what the device threshold works on.  (A two-valued image thresholds to itself: Otsu's first
    maximum on a histogram with spikes at 0 and 255 is 0.)"""
    a = np.asarray(image)
    if a.dtype == bool:
        return np.where(a, 0, 255).astype(np.uint8)
    if a.ndim == 3:
        a = a[..., :3].mean(axis=2)
    if a.dtype != np.uint8:
        a = np.clip(a * (255.0 if a.max() <= 1.0 else 1.0), 0, 255).astype(np.uint8)
    return a


def reported_angle(skew):
    """The angle `preprocess_images` hands to `process`.  `process` maps syllable boxes back onto the
    raw page with rotate_bbox(box, -angle, ...) (reference alignToOCR.py:327-328), whose rotation
    x' = x cos a - y sin a, y' = x sin a + y cos a (image coordinates, y down; alignToOCR.py:104-112)
    turns the opposite way from the device rotation (which follows scipy.ndimage.rotate(img, a)), so
    the angle that makes rotate_bbox(-angle) the exact inverse of the deskewing is minus the angle
    that was applied."""
    return -skew if skew != 0 else 0.0


# --------------------------------------------------------------------------- the two entry points
def _pixels(input_image):
    px = getattr(input_image, "pixels", input_image)
    if type(px).__module__.split(".")[0] == "torch":           # a torch tensor: a 2-D uint8 page is taken where it lies (on the
        if px.dim() == 2 and str(px.dtype) == "torch.uint8":   # device, or in page-locked host memory); anything else comes
            return px                                          # to the host and is reduced there
        px = px.cpu().numpy()
    px = np.asarray(px)
    if px.ndim not in (2, 3):
        raise TypeError("a page image is a 2-D (greyscale / onebit) or 3-D (colour) array")
    return to_grey_u8(px)


def preprocess_images(input_image, despeckle_amt=despeckle_amt, filter_runs=1, filter_runs_amt=2,
                      correct_rotation=True):
    '''denoise and deskew the text layer before text-line segmentation (reference :160-195), on the
    GPU.  Returns (image_bin, image_eroded, angle); the two images stay on the device
    (preproc_gpu.DeviceBinImage: `.dim` / `.ncols` / `.nrows`, `.ink` downloads the bits).
    A PreparedPage passes straight through.'''
    if isinstance(input_image, page_mod.PreparedPage):
        return page_mod.preprocess_images(input_image)
    from . import preproc_gpu
    d, ink, eroded, angle = preproc_gpu.preprocess_images(_pixels(input_image), despeckle_amt=despeckle_amt,
                                                          filter_runs=filter_runs, filter_runs_amt=filter_runs_amt,
                                                          correct_rotation=correct_rotation)
    return preproc_gpu.DeviceBinImage(ink, d), preproc_gpu.DeviceBinImage(eroded, d), angle


def identify_text_lines(image_bin, image_eroded):
    '''text lines of a preprocessed page (reference :198-285): peaks of the smoothed row
    projection; a white line at the projection minimum between neighbouring peaks; connected
    components; per peak the union of the components a strip around the peak touches.
    Returns (line_strips, peak_locations, smoothed_projection).'''
    if hasattr(image_bin, "_page"):
        return page_mod.identify_text_lines(image_bin, image_eroded)
    from . import preproc_gpu
    return preproc_gpu.identify_text_lines(image_bin, image_eroded)


def find_lines(input_image):
    """preprocess_images + identify_text_lines of one page in one call: (image_bin, image_eroded,
    angle, line strips, peak locations)."""
    return find_lines_many([input_image])[0]


# pages whose preprocessing shares its stage calls and waits for the device (~0.2 GB of planes each), and batches in flight
# at once, each driven by a host thread on a HIP stream of its own.  8 x 2 while every page's launches were made from
# Python (more threads only fought for the interpreter lock); with a stage per library call (preproc_gpu, round 6) a thread
# holds the lock for ~0.3 ms per page and 4 x 4 kept the device fed better (64 pages, same box, 8 x 2: 682-698 pages/s,
# 4 x 4: 761-800); once every kernel of a stage is ONE launch for the batch's pages (csrc/ta_preproc.hip, blockIdx.z =
# page: a stream runs a page's small kernels one after the other, so larger batches no longer lengthen the chain) 8 x 2
# again: 950-990 pages/s against 890-925 for 4 x 4, with 1.6 instead of 2.3 ms of host time per page
PAGES_PER_BATCH = 8
PAGE_THREADS = 2


def find_lines_many(pages):
    """find_lines of every page.  The page images go through the device pipeline in batches whose
    data-dependent host decisions share their waits for the device (preproc_gpu.find_lines_batch),
    and a few batches are in flight at once -- one host thread and one HIP stream each -- so that one
    batch's Python (peak finding, component selection) runs while another's kernels do; ctypes and
    torch release the GIL while a call waits for the device.  PreparedPages pass through.  Per page
    the result is that of find_lines, whatever the interleaving."""
    out = [None] * len(pages)
    todo = []
    for k, pg in enumerate(pages):
        if isinstance(pg, page_mod.PreparedPage):
            image, eroded, angle = page_mod.preprocess_images(pg)
            strips, peaks, _ = page_mod.identify_text_lines(image, eroded)
            out[k] = (image, eroded, angle, strips, peaks)
        else:
            todo.append(k)
    if not todo:
        return out
    from . import preproc_gpu
    grey = [_pixels(pages[k]) for k in todo]                   # bad page types fail before any GPU work
    spans = [(a, min(a + PAGES_PER_BATCH, len(todo))) for a in range(0, len(todo), PAGES_PER_BATCH)]

    def run(span):
        res = preproc_gpu.find_lines_batch(grey[span[0]:span[1]])
        for k, r in zip(todo[span[0]:span[1]], res):
            out[k] = r
    nthreads = min(PAGE_THREADS, len(spans))
    if nthreads <= 1:
        for span in spans:
            run(span)
        return out
    import torch
    main = torch.cuda.current_stream()
    device = main.device
    streams = _page_streams(device, nthreads)

    def worker(t):
        torch.cuda.set_device(device)
        streams[t].wait_stream(main)
        with torch.cuda.stream(streams[t]):
            for span in spans[t::nthreads]:
                run(span)
        streams[t].synchronize()                  # the planes are read on the caller's stream afterwards
    list(_page_pool(nthreads).map(worker, range(nthreads)))
    return out


_pool_state = {}


def _page_pool(n):
    from concurrent.futures import ThreadPoolExecutor
    if _pool_state.get("n", 0) < n:
        _pool_state["pool"] = ThreadPoolExecutor(n, thread_name_prefix="ta-page")
        _pool_state["n"] = n
    return _pool_state["pool"]


def _page_streams(device, n):
    import torch
    have = _pool_state.setdefault(("streams", str(device)), [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=device))
    return have
