"""End-to-end timing of alignToOCR.process_batch on synthetic pages (BASELINE configs[2]/[4]
shape): 30 strips per page of 800-2000 columns, ~1200-character transcripts.

    python tools/pages_bench.py [npages] [--profile]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

VOCAB = ("dominus deus meus alleluia gloria patri et filio spiritui sancto sicut erat in principio "
         "nunc semper saecula saeculorum amen laudate eum omnes gentes quoniam confirmata est "
         "super nos misericordia eius veritas manet aeternum").split()
PARAMS = [8, -1, -9, -9, -4, -4]     # cheap mismatches: a random-weight model's text still pairs up


def page_meta(seed, nlines=30):
    """(strip widths, transcript) of synthetic page `seed` -- all a rank needs to cost a page and to
    size the gather, without building its pixels (sharding.shard_plan)."""
    rng = np.random.default_rng([seed, 0])
    widths = [int(w) for w in rng.integers(800, 2001, size=nlines)]
    tr = " ".join(VOCAB[int(i)] for i in rng.integers(0, len(VOCAB), size=180))
    return widths, tr


def make_page(seed, nlines=30, raw=False, block=None):
    """raw = False: strips carry already-normalised (T, 48) rows (SURVEY 8d's OCR input) -- as pageable numpy arrays,
    or, with `block` (a page.RowBlock, page-locked or device memory), as spans of that block (same values);
    raw = True: strips are 60-row uint8 images as the page cutter saves them (device normaliser)."""
    from text_alignment_amd import page as page_mod
    widths, tr = page_meta(seed, nlines)
    rng = np.random.default_rng([seed, 1])
    strips = []
    for k, w in enumerate(widths):
        if raw:
            yy = np.arange(60)[:, None]
            dens = 0.6 * np.exp(-0.5 * ((yy - 30.0) / 8.0) ** 2)
            px = np.where(rng.random((60, w)) < dens, 0, 255).astype(np.uint8)
            strips.append(page_mod.Strip(40, 100 + 120 * k, 60, width=w, pixels=px))
            continue
        xs = np.zeros((w + 32, 48), dtype=np.float32)
        xs[16:16 + w] = (rng.random((w, 48), dtype=np.float32) < 0.15) * rng.random((w, 48), dtype=np.float32)
        if block is not None:
            sp = block.take(w + 32)
            if block.kind == "pinned":
                block.host[sp.start:sp.stop] = xs
            else:
                import torch
                block.tensor[sp.start:sp.stop] = torch.from_numpy(xs).to(block.tensor.device)
            xs = sp
        strips.append(page_mod.Strip(40, 100 + 120 * k, 60, width=2 * w, prepared=xs))
    peaks = [130 + 120 * k for k in range(nlines + 1)]
    return page_mod.PreparedPage((2200, 3300), (2200, 3300), 0, strips, peaks), tr


def make_pages_in_blocks(seeds, kind="pinned", pages_per_block=16, nlines=30):
    """make_page for every seed, the rows of `pages_per_block` consecutive pages in one RowBlock of `kind`;
    returns (pages, transcripts, blocks)"""
    from text_alignment_amd import page as page_mod
    pages, trs, blocks = [], [], []
    for a in range(0, len(seeds), pages_per_block):
        part = seeds[a:a + pages_per_block]
        rows = sum(sum(w + 32 for w in page_meta(sd, nlines)[0]) for sd in part)
        block = page_mod.RowBlock(rows, kind=kind)
        blocks.append(block)
        for sd in part:
            pg, tr = make_page(sd, nlines, block=block)
            pages.append(pg)
            trs.append(tr)
    return pages, trs, blocks


def make_page_image(seed, nlines=30):
    """a whole text-layer page as a uint8 image (white, rows of word-like ink blobs, slightly
    skewed): the input of the complete pipeline -- preprocessing, line finding, normaliser,
    recogniser, aligner -- all on the device"""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    h, w = 200 + 140 * nlines, 1400
    ink = np.zeros((h, w), dtype=bool)
    yy = np.arange(h)[:, None]
    for k in range(nlines):
        cy = 150 + 140 * k
        dens = 0.55 * np.exp(-0.5 * ((yy - cy) / 8.0) ** 2)
        x = 80
        while x < w - 160:
            ww = int(rng.integers(50, 120))
            ink[:, x:x + ww] |= rng.random((h, ww)) < dens
            x += ww + int(rng.integers(24, 40))
    ink = ndimage.binary_closing(ink, structure=np.ones((3, 3), bool), iterations=2)
    img = np.where(ink, 0, 255).astype(np.uint8)
    img = ndimage.rotate(img, float(rng.uniform(-2, 2)), reshape=False, order=1, mode='constant', cval=255)
    return img.astype(np.uint8)


class RawPage(object):
    def __init__(self, px):
        from text_alignment_amd.page import Image
        self.pixels = px
        self.dim = Image(px.shape[1], px.shape[0]).dim


def make_recognizer(seed=7001, no=40, precision=None):
    from text_alignment_amd import ocr
    model = ocr.LineModel.random(seed, no=no)
    model.W2[0, 0] += 4.0            # favour blanks: many short runs -> many characters
    return ocr.LineRecognizer(model, precision=precision or ocr.DEFAULT_PRECISION)


ROWS_INPUT = {
    "numpy": "prepared rows as pageable numpy arrays, one per strip (copied into page-locked staging by the copy pool)",
    "pinned": "prepared rows in page-locked RowBlocks of 16 pages (text_alignment_amd.page.RowBlock): one DMA transfer per "
              "block and chunk straight from where the loader wrote them, lines permuted on the device (csrc/ta_rows.hip)",
    "device": "prepared rows in device-resident RowBlocks of 16 pages: nothing crosses PCIe"}


def setup_sharded(pages_per_rank, rank, world, seed0=100, rows="pinned"):
    """BASELINE configs[4]: `pages_per_rank` x world synthetic pages, half read with a
    Salzinnes-shaped model (96 classes), half with a St-Gall-shaped one (64), sharded over the
    ranks by sharding.shard_plan.  Every rank builds only its own pages (the plan needs strip
    widths and transcripts only).  rows: where the strips' prepared rows lie (ROWS_INPUT).
    Returns the arguments of sharding.process_shard + the transcripts."""
    import torch
    from text_alignment_amd import sharding
    total = pages_per_rank * world
    metas = [page_meta(seed0 + k) for k in range(total)]
    transcripts = [m[1] for m in metas]
    costs = [sharding.page_cost(m[0], len(m[1])) for m in metas]
    shards, capacity = sharding.shard_plan(costs, transcripts, world)
    mine = shards[rank]
    recs = [make_recognizer(7001, 96), make_recognizer(7002, 64)]
    if rows == "numpy":
        pages, blocks = [make_page(seed0 + k)[0] for k in mine], []
    else:
        pages, _, blocks = make_pages_in_blocks([seed0 + k for k in mine], rows)
    job = {"pages": pages, "transcripts": [transcripts[k] for k in mine], "blocks": blocks,
           "ids": mine, "models": [recs[k % 2] for k in mine], "capacity": capacity,
           "all_transcripts": transcripts, "total_pages": total, "rows": rows, "input": ROWS_INPUT[rows]}
    for _ in range(3):                    # warm-up at full size (collective: every rank); the first calls of a process
        sharding.process_shard(job["pages"], job["transcripts"], mine, job["models"],      # pay for stream / queue
                               capacity, PARAMS)                                           # creation and allocator growth
    torch.cuda.synchronize()
    return job


def run_sharded(job, timings=None):
    """one timed pass: this rank's share through process_batch (one batch per model) and the single
    gather; returns the gathered records on rank 0, None elsewhere"""
    from text_alignment_amd import sharding
    return sharding.process_shard(job["pages"], job["transcripts"], job["ids"], job["models"],
                                  job["capacity"], PARAMS, timings=timings)


def _device_busy_ms(fn):
    """device time of one call of fn (kernels + copies of every stream it uses) from a torch profiler trace:
    the union of the device intervals -- the GPU-busy time of a pass"""
    import torch
    from torch.profiler import profile, ProfilerActivity
    try:
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            fn()
            torch.cuda.synchronize()
        iv = sorted((e.time_range.start, e.time_range.end) for e in prof.events()
                    if getattr(e, "device_type", None) is not None and "CUDA" in str(e.device_type))
        busy, cur_a, cur_b = 0.0, None, None
        for a, b in iv:
            if cur_b is None or a > cur_b:
                if cur_b is not None:
                    busy += cur_b - cur_a
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        if cur_b is not None:
            busy += cur_b - cur_a
        return busy / 1e3 if busy > 0 else None
    except Exception:                                   # (no profiler support: the figure is simply absent)
        return None


def run(npages, seed0=100):
    import threading
    import torch
    from text_alignment_amd import alignToOCR as atocr, ocr
    rec = make_recognizer()
    pages, trs = zip(*[make_page(seed0 + k) for k in range(npages)])

    import gc

    def settle():
        """before a timed loop: everything alive so far (pages, blocks, recognisers -- hundreds of thousands of objects) out
        of the collector's way, so that a full collection in mid-pass costs what the pass itself allocated, not 20 ms"""
        gc.collect()
        gc.freeze()

    def median_of(n, pages_, trs_, rec_=None):
        """median of n passes (median of >= 10), each a whole process_batch call incl. the final
        synchronize; the host side of a pass -- numpy, uploads from pageable memory -- varies by +-20 % from call to
        call on a shared box, the device work does not.  Also the host CPU seconds of the median pass's neighbours
        (time.process_time: all threads of this process)."""
        ts, cpu, out = [], [], None
        for _ in range(n):
            c0, t0 = time.process_time(), time.perf_counter()
            out = atocr.process_batch(pages_, trs_, rec_ or rec, PARAMS)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            cpu.append(time.process_time() - c0)
        return float(np.median(ts)), out, float(np.median(cpu))
    for _ in range(3):                       # warm-up at full size (staging buffers, streams, allocator)
        atocr.process_batch(list(pages), list(trs), rec, PARAMS)
    torch.cuda.synchronize()
    settle()
    dt, res, cpu_s = median_of(10, list(pages), list(trs))
    busy_ms = _device_busy_ms(lambda: atocr.process_batch(list(pages), list(trs), rec, PARAMS))
    # the same pages with their rows where a GPU-side loader puts them: page-locked blocks (no host copy), device blocks
    from tools import pages_check
    in_place = {}
    for kind in ("pinned", "device"):
        bpages, btrs, blocks = make_pages_in_blocks([seed0 + k for k in range(npages)], kind)
        for _ in range(3):
            atocr.process_batch(bpages, btrs, rec, PARAMS)
        torch.cuda.synchronize()
        settle()
        bdt, bres, bcpu = median_of(10, bpages, btrs)
        bbusy = _device_busy_ms(lambda: atocr.process_batch(bpages, btrs, rec, PARAMS))
        in_place[kind] = {"input": ROWS_INPUT[kind], "pages_per_s": npages / bdt, "seconds": bdt,
                          "host": {"cpu_ms_per_page": 1e3 * bcpu / npages, "cpu_over_wall": bcpu / bdt,
                                   "gpu_busy_ms_per_pass": bbusy, "gpu_busy_frac": (bbusy * 1e-3 / bdt) if bbusy else None},
                          "equal_to_numpy_input": [atocr.to_JSON_dict(r[0], r[2]) for r in bres] ==
                                                  [atocr.to_JSON_dict(r[0], r[2]) for r in res]}
        if kind == "pinned":
            in_place[kind].update(pages_check.check_pages(
                [atocr.to_JSON_dict(r[0], r[2]) for r in bres], bpages, btrs, [rec.model] * npages, PARAMS, [1, npages - 1]))
        del bpages, bres, blocks
    checked = pages_check.check_pages([atocr.to_JSON_dict(r[0], r[2]) for r in res], pages, trs, [rec.model] * npages,
                                      PARAMS, [0, npages // 2])
    # BASELINE configs[2]: one page end to end (30 strips + one NW problem), latency of a lone call
    gc.collect()                               # a generation-2 collection in mid-call costs ~20 ms

    def single_page(rec_):
        lat = []
        for k in range(5):
            t1 = time.perf_counter()
            atocr.process_batch([pages[k]], [trs[k]], rec_, PARAMS)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t1)
        return 1e3 * sorted(lat)[2]
    lat_ms = single_page(rec)
    # the opt-in exact-f32 mode on the same pages (the default is float64 since round 5)
    other = None
    if rec.mode != 0:
        rec32 = make_recognizer(precision="f32")
        for _ in range(2):
            atocr.process_batch(list(pages), list(trs), rec32, PARAMS)
        torch.cuda.synchronize()
        dt32, _, cpu32 = median_of(10, list(pages), list(trs), rec32)
        other = {"pages_per_s": npages / dt32, "seconds": dt32, "single_page_ms": single_page(rec32),
                 "host_cpu_ms_per_page": 1e3 * cpu32 / npages,
                 "note": "LineRecognizer(model, precision='f32'): opt-in fast mode, same pages"}
        del rec32
    # the same from raw strips: line normaliser on the device in front of the recogniser
    rpages, rtrs = zip(*[make_page(seed0 + 5000 + k, raw=True) for k in range(npages)])
    for _ in range(2):
        atocr.process_batch(list(rpages), list(rtrs), rec, PARAMS)
    torch.cuda.synchronize()
    raw_dt, _, raw_cpu = median_of(10, list(rpages), list(rtrs))
    # and from whole page images: preprocessing and line finding on the device as well
    nimg = npages
    ipages = [RawPage(make_page_image(seed0 + 9000 + k)) for k in range(nimg)]
    itrs = list(trs[:nimg])
    atocr.process_batch(ipages, itrs, rec, PARAMS)         # warm: the page planes come out of torch's caching allocator
    torch.cuda.synchronize()
    img_dt, _, img_cpu = median_of(10, ipages, itrs)
    img_where = {}
    for kind, move in (("pinned_pages", lambda a: torch.from_numpy(a).pin_memory()), ("device_pages", lambda a: torch.from_numpy(a).cuda())):
        tpages = [RawPage(move(pg.pixels)) for pg in ipages]                 # the same pages as uint8 tensors, taken where they lie
        atocr.process_batch(tpages, itrs, rec, PARAMS)
        torch.cuda.synchronize()
        t_dt, _, _ = median_of(10, tpages, itrs)
        img_where[kind] = {"pages_per_s": nimg / t_dt, "seconds": t_dt}
        del tpages
    img_lat = []
    for k in range(5):                                     # the reference's own call shape: process(raw_image, transcript, model)
        t1 = time.perf_counter()
        atocr.process_batch([ipages[k]], [itrs[k]], rec, PARAMS)
        torch.cuda.synchronize()
        img_lat.append(time.perf_counter() - t1)
    out = {"pages": npages, "precision": {0: "f32", 1: "split", 3: "f64"}[rec.mode], "seconds": dt,
           "input": ROWS_INPUT["numpy"],
           "pages_per_s": npages / dt, "lines_per_s": npages * 30 / dt,
           "pinned_rows": in_place["pinned"], "device_rows": in_place["device"],
           "single_page_ms": lat_ms,
           "host": {"cpu_ms_per_page": 1e3 * cpu_s / npages, "cpu_s_per_pass": cpu_s,
                    "cpu_over_wall": cpu_s / dt, "threads_alive": threading.active_count(),
                    "copy_pool_threads": ocr.COPY_THREADS if ocr._pool is not None else 0,
                    "gpu_busy_ms_per_pass": busy_ms, "gpu_busy_frac": (busy_ms * 1e-3 / dt) if busy_ms else None,
                    "pipeline_chunk_pages": atocr.PIPELINE_CHUNK_PAGES,
                    "note": "cpu: time.process_time over the pass (all threads of the process: the interpreter's one "
                            "thread + the strip-copy pool); gpu_busy: union of device intervals of one pass (torch profiler)"},
           "page_images": {"pages": nimg, "pages_per_s": nimg / img_dt, "seconds": img_dt,
                           "host_cpu_ms_per_page": 1e3 * img_cpu / nimg, "single_page_ms": 1e3 * sorted(img_lat)[2],
                           "input": "pageable numpy arrays (staged through one page-locked buffer per batch of pages)",
                           "pinned_pages": img_where["pinned_pages"], "device_pages": img_where["device_pages"],
                           "note": "4400 x 1400 uint8 page images: csrc/ta_preproc.hip + ta_lineest.hip in front"},
           "raw_strips": {"pages_per_s": npages / raw_dt, "seconds": raw_dt, "host_cpu_ms_per_page": 1e3 * raw_cpu / npages,
                          "note": "strips as 60-row uint8 images, normalised by csrc/ta_lineest.hip"},
           "syllable_boxes": sum(len(r[0]) for r in res),
           "pages_checked": checked["pages_checked"], "pages_equal_to_oracle": checked["pages_equal_to_oracle"],
           "check": checked,
           "timing": "median of 10 passes after warm-up, each a whole process_batch call incl. the final synchronize",
           "note": "process_batch end to end: host upload + K3/K4/K5 + NW + host glue (numpy arrays, page_batch.py), "
                   "chunks of pages pipelined (host stage of one chunk under the device stage of the next)"}
    if other is not None:
        out["f32_mode"] = other
    return out


if __name__ == "__main__":
    from tools import switches
    switches.apply()             # TA_* environment variables -> the product modules' attributes
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
    print(run(n))
    if "--profile" in sys.argv:
        import cProfile
        import pstats
        from text_alignment_amd import alignToOCR as atocr
        rec = make_recognizer()
        pages, trs = zip(*[make_page(100 + k) for k in range(n)])
        atocr.process_batch(list(pages), list(trs), rec, PARAMS)
        pr = cProfile.Profile()
        pr.enable()
        atocr.process_batch(list(pages), list(trs), rec, PARAMS)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(30)
