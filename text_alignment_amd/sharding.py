"""Page sharding across the GPUs of one node (one process per GPU, torch.distributed).

The reference processes pages in a plain serial loop (reference alignToOCR.py:407-438) and pages
share nothing, so ranks never exchange data while computing: rank r takes its share of the pages
(sorted by estimated cost, dealt round-robin), and the only collective is one variable-length
gather of syllable-box records to rank 0 at the end: every rank packs its int32 records into a
tensor of a capacity all ranks derive from the transcripts alone (no size exchange, no host
synchronisation), and one `gather` moves them (RCCL over xGMI when the backend is "nccl"; the same
code runs on "gloo" for the CPU tests).  The payload is ~150 records x 24 B per page, i.e.
latency-bound; there is deliberately no bucketing or overlap machinery.  (`gather_records`, the
two-step all-gather for callers that cannot bound their record count, is kept for tools.)
"""
import numpy as np
import torch

RECORD_FIELDS = 6        # page_id, syl_index, ulx, uly, lrx, lry


# ---------------------------------------------------------------------------------------------------------------
# Host placement of a rank.  A rank of the page-sharded job is HOST-bound as often as device-bound: per page it
# lays out ~8 MB of rows, decodes characters and assembles boxes while its GPU runs the previous chunk.  Eight ranks on a
# two-socket host that the scheduler is free to migrate share caches, memory channels and -- with rows staged through the
# copy pool -- cross the socket link twice per byte.  Each rank therefore pins itself (and thereby every thread it
# starts: the copy pool, torch's helper threads) to the cores of the NUMA node its GPU hangs off, an equal slice of them
# per rank when several ranks share the node.
def _cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_node(device_index, sysfs="/sys/bus/pci/devices"):
    """(numa node, cpus of that node) of HIP device `device_index`, from its PCI address; (None, None) if the host does
    not say (no sysfs entry, node -1: a single-node box or a VM that hides the topology)."""
    import ctypes
    import os
    from . import _native
    buf = ctypes.create_string_buffer(64)
    if _native.lib.ta_device_pci_bus_id(int(device_index), buf, 64) != 0:
        return None, None
    bdf = buf.value.decode("ascii", "replace").lower()
    try:
        with open(os.path.join(sysfs, bdf, "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return None, None
        with open(os.path.join(sysfs, bdf, "local_cpulist")) as f:
            cpus = _cpulist(f.read())
    except (OSError, ValueError):
        return None, None
    return node, cpus


def cpu_cores(cpus, sysfs="/sys/devices/system/cpu"):
    """cpu -> the lowest-numbered hardware thread of its physical core (thread_siblings_list); {} where sysfs is silent"""
    import os
    out = {}
    for c in cpus:
        try:
            with open(os.path.join(sysfs, "cpu%d" % c, "topology", "thread_siblings_list")) as f:
                out[c] = min(_cpulist(f.read()))
        except (OSError, ValueError):
            pass
    return out


def plan_binding(node_of_device, cpus_of_node, allowed, local_rank, local_world, min_slice=4, core_of=None):
    """The cpus rank `local_rank` binds to (pure function; tests/test_sharding.py): the cpus of its GPU's NUMA node that
    the process may use at all (`allowed`: the affinity mask it was started with, i.e. the container's share), cut into
    equal contiguous slices among the local ranks whose GPUs sit on the same node -- unless a slice would be smaller than
    `min_slice` cpus (the main thread, the copy pool, the runtime's helper threads), in which case the ranks of a node
    share all of it.  core_of (cpu -> id of its physical core, `cpu_cores()`): slices are cut along cores, SMT siblings
    together.  None: no binding (unknown topology, or nothing left after the intersection)."""
    node = node_of_device.get(local_rank)
    if node is None or node not in cpus_of_node:
        return None
    cpus = sorted(set(cpus_of_node[node]) & set(allowed))
    if not cpus:
        return None
    peers = sorted(r for r in range(local_world) if node_of_device.get(r) == node)
    share = len(cpus) // max(len(peers), 1)
    if len(peers) <= 1 or share < min_slice:
        return cpus
    # whole CORES per rank: the hardware threads of a core stay together (cpu 64 and its sibling 192 in one slice), so that
    # two ranks' main threads never time-share one core's pipelines
    core = core_of or {}
    cpus.sort(key=lambda c: (core.get(c, c), c))
    at = peers.index(local_rank)
    return sorted(cpus[at * share:(at + 1) * share])


def bind_to_gpu_node(device_index=None, local_rank=None, local_world=None, num_threads=1):
    """Bind THIS process to host cores next to its GPU and stop torch's intra-op pool from spreading (`num_threads`;
    the glue's numpy / torch calls are small and a pool of idle spinners per rank is exactly the contention to avoid).
    Call it before the first kernel launch of the rank -- threads started later inherit the mask.
    device_index: the rank's HIP device (default: torch's current one); local_rank / local_world: its place among the
    ranks of this host (default: LOCAL_RANK / LOCAL_WORLD_SIZE of the launcher, else 0 / 1).  Devices that do not exist
    (a rehearsal with more ranks than GPUs) count as sharing the device they wrap to.
    Returns a dict saying what was done: {"bound": bool, "numa_node", "cpus": "a-b,c", "ncpus", "reason"}; never raises
    for a host that cannot say or will not let us."""
    import os
    out = {"bound": False, "numa_node": None, "cpus": _fmt_cpus(os.sched_getaffinity(0)), "ncpus": len(os.sched_getaffinity(0)),
           "torch_threads": None, "reason": None}
    try:
        if num_threads:
            torch.set_num_threads(int(num_threads))
        out["torch_threads"] = torch.get_num_threads()
    except RuntimeError as exc:                      # (set after parallel work has started)
        out["reason"] = "torch.set_num_threads: %s" % exc
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
    ndev = torch.cuda.device_count()
    if ndev == 0:
        out["reason"] = "no GPU"
        return out
    if device_index is None:
        device_index = torch.cuda.current_device()
    node_of, cpus_of = {}, {}
    for r in range(local_world):
        dev = device_index if r == local_rank else r % ndev
        node, cpus = gpu_numa_node(dev)
        node_of[r] = node
        if node is not None:
            cpus_of[node] = cpus
    out["numa_node"] = node_of.get(local_rank)
    want = plan_binding(node_of, cpus_of, os.sched_getaffinity(0), local_rank, local_world,
                        core_of=cpu_cores(os.sched_getaffinity(0)))
    if want is None:
        out["reason"] = "the host does not name a NUMA node for this GPU (or none of its cpus is ours): affinity left as started"
        return out
    try:
        os.sched_setaffinity(0, want)
    except OSError as exc:
        out["reason"] = "sched_setaffinity refused: %s" % exc
        return out
    now = os.sched_getaffinity(0)
    out.update(bound=True, cpus=_fmt_cpus(now), ncpus=len(now))
    return out


def _fmt_cpus(cpus):
    """{0,1,2,3,8,9} -> '0-3,8-9'"""
    cpus = sorted(cpus)
    parts, a = [], None
    for k, c in enumerate(cpus):
        if a is None:
            a = c
        if k + 1 == len(cpus) or cpus[k + 1] != c + 1:
            parts.append("%d" % a if a == c else "%d-%d" % (a, c))
            a = None
    return ",".join(parts)


def page_cost(line_widths, n_transcript, n_ocr_estimate=None):
    """Work estimate of one page: LSTM timesteps + DP cells."""
    m = n_transcript if n_ocr_estimate is None else n_ocr_estimate
    return float(sum(int(w) + 32 for w in line_widths)) * 2.4e5 / 50.0 + float(n_transcript) * float(m)


def shard_indices(costs, world_size, rank):
    """Indices of the pages rank `rank` processes: heaviest first, dealt round-robin."""
    order = sorted(range(len(costs)), key=lambda k: (-costs[k], k))
    return order[rank::world_size]


def boxes_to_records(page_id, syl_boxes, syl_indices=None):
    """syl_boxes (CharBox list of one page) -> int32 [k, 6] records.  The syllable text is not
    shipped: syl_index is the box's index among the non-empty syllables of the transcript, and
    rank 0 recomputes the text with latinSyllabification.syllabify_text."""
    rec = np.zeros((len(syl_boxes), RECORD_FIELDS), dtype=np.int32)
    for k, b in enumerate(syl_boxes):
        idx = k if syl_indices is None else syl_indices[k]
        rec[k] = (page_id, idx, int(b.ul[0]), int(b.ul[1]), int(b.lr[0]), int(b.lr[1]))
    return rec


def gather_records(local, group=None, device=None):
    """Gather variable-length int32 [k, 6] record arrays to every rank (rank 0 uses them).

    Returns the concatenation in rank order as a numpy array.  With no process group
    initialised (single GPU) it returns `local` unchanged.
    """
    import torch.distributed as dist
    local = np.ascontiguousarray(local, dtype=np.int32).reshape(-1, RECORD_FIELDS)
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) \
            if dist.get_backend(group) == "nccl" else torch.device("cpu")
    count = torch.tensor([local.shape[0]], dtype=torch.int32, device=device)
    counts = torch.zeros(world, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(counts, count, group=group)
    counts_h = counts.cpu().numpy()
    cap = int(counts_h.max()) if world else 0
    if cap == 0:
        return np.zeros((0, RECORD_FIELDS), dtype=np.int32)
    padded = torch.zeros((cap, RECORD_FIELDS), dtype=torch.int32, device=device)
    if local.shape[0]:
        padded[:local.shape[0]] = torch.from_numpy(local).to(device)
    allrec = torch.zeros((world * cap, RECORD_FIELDS), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(allrec, padded, group=group)
    allrec_h = allrec.cpu().numpy().reshape(world, cap, RECORD_FIELDS)
    return np.concatenate([allrec_h[r, :int(counts_h[r])] for r in range(world)], axis=0)


def pack_records_device(local, cap, device, strict=True):
    """Host records -> a device tensor of fixed capacity [cap + 1, 6] whose row 0 carries the
    record count.  With a capacity agreed beforehand (pages per rank x an upper bound of boxes per
    page) the gather needs no size exchange and no host synchronisation.
    More records than the capacity: ValueError -- or, with strict=False (process_shard: a rank
    must still enter the collective), a buffer whose count word is MINUS the number of records it
    would have needed and that carries none; unpack_gathered raises on it, after the gather."""
    local = np.ascontiguousarray(local, dtype=np.int32).reshape(-1, RECORD_FIELDS)
    buf = np.zeros((cap + 1, RECORD_FIELDS), dtype=np.int32)
    if local.shape[0] > cap:
        if strict:
            raise ValueError("more records (%d) than the agreed capacity (%d)" % (local.shape[0], cap))
        buf[0, 0] = -local.shape[0]
    else:
        buf[0, 0] = local.shape[0]
        buf[1:1 + local.shape[0]] = local
    return torch.from_numpy(buf).to(device)


def gather_to_root(packed, group=None, dst=0, async_op=False):
    """The single collective of the path: gather every rank's packed record tensor
    ([cap + 1, 6] int32, see pack_records_device) to rank `dst`.  Returns (work, out): `out` is
    the [world, cap + 1, 6] tensor on `dst` (None elsewhere); with async_op=True the caller
    waits on `work` before reading it, so the gather overlaps whatever is enqueued next."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None, packed.unsqueeze(0)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    out, parts = None, None
    if rank == dst:
        out = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
        parts = [out[r] for r in range(world)]
    work = dist.gather(packed, parts, dst=dst, group=group, async_op=async_op)
    return work, out


def unpack_gathered(out):
    """[world, cap + 1, 6] gathered tensor -> concatenated host records in rank order."""
    h = out.cpu().numpy()
    over = [(r, -int(h[r, 0, 0])) for r in range(h.shape[0]) if int(h[r, 0, 0]) < 0]
    if over:
        raise ValueError("rank(s) %s produced more records than the agreed capacity of %d: %s"
                         % ([r for r, _ in over], h.shape[1] - 1, [c for _, c in over]))
    return np.concatenate([h[r, 1:1 + int(h[r, 0, 0])] for r in range(h.shape[0])], axis=0)


HEADER = -1              # syl_index of a page's header record
FAILED = -2              # syl_index of the status record of a page its rank could not process


def page_header(page_id, lines_peak_locs, nboxes):
    """One record per processed page: the float64 `median_line_spacing` of alignToOCR.to_JSON_dict
    (reference alignToOCR.py:338: the 75th percentile of the line gaps) as two int32 words, and the
    page's box count -- so that rank 0 can rebuild every page's JSON from the gathered records and
    the transcripts alone, pages without boxes included."""
    q = np.float64(np.quantile(np.diff(lines_peak_locs), 0.75))
    lo, hi = np.frombuffer(q.tobytes(), dtype=np.int32)
    return np.array([[page_id, HEADER, int(lo), int(hi), int(nboxes), 0]], dtype=np.int32)


def page_failed(page_id):
    """The status record a rank sends for a page it could not process (the reference's loop skips
    such a page: alignToOCR.py:240-243, :430-431): rank `dst` reports the page as None."""
    return np.array([[page_id, FAILED, 0, 0, 0, 0]], dtype=np.int32)


def record_capacity(transcript):
    """Upper bound of the records one page can emit, computable on every rank without talking:
    the header plus one box per non-empty syllable the pipeline can name -- counted with the
    syllabifier itself (words split on ' ' only, and a vowel-less word comes back whole, so a token
    made of tabs, newlines or no-break spaces IS a syllable and can get a box)."""
    from . import latinSyllabification as latsyl
    return 1 + sum(1 for s in latsyl.syllabify_text(transcript) if len(s) >= 1)


def records_to_json(records, transcripts, lines_peak_locs=None):
    """Rank-0 side: the gathered records -> {page_id: dict laid out as alignToOCR.to_JSON_dict}.
    transcripts[page_id] is the page's transcript string; syllable texts are recomputed here.
    The line spacing comes from the page's header record (or, for records without headers, from
    lines_peak_locs[page_id])."""
    from . import latinSyllabification as latsyl
    out = {}
    records = np.asarray(records).reshape(-1, RECORD_FIELDS)
    order = np.argsort(records[:, 0], kind="stable")
    records = records[order]
    bounds = np.flatnonzero(np.diff(records[:, 0])) + 1
    for rows in np.split(records, bounds) if len(records) else []:
        pid = int(rows[0, 0])
        if (rows[:, 1] == FAILED).any():           # the page's rank could not process it
            out[pid] = None
            continue
        head = rows[rows[:, 1] == HEADER]
        rows = rows[rows[:, 1] != HEADER]
        rows = rows[np.argsort(rows[:, 1], kind="stable")]
        if len(head):
            spacing = np.frombuffer(np.array(head[0, 2:4], dtype=np.int32).tobytes(), dtype=np.float64)[0]
            assert int(head[0, 4]) == len(rows), \
                "page %d: %d boxes announced, %d gathered" % (pid, head[0, 4], len(rows))
        else:
            spacing = np.quantile(np.diff(lines_peak_locs[pid]), 0.75)
        texts = [s for s in latsyl.syllabify_text(transcripts[pid]) if len(s) >= 1]
        out[pid] = {
            "median_line_spacing": spacing,
            "syl_boxes": [{"syl": texts[int(r[1])], "ul": [int(r[2]), int(r[3])],
                           "lr": [int(r[4]), int(r[5])]} for r in rows]}
    return out


def estimate_cost(pg, transcript):
    """page_cost from whatever the page object offers: strip widths of a PreparedPage, or the
    pixel count of a raw page image (about one text line per 140 rows in the reference's scans)."""
    strips = getattr(pg, "strips", None)
    if strips is not None:
        return page_cost([getattr(s, "width", 1000) for s in strips], len(transcript))
    px = np.asarray(getattr(pg, "pixels", pg))
    return page_cost([px.shape[1]] * max(1, px.shape[0] // 140), len(transcript))


def _world(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def shard_plan(costs, transcripts, world):
    """(shards, capacity): the page indices of every rank and the record capacity of the gather,
    both functions of the inputs alone, so every rank computes the same plan without talking."""
    shards = [shard_indices(costs, world, r) for r in range(world)]
    capacity = max([sum(record_capacity(transcripts[k]) for k in sh) for sh in shards] + [1])
    return shards, capacity


def _page_errors():
    """What a page's DATA can raise in alignToOCR.process_batch: a text line too long for the recogniser
    (ocr.RecognitionError, the reference's 'OCRopus failed'), an empty / constant / mistyped strip (ValueError,
    TypeError: page.raw_strip_pixels, lineest_gpu), a transcript / OCR length mismatch (the reference's assert,
    alignToOCR.py:291) and a syllable the search cannot place (AttributeError, alignToOCR.py:307)."""
    from . import ocr
    return (ocr.RecognitionError, ValueError, TypeError, AssertionError, AttributeError)


def process_shard(my_pages, my_transcripts, my_ids, my_models, capacity, seq_align_params=None,
                  group=None, dst=0, device=None, timings=None):
    """This rank's pages (global indices `my_ids`) through alignToOCR.process_batch, one batch per
    distinct recogniser model, then THE collective of the path: one fixed-capacity gather of the
    box records to rank `dst`.  `capacity` must be the same on every rank (shard_plan).  Returns
    the concatenated records on `dst`, None elsewhere.  timings (a dict, optional) receives this rank's
    wall seconds of the two parts: "pages_s" (its share, results on the host) and "gather_s" (packing, the
    collective, unpacking on `dst` -- including the wait for the slowest rank to arrive), and "device_wait_s": the part
    of pages_s this thread spent waiting for its GPU (pages_s minus it = the share's host work)."""
    from . import alignToOCR as atocr
    import time
    import torch.distributed as dist
    import warnings
    t_start, w_start = time.perf_counter(), atocr.WAIT_SECONDS[0]
    recs = []
    by_model = {}
    for k, mdl in enumerate(my_models):
        by_model.setdefault(id(mdl), (mdl, []))[1].append(k)

    def run(mdl, ks):
        """records of pages ks through one process_batch (mdl: one model, or a list with one per page of ks)"""
        out, idx, arrs = [], [], []
        res = atocr.process_batch([my_pages[k] for k in ks], [my_transcripts[k] for k in ks], mdl,
                                  seq_align_params, indices_out=idx, arrays_out=arrs)
        for j, (k, r, ix) in enumerate(zip(ks, res, idx)):
            out.append(page_header(my_ids[k], r[2], len(r[0])))
            if j < len(arrs):                     # the array pipeline hands the boxes over as they are
                rec = np.empty((len(ix), RECORD_FIELDS), dtype=np.int32)
                rec[:, 0] = my_ids[k]
                rec[:, 1] = ix
                rec[:, 2:6] = arrs[j]
                out.append(rec)
            else:
                out.append(boxes_to_records(my_ids[k], r[0], ix))
        return out

    # Whatever happens on this rank, it must reach the collective: the other ranks are (or will be)
    # waiting in it.  A batch that fails ON ITS DATA is retried page by page, and a page that still
    # fails is reported by a status record instead of boxes -- one bad page (a blank strip, an over-long
    # line, a syllable the search cannot place) costs that page, as in the reference's loop
    # (alignToOCR.py:240-243: 'OCRopus failed! Skipping current file.'), not the rank's share.
    # Anything else -- a failed native call (kernel fault, sticky HIP error, out of memory), a torch
    # RuntimeError, a programming error -- is NOT a page to skip: the rank remembers it, reports its
    # remaining pages as failed so that rank `dst` sees every page, still enters the gather IF IT CAN, and
    # re-raises afterwards, so the process ends non-zero instead of returning 'skipped pages'.  "If it can":
    # after a HOST-side failure (a bad argument, an exception in the glue) the gather works and nobody waits; after a
    # DEVICE fault under nccl the packing or the collective itself raises on the poisoned context -- the first
    # error is then raised at once, the process exits non-zero and the launcher (torchrun) tears the group down.
    page_errors = _page_errors()
    fatal = None

    def is_page_error(exc):
        from . import _native
        return isinstance(exc, page_errors) and not isinstance(exc, _native.NativeArgumentError)
    # first the whole share in ONE call -- process_batch groups the pages by model itself and runs its chunk pipeline on
    # across the groups (two models x 32 pages: 1 280 -> ~1 500 pages/s); only if that fails, model by model, then page by page
    if len(by_model) > 1:
        try:
            every = [k for _, ks in by_model.values() for k in ks]
            recs += run([my_models[k] for k in every], every)
            by_model = {}
        except Exception as exc:                  # noqa: BLE001 -- sorted out below, batch by batch
            if not is_page_error(exc):
                fatal = exc
    for mdl, ks in by_model.values():
        if fatal is not None:
            recs += [page_failed(my_ids[k]) for k in ks]
            continue
        try:
            recs += run(mdl, ks)
        except Exception as exc:                  # noqa: BLE001 -- sorted into page / fatal just below
            if not is_page_error(exc):
                fatal = exc
                recs += [page_failed(my_ids[k]) for k in ks]
                continue
            for k in ks:
                if fatal is not None:
                    recs.append(page_failed(my_ids[k]))
                    continue
                try:
                    recs += run(mdl, [k])
                except Exception as exc1:         # noqa: BLE001
                    if is_page_error(exc1):
                        warnings.warn("page %d failed on this rank and is skipped: %r" % (my_ids[k], exc1))
                    else:
                        fatal = exc1
                    recs.append(page_failed(my_ids[k]))
    local = np.concatenate(recs, axis=0) if recs else np.zeros((0, RECORD_FIELDS), np.int32)
    t_pages = time.perf_counter()
    if device is None:
        nccl = dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"
        device = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    try:
        work, out = gather_to_root(pack_records_device(local, capacity, device, strict=False), group=group, dst=dst)
    except Exception:                             # noqa: BLE001
        if fatal is not None:                     # the device is gone: report the FIRST failure, promptly
            raise fatal
        raise
    if fatal is not None:                         # after the collective: nobody is left waiting
        raise fatal
    if local.shape[0] > capacity:
        raise ValueError("more records (%d) than the agreed capacity (%d)" % (local.shape[0], capacity))
    res = None if out is None else unpack_gathered(out)
    if timings is not None:
        if out is None and device.type == "cuda":
            torch.cuda.current_stream(device).synchronize()      # (a sender's part of the gather is over when its stream is)
        timings["pages_s"] = t_pages - t_start
        timings["device_wait_s"] = atocr.WAIT_SECONDS[0] - w_start
        timings["gather_s"] = time.perf_counter() - t_pages
    return res


def process_pages(pages, transcripts, ocropus_model, seq_align_params=None, group=None, dst=0):
    """The reference's page loop (alignToOCR.py:407-438) sharded over the ranks of the process
    group: rank r runs alignToOCR.process_batch on its share of `pages` (heaviest first, dealt
    round-robin), then one gather of syllable-box records to rank `dst`.  `ocropus_model` is one
    model for all pages or a list with one per page (the reference's two manuscripts have one
    each, alignToOCR.py:390-405).  Returns {page index: JSON dict as alignToOCR.to_JSON_dict} on
    rank `dst` (every page present, pages without boxes with an empty list, pages that failed on
    their rank -- skipped, as the reference's loop skips them -- with None), None on other ranks."""
    world, rank = _world(group)
    models = list(ocropus_model) if isinstance(ocropus_model, (list, tuple)) else [ocropus_model] * len(pages)
    if len(models) != len(pages) or len(transcripts) != len(pages):
        raise ValueError("need one transcript (and, if a list is given, one model) per page")
    costs = [estimate_cost(pg, tr) for pg, tr in zip(pages, transcripts)]
    shards, capacity = shard_plan(costs, transcripts, world)
    mine = shards[rank]
    allrec = process_shard([pages[k] for k in mine], [transcripts[k] for k in mine], mine,
                           [models[k] for k in mine], capacity, seq_align_params, group, dst)
    if allrec is None:
        return None
    out = records_to_json(allrec, {k: transcripts[k] for k in range(len(pages))})
    assert sorted(out) == list(range(len(pages))), "a page's header record is missing from the gather"
    return out
