"""Page sharding across the GPUs of one node (one process per GPU, torch.distributed).

The reference processes pages in a plain serial loop (reference alignToOCR.py:407-438) and pages
share nothing, so ranks never exchange data while computing: rank r takes its share of the pages
(sorted by estimated cost, dealt round-robin), and the only collective is one variable-length
gather of syllable-box records to rank 0 at the end -- an all-gather of the per-rank record
counts followed by one all-gather of the padded int32 records (RCCL over xGMI when the backend
is "nccl"; the same code runs on "gloo" for the CPU tests).  The payload is ~150 records x 24 B
per page, i.e. latency-bound; there is deliberately no bucketing or overlap machinery.
"""
import numpy as np
import torch

RECORD_FIELDS = 6        # page_id, syl_index, ulx, uly, lrx, lry


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


def pack_records_device(local, cap, device):
    """Host records -> a device tensor of fixed capacity [cap + 1, 6] whose row 0 carries the
    record count.  With a capacity agreed beforehand (pages per rank x an upper bound of boxes per
    page) the gather needs no size exchange and no host synchronisation."""
    local = np.ascontiguousarray(local, dtype=np.int32).reshape(-1, RECORD_FIELDS)
    if local.shape[0] > cap:
        raise ValueError("more records (%d) than the agreed capacity (%d)" % (local.shape[0], cap))
    buf = np.zeros((cap + 1, RECORD_FIELDS), dtype=np.int32)
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
    return np.concatenate([h[r, 1:1 + int(h[r, 0, 0])] for r in range(h.shape[0])], axis=0)


def records_to_json(records, transcripts, lines_peak_locs):
    """Rank-0 side: the gathered records -> {page_id: dict laid out as alignToOCR.to_JSON_dict}.
    transcripts[page_id] is the page's transcript string; syllable texts are recomputed here."""
    from . import latinSyllabification as latsyl
    out = {}
    records = np.asarray(records).reshape(-1, RECORD_FIELDS)
    for pid in sorted(set(int(r[0]) for r in records)):
        rows = records[records[:, 0] == pid]
        rows = rows[np.argsort(rows[:, 1], kind="stable")]
        texts = [s for s in latsyl.syllabify_text(transcripts[pid]) if len(s) >= 1]
        out[pid] = {
            "median_line_spacing": np.quantile(np.diff(lines_peak_locs[pid]), 0.75),
            "syl_boxes": [{"syl": texts[int(r[1])], "ul": [int(r[2]), int(r[3])],
                           "lr": [int(r[4]), int(r[5])]} for r in rows]}
    return out


def process_pages(pages, transcripts, ocropus_model, seq_align_params=None, group=None):
    """Shard `pages` over the ranks of the process group, run alignToOCR.process_batch on this
    rank's share, gather the syllable boxes.  Returns {page index: JSON dict} on every rank
    (rank 0 is the consumer); page indices with no boxes map to an empty syl_boxes list."""
    import torch.distributed as dist
    from . import alignToOCR as atocr
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    costs = [page_cost([getattr(s, "width", 1000) for s in pg.strips], len(tr))
             for pg, tr in zip(pages, transcripts)]
    mine = shard_indices(costs, world, rank)
    idx = []
    res = atocr.process_batch([pages[k] for k in mine], [transcripts[k] for k in mine],
                              ocropus_model, seq_align_params, indices_out=idx)
    recs = [boxes_to_records(pid, r[0], ix) for pid, r, ix in zip(mine, res, idx)]
    local = np.concatenate(recs, axis=0) if recs else np.zeros((0, RECORD_FIELDS), np.int32)
    allrec = gather_records(local, group)
    peaks = {k: pages[k].lines_peak_locs for k in range(len(pages))}
    out = records_to_json(allrec, {k: transcripts[k] for k in range(len(pages))}, peaks)
    for k in range(len(pages)):
        if k not in out:
            out[k] = {"median_line_spacing": np.quantile(np.diff(peaks[k]), 0.75), "syl_boxes": []}
    return out
