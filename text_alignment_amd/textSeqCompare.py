"""Affine-gap Needleman-Wunsch aligner on MI355X -- drop-in for the reference module of the
same name (reference textSeqCompare.py:1-177).

    from text_alignment_amd import textSeqCompare as tsc
    tra_align, ocr_align = tsc.perform_alignment(list(transcript), list(ocr), scoring_system)

`perform_alignment` keeps the reference's signature, return value and ValueError; the DP fill
and the traceback run in the HIP kernels of csrc/ta_nw.hip through the C ABI
(`ta_nw_batch`, include/text_alignment_amd.h).  `NWBatch` is the batched form behind it: many
independent problems (pages, or one page under many scoring systems as in the reference's
parameter grid search, evaluate_text_alignment.py:178-198) in one launch.
"""
import numpy as np
import torch

from . import _native

# module attributes of the reference (textSeqCompare.py:5-10)
default_match = 10
default_mismatch = -5
gap_open = -10
gap_extend = -1          # the boundary rows always use this one (textSeqCompare.py:54-59)
default_sys = [8, -4, -7, -7, -3, 0]

GAP = '_'


def parse_scoring_system(scoring_system):
    """The three accepted forms of textSeqCompare.py:24-42.

    Returns (params, fn): params = [match, mismatch, gap_open_x, gap_open_y, gap_extend_x,
    gap_extend_y]; fn is the caller's scoring callable for the 5-element form, else None.
    """
    if scoring_system is None:
        scoring_system = default_sys
    size = len(scoring_system)
    if size == 5 and callable(scoring_system[0]):
        return [0, 0] + [scoring_system[k] for k in range(1, 5)], scoring_system[0]
    if size == 6:
        return [scoring_system[k] for k in range(6)], None
    if size == 4:
        hit, miss, opn, ext = (scoring_system[k] for k in range(4))
        return [hit, miss, opn, opn, ext, ext], None
    raise ValueError('scoring_system {} invalid'.format(scoring_system))


def _is_integral(params):
    try:
        return all(float(v) == int(v) for v in params)
    except (TypeError, ValueError, OverflowError):
        return False


def encode_tokens(*seqs):
    """Dense int32 ids for arbitrary hashable tokens; equal tokens <=> equal ids
    (the aligner only ever compares tokens with ==, textSeqCompare.py:32).  Sequences of single characters -- what
    alignToOCR.py:273 passes, `list(transcript)` -- are encoded by their code points in one numpy pass; anything else
    (the bigrams of textSeqCompare.py:185-186, numbers, tuples) token by token through a dict."""
    seqs = [seq if isinstance(seq, (list, tuple, str)) else list(seq) for seq in seqs]    # an iterator is read ONCE
    try:
        if all(type(tok) is str for seq in seqs for tok in seq) and \
                all(len(seq) == 0 or max(map(len, seq)) == 1 == min(map(len, seq)) for seq in seqs):
            cps = [np.frombuffer("".join(seq).encode("utf-32-le"), dtype="<u4") for seq in seqs]
            alphabet, first = np.unique(np.concatenate(cps) if cps else np.zeros(0, "<u4"), return_index=True)
            # ids in order of first appearance, as the dict path numbers them
            rank = np.empty(len(alphabet), dtype=np.int32)
            rank[np.argsort(first, kind="stable")] = np.arange(len(alphabet), dtype=np.int32)
            out = [rank[np.searchsorted(alphabet, c)] if len(c) else np.zeros(0, np.int32) for c in cps]
            ids = {chr(int(c)): int(r) for c, r in zip(alphabet, rank)}
            return out, ids
    except (TypeError, ValueError):       # ValueError: UnicodeEncodeError -- a lone surrogate is a token like any other
        pass
    ids = {}
    out = []
    for seq in seqs:
        arr = np.empty(len(seq), dtype=np.int32)
        for k, tok in enumerate(seq):
            arr[k] = ids.setdefault(tok, len(ids))
        out.append(arr)
    return out, ids


def ops_to_alignment(ops, transcript, ocr):
    """Alignment columns (0 pair, 1 transcript token over a gap, 2 gap over an OCR token) ->
    the reference's two token lists with '_' gap markers (textSeqCompare.py:116-162).  On arrays: the token a column
    shows is the running count of the columns that consumed one."""
    ops = np.asarray(ops)
    if ops.size == 0:
        return [], []
    has_t, has_o = ops != 2, ops != 1
    t_obj = np.fromiter(transcript, dtype=object, count=len(transcript))
    o_obj = np.fromiter(ocr, dtype=object, count=len(ocr))
    tra = np.full(ops.size, GAP, dtype=object)
    oc = np.full(ops.size, GAP, dtype=object)
    tra[has_t] = t_obj[:int(has_t.sum())]
    oc[has_o] = o_obj[:int(has_o.sum())]
    return tra.tolist(), oc.tolist()


class NWBatch(object):
    """A batch of independent NW problems resident in HBM.

    t_list / o_list: per-problem int32 id arrays (host).  params: one scoring system
    (6 integers) or one per problem.  All device buffers are torch tensors on `device`;
    `run()` only enqueues kernels on torch's current stream.
    """

    def __init__(self, t_list, o_list, params, device="cuda", two_phase=None, wide=None):
        assert len(t_list) == len(o_list)
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.nprob = len(t_list)
        self.n = np.array([len(t) for t in t_list], dtype=np.int64)
        self.m = np.array([len(o) for o in o_list], dtype=np.int64)
        self.max_n = int(self.n.max()) if self.nprob else 0
        self.max_m = int(self.m.max()) if self.nprob else 0
        self.cells = int((self.n * self.m).sum())
        # two-phase aligner (score-only fill + chunked pointer re-derivation, csrc/ta_nw2.hip):
        # its fill is half as expensive per cell, its traceback costs ~60 us more per 256-row strip
        # of the tallest problem (one wave walks the strips one after the other), and its workspace
        # is 8x smaller.  Measured break-even on MI355X (tools/nw_breakeven.py): total cells
        # ~ 1.8e8 x strips, i.e. once the batch fills the chip about twice (re-measured after the
        # round-2 changes of both paths: tools/nw_breakeven.py).
        if two_phase is None:
            nstrips = (self.max_n + 255) // 256
            two_phase = (self.cells > 1.8e8 * nstrips) or (self.cells > 64e9) or \
                (self.max_m > _native.lib.ta_nw_max_m())       # wider than the one-pass kernel's LDS row
        self.two_phase = bool(two_phase)
        # one-pass launch shape: None = library default (a problem is spread over several
        # workgroups when the batch has fewer problems than the GPU has CUs), True / False force it
        self.wide = wide
        # two-phase launch-shape overrides (tests, A/B timing): phase 1 without the score profile;
        # phase 1 with exactly this many waves per workgroup (None = the library's own choice)
        self.no_profile = False
        self.waves = None
        # phase 2 launch shape (TA_NW_TBWAVES: 1, 2, 4 waves per problem; 3 half-strip pairs; 5, 6 pairs with 2, 4 waves;
        # None = the library's choice by batch size)
        self.tb_waves = None
        # one-pass fill: rows per lane (2 or 4; None = the library's choice by batch size)
        self.rows = None
        p = np.asarray(params, dtype=np.int64)
        if p.ndim == 1:
            p = p.reshape(1, 6)
        if p.shape[1] != 6 or p.shape[0] not in (1, self.nprob):
            raise ValueError("params must be 6 integers, or one row of 6 per problem")
        self.params_stride = 0 if p.shape[0] == 1 else 6
        pmax = int(np.abs(p).max()) if p.size else 0
        self.score_bound = (self.max_n + self.max_m + 2) * (3 * pmax + 2)
        if np.abs(p).max(initial=0) > 2 ** 20:
            raise OverflowError("scoring parameters too large for the integer kernels")
        if self.score_bound >= 2 ** 23:
            raise OverflowError("(n+m+2)*max|param| does not fit the 32-bit encoded scores")
        if self.max_m > (_native.lib.ta_nw2_max_m() if self.two_phase else _native.lib.ta_nw_max_m()):
            raise OverflowError("OCR string longer than the integer kernels take")

        lib = _native.lib
        t_off = np.zeros(self.nprob + 1, dtype=np.int64); np.cumsum(self.n, out=t_off[1:])
        o_off = np.zeros(self.nprob + 1, dtype=np.int64); np.cumsum(self.m, out=o_off[1:])
        ws_fn = lib.ta_nw2_workspace_bytes if self.two_phase else lib.ta_nw_workspace_bytes
        ws_sizes = np.array([ws_fn(int(a), int(b)) for a, b in zip(self.n, self.m)], dtype=np.int64)
        ws_off = np.zeros(self.nprob + 1, dtype=np.int64); np.cumsum(ws_sizes, out=ws_off[1:])
        self.ws_bytes = int(ws_off[-1])
        cap = self.n + self.m
        ops_off = np.zeros(self.nprob + 1, dtype=np.int64); np.cumsum(cap, out=ops_off[1:])
        self.ops_off_host = ops_off
        self.cap_host = cap

        cat_t = np.concatenate(t_list) if self.nprob and t_off[-1] else np.zeros(1, np.int32)
        cat_o = np.concatenate(o_list) if self.nprob and o_off[-1] else np.zeros(1, np.int32)
        if (cat_t.max(initial=0) >= 65535) or (cat_o.max(initial=0) >= 65535):
            raise OverflowError("more than 65534 distinct tokens in one batch")
        max_code = int(max(cat_t.max(initial=0), cat_o.max(initial=0)))
        self.codes8 = bool(max_code < 255)
        # hints for phase 1 of the two-phase aligner (include/text_alignment_amd.h): with a small
        # alphabet, byte-sized substitution scores and no free gap opens it keeps a score profile
        # in LDS instead of comparing token ids per cell
        self.hints = 0
        if p.size:
            sub = p[:, :2] - p[:, 4:5] - p[:, 5:6]
            if max_code < 254 and (p[:, 2:4] <= 0).all() and (np.abs(sub) <= 127).all():
                self.hints |= (max_code + 1) << _native.TA_NW_ALPHABET_SHIFT
            if (p[:, 2] == p[:, 3]).all():
                self.hints |= _native.TA_NW_OPENS_SAME
        (self.t_codes, self.o_codes, self.t_off, self.o_off, self.params, self.ws_off, self.ops_off) = _native.upload_packed(
            [cat_t.astype(np.int32), cat_o.astype(np.int32), t_off, o_off, p.astype(np.int32),
             ws_off[:-1].copy() if self.nprob else ws_off, ops_off[:-1].copy() if self.nprob else ops_off], self.device)
        self.ws = torch.empty(max(self.ws_bytes, 16), dtype=torch.uint8, device=self.device)
        self.ops = torch.empty(max(int(ops_off[-1]), 16), dtype=torch.uint8, device=self.device)
        self.ops_len = torch.zeros(max(self.nprob, 1), dtype=torch.int32, device=self.device)

    def run(self, fill=True, traceback=True):
        if self.nprob == 0:
            return
        flags = (_native.TA_NW_FILL if fill else 0) | (_native.TA_NW_TRACEBACK if traceback else 0)
        if self.codes8:
            flags |= _native.TA_NW_CODES8
        if self.two_phase:
            flags |= self.phase1_flags()
        if self.wide is not None and not self.two_phase:
            flags |= _native.TA_NW_WIDE if self.wide else _native.TA_NW_NARROW
        if self.rows and not self.two_phase:
            flags |= (int(self.rows) & 0x7) << _native.TA_NW_ROWS_SHIFT
        stream = torch.cuda.current_stream(self.device).cuda_stream
        entry = _native.lib.ta_nw2_batch if self.two_phase else _native.lib.ta_nw_batch
        rc = entry(
            self.t_codes.data_ptr(), self.t_off.data_ptr(),
            self.o_codes.data_ptr(), self.o_off.data_ptr(), self.nprob,
            self.params.data_ptr(), self.params_stride,
            self.ws.data_ptr(), self.ws_off.data_ptr(),
            self.ops.data_ptr(), self.ops_off.data_ptr(), self.ops_len.data_ptr(),
            self.max_n, self.max_m, self.score_bound, flags, stream)
        _native.check(rc, "ta_nw2_batch" if self.two_phase else "ta_nw_batch")

    def phase1_flags(self):
        """Hint and override bits of a ta_nw2_batch / ta_nw2_phase1_plan_batch call for this batch."""
        flags = self.hints
        if getattr(self, "check_ids", False):        # debug guard: the library verifies the id bounds the hints assert
            flags |= _native.TA_NW_CHECK_IDS
        if self.no_profile:
            flags |= _native.TA_NW_NO_PROFILE
        if self.waves:
            flags |= (int(self.waves) & 0xF) << _native.TA_NW_WAVES_SHIFT
        if self.tb_waves:
            flags |= (int(self.tb_waves) & 0x7) << _native.TA_NW_TBWAVES_SHIFT
        return flags

    def traceback_kernel(self):
        """name of the traceback kernel run() launches for this batch (bench / profile labels)"""
        if not self.two_phase:
            return "nw_traceback_kernel"
        w = _native.lib.ta_nw2_traceback_plan(self.nprob, self.params_stride, self.phase1_flags())
        return {1: "nw_trace2_kernel", 2: "nw_trace2w_kernel<2>", 4: "nw_trace2w_kernel<4>", 3: "nw_trace2h_kernel",
                5: "nw_trace2hw_kernel<2>", 6: "nw_trace2hw_kernel<4>"}[w]

    def fetch_begin(self):
        """start the download of the alignment columns (pinned buffers, an event on the current stream); `results()` then
        only waits for that event -- a caller with other host work puts it in between"""
        if self.nprob == 0:
            return
        self._host_ops = torch.empty(self.ops.shape, dtype=self.ops.dtype, pin_memory=True)
        self._host_len = torch.empty(self.ops_len.shape, dtype=self.ops_len.dtype, pin_memory=True)
        self._host_ops.copy_(self.ops, non_blocking=True)
        self._host_len.copy_(self.ops_len, non_blocking=True)
        self._fetched = torch.cuda.Event()
        self._fetched.record()

    def results(self):
        """Host copies of the alignment columns, one uint8 array per problem."""
        if self.nprob == 0:
            return []
        if getattr(self, "_fetched", None) is not None:
            self._fetched.synchronize()
            ops, lens = self._host_ops.numpy(), self._host_len.numpy()
            self._fetched = None
        else:
            ops = self.ops.cpu().numpy()
            lens = self.ops_len.cpu().numpy()
        if (lens[:self.nprob] < 0).any():         # ta_nw2_batch resets the lengths to -1 before its traceback launch
            raise RuntimeError("traceback of problem %d did not finish (a bounded wait between its waves ran out)"
                               % int(np.nonzero(lens[:self.nprob] < 0)[0][0]))
        out = []
        for k in range(self.nprob):
            end = int(self.ops_off_host[k] + self.cap_host[k])
            out.append(ops[end - int(lens[k]):end].copy())
        return out


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("text_alignment_amd needs an AMD GPU (MI355X): torch.cuda.is_available() "
                           "is False and there is no CPU fallback")


def perform_alignment_batch(pairs, scoring_systems=None):
    """Align many (transcript, ocr) token-list pairs in one launch.

    scoring_systems: None, one scoring system for all, or a list with one per pair (integer
    match/mismatch forms only).  Returns a list of (tra_align, ocr_align).
    """
    _require_gpu()
    pairs = [(list(t), list(o)) for t, o in pairs]
    if scoring_systems is None or not isinstance(scoring_systems, (list, tuple)) or \
            (len(scoring_systems) in (4, 6) and not isinstance(scoring_systems[0], (list, tuple, np.ndarray))) or \
            (len(scoring_systems) == 5 and callable(scoring_systems[0])):
        systems = [scoring_systems] * len(pairs)
    else:
        systems = list(scoring_systems)
        if len(systems) != len(pairs):
            raise ValueError("need one scoring system per pair")
    parsed = [parse_scoring_system(s) for s in systems]
    if any(fn is not None for _, fn in parsed):      # a scoring callable: its table is per pair
        return [perform_alignment(t, o, s) for (t, o), s in zip(pairs, systems)]
    t_list, o_list = [], []
    for t, o in pairs:
        (ti, oi), _ = encode_tokens(t, o)
        t_list.append(ti); o_list.append(oi)
    all_ops = None
    if all(_is_integral(p) for p, _ in parsed):
        params = np.array([[int(v) for v in p] for p, _ in parsed], dtype=np.int64)
        if len(pairs) and (params == params[0]).all():
            params = params[:1]
        try:
            batch = NWBatch(t_list, o_list, params)
            batch.run()
            all_ops = batch.results()
        except OverflowError:       # beyond the 32-bit kernels' limits: the float64 kernel below
            pass
    if all_ops is None:             # non-integral numbers (or too large): float64 kernel, still one launch
        from . import nw_general
        all_ops = nw_general.align_batch(t_list, o_list, [[float(v) for v in p] for p, _ in parsed])
    return [ops_to_alignment(ops, t, o) for ops, (t, o) in zip(all_ops, pairs)]


def perform_alignment(transcript, ocr, scoring_system=None, verbose=False):
    '''
    @scoring_system must be array-like, of one of the following forms:
    [match_func(a,b), gap_open_x, gap_open_y, gap_extend_x, gap_extend_y]
    [match, mismatch, gap_open_x, gap_open_y, gap_extend_x, gap_extend_y]
    [match, mismatch, gap_open, gap_extend]

    Returns (tra_align, ocr_align): two equal-length token lists with '_' where the other
    sequence has no partner (reference textSeqCompare.py:13, :177).  Inputs are not mutated.
    '''
    params, fn = parse_scoring_system(scoring_system)      # raises ValueError like the reference
    _require_gpu()
    transcript = list(transcript)
    ocr = list(ocr)
    (t_ids, o_ids), ids = encode_tokens(transcript, ocr)
    if fn is None and _is_integral(params):
        try:
            batch = NWBatch([t_ids], [o_ids], [int(v) for v in params])
            batch.run()
            ops = batch.results()[0]
        except OverflowError:       # scores would not fit the 32-bit encoding: float64 kernel
            ops = _general_alignment(t_ids, o_ids, ids, params, fn)
    else:
        ops = _general_alignment(t_ids, o_ids, ids, params, fn)
    tra_align, ocr_align = ops_to_alignment(ops, transcript, ocr)
    if verbose:
        for a, b in zip(tra_align, ocr_align):
            mark = ' ' if (a == GAP or b == GAP) else ('O' if a == b else '~')
            print('{} {} {}'.format(a, b, mark))
    return (tra_align, ocr_align)


def _general_alignment(t_ids, o_ids, ids, params, fn):
    """float64 / substitution-table kernel for callable or non-integral scoring systems."""
    from . import nw_general
    return nw_general.align(t_ids, o_ids, ids, params, fn)
