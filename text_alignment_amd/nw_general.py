"""Host side of ta_nw_general (csrc/ta_nw_general.hip): callable or non-integral scoring
systems (reference textSeqCompare.py:27-29 and the float case of :30-40)."""
import numpy as np
import torch

from . import _native


def align(t_ids, o_ids, ids, params, fn):
    """Returns the alignment columns (uint8 array) for one problem."""
    dev = torch.device("cuda")
    n, m = len(t_ids), len(o_ids)
    table_d = None
    tm = 0
    if fn is not None:
        toks = [None] * len(ids)
        for tok, k in ids.items():
            toks[k] = tok
        tm = len(ids)
        table = np.zeros((tm, tm), dtype=np.float64)
        for a in sorted(set(int(v) for v in t_ids)):
            for b in sorted(set(int(v) for v in o_ids)):
                table[a, b] = fn(toks[a], toks[b])      # textSeqCompare.py:67
        table_d = torch.from_numpy(table).to(dev)
    lib = _native.lib
    p_d = torch.tensor([float(v) for v in params], dtype=torch.float64, device=dev)
    t_d = torch.from_numpy(np.ascontiguousarray(t_ids, dtype=np.int32)).to(dev) if n else None
    o_d = torch.from_numpy(np.ascontiguousarray(o_ids, dtype=np.int32)).to(dev) if m else None
    sc = torch.empty(max(lib.ta_nw_general_score_bytes(n) // 8, 1), dtype=torch.float64, device=dev)
    ptr = torch.empty(max(lib.ta_nw_general_ptr_bytes(n, m), 1), dtype=torch.uint8, device=dev)
    ops = torch.empty(max(n + m, 1), dtype=torch.uint8, device=dev)
    ln = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = lib.ta_nw_general(t_d.data_ptr() if n else None, n, o_d.data_ptr() if m else None, m,
                           p_d.data_ptr(), table_d.data_ptr() if table_d is not None else None, tm,
                           sc.data_ptr(), ptr.data_ptr(), ops.data_ptr(), ln.data_ptr(),
                           torch.cuda.current_stream(dev).cuda_stream)
    _native.check(rc, "ta_nw_general")
    k = int(ln.item())
    return ops.cpu().numpy()[n + m - k:n + m].copy()


def align_batch(t_list, o_list, params):
    """Alignment columns of many problems with match/mismatch scoring in float64 (non-integral
    parameters, or integer ones too large for the 32-bit kernels), one launch: params is one system
    (6 numbers) or one per problem."""
    dev = torch.device("cuda")
    lib = _native.lib
    nprob = len(t_list)
    if nprob == 0:
        return []
    n = np.array([len(t) for t in t_list], dtype=np.int64)
    m = np.array([len(o) for o in o_list], dtype=np.int64)
    p = np.asarray(params, dtype=np.float64).reshape(-1, 6)
    if p.shape[0] not in (1, nprob):
        raise ValueError("need one scoring system, or one per problem")

    def offsets(sizes):
        off = np.zeros(nprob + 1, dtype=np.int64)
        np.cumsum(sizes, out=off[1:])
        return off
    t_off, o_off = offsets(n), offsets(m)
    sc_off = offsets([lib.ta_nw_general_score_bytes(int(a)) // 8 for a in n])
    ptr_off = offsets([lib.ta_nw_general_ptr_bytes(int(a), int(b)) for a, b in zip(n, m)])
    ops_off = offsets(n + m)

    def d(a, dt):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    cat_t = np.concatenate(t_list) if t_off[-1] else np.zeros(1, np.int32)
    cat_o = np.concatenate(o_list) if o_off[-1] else np.zeros(1, np.int32)
    bufs = (d(cat_t, np.int32), d(t_off, np.int64), d(cat_o, np.int32), d(o_off, np.int64), d(p, np.float64),
            torch.empty(max(int(sc_off[-1]), 1), dtype=torch.float64, device=dev), d(sc_off[:-1], np.int64),
            torch.empty(max(int(ptr_off[-1]), 1), dtype=torch.uint8, device=dev), d(ptr_off[:-1], np.int64),
            torch.empty(max(int(ops_off[-1]), 1), dtype=torch.uint8, device=dev), d(ops_off[:-1], np.int64),
            torch.zeros(nprob, dtype=torch.int32, device=dev))
    rc = lib.ta_nw_general_batch(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr(),
                                 nprob, bufs[4].data_ptr(), 0 if p.shape[0] == 1 else 6,
                                 bufs[5].data_ptr(), bufs[6].data_ptr(), bufs[7].data_ptr(), bufs[8].data_ptr(),
                                 bufs[9].data_ptr(), bufs[10].data_ptr(), bufs[11].data_ptr(),
                                 torch.cuda.current_stream(dev).cuda_stream)
    _native.check(rc, "ta_nw_general_batch")
    ops = bufs[9].cpu().numpy()
    lens = bufs[11].cpu().numpy()
    out = []
    for k in range(nprob):
        end = int(ops_off[k + 1])
        out.append(ops[end - int(lens[k]):end].copy())
    return out
