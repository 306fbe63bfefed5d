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
