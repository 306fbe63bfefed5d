"""ctypes binding of libta_hip.so (C ABI: include/text_alignment_amd.h).

The HIP library is the product: if it is missing or does not load, importing this module
raises -- there is no CPU fallback anywhere in the package.
"""
import ctypes
import os

# torch first: its wheel bundles the HIP runtime (libamdhip64, soname .so.7).  Loading
# libta_hip.so before torch would pull in /opt/rocm's copy and leave two HIP runtimes in one
# process (kernel launches then fail with "no ROCm-capable device").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# TA_HIP_LIB: an alternative build of the same library (kernel timing experiments, tools/p1_ablate.sh)
LIB_PATH = os.environ.get("TA_HIP_LIB") or os.path.join(_HERE, "libta_hip.so")

TA_OK = 0
TA_PP_LABEL_PIXELS = 1          # flags of ta_pp_binarise_batch / ta_pp_line_components_batch (see the header)
TA_EINVAL, TA_ERANGE, TA_EHIP, TA_ELIMIT = -1, -2, -3, -4
TA_NW_FILL, TA_NW_TRACEBACK, TA_NW_CODES8, TA_NW_WIDE, TA_NW_NARROW = 1, 2, 4, 8, 16
TA_NW_OPENS_SAME, TA_NW_ALPHABET_SHIFT = 32, 8
TA_NW_NO_PROFILE, TA_NW_WAVES_SHIFT, TA_NW_ROWS_SHIFT, TA_NW_TBWAVES_SHIFT = 64, 16, 20, 24
TA_NW_CHECK_IDS = 128


class NativeLibraryError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            "libta_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C text_alignment_amd/csrc`). There is no CPU fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise NativeLibraryError("cannot load %s: %s" % (LIB_PATH, e))
    vp, i32, i64, u32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32
    lib.ta_version.restype = ctypes.c_int
    lib.ta_version.argtypes = []
    lib.ta_last_error.restype = ctypes.c_char_p
    lib.ta_last_error.argtypes = []
    lib.ta_nw_workspace_bytes.restype = i64
    lib.ta_nw_workspace_bytes.argtypes = [i32, i32]
    lib.ta_nw_max_m.restype = i32
    lib.ta_nw_max_m.argtypes = []
    lib.ta_nw_batch.restype = ctypes.c_int
    lib.ta_nw_batch.argtypes = [vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp,
                                i32, i32, i64, u32, vp]
    lib.ta_nw2_max_m.restype = i32
    lib.ta_nw2_max_m.argtypes = []
    lib.ta_nw2_workspace_bytes.restype = i64
    lib.ta_nw2_workspace_bytes.argtypes = [i32, i32]
    lib.ta_nw2_batch.restype = ctypes.c_int
    lib.ta_nw2_batch.argtypes = lib.ta_nw_batch.argtypes
    lib.ta_nw2_traceback_plan.restype = i32
    lib.ta_nw2_traceback_plan.argtypes = [i32, i32, ctypes.c_uint32]
    lib.ta_nw2_phase1_plan.restype = ctypes.c_int
    lib.ta_nw2_phase1_plan.argtypes = [i32, i32, ctypes.c_uint32, ctypes.c_void_p]
    lib.ta_nw2_phase1_plan_batch.restype = ctypes.c_int
    lib.ta_nw2_phase1_plan_batch.argtypes = [i32, i32, i32, ctypes.c_uint32, ctypes.c_void_p]
    lib.ta_nw_general_score_bytes.restype = i64
    lib.ta_nw_general_score_bytes.argtypes = [i32]
    lib.ta_nw_general_ptr_bytes.restype = i64
    lib.ta_nw_general_ptr_bytes.argtypes = [i32, i32]
    lib.ta_nw_general.restype = ctypes.c_int
    lib.ta_nw_general.argtypes = [vp, i32, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.ta_nw_general_batch.restype = ctypes.c_int
    lib.ta_nw_general_batch.argtypes = [vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    f32 = ctypes.c_float
    lib.ta_lstm_packed_weight_floats.restype = i32
    lib.ta_lstm_packed_weight_floats.argtypes = [i32]
    lib.ta_lstm_forward.restype = ctypes.c_int
    lib.ta_lstm_forward.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp]
    lib.ta_lstm_f64_weight_doubles.restype = i64
    lib.ta_lstm_f64_weight_doubles.argtypes = [i32]
    lib.ta_lstm_f64_gx_bytes.restype = i64
    lib.ta_lstm_f64_gx_bytes.argtypes = [i64]
    lib.ta_lstm_xproj_f64.restype = ctypes.c_int
    lib.ta_lstm_xproj_f64.argtypes = [vp, i64, vp, vp, vp]
    lib.ta_lstm_forward_f64.restype = ctypes.c_int
    lib.ta_lstm_forward_f64.argtypes = [vp, i64, i64, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ta_lstm_forward_f64_g4.restype = ctypes.c_int
    lib.ta_lstm_forward_f64_g4.argtypes = [vp, i64, i64, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ta_lstm_output.restype = ctypes.c_int
    lib.ta_lstm_output.argtypes = [vp, i64, vp, i32, vp, vp, vp, vp]
    lib.ta_lstm_output_split_weight_bytes.restype = i64
    lib.ta_lstm_output_split_weight_bytes.argtypes = [i32]
    lib.ta_lstm_output_split.restype = ctypes.c_int
    lib.ta_lstm_output_split.argtypes = [vp, i64, vp, vp, i32, vp, vp, vp, vp]
    lib.ta_decode_summary.restype = ctypes.c_int
    lib.ta_decode_summary.argtypes = [vp, vp, vp, i32, f32, vp, vp, vp, vp, vp]
    lib.ta_decode.restype = ctypes.c_int
    lib.ta_decode.argtypes = [vp, vp, vp, i32, i32, f32, vp, vp, vp, vp, vp]
    lib.ta_device_pci_bus_id.restype = ctypes.c_int
    lib.ta_device_pci_bus_id.argtypes = [i32, ctypes.c_char_p, i32]
    lib.ta_host_copy_pieces.restype = ctypes.c_int
    lib.ta_host_copy_pieces.argtypes = [vp, vp, vp, vp, i32]
    lib.ta_host_chars_of_batch.restype = ctypes.c_int
    lib.ta_host_chars_of_batch.argtypes = [vp] * 10 + [i32, i32, i32, i64] + [vp] * 4
    lib.ta_host_syllable_boxes.restype = ctypes.c_int
    lib.ta_host_syllable_boxes.argtypes = [vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, vp]
    lib.ta_host_peak_candidates.restype = ctypes.c_int
    lib.ta_host_peak_candidates.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp]
    lib.ta_host_peak_select.restype = ctypes.c_int
    lib.ta_host_peak_select.argtypes = [vp, vp, vp, i32, vp, vp, vp, ctypes.c_double, vp, vp, vp, vp]
    lib.ta_host_otsu_batch.restype = ctypes.c_int
    lib.ta_host_otsu_batch.argtypes = [vp, i32, vp]
    lib.ta_host_sharpest_rows.restype = ctypes.c_int
    lib.ta_host_sharpest_rows.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp]
    lib.ta_host_line_boxes.restype = ctypes.c_int
    lib.ta_host_line_boxes.argtypes = [vp, i64, vp, i64, i64, vp, vp]
    lib.ta_rows_gather.restype = ctypes.c_int
    lib.ta_rows_gather.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.ta_linenorm_measure.restype = ctypes.c_int
    lib.ta_linenorm_measure.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ta_linenorm_resample.restype = ctypes.c_int
    lib.ta_linenorm_resample.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    pp = {"ta_pp_histogram": [vp, i64, vp, vp], "ta_pp_threshold": [vp, i64, i32, i32, vp, vp],
          "ta_pp_label": [vp, i32, i32, vp, vp, vp, vp], "ta_pp_label_batch": [i32, vp, vp, vp, vp, vp, vp, vp],
          "ta_pp_components": [vp, vp, i32, i32, vp, i32, vp, vp],
          "ta_pp_filter_components": [vp, vp, vp, i32, i32, i32, i32, vp], "ta_pp_invert": [vp, i64, vp],
          "ta_pp_angle_histograms": [vp, i32, i32, i32, vp, i32, vp, vp],
          "ta_pp_rotate": [vp, i32, i32, vp, i32, i32, vp, vp], "ta_pp_open_runs": [vp, vp, i32, i32, i32, i32, vp],
          "ta_pp_row_sums": [vp, i32, i32, vp, vp], "ta_pp_clear_rows": [vp, i32, vp, i32, vp],
          "ta_pp_cut_strips": [vp, i32, i32, vp, i32, vp, vp],
          "ta_pp_peak_prominence_args": [vp, i32, vp, i32, ctypes.c_double, vp],
          "ta_pp_ink_points": [vp, i32, i32, i32, vp, vp, vp],
          "ta_pp_angle_histograms_points": [vp, vp, i32, i32, vp, i32, vp, vp],
          "ta_pp_histogram_batch": [i32, vp, vp, vp, vp],
          "ta_pp_binarise_batch": [i32, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp],
          "ta_pp_angle_histograms_points_batch": [i32, vp, vp, vp, vp, vp, vp, vp, vp],
          "ta_pp_deskew_batch": [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp],
          "ta_pp_line_components_batch": [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp],
          "ta_pp_cut_strips_batch": [i32, vp, vp, vp, vp, vp, vp, vp]}
    for name, args in pp.items():
        getattr(lib, name).restype = ctypes.c_int
        getattr(lib, name).argtypes = args
    return lib


lib = _load()

EXPORTS = ["ta_version", "ta_last_error", "ta_device_pci_bus_id", "ta_host_copy_pieces", "ta_host_chars_of_batch", "ta_host_syllable_boxes", "ta_host_peak_candidates", "ta_host_peak_select", "ta_host_otsu_batch", "ta_host_sharpest_rows", "ta_host_line_boxes", "ta_nw_workspace_bytes", "ta_nw_max_m", "ta_nw_batch", "ta_nw2_workspace_bytes", "ta_nw2_max_m", "ta_nw2_batch", "ta_nw2_phase1_plan", "ta_nw2_phase1_plan_batch", "ta_nw2_traceback_plan",
           "ta_nw_general_score_bytes", "ta_nw_general_ptr_bytes", "ta_nw_general", "ta_nw_general_batch",
           "ta_lstm_packed_weight_floats", "ta_lstm_forward", "ta_lstm_f64_weight_doubles", "ta_lstm_f64_gx_bytes", "ta_lstm_xproj_f64", "ta_lstm_forward_f64", "ta_lstm_forward_f64_g4", "ta_lstm_output", "ta_lstm_output_split_weight_bytes", "ta_lstm_output_split", "ta_decode",
           "ta_decode_summary", "ta_rows_gather", "ta_linenorm_measure", "ta_linenorm_resample",
           "ta_pp_histogram", "ta_pp_threshold", "ta_pp_label", "ta_pp_label_batch", "ta_pp_components", "ta_pp_filter_components",
           "ta_pp_invert", "ta_pp_angle_histograms", "ta_pp_rotate", "ta_pp_open_runs", "ta_pp_row_sums",
           "ta_pp_clear_rows", "ta_pp_cut_strips", "ta_pp_peak_prominence_args", "ta_pp_ink_points",
           "ta_pp_angle_histograms_points", "ta_pp_histogram_batch", "ta_pp_binarise_batch",
           "ta_pp_angle_histograms_points_batch", "ta_pp_deskew_batch", "ta_pp_line_components_batch", "ta_pp_cut_strips_batch"]


class NativeArgumentError(ValueError):
    """TA_EINVAL from the library: the CALLER passed a bad argument -- a programming error, not a property of
    a page's data (sharding.process_shard tells the two apart)"""


def check(rc, what):
    if rc != TA_OK:
        msg = lib.ta_last_error().decode("utf-8", "replace")
        if rc == TA_EINVAL:
            raise NativeArgumentError("%s: %s" % (what, msg))
        if rc == TA_ERANGE:
            raise OverflowError("%s: %s" % (what, msg))
        raise RuntimeError("%s failed (%d): %s" % (what, rc, msg))


def upload_packed(arrays, device):
    """Several small host arrays to the device in ONE asynchronous transfer: packed (16-byte aligned) into a pinned
    buffer, copied non_blocking on torch's current stream, returned as typed views of one device buffer.  A
    `tensor.to(device)` from pageable memory costs ~0.25 ms of host time each; a batch's metadata was ten of them."""
    import numpy as np
    import torch
    arrays = [np.ascontiguousarray(a) for a in arrays]
    offs, total = [], 0
    for a in arrays:
        offs.append(total)
        total += (a.nbytes + 15) // 16 * 16
    host = torch.empty(max(total, 16), dtype=torch.uint8, pin_memory=True)
    hv = host.numpy()
    for a, off in zip(arrays, offs):
        if a.nbytes:
            hv[off:off + a.nbytes] = a.reshape(-1).view(np.uint8)
    dev = host.to(device, non_blocking=True)
    out = []
    for a, off in zip(arrays, offs):
        t = dev[off:off + a.nbytes].view(getattr(torch, a.dtype.name))
        out.append(t.reshape(a.shape))
    return out


def host_copy_pieces(dst_view, pieces, dst_off_bytes):
    """pieces (C-contiguous numpy arrays) -> dst_view (a writable, C-contiguous numpy array, e.g. the view of a page-locked
    tensor) at byte offsets dst_off_bytes[k], in ONE native call that holds no interpreter lock (ta_host_copy_pieces)."""
    import numpy as np
    n = len(pieces)
    if n == 0:
        return
    src = np.fromiter((p.ctypes.data for p in pieces), dtype=np.uint64, count=n)
    nb = np.fromiter((p.nbytes for p in pieces), dtype=np.int64, count=n)
    off = np.ascontiguousarray(dst_off_bytes, dtype=np.int64)
    end = int((off + nb).max())
    if end > dst_view.nbytes or int(off.min()) < 0:
        raise ValueError("a piece does not fit the staging buffer")
    check(lib.ta_host_copy_pieces(dst_view.ctypes.data, src.ctypes.data, off.ctypes.data, nb.ctypes.data, n), "ta_host_copy_pieces")
