"""GPU parity tests of the line recogniser (K3 BiLSTM, K4 output layer + softmax, K5 decode)
against the float64 restatement oracle/ocr_ref_f64.py (parity unpinned:
the reference's OCR arithmetic is third-party and absent).  Tolerance from BASELINE.json's
north_star: logits within 1e-3 of the float64 result; decoded (t, class) lists identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL = 1e-3


def _tame(om):
    """Contractive variant of the synthetic model: recurrent columns and peepholes scaled by 1/4.
    The SURVEY-spec random weights (U(-0.5, 0.5) on 149 inputs) make the LSTM chaotic -- rounding
    differences of 1e-7 grow to 1e-3 within ~1000 timesteps on some lines, in ANY float32
    implementation -- so long-line parity at 1e-3 is asserted on this stable model, as trained
    recognisers are."""
    for w in (om.fwd, om.rev):
        for k in ("WGI", "WGF", "WGO", "WCI"):
            w[k][:, 49:] *= 0.25
        for k in ("WIP", "WFP", "WOP"):
            w[k] *= 0.25
    return om


def _models(seed, no):
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    om = _tame(R.synthetic_model(seed, no=no))
    pm = ocr.LineModel(om.fwd, om.rev, om.W2, om.codec)
    return R, ocr, om, pm


def _check_lines(R, ocr, om, widths, tol, check_decode=True, precision="f32"):
    pm = ocr.LineModel(om.fwd, om.rev, om.W2, om.codec)
    lines = [R.synthetic_line(8000 + k, width=w) for k, w in enumerate(widths)]
    rec = ocr.LineRecognizer(pm, precision=precision)
    dec, probs, logits, states = rec.recognise(lines, want_probs=True)
    assert rec.recognise(lines, from_probs=True) == dec        # K5 from full probabilities == from summaries
    errs = []
    for k, xs in enumerate(lines):
        ref = R.recognise(om, xs)
        assert states[k].shape == ref["states"].shape
        e_z = float(np.abs(logits[k] - ref["logits"]).max())
        e_p = float(np.abs(probs[k] - ref["probs"]).max())
        errs.append(e_z)
        assert e_z < tol, (k, widths[k], e_z)
        assert e_p < tol, (k, widths[k], e_p)
        if check_decode:
            assert dec[k] == ref["decoded"], (k, widths[k])
            ll = rec.llocs(dec[k], xs.shape[0], widths[k])
            assert ocr.llocs_text(ll) == R.llocs_text(ref["llocs"])
    print("logit errors: max %.3g median %.3g" % (max(errs), float(np.median(errs))))
    return errs


@pytest.mark.parametrize("seed,no", [(7001, 96), (7002, 64)])
def test_spec_model_short_lines(seed, no):
    """SURVEY section 8d weights, lines up to 333 columns: logits within 1e-3, decode identical."""
    assert torch.cuda.is_available()
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    _check_lines(R, ocr, R.synthetic_model(seed, no=no),
                 [1, 2, 3, 17, 31, 40, 64, 65, 100, 128, 200, 256, 300, 333], TOL)


@pytest.mark.parametrize("seed,no", [(7001, 96), (7002, 64)])
def test_stable_model_long_lines(seed, no):
    """Contractive model, lines of the benchmark's widths (up to 2000 columns): 1e-3, decode identical."""
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    errs = _check_lines(R, ocr, _tame(R.synthetic_model(seed, no=no)),
                        [499, 500, 501, 777, 800, 900, 1000, 1200, 1600, 2000], TOL)
    assert max(errs) < 2e-4


def test_split_mode():
    """The split-operand mode (16-bit matrix cores, f32 accumulation): 1e-3 on the contractive model
    at benchmark widths and on short lines of the spec model, decode identical."""
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    errs = _check_lines(R, ocr, _tame(R.synthetic_model(7001, no=96)), [100, 500, 1000, 2000], TOL,
                        precision="split")
    assert max(errs) < 1e-4
    _check_lines(R, ocr, R.synthetic_model(7001, no=96), [1, 17, 40, 64, 100], TOL, precision="split")


def _segmented_states(rec, st, om, lines, seg, group=16):
    """Re-run K3 over `st` (a prepared batch of `lines`) in segments of `seg` timesteps, every
    segment restarted from the float64 oracle's LSTM state at its boundary: the kernel's own drift
    is then bounded by what `seg` steps can accumulate, however chaotic the model is.  The segments
    are "lines" of the same row layout (row offset inside the parent line), so hout comes out in the
    parents' layout and K4 / K5 run unchanged."""
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import _native
    dev = rec.device
    row_off, T, h0, c0, ts = [], [], [], [], []
    for b, xs in enumerate(lines):
        Tl = xs.shape[0]
        f_h, f_c = R.lstm_forward(om.fwd, xs, return_cell=True)
        r_h, r_c = R.lstm_forward(om.rev, xs[::-1], return_cell=True)      # index = reversed time
        for a in range(0, Tl, seg):
            e = min(a + seg, Tl)
            row_off.append(int(st["row_start_host"][b]) + a)
            T.append(e - a)
            zero = np.zeros(100)
            done_rev = Tl - e                                  # reversed steps before this segment
            h0.append([f_h[a - 1] if a > 0 else zero, r_h[done_rev - 1] if done_rev > 0 else zero])
            c0.append([f_c[a - 1] if a > 0 else zero, r_c[done_rev - 1] if done_rev > 0 else zero])
            ts.append([a, done_rev])
    nseg = len(T)
    order = np.argsort(-np.asarray(T), kind="stable")
    ngroups = (nseg + group - 1) // group
    group_lines = np.full((ngroups, group), -1, dtype=np.int32)
    group_lines.reshape(-1)[:nseg] = order

    def d(a, dt):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=dt)).to(dev)
    args = (d(row_off, np.int64), d(T, np.int32), d(group_lines, np.int32),
            d(h0, np.float32), d(c0, np.float32), d(ts, np.int32))
    rc = _native.lib.ta_lstm_forward(st["x"].data_ptr(), args[0].data_ptr(), args[1].data_ptr(),
                                     args[2].data_ptr(), ngroups, (rec.wp4 if group == 4 else rec.wp).data_ptr(),
                                     rec.peep.data_ptr(), st["hout"].data_ptr(), 2 if group == 4 else rec.mode,
                                     args[3].data_ptr(), args[4].data_ptr(),
                                     args[5].data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    _native.check(rc, "ta_lstm_forward")
    torch.cuda.synchronize()
    return nseg


@pytest.mark.parametrize("seed,no", [(7001, 96), (7002, 64)])
@pytest.mark.parametrize("precision", ["f32", "split"])
def test_spec_model_benchmark_widths_per_segment(seed, no, precision):
    """SURVEY section 8(d)'s model AS SPECIFIED (no contraction) at the benchmark's widths (800 .. 2000
    columns): with the recurrence restarted from the float64 state every 128 timesteps, logits and
    probabilities stay within 1e-3 of the float64 restatement on every line and the decoded
    (t, class) lists are identical -- in both modes (measured: f32 5e-5; split operands see the
    printed value)."""
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    om = R.synthetic_model(seed, no=no)
    widths = [800, 977, 1200, 1433, 1601, 2000]
    lines = [R.synthetic_line(8100 + k, width=w) for k, w in enumerate(widths)]
    rec = ocr.LineRecognizer(ocr.LineModel(om.fwd, om.rev, om.W2, om.codec), precision=precision)
    st = rec.prepare(lines)
    nseg = _segmented_states(rec, st, om, lines, 128)
    assert nseg == sum((w + 32 + 127) // 128 for w in widths)
    rec.run(st, want_logits=True, lstm=False)
    dec = rec.decoded(st)
    logits, probs, states = st["logits"].cpu().numpy(), st["probs"].cpu().numpy(), st["hout"].cpu().numpy()
    worst = 0.0
    for k, xs in enumerate(lines):
        ref = R.recognise(om, xs)
        sl = slice(int(st["row_start_host"][k]), int(st["row_start_host"][k] + st["T_host"][k]))
        e_h = float(np.abs(states[sl] - ref["states"]).max())
        e_z = float(np.abs(logits[sl] - ref["logits"]).max())
        e_p = float(np.abs(probs[sl] - ref["probs"]).max())
        worst = max(worst, e_z)
        assert e_z < TOL and e_p < TOL and e_h < TOL, (k, widths[k], e_h, e_z, e_p)
        assert dec[k] == ref["decoded"], (k, widths[k])
    print("%s: worst logit error over %d segments: %.3g" % (precision, nseg, worst))


def test_continuation_equals_one_run_on_the_stable_model():
    """The continuation inputs themselves: a line cut into segments that are restarted from the
    KERNEL's own states reproduces the uncut run (contractive model: no amplification either way)."""
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    om = _tame(R.synthetic_model(7001, no=96))
    lines = [R.synthetic_line(8200 + k, width=w) for k, w in enumerate([300, 517])]
    rec = ocr.LineRecognizer(ocr.LineModel(om.fwd, om.rev, om.W2, om.codec), precision="f32")
    st = rec.prepare(lines)
    rec.run(st, output=False, decode=False)
    whole = st["hout"].cpu().numpy().copy()
    st["hout"].zero_()
    _segmented_states(rec, st, om, lines, 100)
    assert float(np.abs(st["hout"].cpu().numpy() - whole).max()) < 1e-5


@pytest.mark.parametrize("seed,no", [(7001, 96), (7002, 64)])
@pytest.mark.parametrize("precision", ["f32", "split"])
def test_spec_model_benchmark_widths_free_running(seed, no, precision):
    """SURVEY section 8(d)'s model AS SPECIFIED at the benchmark's widths (800 .. 2000 columns),
    FREE-RUNNING (the kernels' own state all the way), the two opt-in float32 modes (the default, float64, has the
    strict test below):
    what the north_star's "logits within 1e-3" looks like on a chaotic random-weight model.  Measured
    on 96 lines per model (tools/ocr_mode_agreement.py, profiles/r05_ocr_mode_agreement.json): f32 median
    1.3e-4 / 2.4e-5, 89 / 94 lines within 1e-3, worst line 1.5e-2; split median 3.5e-4 / 3.4e-5, 72 / 92
    lines, worst 6.6e-2; decode identical on every line in both modes.  Asserted here on 16 lines:
    the median and the share of lines within 1e-3 (per mode), a loose bound on the worst line, and that
    every decode difference -- none so far -- is one the measured probability difference explains.
    The per-segment test above is the parity statement that holds on EVERY line."""
    import ocr_compare
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    om = R.synthetic_model(seed, no=no)
    rng = np.random.default_rng(seed + 5)
    widths = [800, 2000] + [int(w) for w in rng.integers(800, 2001, size=14)]
    lines = [R.synthetic_line(8300 + k, width=w) for k, w in enumerate(widths)]
    kw = {} if precision is None else {"precision": precision}
    rec = ocr.LineRecognizer(ocr.LineModel(om.fwd, om.rev, om.W2, om.codec), **kw)
    c = ocr_compare.compare_lines(R, om, rec, lines)
    errs = c["logit_err"]
    within = sum(e < TOL for e in errs)
    print("%s seed %d: logit error median %.3g, p90 %.3g, max %.3g; %d / %d lines within 1e-3; characters "
          "%d / %d agree, %d lines with explained differences"
          % (precision or ocr.DEFAULT_PRECISION, seed, float(np.median(errs)), float(np.quantile(errs, 0.9)),
             max(errs), within, len(errs), c["chars_agree"], c["chars"], sum(1 for e in c["explained"] if e)))
    f32 = (precision or ocr.DEFAULT_PRECISION) == "f32"
    assert float(np.median(errs)) < (5e-4 if f32 else 1e-3)
    assert within >= (12 if f32 else 9)
    assert max(errs) < 0.2
    assert not any(c["unexplained"]), c["unexplained"]
    assert c["chars_agree"] >= 0.99 * c["chars"]


@pytest.mark.parametrize("seed,no", [(7001, 96), (7002, 64)])
def test_spec_model_benchmark_widths_free_running_f64(seed, no):
    """The north_star's tolerance AS STATED, on the model SURVEY section 8(d) specifies, at the widths the benchmark
    times, FREE-RUNNING: float64 mode (precision="f64": hoisted float64 input projection + float64 recurrence on
    v_mfma_f64_16x16x4_f64, csrc/ta_lstm_f64.hip).  EVERY line: LSTM outputs, logits and probabilities within 1e-3 of
    the float64 restatement (the reference's recogniser computes in float64; call site
    alignToOCR.py:142-147) and the decoded (t, class) lists identical -- no per-segment restart, no tamed model,
    no "explained" differences.  Same 16 lines per model as the f32 / split test above, plus the group edges."""
    import ocr_compare
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    om = R.synthetic_model(seed, no=no)
    rng = np.random.default_rng(seed + 5)
    widths = [800, 2000] + [int(w) for w in rng.integers(800, 2001, size=14)] + [1, 17, 333]      # 19 lines: two groups
    lines = [R.synthetic_line(8300 + k, width=w) for k, w in enumerate(widths)]
    rec = ocr.LineRecognizer(ocr.LineModel(om.fwd, om.rev, om.W2, om.codec), precision="f64")
    dec, probs, logits, states = rec.recognise(lines, want_probs=True)
    worst = [0.0, 0.0, 0.0]
    for k, xs in enumerate(lines):
        ref = R.recognise(om, xs)
        e_h = float(np.abs(states[k] - ref["states"]).max())
        e_z = float(np.abs(logits[k] - ref["logits"]).max())
        e_p = float(np.abs(probs[k] - ref["probs"]).max())
        worst = [max(worst[0], e_h), max(worst[1], e_z), max(worst[2], e_p)]
        assert e_h < TOL and e_z < TOL and e_p < TOL, (k, widths[k], e_h, e_z, e_p)
        assert dec[k] == ref["decoded"], (k, widths[k])
    print("f64 seed %d: worst state / logit / probability error over %d lines: %.3g / %.3g / %.3g"
          % (seed, len(lines), *worst))
    # the output layer runs in float32 on the float64 states rounded once: its own error bounds the total
    assert worst[1] < 1e-4


def test_f64_mode_group_edges_continuation_and_chunks(monkeypatch):
    """float64 mode plumbing: 1 .. 33 lines (every fill of a group of 16, empty slots), results independent of the
    batch; the hoisted projection held for a bounded number of rows at a time (F64_GX_MAX_ROWS forced small: many
    chunks) gives bit-identical states; a line cut into segments restarted from the float64 oracle's states
    (double h0 / c0, tstart) reproduces the oracle to 1e-9."""
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import _native, ocr
    om = R.synthetic_model(7002, no=64)
    rec = ocr.LineRecognizer(ocr.LineModel(om.fwd, om.rev, om.W2, om.codec), precision="f64")
    base = [R.synthetic_line(9000 + k, width=30 + 9 * k) for k in range(33)]
    ref = [R.recognise(om, xs) for xs in base]
    for cnt in (1, 15, 16, 17, 33):
        dec, probs, logits, states = rec.recognise(base[:cnt], want_probs=True)
        for k in range(cnt):
            assert float(np.abs(states[k] - ref[k]["states"]).max()) < 1e-6, (cnt, k)     # float32 rounding of the outputs
            assert dec[k] == ref[k]["decoded"], (cnt, k)
    st = rec.prepare(base)
    rec.run(st, output=False, decode=False)
    whole = st["hout"].clone()
    monkeypatch.setattr(ocr, "F64_GX_MAX_ROWS", 400)
    st["hout"].zero_()
    rec.run(st, output=False, decode=False)
    assert torch.equal(st["hout"], whole)
    monkeypatch.undo()
    # continuation: segments of 50 steps from the oracle's float64 states
    lines = base[20:27]
    st = rec.prepare(lines)
    dev = rec.device
    row_off, T, h0, c0, ts = [], [], [], [], []
    for b, xs in enumerate(lines):
        Tl = xs.shape[0]
        f_h, f_c = R.lstm_forward(om.fwd, xs, return_cell=True)
        r_h, r_c = R.lstm_forward(om.rev, xs[::-1], return_cell=True)
        for a in range(0, Tl, 50):
            e = min(a + 50, Tl)
            row_off.append(int(st["row_start_host"][b]) + a); T.append(e - a)
            zero, done_rev = np.zeros(100), Tl - e
            h0.append([f_h[a - 1] if a > 0 else zero, r_h[done_rev - 1] if done_rev > 0 else zero])
            c0.append([f_c[a - 1] if a > 0 else zero, r_c[done_rev - 1] if done_rev > 0 else zero])
            ts.append([a, done_rev])
    nseg = len(T)
    order = np.argsort(-np.asarray(T), kind="stable")
    ngroups = (nseg + 15) // 16
    gl = np.full((ngroups, 16), -1, dtype=np.int32)
    gl.reshape(-1)[:nseg] = order

    def d(a, dt):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=dt)).to(dev)
    args = (d(row_off, np.int64), d(T, np.int32), d(gl, np.int32), d(h0, np.float64), d(c0, np.float64), d(ts, np.int32))
    rows = st["rows"]
    gx = torch.empty(_native.lib.ta_lstm_f64_gx_bytes(rows) // 8, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _native.check(_native.lib.ta_lstm_xproj_f64(st["x"].data_ptr(), rows, rec.wx64.data_ptr(), gx.data_ptr(), stream), "xproj")
    st["hout"].zero_()
    _native.check(_native.lib.ta_lstm_forward_f64(gx.data_ptr(), 0, rows, args[0].data_ptr(), args[1].data_ptr(),
                                                  args[2].data_ptr(), ngroups, rec.wh64.data_ptr(), rec.peep64.data_ptr(),
                                                  st["hout"].data_ptr(), args[3].data_ptr(), args[4].data_ptr(),
                                                  args[5].data_ptr(), None, stream), "forward_f64")
    torch.cuda.synchronize()
    hout = st["hout"].cpu().numpy()
    for b, xs in enumerate(lines):
        s0 = int(st["row_start_host"][b])
        want = R.bilstm_states(om, xs)
        assert float(np.abs(hout[s0:s0 + xs.shape[0]] - want).max()) < 1e-6, b
    # the projection itself against numpy, to float64 rounding
    x = st["x"].cpu().numpy().astype(np.float64)
    # a row of Gx: per tile of 4 units [gate pair][unit in tile][2] (csrc/ta_lstm_f64.hip: gx_index)
    g = gx.cpu().numpy().reshape(2, rows, 25, 2, 4, 2).transpose(0, 1, 2, 4, 3, 5).reshape(2, rows, 100, 4)
    for dname, w in enumerate((om.fwd, om.rev)):
        for gi, name in enumerate(("WGI", "WGF", "WGO", "WCI")):
            want = w[name][:, 0][None, :] + x.dot(w[name][:, 1:49].T)
            assert float(np.abs(g[dname, :, :, gi] - want).max()) < 1e-12, (dname, name)


def test_f64_mode_class_pipeline_is_bit_identical(monkeypatch):
    """float64 mode on a batch large enough for the per-class pipeline (projection class by class, the recurrence of
    each class on a side stream as soon as its projection is done, own pieces of the Gx buffer): LSTM outputs bit for bit
    those of the one-after-the-other order, also when the buffer bound cuts the batch into several runs; and a
    few lines against the oracle, so that a wrong piece of the buffer cannot pass as 'equal to itself'."""
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    om = R.synthetic_model(7001, no=96)
    rec = ocr.LineRecognizer(ocr.LineModel(om.fwd, om.rev, om.W2, om.codec), precision="f64")
    lines = [R.synthetic_line(9300 + k, width=40 + (37 * k) % 160) for k in range(20 * 16 + 5)]
    st = rec.prepare(lines)
    assert st["ngroups"] >= ocr.F64_CLASS_MIN_GROUPS
    monkeypatch.setattr(ocr, "F64_CLASS_PIPELINE", False)
    rec.run(st, output=False, decode=False)
    torch.cuda.synchronize()
    serial = st["hout"].clone()
    monkeypatch.setattr(ocr, "F64_CLASS_PIPELINE", True)
    for max_rows in (ocr.F64_GX_MAX_ROWS, int(st["rows"]) // 3):
        monkeypatch.setattr(ocr, "F64_GX_MAX_ROWS", max_rows)
        st["hout"].zero_()
        rec.run(st, output=False, decode=False)
        torch.cuda.synchronize()
        assert torch.equal(st["hout"], serial), max_rows
    hout = serial.cpu().numpy()
    for b in (0, 7, 160, len(lines) - 1):
        s0 = int(st["row_start_host"][b])
        want = R.bilstm_states(om, lines[b])
        assert float(np.abs(hout[s0:s0 + lines[b].shape[0]] - want).max()) < 1e-6, b


def test_group_boundaries_and_order():
    """15, 16, 17 and 33 lines (group edges), results independent of batch composition."""
    R, ocr, om, pm = _models(7001, 96)
    rec = ocr.LineRecognizer(pm)
    base = [R.synthetic_line(9000 + k, width=60 + 7 * k) for k in range(33)]
    ref = [R.recognise(om, xs)["decoded"] for xs in base]
    for cnt in (1, 15, 16, 17, 33):
        dec = rec.recognise(base[:cnt])
        assert dec == ref[:cnt], cnt
    assert rec.recognise([]) == []


@pytest.mark.parametrize("precision,group", [("f32", None), ("f32", "16"), ("split", None), ("f64", None), ("f64", "16")])
def test_class_split_launches_equal_single_launches(precision, group, monkeypatch):
    """run(): the recurrence and the output layer per length class on side streams (large batches) against
    one launch each -- same kernels on the same rows, so states, summaries, probabilities and decode are
    bit for bit the same; rows are laid out by groups (longest lines first), every line at row_start ..
    + T, whatever the order the lines came in; and the one-off timing check leaves a verdict behind."""
    R, ocr, om, pm = _models(7001, 96)
    rec = ocr.LineRecognizer(pm, precision=precision)
    rng = np.random.default_rng(3)
    lines = [R.synthetic_line(9500 + k, width=int(w)) for k, w in enumerate(rng.integers(20, 200, size=16 * 26 + 5))]
    if group:
        monkeypatch.setattr(ocr, "FORCE_GROUP", int(group))   # the kernel large batches take, on a batch a test can afford
    st = rec.prepare(lines)
    G = st["group_size"]
    assert G == (4 if precision in ("f32", "f64") and not group else 16) and st["ngroups"] == (len(lines) + G - 1) // G
    T, start = st["T_host"], st["row_start_host"]
    order = np.argsort(-T, kind="stable")
    assert np.array_equal(start[order], np.cumsum(T[order]) - T[order])           # sorted layout, no holes
    assert np.array_equal(st["group_row_host"][:-1], start[order[::G]]) and st["group_row_host"][-1] == T.sum()
    out = {}
    for split in (False, True):
        for key in ("hout", "summary", "dec_t", "dec_c", "dec_n"):
            st[key].fill_(0)
        rec.run(st, want_logits=True, class_split=split)
        torch.cuda.synchronize()
        out[split] = {k: st[k].clone() for k in ("hout", "summary", "probs", "logits", "dec_t", "dec_c", "dec_n")}
    for k, v in out[False].items():
        assert torch.equal(v, out[True][k]), k
    dec = rec.decoded(st)
    few = [0, 5, 77, 300, len(lines) - 1]
    alone = rec.recognise([lines[k] for k in few])
    assert [dec[k] for k in few] == alone
    rec.run(st)                                               # the default path: split mode checks itself once, then decides
    if precision == "split":
        key = ocr._device_key(rec.device) + (rec.mode,)
        assert ocr._split_state["ok"][key] in (True, False) and set(ocr._split_state["times_ms"][key]) == {True, False}
        print("class split timing check:", ocr._split_state["times_ms"][key], "->", ocr._split_state["ok"][key])


def test_four_line_groups_equal_sixteen_line_groups(monkeypatch):
    """Exact-f32 mode has two recurrence kernels: groups of 16 lines on v_mfma_f32_16x16x4_f32 and groups of 4
    on v_mfma_f32_4x4x1_16B_f32 (small and medium batches).  One k per instruction in ascending order is the
    fmaf chain the 16 x 16 x 4 form computes too and the gate arithmetic is the same code, so the LSTM outputs
    are equal to the BIT -- which kernel a batch takes never shows in a result.  Ragged lengths, 1 .. 21
    lines (every fill of the last group), both directions, and a continued sequence (h0 / c0 / tstart)."""
    R, ocr, om, pm = _models(7002, 64)
    rec = ocr.LineRecognizer(pm, precision="f32")
    rng = np.random.default_rng(17)
    pool = [R.synthetic_line(9900 + k, width=int(w)) for k, w in enumerate(rng.integers(1, 260, size=21))]
    for cnt in (1, 2, 3, 4, 5, 7, 8, 16, 17, 21):
        got = {}
        for G in ("4", "16"):
            monkeypatch.setattr(ocr, "FORCE_GROUP", int(G))
            st = rec.prepare(pool[:cnt])
            assert st["group_size"] == int(G)
            rec.run(st, want_logits=True)
            torch.cuda.synchronize()
            # rows are laid out by groups of G: compare line by line
            got[G] = [(st["hout"][int(s):int(s + t)].clone(), st["logits"][int(s):int(s + t)].clone())
                      for s, t in zip(st["row_start_host"], st["T_host"])] + [rec.decoded(st)]
        for a, b in zip(got["4"][:-1], got["16"][:-1]):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), cnt
        assert got["4"][-1] == got["16"][-1]
    # continued sequences (h0 / c0 / tstart): segments of 37 steps restarted from the float64 states, both kernels
    monkeypatch.setattr(ocr, "FORCE_GROUP", None)
    lines = [ln for ln in pool if ln.shape[0] >= 60][:7]
    st = rec.prepare(lines)
    seg = {}
    for G in (4, 16):
        st["hout"].zero_()
        _segmented_states(rec, st, om, lines, 37, group=G)
        seg[G] = st["hout"].clone()
    assert torch.equal(seg[4], seg[16])


def test_f64_four_line_groups_equal_sixteen_line_groups(monkeypatch):
    """Float64 mode has two recurrence kernels as well: groups of 16 lines on v_mfma_f64_16x16x4_f64 and groups of 4 on
    v_mfma_f64_4x4x4_4b_f64 (csrc/ta_lstm_f64.hip: lstm_seq_f64_kernel / lstm_seq4_f64_kernel).  A block's 4-term product is
    the fma chain k = 0..3 of the 16 x 16 x 4 form, the accumulators start from the same Gx, the 25th tile is summed
    from the same three partial chains and the cell update is the same function: LSTM outputs equal to the BIT.  Ragged
    lengths, 1 .. 21 lines (every fill of the last group), both directions, a continued sequence (double h0 / c0 /
    tstart) through both entry points, and a few lines against the float64 oracle so that 'equal' is not 'equally wrong'."""
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import _native, ocr
    om = R.synthetic_model(7002, no=64)
    rec = ocr.LineRecognizer(ocr.LineModel(om.fwd, om.rev, om.W2, om.codec), precision="f64")
    rng = np.random.default_rng(23)
    pool = [R.synthetic_line(9700 + k, width=int(w)) for k, w in enumerate(rng.integers(1, 260, size=21))]
    for cnt in (1, 2, 3, 4, 5, 7, 8, 16, 17, 21):
        got = {}
        for G in (4, 16):
            monkeypatch.setattr(ocr, "FORCE_GROUP", G)
            st = rec.prepare(pool[:cnt])
            assert st["group_size"] == G
            rec.run(st, want_logits=True)
            torch.cuda.synchronize()
            got[G] = [(st["hout"][int(s):int(s + t)].clone(), st["logits"][int(s):int(s + t)].clone())
                      for s, t in zip(st["row_start_host"], st["T_host"])] + [rec.decoded(st)]
        for a, b in zip(got[4][:-1], got[16][:-1]):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), cnt
        assert got[4][-1] == got[16][-1]
        if cnt in (5, 21):
            for k in (0, cnt - 1):
                want = R.bilstm_states(om, pool[k])
                assert float(np.abs(got[4][k][0].cpu().numpy() - want).max()) < 1e-6, (cnt, k)
    monkeypatch.setattr(ocr, "FORCE_GROUP", None)
    assert rec.prepare(pool[:5])["group_size"] == 4           # the product's own choice for a small batch
    # continued sequences: segments of 37 steps from the oracle's float64 states, through both entry points
    lines = [ln for ln in pool if ln.shape[0] >= 60][:7]
    st = rec.prepare(lines)
    dev = rec.device
    row_off, T, h0, c0, ts = [], [], [], [], []
    for b, xs in enumerate(lines):
        Tl = xs.shape[0]
        f_h, f_c = R.lstm_forward(om.fwd, xs, return_cell=True)
        r_h, r_c = R.lstm_forward(om.rev, xs[::-1], return_cell=True)
        for a in range(0, Tl, 37):
            e = min(a + 37, Tl)
            row_off.append(int(st["row_start_host"][b]) + a); T.append(e - a)
            zero, done_rev = np.zeros(100), Tl - e
            h0.append([f_h[a - 1] if a > 0 else zero, r_h[done_rev - 1] if done_rev > 0 else zero])
            c0.append([f_c[a - 1] if a > 0 else zero, r_c[done_rev - 1] if done_rev > 0 else zero])
            ts.append([a, done_rev])
    nseg = len(T)
    order = np.argsort(-np.asarray(T), kind="stable")

    def d(a, dt):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=dt)).to(dev)
    common = (d(row_off, np.int64), d(T, np.int32), d(h0, np.float64), d(c0, np.float64), d(ts, np.int32))
    rows = st["rows"]
    gx = torch.empty(_native.lib.ta_lstm_f64_gx_bytes(rows) // 8, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _native.check(_native.lib.ta_lstm_xproj_f64(st["x"].data_ptr(), rows, rec.wx64.data_ptr(), gx.data_ptr(), stream), "xproj")
    seg = {}
    for G, fn, wh in ((4, _native.lib.ta_lstm_forward_f64_g4, rec.wh64g4), (16, _native.lib.ta_lstm_forward_f64, rec.wh64)):
        ngroups = (nseg + G - 1) // G
        gl = np.full((ngroups, G), -1, dtype=np.int32)
        gl.reshape(-1)[:nseg] = order
        gld = d(gl, np.int32)
        st["hout"].zero_()
        _native.check(fn(gx.data_ptr(), 0, rows, common[0].data_ptr(), common[1].data_ptr(), gld.data_ptr(), ngroups,
                         wh.data_ptr(), rec.peep64.data_ptr(), st["hout"].data_ptr(), common[2].data_ptr(),
                         common[3].data_ptr(), common[4].data_ptr(), None, stream), "forward_f64 G=%d" % G)
        torch.cuda.synchronize()
        seg[G] = st["hout"].clone()
    assert torch.equal(seg[4], seg[16])
    hout = seg[4].cpu().numpy()
    for b, xs in enumerate(lines):
        s0 = int(st["row_start_host"][b])
        assert float(np.abs(hout[s0:s0 + xs.shape[0]] - R.bilstm_states(om, xs)).max()) < 1e-6, b


def test_input_too_large():
    R, ocr, om, pm = _models(7001, 96)
    rec = ocr.LineRecognizer(pm)
    with pytest.raises(ocr.RecognitionError):
        rec.recognise([np.zeros((5001, 48))])


def test_decode_kernel_on_crafted_probabilities():
    """K5 alone on hand-made outputs: ties go to the earlier t, then the smaller class; a run
    open at the end of the line is emitted."""
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import _native
    no = 70
    T = 12
    p = np.full((T, no), 0.001, dtype=np.float32)
    p[:, 0] = 0.9
    p[2:5, 0] = 0.1; p[3, 5] = 0.6; p[4, 5] = 0.6; p[3, 69] = 0.6       # tie: (3,5) wins
    p[7, 0] = 0.69; p[7, 1] = 0.3
    p[10:, 0] = 0.2; p[11, 66] = 0.5                                    # run open at the end
    want = R.translate_back(p.astype(np.float64))
    dev = torch.device("cuda")
    pd = torch.from_numpy(p).to(dev)
    ro = torch.zeros(1, dtype=torch.int64, device=dev)
    Td = torch.tensor([T], dtype=torch.int32, device=dev)
    dt = torch.zeros(T, dtype=torch.int32, device=dev); dc = torch.zeros(T, dtype=torch.int32, device=dev)
    dn = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = _native.lib.ta_decode(pd.data_ptr(), ro.data_ptr(), Td.data_ptr(), 1, no, 0.7,
                               dt.data_ptr(), dc.data_ptr(), dn.data_ptr(), ro.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    k = int(dn.item())
    got = list(zip(dt.cpu().numpy()[:k].tolist(), dc.cpu().numpy()[:k].tolist()))
    assert got == want == [(3, 5), (7, 0), (11, 66)]


@pytest.mark.parametrize("precision", ["f32", "split"])
@pytest.mark.parametrize("no", [5, 16, 17, 33, 80, 112, 128])
def test_class_counts_cover_every_output_tile_variant(no, precision):
    """The output layer is compiled once per number of 16-class tiles (1..8), in both forms (f32
    MFMA; split 16-bit operands, which also take 4, 3 or 2 row tiles per wave depending on the
    class count): class counts on both sides of the tile boundaries, up to the maximum of 128, and
    row counts that leave the last row tiles of a wave partly or wholly empty."""
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    _check_lines(R, ocr, _tame(R.synthetic_model(7100 + no, no=no)), [40, 129, 300, 1, 7], TOL, precision=precision)


def test_split_output_layer_matches_f32_output_layer():
    """K4 alone: the split-operand output layer against the f32-input one on the same LSTM states
    (random states in (-1, 1), the spec model's W2): logits within 2e-5, far inside the 1e-3 budget."""
    import torch
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import _native, ocr
    om = R.synthetic_model(7001, no=96)
    rec = ocr.LineRecognizer(ocr.LineModel(om.fwd, om.rev, om.W2, om.codec), precision="split")
    rng = np.random.default_rng(5)
    for rows in (1, 47, 48, 49, 1000, 4099):
        h = torch.from_numpy(rng.uniform(-1, 1, size=(rows, 200)).astype(np.float32)).cuda()
        outs = []
        for split in (False, True):
            probs = torch.empty((rows, 96), dtype=torch.float32, device="cuda")
            logits = torch.empty_like(probs)
            summary = torch.empty((rows, 4), dtype=torch.float32, device="cuda")
            if split:
                rc = _native.lib.ta_lstm_output_split(h.data_ptr(), rows, rec.w2s.data_ptr(), rec.w2bias.data_ptr(), 96,
                                                      probs.data_ptr(), logits.data_ptr(), summary.data_ptr(), None)
            else:
                rc = _native.lib.ta_lstm_output(h.data_ptr(), rows, rec.w2p.data_ptr(), 96,
                                                probs.data_ptr(), logits.data_ptr(), summary.data_ptr(), None)
            assert rc == 0
            torch.cuda.synchronize()
            outs.append((probs.cpu().numpy(), logits.cpu().numpy(), summary.cpu().numpy()))
        want = h.cpu().numpy().astype(np.float64).dot(om.W2[:, 1:].T) + om.W2[:, 0]
        assert np.abs(outs[0][1] - want).max() < 2e-5
        assert np.abs(outs[1][1] - want).max() < 2e-5, np.abs(outs[1][1] - want).max()
        assert np.abs(outs[1][0] - outs[0][0]).max() < 3e-5       # probabilities: |dp| <= p |dz|
        assert np.array_equal(outs[1][2][:, 2].view(np.int32), outs[0][2][:, 2].view(np.int32)) or \
            np.abs(outs[1][2][:, 1] - outs[0][2][:, 1]).max() < 3e-5
