#!/usr/bin/env python3
"""bench.py -- throughput of the HIP hot path (DESIGN.md "Measurement").

Headline (the JSON line's `value`): NW DP cells/s.  One "step" = fill + traceback of one batch
of synthetic 4096 x 4096 affine-gap problems per GPU (default scoring, SURVEY.md section 8d
generator), inputs resident in HBM, plus -- with more than one GPU -- the single gather of
syllable-box records to rank 0 (the only collective on the path).  Pages/problems shard across
ranks with no other communication, so scaling is "weak": every rank runs the same batch size.

`--gpus N` without a launcher (no RANK in the environment) starts N ranks itself -- a fresh
`python -m torch.distributed.run` child, before this process has touched the GPU -- and exits with
the child's code; under the driver's own torchrun the ranks are already there.

Also on the line: `pages_sharded` (BASELINE configs[4]: 64 synthetic pages per GPU, Salzinnes- and
St-Gall-shaped models, sharding.process_shard with its single gather of syllable-box records),
`configs` (SURVEY.md 8(d)'s named NW shapes, N = 1 only), `roofline` for the dominant kernel (nw_score_kernel of the two-phase aligner, or
nw_fill_kernel with --one-pass; HIP events around its launches on the launch stream),
`cpu_baseline` (the reference's algorithm on this host's cores, bounded sample, at N = 1 only),
`ocr` (text-lines/s of the line recogniser on synthetic lines, timed separately, with its own MFMA
roofline) and `pages_end_to_end` (process_batch from normalised strips, raw strips and whole page
images).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
F32_MFMA_PEAK_TF = 157.3       # MI355X_MICROARCH.md: f32-input MFMA peak (spec)
BF16_MFMA_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (spec)
F64_MFMA_PEAK_TF = 78.6        # f64 matrix peak (spec); measured: v_mfma_f64_16x16x4_f64 at 64 cycles per SIMD = 78.6 TF
                               # at 2.4 GHz (tools/ubench/mfma_f64.hip, profiles/r04_mfma_f64.txt)
DEFAULT_SYS = [8, -4, -7, -7, -3, 0]
P1_VALU_PER_BLOCK = 382        # nw_score_kernel, score profile + one gap open: VALU instructions per 64 cells of a lane
P1_NS_PER_VALU = 1.72          # measured issue interval of that instruction mix (profiles/r05_valu_issue_rates.txt: 1.722 at 8 waves per SIMD; round 2: 1.72)


def _profile_file(*stems):
    """(name, path) of the NEWEST round's copy of a profile summary: for each stem ("nw2_hbm_traffic.json", ...) the file
    profiles/rNN_<stem> with the highest NN present -- the line never cites an older round than profiles/ holds
    (tests/test_bench_profiles.py).  Several stems: the first that exists in any round."""
    import re
    try:
        listing = os.listdir(os.path.join(REPO, "profiles"))
    except OSError:
        return None, None
    for stem in stems:
        rounds = sorted((int(m.group(1)), f) for f in listing for m in [re.match(r"r(\d+)_" + re.escape(stem) + "$", f)] if m)
        if rounds:
            name = rounds[-1][1]
            return name, os.path.join(REPO, "profiles", name)
    return None, None


def measured_traffic(batch, n, m, kernel):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes kept under
    profiles/ (WRITE_SIZE + 2 x FETCH_SIZE, MI355X_MICROARCH.md HBM section) -- a number read from
    that file, not measured in this run; (None, None) for configs that were not profiled."""
    stem = "nw2_hbm_traffic.json" if kernel == "nw_score_kernel" else "nw_hbm_traffic.json"
    name, path = _profile_file(stem)
    try:
        with open(path) as f:
            d = json.load(f)
        if d["config"] == {"batch": batch, "n": n, "m": m}:
            for k, v in d["kernels"].items():
                if kernel in k:
                    return v["hbm_bytes_per_launch"], "profiles/" + name
    except (OSError, KeyError, ValueError, TypeError):
        pass
    return None, None


def measured_mfma_busy(kernel):
    """MFMA pipe utilisation of a recogniser kernel from the counter pass kept under profiles/
    (SQ_VALU_MFMA_BUSY_CYCLES over all SIMD-cycles, 1920-line workload); (None, None) if absent."""
    name, path = _profile_file("ocr_pmc_mfma.json")
    try:
        with open(path) as f:
            for k, v in json.load(f)["kernels"].items():
                if (kernel + "(") in k or k.endswith(kernel):
                    return v["mfma_pipe_utilisation"], "profiles/" + name
    except (OSError, KeyError, ValueError, TypeError):
        pass
    return None, None


# float64 recurrence on groups of four lines: what a SIMD must ISSUE per timestep of a workgroup.  Its float64 matrix and
# float64 vector instructions do not overlap (profiles/r05_simd_pair.txt: 2 765 + 1 985 = 4 795 cycles), so the floor of a
# step is their sum: 150 + 9 v_mfma_f64_4x4x4_4b_f64 at 16 cycles (profiles/r05_mfma_f64_4x4.txt) on the busiest SIMD and
# the 127 + 120 vector instructions of its two waves at 4 cycles each (tools/isa_loop_counts.py on csrc/ta_lstm_f64.hip).
F64G4_MFMA_PER_SIMD_STEP, F64G4_VALU_PER_SIMD_STEP = 159, 247


def f64_issue_floor():
    """{issue floor, measured cycles per step, their ratio} of lstm_seq4_f64_kernel from the cycle counters kept under
    profiles/ (f64_step_cycles.txt, a -DTA_F64_PROFILE build) -- read from that file, not measured in this run."""
    import re
    name, path = _profile_file("f64_step_cycles.txt")
    try:
        with open(path) as f:
            m = re.search(r"total (\d+)", f.read())             # the first entry is the four-line kernel
        measured = int(m.group(1))
    except (OSError, TypeError, AttributeError, ValueError):
        return {}
    floor = 16 * F64G4_MFMA_PER_SIMD_STEP + 4 * F64G4_VALU_PER_SIMD_STEP
    return {"binding_unit": "f64 datapath of a SIMD (its float64 matrix and vector instructions do not overlap)",
            "issue_floor_cycles_per_step": floor, "measured_cycles_per_step": measured, "issue_floor_frac": floor / measured,
            "issue_floor_source": "profiles/%s (step), profiles/%s (no overlap), %d MFMAs x 16 + %d VALU x 4 cycles "
                                  "per SIMD and step" % (name, _profile_file("simd_pair.txt")[0], F64G4_MFMA_PER_SIMD_STEP,
                                                         F64G4_VALU_PER_SIMD_STEP)}


def make_nw_batch(tsc, nprob, n, m, seed0, distinct=256, two_phase=False):
    from tools.synth import synth_pair_ids      # seeded input generator shared with the tests
    uniq = [synth_pair_ids(n, m, seed0 + k) for k in range(min(nprob, distinct))]
    probs = [uniq[k % len(uniq)] for k in range(nprob)]
    return tsc.NWBatch([p[0] for p in probs], [p[1] for p in probs], DEFAULT_SYS, two_phase=two_phase), uniq


def synthetic_lines(nlines, seed0):
    """Already-normalised strips 48 x W', W' ~ U[800, 2000]: (T, 48)."""
    rng = np.random.default_rng(seed0)
    lines = []
    for _ in range(nlines):
        w = int(rng.integers(800, 2001))
        xs = np.zeros((w + 32, 48), dtype=np.float32)
        xs[16:16 + w] = (rng.random((w, 48)) < 0.15) * rng.random((w, 48))
        lines.append(xs)
    return lines


def cpu_baseline(seconds=12.0):
    """oracle/nw_ref_py.py -- behavioural port of textSeqCompare.py:13-177 (per-cell Python loop
    over float64 numpy matrices) -- on one host core, config-1-shaped problems."""
    from oracle import nw_ref_py, nw_oracle
    from tools.synth import synth_pair, synth_pair_ids
    n = m = 500
    done, t0 = 0, time.perf_counter()
    while True:
        t, o = synth_pair(n, m, 1234 + done)
        nw_ref_py.perform_alignment(t, o)
        done += 1
        dt = time.perf_counter() - t0
        if dt > seconds or done >= 64:
            break
    # SURVEY 8(d)'s second single-process point: ONE 2048 x 2048 problem (six float64 matrices of 2049^2: 200 MB)
    t2, o2 = synth_pair(2048, 2048, 1234)
    t1 = time.perf_counter()
    a2 = nw_ref_py.perform_alignment(t2, o2)
    dt2 = time.perf_counter() - t1
    big = {"value": 2048 * 2048 / dt2, "unit": "cells/s", "cores": 1, "seconds": dt2,
           "sample": "one problem of 2048x2048 (the shape of BASELINE configs[1]) through oracle/nw_ref_py.py",
           "alignment_columns": len(a2[0])}
    out = {"value": done * n * m / dt, "unit": "cells/s", "cores": 1, "kind": "port",
           "sample": "%d problems of %dx%d (BASELINE configs[0] shape) through oracle/nw_ref_py.py, "
                     "%.1f s" % (done, n, m, dt),
           "at_2048x2048": big,
           "c_restatement_cells_per_s": nw_oracle.fill_only_rate(*synth_pair_ids(2048, 2048, 1234)),
           "cpu_model": _cpu_model()}
    # the same port on every host core this process may use: one plain child interpreter per core
    # (the reference itself is single-threaded; SURVEY.md 8d asks for both figures)
    import subprocess
    cores = _usable_cores()
    per = 4
    code = ("import sys; sys.path.insert(0, %r); from oracle import nw_ref_py; "
            "from tools.synth import synth_pair; s0 = int(sys.argv[1]); "
            "[nw_ref_py.perform_alignment(*synth_pair(%d, %d, s0 + k)) for k in range(%d)]" % (REPO, n, m, per))
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(5000 + per * k)]) for k in range(cores)]
    ok = True
    for pr in procs:
        try:
            ok = (pr.wait(timeout=max(1.0, 120.0 - (time.perf_counter() - t0))) == 0) and ok
        except subprocess.TimeoutExpired:
            pr.kill()
            ok = False
    dt = time.perf_counter() - t0
    out["all_cores"] = ({"value": cores * per * n * m / dt, "unit": "cells/s", "cores": cores,
                         "sample": "%d processes x %d problems of %dx%d, %.1f s incl. interpreter start"
                                   % (cores, per, n, m, dt)} if ok else {"error": "a worker failed or timed out"})
    return out


def _usable_cores(cap=16):
    """host cores this job may actually burn: the cgroup CPU quota if there is one, never more
    than the affinity mask, and at most `cap` (a one-GPU box's share of its host)"""
    cores = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(cores, cap))


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def ocr_cpu_baseline(model_seed, no, nlines=60):
    """oracle/ocr_ref_f64.py on ONE host core, and on TWO processes -- the reference runs `ocropus-rpred -Q 2`
    (alignToOCR.py:24, :142-143): two worker processes, one line each at a time.  Both as child interpreters with one
    BLAS thread each (the 100 x 149 products of the loop are far too small for a threaded BLAS, which only slows them
    down); each child times its own loop, interpreter start excluded."""
    import subprocess
    code = ("import sys, os, time; sys.path.insert(0, %r); os.environ['OMP_NUM_THREADS'] = '1'; "
            "os.environ['OPENBLAS_NUM_THREADS'] = '1'; os.environ['MKL_NUM_THREADS'] = '1'; "
            "from oracle import ocr_ref_f64 as R; om = R.synthetic_model(%d, no=%d); s0, k = int(sys.argv[1]), int(sys.argv[2]); "
            "lines = [R.synthetic_line(s0 + i, width=1000) for i in range(k)]; t0 = time.perf_counter(); "
            "[R.recognise(om, xs) for xs in lines]; print(time.perf_counter() - t0)" % (REPO, model_seed, no))

    def run(nproc, per):
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, "-c", code, str(8000 + per * k), str(per)], stdout=subprocess.PIPE, text=True)
                 for k in range(nproc)]
        secs = []
        for pr in procs:
            try:
                out, _ = pr.communicate(timeout=max(1.0, 120.0 - (time.perf_counter() - t0)))
                secs.append(float(out.strip().splitlines()[-1]))
            except (subprocess.TimeoutExpired, ValueError, IndexError):
                pr.kill()
                return None
        return max(secs)
    one = run(1, nlines)
    if one is None:
        return {"error": "the baseline child failed or timed out"}
    out = {"value": nlines / one, "unit": "lines/s", "cores": 1, "kind": "port",
           "sample": "%d lines of width 1000 (T = 1032) through oracle/ocr_ref_f64.py in one process, one BLAS thread, %.1f s"
                     % (nlines, one)}
    two = run(2, nlines // 2)
    out["two_processes"] = ({"value": 2 * (nlines // 2) / two, "unit": "lines/s", "cores": 2,
                             "sample": "2 processes x %d lines of width 1000 (the reference's -Q 2), %.1f s" % (nlines // 2, two)}
                            if two is not None else {"error": "a worker failed or timed out"})
    return out


def bench_ocr(args, rank, precision=None, nlines=None):
    """K3 + K4 + K5 on synthetic lines; precision None = the recogniser's default mode (the mode the
    page pipeline runs and tests/test_page_gpu.py compares with the oracle)."""
    from text_alignment_amd import ocr
    no = 96
    nlines = args.ocr_lines if nlines is None else nlines
    precision = precision or ocr.DEFAULT_PRECISION
    rec = ocr.LineRecognizer(ocr.LineModel.random(7001, no=no), precision=precision)
    lines = synthetic_lines(nlines, 8000 + 7919 * rank)
    st = rec.prepare(lines)
    tsteps = int(st["rows"])
    for _ in range(2):
        rec.run(st)
    torch.cuda.synchronize()
    reps = 3
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
    t0 = time.perf_counter()
    for r in range(reps):
        ev[r][0].record(); rec.run(st, lstm=True, output=False, decode=False)
        ev[r][1].record(); rec.run(st, lstm=False, output=True, decode=False)
        ev[r][2].record(); rec.run(st, lstm=False, output=False, decode=True)
        ev[r][3].record()
    torch.cuda.synchronize()
    # the whole pass as the product enqueues it: for a batch of this size K3 and K4 per length class on side
    # streams, so that the output layer of the short classes runs under the recurrence of the long ones
    # (ocr.LineRecognizer.run) -- wall clock around `reps` passes, device idle before and after
    t0 = time.perf_counter()
    for r in range(reps):
        rec.run(st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    lstm_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    out_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    dec_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))
    f32 = precision == "f32"
    if precision == "f64":
        # float64 mode: hoisted input projection (a f64 GEMM) + the recurrence on v_mfma_f64_16x16x4_f64 (timed
        # together as "lstm": one run() issues both per run of groups); algorithmic flops per timestep as in f32 mode
        tf = tsteps * 238400.0 / (lstm_ms * 1e-3) / 1e12
        traffic, traffic_src = measured_ocr_traffic(nlines, "f64", st["group_size"])
        roof = {"bound": "mfma", "achieved": tf, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": tf / F64_MFMA_PEAK_TF, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "%s (+ lstm_xproj_f64_kernel)" % ("lstm_seq4_f64_kernel" if st["group_size"] == 4 else "lstm_seq_f64_kernel"),
                "flops": "algorithmic: 238400 per timestep, all of them float64 (no padding: 25 x 25 tiles are exactly 400 x 100; the x part is twelve k-steps of four over the 48 inputs, the bias is what the accumulators start from)",
                "peak_is": "f64 matrix peak: v_mfma_f64_16x16x4_f64 at 64 cycles per SIMD, v_mfma_f64_4x4x4_4b_f64 at 16 -- the same 16 "
                           "multiply-adds per cycle and SIMD (measured, profiles/r04_mfma_f64.txt, r05_mfma_f64_4x4.txt)"}
        if st["group_size"] == 4:
            roof.update(f64_issue_floor())
    elif f32:
        # exact f32 arithmetic of this recurrence is bounded by the f32-input MFMA rate; algorithmic
        # flops per timestep: 2 dirs x 4 gates x 100 units x 149 inputs x 2
        tf = tsteps * 238400.0 / (lstm_ms * 1e-3) / 1e12
        kern = "lstm_seq4_kernel" if st["group_size"] == 4 else "lstm_seq_kernel"
        busy, busy_src = measured_mfma_busy(kern) if nlines == 1920 else (None, None)
        traffic, traffic_src = measured_ocr_traffic(nlines, "f32", st["group_size"])
        roof = {"bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": tf / F32_MFMA_PEAK_TF, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "lstm_seq4_kernel" if st["group_size"] == 4 else "lstm_seq_kernel",
                "flops": "algorithmic: 238400 per timestep (the kernel executes 14 % more on padded tiles)",
                "peak_is": "f32-input MFMA (v_mfma_f32_4x4x1_16B_f32 for groups of 4 lines, v_mfma_f32_16x16x4_f32 for "
                           "groups of 16: the same 256 flop per cycle and CU), the instruction the kernel issues",
                "mfma_pipe_busy_rocprof": busy, "mfma_pipe_busy_source": busy_src}
    else:
        # what the 16-bit matrix pipe executes in this mode: 4 products per k-step on 16x16x32 tiles over
        # 160 padded inputs and 112 padded units, both directions: 2 x 7 x 80 MFMAs of 16384 flop per 16
        # lines and timestep -- priced against the pipe it runs on
        tf = tsteps / 16.0 * 2 * 7 * 80 * 16384.0 / (lstm_ms * 1e-3) / 1e12
        traffic, traffic_src = measured_ocr_traffic(nlines, "split", st["group_size"])
        roof = {"bound": "mfma", "achieved": tf, "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": tf / BF16_MFMA_PEAK_TF, "traffic": traffic, "traffic_source": traffic_src, "kernel": "lstm_seq_split_kernel",
                "flops": "executed: 4 split products per k-step, 16-bit operands, padded tiles",
                "peak_is": "dense bf16 / fp16 MFMA (v_mfma_f32_16x16x32_bf16 / _f16), the instructions the kernel issues"}
    return {"lines_per_s": nlines / dt, "timesteps_per_s": tsteps / dt, "lines": nlines,
            "timesteps": tsteps, "classes": no, "precision": precision,
            "dtype": {"f32": "f32", "f64": "f64 recurrence (state, accumulation, gate functions), f32 output layer"}.get(
                precision, "split 16-bit operands (W: bf16 + fp16, a: 3 x bf16 + fp16), f32 accumulate"),
            "ms": {"lstm": lstm_ms, "output_softmax": out_ms, "decode": dec_ms, "pass": 1e3 * dt,
                   "note": "lstm / output_softmax / decode: each kernel launched alone; pass: one run() of all three"},
            "class_split": bool(ocr._split_state["ok"].get(ocr._device_key(rec.device) + (rec.mode,))) and
                           st["n"] >= ocr.CLASS_SPLIT_MIN_LINES and rec.mode == 1,
            "lines_per_workgroup": st["group_size"],
            "roofline": roof}


def measured_ocr_traffic(nlines, precision, group):
    """HBM bytes of one pass of the recurrence kernels of a mode (float64: projection + recurrence) from the separate
    --pmc WRITE_SIZE / FETCH_SIZE passes kept under profiles/ (tools/profile_round.sh); read, not measured in this run."""
    name, path = _profile_file("ocr_hbm_traffic.json", "ocr_f64_hbm_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if "modes" in d:
            e = d["modes"].get("%s_g%d_%d" % (precision, group, nlines))
            if e:
                return float(e["hbm_bytes_per_pass"]), "profiles/" + name
        elif precision == "f64":
            e = d.get(str(nlines))
            if e:
                return float(e["hbm_bytes_per_pass"]), "profiles/" + name
    except (OSError, KeyError, ValueError, TypeError):
        pass
    return None, None


def ocr_mode_agreement():
    """Free-running agreement of the recogniser's modes (f32, split, f64) with oracle/ocr_ref_f64.py, as measured by
    tools/ocr_mode_agreement.py on the GPU box and kept under profiles/ (read, not measured here)."""
    name, path = _profile_file("ocr_mode_agreement.json")
    try:
        with open(path) as f:
            d = json.load(f)
        keep = ("lines", "widths", "logit_err_median", "logit_err_p90", "logit_err_max", "lines_within_1e3",
                "lines_decode_identical", "chars_ref", "chars_agree")
        return {"source": "profiles/" + name, "vs": "oracle/ocr_ref_f64.py, free-running, spec model",
                "by_model_and_mode": {k: {q: v[q] for q in keep} for k, v in d.items()}}
    except (OSError, KeyError, ValueError, TypeError):
        return None


def nw_roofline(batch, kname, fill_ms, tb_ms, traffic, traffic_src):
    """The contract's roofline block for the dominant kernel (algorithmic bytes = 1 B per cell,
    SURVEY.md 8d, over the kernel's mean launch duration against the 8 TB/s HBM peak) -- and, beside
    it, what the numbers mean for THIS kernel: the step (fill + traceback, the metric as 8(d) times
    it) against the same yardstick, the HBM traffic the counters actually see, and the unit that
    binds the two-phase fill, VALU issue."""
    cells = batch.cells
    fill_rate = cells / (fill_ms * 1e-3)
    step_ms = fill_ms + tb_ms
    roof = {"bound": "hbm", "achieved": fill_rate / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": fill_rate / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_src,
            "kernel": kname, "kernel_ms": fill_ms, "traceback_ms": tb_ms,
            "traceback_kernel": batch.traceback_kernel(),
            "algorithmic_bytes_per_cell": 1,
            "step_ms": step_ms, "step_frac": cells / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "step_is": "fill + traceback device time, the metric as SURVEY.md 8(d) defines it; frac is the "
                       "dominant kernel alone"}
    if traffic is not None:
        roof["measured_hbm_GBps"] = traffic / (fill_ms * 1e-3) / 1e9
        roof["measured_hbm_frac"] = roof["measured_hbm_GBps"] / HBM_PEAK_GBS
    if batch.two_phase:
        # the score kernel never writes the algorithmic byte (0.22 B/cell of checkpoints instead): HBM is a
        # yardstick for it, VALU issue is what binds it.  Floor of its instruction mix: 382 VALU instructions
        # per 64 cells of a lane at 1.72 ns per wave-instruction and SIMD (profiles/r05_valu_issue_rates.txt)
        floor_ms = cells / 4096.0 * P1_VALU_PER_BLOCK * P1_NS_PER_VALU * 1e-6 / 1024.0
        roof.update({"binding_unit": "valu-issue", "valu_issue_floor_ms": floor_ms,
                     "valu_issue_frac": floor_ms / fill_ms,
                     "valu_issue_source": "profiles/%s; %d VALU per 64 cells x 64 lanes at %.2f ns per wave-instruction "
                                          "per SIMD" % (_profile_file("valu_issue_rates.txt")[0], P1_VALU_PER_BLOCK, P1_NS_PER_VALU)})
    return roof


def nw_configs(tsc, torch):
    """SURVEY.md 8(d)'s named NW shapes beside the headline batch: device time of fill + traceback
    (median of 5 after 2 warm-ups, events on the launch stream), every DISTINCT problem of each
    batch checked bit-exact against the C oracle."""
    from oracle import nw_oracle
    from tools.synth import synth_pair_ids
    rows = []
    for name, nprob, n, m, distinct in [("headline 1x4096^2", 1, 4096, 4096, 1),
                                        ("headline 64x4096^2", 64, 4096, 4096, 4),
                                        ("C2 1024x2048^2", 1024, 2048, 2048, 8),
                                        ("C4 1x8192^2", 1, 8192, 8192, 1)]:
        uniq = [synth_pair_ids(n, m, 4321 + k) for k in range(distinct)]
        batch = tsc.NWBatch([uniq[k % distinct][0] for k in range(nprob)],
                            [uniq[k % distinct][1] for k in range(nprob)], DEFAULT_SYS)
        for _ in range(2):
            batch.run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record(); batch.run(fill=True, traceback=False)
            e[1].record(); batch.run(fill=False, traceback=True)
            e[2].record()
            torch.cuda.synchronize()
            ts.append((e[0].elapsed_time(e[2]), e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
        total, fill, tb = sorted(ts)[len(ts) // 2]
        res = batch.results()
        want = [nw_oracle.align_ids(t, o, DEFAULT_SYS).tolist() for t, o in uniq]
        ok = all(res[k].tolist() == want[k % distinct] for k in range(nprob))
        rows.append({"config": name, "problems": nprob, "n": n, "m": m,
                     "mode": "two-phase" if batch.two_phase else "one-pass", "traceback_kernel": batch.traceback_kernel(),
                     "ms": total, "fill_ms": fill, "traceback_ms": tb,
                     "cells_per_s": batch.cells / (total * 1e-3),
                     "frac": batch.cells / (total * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "fill_frac": batch.cells / (fill * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "bit_exact": bool(ok), "problems_checked": nprob})
        del batch
        torch.cuda.empty_cache()
    return rows


def nw_end_to_end(tsc, torch, batch_n):
    """"also report ... end-to-end incl. H2D/D2H and Python wrapper".  The call being replaced is
    `perform_alignment(list(transcript), list(ocr), params)` (alignToOCR.py:273), one pair per call, Python lists of
    single characters in and out.  Two figures per batch shape (the headline batch and BASELINE configs[1]):
      lists:  ONE perform_alignment_batch call on host lists -- token encoding, H2D of the codes, both kernels, D2H of
              the alignment columns, the two '_'-marked token lists per pair rebuilt (what a caller of the reference's
              surface gets); the Python objects in and out are the cost here, not the GPU;
      arrays: NWBatch on host id arrays -> host column arrays (H2D + kernels + D2H, no per-token Python objects).
    Median of 3 calls after one warm-up; two pairs of each run compared with the oracle's lists."""
    from oracle import nw_oracle
    from tools.synth import synth_pair_ids
    alphabet = "abcdefghijklmnopqrstuvwxyz "
    rows = []
    for name, nprob, n, m, distinct in [("headline %dx4096^2" % batch_n, batch_n, 4096, 4096, 16),
                                        ("C2 1024x2048^2", 1024, 2048, 2048, 8)]:
        uniq = [synth_pair_ids(n, m, 7321 + k) for k in range(distinct)]
        as_list = [([alphabet[c] for c in t], [alphabet[c] for c in o]) for t, o in uniq]
        pairs = [as_list[k % distinct] for k in range(nprob)]
        cells = float(nprob) * n * m

        def lists():
            return tsc.perform_alignment_batch(pairs, DEFAULT_SYS)

        def arrays():
            b = tsc.NWBatch([uniq[k % distinct][0] for k in range(nprob)], [uniq[k % distinct][1] for k in range(nprob)],
                            DEFAULT_SYS)
            b.run()
            return b.results()
        out = {"config": name, "problems": nprob, "n": n, "m": m}
        for key, fn in (("lists", lists), ("arrays", arrays)):
            fn()
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = fn()
                ts.append(time.perf_counter() - t0)
            dt = sorted(ts)[1]
            out[key] = {"seconds": dt, "cells_per_s": cells / dt}
            if key == "lists":
                ok = all(res[k] == nw_oracle.perform_alignment(*as_list[k % distinct]) for k in (0, nprob - 1))
                out["lists"]["equal_to_oracle_lists"] = bool(ok)
            del res
        rows.append(out)
        torch.cuda.empty_cache()
    return {"shapes": rows,
            "note": "host lists / host arrays in and out of ONE call; the resident rate is `value` (headline) and "
                    "`configs` (C2).  lists: Python token lists as the reference's surface takes and returns them "
                    "(alignToOCR.py:273) -- 8 192 + ~8 800 Python objects per 4096^2 pair; arrays: int32 ids in, uint8 "
                    "alignment columns out"}


def nw_grid_search(tsc, torch):
    """SURVEY.md 8(d)'s secondary run / row N2: the reference's grid search (evaluate_text_alignment.py:134-198) is
    2 187 page-sized alignments -- 3 pages x the 729 scoring systems of :181-188 -- one `perform_alignment` call each
    (:163 via process, ~4 s per call on a CPU core).  Here: ONE launch with a scoring system per problem
    (params_stride = 6).  Device time median of 10 after 3 warm-ups; EVERY problem checked bit-exact against the C oracle."""
    from oracle import nw_oracle
    from tools.nw_configs import grid_systems, time_batch
    from tools.synth import synth_pair_ids
    grid = grid_systems()
    n, m = 800, 900
    pages = [synth_pair_ids(n, m, 4400 + k) for k in range(3)]
    params = np.array(grid * 3, dtype=np.int64)
    nprob = len(params)
    batch = tsc.NWBatch([pages[k // len(grid)][0] for k in range(nprob)],
                        [pages[k // len(grid)][1] for k in range(nprob)], params)
    total, fill, tb = time_batch(torch, batch)
    res = batch.results()
    sample = list(range(nprob))          # every problem (2 187 x 0.72 M cells: ~5 s of the C oracle); until round 5 every 27th,
    ok = all(                            # which on a 3^6 product grid saw ONE value of the three fastest-varying parameters
    res[k].tolist() == nw_oracle.align_ids(pages[k // len(grid)][0], pages[k // len(grid)][1],
                                                     [int(v) for v in params[k]]).tolist() for k in sample)
    out = {"problems": nprob, "n": n, "m": m, "scoring_systems": len(grid), "pages": 3,
           "mode": "two-phase" if batch.two_phase else "one-pass", "traceback_kernel": batch.traceback_kernel(),
           "launches": 1,
           "ms": total, "fill_ms": fill, "traceback_ms": tb, "cells_per_s": batch.cells / (total * 1e-3),
           "frac": batch.cells / (total * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "bit_exact": bool(ok), "problems_checked": len(sample),
           "reference": "evaluate_text_alignment.py:181-198: one perform_alignment call per (page, system)"}
    del batch
    torch.cuda.empty_cache()
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks as a fresh child (one process
    per GPU over RCCL) BEFORE this process has made any GPU call, and hand back its exit code."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=4096, help="NW problems per GPU per step")
    ap.add_argument("--ocr-lines", type=int, default=1920, help="text lines per GPU (64 pages x 30)")
    ap.add_argument("--ocr-lines-large", type=int, default=5760,
                    help="second OCR measurement with more lines than CUs x 16 (0 = skip)")
    ap.add_argument("--no-pipelined", dest="pipelined", action="store_false",
                    help="accepted and ignored (the two-stream leg of rounds 2-3 is gone: fill and traceback are both "
                         "bound by VALU issue, so running them side by side only ever lost, 14.5 against 14.3 ms)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ocr", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the SURVEY 8(d) NW shapes (N = 1 only)")
    ap.add_argument("--pages", type=int, default=64,
                    help="synthetic pages per GPU for the sharded page pipeline and the end-to-end "
                         "process_batch timings (0 = skip)")
    ap.add_argument("--one-pass", action="store_true",
                    help="use the single-pass fill (1 B/cell pointer matrix) instead of the two-phase aligner")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group even at world size 1 (rehearses the RCCL path)")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--no-bind", action="store_true", help="leave the process's cpu affinity as started (A/B of the NUMA binding)")
    ap.add_argument("--page-rows", default="pinned", choices=["pinned", "numpy", "device"],
                    help="pages_sharded leg: where the strips' prepared rows lie (tools/pages_bench.ROWS_INPUT)")
    ap.add_argument("--pages-only", action="store_true",
                    help="rehearsals: a minimal NW step, no OCR / config legs -- the pages_sharded leg is what is looked at")
    args = ap.parse_args()

    if args.pages_only:
        args.batch, args.steps, args.warmup = min(args.batch, 8), 1, 0
        args.no_ocr = args.no_configs = args.no_cpu_baseline = True
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # ranks may outnumber the box's GPUs only in rehearsals (gloo): they then share device 0
    ndev = torch.cuda.device_count()
    dev_index = local if local < ndev else local % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    # before the first kernel launch: this rank -- and every thread it starts -- onto the host cores next to its GPU,
    # torch's intra-op pool down to one thread (text_alignment_amd.sharding.bind_to_gpu_node; silent where the host
    # does not name a NUMA node for the device)
    from text_alignment_amd import sharding as _sh
    try:
        placement = {"bound": False, "reason": "--no-bind"} if args.no_bind else _sh.bind_to_gpu_node(
            dev_index, local_rank=local, local_world=int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    except Exception as exc:        # placement is an optimisation: a host it cannot read must not cost the run
        placement = {"bound": False, "reason": "bind_to_gpu_node raised %r" % (exc,)}
    dist = None
    if world > 1 or (args.force_dist and "RANK" in os.environ):
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)

    from text_alignment_amd import sharding, textSeqCompare as tsc
    batch, uniq = make_nw_batch(tsc, args.batch, args.n, args.m, 1234 + rank * 100000,
                               two_phase=False if args.one_pass else None)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(timed=None):
        if timed is not None:
            timed[0].record()
        batch.run(fill=True, traceback=False)
        if timed is not None:
            timed[1].record()
        batch.run(fill=False, traceback=True)
        if timed is not None:
            timed[2].record()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(ev[k])
    barrier()
    dt = time.perf_counter() - t0
    red_dev = torch.device("cuda", dev_index) if args.backend == "nccl" else torch.device("cpu")

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_sum(xs):
        t = torch.tensor(xs, dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t]

    dt = reduce_max(dt)
    fill_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    tb_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))

    # ---- BASELINE configs[4]: pages sharded over the ranks, one gather of syllable-box records ----
    sharded = None
    if args.pages > 0:
        from tools import pages_bench
        job = pages_bench.setup_sharded(args.pages, rank, world, seed0=100, rows=args.page_rows)
        import gc
        gc.collect()
        gc.freeze()                  # the job's pages and blocks out of the collector's way for the timed passes
        sh_all, mine_s, gather_s, cpu_s, wait_s = [], [], [], [], []
        for _ in range(10):         # median of ten passes, each barrier to barrier over every rank incl. the gather
            barrier()
            tm = {}
            c1, t1 = time.process_time(), time.perf_counter()
            allrec = pages_bench.run_sharded(job, timings=tm)
            cpu_s.append(time.process_time() - c1)
            barrier()
            sh_all.append(reduce_max(time.perf_counter() - t1))
            mine_s.append(tm["pages_s"])
            gather_s.append(tm["gather_s"])
            wait_s.append(tm["device_wait_s"])
        sh_dt = float(np.median(sh_all))
        # per-rank diagnosis, after the timed passes: one pass under the device profiler (every rank: the gather is collective)
        # and ONE all_gather of a fixed vector per rank -- a straggler or a starved rank must be readable from this line
        busy_ms = pages_bench._device_busy_ms(lambda: pages_bench.run_sharded(job))
        cpus_now = sorted(os.sched_getaffinity(0))
        vec = [float(np.median(mine_s)), float(np.median(gather_s)), 1e3 * float(np.median(cpu_s)) / max(len(job["ids"]), 1),
               (busy_ms * 1e-3 / float(np.median(mine_s))) if busy_ms else -1.0, float(len(cpus_now)), float(cpus_now[0]),
               float(cpus_now[-1]), float(-1 if placement.get("numa_node") is None else placement["numa_node"]),
               1.0 if placement.get("bound") else 0.0, float(len(job["ids"])), float(dev_index),
               1e3 * float(np.median(np.array(mine_s) - np.array(wait_s))) / max(len(job["ids"]), 1)]
        mine_t = torch.tensor(vec, dtype=torch.float64, device=red_dev)
        if dist is not None:
            every = torch.zeros(world * len(vec), dtype=torch.float64, device=red_dev)
            dist.all_gather_into_tensor(every, mine_t)
            every = every.cpu().numpy().reshape(world, len(vec))
        else:
            every = mine_t.cpu().numpy().reshape(1, len(vec))
        if rank == 0:
            import hashlib
            heads = allrec[allrec[:, 1] == sharding.HEADER]
            boxes = int((allrec[:, 1] != sharding.HEADER).sum())
            canon = allrec[np.lexsort(allrec.T[::-1])]          # rank order does not matter: sorted by (page, syllable, ...)
            sharded = {"pages": job["total_pages"], "pages_per_gpu": args.pages, "ranks": world,
                       "backend": (args.backend if dist is not None else "none (single process)"),
                       "seconds": sh_dt, "pages_per_s": job["total_pages"] / sh_dt,
                       "lines_per_s": job["total_pages"] * 30 / sh_dt,
                       "models": "half the pages 96 classes (Salzinnes-shaped), half 64 (St-Gall-shaped)",
                       "gathered_records": int(allrec.shape[0]), "syllable_boxes": boxes,
                       "records_sha16": hashlib.sha256(np.ascontiguousarray(canon, dtype=np.int32).tobytes()).hexdigest()[:16],
                       "gather_capacity_records_per_rank": job["capacity"],
                       "gather_ok": bool(sorted(int(v) for v in heads[:, 0]) == list(range(job["total_pages"]))
                                         and int(heads[:, 4].sum()) == boxes),
                       "timing": "median of 10 barrier-to-barrier passes (max over ranks each); shortest %.4f s, longest %.4f s"
                                 % (min(sh_all), max(sh_all)),
                       "input": job["input"],
                       "rank_seconds": {"min": float(every[:, 0].min()), "median": float(np.median(every[:, 0])),
                                        "max": float(every[:, 0].max()),
                                        "is": "each rank's own share (process_batch, results on the host), median of its 10 passes"},
                       "gather_seconds": {"min": float(every[:, 1].min()), "median": float(np.median(every[:, 1])),
                                          "max": float(every[:, 1].max()),
                                          "is": "pack + the ONE gather + unpack on rank 0, per rank; includes waiting for the slowest rank to arrive"},
                       "per_rank": [{"rank": r, "device": int(every[r, 10]), "pages": int(every[r, 9]), "seconds": float(every[r, 0]),
                                     "gather_seconds": float(every[r, 1]), "host_cpu_ms_per_page": float(every[r, 2]),
                                     "host_work_ms_per_page": float(every[r, 11]),
                                     "gpu_busy_frac": (float(every[r, 3]) if every[r, 3] >= 0 else None),
                                     "cpus": "%d in %d..%d" % (every[r, 4], every[r, 5], every[r, 6]),
                                     "numa_node": (int(every[r, 7]) if every[r, 7] >= 0 else None), "bound": bool(every[r, 8])}
                                    for r in range(world)],
                       "per_rank_is": "seconds: the rank's own share; host_work_ms_per_page: (that - the main thread's waits for its GPU) "
                                      "/ pages -- the figure that must stay flat as ranks are added; host_cpu_ms_per_page: process CPU "
                                      "time of ALL threads (runtime helpers and any spin-waiting included)",
                       "placement_rank0": placement,
                       "note": "sharding.process_shard: process_batch per model on this rank's pages + ONE "
                               "gather of [page, syllable, ulx, uly, lrx, lry] records to rank 0"}
            # the timed output under the checkers: two of rank 0's own pages (one per model) rebuilt from the float64
            # recogniser restatement, the C aligner and the reference-pinned glue -- after the timed region
            from tools import pages_check
            got = sharding.records_to_json(allrec, {k: t for k, t in enumerate(job["all_transcripts"])})
            local = {gid: j for j, gid in enumerate(job["ids"])}
            which = [job["ids"][0], job["ids"][min(1, len(job["ids"]) - 1)]]
            chk = pages_check.check_pages({g: got[g] for g in which}, {g: job["pages"][local[g]] for g in which},
                                          {g: job["transcripts"][local[g]] for g in which},
                                          {g: job["models"][local[g]].model for g in which}, pages_bench.PARAMS, sorted(set(which)))
            sharded.update({"pages_checked": chk["pages_checked"], "pages_equal_to_oracle": chk["pages_equal_to_oracle"], "check": chk})
        del job

    ocr_res = None
    if not args.no_ocr:
        ocr_res = bench_ocr(args, rank)                      # the default mode (float64): what pages run and tests compare
        notes = {"f32": "opt-in fast mode: LineRecognizer(model, precision='f32') -- exact float32 arithmetic (a k-ordered "
                        "fmaf chain); narrower than the reference's float64, agreement under `agreement`",
                 "split": "opt-in fastest mode: LineRecognizer(model, precision='split') -- 16-bit matrix cores, split "
                          "operands; agreement under `agreement`",
                 "f64": "LineRecognizer(model, precision='f64') -- the reference's arithmetic type; the mode in which the "
                        "north_star's 1e-3 logit tolerance holds FREE-RUNNING on every line of this (chaotic, "
                        "random-weight) model: tests/test_ocr_gpu.py::test_spec_model_benchmark_widths_free_running_f64"}
        ocr_res["note"] = "default mode (ocr.DEFAULT_PRECISION = %r): %s" % (ocr_res["precision"], notes[ocr_res["precision"]])
        for other in ("f64", "f32", "split"):
            if other == ocr_res["precision"]:
                continue
            alt = bench_ocr(args, rank, precision=other)
            ocr_res["%s_mode" % other] = {
                "lines_per_s": alt["lines_per_s"], "ms": alt["ms"], "roofline": alt["roofline"], "dtype": alt["dtype"],
                "lines_per_workgroup": alt["lines_per_workgroup"], "note": notes[other]}
        if args.ocr_lines_large > args.ocr_lines:
            # three times the lines: several rounds of workgroups per CU (longest first)
            big = bench_ocr(args, rank, nlines=args.ocr_lines_large)
            ocr_res["large_batch"] = {"lines": big["lines"], "precision": big["precision"], "lines_per_s": big["lines_per_s"],
                                      "ms": big["ms"], "roofline_frac": big["roofline"]["frac"],
                                      "lines_per_workgroup": big["lines_per_workgroup"]}
            if big["precision"] != "f32":
                big32 = bench_ocr(args, rank, nlines=args.ocr_lines_large, precision="f32")
                ocr_res["large_batch"]["f32_mode"] = {"lines_per_s": big32["lines_per_s"], "ms": big32["ms"],
                                                      "roofline_frac": big32["roofline"]["frac"],
                                                      "lines_per_workgroup": big32["lines_per_workgroup"]}
        agree = ocr_mode_agreement()
        if agree is not None:
            ocr_res["agreement"] = agree
        if dist is not None:
            ocr_res["lines_per_s_all_gpus"] = reduce_sum([ocr_res["lines_per_s"]])[0]

    pages_res = None
    if args.pages > 0 and not args.no_ocr and world == 1:
        from tools import pages_bench
        pages_res = pages_bench.run(args.pages, seed0=100)

    configs = grid = e2e = None
    if world == 1 and not args.no_configs and not args.one_pass:
        configs = nw_configs(tsc, torch)
        grid = nw_grid_search(tsc, torch)
        e2e = nw_end_to_end(tsc, torch, args.batch)

    if rank == 0:
        # bit-exact check of the timed output against the oracle (checker only): every distinct
        # problem of the batch, and every replica against its original
        from oracle import nw_oracle
        res = batch.results()
        want = [nw_oracle.align_ids(t, o, DEFAULT_SYS).tolist() for t, o in uniq]
        ok = all(res[k].tolist() == want[k % len(uniq)] for k in range(len(res)))
        kname = "nw_score_kernel" if batch.two_phase else "nw_fill_kernel"
        cells_step = batch.cells * world
        fill_rate = batch.cells / (fill_ms * 1e-3)
        traffic, traffic_src = measured_traffic(args.batch, args.n, args.m, kname)
        out = {
            "metric": "nw_dp_cells_per_s", "value": cells_step * args.steps / dt, "unit": "cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "affine-gap NW, %d problems of %dx%d per GPU per step, default "
                                   "scoring [8,-4,-7,-7,-3,0], %s"
                                   % (args.batch, args.n, args.m,
                                      "two-phase (score fill + windowed traceback)" if batch.two_phase
                                      else "one-pass fill + traceback"),
                       "cells_per_step": cells_step, "parallelism": "problems sharded x%d, no data-path collective" % world,
                       "bit_exact_vs_oracle": ok, "problems_checked": len(res),
                       "distinct_problems": len(uniq)},
            "roofline": nw_roofline(batch, kname, fill_ms, tb_ms, traffic, traffic_src),
        }
        if sharded is not None:
            out["pages_sharded"] = sharded
        if configs is not None:
            out["configs"] = configs
        if grid is not None:
            out["grid_search"] = grid
        if e2e is not None:
            out["nw_end_to_end"] = e2e
        if ocr_res is not None:
            out["ocr"] = ocr_res
        if pages_res is not None:
            out["pages_end_to_end"] = pages_res
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline()
            if ocr_res is not None:
                out["ocr"]["cpu_baseline"] = ocr_cpu_baseline(7001, 96)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
