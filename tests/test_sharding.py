"""Multi-process (gloo, world_size 2, CPU) test of the page sharding + gather path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from text_alignment_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    costs = [5.0, 1.0, 9.0, 3.0, 7.0, 2.0, 8.0]
    mine = sharding.shard_indices(costs, world, rank)
    recs = []
    for pid in mine:                       # page pid emits pid + 1 boxes (rank 1 may hold fewer)
        for k in range(pid + 1):
            recs.append((pid, k, 10 * pid + k, 90, 10 * pid + k + 8, 130))
    local = np.array(recs, dtype=np.int32).reshape(-1, 6)
    allrec = sharding.gather_records(local)
    empty = sharding.gather_records(np.zeros((0, 6), np.int32) if rank == 1 else local)
    # fixed-capacity form: one gather to rank 0, no size exchange, asynchronous
    packed = sharding.pack_records_device(local, 64, torch.device("cpu"))
    work, out = sharding.gather_to_root(packed, async_op=True)
    work.wait()
    if rank == 0:
        assert sharding.unpack_gathered(out).tolist() == allrec.tolist()
    else:
        assert out is None
    if rank == 0:
        q.put((mine, allrec.tolist(), empty.shape[0]))
    else:
        q.put((mine, None, None))
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shards = sorted((g[0] for g in got), key=lambda s: s[0])
    assert sorted(shards[0] + shards[1]) == list(range(7))            # a partition of the pages
    assert sorted([shards[0], shards[1]]) == sorted([[2, 4, 3, 1], [6, 0, 5]])   # heaviest first, dealt
    allrec = [g[1] for g in got if g[1] is not None][0]
    assert len(allrec) == sum(p + 1 for p in range(7))
    by_page = {}
    for r in allrec:
        by_page.setdefault(r[0], []).append(r)
    for pid, rows in by_page.items():
        assert [r[1] for r in rows] == list(range(pid + 1))
        assert all(r[2] == 10 * pid + r[1] for r in rows)
    n_empty = [g[2] for g in got if g[2] is not None][0]
    assert n_empty == sum(p + 1 for p in [2, 4, 3, 1]) or n_empty == sum(p + 1 for p in [6, 0, 5])


def test_single_process_passthrough_and_cost():
    from text_alignment_amd import sharding
    local = np.arange(12, dtype=np.int32).reshape(2, 6)
    assert np.array_equal(sharding.gather_records(local), local)
    assert sharding.shard_indices([1, 5, 3], 1, 0) == [1, 2, 0]
    assert sharding.page_cost([100, 200], 800) > sharding.page_cost([100], 800)


def _fake_process_batch(pages, transcripts, model, seq_align_params=None, indices_out=None, parallel=2,
                        arrays_out=None):
    """stands in for alignToOCR.process_batch on a CPU-only box: boxes are a function of the page
    and the transcript only (every second non-empty syllable gets one), so any rank computes the
    same boxes for the same page"""
    from text_alignment_amd import alignToOCR as atocr, latinSyllabification as latsyl
    out = []
    models = model if isinstance(model, (list, tuple)) else [model] * len(pages)
    for pg, tr, model in zip(pages, transcripts, models):
        if pg.get("bad"):                      # e.g. page.prepared_lines' 'empty or constant text-line image'
            raise ValueError("empty or constant text-line image")
        if pg.get("fatal"):                    # e.g. _native.check after a kernel fault
            raise RuntimeError("ta_nw2_batch failed (-3): hipErrorLaunchFailure")
        syls = [s for s in latsyl.syllabify_text(tr) if len(s) >= 1]
        boxes, idx = [], []
        for k, s in enumerate(syls):
            if (k + pg["seed"]) % 2 == 0 and not pg["empty"]:
                boxes.append(atocr.CharBox(s, (10 * k + pg["seed"], 100 + model["shift"]),
                                           (10 * k + 8 + pg["seed"], 140 + model["shift"])))
                idx.append(k)
        if indices_out is not None:
            indices_out.append(idx)
        out.append((boxes, None, pg["peaks"], []))
    return out


_TEXTS = ["dominus dixit ad me", "filius meus es tu alleluia", "gloria patri et filio", "amen",
          "laudate eum omnes gentes", "quoniam confirmata est super nos misericordia eius", "et"]


def _pages_worker(rank, world, port, q, bad_page=None, capacity_bug=False, fatal_page=None):
    import warnings
    import torch.distributed as dist
    from text_alignment_amd import alignToOCR as atocr, sharding
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    atocr.process_batch = _fake_process_batch
    sharding.estimate_cost = lambda pg, tr: float(len(tr))
    pages = [{"seed": k, "empty": k == 3, "peaks": [100, 220 + k, 340, 470 + 3 * k], "bad": k == bad_page,
              "fatal": k == fatal_page} for k in range(len(_TEXTS))]
    if capacity_bug:                           # a capacity that one rank's records exceed
        sharding.record_capacity = lambda tr: 2
    warnings.simplefilter("ignore")
    models = [{"shift": 0} if k % 2 == 0 else {"shift": 1000} for k in range(len(_TEXTS))]
    models = [models[0] if k % 2 == 0 else models[1] for k in range(len(_TEXTS))]     # two distinct model objects
    try:
        out = sharding.process_pages(pages, _TEXTS, models, None)
    except ValueError as exc:                  # raised AFTER the collective: no rank is left waiting in it
        out = "ValueError: %s" % exc
    except RuntimeError as exc:                # a failed native call: re-raised AFTER the collective
        out = "RuntimeError: %s" % exc
    q.put((rank, out))
    if world > 1:
        dist.destroy_process_group()


def test_process_pages_world2_equals_single_process():
    """sharding.process_pages -- the driver BASELINE configs[4] runs -- with two gloo ranks: rank 0
    receives every page's JSON through the one fixed-capacity gather (two models, a page without
    boxes, uneven shards), identical to what a single process computes; rank 1 gets None."""
    ctx = mp.get_context("spawn")
    results = {}
    for world in (1, 2):
        port = _free_port()
        q = ctx.Queue()
        procs = [ctx.Process(target=_pages_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = dict(q.get(timeout=120) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        results[world] = got
    single = results[1][0]
    assert sorted(single) == list(range(len(_TEXTS)))
    assert single[3]["syl_boxes"] == [] and single[3]["median_line_spacing"] == np.quantile(np.diff([100, 223, 340, 479]), 0.75)
    assert sum(len(v["syl_boxes"]) for v in single.values()) > 15
    assert single[1]["syl_boxes"][0]["ul"][1] == 1100 and single[0]["syl_boxes"][0]["ul"][1] == 100
    assert results[2][1] is None
    assert results[2][0] == single


def _run_pages(world, **kw):
    ctx = mp.get_context("spawn")
    port = _free_port()
    q = ctx.Queue()
    procs = [ctx.Process(target=_pages_worker, args=(r, world, port, q), kwargs=kw) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_bad_page_on_one_rank_is_skipped_and_nobody_hangs():
    """A page that raises on ITS rank (here: page 5, a blank strip) must cost that page only: the rank
    retries its batch page by page, sends a status record for the page that still fails and enters
    the one collective like every other rank; rank 0 reports the page as None (the reference's loop
    prints 'OCRopus failed! Skipping current file.' and goes on, alignToOCR.py:240-243, :430-431)
    and every other page exactly as a run without the failure."""
    good = _run_pages(1)[0]
    for world in (1, 2):
        got = _run_pages(world, bad_page=5)
        out = got[0]
        assert sorted(out) == list(range(len(_TEXTS)))
        assert out[5] is None
        assert all(out[k] == good[k] for k in range(len(_TEXTS)) if k != 5)
        if world == 2:
            assert got[1] is None


def test_a_failed_native_call_is_not_a_skipped_page():
    """What is NOT a page to skip (the reference skips only on the OCR step's own failure,
    alignToOCR.py:240-243): a RuntimeError out of a native call -- a kernel fault, a sticky HIP error -- on a
    rank.  That rank must not report 'skipped pages' and return normally: it remembers the error, reports the
    pages it did not finish as failed, still enters the one gather (nobody hangs) and re-raises afterwards.  The
    other rank's pages arrive as usual."""
    from text_alignment_amd import sharding
    costs = [float(len(t)) for t in _TEXTS]
    shards, _ = sharding.shard_plan(costs, _TEXTS, 2)
    fatal = shards[1][0]                                   # a page of rank 1
    good = _run_pages(1)[0]
    got = _run_pages(2, fatal_page=fatal)
    assert isinstance(got[1], str) and got[1].startswith("RuntimeError") and "hipErrorLaunchFailure" in got[1]
    out = got[0]                                           # rank 0 finished: its own pages, and None for rank 1's unfinished ones
    assert sorted(out) == list(range(len(_TEXTS)))
    assert out[fatal] is None
    assert all(out[k] == good[k] for k in shards[0])
    assert all(out[k] is None or out[k] == good[k] for k in shards[1])
    # alone (one process): the error reaches the caller
    got = _run_pages(1, fatal_page=fatal)
    assert isinstance(got[0], str) and got[0].startswith("RuntimeError")
    # and a bad ARGUMENT to the library (a programming error) is not a page error either
    from text_alignment_amd import _native
    assert issubclass(_native.NativeArgumentError, ValueError)
    assert not any(issubclass(_native.NativeArgumentError, e) and e is not ValueError for e in sharding._page_errors())


def test_capacity_overflow_raises_after_the_collective_on_every_rank_involved():
    """Records beyond the agreed capacity (a capacity bug): the offending rank still enters the
    gather -- with a buffer that says how many records it had -- and raises afterwards; rank 0
    raises when it unpacks.  No rank waits for a peer that has already left."""
    got = _run_pages(2, capacity_bug=True)
    assert isinstance(got[0], str) and got[0].startswith("ValueError")
    assert got[1] is None or (isinstance(got[1], str) and got[1].startswith("ValueError"))


def test_capacity_is_an_upper_bound_and_plan_is_deterministic():
    from text_alignment_amd import sharding, latinSyllabification as latsyl
    # tokens made of whitespace other than ' ' are syllables of their own (words split on ' ' only, a
    # vowel-less word comes back whole): they can get a box and must be counted
    for tr in ["a \n b", "a\tb \t c", "dominus \u00a0 deus", "\n", " \t "]:
        nsyl = len([s for s in latsyl.syllabify_text(tr) if len(s) >= 1])
        assert sharding.record_capacity(tr) == 1 + nsyl
    assert sharding.record_capacity("a \n b") == 4
    for tr in _TEXTS + ["", "  ", "a e i o u"]:
        nsyl = len([s for s in latsyl.syllabify_text(tr) if len(s) >= 1])
        assert sharding.record_capacity(tr) >= 1 + nsyl
    costs = [float(len(t)) for t in _TEXTS]
    shards, cap = sharding.shard_plan(costs, _TEXTS, 3)
    assert sorted(k for sh in shards for k in sh) == list(range(len(_TEXTS)))
    assert cap == max(sum(sharding.record_capacity(_TEXTS[k]) for k in sh) for sh in shards)
    with pytest.raises(ValueError):
        sharding.pack_records_device(np.zeros((5, 6), np.int32), 4, torch.device("cpu"))


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_on_one_gpu():
    """The multi-GPU path of bench.py rehearsed by a test, not by hand (the pool offers one GPU per box and the
    driver's 8-GPU run is the only place the RCCL path runs): `python bench.py --gpus 2 --backend gloo ...` as a fresh
    child -- it starts its own `torch.distributed.run` before touching the GPU; both ranks share device 0 -- times the
    NW step per rank and runs BASELINE configs[4]'s leg, sharding.process_shard with its ONE gather.  Asserted: two
    ranks, every page's header gathered, and the gathered records equal to a single process's over the same pages
    (the loop being sharded: alignToOCR.py:407-438)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--batch", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-configs", "--no-ocr"]
    lines = {}
    for name, extra in (("two", ["--gpus", "2", "--backend", "gloo", "--pages", "4"]),
                        ("one", ["--gpus", "1", "--pages", "8"])):
        r = subprocess.run([sys.executable, os.path.join(repo, "bench.py")] + extra + common, cwd=repo,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        last = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        lines[name] = json.loads(last)
    two, one = lines["two"], lines["one"]
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["bit_exact_vs_oracle"] and one["config"]["bit_exact_vs_oracle"]
    assert two["value"] > 0 and two["config"]["cells_per_step"] == 2 * one["config"]["cells_per_step"]
    ps2, ps1 = two["pages_sharded"], one["pages_sharded"]
    assert ps2["ranks"] == 2 and ps2["backend"] == "gloo" and ps2["pages"] == 8 and ps1["pages"] == 8
    assert ps2["gather_ok"] and ps1["gather_ok"]
    assert ps2["syllable_boxes"] == ps1["syllable_boxes"] > 0
    assert ps2["records_sha16"] == ps1["records_sha16"]
    # the timed output under the checkers, and the per-rank diagnosis the single line of an 8-GPU run has to carry
    assert ps2["pages_checked"] >= 1 and ps2["pages_equal_to_oracle"] == ps2["pages_checked"]
    assert ps1["pages_checked"] >= 1 and ps1["pages_equal_to_oracle"] == ps1["pages_checked"]
    assert "page-locked" in ps2["input"]
    assert [r["rank"] for r in ps2["per_rank"]] == [0, 1] and sum(r["pages"] for r in ps2["per_rank"]) == 8
    for r in ps2["per_rank"]:
        assert 0 < r["seconds"] <= ps2["rank_seconds"]["max"] and r["gather_seconds"] >= 0 and r["host_cpu_ms_per_page"] > 0
        assert r["gpu_busy_frac"] is None or 0 < r["gpu_busy_frac"] <= 1.5
        assert set(r) >= {"device", "cpus", "numa_node", "bound"}
    assert ps2["rank_seconds"]["min"] <= ps2["rank_seconds"]["median"] <= ps2["rank_seconds"]["max"] <= ps2["seconds"] * 1.5
    assert set(ps2["placement_rank0"]) >= {"bound", "numa_node", "cpus", "ncpus", "reason"}


def test_binding_plan_slices_a_numa_node_among_its_ranks():
    """sharding.plan_binding (pure): the cpus of the GPU's NUMA node that the process may use, an equal contiguous slice per
    local rank on that node; the whole node when a slice would be too small; None when the host names no node."""
    from text_alignment_amd import sharding as sh
    node_of = {0: 0, 1: 0, 2: 0, 3: 0, 4: 1, 5: 1, 6: 1, 7: 1}
    cpus_of = {0: set(range(0, 64)) | set(range(128, 192)), 1: set(range(64, 128)) | set(range(192, 256))}
    every = set(range(256))
    got = [sh.plan_binding(node_of, cpus_of, every, r, 8) for r in range(8)]
    assert all(len(g) == 32 for g in got)
    assert set().union(*got[:4]) == cpus_of[0] and set().union(*got[4:]) == cpus_of[1]
    assert all(not (set(got[a]) & set(got[b])) for a in range(8) for b in range(a))
    # with the core map, a slice is whole cores: cpu c and its SMT sibling c + 128 always land in the same rank's slice
    core_of = {c: c % 128 for c in range(256)}
    smt = [sh.plan_binding(node_of, cpus_of, every, r, 8, core_of=core_of) for r in range(8)]
    assert all(len(g) == 32 and {c % 128 for c in g} == {c % 128 for c in g if c < 128} and len({c % 128 for c in g}) == 16 for g in smt)
    assert all(not (set(smt[a]) & set(smt[b])) for a in range(8) for b in range(a))
    # a container that owns 16 cpus of node 0 only: ranks on node 0 share them (slices of 4 are the floor), node 1 has none
    few = set(range(0, 12))
    assert sh.plan_binding(node_of, cpus_of, few, 1, 8) == sorted(few)
    assert sh.plan_binding(node_of, cpus_of, few, 5, 8) is None
    assert sh.plan_binding({0: None}, {}, every, 0, 1) is None
    assert sh._fmt_cpus({0, 1, 2, 3, 8, 9, 11}) == "0-3,8-9,11" and sh._cpulist("0-3,8-9,11\n") == {0, 1, 2, 3, 8, 9, 11}
    # no GPU here: the helper says so and leaves the mask alone
    import os
    before = os.sched_getaffinity(0)
    import torch
    if not torch.cuda.is_available():
        out = sh.bind_to_gpu_node()
        assert out["bound"] is False and out["reason"] == "no GPU" and os.sched_getaffinity(0) == before


@pytest.mark.gpu
def test_bench_rccl_path_at_world_size_one():
    """The path `bench.py --gpus 8` takes -- `torch.distributed.run`, backend "nccl" (= RCCL), device tensors through
    `dist.gather` in sharding.gather_to_root, the all_reduce's of the timing on a device tensor -- executed once on the
    one GPU there is: world size 1, launched exactly as the driver launches ranks.  Until round 5 every rehearsal of the
    gather ran on gloo with host tensors; the first RCCL execution must not be the driver's 8-GPU run.  Asserted: the
    backend really is nccl, the gather is complete, and the records equal a plain single process's over the same pages
    (the loop being sharded: alignToOCR.py:407-438)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--batch", "8", "--steps", "1", "--warmup", "0", "--pages", "4", "--no-cpu-baseline", "--no-configs", "--no-ocr"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    lines = {}
    for name, cmd in (("rccl", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                                "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(repo, "bench.py"),
                                "--force-dist", "--backend", "nccl"]),
                      ("plain", [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1"])):
        r = subprocess.run(cmd + common, cwd=repo, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines[name] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    rccl, plain = lines["rccl"], lines["plain"]
    assert rccl["n_gpus"] == 1 and rccl["config"]["bit_exact_vs_oracle"]
    ps, pp = rccl["pages_sharded"], plain["pages_sharded"]
    assert ps["backend"] == "nccl" and ps["ranks"] == 1 and ps["pages"] == 4
    assert ps["gather_ok"] and pp["gather_ok"]
    assert ps["syllable_boxes"] == pp["syllable_boxes"] > 0
    assert ps["records_sha16"] == pp["records_sha16"]
