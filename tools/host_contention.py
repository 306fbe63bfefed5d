"""Host-contention rehearsal of BASELINE configs[4] on ONE GPU: `bench.py --gpus N --backend gloo --pages-only --pages P`
for N = 1, 2, 4 -- N ranks as the driver's launcher starts them, all sharing device 0 -- and the host CPU milliseconds
each rank spends per page.  This is NOT a scaling number (the ranks share one GPU, so pages/s cannot grow); it answers
one question the 8-GPU run depends on and this pool cannot otherwise show: does a rank's HOST cost per page stay flat
when more ranks run beside it on the same socket?  (The loop being sharded: reference alignToOCR.py:407-438.)

    python tools/host_contention.py [pages per rank = 16] [rows = pinned|numpy]   ->  one JSON object on stdout
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    pages = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    rows = sys.argv[2] if len(sys.argv) > 2 else "pinned"
    out = {"what": "host work ms per page per rank (wall of its share minus its waits for the GPU), N ranks sharing ONE GPU (gloo): "
                   "a host-contention rehearsal, NOT a scaling curve",
           "pages_per_rank": pages, "rows": rows, "runs": []}
    for n in (1, 2, 4):
        cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--backend", "gloo", "--pages-only",
               "--pages", str(pages), "--page-rows", rows]
        r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            out["runs"].append({"ranks": n, "error": r.stderr[-600:]})
            continue
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        ps = line["pages_sharded"]
        out["runs"].append({"ranks": n, "host_work_ms_per_page": [round(p["host_work_ms_per_page"], 3) for p in ps["per_rank"]],
                            "host_cpu_ms_per_page": [round(p["host_cpu_ms_per_page"], 3) for p in ps["per_rank"]],
                            "rank_seconds": [round(p["seconds"], 4) for p in ps["per_rank"]],
                            "cpus": [p["cpus"] for p in ps["per_rank"]], "bound": [p["bound"] for p in ps["per_rank"]],
                            "pages_per_s_all_ranks_one_gpu": round(ps["pages_per_s"], 1),
                            "pages_equal_to_oracle": "%d / %d" % (ps["pages_equal_to_oracle"], ps["pages_checked"])})
    base = out["runs"][0].get("host_work_ms_per_page", [None])[0]
    if base:
        out["worst_host_work_over_one_rank"] = max(max(r.get("host_work_ms_per_page", [0])) for r in out["runs"]) / base
    print(json.dumps(out))


if __name__ == "__main__":
    main()
