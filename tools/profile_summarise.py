"""Turns the rocprofv3 CSVs written by tools/profile_round.sh into the small summaries kept under
profiles/: per-kernel mean WRITE_SIZE / FETCH_SIZE and HBM bytes per launch
(WRITE_SIZE*1024 + 2*FETCH_SIZE*1024: counter unit KiB, gfx950 FETCH_SIZE correction, see
MI355X_MICROARCH.md HBM section), and a copy of the kernel-stats table."""
import csv
import glob
import json
import os
import shutil
import sys


def counter_means(path, counter, only="nw_"):
    acc = {}
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"]
                if only and only not in name:
                    continue
                s = acc.setdefault(name, [0.0, 0])
                s[0] += float(row["Counter_Value"])
                s[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


def _kept(path, key):
    """entries of an existing summary that a PARTIAL collection (one part of tools/profile_round.sh re-run after a kernel
    changed) does not replace: the new entries are laid over them"""
    if not os.path.exists(path):
        return {}
    with open(path) as fh:
        return json.load(fh).get(key, {})


def main():
    rnd, out = sys.argv[1], sys.argv[2]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(repo, "profiles")
    for sub in sorted(glob.glob(os.path.join(out, "kt_*"))):
        if not os.path.isdir(sub):
            continue
        stats = glob.glob(os.path.join(sub, "**", "*kernel_stats.csv"), recursive=True)
        if stats:
            tag = os.path.basename(sub)[3:]
            name = "nw_headline" if tag == "nw" else tag
            shutil.copy(stats[0], os.path.join(prof, "%s_kernel_stats_%s.csv" % (rnd, name)))
    for mode, fname, desc in (("two", "%s_nw2_hbm_traffic.json", "two-phase aligner"),
                              ("one", "%s_nw_hbm_traffic.json", "one-pass aligner (--one-pass)")):
        wr = counter_means(os.path.join(out, mode + "_WRITE_SIZE"), "WRITE_SIZE")
        rd = counter_means(os.path.join(out, mode + "_FETCH_SIZE"), "FETCH_SIZE")
        if not wr and not rd:           # a partial collection (tools/profile_round.sh <round> f64): the file of the full one stays
            continue
        kernels = {}
        for k in sorted(set(wr) | set(rd)):
            w, r = wr.get(k, 0.0), rd.get(k, 0.0)
            kernels[k] = {"WRITE_SIZE_KiB_mean": w, "FETCH_SIZE_KiB_mean": r,
                          "hbm_bytes_per_launch": w * 1024 + 2 * r * 1024}
        doc = {"command": "rocprofv3 --pmc WRITE_SIZE (and, separately, FETCH_SIZE) --output-format csv -- python3 "
                          "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined --no-configs --no-ocr --pages 0%s   [4096 problems of "
                          "4096x4096, %s]" % (" --one-pass" if mode == "one" else "", desc),
               "units": "counter values are KiB (x1024 -> bytes); FETCH_SIZE is doubled for wide coalesced reads "
                        "per MI355X_MICROARCH.md (HBM section); WRITE_SIZE is exact for 16-B-per-lane streaming stores",
               "config": {"batch": 4096, "n": 4096, "m": 4096}, "kernels": kernels}
        with open(os.path.join(prof, fname % rnd), "w") as fh:
            json.dump(doc, fh, indent=1)
    # float64 recogniser: HBM bytes per launch of the projection and the recurrence kernels (1 920 lines), both group sizes
    f64 = {}
    for g in (4, 16):
        wr = counter_means(os.path.join(out, "ocr_f64g%d_WRITE_SIZE" % g), "WRITE_SIZE", only="lstm_")
        rd = counter_means(os.path.join(out, "ocr_f64g%d_FETCH_SIZE" % g), "FETCH_SIZE", only="lstm_")
        if not wr and not rd:
            continue
        kernels = {}
        for k in sorted(set(wr) | set(rd)):
            w, r = wr.get(k, 0.0), rd.get(k, 0.0)
            kernels[k] = {"WRITE_SIZE_KiB_mean": w, "FETCH_SIZE_KiB_mean": r, "hbm_bytes_per_launch": w * 1024 + 2 * r * 1024}
        f64["groups_of_%d" % g] = {"kernels": kernels,
                                  "hbm_bytes_per_pass": sum(v["hbm_bytes_per_launch"] for k, v in kernels.items()
                                                            if "xproj_f64" in k or "seq_f64" in k or "seq4_f64" in k)}
    # every mode's recurrence kernels in one file (bench.py: ocr.*.roofline.traffic)
    modes = {}
    for prec, g, pick in (("f64", 4, ("xproj_f64", "seq4_f64")), ("f64", 16, ("xproj_f64", "seq_f64_kernel")),
                          ("f32", 4, ("lstm_seq4_kernel",)), ("split", 16, ("lstm_seq_split_kernel",))):
        wr = counter_means(os.path.join(out, "ocr_%sg%d_WRITE_SIZE" % (prec, g)), "WRITE_SIZE", only="lstm_")
        rd = counter_means(os.path.join(out, "ocr_%sg%d_FETCH_SIZE" % (prec, g)), "FETCH_SIZE", only="lstm_")
        ks = {k: wr.get(k, 0.0) * 1024 + 2 * rd.get(k, 0.0) * 1024 for k in set(wr) | set(rd) if any(p in k for p in pick)}
        if ks:
            modes["%s_g%d_1920" % (prec, g)] = {"kernels": ks, "hbm_bytes_per_pass": sum(ks.values())}
    if modes:
        modes = dict(_kept(os.path.join(prof, "%s_ocr_hbm_traffic.json" % rnd), "modes"), **modes)
        with open(os.path.join(prof, "%s_ocr_hbm_traffic.json" % rnd), "w") as fh:
            json.dump({"command": "TA_OCR_CLASS_SPLIT=0 TA_OCR_F64_PIPE=0 TA_OCR_GROUP=<g> rocprofv3 --pmc WRITE_SIZE (and, separately, "
                                  "FETCH_SIZE) --output-format csv -- python3 tools/ocr_only.py 1920 <mode>",
                       "units": "bytes per launch = WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 (counter unit KiB; FETCH_SIZE doubled per "
                                "MI355X_MICROARCH.md, HBM section); summed over the mode's recurrence kernels (float64: projection + recurrence)",
                       "modes": modes}, fh, indent=1)
    if f64:
        doc = {"command": "TA_OCR_F64_PIPE=0 TA_OCR_GROUP=<4|16> rocprofv3 --pmc WRITE_SIZE (and, separately, FETCH_SIZE) "
                          "--output-format csv -- python3 tools/ocr_only.py 1920 f64",
               "units": "counter values are KiB (x1024 -> bytes); FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section)",
               "algorithmic": "Gx: 2 directions x 2 763 143 rows x 400 doubles = 17.7 GB written by the projection and read "
                              "once by the recurrence; x: 0.53 GB read; hout: 2.2 GB written",
               "1920": {"hbm_bytes_per_pass": f64.get("groups_of_4", f64.get("groups_of_16"))["hbm_bytes_per_pass"],
                        "is": "projection + recurrence kernels of the product's choice (groups of four lines)"}}
        doc.update(f64)
        with open(os.path.join(prof, "%s_ocr_f64_hbm_traffic.json" % rnd), "w") as fh:
            json.dump(doc, fh, indent=1)
    # matrix-pipe counters of the recogniser kernels
    kernels = {}
    for prec in ("split", "f32", "f32g4", "f64", "f64g4"):
        path = os.path.join(out, "ocr_pmc_" + prec)
        if not os.path.isdir(path):
            continue
        names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES",
                 "GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU"]
        per = {}
        for c in names:
            for k, v in counter_means(path, c, only="lstm_").items():
                per.setdefault(k, {})[c] = v
        for k, d in per.items():
            if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_INSTS_MFMA"):
                d["mfma_pipe_utilisation"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * d["GRBM_GUI_ACTIVE"] / 8.0)
                d["busy_cycles_per_mfma"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_INSTS_MFMA"]
            kernels["%s [precision=%s]" % (k, prec)] = d
    if kernels:
        kernels = dict(_kept(os.path.join(prof, "%s_ocr_pmc_mfma.json" % rnd), "kernels"), **kernels)
        with open(os.path.join(prof, "%s_ocr_pmc_mfma.json" % rnd), "w") as fh:
            json.dump({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU "
                                  "SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -- python3 "
                                  "tools/ocr_only.py 1920 <split|f32>   (counter pass only; TA_OCR_CLASS_SPLIT=0, TA_OCR_GROUP=16, "
                                  "and =4 for the f32g4 entry: the 4-line recurrence kernel)",
                       "reading": "GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES"
                                  " / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)",
                       "kernels": kernels}, fh, indent=1)
    # SQ counters of the NW kernels (headline batch) and the wait counters of the four-line recurrence kernel
    for sub, fname, only, note in (
            ("nw_pmc_sq", "%s_nw_pmc_sq.json", "nw_",
             "SQ_* counters count quad-cycles (MI355X_MICROARCH.md); VALU issue share = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES; "
             "instructions per wave-cycle etc. follow from the raw means"),
            ("ocr_pmc_f32g4_waits", "%s_ocr_pmc_waits_f32g4.json", "lstm_",
             "SQ_WAIT_ANY = wave parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY = issue stall, SQ_WAIT_INST_LDS = LDS issue stall; "
             "WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES")):
        path = os.path.join(out, sub)
        if not os.path.isdir(path):
            continue
        per = {}
        for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    if only not in row["Kernel_Name"]:
                        continue
                    d = per.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], [0.0, 0])
                    d[0] += float(row["Counter_Value"]); d[1] += 1
        kernels = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in per.items()}
        for k, d in kernels.items():
            if d.get("SQ_WAVE_CYCLES"):
                for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS"):
                    if c in d:
                        d[c + "_share_of_wave_cycles"] = d[c] / d["SQ_WAVE_CYCLES"]
        with open(os.path.join(prof, fname % rnd), "w") as fh:
            json.dump({"command": "rocprofv3 --pmc <counters below> --output-format csv -- see tools/profile_round.sh (" + sub + ")",
                       "reading": note, "kernels": kernels}, fh, indent=1)
    print("profiles updated")


if __name__ == "__main__":
    main()
