#!/usr/bin/env python3
"""Turn gpurun_out/refresh_<tag>/ (tools/refresh_profiles.sh) into the committed files under profiles/.

  --stage TAG    (on the GPU box) derive profiles/trace_pmc_TAG.json from the PMC summary
  --collect TAG --round rNN   (here) copy kernel stats / PMC summary / bench line into profiles/
"""
import argparse
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stage(tag):
    src = os.path.join(ROOT, "gpurun_out", "refresh_" + tag, "pmc_summary.json")
    d = json.load(open(src))
    # the TIMED instance (k_trace<false, ..>); the profiled command also runs the counting instance k_trace<true, ..>
    keys = [k for k in d if k.startswith("rtd::k_trace<false")]
    assert len(keys) == 1, keys
    key = keys[0]
    t = d[key]
    n = t["calls"]
    rd_raw = t["FETCH_SIZE"] * 1024 / n
    wr = t["WRITE_SIZE"] * 1024 / n
    # every kernel of the profiled step: HBM bytes of the whole step (bench.py: roofline.hbm_frac_rocprof_step)
    # The profiled command (tools/prof_pmc.sh: bench.py --steps 1 --warmup 0) renders the frame TWICE: the timed step and
    # the instrumented counting pass bench.py always adds.  Kernels with a COUNT template flag exist once per pass
    # (k_trace<false, ..> / k_tail<.., false> = the timed step, <true, ..> / <.., true> = the counting pass, left out);
    # every other kernel ran in both passes under one name, so its sums are halved (the two passes do identical work).
    def passes(name):
        if name.startswith(("rtd::k_trace<true", "rtd32::k_trace<true")) or (name.startswith(("rtd::k_tail", "rtd32::k_tail")) and name.rstrip(">").endswith("true")):
            return 0
        if name.startswith(("rtd::k_trace<false", "rtd32::k_trace<false")) or name.startswith(("rtd::k_tail", "rtd32::k_tail")):
            return 1
        return 2
    per_kernel = {k: (v.get("FETCH_SIZE", 0.0) * 2048 / passes(k), v.get("WRITE_SIZE", 0.0) * 1024 / passes(k))
                  for k, v in d.items() if passes(k)}
    step_rd = sum(r for r, _ in per_kernel.values())
    step_wr = sum(w for _, w in per_kernel.values())
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_hash import kernel_hash
    try:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                text=True).stdout.strip() or None
    except Exception:
        commit = None
    out = {
        # which build the counters belong to: bench.py compares these hashes with the library it is running and flags
        # the figure as stale when they differ (the GPU box has no .git: `commit` is then filled in by --collect)
        "kernel_object_hash": kernel_hash("k_trace"), "library_kernels_hash": kernel_hash("k_"), "commit": commit,
        "steps_profiled": 1,
        "hbm_bytes_per_step_all_kernels": step_rd + step_wr,
        "hbm_read_bytes_per_step_all_kernels": step_rd, "hbm_write_bytes_per_step_all_kernels": step_wr,
        "per_kernel_hbm_bytes_per_step": {k: r + w for k, (r, w) in per_kernel.items()},
        "per_kernel_hbm_read_bytes_per_step": {k: r for k, (r, w) in per_kernel.items()},
        "per_kernel_hbm_write_bytes_per_step": {k: w for k, (r, w) in per_kernel.items()},
        "kernel": key, "workload": tag, "launches": n,
        "FETCH_SIZE_KB_sum": t["FETCH_SIZE"], "WRITE_SIZE_KB_sum": t["WRITE_SIZE"],
        "hbm_read_bytes_per_launch_raw": rd_raw,
        "hbm_read_bytes_per_launch_corrected_x2": 2 * rd_raw,
        "hbm_write_bytes_per_launch": wr,
        "hbm_bytes_per_launch": 2 * rd_raw + wr,
        "l2_hit_rate": t.get("l2_hit_rate"),
        "note": "rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum / WRITE_SIZE TCC_MISS_sum TCC_REQ_sum in separate passes with "
                "--kernel-trace only; read side doubled per the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md "
                "(HBM section); Infinity-Cache hits are counted, not excluded",
    }
    json.dump(out, open(os.path.join(ROOT, "profiles", f"trace_pmc_{tag}.json"), "w"), indent=1)


def collect(tag, rnd):
    src = os.path.join(ROOT, "gpurun_out", "refresh_" + tag)
    prof = os.path.join(ROOT, "profiles")
    ks = sorted(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)
    assert ks, "no kernel_stats.csv"   # (gpurun merges into gpurun_out/, so older runs may still be there: newest wins)
    shutil.copy(ks[0], os.path.join(prof, f"{rnd}_kernel_stats_{tag}.csv"))
    shutil.copy(os.path.join(src, "pmc_summary.json"), os.path.join(prof, f"{rnd}_pmc_summary_{tag}.json"))
    shutil.copy(os.path.join(src, "bench_full.json"), os.path.join(prof, f"{rnd}_bench_{tag}.json"))
    tp = json.load(open(os.path.join(src, f"trace_pmc_{tag}.json")))
    if not tp.get("commit"):  # staged on the GPU box (no .git there): the commit whose tree was profiled = HEAD here, if the
        import subprocess     # library's kernels still hash to what was profiled
        sys_path = os.path.join(ROOT, "tools")
        import sys
        sys.path.insert(0, sys_path)
        from kernel_hash import kernel_hash
        if kernel_hash("k_") == tp.get("library_kernels_hash"):
            dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "rustraytracer_amd", "include"],
                                   stdout=subprocess.PIPE, text=True).stdout.strip()
            head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], stdout=subprocess.PIPE, text=True).stdout.strip()
            tp["commit"] = head + ("+uncommitted" if dirty else "")
    json.dump(tp, open(os.path.join(prof, f"trace_pmc_{tag}.json"), "w"), indent=1)
    b = json.load(open(os.path.join(src, "bench_full.json")))
    sb = json.load(open(os.path.join(src, "stats_bench.json")))
    for r in csv.DictReader(open(ks[0])):
        if "k_trace" in r["Name"]:
            print("rocprof k_trace avg ms", float(r["AverageNs"]) / 1e6, "calls", r["Calls"],
                  "| bench-under-rocprof avg_launch_ms", sb["roofline"]["k_trace_detail"]["avg_launch_ms"],
                  "| plain bench avg_launch_ms", b["roofline"]["k_trace_detail"]["avg_launch_ms"])
    print(json.dumps({k: b[k] for k in ("value", "ms_per_step", "roofline", "cpu_baseline")}, indent=1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage")
    ap.add_argument("--collect")
    ap.add_argument("--round", default="r01")
    a = ap.parse_args()
    if a.stage:
        stage(a.stage)
    if a.collect:
        collect(a.collect, a.round)
