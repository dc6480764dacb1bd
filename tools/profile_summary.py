#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/collect_profiles.sh into the summaries kept under profiles/:

  profiles/<tag>_<config>.txt     per kernel: dispatches, time, FETCH_SIZE / WRITE_SIZE, SQ ratios, per frame
  profiles/<tag>_pmc_hbm.txt      the calibration of the two traffic counters and the per-frame totals
  profiles/hbm_traffic.json       HBM bytes per frame for each configuration, stamped with a hash of the kernel sources
                                  (bench.py reads it, and ignores entries taken with other kernels)

Usage: profile_summary.py <gpurun_out/dir> <tag>

FETCH_SIZE and WRITE_SIZE are in KiB.  The calibration runs (tools/hbm_calib.hip: 1 GiB streamed in or
out in this code's access widths) give the factor between counter and bytes; reads are scaled by the
factor measured for 8-byte loads, writes by the one for 8-byte stores."""
import collections, csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(ROOT, "profiles")
GIB_KB = float(1 << 20)


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n if len(n) <= 64 else n[:61] + "..."


def counters(path):
    """{kernel: {counter: sum}}, {kernel: dispatches}, {kernel: (vgpr, scratch, lds)}"""
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    meta = {}
    if not files:
        return tot, {}, meta
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
        meta[k] = (r["VGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"])
    return tot, {k: len(v) for k, v in seen.items()}, meta


def by_pass(path, counter):
    """Counter sums of the frame's two passes: dispatches up to k_aa_detect belong to pass 1, the rest to the AA resolve."""
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    out = {"pass1": 0.0, "aa": 0.0}
    if not files:
        return out
    rows = sorted((r for r in csv.DictReader(open(files[0])) if r["Counter_Name"] == counter), key=lambda r: int(r["Dispatch_Id"]))
    phase = "pass1"
    for r in rows:
        out[phase] += float(r["Counter_Value"])
        if short(r["Kernel_Name"]).startswith("k_aa_detect"):
            phase = "aa"
    return out


def calibration():
    fetch, _, _ = counters(os.path.join(src, "calib", "fetch"))
    write, _, _ = counters(os.path.join(src, "calib", "write"))
    rows = []
    for k in sorted(set(fetch) | set(write)):
        if not (k.startswith("k_read") or k.startswith("k_write") or k.startswith("k_atomic64")):
            continue
        rows.append((k, fetch[k].get("FETCH_SIZE", 0.0), write[k].get("WRITE_SIZE", 0.0)))
    f_read = next((GIB_KB / f for k, f, w in rows if k.startswith("k_read<double>") and f > 0), 2.0)
    f_write = next((GIB_KB / w for k, f, w in rows if k.startswith("k_write<double>") and w > 0), 1.0)
    return rows, f_read, f_write


def main():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, ROOT)
    from bench import kernel_sources_hash
    kernels = kernel_sources_hash()             # bench.py uses an entry only while the kernel sources still hash to this
    calib_rows, f_read, f_write = calibration()
    traffic = {}
    hbm_lines = ["HBM traffic counters (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, one pass each), tools/collect_profiles.sh + tools/profile_summary.py",
                 "", "Calibration: tools/hbm_calib.hip moves 1 GiB (1048576 KiB) per kernel in this code's access widths",
                 "%-64s %14s %14s" % ("kernel", "FETCH_SIZE KiB", "WRITE_SIZE KiB")]
    for k, f, w in calib_rows:
        hbm_lines.append("%-64s %14.0f %14.0f" % (k, f, w))
    hbm_lines += ["", "bytes read  = FETCH_SIZE x 1024 x %.3f   (8-byte loads; 4- and 16-byte loads read the same)" % f_read,
                  "bytes written = WRITE_SIZE x 1024 x %.3f (8-byte stores)" % f_write, ""]
    for cdir in sorted(glob.glob(os.path.join(src, "*", "kt"))):
        config = os.path.basename(os.path.dirname(cdir))
        kt = glob.glob(os.path.join(cdir, "**", "*kernel_trace.csv"), recursive=True)[0]
        rows = list(csv.DictReader(open(kt)))
        log = open(os.path.join(src, config + ".kt.log")).read().splitlines()
        frames = max(1, sum(1 for l in log if l.startswith("frame ")))
        per = collections.defaultdict(list)
        for r in rows:
            per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
        fetch, nf, _ = counters(os.path.join(src, config, "fetch"))
        write, _, _ = counters(os.path.join(src, config, "write"))
        sq, _, meta = counters(os.path.join(src, config, "sq"))
        lines = ["%s: rocprofv3 --kernel-trace over %d frames (tools/prof_frame.py, no work counting), separate --pmc passes over 1 frame" % (config, frames),
                 "host-timed frames under the profiler: " + "; ".join(l for l in log if l.startswith("frame ")),
                 "", "%-64s %7s %9s %9s %9s %11s %11s %6s %6s %6s %s" % ("kernel (per frame)", "launch", "ms total", "us avg", "ms max", "read MB", "written MB", "lanes", "valu%", "wait%", "vgpr/scratch/lds")]
        tot_ms = tot_r = tot_w = 0.0
        for k in sorted(per, key=lambda k: -sum(per[k])):
            d = per[k]
            rb = fetch[k].get("FETCH_SIZE", 0.0) * 1024 * f_read
            wb = write[k].get("WRITE_SIZE", 0.0) * 1024 * f_write
            c = sq.get(k, {})
            lanes = c.get("SQ_THREAD_CYCLES_VALU", 0) / c["SQ_INSTS_VALU"] if c.get("SQ_INSTS_VALU") else 0
            busy = 100 * c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else 0
            wait = 100 * c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else 0
            lines.append("%-64s %7.1f %9.3f %9.1f %9.3f %11.2f %11.2f %6.1f %6.0f %6.0f %s" % (
                k, len(d) / frames, sum(d) / frames, 1e3 * sum(d) / len(d), max(d), rb / 1e6, wb / 1e6, lanes, busy, wait, "/".join(meta.get(k, ("", "", "")))))
            tot_ms += sum(d) / frames; tot_r += rb; tot_w += wb
        span = []
        starts = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
        busy_ns, cur_s, cur_e = 0, None, None
        for s, e in starts:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy_ns += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        if cur_e is not None:
            busy_ns += cur_e - cur_s
        lines += ["", "sum of kernel times %.3f ms per frame; GPU busy (union of the intervals) %.3f ms per frame" % (tot_ms, busy_ns * 1e-6 / frames),
                  "HBM per frame: %.1f MB read + %.1f MB written = %.1f MB  (lanes = VALU thread-cycles / VALU instructions; valu%% = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES; wait%% = SQ_WAIT_ANY / SQ_WAVE_CYCLES)"
                  % (tot_r / 1e6, tot_w / 1e6, (tot_r + tot_w) / 1e6)]
        open(os.path.join(OUT, "%s_%s.txt" % (tag, config)), "w").write("\n".join(lines) + "\n")
        fp, wp = by_pass(os.path.join(src, config, "fetch"), "FETCH_SIZE"), by_pass(os.path.join(src, config, "write"), "WRITE_SIZE")
        per_pass = {k: fp[k] * 1024 * f_read + wp[k] * 1024 * f_write for k in fp}
        lines.append("HBM per pass: pass 1 %.1f MB, AA resolve %.1f MB" % (per_pass["pass1"] / 1e6, per_pass["aa"] / 1e6))
        open(os.path.join(OUT, "%s_%s.txt" % (tag, config)), "w").write("\n".join(lines) + "\n")
        traffic[config] = {"kernels": kernels, "bytes_per_frame": tot_r + tot_w, "bytes_pass1": per_pass["pass1"], "bytes_aa": per_pass["aa"], "read_bytes": tot_r, "written_bytes": tot_w,
                           "fetch_size_kib": tot_r / 1024 / f_read, "write_size_kib": tot_w / 1024 / f_write,
                           "read_factor": f_read, "write_factor": f_write,
                           "source": "profiles/%s_pmc_hbm.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), calibrated on 1 GiB streams (tools/hbm_calib.hip)" % tag}
        hbm_lines.append("%-26s read %9.1f MB  written %9.1f MB  total %9.1f MB per frame   (FETCH_SIZE %.0f KiB, WRITE_SIZE %.0f KiB)" % (
            config, tot_r / 1e6, tot_w / 1e6, (tot_r + tot_w) / 1e6, tot_r / 1024 / f_read, tot_w / 1024 / f_write))
        print(config, "%.1f ms/frame, %.1f MB/frame" % (tot_ms, (tot_r + tot_w) / 1e6))
    open(os.path.join(OUT, "%s_pmc_hbm.txt" % tag), "w").write("\n".join(hbm_lines) + "\n")
    path = os.path.join(OUT, "hbm_traffic.json")
    try:
        merged = json.load(open(path))          # configurations this run did not profile keep their (hash-stamped) entries
    except (OSError, ValueError):
        merged = {}
    merged.update(traffic)
    json.dump(merged, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
