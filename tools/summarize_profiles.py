"""Turns the rocprofv3 outputs merged into gpurun_out/ into the small, tracked summaries under profiles/.

    python tools/summarize_profiles.py <kernel-stats dir> <tag>         # e.g. gpurun_out/prof8 r01_final
    python tools/summarize_profiles.py --replay <kernel-trace dir> <tag> # replayed (timed) steps only
    python tools/summarize_profiles.py --pmc gpurun_out <tag>           # pmc_fetch / pmc_write / pmc_mfma passes
    python tools/summarize_profiles.py --roofline <tag>                 # joins the replay and PMC summaries
    python tools/summarize_profiles.py --pmc-sq <pmc_sq dir> <tag>      # SQ issue / wait counters per kernel

HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced reads, so reads = 2 * FETCH_SIZE KiB; the two counters need separate passes (TCC slots).
"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# where the summaries go: profiles/ of the repo, or (on the GPU box, whose repo copy does not travel back) a directory under
# gpurun_out/ named by OTVAE_PROFILES_OUT
OUT = os.environ.get("OTVAE_PROFILES_OUT") or os.path.join(ROOT, "profiles")


def kernel_stats(src, tag):
    """Whole-process view of `rocprofv3 --stats` (eager warm-up, capture, every replay, the parity check and the
    dominant-kernel timing loop of bench.py together): calls, average and total duration per kernel."""
    f = max(glob.glob(os.path.join(src, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    out = os.path.join(OUT, f"{tag}_kernel_stats.csv")
    with open(out, "w") as w:
        w.write("kernel,calls,avg_us,total_ms,percent\n")
        tot = sum(int(r["TotalDurationNs"]) for r in rows)
        for r in rows:
            w.write('"%s",%d,%.2f,%.4f,%.2f\n' % (r["Name"].split("(")[0], int(r["Calls"]), float(r["AverageNs"]) / 1e3,
                                                 int(r["TotalDurationNs"]) / 1e6, 100.0 * int(r["TotalDurationNs"]) / tot))
    print("wrote", out, "total kernel ms %.3f" % (tot / 1e6))


def replay_stats(src, tag, steps=20):
    """Per-kernel statistics over `steps` consecutive hipGraph replays only (the timed region of bench.py), cut out of the
    kernel trace at the adam_kernel launches: the stats file of the same run also contains the eager warm-up, the
    capture warm-up, the parity check and the dominant-kernel timing loop."""
    f = max(glob.glob(os.path.join(src, "*", "*_kernel_trace.csv")), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
    # the timed region = the `steps` consecutive steps with the smallest wall time (graph replays run back to back; the eager
    # warm-up, the eager-route timing and the parity step of the same process are slower per step)
    t_end = [int(rows[i]["End_Timestamp"]) for i in ends]
    best = min(range(len(ends) - steps), key=lambda a: t_end[a + steps] - t_end[a])
    lo, hi = ends[best] + 1, ends[best + steps] + 1
    seg = rows[lo:hi]
    acc = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        k = r["Kernel_Name"].split("(")[0]
        acc[k][0] += 1
        acc[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(v[1] for v in acc.values())
    wall = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    out = os.path.join(OUT, f"{tag}_replay_kernel_stats.csv")
    with open(out, "w") as w:
        w.write("# csrc_sha256=%s\n" % csrc_digest())
        w.write("# %d consecutive graph replays: %d launches/step, kernel-busy %.3f ms/step, wall %.3f ms/step\n" %
                (steps, len(seg) // steps, tot / 1e6 / steps, wall / 1e6 / steps))
        w.write("kernel,calls_per_step,avg_us,ms_per_step,percent\n")
        for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            w.write('"%s",%.1f,%.2f,%.4f,%.2f\n' % (k, n / steps, t / n / 1e3, t / 1e6 / steps, 100.0 * t / tot))
    print("wrote", out, "launches/step", len(seg) // steps, "busy ms/step %.3f wall %.3f" % (tot / 1e6 / steps, wall / 1e6 / steps))


def timeline(src, tag):
    """ONE graph replay (the 10th of the fastest 20-step window) as a timeline: start / end of every launch in us from the step's first
    kernel, the queue it ran on, the idle gap on that queue in front of it: where the chain waits, what ends the step."""
    f = max(glob.glob(os.path.join(src, "*", "*_kernel_trace.csv")), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
    t_end = [int(rows[i]["End_Timestamp"]) for i in ends]
    steps = 20
    best = min(range(len(ends) - steps), key=lambda a: t_end[a + steps] - t_end[a])
    seg = rows[ends[best + 9] + 1: ends[best + 10] + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    qkey = "Queue_Id" if "Queue_Id" in seg[0] else ("Stream_Id" if "Stream_Id" in seg[0] else None)
    last_end = {}
    out = os.path.join(OUT, f"{tag}_step_timeline.txt")
    with open(out, "w") as w:
        w.write("# one graph replay; us from the first kernel's start.  gap = idle time on the same queue in front of the launch\n")
        w.write("# %5s %9s %9s %8s %7s  kernel\n" % ("queue", "start", "end", "dur", "gap"))
        for r in seg:
            q = r[qkey] if qkey else "?"
            a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
            gap = a - last_end.get(q, a)
            last_end[q] = b
            w.write("  %5s %9.2f %9.2f %8.2f %7.2f  %s\n" % (q, a, b, b - a, gap, r["Kernel_Name"].split("(")[0][:70]))
    print("wrote", out, len(seg), "launches")


def csrc_digest():
    """sha256 over the kernel sources (csrc/*.hip, *.h, *.cpp, sorted by name): the same function as bench.csrc_digest"""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ot_vae_lightning_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()


def pmc(src, tag):
    def load(d, counter):
        f = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))[0]
        a = collections.defaultdict(lambda: [0, 0.0, 0.0])
        for x in csv.DictReader(open(f)):
            if x["Counter_Name"] == counter:
                k = x["Kernel_Name"].split("(")[0]
                a[k][0] += 1
                a[k][1] += float(x["Counter_Value"])
                a[k][2] += (int(x["End_Timestamp"]) - int(x["Start_Timestamp"]))
        return a
    F, W = load("pmc_fetch", "FETCH_SIZE"), load("pmc_write", "WRITE_SIZE")
    MF, GA = load("pmc_mfma", "SQ_VALU_MFMA_BUSY_CYCLES"), load("pmc_mfma", "GRBM_GUI_ACTIVE")
    out = os.path.join(OUT, f"{tag}_pmc_per_kernel.csv")
    with open(out, "w") as w:
        # the kernel sources these counters were collected from: bench.py reports `roofline.traffic` from this file only while
        # the digest still matches the tree (a changed kernel makes the committed bytes stale)
        w.write("# csrc_sha256=%s\n" % csrc_digest())
        w.write("kernel,launches,FETCH_SIZE_KiB_per_launch,read_MB_per_launch(2x_gfx950_correction),WRITE_SIZE_KiB_per_launch,"
                "hbm_MB_per_launch,SQ_VALU_MFMA_BUSY_CYCLES_per_launch,GRBM_GUI_ACTIVE_per_launch,"
                "mfma_busy_frac(=busy/(gui_active/8*1024 SIMDs))\n")
        for k in sorted(F, key=lambda k: -F[k][1]):
            n, fs, _ = F[k]
            ws = W.get(k, [1, 0.0, 0])
            mf, ga = MF.get(k, [1, 0.0, 0]), GA.get(k, [1, 0.0, 0])
            fetch = fs / n
            write = ws[1] / max(ws[0], 1)
            mfb, gui = mf[1] / max(mf[0], 1), ga[1] / max(ga[0], 1)
            frac = mfb / (gui / 8 * 1024) if gui else 0.0
            w.write('"%s",%d,%.1f,%.3f,%.1f,%.3f,%.0f,%.0f,%.4f\n' % (k, n, fetch, 2 * fetch / 1024, write,
                                                                     (2 * fetch + write) / 1024, mfb, gui, frac))
    print("wrote", out)


def pmc_sq(src, tag):
    """Instruction-issue evidence per kernel from one SQ pass (--pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS
    SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES).  Units (MI355X_MICROARCH.md, PMC table):
    SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_INSTS_* count wave-instructions."""
    f = glob.glob(os.path.join(src, "*", "*_counter_collection.csv"))[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    meta = {}
    for x in csv.DictReader(open(f)):
        k = x["Kernel_Name"].split("(")[0]
        acc[k][x["Counter_Name"]] += float(x["Counter_Value"])
        if x["Counter_Name"] == "SQ_WAVES":
            cnt[k] += 1
            acc[k]["dur_ns"] += int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
            meta[k] = (x["VGPR_Count"], x["Accum_VGPR_Count"], x["LDS_Block_Size"], x["Workgroup_Size"])
    out = os.path.join(OUT, f"{tag}_pmc_sq_per_kernel.csv")
    with open(out, "w") as w:
        w.write("# per launch averages; valu_active_frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of a wave's life spent issuing\n"
                "# vector instructions), wait_any_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked on s_waitcnt / barrier),\n"
                "# lds_wait_frac = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES, valu_per_wave = SQ_INSTS_VALU / SQ_WAVES,\n"
                "# occupancy_waves_per_simd = SQ_WAVE_CYCLES / SQ_BUSY_CYCLES-normalised estimate (wave-quad-cycles per busy SE cycle / SIMDs)\n")
        w.write("kernel,launches,avg_us_under_pmc,vgpr,agpr,lds_bytes,wg_size,waves,valu_insts_per_wave,lds_insts_per_wave,"
                "valu_active_frac,wait_any_frac,lds_wait_frac\n")
        for k in sorted(acc, key=lambda k: -acc[k]["dur_ns"]):
            a, n = acc[k], max(cnt[k], 1)
            wc = a["SQ_WAVE_CYCLES"] or 1.0
            wv = a["SQ_WAVES"] or 1.0
            w.write('"%s",%d,%.2f,%s,%s,%s,%s,%.0f,%.1f,%.1f,%.3f,%.3f,%.3f\n' % (
                k, n, a["dur_ns"] / n / 1e3, *meta.get(k, ("", "", "", "")), wv / n, a["SQ_INSTS_VALU"] / wv, a["SQ_INSTS_LDS"] / wv,
                a["SQ_ACTIVE_INST_VALU"] / wc, a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_LDS"] / wc))
    print("wrote", out)


def roofline(tag):
    """Joins <tag>_final_replay_kernel_stats.csv and <tag>_pmc_per_kernel.csv into <tag>_kernel_roofline.csv."""
    rep = {r["kernel"]: r for r in csv.DictReader(l for l in open(os.path.join(OUT, f"{tag}_final_replay_kernel_stats.csv"))
                                                  if not l.startswith("#"))}
    pm = list(csv.DictReader(l for l in open(os.path.join(OUT, f"{tag}_pmc_per_kernel.csv")) if not l.startswith("#")))
    busy_key = [k for k in pm[0] if k.startswith("mfma_busy_frac")][0]
    pm = {r["kernel"]: r for r in pm}
    rows = []
    for k, r in rep.items():
        if k in pm:
            us, mib = float(r["avg_us"]), float(pm[k]["hbm_MB_per_launch"])
            rows.append((float(r["ms_per_step"]), k, float(r["calls_per_step"]), us, mib, mib * 1.048576 / us if us else 0.0,
                         float(pm[k][busy_key])))
    rows.sort(reverse=True)
    out = os.path.join(OUT, f"{tag}_kernel_roofline.csv")
    with open(out, "w") as w:
        w.write("# per kernel of the timed step: average duration (kernel trace, timed replays), HBM MiB per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE),\n"
                "# (the x2 read correction is calibrated for 16-byte-per-lane loads; kernels reading 4 bytes per lane, e.g. the reductions, are\n"
                "#  over-counted by up to 2x),\n"
                "# achieved HBM TB/s = bytes / duration (peak 8 TB/s), MFMA busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs))\n")
        w.write("kernel,calls_per_step,avg_us,ms_per_step,hbm_MiB_per_launch,achieved_TB_per_s,frac_of_hbm_peak,mfma_busy_frac\n")
        for ms, k, c, us, mib, tbs, mf in rows:
            w.write('"%s",%.0f,%.2f,%.4f,%.2f,%.3f,%.3f,%.4f\n' % (k, c, us, ms, mib, tbs, tbs / 8.0, mf))
    print("wrote", out)


if __name__ == "__main__":
    if sys.argv[1] == "--roofline":
        roofline(sys.argv[2])
        sys.exit(0)
    if sys.argv[1] == "--pmc-sq":
        pmc_sq(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "--pmc":
        pmc(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "--replay":
        replay_stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "--timeline":
        timeline(sys.argv[2], sys.argv[3])
    else:
        kernel_stats(sys.argv[1], sys.argv[2])
