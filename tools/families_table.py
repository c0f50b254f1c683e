"""profiles/<round>_families.txt from the committed bench line (profiles/<round>_bench.json, written by tools/final_profile.sh and then
re-run with the fresh summaries in place so that `families` and `roofline.traffic` carry the measured columns):

    python tools/families_table.py [r04] > profiles/r04_families.txt
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
d = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench.json")))
fam = d["families"]
replay = open(os.path.join(ROOT, "profiles", f"{tag}_final_replay_kernel_stats.csv")).read().splitlines()[1].lstrip("# ")
print(f"# The step by kernel family ({tag}): launches, kernel ms per step and PMC bytes from {tag}_final_replay_kernel_stats.csv / {tag}_pmc_per_kernel.csv")
print("# (the same bench command under rocprofv3; " + replay + "); algorithmic bytes / FLOPs from the bench line's `families`")
print("# (convolutions: the step's own recorded jobs, live multiply-adds; attention: analytic; BatchNorm launches hold no algorithmic work).")
print("# The weight-gradient family runs on the side stream beside the chain, so the ms column adds up to more than the step.")
print("# frac = algorithmic work / family ms against the 8 TB/s HBM and 157.3 TFLOP/s fp32 peaks.")
hdr = f"{'family':22s} {'launches':>8s} {'ms/step':>8s} {'alg MB':>9s} {'PMC MB':>9s} {'PMC/alg':>8s} {'alg GFLOP live':>15s} {'padded':>8s} {'frac HBM':>9s} {'frac fp32':>10s}"
print(hdr)
tot = [0.0, 0.0, 0.0, 0.0, 0.0]


def f(v, spec):
    return format(v, spec) if v is not None else "n/a".rjust(int(spec.split(".")[0]))


for r in fam["rows"]:
    ratio = (r["pmc_mbytes"] / r["alg_mbytes"]) if (r["pmc_mbytes"] is not None and r["alg_mbytes"]) else None
    print(f"{r['family']:22s} {f(r['launches'], '8.0f')} {f(r['ms_per_step'], '8.4f')} {r['alg_mbytes']:9.1f} {f(r['pmc_mbytes'], '9.1f')} "
          f"{f(ratio, '8.2f')} {r['alg_gflops_live']:15.3f} {r['alg_gflops_padded']:8.3f} {f(r['frac_of_hbm_peak'], '9.4f')} {f(r['frac_of_fp32_peak'], '10.4f')}")
    for i, k in enumerate(("launches", "ms_per_step", "alg_mbytes", "pmc_mbytes", "alg_gflops_live")):
        tot[i] += r[k] or 0.0
print(f"{'sum':22s} {tot[0]:8.0f} {tot[1]:8.4f} {tot[2]:9.1f} {tot[3]:9.1f} {(tot[3] / tot[2] if tot[2] else 0):8.2f} {tot[4]:15.3f}")
sr = d["step_roofline"]
print(f"whole step (un-profiled, {d['ms_per_step']} ms): {sr['frac_of_hbm_peak']:.4f} of the HBM peak, {sr['frac_of_fp32_peak']:.4f} of the fp32 peak (live multiply-adds)")
