"""Per-launch table of the step's packed convolution launches (conv_jobs_kernel): geometry of every job, algorithmic bytes
and FLOPs, measured duration (each call alone, 20 launches in one hipGraph, HIP events).  Reuses bench.py's job re-issue."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ot_vae_lightning_amd as A  # noqa: E402
from ot_vae_lightning_amd import _lib as L  # noqa: E402
from ot_vae_lightning_amd import functional as HF  # noqa: E402
from ot_vae_lightning_amd.utils.synthetic import mnist_like  # noqa: E402

KIND = {0: "fwd", 1: "dgrad", 2: "wgrad"}


def main(all_calls=False):
    lib = L.load()
    model = bench.build_model(A, seed=2).cuda().train()
    tr = A.HipTrainer(model, batch_shape=(1024, 1, 32, 32), use_graph=False)
    x = mnist_like(1024, seed=77).cuda()
    tr.step(x)
    HF.WGRAD_SIDE_STREAM = 2 if HF.WGRAD_SIDE_STREAM else 0  # the launches of the captured step (weight gradients on the side stream)
    HF.JOB_TRACE = []
    tr.step(x)
    torch.cuda.synchronize()
    trace, HF.JOB_TRACE = HF.JOB_TRACE, None
    print(f"{len(trace)} otvae_conv_multi calls per step")
    rows = []
    for ci, c in enumerate(trace):
        jobs = c["jobs"]
        desc = []
        by = fl = flx = flp = 0
        for i, j in enumerate(jobs):
            g = dict(zip(bench.GEOM_FIELDS, j["geom"]))
            b_, f_ = bench._job_algorithmic(j)      # bytes, LIVE flops (operands that exist; round 3 printed the padded count)
            by += b_
            fl += f_
            flx += bench._job_flops(j)["executed"]
            flp += bench._job_flops(j)["padded"]
            desc.append(f"{KIND[j['kind']]}{'*' if c['packed_mask'] >> i & 1 else ''} {g['Cs']}->{g['Cn']} k{g['KH']} s{g['stride']} up{g['up']} "
                        f"{g['Hs']}x{g['Ws']}->{g['Ho']}x{g['Wo']}")
        rows.append((ci, c["uniform_tap"], c["packed_mask"], by, (fl, flx, flp), " | ".join(desc)))
    # time every call alone (re-issued on synthetic tensors through bench's builder, one call per graph)
    print("FLOP columns: live (operands that exist) / executed by the kernel (dead taps dropped) / padded (2*y*KH*KW*Cs, rounds 1-3); TF/s and the\n"
          "fraction of the 157.3 TFLOP/s fp32 matrix peak are for the LIVE count; no launch may exceed 1.0 of a hardware peak")
    worst = 0.0
    for ci, ut, mask, by, (fl, flx, flp), desc in rows:
        c = trace[ci]
        one = dict(jobs=c["jobs"], packed_mask=(1 << len(c["jobs"])) - 1, uniform_tap=1)
        t = time_call(lib, one)
        each = ""
        if len(c["jobs"]) > 1 and mask:
            each = "  alone: " + " ".join(f"{time_call(lib, dict(jobs=[j])) * 1e6:.1f}" for j in c["jobs"])
        worst = max(worst, fl / t / 1e12 / bench.FP32_MFMA_PEAK_TFLOPS, by / t / 1e9 / bench.HBM_PEAK_GBS)
        print(f"call {ci:3d} ut={ut:2d} mask={mask:04b} {t*1e6:7.1f} us  {by/1e6:7.2f} MB {fl/1e9:6.3f}/{flx/1e9:6.3f}/{flp/1e9:6.3f} GF  "
              f"{by/t/1e9:7.0f} GB/s {fl/t/1e12:6.1f} TF/s ({fl/t/1e12/bench.FP32_MFMA_PEAK_TFLOPS:5.3f} of peak; padded convention "
              f"{flp/t/1e12/bench.FP32_MFMA_PEAK_TFLOPS:5.3f})  {desc}{each}", flush=True)
    print(f"largest fraction of a hardware peak over all calls (live FLOPs / algorithmic bytes): {worst:.3f}")


def time_call(lib, c, iters=20):
    dev = "cuda"
    jobs = c["jobs"]
    arr = (L.ConvJob * len(jobs))()
    held = []
    for jb, j in zip(arr, jobs):
        g = L.ConvGeom(*j["geom"])
        gd = dict(zip(bench.GEOM_FIELDS, j["geom"]))
        t_x = torch.randn(gd["N"], gd["Hs"], gd["Ws"], gd["Cs"], device=dev)
        t_y = torch.randn(gd["N"], gd["Ho"], gd["Wo"], gd["Cn"], device=dev)
        t_w = torch.randn(gd["KH"] * gd["KW"] * gd["Cs"] * gd["Cn"], device=dev) * 0.05
        vec = lambda: torch.rand(gd["Cs"], device=dev) + 0.5  # noqa: E731
        jb.kind, jb.relu, jb.has_bias, jb.geom = j["kind"], j["relu"], j["has_bias"], g
        held += [t_x, t_y, t_w]
        if j["has_norm"]:
            sc, sh = vec(), vec()
            jb.scale, jb.shift = L.ptr(sc), L.ptr(sh)
            held += [sc, sh]
        if j["kind"] == L.JOB_FWD:
            jb.x, jb.w, jb.y = L.ptr(t_x), L.ptr(t_w), L.ptr(t_y)
        elif j["kind"] == L.JOB_BWD_DATA:
            gv = torch.empty_like(t_x)
            jb.gy, jb.w, jb.x, jb.gv = L.ptr(t_y), L.ptr(t_w), L.ptr(t_x), L.ptr(gv)
            held.append(gv)
            if j["bn_sums"]:
                p_d, cp = C.c_int(0), C.c_int(0)
                L.check(lib.otvae_conv_bwd_data_ws(C.byref(g), C.byref(p_d), C.byref(cp)), "ws")
                mean, invstd = vec(), vec()
                part = torch.empty((p_d.value, 2, cp.value), device=dev, dtype=torch.float64)
                jb.mean, jb.invstd, jb.bn_partial = L.ptr(mean), L.ptr(invstd), L.ptr(part)
                held += [mean, invstd, part]
        else:
            p_w = C.c_int(0)
            L.check(lib.otvae_conv_bwd_weight_ws(C.byref(g), j["has_bias"], C.byref(p_w)), "ws")
            kk = gd["KH"] * gd["KW"] * gd["Cs"] + (1 if j["has_bias"] else 0)
            wpart = torch.empty((p_w.value, kk, gd["Cn"]), device=dev)
            gw, gb = torch.empty_like(t_w), torch.empty(gd["Cn"], device=dev)
            jb.x, jb.gy, jb.wpartial, jb.gw, jb.gb = L.ptr(t_x), L.ptr(t_y), L.ptr(wpart), L.ptr(gw), L.ptr(gb)
            jb.defer_reduce = L.DEFER_SPARSE
            held += [wpart, gw, gb]

    def issue():
        L.check(lib.otvae_conv_multi(len(jobs), arr, L.stream()), "otvae_conv_multi")

    issue()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        for _ in range(iters):
            issue()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * iters) * 1e-3


if __name__ == "__main__":
    main()
