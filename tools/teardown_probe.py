"""Finds what aborts the interpreter in ``torch.distributed.destroy_process_group`` after the 1-rank RCCL rehearsal of
the data-parallel step (gpurun_out/t40.log of round 1).  Every variant runs in a child process of its own with stderr
kept (pytest's fd capture swallowed the native message in round 1) and reports its exit code.

    python tools/teardown_probe.py            # all variants, logs under gpurun_out/teardown/
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = r"""
import faulthandler, gc, os, sys, torch
faulthandler.enable()
import torch.distributed as dist
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import ot_vae_lightning_amd as A
from detfill import mnist_like, normal
x = [mnist_like(64, 90 + i).cuda() for i in range(3)]
eps = [normal((64, 128, 1, 1), 95 + i).cuda() for i in range(3)]
KEEP = []

def run(overlap, graph, keep):
    torch.manual_seed(11)
    enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
    dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
    model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
    tr = A.HipTrainer(model, batch_shape=(64, 1, 32, 32), use_graph=graph, dp_overlap=overlap)
    for i in range(3):
        tr.step(x[i], eps[i])
    torch.cuda.synchronize()
    if keep:
        KEEP.append(tr)
    return tr

dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % PORT, rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
"""

VARIANTS = {
    # nothing but a communicator and one collective
    "a_plain": "t = torch.ones(1024, device='cuda'); dist.all_reduce(t); torch.cuda.synchronize(); dist.destroy_process_group()",
    # eager trainer (reducer side stream, no graphs), trainer still alive at destroy
    "b_eager_alive": "run(True, False, True); dist.destroy_process_group()",
    # captured trainer (3 graphs sharing a pool) still alive at destroy: the round-1 situation
    "c_graph_alive": "run(True, True, True); dist.destroy_process_group()",
    # captured trainer dropped + gc + sync before destroy
    "d_graph_dropped": "run(True, True, False); gc.collect(); torch.cuda.synchronize(); dist.destroy_process_group()",
    # captured trainer alive, but device idle and reducer stream joined
    "e_graph_alive_synced": "tr = run(True, True, True); tr.reducer.stream.synchronize(); torch.cuda.synchronize(); dist.destroy_process_group()",
    # two-graph path (collective between forward/backward graph and Adam graph), alive
    "f_twograph_alive": "run(False, True, True); dist.destroy_process_group()",
    # no explicit destroy at all: interpreter exit with live communicator + graphs
    "g_no_destroy": "run(True, True, True)",
    # (a variant that looped four global-mode trainers "so that a poll lands in a capture" was removed in round 3: it exited 0, i.e.
    # it did not demonstrate the cause, and repeating a run on the GPU box until it aborts is not a diagnostic -- DESIGN section 5)
    # teardown through HipTrainer.close(), then an explicit destroy, then a normal interpreter exit
    "i_close_then_destroy": "tr = run(True, True, True); tr.close(); dist.destroy_process_group()",
}


def main():
    out = os.path.join(ROOT, "gpurun_out", "teardown")
    os.makedirs(out, exist_ok=True)
    only = sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN", TORCH_SHOW_CPP_STACKTRACES="1",
               TORCH_CPP_LOG_LEVEL="INFO")
    summary = []
    for i, (name, tail) in enumerate(VARIANTS.items()):
        if only and name not in only:
            continue
        code = f"ROOT = {ROOT!r}\nPORT = {29810 + i}\n" + BODY + tail + "\nprint('PROBE-END', flush=True)\n"
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        with open(os.path.join(out, name + ".log"), "w") as f:
            f.write(f"rc={r.returncode}\n==== stdout ====\n{r.stdout}\n==== stderr ====\n{r.stderr}")
        line = f"{name}: rc={r.returncode} end={'PROBE-END' in r.stdout}"
        print(line, flush=True)
        summary.append(line)
    with open(os.path.join(out, "summary.txt"), "w") as f:
        f.write("\n".join(summary) + "\n")


if __name__ == "__main__":
    main()
