"""Host cost of the data-parallel step's call sequence: three captured graphs + the side-stream all-reduces of the
flat gradient buffer, rehearsed on ONE GPU with a 1-rank RCCL process group (the collectives really run; they move no data between
GPUs, so the DEVICE time below is a lower bound of the multi-GPU step, the HOST enqueue time is what it is on any world size).

    python tools/dp_host_enqueue.py        # prints ms/step and host enqueue ms/step for the overlapped and the plain DP sequence
"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import ot_vae_lightning_amd as A  # noqa: E402
from ot_vae_lightning_amd.utils.synthetic import mnist_like  # noqa: E402


def run(tag, steps=60, warmup=15, batch=1024, **kw):
    torch.manual_seed(0)
    enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
    dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
    model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
    tr = A.HipTrainer(model, batch_shape=(batch, 1, 32, 32), **kw)
    pool = [mnist_like(batch, seed=5 + i).cuda() for i in range(4)]
    for i in range(warmup):
        tr.step(pool[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(pool[i % 4])
    host = (time.perf_counter() - t0) / steps * 1e3
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / steps * 1e3
    print(f"{tag:72s} {total:7.3f} ms/step   host enqueue {host:6.3f} ms/step", flush=True)
    tr.close()


if __name__ == "__main__":
    torch.cuda.set_device(0)
    run("single process, no process group: one captured step (segments)", data_parallel=False)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29877", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    run("1-rank RCCL: 3 graphs + 3 all-reduce calls (decoder half overlapped)", dp_overlap=True)
    run("1-rank RCCL: 2 graphs + 1 all-reduce", dp_overlap=False)
    run("1-rank RCCL: 3 graphs + 3 all-reduce calls + global-norm clipping", dp_overlap=True, gradient_clip_val=1.0)
    torch.cuda.synchronize()
    dist.destroy_process_group()
