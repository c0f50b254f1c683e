"""Host-side profile of the eagerly issued training step (what a Lightning loop pays per step)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
import ot_vae_lightning_amd as A
from ot_vae_lightning_amd.utils.synthetic import mnist_like

model = B.build_model(A, seed=2, workload="gaussian").cuda().train()
tr = A.HipTrainer(model, batch_shape=(1024, 1, 32, 32), use_graph=False, data_parallel=False)
x = mnist_like(1024, seed=77).cuda()
for _ in range(5):
    tr.step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.step(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("eager: host %.3f ms/step, done %.3f ms/step" % ((t1 - t0) / 20 * 1e3, (time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    tr.step(x)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
