"""How far ahead of the GPU is the host?  Issues 20 steps without synchronising and reports issue time vs completion time."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'unsupervised-pseuso-lidar_amd'))
import torch
import bench
dev = torch.device('cuda', 0)
depth, pose, opt, crit = bench.build(dev)
s = bench.synthetic_samples(12, 192, 640, 0)
samples = {"tgt": s["tgt"].to(dev), "ref_imgs": [r.to(dev) for r in s["ref_imgs"]], "intrinsics": s["intrinsics"].to(dev)}
step = bench.make_step(depth, pose, opt, crit, samples, pair=True, graph=False)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue %.2f ms/step, total %.2f ms/step, GPU tail after last issue %.2f ms" % ((t1-t0)/20*1e3, (t2-t0)/20*1e3, (t2-t1)*1e3))
