import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from cinema_amd.segmentation.convunetr import ConvUNetR
from cinema_amd.segmentation.train import SegTrainStep
kw = bench.seg_kwargs("base", (256, 256, 12))
torch.manual_seed(0)
model = ConvUNetR(**kw).to("cuda").train()
step = SegTrainStep(model, ["sax"], lr=1e-4, layer_decay=0.75)
g = torch.Generator().manual_seed(1)
img = torch.rand(4, 1, 256, 256, 12, generator=g)
batch = {"sax_image": img.cuda(), "sax_label": torch.clamp((img * 4).long(), 0, 3).to(torch.int8).cuda()}
for _ in range(8): step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/20:.2f} ms/step, wall {1e3*(t2-t0)/20:.2f} ms/step")
