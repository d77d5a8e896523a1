"""First timing of the ConvUNetR segmentation step at BASELINE config 4 shape (ACDC-like SAX 256x256x12, 4 classes, ViT-Base): dev tooling."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd.optim import FlatModel, FusedAdamW  # noqa: E402
from cinema_amd.segmentation.convunetr import ConvUNetR  # noqa: E402
from cinema_amd.segmentation.train import _segmentation_loss  # noqa: E402
from cinema_amd.vit import get_vit_config  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 2
vit = get_vit_config("base")
torch.manual_seed(0)
model = ConvUNetR(image_size_dict={"sax": (256, 256, 12)}, in_chans_dict={"sax": 1}, out_chans=4, enc_patch_size_dict={"sax": (4, 4, 1)},
                  enc_scale_factor_dict={"sax": (2, 2, 1)}, enc_conv_chans=[64, 128], enc_conv_n_blocks=2, enc_embed_dim=vit["enc_embed_dim"],
                  enc_depth=vit["enc_depth"], enc_n_heads=vit["enc_n_heads"], dec_chans=(32, 64, 128, 256, 512),
                  dec_patch_size_dict={"sax": (2, 2, 1)}, dec_scale_factor_dict={"sax": (2, 2, 1)}).to("cuda").eval()
flat = FlatModel(model, 0.05)
opt = FusedAdamW(flat, lr=1e-4)
g = torch.Generator().manual_seed(1)
img = torch.rand(b, 1, 256, 256, 12, generator=g).cuda()
lab = torch.randint(0, 4, (b, 1, 256, 256, 12), generator=g).cuda()


def step():
    logits = model({"sax": img})["sax"]
    loss, _ = _segmentation_loss(logits, lab)
    loss.backward()
    opt.step(1.0)
    opt.zero_grad()
    return loss


for _ in range(3):
    loss = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"ConvUNetR base, SAX 256x256x12, batch {b}: {dt * 1e3:.1f} ms/step ({b / dt:.2f} samples/s), loss {float(loss):.4f}, "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
