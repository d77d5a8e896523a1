"""Per-shape GEMM table of one ConvUNetR segmentation step (which launches fall to the generic kernel, and why): dev tooling."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from cinema_amd import tape as T  # noqa: E402
from cinema_amd.optim import FlatModel, FusedAdamW  # noqa: E402
from cinema_amd.segmentation.convunetr import ConvUNetR  # noqa: E402
from cinema_amd.segmentation.train import _segmentation_loss  # noqa: E402
from cinema_amd.vit import get_vit_config  # noqa: E402

b = 2
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


T.SIDE_WGRAD = False
for _ in range(2):
    step()
K.GEMM_PROFILE = []
step()
torch.cuda.synchronize()
prof, K.GEMM_PROFILE = K.GEMM_PROFILE, None
agg = {}
for kind, flops, e0, e1, shape, *_ in prof:
    a = agg.setdefault((kind, shape), [0.0, 0.0, 0])
    a[0] += flops
    a[1] += e0.elapsed_time(e1) * 1e-3
    a[2] += 1
print(f"all GEMM launches: {sum(v[1] for v in agg.values()) * 1e3:.1f} ms/step")
for (kind, (m, n, k, ak, bk, sk, _alg)), (fl, secs, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{K.GEMM_KERNEL_NAMES[kind]:36s} m{m:8d} n{n:5d} k{k:8d} aK{ak} bK{bk} sk{sk:3d} x{cnt:3d} {secs / cnt * 1e6:9.1f} us {fl / secs / 1e12:6.1f} TF {secs * 1e3:7.2f} ms")
