import sys, time, json
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import torch
import bench
from parity import mae_fp8_grad_parity
from cinema_amd import CineMA
kw5 = bench.base_kwargs("large", (256, 256, 24), (256, 256))
torch.manual_seed(0)
sd5 = {k: v.detach().clone() for k, v in CineMA(**kw5).state_dict().items()}
t0 = time.time()
names = ("encoder.blocks.0.attn.kv.weight", "encoder.blocks.23.mlp.fc1.weight", "decoder.blocks.0.attn.q.weight", "decoder.blocks.7.mlp.fc2.weight", "enc_down_dict.sax.conv_blocks.0.conv.0.mlp.fc1.weight", "pred_head_dict.sax.weight")
par = mae_fp8_grad_parity(kw5, sd5, batch=1, seed=17, device="cuda", threads=16, modes=("bf16", "fp8_wgrad"), report=names)
print("seconds", time.time() - t0, "oracle", par["oracle_seconds"])
for m in ("bf16", "fp8_wgrad"):
    r = par[m]
    print(m, "loss_rel", r["loss_rel"], "gn", r["grad_norm_rel"], "whole", r["whole_grad_rel_l2"], "worst", r["worst_matrix_rel_l2"], r["named_rel_l2"])
