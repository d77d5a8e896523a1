#!/bin/bash
# main-loop form of the persistent 256x256 GEMM (CINEMA_P256_LOOP: 1 = LDS-DMA issued between the MFMAs, 2 = issued by the reading wave): fp8 kernel A/B, then the step A/B
# (interleaved processes) on config 2 and on config 5 (fp8)
mkdir -p gpurun_out
OUT=gpurun_out/r04_ag_loop_step_ab.txt
: > $OUT
python - >> $OUT 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
from cinema_amd import hip as K
from tools.bench_p256_loop import bench
dev = "cuda"
for name, gs in {"large enc": [(13824, 3072, 1024), (13824, 1024, 1024), (13824, 4096, 1024), (13824, 1024, 4096)], "enc block": [(10960, 2304, 768), (10960, 768, 768), (10960, 3072, 768), (10960, 768, 3072)], "8192^3": [(8192, 8192, 8192)]}.items():
    q8 = []
    for rows, n, k in gs:
        dy = (torch.randn(rows, n, device=dev) * 0.5).to(torch.bfloat16)
        x = (torch.randn(rows, k, device=dev) * 0.5).to(torch.bfloat16)
        q8.append((*K.quantize_fp8(dy), *K.quantize_fp8(x), torch.zeros(n, k, dtype=torch.float32, device=dev)))
    flops = sum(2.0 * r * n * k for r, n, k in gs)
    outs = {}
    def mk(f):
        def run():
            os.environ["CINEMA_P256_LOOP"] = str(f)
            K.gemm_fp8_wgrad_grouped(q8)
        return run
    for f in (1, 2):
        for t in q8: t[-1].zero_()
        mk(f)(); torch.cuda.synchronize()
        outs[f] = [t[-1].clone() for t in q8]
    same = all(torch.equal(a, b) for a, b in zip(outs[1], outs[2]))
    r = bench({f: mk(f) for f in (1, 2)})
    print(f"fp8 wgrad {name:10s} | form 1 {flops / r[1] / 1e12:7.1f} TF ({r[1] * 1e6:6.1f} us)  form 2 {flops / r[2] / 1e12:7.1f} TF ({r[2] * 1e6:6.1f} us) | identical {same}", flush=True)
PY
for r in 1 2; do for w in 1 2; do
  CINEMA_P256_LOOP=$w timeout -s KILL 200 python bench.py --steps 30 --warmup 10 --cpu-budget 0 --profile-steps 0 --no-secondary 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 2 P256_LOOP=$w ms_per_step', d['ms_per_step'], 'loss', d['config']['final_loss'])"
done; done >> $OUT 2>&1
for r in 1 2; do for w in 1 2; do
  CINEMA_P256_LOOP=$w timeout -s KILL 300 python bench.py --steps 15 --warmup 5 --prewarm 5 --cpu-budget 0 --profile-steps 0 --no-secondary --size large --sax 256,256,24 --lax 256,256 --batch 8 --dtype fp8 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 5 fp8 P256_LOOP=$w ms_per_step', d['ms_per_step'], 'loss', d['config']['final_loss'])"
done; done >> $OUT 2>&1
cat $OUT
