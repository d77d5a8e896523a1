"""Where a wave of the persistent 256x256 GEMM spends a phase: READ segment / first barrier / MFMA segment / second barrier, in shader clocks (dev tooling).

Needs the dev build of the library with -DCINEMA_P256_PROBE as cinema_amd/csrc/build/libcinema_hip_probe.so:
  cd cinema_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DCINEMA_P256_PROBE -c gemm256.hip -o build/gemm256_probe.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/libcinema_hip_probe.so build/gemm256_probe.o $(ls build/*.o | grep -v gemm256)
The stamps themselves cost ~4 x (s_memtime + wait) per phase: read the numbers as proportions.
   python tools/p256_phase_probe.py [forms, default 1,2]"""
import ctypes as C
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

K._LIB_PATH = Path(K.__file__).resolve().parent / "csrc" / "build" / "libcinema_hip_probe.so"
lib = K.load()
lib.cinema_debug_p256_probe.argtypes = [C.c_void_p]
dev = "cuda"
forms = [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "2"])]
blocks = {
    "enc block wgrad": [(10960, 2304, 768), (10960, 768, 768), (10960, 3072, 768), (10960, 768, 3072)],
    "8192^3 wgrad": [(8192, 8192, 8192)],
}
for name, gs in blocks.items():
    probs = []
    for rows, n, k in gs:
        dy = (torch.randn(rows, n, device=dev) * 0.5).to(torch.bfloat16)
        x = (torch.randn(rows, k, device=dev) * 0.5).to(torch.bfloat16)
        probs.append((dy, x, torch.zeros(n, k, dtype=torch.float32, device=dev), None))
    for f in forms:
        os.environ["CINEMA_P256_LOOP"] = str(f)
        for _ in range(3):
            K.gemm_wgrad_grouped(probs, p256=True)
        torch.cuda.synchronize()
        buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
        lib.cinema_debug_p256_probe(buf.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.gemm_wgrad_grouped(probs, p256=True)
        e1.record()
        torch.cuda.synchronize()
        lib.cinema_debug_p256_probe(None)
        t = buf.view(256, 8, 8).cpu().double()
        for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
            g = t[:, sl, :].reshape(-1, 8)
            g = g[g[:, 4] > 0]
            per = g[:, :4].sum(0) / g[:, 4].sum()
            pre = g[:, 5].sum() / g[:, 4].sum()
            print(f"{name} form {f} {grp}: per phase READ {per[0]:.0f} (reads + DMA issue {pre:.0f}, then the vmcnt wait) | barrier {per[1]:.0f} | MFMA {per[2]:.0f} | barrier {per[3]:.0f} = {per.sum():.0f} clk"
                  f"  ({int(g[:, 4].sum() / g.shape[0])} phases per wave; launch {e0.elapsed_time(e1) * 1e3:.0f} us)", flush=True)
