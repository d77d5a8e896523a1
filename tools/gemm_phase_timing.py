"""Where a workgroup of the one-shot 128x128 GEMM spends its time: prologue (first k-tile lands) / MFMA loop / epilogue issue / store drain.

Needs the dev build of the library with -DCINEMA_GEMM_TIMING as cinema_amd/csrc/build/libcinema_hip_timing.so:
  cd cinema_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DCINEMA_GEMM_TIMING -c gemm.hip -o build/gemm_timing.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/libcinema_hip_timing.so build/gemm_timing.o $(ls build/*.o | grep -v gemm)
wall_clock64() ticks at 100 MHz (10 ns)."""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

K._LIB_PATH = Path(K.__file__).resolve().parent / "csrc" / "build" / "libcinema_hip_timing.so"
lib = K.load()
lib.cinema_debug_gemm_timing.argtypes = [C.c_void_p]
dev = "cuda"
for m, n, k, epi in ((32848, 2048, 512, "gelu"), (32848, 2048, 512, "plain"), (10960, 3072, 768, "plain"), (10960, 768, 3072, "res"), (10752, 768, 768, "plain")):
    x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    y16 = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    y32 = torch.empty(m, n, dtype=torch.float32, device=dev)
    pre = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    res = torch.randn(m, n, device=dev)
    if epi == "gelu":
        fn = lambda: K.gemm(x, w, out=y16, bias=bias, act=1, aux_out=pre)  # noqa: E731
    elif epi == "res":
        fn = lambda: K.gemm(x, w, out=y32, bias=bias, residual=res)  # noqa: E731
    else:
        fn = lambda: K.gemm(x, w, out=y16, bias=bias)  # noqa: E731
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    buf = torch.zeros(tiles * 8, dtype=torch.int64, device=dev)
    for _ in range(3):
        fn()
    lib.cinema_debug_gemm_timing(buf.data_ptr())
    fn()
    torch.cuda.synchronize()
    lib.cinema_debug_gemm_timing(None)
    t = buf.view(tiles, 8).cpu().double()
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    t = t[t[:, 4] > 0]
    span = (t[:, 4].max() - t0) / 100.0
    pro, loop, epi_t, drain = [(t[:, i + 1] - t[:, i]).mean().item() / 100.0 for i in range(4)]
    life = (t[:, 4] - t[:, 0]).mean().item() / 100.0
    starts = ((t[:, 0] - t0) / 100.0).sort().values
    print(f"{m}x{n}x{k} {epi}: {t.shape[0]} workgroups, kernel span {span:.1f} us; per workgroup: prologue {pro:.2f} + loop {loop:.2f} ({loop / ((k + 63) // 64):.2f}/k-tile) "
          f"+ epilogue {epi_t:.2f} + drain {drain:.2f} = {life:.2f} us; start times p50 {starts[len(starts) // 2]:.1f} p100 {starts[-1]:.1f} us", flush=True)
