"""Code size / VGPR / AGPR / scratch / occupancy per kernel of a .hip file, from hipcc -S (dev tooling): python tools/kernel_stats.py gemm.hip"""
import re
import subprocess
import sys
from pathlib import Path

src = Path(__file__).resolve().parent.parent / "cinema_amd" / "csrc" / sys.argv[1]
out = src.parent / "build" / (src.stem + ".s")
out.parent.mkdir(exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", str(src), "-o", str(out)] + sys.argv[2:],
               check=True, stderr=subprocess.DEVNULL)
s = out.read_text()
names = re.findall(r"^(_Z\S+):\s+; @", s, flags=re.M)
stats = re.findall(r"; codeLenInByte = (\d+)\n(?:;.*\n)*?; NumVgprs: (\d+)\n; NumAgprs: (\d+)\n(?:;.*\n)*?; ScratchSize: (\d+)\n(?:;.*\n)*?; Occupancy: (\d+)", s)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for d, st in zip(dem, stats):
    d = d.replace("(anonymous namespace)::", "")
    print(f"{d[:72]:72s} code {st[0]:>6s} B  vgpr {st[1]:>3s} agpr {st[2]:>3s} scratch {st[3]:>4s} occupancy {st[4]}")
