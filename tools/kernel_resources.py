"""VGPRs / scratch / LDS of every kernel in libcinema_hip.so, read from the code-object metadata; flags kernels that spill (dev tooling, also used by
tests/test_host_cpu.py::test_hot_kernels_have_no_scratch).   python tools/kernel_resources.py [filter]"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

import yaml

LLVM = Path("/opt/rocm/lib/llvm/bin")
SO = Path(__file__).resolve().parent.parent / "cinema_amd" / "libcinema_hip.so"


def kernel_table(so: Path = SO) -> list:
    """[{name, vgpr_count, agpr_count, vgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size, ...}] for every kernel of every
    translation unit (the .hip_fatbin section holds one offload bundle per object file)."""
    rows = []
    with tempfile.TemporaryDirectory() as d:
        tmp = Path(d)
        sec = tmp / "fatbin"
        subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(so), str(sec)], check=True)
        blob = sec.read_bytes()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for n, st in enumerate(starts):
            part = tmp / f"bundle{n}"
            part.write_bytes(blob[st:starts[n + 1] if n + 1 < len(starts) else len(blob)])
            out = tmp / f"gfx950_{n}.co"
            subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}", f"--output={out}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(out)], check=True, capture_output=True, text=True).stdout
            for doc in re.findall(r"^\s*---\n(.*?)^\s*\.\.\.", notes, flags=re.S | re.M):
                for kern in yaml.safe_load(doc).get("amdhsa.kernels", []):
                    rows.append({k.lstrip("."): v for k, v in kern.items() if not isinstance(v, (list, dict))})
    names = subprocess.run(["c++filt"], input="\n".join(r["symbol"].replace(".kd", "") for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, name in zip(rows, names):
        r["name"] = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return rows


if __name__ == "__main__":
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    print(f"{'vgpr':>5s} {'agpr':>5s} {'spill':>5s} {'scratch':>7s} {'lds':>7s}  kernel")
    for r in kernel_table():
        if flt in r["name"]:
            flag = "  <-- SPILLS" if int(r.get("private_segment_fixed_size", 0)) > 0 else ""
            print(f"{r.get('vgpr_count', '?'):>5} {r.get('agpr_count', 0):>5} {r.get('vgpr_spill_count', 0):>5} {r.get('private_segment_fixed_size', '?'):>7} "
                  f"{r.get('group_segment_fixed_size', '?'):>7}  {r['name'][:100]}{flag}")
