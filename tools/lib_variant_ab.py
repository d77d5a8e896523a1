"""bench.py under a variant build of the library (dev tooling): python tools/lib_variant_ab.py <path/to/libcinema_hip_variant.so | default> [bench.py args...]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

lib = sys.argv[1]
if lib != "default":
    K._LIB_PATH = Path(lib).resolve()
sys.argv = ["bench.py"] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
