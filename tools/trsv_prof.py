import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo') else os.getcwd())
sys.argv = ["bench.py", "--no-cpu-baseline", "--workload", "dense-50k", "--steps", "5", "--warmup", "2"]
from libdogleg_amd import capi
capi.LIB_PATH = os.path.abspath("tools/micro/libvar.so")
import bench, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
capi.lib().dlg_trsv_profile_dump(32)
