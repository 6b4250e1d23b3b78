"""tools only: time line of the diagonal owners of the one-launch dense factorisation (k_potrf_tiles) -- a library built with
-DDLG_POTRF_PROFILE (tools/variant_lib.sh build dense_diag.hip -DDLG_POTRF_PROFILE; cp tools/micro/libvar.so tools/micro/libpotrfprof.so).
Ticks of wall_clock64 (10 ns)."""
import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo') else os.getcwd())
sys.argv = ["bench.py", "--no-cpu-baseline", "--workload", "dense-50k", "--steps", "5", "--warmup", "2"]
from libdogleg_amd import capi
capi.LIB_PATH = os.path.abspath(os.environ.get("DLG_PROF_LIB", "tools/micro/libpotrfprof.so"))
import bench, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
capi.lib().dlg_potrf_profile_dump(32)
