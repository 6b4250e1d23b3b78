"""tools only: bench.py against another build of the library (tools/variant_lib.sh): python3 tools/bench_lib.py LIB [bench args]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
lib = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[2:]
from libdogleg_amd import capi
capi.LIB_PATH = lib
import bench
bench.main()
