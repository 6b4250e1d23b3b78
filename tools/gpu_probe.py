"""Micro-probes run on the MI355X: fp64 MFMA issue rate and HBM copy ceiling."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libdogleg_amd import capi

L = capi.lib()
v = C.c_double()
assert L.dlg_probe_mfma_f64(C.byref(v)) == 0, L.dlg_last_error()
print(f"fp64 MFMA (v_mfma_f64_16x16x4_f64) sustained: {v.value:.2f} TFLOP/s")
assert L.dlg_probe_hbm_copy(C.byref(v)) == 0, L.dlg_last_error()
print(f"HBM copy (read+write bytes): {v.value:.1f} GB/s")
