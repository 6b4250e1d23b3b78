"""Micro-probes run on the MI355X: fp64 MFMA issue rate (with the shader clock it ran at) and HBM copy ceiling."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libdogleg_amd import capi

L = capi.lib()
v = C.c_double()
c3 = (C.c_double * 3)()
L.dlg_probe_mfma_f64_waves.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
print("fp64 MFMA (v_mfma_f64_16x16x4_f64), every CU filled with ONE resident round of workgroups, four independent accumulators a wave;")
print("shader clock = s_memtime against the 100 MHz s_memrealtime over the loop of one wave in the middle of the grid")
print("waves/SIMD  TFLOP/s  shader MHz  clocks per MFMA and SIMD  TFLOP/s if an MFMA took 64 clocks at that clock")
best = 0.0
for w in (1, 2, 4, 8):
    assert L.dlg_probe_mfma_f64_waves(w, C.byref(v), c3) == 0, L.dlg_last_error()
    ideal = 256 * 4 * 2048.0 / 64.0 * c3[0] * 1e6 / 1e12
    print(f"{w:10d}  {v.value:7.2f}  {c3[0]:10.0f}  {c3[2]:24.1f}  {ideal:8.1f}")
    best = max(best, v.value)
print(f"fp64 MFMA sustained (best of the above): {best:.2f} TFLOP/s; datasheet 78.6 = 64 clocks per MFMA and SIMD at 2400 MHz")
assert L.dlg_probe_hbm_copy(C.byref(v)) == 0, L.dlg_last_error()
print(f"HBM copy (read+write bytes): {v.value:.1f} GB/s")
