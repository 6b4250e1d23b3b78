"""Micro-probes run on the MI355X: fp64 MFMA issue rate and HBM copy ceiling."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libdogleg_amd import capi

L = capi.lib()
v = C.c_double()
c3 = (C.c_double * 3)()
L.dlg_probe_mfma_f64_clock.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
assert L.dlg_probe_mfma_f64_clock(C.byref(v), c3) == 0, L.dlg_last_error()
print(f"fp64 MFMA (v_mfma_f64_16x16x4_f64) sustained: {v.value:.2f} TFLOP/s")
print(f"  shader clock during that loop (s_memtime against the 100 MHz s_memrealtime, one wave in the middle of the grid): {c3[0]:.0f} MHz")
print(f"  clocks per MFMA: {c3[1]:.1f} per wave, {c3[2]:.1f} per SIMD (four waves a SIMD, four independent accumulators each, one resident round)")
peak = 256 * 4 * 2048.0 / 64.0
print(f"  => at that clock a 64-clock MFMA gives {peak * c3[0] * 1e6 / 1e12:.1f} TFLOP/s (the datasheet's 78.6 is 64 clocks at 2400 MHz); "
      f"measured / that = {v.value / (peak * c3[0] * 1e6 / 1e12):.2f}")
assert L.dlg_probe_hbm_copy(C.byref(v)) == 0, L.dlg_last_error()
print(f"HBM copy (read+write bytes): {v.value:.1f} GB/s")
