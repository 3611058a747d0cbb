import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import saugns_amd as sa
L = sa.lib()
L.sauAmd_kat_div_device.restype = C.c_longlong
L.sauAmd_kat_div_device.argtypes = [C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
for w in range(12):
    fb = C.c_uint32()
    print(w, [L.sauAmd_kat_div_device(w, v, C.byref(fb)) for v in (0, 1)], hex(fb.value))
