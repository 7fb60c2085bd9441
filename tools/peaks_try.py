import ctypes as C, os
lib = C.CDLL(os.path.join("vqa_amd", "csrc", "libvqapeaks.so"))
v = C.c_double(0.0); var = (C.c_double * 4)()
print(lib.vqa_peak_hbm_copy_variants(0, C.byref(v), var), v.value, list(var))
for fn in ("vqa_peak_fp64_fma", "vqa_peak_lds"):
    print(fn, getattr(lib, fn)(0, C.byref(v)), v.value)
