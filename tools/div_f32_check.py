# Exhaustive check behind colour_device.h cbrt_lerp<FINITE>: (float) ((double) A / C) == fmaf(A, Rhi, A * Rlo)
# for every float A in [2^-8, 2^27) and C = X0, Y0, Z0 of D65 (XYZ2Lab.c:120-122).  CPU only, ~40 s.
import numpy as np, sys
# n_ref = float32( float64(A) / C )   vs   n_new = fmaf(A, Rhi, fmul(A, Rlo))  for float32 A
def check(C, lo_exp, hi_exp):
    R = 1.0 / np.float64(C)
    Rhi = np.float32(R)
    Rlo = np.float32(R - np.float64(Rhi))
    bad = 0
    m = np.arange(1<<23, 1<<24, dtype=np.int64)
    for e in range(lo_exp, hi_exp+1):
        A = (m.astype(np.float64) * 2.0**(e-23)).astype(np.float32)
        ref = (A.astype(np.float64) / np.float64(C)).astype(np.float32)
        t = (A * Rlo).astype(np.float32)   # fmul in f32
        # fmaf(A, Rhi, t): exact product in longdouble (48 bits) + t exact -> one rounding to f32
        prod = A.astype(np.longdouble) * np.longdouble(Rhi)
        new = (prod + t.astype(np.longdouble)).astype(np.float32)
        nb = int((ref.view(np.uint32) != new.view(np.uint32)).sum())
        bad += nb
        if nb: print(C, e, nb)
    return bad, Rhi, Rlo
assert np.finfo(np.longdouble).nmant >= 63
for C in (95.0470, 100.0, 108.8827):
    b, Rhi, Rlo = check(C, -8, 26)
    print(C, "bad", b, float(Rhi).hex(), float(Rlo).hex())
