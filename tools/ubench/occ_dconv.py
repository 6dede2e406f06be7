import ctypes as C, torch
torch.zeros(1, device="cuda")
lib = C.CDLL("/root/repo/tools/ubench/libdconv_prof.so")
for mg in (1, 2):
    for pool in (0, 1):
        print("mg", mg, "pool", pool, "blocks/CU", lib.mfr_dconv_occupancy(mg, pool))
