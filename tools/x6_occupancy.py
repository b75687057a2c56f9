import ctypes as C, sys
sys.path.insert(0, "/root/repo")
import torch
from bert_vits2_amd import lib as L
lib = L.load()
torch.zeros(1, device="cuda")
a = (C.c_int * 4)()
lib.bv2_test_x6_occupancy.restype = None
lib.bv2_test_x6_occupancy.argtypes = [C.POINTER(C.c_int)]
lib.bv2_test_x6_occupancy(a)
print("x6 occupancy (workgroups per CU): 128x64", a[0], " 128x64+loaders", a[1], " 64x128", a[2], " 32x256", a[3])
