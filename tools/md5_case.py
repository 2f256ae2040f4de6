#!/usr/bin/env python3
"""MD5 of the library on the GPU box's host: alone, and next to memory-heavy neighbours."""
import ctypes as C, os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_bindings
B = load_bindings(); L = B.lib()
L.lrzgpu_hash_buffer.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_char_p]
n = 2 << 30
a = np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8)
out = C.create_string_buffer(64)
for rep in range(2):
    t = time.time(); L.lrzgpu_hash_buffer(1, a.ctypes.data, n, out); dt = time.time() - t
print("MD5 alone: %.2f GB/s" % (n / dt / 1e9))
stop = False
def copier():
    b = np.empty(256 << 20, dtype=np.uint8)
    while not stop:
        b[:] = a[:256 << 20]
th = [threading.Thread(target=copier) for _ in range(8)]
for t in th: t.start()
t = time.time(); L.lrzgpu_hash_buffer(1, a.ctypes.data, n, out); dt = time.time() - t
stop = True
for t in th: t.join()
print("MD5 next to 8 memcpy threads: %.2f GB/s" % (n / dt / 1e9))
import torch
p = torch.empty(n, dtype=torch.uint8, pin_memory=True)
p.copy_(torch.from_numpy(a))
t = time.time(); L.lrzgpu_hash_buffer(1, p.data_ptr(), n, out); dt = time.time() - t
print("MD5 of pinned (hipHostMalloc) memory: %.2f GB/s" % (n / dt / 1e9))
t = time.time(); b2 = p.numpy().copy(); dt = time.time() - t
print("memcpy out of pinned memory: %.2f GB/s" % (n / dt / 1e9))
