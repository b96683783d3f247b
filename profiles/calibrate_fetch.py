#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE calibration on known byte counts (MI355X_MICROARCH.md §HBM: 'calibrate on a known byte count in
your own access pattern').  Run under rocprofv3 --pmc FETCH_SIZE (and again with WRITE_SIZE):
  pattern A: wide coalesced streaming read+write   (torch clone of 1 GiB)          -> the ray-queue stream
  pattern B: random 64-B rows out of a 2 GiB table  (index_select, 8 Mi rows = 512 MiB read) -> BVH node / triangle lines
"""
import torch
torch.manual_seed(0)
dev = torch.device("cuda")
a = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()       # 1 GiB
table = torch.empty((32 * 1024 * 1024, 16), dtype=torch.float32, device=dev).normal_()  # 2 GiB, 64-B rows
idx = torch.randint(0, table.shape[0], (8 * 1024 * 1024,), device=dev)
torch.cuda.synchronize()
for _ in range(3):
    b = a.clone()                       # A: 1 GiB read, 1 GiB written
    torch.cuda.synchronize()
for _ in range(3):
    c = torch.index_select(table, 0, idx)   # B: 512 MiB of random 64-B rows read, 512 MiB written
    torch.cuda.synchronize()
print("done", float(b[0]), float(c[0, 0]))
