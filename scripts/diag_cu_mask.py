"""Which compute units a CU-masked side stream reaches (moc_side_stream_create + moc_cu_census), per XCD.
usage: python scripts/diag_cu_mask.py [n_cus ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moc_amd import engine
from moc_amd._lib import check, lib


def census(n_cus, n_wg=8192, hold_us=30):
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream()
    hist = torch.zeros(16 * 256, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        check(lib().moc_cu_census(hist.data_ptr(), n_wg, hold_us, C.c_void_p(s.cuda_stream)), "moc_cu_census")
    s.synchronize()
    h = hist.cpu().view(16, 256)
    per_xcc = [(int((h[x] > 0).sum()), int(h[x].sum())) for x in range(16) if int(h[x].sum())]
    return int((h > 0).sum()), per_xcc


if __name__ == "__main__":
    for n in [int(v) for v in sys.argv[1:]] or [0, 224, 192, 128, 32]:
        used, per = census(n)
        print(f"n_cus={n or 'all'}: distinct (xcc, se/sh/cu) slots used {used}; per XCD (slots, workgroups): {per}", flush=True)
