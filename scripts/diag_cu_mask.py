"""What moc_cu_census finds on this device -- (XCD, HW_ID[15:8]) slots per XCD and shader engine -- and which of them the
reserved table names for the look-ahead score pass (engine.choose_reserved_slots).
usage: python scripts/diag_cu_mask.py [n_reserved ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moc_amd import engine

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    slots = engine.cu_slots(dev)
    print(f"{len(slots)} compute units seen ({torch.cuda.get_device_properties(dev).multi_processor_count} reported)")
    for x in sorted({t[0] for t in slots}):
        per_se = {}
        for _, s in (t for t in slots if t[0] == x):
            per_se.setdefault(s >> 5, []).append(s & 31)
        print(f"  XCD {x}: " + "  ".join(f"SE{se}: CUs {sorted(v)}" for se, v in sorted(per_se.items())))
    for n in [int(v) for v in sys.argv[1:]] or [64]:
        chosen = engine.choose_reserved_slots(slots, n)
        print(f"reserved {n}: " + ", ".join(f"x{x}:se{s >> 5}:cu{s & 31}" for x, s in chosen[:16]) + (" ..." if len(chosen) > 16 else ""))
