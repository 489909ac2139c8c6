"""The diagnostic build of the library (-DMOC_STAMPS: in-kernel phase stamps), built where the diagnostic runs --
it is never part of the tree that ships.  Import BEFORE moc_amd: sets MOC_HIP_LIB."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "build", "libmoc_hip_stamps.so")


def ensure():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "moc_amd", "csrc"), "stamps"])
    os.environ.setdefault("MOC_HIP_LIB", LIB)
    return LIB


ensure()
