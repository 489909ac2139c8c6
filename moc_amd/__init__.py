"""moc_amd -- MI355X (gfx950) engine for the MOC hot path.

Only the per-slide classifier-bank -> patch-filtering -> meta-learner -> top-K
path of xmed-lab/MOC lives here.  Compute is in
hand-written HIP kernels behind the C ABI declared in include/moc_hip.h
(moc_amd/csrc, built to moc_amd/libmoc_hip.so); this package is the host-side
mirror of the reference's Python interface for that path.

Importing the package does not load the native library; the first call into a
kernel does, and raises if it is missing -- there is no CPU fallback.
"""
__version__ = "0.1.0"
