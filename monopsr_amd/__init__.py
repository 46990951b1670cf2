"""monopsr_amd -- MonoPSR's per-instance hot path (ResNet-101 crop trunk, squash + map decoder, centroid/shape
heads, Chamfer and approximate-EMD point-cloud ops) as hand-written gfx950 HIP kernels behind the reference's
operator API.  The compute lives in libmonopsr_hip.so (C ABI: include/monopsr_hip.h); this package is the
host-side mirror of the reference's Python interface for that path.  PyTorch provides device memory, streams and
torch.distributed only.
"""
__version__ = "0.1.0"
