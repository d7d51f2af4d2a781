"""invesalius3_amd -- MI355X-native implementation of the InVesalius dense-voxel hot path.

Host-side mirror of the reference interface for that path only:

    invesalius3_amd.invesalius_rs      <-> invesalius_rs/__init__.py (floodfill / mips names)
    invesalius3_amd.slice_             <-> invesalius/data/slice_.py threshold + projection call sites
    invesalius3_amd.surface_process    <-> invesalius/data/surface_process.py create_surface_piece
    invesalius3_amd.watershed_process  <-> invesalius/data/watershed_process.py do_watershed
    invesalius3_amd.device             resident-volume pipeline (upload once; threshold -> grow -> MC in HBM)
    invesalius3_amd.parallel           Z-slab sharding across GPUs (one process per GPU)

All arithmetic runs in hand-written HIP kernels inside libivx.so (C ABI: include/ivx.h), reached through
ctypes.  No PyTorch, no VTK, and no CPU fallback.
"""
__version__ = "0.1.0"
