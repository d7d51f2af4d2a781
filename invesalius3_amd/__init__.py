"""invesalius3_amd -- MI355X-native implementation of the InVesalius dense-voxel hot path.

Host-side mirror of the reference interface for that path and the stages next to it:

    invesalius3_amd.invesalius_rs      <-> invesalius_rs/__init__.py (floodfill, mips, transforms, mesh, mask-editing names)
    invesalius3_amd.slice_             <-> invesalius/data/slice_.py threshold / projection / boolean / measurement call sites
    invesalius3_amd.mask               <-> invesalius/data/mask.py fill_holes_auto
    invesalius3_amd.styles             <-> invesalius/data/styles.py region-growing tool (do_3d_seg, do_rg_confidence)
    invesalius3_amd.surface_process    <-> invesalius/data/surface_process.py create_surface_piece, join_process_surface
    invesalius3_amd.watershed_process  <-> invesalius/data/watershed_process.py do_watershed
    invesalius3_amd.project            <-> invesalius/project.py .inv3 container (read / write, no wx / VTK)
    invesalius3_amd.headless           command-line driver: project in -> mask / surface / measurements / STL out
    invesalius3_amd.device             resident-volume pipeline (upload once; threshold -> grow -> MC in HBM)
    invesalius3_amd.parallel           Z-slab sharding across GPUs (one process per GPU)

All arithmetic runs in hand-written HIP kernels inside libivx.so (C ABI: include/ivx.h), reached through
ctypes.  No PyTorch, no VTK, and no CPU fallback.
"""
__version__ = "0.1.0"
