"""mimir_amd: MI355X (gfx950) implementation of Mimir's map-side shuffle
transform (hash -> destination -> stable pack), behind a C ABI.

  mimir_amd.capi     ctypes binding of include/mimir_gpu_shuffle.h
  mimir_amd.packer   device-resident packer on torch tensors
  mimir_amd.datagen  seeded synthetic record streams
  mimir_amd.build    builds the HIP library in-tree
"""
__all__ = ["capi", "packer", "datagen", "build"]
