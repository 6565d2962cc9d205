"""trre_amd — MI355X-native transducer scan engine (scan-mode hot path of c0stya/trre).

Only what the hot path needs lives here:
  csrc/   HIP kernels (gfx950), the C++ pattern front end and the C ABI
  api.py  ctypes mirror of the reference's scan interface
  cli.py  `trre` / `trre_dft` work-alike entry points for scan mode
"""
from .api import (ENGINE_DFT, ENGINE_NFT, KERNEL_AUTO, KERNEL_BYTEMAP, KERNEL_NAMES, KERNEL_STREAM_GEN,  # noqa: F401
                  KERNEL_STREAM_LP, KERNEL_TILE_GEN, KERNEL_TILE_LP, Program, TrreError, build_library, shard_bounds)
