"""trre_amd — MI355X-native transducer scan engine (scan-mode hot path of c0stya/trre).

Only what the hot path needs lives here:
  csrc/       HIP kernels (gfx950), the C++ pattern front end, the C ABI and the `trre` / `trre_dft`
              command-line work-alikes (built into bin/)
  api.py      ctypes mirror of the reference's scan interface
  sharded.py  line sharding over ranks (one process per GPU)
"""
from .api import (ENGINE_DFT, ENGINE_NFT, KERNEL_AUTO, KERNEL_BACKTRACK, KERNEL_BYTEMAP, KERNEL_DFT_LAZY, KERNEL_GUIDED_GEN, KERNEL_GUIDED_LP,  # noqa: F401
                  KERNEL_NAMES, KERNEL_STREAM_GEN, KERNEL_STREAM_LP, KERNEL_TILE_GEN, KERNEL_TILE_LP, Program, TrreError,
                  build_library, shard_bounds)
