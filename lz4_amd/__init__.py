"""lz4_amd -- MI355X-native LZ4 block codec (HIP kernels behind the liblz4 C ABI).

The product is lz4_amd/liblz4_amd.so (C ABI declared in include/*.h).  This package is
only the Python-side binding used by tests and bench.py: ctypes over that C ABI, with torch
tensors as device memory.  There is no CPU codec here; importing works without a GPU, calling
anything that needs the device raises.
"""
from .api import (Context, Plan, OP_COMPRESS, OP_DECOMPRESS, OP_COMPRESS_HC, OP_XXH32, OP_GATHER, compress_bound, hint_bytes, lib, lib_path,
                  BlockTable, compress_blocks, decompress_blocks, Lz4AmdError)

__all__ = ["Context", "Plan", "OP_COMPRESS", "OP_DECOMPRESS", "OP_COMPRESS_HC", "OP_XXH32", "OP_GATHER", "compress_bound", "hint_bytes", "lib", "lib_path",
           "BlockTable", "compress_blocks", "decompress_blocks", "Lz4AmdError"]
