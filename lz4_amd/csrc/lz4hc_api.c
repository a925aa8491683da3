/*
 * lz4hc_api.c -- the reference's one-shot high-compression C ABI (lib/lz4hc.h) on top of the GPU
 * batch codec.  Same names, argument meaning and return conventions as the reference
 * (lz4hc.c:1519 LZ4_compress_HC, 1503 LZ4_compress_HC_extStateHC, 1486 LZ4_sizeofStateHC): host
 * pointers in and out, the block makes a round trip through HBM.  No CPU codec: without a usable
 * HIP device the calls return 0 after a message on stderr.
 */
#include "../../include/lz4hc.h"
#include "../../include/lz4amd.h"
#include "lz4amd_internal.h"
#include <stdint.h>

int LZ4_compress_HC(const char* src, char* dst, int srcSize, int dstCapacity, int compressionLevel)
{
    if (srcSize < 0 || (unsigned)srcSize > (unsigned)LZ4_MAX_INPUT_SIZE) return 0;   /* lz4hc.c:1403 */
    if (dst == NULL || dstCapacity <= 0) return 0;
    if (src == NULL && srcSize != 0) return 0;
    return lz4amd_run_one(LZ4AMD_OP_COMPRESS_HC, src, dst, srcSize, dstCapacity, compressionLevel, 0);
}

int LZ4_sizeofStateHC(void) { return LZ4_STREAMHC_MINSIZE; }

int LZ4_compress_HC_extStateHC(void* stateHC, const char* src, char* dst, int srcSize, int maxDstSize, int compressionLevel)
{
    if (stateHC == NULL || ((uintptr_t)stateHC & (sizeof(void*) - 1)) != 0) return 0;   /* lz4hc.c:1506-1508 */
    return LZ4_compress_HC(src, dst, srcSize, maxDstSize, compressionLevel);
}
