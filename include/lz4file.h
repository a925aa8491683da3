/* lz4file.h -- stdio wrapper over the frame streaming API of liblz4_amd.
 *
 * Same names, argument meaning and error behaviour as the reference's lib/lz4file.h:49-88 (LZ4F_readOpen / LZ4F_read /
 * LZ4F_readClose, LZ4F_writeOpen / LZ4F_write / LZ4F_writeClose), so that callers of the reference's file helper
 * (examples/fileCompress.c) link unchanged.  Host-side plumbing only: every block goes through LZ4F_compressUpdate /
 * LZ4F_decompress of this library, i.e. through the device codec. */
#ifndef LZ4AMD_LZ4FILE_H
#define LZ4AMD_LZ4FILE_H
#include <stdio.h>
#include "lz4frame.h"
#if defined(__cplusplus)
extern "C" {
#endif

typedef struct LZ4_readFile_s LZ4_readFile_t;
typedef struct LZ4_writeFile_s LZ4_writeFile_t;

/* Opens a frame for reading from fp (positioned at the frame's magic number).  *lz4fRead is NULL on failure.
 * lz4file.c:73-138: NULL arguments -> parameter_null, a file shorter than a minimal frame -> io_read. */
LZ4F_errorCode_t LZ4F_readOpen(LZ4_readFile_t** lz4fRead, FILE* fp);
/* Up to `size` decoded bytes into buf; returns how many (0 at the end of the frame) or an error code (lz4file.c:140-181). */
size_t LZ4F_read(LZ4_readFile_t* lz4fRead, void* buf, size_t size);
LZ4F_errorCode_t LZ4F_readClose(LZ4_readFile_t* lz4fRead);

/* Writes the frame header for prefsPtr (NULL = defaults) to fp (lz4file.c:217-279). */
LZ4F_errorCode_t LZ4F_writeOpen(LZ4_writeFile_t** lz4fWrite, FILE* fp, const LZ4F_preferences_t* prefsPtr);
/* Compresses `size` bytes of buf into the frame; returns size or an error code (lz4file.c:281-315). */
size_t LZ4F_write(LZ4_writeFile_t* lz4fWrite, const void* buf, size_t size);
/* Ends the frame (end mark, content checksum) and frees the state (lz4file.c:317-341). */
LZ4F_errorCode_t LZ4F_writeClose(LZ4_writeFile_t* lz4fWrite);

#if defined(__cplusplus)
}
#endif
#endif
