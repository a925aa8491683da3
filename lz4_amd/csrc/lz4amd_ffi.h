/* lz4amd_ffi.h -- the thin extern "C" seam between the C host code (lz4amd_batch.c, lz4_api.c,
 * lz4frame_api.c) and the HIP translation unit (lz4amd_device.hip).  Everything the host code
 * needs from the HIP runtime goes through these functions, so the host side stays plain C. */
#ifndef LZ4AMD_FFI_H
#define LZ4AMD_FFI_H
#include <stddef.h>
#include "lz4amd_params.h"

#ifdef __cplusplus
extern "C" {
#endif

int         lz4amd_hip_init(int device, int* n_cus);            /* 0 ok */
const char* lz4amd_hip_errstr(void);
int         lz4amd_hip_use_device(int device);                  /* hipSetDevice for the calling thread */
void*       lz4amd_hip_malloc(size_t bytes);
void        lz4amd_hip_free(void* d);
int         lz4amd_hip_h2d(void* d, const void* h, size_t n, void* stream);
int         lz4amd_hip_d2h(void* h, const void* d, size_t n, void* stream);
int         lz4amd_hip_memset(void* d, int value, size_t n, void* stream);
int         lz4amd_hip_sync(void* stream);
void*       lz4amd_hip_host_alloc(size_t bytes);                /* page-locked, device-accessible host memory */
void        lz4amd_hip_host_free(void* p);
void*       lz4amd_hip_stream_create(void);                     /* a non-blocking stream, NULL on failure */
void        lz4amd_hip_stream_destroy(void* stream);
void*       lz4amd_hip_event_create(void);
void        lz4amd_hip_event_destroy(void* ev);
int         lz4amd_hip_event_record(void* ev, void* stream);
int         lz4amd_hip_event_sync(void* ev);
float       lz4amd_hip_event_ms(void* start, void* stop);

/* kernel geometry facts the host needs for sizing */
size_t      lz4amd_hip_dec_scratch_bytes(unsigned max_csize, unsigned max_out);
size_t      lz4amd_hip_hc_scratch_bytes(unsigned max_src);

/* launches (asynchronous on `stream`) */
int lz4amd_hip_launch_decompress(const lz4amd_dec_params* p, unsigned grid, void* stream);
int lz4amd_hip_launch_xxh32(const lz4amd_xxh_params* p, void* stream);
int lz4amd_hip_launch_gather(const lz4amd_gather_params* p, void* stream);
int lz4amd_hip_launch_spec_fill(const lz4amd_spec_params* p, void* stream);
int lz4amd_hip_launch_spec(const lz4amd_spec_params* p, const lz4amd_dec_params* dec, unsigned dec_grid,
                           const lz4amd_dec_params* dec_b, unsigned dec_b_grid, unsigned max_cap, void* stream);      /* dec_b: NULL, or the launch of the second copies */
int lz4amd_hip_launch_stream_copy(void* d_dst, const void* d_src, size_t bytes, unsigned grid, unsigned variant, void* stream);   /* variant: granules per thread and trip / non-temporal (lz4amd_device.hip) */
int lz4amd_hip_launch_compress(const lz4amd_comp_params* p, unsigned grid, void* stream);
int lz4amd_hip_launch_compress_hc(const lz4amd_hc_params* p, unsigned grid, void* stream);

#ifdef __cplusplus
}
#endif
#endif
