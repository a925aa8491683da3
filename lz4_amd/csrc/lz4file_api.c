/* lz4file_api.c -- stdio wrapper over the frame streaming API (include/lz4file.h; mirrors lib/lz4file.c of the
 * reference: same entry points, same error codes).  Plain host plumbing: the blocks themselves are coded by
 * LZ4F_compressUpdate / LZ4F_decompress of this library, on the device. */
#include <stdlib.h>
#include <string.h>
#include "../../include/lz4file.h"

static LZ4F_errorCode_t file_err(LZ4F_errorCodes c) { return (LZ4F_errorCode_t)-(ptrdiff_t)c; }

static size_t block_bytes_of(LZ4F_blockSizeID_t id, int* ok) {
    *ok = 1;
    switch (id) {
        case LZ4F_default: case LZ4F_max64KB: return (size_t)64 << 10;
        case LZ4F_max256KB: return (size_t)256 << 10;
        case LZ4F_max1MB: return (size_t)1 << 20;
        case LZ4F_max4MB: return (size_t)4 << 20;
        default: *ok = 0; return 0;
    }
}

/* ------------------------------------------------------------------ reading */
struct LZ4_readFile_s {
    LZ4F_dctx* dctx;
    FILE* fp;
    unsigned char* in;        /* compressed bytes read from the file, not yet consumed: in[at, have) */
    size_t cap, at, have;
    int eof;                  /* the file has nothing more; the decoder may (it decodes in batches: output can trail the input) */
};

static void read_free(LZ4_readFile_t* r) {
    if (!r) return;
    LZ4F_freeDecompressionContext(r->dctx);
    free(r->in);
    free(r);
}

LZ4F_errorCode_t LZ4F_readOpen(LZ4_readFile_t** out, FILE* fp) {
    unsigned char head[LZ4F_HEADER_SIZE_MAX];
    LZ4_readFile_t* r;
    LZ4F_frameInfo_t info;
    size_t got, used;
    LZ4F_errorCode_t rc;
    int ok;
    if (out == NULL || fp == NULL) return file_err(LZ4F_ERROR_parameter_null);
    *out = NULL;
    r = (LZ4_readFile_t*)calloc(1, sizeof *r);
    if (!r) return file_err(LZ4F_ERROR_allocation_failed);
    rc = LZ4F_createDecompressionContext(&r->dctx, LZ4F_VERSION);
    if (LZ4F_isError(rc)) { read_free(r); return rc; }
    r->fp = fp;
    got = fread(head, 1, sizeof head, fp);
    if (got < LZ4F_HEADER_SIZE_MIN + LZ4F_ENDMARK_SIZE) { read_free(r); return file_err(LZ4F_ERROR_io_read); }   /* shorter than an empty frame */
    used = got;
    rc = LZ4F_getFrameInfo(r->dctx, &info, head, &used);
    if (LZ4F_isError(rc)) { read_free(r); return rc; }
    r->cap = block_bytes_of(info.blockSizeID, &ok);
    if (!ok) { read_free(r); return file_err(LZ4F_ERROR_maxBlockSize_invalid); }
    if (r->cap < sizeof head) r->cap = sizeof head;
    r->in = (unsigned char*)malloc(r->cap);
    if (!r->in) { read_free(r); return file_err(LZ4F_ERROR_allocation_failed); }
    /* what was read behind the header belongs to the first block */
    memcpy(r->in, head + used, got - used);
    r->at = 0; r->have = got - used;
    *out = r;
    return LZ4F_OK_NoError;                      /* (lz4file.c:96-140: the result of LZ4F_createDecompressionContext, not getFrameInfo's size hint) */
}

size_t LZ4F_read(LZ4_readFile_t* r, void* buf, size_t size) {
    unsigned char* p = (unsigned char*)buf;
    size_t done = 0;
    if (r == NULL || buf == NULL) return file_err(LZ4F_ERROR_parameter_null);
    while (done < size) {
        size_t dn = size - done, sn;
        size_t rc;
        if (r->at == r->have && !r->eof) {
            const size_t got = fread(r->in, 1, r->cap, r->fp);
            if (got == 0) { if (ferror(r->fp)) return file_err(LZ4F_ERROR_io_read); r->eof = 1; }
            else { r->at = 0; r->have = got; }
        }
        sn = r->have - r->at;
        rc = LZ4F_decompress(r->dctx, p + done, &dn, r->in + r->at, &sn, NULL);       /* (sn == 0 at the end of the file: drains what is decoded) */
        if (LZ4F_isError(rc)) return rc;
        r->at += sn; done += dn;
        if (dn == 0 && sn == 0 && (r->eof || rc == 0)) break;                           /* nothing more to give */
    }
    return done;
}

LZ4F_errorCode_t LZ4F_readClose(LZ4_readFile_t* r) {
    if (r == NULL) return file_err(LZ4F_ERROR_parameter_null);
    read_free(r);
    return LZ4F_OK_NoError;
}

/* ------------------------------------------------------------------ writing */
struct LZ4_writeFile_s {
    LZ4F_cctx* cctx;
    FILE* fp;
    unsigned char* out;
    size_t out_cap, chunk;     /* chunk: the most LZ4F_compressUpdate is given at a time (one block) */
    LZ4F_errorCode_t err;      /* a failed fwrite ends the frame: close only frees */
};

static void write_free(LZ4_writeFile_t* w) {
    if (!w) return;
    LZ4F_freeCompressionContext(w->cctx);
    free(w->out);
    free(w);
}

LZ4F_errorCode_t LZ4F_writeOpen(LZ4_writeFile_t** out, FILE* fp, const LZ4F_preferences_t* prefs) {
    unsigned char head[LZ4F_HEADER_SIZE_MAX];
    LZ4_writeFile_t* w;
    size_t rc;
    int ok;
    if (out == NULL || fp == NULL) return file_err(LZ4F_ERROR_parameter_null);
    *out = NULL;
    w = (LZ4_writeFile_t*)calloc(1, sizeof *w);
    if (!w) return file_err(LZ4F_ERROR_allocation_failed);
    w->chunk = block_bytes_of(prefs ? prefs->frameInfo.blockSizeID : LZ4F_default, &ok);
    if (!ok) { write_free(w); return file_err(LZ4F_ERROR_maxBlockSize_invalid); }
    w->out_cap = LZ4F_compressBound(w->chunk, prefs);
    w->out = (unsigned char*)malloc(w->out_cap);
    if (!w->out) { write_free(w); return file_err(LZ4F_ERROR_allocation_failed); }
    rc = LZ4F_createCompressionContext(&w->cctx, LZ4F_VERSION);
    if (LZ4F_isError(rc)) { write_free(w); return rc; }
    rc = LZ4F_compressBegin(w->cctx, head, sizeof head, prefs);
    if (LZ4F_isError(rc)) { write_free(w); return rc; }
    if (fwrite(head, 1, rc, fp) != rc) { write_free(w); return file_err(LZ4F_ERROR_io_write); }
    w->fp = fp; w->err = LZ4F_OK_NoError;
    *out = w;
    return LZ4F_OK_NoError;
}

size_t LZ4F_write(LZ4_writeFile_t* w, const void* buf, size_t size) {
    const unsigned char* p = (const unsigned char*)buf;
    size_t left = size;
    if (w == NULL || buf == NULL) return file_err(LZ4F_ERROR_parameter_null);
    while (left) {
        const size_t n = left < w->chunk ? left : w->chunk;
        const size_t rc = LZ4F_compressUpdate(w->cctx, w->out, w->out_cap, p, n, NULL);
        if (LZ4F_isError(rc)) { w->err = rc; return rc; }
        if (fwrite(w->out, 1, rc, w->fp) != rc) { w->err = file_err(LZ4F_ERROR_io_write); return w->err; }
        p += n; left -= n;
    }
    return size;
}

LZ4F_errorCode_t LZ4F_writeClose(LZ4_writeFile_t* w) {
    LZ4F_errorCode_t ret = LZ4F_OK_NoError;
    if (w == NULL) return file_err(LZ4F_ERROR_parameter_null);
    if (w->err == LZ4F_OK_NoError) {
        const size_t rc = LZ4F_compressEnd(w->cctx, w->out, w->out_cap, NULL);
        if (LZ4F_isError(rc)) ret = rc;
        else if (fwrite(w->out, 1, rc, w->fp) != rc) ret = file_err(LZ4F_ERROR_io_write);
    }
    write_free(w);
    return ret;
}
