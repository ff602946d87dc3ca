/*
 * cpu_bench.c -- the CPU leg of bench.py's `cpu_baseline`: ParCompress / ParDecompress-style
 * worker threads (src/par/compress.rs:279-294, src/par/decompress.rs:162-186) over a bounded
 * sample, timed natively (pthreads; no Python in the loop).  Test/bench infrastructure only --
 * nothing in the product path links or calls this file.
 *
 * compress: every worker owns a contiguous run of BGZF/Mgzip blocks of the slab and re-encodes
 *           them with the oracle (gzpx_oracle_encode_block) until the deadline.
 * inflate : every worker owns a contiguous run of blocks of the compressed stream and inflates +
 *           CRC-checks them with the libdeflate binary of the image (dlopen; the library gzp binds
 *           through libdeflater) until the deadline.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gzpx_oracle.h"

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    /* compress */
    const uint8_t *slab;
    size_t first_block, n_blocks, slab_len, block;
    int fmt, level, compat;
    /* inflate */
    const uint8_t *comp;
    const uint64_t *offs;
    const uint32_t *sizes;
    size_t hdr_len;
    void *(*alloc_d)(void);
    int (*inflate)(void *, const void *, size_t, void *, size_t, size_t *);
    uint32_t (*crc32)(uint32_t, const void *, size_t);
    void (*free_d)(void *);
    /* compress with the libdeflate binary */
    void *(*alloc_c)(int);
    size_t (*deflate)(void *, const void *, size_t, void *, size_t);
    void (*free_c)(void *);
    /* both */
    double deadline;
    uint64_t bytes;
    int failed;
} worker_t;

static void *compress_worker(void *arg) {
    worker_t *w = (worker_t *)arg;
    const size_t cap = w->block + w->block / 8 + 4096;
    uint8_t *out = (uint8_t *)malloc(cap);
    if (!out) {
        w->failed = 1;
        return NULL;
    }
    do {
        for (size_t b = 0; b < w->n_blocks; b++) {
            const size_t off = (w->first_block + b) * w->block;
            const size_t n = off + w->block <= w->slab_len ? w->block : w->slab_len - off;
            int err = 0;
            if (gzpx_oracle_encode_block(w->fmt, w->level, w->compat, w->slab + off, n, 0, out, cap, &err) == 0 ||
                err)
                w->failed = 1;
            w->bytes += n;
        }
    } while (now_s() < w->deadline);
    free(out);
    return NULL;
}

/* bgzf::compress with the real library: libdeflate_deflate_compress into buf[18..] + libdeflate_crc32
 * (src/bgzf.rs:214-225); header and footer bytes are not worth timing */
static void *compress_ref_worker(void *arg) {
    worker_t *w = (worker_t *)arg;
    const size_t cap = w->block + w->block / 8 + 4096;
    uint8_t *out = (uint8_t *)malloc(cap);
    void *c = w->alloc_c(w->level);
    if (!out || !c) {
        w->failed = 1;
        return NULL;
    }
    uint32_t sink = 0;
    do {
        for (size_t b = 0; b < w->n_blocks; b++) {
            const size_t off = (w->first_block + b) * w->block;
            const size_t n = off + w->block <= w->slab_len ? w->block : w->slab_len - off;
            if (w->deflate(c, w->slab + off, n, out + 18, cap - 26) == 0) w->failed = 1;
            sink ^= w->crc32(0, w->slab + off, n);
            w->bytes += n;
        }
    } while (now_s() < w->deadline);
    out[0] = (uint8_t)sink;
    w->free_c(c);
    free(out);
    return NULL;
}

static void *inflate_worker(void *arg) {
    worker_t *w = (worker_t *)arg;
    void *d = w->alloc_d();
    size_t cap = 1 << 16;
    uint8_t *out = (uint8_t *)malloc(cap);
    if (!d || !out) {
        w->failed = 1;
        return NULL;
    }
    do {
        for (size_t b = 0; b < w->n_blocks; b++) {
            const uint8_t *blk = w->comp + w->offs[w->first_block + b];
            const uint32_t sz = w->sizes[w->first_block + b];
            const uint8_t *f = blk + sz - 8;
            const uint32_t want = (uint32_t)f[0] | ((uint32_t)f[1] << 8) | ((uint32_t)f[2] << 16) | ((uint32_t)f[3] << 24);
            const uint32_t isize = (uint32_t)f[4] | ((uint32_t)f[5] << 8) | ((uint32_t)f[6] << 16) | ((uint32_t)f[7] << 24);
            if (isize > cap) {
                free(out);
                cap = isize;
                out = (uint8_t *)malloc(cap);
                if (!out) {
                    w->failed = 1;
                    return NULL;
                }
            }
            if (isize) {
                if (w->inflate(d, blk + w->hdr_len, sz - w->hdr_len - 8, out, isize, NULL) != 0) w->failed = 1;
                if (w->crc32(0, out, isize) != want) w->failed = 1;
            }
            w->bytes += isize;
        }
    } while (now_s() < w->deadline);
    w->free_d(d);
    free(out);
    return NULL;
}

static int run(worker_t *ws, int threads, void *(*fn)(void *), double *elapsed, uint64_t *bytes) {
    pthread_t *tid = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    if (!tid) return -1;
    const double t0 = now_s();
    int started = 0;
    for (int i = 0; i < threads; i++) {
        if (pthread_create(&tid[i], NULL, fn, &ws[i]) != 0) break;
        started++;
    }
    int failed = started != threads;
    *bytes = 0;
    for (int i = 0; i < started; i++) {
        pthread_join(tid[i], NULL);
        *bytes += ws[i].bytes;
        failed |= ws[i].failed;
    }
    *elapsed = now_s() - t0;
    free(tid);
    return failed ? -1 : 0;
}

/* Returns 0 on success; *bytes = input bytes compressed by all workers in *elapsed seconds. */
int gzpx_cpu_bench_compress(int fmt, int level, int compat, size_t block, const uint8_t *slab, size_t slab_len,
                            int threads, double wall_s, double *elapsed, uint64_t *bytes, int *threads_used) {
    const size_t nb = (slab_len + block - 1) / block;
    if (threads < 1 || nb == 0) return -1;
    if ((size_t)threads > nb) threads = (int)nb;
    worker_t *ws = (worker_t *)calloc((size_t)threads, sizeof(worker_t));
    if (!ws) return -1;
    const double deadline = now_s() + wall_s;
    for (int i = 0; i < threads; i++) {
        ws[i].slab = slab;
        ws[i].slab_len = slab_len;
        ws[i].block = block;
        ws[i].first_block = nb * (size_t)i / (size_t)threads;
        ws[i].n_blocks = nb * (size_t)(i + 1) / (size_t)threads - ws[i].first_block;
        ws[i].fmt = fmt;
        ws[i].level = level;
        ws[i].compat = compat;
        ws[i].deadline = deadline;
    }
    const int rc = run(ws, threads, compress_worker, elapsed, bytes);
    *threads_used = threads;
    free(ws);
    return rc;
}

/* The same with the libdeflate binary of the image doing the work (the library gzp calls).
 * Returns 0 on success, -2 when no libdeflate binary can be loaded on this box. */
int gzpx_cpu_bench_compress_ref(int level, size_t block, const uint8_t *slab, size_t slab_len, int threads,
                                double wall_s, double *elapsed, uint64_t *bytes, int *threads_used) {
    void *h = dlopen("libdeflate.so.0", RTLD_NOW);
    if (!h) h = dlopen("/lib/x86_64-linux-gnu/libdeflate.so.0", RTLD_NOW);
    if (!h) return -2;
    worker_t proto;
    memset(&proto, 0, sizeof(proto));
    *(void **)(&proto.alloc_c) = dlsym(h, "libdeflate_alloc_compressor");
    *(void **)(&proto.deflate) = dlsym(h, "libdeflate_deflate_compress");
    *(void **)(&proto.crc32) = dlsym(h, "libdeflate_crc32");
    *(void **)(&proto.free_c) = dlsym(h, "libdeflate_free_compressor");
    const size_t nb = (slab_len + block - 1) / block;
    if (!proto.alloc_c || !proto.deflate || !proto.crc32 || !proto.free_c || threads < 1 || nb == 0) return -2;
    if ((size_t)threads > nb) threads = (int)nb;
    worker_t *ws = (worker_t *)calloc((size_t)threads, sizeof(worker_t));
    if (!ws) return -1;
    const double deadline = now_s() + wall_s;
    for (int i = 0; i < threads; i++) {
        ws[i] = proto;
        ws[i].slab = slab;
        ws[i].slab_len = slab_len;
        ws[i].block = block;
        ws[i].first_block = nb * (size_t)i / (size_t)threads;
        ws[i].n_blocks = nb * (size_t)(i + 1) / (size_t)threads - ws[i].first_block;
        ws[i].level = level;
        ws[i].deadline = deadline;
    }
    const int rc = run(ws, threads, compress_ref_worker, elapsed, bytes);
    *threads_used = threads;
    free(ws);
    return rc;
}

/* Returns 0 on success, -2 when no libdeflate binary can be loaded on this box. */
int gzpx_cpu_bench_inflate(const uint8_t *comp, const uint64_t *offs, const uint32_t *sizes, size_t nb,
                           size_t hdr_len, int threads, double wall_s, double *elapsed, uint64_t *bytes,
                           int *threads_used) {
    void *h = dlopen("libdeflate.so.0", RTLD_NOW);
    if (!h) h = dlopen("/lib/x86_64-linux-gnu/libdeflate.so.0", RTLD_NOW);
    if (!h) return -2;
    worker_t proto;
    memset(&proto, 0, sizeof(proto));
    *(void **)(&proto.alloc_d) = dlsym(h, "libdeflate_alloc_decompressor");
    *(void **)(&proto.inflate) = dlsym(h, "libdeflate_deflate_decompress");
    *(void **)(&proto.crc32) = dlsym(h, "libdeflate_crc32");
    *(void **)(&proto.free_d) = dlsym(h, "libdeflate_free_decompressor");
    if (!proto.alloc_d || !proto.inflate || !proto.crc32 || !proto.free_d || threads < 1 || nb == 0) return -2;
    if ((size_t)threads > nb) threads = (int)nb;
    worker_t *ws = (worker_t *)calloc((size_t)threads, sizeof(worker_t));
    if (!ws) return -1;
    const double deadline = now_s() + wall_s;
    for (int i = 0; i < threads; i++) {
        ws[i] = proto;
        ws[i].comp = comp;
        ws[i].offs = offs;
        ws[i].sizes = sizes;
        ws[i].hdr_len = hdr_len;
        ws[i].first_block = nb * (size_t)i / (size_t)threads;
        ws[i].n_blocks = nb * (size_t)(i + 1) / (size_t)threads - ws[i].first_block;
        ws[i].deadline = deadline;
    }
    const int rc = run(ws, threads, inflate_worker, elapsed, bytes);
    *threads_used = threads;
    free(ws);
    return rc;
}
