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

/* ------------------------------------------------------------------------------------------------
 * ParCompress<Bgzf> as gzp runs it, over the libdeflate binary (round 5; BASELINE.md 3, SURVEY 8(d) "CPU baseline"):
 * the orchestration of src/par/compress.rs restated with pthreads --
 *   caller thread   write_all(64 KiB chunks, benches/bench.rs:36-45,121) -> extend_from_slice (one copy per byte,
 *                   :414), and while strictly MORE than buffer_size is buffered (:415) split a block off, send the
 *                   order token to the writer queue FIRST and the work item to the compressor queue second
 *                   (:423-454); finish(): flush_last(true) (:332-362) = the remaining bytes (at least one block,
 *                   maybe empty) with is_last, then both channels are closed and the threads joined (:377-388)
 *   N workers       recv -> Bgzf::encode (src/deflate.rs:613-626) = bgzf::compress (src/bgzf.rs:204-237: zeroed
 *                   buffer of 18 + n + max(128, n / 10) + 8, libdeflate_deflate_compress into [18..], the
 *                   BlockSizeExceeded guard, libdeflate_crc32, header, truncate, CRC32 + ISIZE) + BGZF_EOF when
 *                   is_last -> oneshot (src/par/compress.rs:279-294)
 *   writer thread   recv the order tokens in submission order, wait for each one's result, write_all into an
 *                   in-memory sink (:302-313)
 * with both queues bounded at 2 N (:111-112).  Whole passes over the slab (each a complete ParCompress lifetime:
 * spawn, write, finish, join) until `wall_s` has elapsed; the last pass's stream is left in `sink` so that the
 * caller can compare it with the GPU's.  Bench infrastructure only.
 * ---------------------------------------------------------------------------------------------- */
typedef struct pc_msg {
    uint8_t *base;       /* allocation the block lives in (shared with the caller's buffer; freed by refcount) */
    int *refs;           /* references to `base` (guarded by pc_t.mu_ref) */
    const uint8_t *data; /* the block */
    size_t n;
    int is_last;
    /* oneshot */
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int done;
    uint8_t *out;
    size_t out_len;
    int err;
} pc_msg;

typedef struct {
    pc_msg **q;
    size_t cap, head, count;
    int closed;
    pthread_mutex_t mu;
    pthread_cond_t not_empty, not_full;
} pc_queue;

typedef struct {
    pc_queue work, order;
    pthread_mutex_t mu_ref;
    int level;
    void *(*alloc_c)(int);
    size_t (*deflate)(void *, const void *, size_t, void *, size_t);
    uint32_t (*crc32)(uint32_t, const void *, size_t);
    void (*free_c)(void *);
    uint8_t *sink;
    size_t sink_cap, sink_len;
    int failed;
} pc_t;

static int pcq_init(pc_queue *q, size_t cap) {
    memset(q, 0, sizeof(*q));
    q->q = (pc_msg **)calloc(cap, sizeof(pc_msg *));
    q->cap = cap;
    pthread_mutex_init(&q->mu, NULL);
    pthread_cond_init(&q->not_empty, NULL);
    pthread_cond_init(&q->not_full, NULL);
    return q->q ? 0 : -1;
}
static void pcq_destroy(pc_queue *q) {
    free(q->q);
    pthread_mutex_destroy(&q->mu);
    pthread_cond_destroy(&q->not_empty);
    pthread_cond_destroy(&q->not_full);
}
static void pcq_send(pc_queue *q, pc_msg *m) { /* bounded(2 N): blocks while full */
    pthread_mutex_lock(&q->mu);
    while (q->count == q->cap) pthread_cond_wait(&q->not_full, &q->mu);
    q->q[(q->head + q->count++) % q->cap] = m;
    pthread_cond_signal(&q->not_empty);
    pthread_mutex_unlock(&q->mu);
}
static pc_msg *pcq_recv(pc_queue *q) { /* NULL = closed and drained */
    pthread_mutex_lock(&q->mu);
    while (q->count == 0 && !q->closed) pthread_cond_wait(&q->not_empty, &q->mu);
    pc_msg *m = NULL;
    if (q->count) {
        m = q->q[q->head];
        q->head = (q->head + 1) % q->cap;
        q->count--;
        pthread_cond_signal(&q->not_full);
    }
    pthread_mutex_unlock(&q->mu);
    return m;
}
static void pcq_close(pc_queue *q) {
    pthread_mutex_lock(&q->mu);
    q->closed = 1;
    pthread_cond_broadcast(&q->not_empty);
    pthread_mutex_unlock(&q->mu);
}

static const uint8_t PC_BGZF_EOF[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0,
                                        0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

static void *pc_worker(void *arg) {
    pc_t *pc = (pc_t *)arg;
    void *c = pc->alloc_c(pc->level); /* create_compressor: once per worker (src/par/compress.rs:278) */
    pc_msg *m;
    while ((m = pcq_recv(&pc->work)) != NULL) {
        const size_t n = m->n;
        const size_t cap = 18 + n + (n / 10 > 128 ? n / 10 : 128) + 8; /* src/bgzf.rs:206-212 */
        uint8_t *out = (uint8_t *)calloc(cap + 28, 1);
        size_t clen = 0;
        int err = !out || !c;
        if (!err) {
            clen = pc->deflate(c, m->data, n, out + 18, cap - 18 - 8);
            if (clen == 0 || clen >= 65536) err = 1; /* InsufficientSpace / BlockSizeExceeded */
        }
        if (!err) {
            const uint32_t crc = pc->crc32(0, m->data, n);
            const uint32_t xfl = pc->level >= 9 ? 2u : pc->level <= 1 ? 4u : 0u;
            const uint8_t hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, (uint8_t)xfl, 0xff, 6, 0, 'B', 'C', 2, 0,
                                     (uint8_t)((clen + 25) & 0xff), (uint8_t)((clen + 25) >> 8)};
            memcpy(out, hdr, 18);
            uint8_t *f = out + 18 + clen;
            for (int i = 0; i < 4; i++) f[i] = (uint8_t)(crc >> (8 * i));
            for (int i = 0; i < 4; i++) f[4 + i] = (uint8_t)((uint32_t)n >> (8 * i));
            clen += 26;
            if (m->is_last) {
                memcpy(out + clen, PC_BGZF_EOF, 28);
                clen += 28;
            }
        }
        /* the worker drops its reference to the input block (Bytes: ref-counted) */
        pthread_mutex_lock(&pc->mu_ref);
        const int left = --*m->refs;
        pthread_mutex_unlock(&pc->mu_ref);
        if (left == 0) {
            free(m->base);
            free(m->refs);
        }
        pthread_mutex_lock(&m->mu);
        m->out = out;
        m->out_len = clen;
        m->err = err;
        m->done = 1;
        pthread_cond_signal(&m->cv);
        pthread_mutex_unlock(&m->mu);
    }
    if (c) pc->free_c(c);
    return NULL;
}

static void *pc_writer(void *arg) {
    pc_t *pc = (pc_t *)arg;
    pc_msg *m;
    while ((m = pcq_recv(&pc->order)) != NULL) {
        pthread_mutex_lock(&m->mu);
        while (!m->done) pthread_cond_wait(&m->cv, &m->mu);
        pthread_mutex_unlock(&m->mu);
        if (m->err || pc->sink_len + m->out_len > pc->sink_cap) {
            pc->failed = 1;
        } else {
            memcpy(pc->sink + pc->sink_len, m->out, m->out_len); /* write_all into the in-memory sink */
            pc->sink_len += m->out_len;
        }
        free(m->out);
        pthread_mutex_destroy(&m->mu);
        pthread_cond_destroy(&m->cv);
        free(m);
    }
    return NULL;
}

typedef struct { /* the caller's BytesMut */
    uint8_t *base;
    int *refs;
    size_t off, len, cap;
} pc_buf;

static int pc_buf_fresh(pc_buf *b, size_t cap, const uint8_t *carry, size_t carry_len) {
    b->base = (uint8_t *)malloc(cap);
    b->refs = (int *)malloc(sizeof(int));
    if (!b->base || !b->refs) return -1;
    *b->refs = 1;
    b->off = 0;
    b->cap = cap;
    b->len = carry_len;
    if (carry_len) memcpy(b->base, carry, carry_len);
    return 0;
}

static int pc_send_block(pc_t *pc, pc_buf *b, size_t n, int is_last) {
    pc_msg *m = (pc_msg *)calloc(1, sizeof(pc_msg));
    if (!m) return -1;
    pthread_mutex_init(&m->mu, NULL);
    pthread_cond_init(&m->cv, NULL);
    m->base = b->base;
    m->refs = b->refs;
    m->data = b->base + b->off; /* split_to(n).freeze(): O(1), shares the allocation */
    m->n = n;
    m->is_last = is_last;
    pthread_mutex_lock(&pc->mu_ref);
    ++*b->refs;
    pthread_mutex_unlock(&pc->mu_ref);
    b->off += n;
    b->len -= n;
    pcq_send(&pc->order, m); /* the order token first (src/par/compress.rs:423) ... */
    pcq_send(&pc->work, m);  /* ... then the work item (:439) */
    return 0;
}

static void pc_buf_release(pc_t *pc, pc_buf *b) {
    pthread_mutex_lock(&pc->mu_ref);
    const int left = --*b->refs;
    pthread_mutex_unlock(&pc->mu_ref);
    if (left == 0) {
        free(b->base);
        free(b->refs);
    }
}

/* One ParCompress lifetime over the slab.  Returns 0 on success. */
static int pc_one_pass(pc_t *pc, const uint8_t *slab, size_t slab_len, size_t block, size_t chunk, int threads) {
    pthread_t *tid = (pthread_t *)calloc((size_t)threads + 1, sizeof(pthread_t));
    if (!tid || pcq_init(&pc->work, 2 * (size_t)threads) || pcq_init(&pc->order, 2 * (size_t)threads)) return -1;
    pc->sink_len = 0;
    pc->failed = 0;
    int started = 0;
    for (int i = 0; i < threads; i++)
        if (pthread_create(&tid[i], NULL, pc_worker, pc) == 0) started++;
    const int have_writer = pthread_create(&tid[threads], NULL, pc_writer, pc) == 0;
    int rc = (started == threads && have_writer) ? 0 : -1;
    pc_buf b;
    memset(&b, 0, sizeof(b));
    const size_t alloc = block + chunk + 64;
    if (rc == 0 && pc_buf_fresh(&b, alloc, NULL, 0)) rc = -1;
    for (size_t pos = 0; rc == 0 && pos < slab_len; pos += chunk) { /* write_all(&buf[..64 KiB]) */
        const size_t n = pos + chunk <= slab_len ? chunk : slab_len - pos;
        if (b.off + b.len + n > b.cap) { /* BytesMut::reserve: a new allocation, what is buffered moves over */
            pc_buf nb;
            if (pc_buf_fresh(&nb, alloc, b.base + b.off, b.len)) {
                rc = -1;
                break;
            }
            pc_buf_release(pc, &b);
            b = nb;
        }
        memcpy(b.base + b.off + b.len, slab + pos, n); /* extend_from_slice */
        b.len += n;
        while (rc == 0 && b.len > block) rc = pc_send_block(pc, &b, block, 0); /* strictly more than a block */
    }
    /* finish() -> flush_last(true): what is left, at least one (maybe empty) block, the last one is_last */
    while (rc == 0) {
        const size_t n = b.len < block ? b.len : block;
        rc = pc_send_block(pc, &b, n, b.len == n);
        if (b.len == 0) break;
    }
    if (b.base) pc_buf_release(pc, &b);
    pcq_close(&pc->work);
    pcq_close(&pc->order);
    for (int i = 0; i < started; i++) pthread_join(tid[i], NULL);
    if (have_writer) pthread_join(tid[threads], NULL);
    pcq_destroy(&pc->work);
    pcq_destroy(&pc->order);
    free(tid);
    return rc || pc->failed ? -1 : 0;
}

/* Returns 0 on success, -2 when no libdeflate binary can be loaded.  *bytes = input bytes of the completed
 * passes, *elapsed their wall time, *sink_len = length of the last pass's stream in sink[0 .. sink_cap). */
int gzpx_cpu_bench_parcompress_ref(int level, size_t block, size_t chunk, const uint8_t *slab, size_t slab_len,
                                   int threads, double wall_s, uint8_t *sink, size_t sink_cap, size_t *sink_len,
                                   double *elapsed, uint64_t *bytes, int *passes) {
    void *h = dlopen("libdeflate.so.0", RTLD_NOW);
    if (!h) h = dlopen("/lib/x86_64-linux-gnu/libdeflate.so.0", RTLD_NOW);
    if (!h) return -2;
    pc_t pc;
    memset(&pc, 0, sizeof(pc));
    *(void **)(&pc.alloc_c) = dlsym(h, "libdeflate_alloc_compressor");
    *(void **)(&pc.deflate) = dlsym(h, "libdeflate_deflate_compress");
    *(void **)(&pc.crc32) = dlsym(h, "libdeflate_crc32");
    *(void **)(&pc.free_c) = dlsym(h, "libdeflate_free_compressor");
    if (!pc.alloc_c || !pc.deflate || !pc.crc32 || !pc.free_c) return -2;
    if (threads < 1 || block == 0 || chunk == 0 || !sink) return -1;
    pc.level = level;
    pc.sink = sink;
    pc.sink_cap = sink_cap;
    pthread_mutex_init(&pc.mu_ref, NULL);
    const double t0 = now_s();
    *bytes = 0;
    *passes = 0;
    int rc = 0;
    do {
        rc = pc_one_pass(&pc, slab, slab_len, block, chunk, threads);
        if (rc) break;
        *bytes += slab_len;
        ++*passes;
    } while (now_s() - t0 < wall_s);
    *elapsed = now_s() - t0;
    *sink_len = pc.sink_len;
    pthread_mutex_destroy(&pc.mu_ref);
    return rc;
}
