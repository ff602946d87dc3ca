// gzpx_api.cpp -- the C ABI of include/gzpx.h over the HIP pipeline of gzpx_kernels.hip.
//
// Host-side mirror of what a gzp worker thread does per block (src/par/compress.rs:279-294:
// create_compressor once, encode per block) lifted to whole slabs: one call = every block of the
// slab through the kernel pipeline, blocks cut and ordered exactly like ParCompress::write /
// flush_last (src/par/compress.rs:413-463, 332-362).  No CPU fallback exists on purpose: without
// a HIP device every entry point fails with GZPX_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <mutex>
#include <new>
#include <string.h>
#include <vector>

#include "../../include/gzpx.h"
#include "gzpx_device.h"

using namespace gzpx;

namespace {

constexpr size_t kDictSize = 32768;  // DICT_SIZE, src/lib.rs:108
constexpr uint32_t kMaxBatchBlocks = 16384;
constexpr size_t kMaxScratchBytes = (size_t)24 << 30;  // per-context cap of the per-block scratch

// ---- GF(2) polynomial helpers for CRC-32 combination (reflected representation) ----
uint32_t multmodp(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}

// x^(2^k) mod P
uint32_t x2k(unsigned k) {
    uint32_t p = 1u << 30;  // x^1
    for (unsigned i = 0; i < k; i++) p = multmodp(p, p);
    return p;
}

// x^(8 * len) mod P
uint32_t x8n(uint64_t len) {
    uint32_t r = 1u << 31;  // x^0
    unsigned k = 3;
    while (len) {
        if (len & 1) r = multmodp(x2k(k), r);
        len >>= 1;
        k++;
    }
    return r;
}

uint32_t crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2) {
    return multmodp(x8n(len2), crc1) ^ crc2;
}

struct StageEvents {
    hipEvent_t ev[GZPX_N_STAGES + 1];
    bool created = false;
};

}  // namespace

struct gzpx_ctx {
    gzpx_config cfg;
    Config dcfg;
    CrcConsts crc_consts;
    hipStream_t stream = nullptr;
    uint32_t batch_blocks = 0;
    Scratch scratch = {};
    uint8_t *d_in = nullptr;
    size_t d_in_cap = 0;
    uint8_t *d_out = nullptr;
    size_t d_out_cap = 0;
    BlockMeta *h_meta = nullptr;  // pinned, batch_blocks entries
    SubMeta *h_sub = nullptr;     // pinned, batch_blocks * max_sub entries (debug hooks)
    uint64_t *h_total = nullptr;  // pinned
    StageEvents events;
    bool profiling = false;
    float stage_ms[GZPX_N_STAGES] = {0};
    // last call bookkeeping for the debug hooks
    uint32_t last_nb = 0;
    char devname[256] = {0};
    std::mutex mu;
};

namespace {

#define HIP_TRY(expr)                          \
    do {                                       \
        hipError_t _e = (expr);                \
        if (_e != hipSuccess) return GZPX_ERR_DEVICE; \
    } while (0)

size_t extra_amount(size_t n) {  // src/bgzf.rs:45,50-52
    size_t e = (size_t)((double)n * 0.1);
    return e < 128 ? 128 : e;
}

size_t framed_bound_per_block(const gzpx_ctx *ctx) {
    const size_t hdr = ctx->cfg.format == GZPX_FORMAT_BGZF ? 18 : 20;
    return hdr + ctx->cfg.buffer_size + extra_amount(ctx->cfg.buffer_size) + 8;
}

uint64_t blocks_of(const gzpx_ctx *ctx, size_t in_len) {
    const size_t bs = ctx->cfg.buffer_size;
    return in_len == 0 ? 1 : (in_len + bs - 1) / bs;
}

// bytes of device scratch one block needs (see gzpx_device.h Scratch)
size_t scratch_bytes_per_block(const Config &c) {
    return (size_t)c.stride * (2 + 1 + 2 + 4 + (c.level >= 2 ? 2 : 0)) + c.stride / 8 +
           (size_t)c.max_sub * (sizeof(SubMeta) + (kHistStride + kCodeWords + kHdrWords) * 4) +
           sizeof(BlockMeta) + 8;
}

int alloc_scratch(gzpx_ctx *ctx) {
    const size_t nb = ctx->batch_blocks;
    const Config &c = ctx->dcfg;
    Scratch &s = ctx->scratch;
    HIP_TRY(hipMalloc((void **)&s.meta, nb * sizeof(BlockMeta)));
    HIP_TRY(hipMalloc((void **)&s.sub, nb * (size_t)c.max_sub * sizeof(SubMeta)));
    HIP_TRY(hipMalloc((void **)&s.cand, nb * (size_t)c.stride * sizeof(uint16_t)));
    HIP_TRY(hipMalloc((void **)&s.len8, nb * (size_t)c.stride));
    HIP_TRY(hipMalloc((void **)&s.which, nb * (size_t)(c.stride / 32) * 4));
    HIP_TRY(hipMalloc((void **)&s.alt, nb * (size_t)c.stride * sizeof(uint16_t)));
    HIP_TRY(hipMalloc((void **)&s.tok, nb * (size_t)c.stride * 4));
    if (c.level >= 2) {  // hc_matchfinder levels: hash4 chain links + per-block parse state
        HIP_TRY(hipMalloc((void **)&s.d4, nb * (size_t)c.stride * sizeof(uint16_t)));
        HIP_TRY(hipMalloc((void **)&s.hc, nb * sizeof(HcState)));
        HIP_TRY(hipMalloc((void **)&s.pending, 64));
    }
    HIP_TRY(hipMalloc((void **)&s.hist, nb * (size_t)c.max_sub * kHistStride * 4));
    HIP_TRY(hipMalloc((void **)&s.codes, nb * (size_t)c.max_sub * kCodeWords * 4));
    HIP_TRY(hipMalloc((void **)&s.hdr, nb * (size_t)c.max_sub * kHdrWords * 4));
    HIP_TRY(hipMalloc((void **)&s.out_off, (nb + 1) * sizeof(uint64_t)));
    HIP_TRY(hipHostMalloc((void **)&ctx->h_meta, nb * sizeof(BlockMeta), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&ctx->h_sub, nb * (size_t)c.max_sub * sizeof(SubMeta), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&ctx->h_total, 64, hipHostMallocDefault));
    return GZPX_OK;
}

void free_scratch(gzpx_ctx *ctx) {
    Scratch &s = ctx->scratch;
    if (s.meta) (void)hipFree(s.meta);
    if (s.sub) (void)hipFree(s.sub);
    if (ctx->h_sub) (void)hipHostFree(ctx->h_sub);
    if (s.cand) (void)hipFree(s.cand);
    if (s.tok) (void)hipFree(s.tok);
    if (s.len8) (void)hipFree(s.len8);
    if (s.which) (void)hipFree(s.which);
    if (s.alt) (void)hipFree(s.alt);
    if (s.d4) (void)hipFree(s.d4);
    if (s.hc) (void)hipFree(s.hc);
    if (s.pending) (void)hipFree(s.pending);
    if (s.hist) (void)hipFree(s.hist);
    if (s.codes) (void)hipFree(s.codes);
    if (s.hdr) (void)hipFree(s.hdr);
    if (s.out_off) (void)hipFree(s.out_off);
    if (ctx->h_meta) (void)hipHostFree(ctx->h_meta);
    if (ctx->h_total) (void)hipHostFree(ctx->h_total);
    if (ctx->d_in) (void)hipFree(ctx->d_in);
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    s = Scratch{};
}

// One batch of blocks through the pipeline.  d_in/d_out are device pointers for this batch.
int run_batch(gzpx_ctx *ctx, const uint8_t *d_in, size_t in_len, uint32_t nb, int is_last,
              uint8_t *d_out, size_t out_cap, hipStream_t stream, size_t *produced,
              uint32_t *block_sizes, size_t *fail_block) {
    const Config &c = ctx->dcfg;
    const Scratch &s = ctx->scratch;
    const bool prof = ctx->profiling;
    hipEvent_t *ev = ctx->events.ev;
    int k = 0;
    if (prof) HIP_TRY(hipEventRecord(ev[k], stream));
    launch_init_meta(c, in_len, nb, is_last, s, stream);
    if (prof) HIP_TRY(hipEventRecord(ev[++k], stream));
    launch_candidates(c, d_in, in_len, nb, s, stream);
    if (prof) HIP_TRY(hipEventRecord(ev[++k], stream));
    if (c.level <= 1) {  // (at level 0 every block is a passthrough block: both return at once)
        launch_match(c, d_in, in_len, nb, s, stream);
        if (prof) HIP_TRY(hipEventRecord(ev[++k], stream));
        launch_parse(c, d_in, in_len, nb, s, stream);
        if (prof) HIP_TRY(hipEventRecord(ev[++k], stream));
    } else {
        // levels 2-4: match + parse rounds until no block needs its tail redone with another
        // min_len (one round unless should_end_block splits a block into unlike halves)
        for (uint32_t round = 0;; round++) {
            launch_hc_round(c, d_in, nb, s, round == 0, stream);
            HIP_TRY(hipMemcpyAsync(ctx->h_total, s.pending, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (*(const uint32_t *)ctx->h_total == 0) break;
            if (round > c.max_sub + 2) return GZPX_ERR_DEVICE;  // cannot happen: one sub-block per round
        }
        if (prof) HIP_TRY(hipEventRecord(ev[++k], stream));
        if (prof) HIP_TRY(hipEventRecord(ev[++k], stream));
    }
    launch_hist(c, nb, s, stream);
    if (prof) HIP_TRY(hipEventRecord(ev[++k], stream));
    launch_huffman(c, nb, s, stream);
    if (prof) HIP_TRY(hipEventRecord(ev[++k], stream));
    launch_crc32(c, d_in, in_len, nb, s, ctx->crc_consts, stream);
    if (prof) HIP_TRY(hipEventRecord(ev[++k], stream));
    launch_scan(nb, s, stream);
    if (prof) HIP_TRY(hipEventRecord(ev[++k], stream));
    launch_emit(c, d_in, in_len, nb, s, d_out, out_cap, stream);
    if (prof) HIP_TRY(hipEventRecord(ev[++k], stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ctx->h_total, s.out_off + nb, sizeof(uint64_t), hipMemcpyDeviceToHost,
                           stream));
    HIP_TRY(hipMemcpyAsync(ctx->h_meta, s.meta, nb * sizeof(BlockMeta), hipMemcpyDeviceToHost,
                           stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (prof) {
        for (int i = 0; i < GZPX_N_STAGES; i++) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            ctx->stage_ms[i] += ms;
        }
    }
    ctx->last_nb = nb;
    for (uint32_t b = 0; b < nb; b++) {
        if (ctx->h_meta[b].status == kStatusBlockSizeExceeded) {
            *fail_block = b;
            return GZPX_ERR_BLOCK_SIZE_EXCEEDED;
        }
        if (block_sizes) block_sizes[b] = ctx->h_meta[b].framed_bytes;
    }
    *produced = (size_t)*ctx->h_total;
    if (*produced > out_cap) return GZPX_ERR_INSUFFICIENT_SPACE;
    return GZPX_OK;
}

int compress_device_locked(gzpx_ctx *ctx, const uint8_t *d_in, size_t in_len, int mode,
                           uint8_t *d_out, size_t out_cap, size_t *out_len, uint32_t *block_sizes,
                           size_t max_blocks, size_t *n_blocks, hipStream_t stream) {
    const size_t bs = ctx->cfg.buffer_size;
    if (mode != GZPX_SLAB_FULL_BLOCKS && mode != GZPX_SLAB_LAST && mode != GZPX_SLAB_FLUSH)
        return GZPX_ERR_INVALID_ARG;
    if (mode == GZPX_SLAB_FULL_BLOCKS && (in_len == 0 || in_len % bs != 0)) return GZPX_ERR_INVALID_ARG;
    const int is_last = mode == GZPX_SLAB_LAST;
    if ((in_len && !d_in) || !d_out || !out_len) return GZPX_ERR_INVALID_ARG;
    const uint64_t total_nb = blocks_of(ctx, in_len);
    if (block_sizes && max_blocks < total_nb) return GZPX_ERR_INVALID_ARG;
    if (!stream) stream = ctx->stream;
    memset(ctx->stage_ms, 0, sizeof(ctx->stage_ms));
    size_t produced_total = 0;
    for (uint64_t b0 = 0; b0 < total_nb; b0 += ctx->batch_blocks) {
        const uint32_t nb = (uint32_t)((total_nb - b0 < ctx->batch_blocks) ? total_nb - b0
                                                                          : ctx->batch_blocks);
        const size_t in_begin = (size_t)b0 * bs;
        size_t in_batch = in_len > in_begin ? in_len - in_begin : 0;
        if (in_batch > (size_t)nb * bs) in_batch = (size_t)nb * bs;
        const int last_batch = (b0 + nb == total_nb) ? is_last : 0;
        size_t produced = 0, fail = 0;
        int rc = run_batch(ctx, d_in + in_begin, in_batch, nb, last_batch, d_out + produced_total,
                           out_cap - produced_total, stream, &produced,
                           block_sizes ? block_sizes + b0 : nullptr, &fail);
        if (rc != GZPX_OK) {
            if (n_blocks) *n_blocks = (size_t)(b0 + fail);
            return rc;
        }
        produced_total += produced;
    }
    *out_len = produced_total;
    if (n_blocks) *n_blocks = (size_t)total_nb;
    return GZPX_OK;
}

int ensure_buffers(gzpx_ctx *ctx, size_t in_len, size_t out_need) {
    if (in_len + 16 > ctx->d_in_cap) {
        if (ctx->d_in) (void)hipFree(ctx->d_in);
        ctx->d_in = nullptr;
        ctx->d_in_cap = 0;
        const size_t cap = in_len + in_len / 8 + 4096;
        HIP_TRY(hipMalloc((void **)&ctx->d_in, cap));
        ctx->d_in_cap = cap;
    }
    if (out_need > ctx->d_out_cap) {
        if (ctx->d_out) (void)hipFree(ctx->d_out);
        ctx->d_out = nullptr;
        ctx->d_out_cap = 0;
        const size_t cap = out_need + out_need / 8 + 4096;
        HIP_TRY(hipMalloc((void **)&ctx->d_out, cap));
        ctx->d_out_cap = cap;
    }
    return GZPX_OK;
}

}  // namespace

extern "C" {

void gzpx_config_default(gzpx_config *cfg, int format) {
    if (!cfg) return;
    cfg->device = 0;
    cfg->format = format;
    cfg->level = 3;  // ParCompressBuilder::new: Compression::new(3), src/par/compress.rs:54-62
    cfg->compat = GZPX_COMPAT_LIBDEFLATE_1_24;
    // Bgzf::DEFAULT_BUFSIZE = 65280 (src/deflate.rs:583); Mgzip keeps the trait default
    // DEFAULT_BUFSIZE = BUFSIZE = 128 KiB (src/lib.rs:330, :105)
    cfg->buffer_size = format == GZPX_FORMAT_BGZF ? 65280 : 131072;
    cfg->max_slab_bytes = (size_t)1 << 30;
}

int gzpx_ctx_create(const gzpx_config *cfg, gzpx_ctx **out) {
    if (!cfg || !out) return GZPX_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->format != GZPX_FORMAT_BGZF && cfg->format != GZPX_FORMAT_MGZIP) return GZPX_ERR_INVALID_ARG;
    if (cfg->buffer_size < kDictSize) return GZPX_ERR_BUFFER_SIZE;  // src/par/compress.rs:68-74
    if (cfg->level < 0 || cfg->level > 12) return GZPX_ERR_COMPRESSION_LEVEL;
    if (cfg->compat != GZPX_COMPAT_LIBDEFLATE_1_24 && cfg->compat != GZPX_COMPAT_LIBDEFLATE_1_10)
        return GZPX_ERR_INVALID_ARG;
    if (cfg->level > 4) return GZPX_ERR_UNSUPPORTED;  // 5..12 (lazy / near-optimal parsers): not built yet
    if (cfg->buffer_size > kMaxBlockSize) return GZPX_ERR_UNSUPPORTED;  // > 16 MiB blocks: not built
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GZPX_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return GZPX_ERR_INVALID_ARG;
    if (hipSetDevice(cfg->device) != hipSuccess) return GZPX_ERR_DEVICE;

    gzpx_ctx *ctx = new (std::nothrow) gzpx_ctx();
    if (!ctx) return GZPX_ERR_DEVICE;
    ctx->cfg = *cfg;
    ctx->dcfg.format = (uint32_t)cfg->format;
    ctx->dcfg.level = (uint32_t)cfg->level;
    ctx->dcfg.compat = (uint32_t)cfg->compat;
    ctx->dcfg.block_size = (uint32_t)cfg->buffer_size;
    ctx->dcfg.xfl = cfg->level >= 9 ? 2u : cfg->level <= 1 ? 4u : 0u;  // src/bgzf.rs:278-284
    ctx->dcfg.debug = 0;
    // per-block strides: padding for k_candidates' last iteration / dword-wide tile loads
    ctx->dcfg.stride = (uint32_t)((cfg->buffer_size + 2047) / 2048 * 2048 + 2048);
    // level 1 sub-blocks hold 8192 matches (>= 32768 bytes); the block splitter of levels 2-4 may
    // cut every MIN_BLOCK_LENGTH = 5000 bytes
    ctx->dcfg.max_sub = (uint32_t)(cfg->buffer_size / (cfg->level == 1 ? 32768 : 5000) + 2);
    // deflate_compress: inputs up to 55 - 4*level bytes -- and everything at level 0 -- take
    // deflate_compress_none
    ctx->dcfg.passthrough = cfg->level == 0 ? 0xFFFFFFFFu : (uint32_t)(55 - 4 * cfg->level);
    {
        static const uint32_t depth[5] = {0, 0, 6, 12, 16}, nice[5] = {0, 0, 10, 14, 30};
        ctx->dcfg.hc_depth = depth[cfg->level];
        ctx->dcfg.hc_nice = nice[cfg->level];
    }
    for (unsigned l = 0; l < 10; l++) ctx->crc_consts.pow64[l] = x2k(9 + l);
    ctx->crc_consts.pow_tile = x2k(19);  // x^(8 * 65536) = x^(2^19)
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) {
        snprintf(ctx->devname, sizeof(ctx->devname), "%s (%s, %d CUs)", prop.name, prop.gcnArchName,
                 prop.multiProcessorCount);
    }
    const uint64_t want = blocks_of(ctx, cfg->max_slab_bytes ? cfg->max_slab_bytes : 1);
    ctx->batch_blocks = (uint32_t)(want < kMaxBatchBlocks ? want : kMaxBatchBlocks);
    {
        const size_t fit = kMaxScratchBytes / scratch_bytes_per_block(ctx->dcfg);
        if (ctx->batch_blocks > fit) ctx->batch_blocks = (uint32_t)fit;
    }
    if (ctx->batch_blocks == 0) ctx->batch_blocks = 1;
    int rc = GZPX_OK;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) rc = GZPX_ERR_DEVICE;
    if (rc == GZPX_OK) rc = alloc_scratch(ctx);
    if (rc == GZPX_OK) {
        for (int i = 0; i <= GZPX_N_STAGES; i++)
            if (hipEventCreate(&ctx->events.ev[i]) != hipSuccess) rc = GZPX_ERR_DEVICE;
        ctx->events.created = (rc == GZPX_OK);
    }
    if (rc != GZPX_OK) {
        gzpx_ctx_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return GZPX_OK;
}

void gzpx_ctx_destroy(gzpx_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_scratch(ctx);
    if (ctx->events.created)
        for (int i = 0; i <= GZPX_N_STAGES; i++) (void)hipEventDestroy(ctx->events.ev[i]);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

size_t gzpx_slab_bound(const gzpx_ctx *ctx, size_t in_len) {
    if (!ctx) return 0;
    return (size_t)blocks_of(ctx, in_len) * framed_bound_per_block(ctx) + 28 + 64;
}

int gzpx_compress_slab_device(gzpx_ctx *ctx, const void *d_in, size_t in_len, int mode,
                              void *d_out, size_t out_cap, size_t *out_len, uint32_t *block_sizes,
                              size_t max_blocks, size_t *n_blocks, void *hip_stream) {
    if (!ctx) return GZPX_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (hipSetDevice(ctx->cfg.device) != hipSuccess) return GZPX_ERR_DEVICE;
    return compress_device_locked(ctx, (const uint8_t *)d_in, in_len, mode, (uint8_t *)d_out,
                                  out_cap, out_len, block_sizes, max_blocks, n_blocks,
                                  (hipStream_t)hip_stream);
}

int gzpx_compress_slab(gzpx_ctx *ctx, const uint8_t *in, size_t in_len, int mode, uint8_t *out,
                       size_t out_cap, size_t *out_len, uint32_t *block_sizes, size_t max_blocks,
                       size_t *n_blocks) {
    if (!ctx || (in_len && !in) || !out || !out_len) return GZPX_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (hipSetDevice(ctx->cfg.device) != hipSuccess) return GZPX_ERR_DEVICE;
    const size_t need = gzpx_slab_bound(ctx, in_len);
    int rc = ensure_buffers(ctx, in_len, need);
    if (rc != GZPX_OK) return rc;
    if (in_len) HIP_TRY(hipMemcpyAsync(ctx->d_in, in, in_len, hipMemcpyHostToDevice, ctx->stream));
    size_t produced = 0;
    rc = compress_device_locked(ctx, ctx->d_in, in_len, mode, ctx->d_out, ctx->d_out_cap, &produced,
                                block_sizes, max_blocks, n_blocks, ctx->stream);
    if (rc != GZPX_OK) return rc;
    if (produced > out_cap) return GZPX_ERR_INSUFFICIENT_SPACE;
    HIP_TRY(hipMemcpyAsync(out, ctx->d_out, produced, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out_len = produced;
    return GZPX_OK;
}

int gzpx_encode_block(gzpx_ctx *ctx, const uint8_t *in, size_t n, int is_last, uint8_t *out,
                      size_t out_cap, size_t *out_len) {
    if (!ctx) return GZPX_ERR_INVALID_ARG;
    if (n > ctx->cfg.buffer_size) return GZPX_ERR_INVALID_ARG;
    // Bgzf::encode runs on one block; the EOF marker is the only is_last effect
    // (src/deflate.rs:613-626).  A non-last block may be short (flush()).
    std::vector<uint8_t> tmp(gzpx_slab_bound(ctx, n));
    size_t got = 0, nb = 0;
    int rc = gzpx_compress_slab(ctx, in, n, is_last ? GZPX_SLAB_LAST : GZPX_SLAB_FLUSH, tmp.data(),
                                tmp.size(), &got, nullptr, 0, &nb);
    if (rc != GZPX_OK) return rc;
    if (got > out_cap) return GZPX_ERR_INSUFFICIENT_SPACE;
    memcpy(out, tmp.data(), got);
    *out_len = got;
    return GZPX_OK;
}

// ---------------------------------------------------------------- libdeflate-shaped ABI
struct gzpx_compressor {
    int level;
    int compat;
    gzpx_ctx *ctx;
    std::vector<uint8_t> tmp;
};

gzpx_compressor *gzpx_alloc_compressor(int level) {
    if (level < 0 || level > 12) return nullptr;
    gzpx_compressor *c = new (std::nothrow) gzpx_compressor();
    if (!c) return nullptr;
    c->level = level;
    c->compat = GZPX_COMPAT_LIBDEFLATE_1_24;
    c->ctx = nullptr;
    return c;
}

int gzpx_compressor_set_compat(gzpx_compressor *c, int compat) {
    if (!c || c->ctx) return GZPX_ERR_INVALID_ARG;
    c->compat = compat;
    return GZPX_OK;
}

static int compressor_ctx(gzpx_compressor *c, size_t n) {
    if (c->ctx && n <= c->ctx->cfg.buffer_size) return GZPX_OK;
    if (c->ctx) gzpx_ctx_destroy(c->ctx);
    c->ctx = nullptr;
    gzpx_config cfg;
    gzpx_config_default(&cfg, GZPX_FORMAT_MGZIP);  // Mgzip framing has no payload size limit
    cfg.level = c->level;
    cfg.compat = c->compat;
    size_t bs = 65536;  // one whole-buffer DEFLATE call = one "block" of that size
    while (bs < n) bs *= 2;
    cfg.buffer_size = bs;
    cfg.max_slab_bytes = bs;
    return gzpx_ctx_create(&cfg, &c->ctx);
}

size_t gzpx_deflate_compress(gzpx_compressor *c, const void *in, size_t n, void *out, size_t cap) {
    if (!c || (!in && n) || !out) return 0;
    if (n > kMaxBlockSize) return 0;  // whole-buffer inputs above 16 MiB: not built
    if (compressor_ctx(c, n) != GZPX_OK) return 0;
    c->tmp.resize(gzpx_slab_bound(c->ctx, n));
    size_t got = 0, nb = 0;
    if (gzpx_compress_slab(c->ctx, (const uint8_t *)in, n, GZPX_SLAB_LAST, c->tmp.data(), c->tmp.size(),
                           &got, nullptr, 0, &nb) != GZPX_OK)
        return 0;
    const size_t payload = got - 20 - 8;
    if (payload > cap) return 0;  // libdeflate: 0 when the output does not fit
    memcpy(out, c->tmp.data() + 20, payload);
    return payload;
}

size_t gzpx_deflate_compress_bound(gzpx_compressor *, size_t n) {
    // libdeflate_deflate_compress_bound: stored blocks of >= 10000 bytes, 5 bytes each, + slack
    size_t max_blocks = (n + 9999) / 10000;
    if (max_blocks < 1) max_blocks = 1;
    return 5 * max_blocks + n + 1 + 8;
}

void gzpx_free_compressor(gzpx_compressor *c) {
    if (!c) return;
    if (c->ctx) gzpx_ctx_destroy(c->ctx);
    delete c;
}

uint32_t gzpx_crc32(uint32_t crc, const void *buf, size_t n) {
    // libdeflate_crc32 semantics: crc32(crc, buf) = combine(crc, crc32(0, buf), n)
    static std::mutex mu;
    static gzpx_ctx *ctx = nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!ctx) {
        gzpx_config cfg;
        gzpx_config_default(&cfg, GZPX_FORMAT_MGZIP);
        cfg.level = 1;
        cfg.buffer_size = kTile;
        cfg.max_slab_bytes = (size_t)64 << 20;
        if (gzpx_ctx_create(&cfg, &ctx) != GZPX_OK) return 0;
    }
    if (n == 0) return crc;
    std::lock_guard<std::mutex> lock2(ctx->mu);
    if (hipSetDevice(ctx->cfg.device) != hipSuccess) return 0;
    const uint8_t *p = (const uint8_t *)buf;
    const size_t slab_max = (size_t)ctx->batch_blocks * kTile;
    while (n) {
        const size_t take = n < slab_max ? n : slab_max;
        if (ensure_buffers(ctx, take, 64) != GZPX_OK) return 0;
        const uint32_t nb = (uint32_t)((take + kTile - 1) / kTile);
        if (hipMemcpyAsync(ctx->d_in, p, take, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return 0;
        launch_init_meta(ctx->dcfg, take, nb, 0, ctx->scratch, ctx->stream);
        launch_crc32(ctx->dcfg, ctx->d_in, take, nb, ctx->scratch, ctx->crc_consts, ctx->stream);
        if (hipMemcpyAsync(ctx->h_meta, ctx->scratch.meta, nb * sizeof(BlockMeta), hipMemcpyDeviceToHost,
                           ctx->stream) != hipSuccess)
            return 0;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return 0;
        for (uint32_t b = 0; b < nb; b++)
            crc = crc32_combine(crc, ctx->h_meta[b].crc, ctx->h_meta[b].n);
        p += take;
        n -= take;
    }
    return crc;
}

// ---------------------------------------------------------------- ParDecompress side
struct gzpx_dctx {
    int device = 0;
    int format = 0;
    CrcConsts cc;
    hipStream_t stream = nullptr;
    size_t cap_blocks = 0;
    uint64_t *d_offsets = nullptr, *d_out_off = nullptr;
    uint32_t *d_sizes = nullptr, *d_crc = nullptr;
    DBlockHost *d_blk = nullptr;
    DBlockHost *h_blk = nullptr;
    uint32_t *h_crc = nullptr;
    uint64_t *h_total = nullptr;
    uint8_t *d_in = nullptr, *d_out = nullptr;
    size_t d_in_cap = 0, d_out_cap = 0;
    bool debug = false;
    size_t last_nb = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};  // around k_inflate of the last launch
    std::mutex mu;
};

namespace {

void dctx_free_tables(gzpx_dctx *c) {
    if (c->d_offsets) (void)hipFree(c->d_offsets);
    if (c->d_out_off) (void)hipFree(c->d_out_off);
    if (c->d_sizes) (void)hipFree(c->d_sizes);
    if (c->d_crc) (void)hipFree(c->d_crc);
    if (c->d_blk) (void)hipFree(c->d_blk);
    if (c->h_blk) (void)hipHostFree(c->h_blk);
    if (c->h_crc) (void)hipHostFree(c->h_crc);
    c->d_offsets = c->d_out_off = nullptr;
    c->d_sizes = c->d_crc = nullptr;
    c->d_blk = c->h_blk = nullptr;
    c->h_crc = nullptr;
    c->cap_blocks = 0;
}

int dctx_reserve(gzpx_dctx *c, size_t nb) {
    if (nb <= c->cap_blocks) return GZPX_OK;
    dctx_free_tables(c);
    const size_t cap = nb + nb / 4 + 64;
    HIP_TRY(hipMalloc((void **)&c->d_offsets, cap * 8));
    HIP_TRY(hipMalloc((void **)&c->d_out_off, (cap + 1) * 8));
    HIP_TRY(hipMalloc((void **)&c->d_sizes, cap * 4));
    HIP_TRY(hipMalloc((void **)&c->d_crc, cap * 4));
    HIP_TRY(hipMalloc((void **)&c->d_blk, cap * sizeof(DBlockHost)));
    HIP_TRY(hipHostMalloc((void **)&c->h_blk, cap * sizeof(DBlockHost), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&c->h_crc, cap * 4, hipHostMallocDefault));
    c->cap_blocks = cap;
    return GZPX_OK;
}

int decompress_device_locked(gzpx_dctx *c, const uint8_t *d_in, size_t in_len, const uint64_t *offsets,
                             const uint32_t *sizes, size_t nb, uint8_t *d_out, size_t out_cap,
                             size_t *out_len, gzpx_check_info *info, hipStream_t stream) {
    if (!out_len || (nb && (!offsets || !sizes || !d_in))) return GZPX_ERR_INVALID_ARG;
    *out_len = 0;
    if (nb == 0) return GZPX_OK;
    const uint32_t hdr_len = c->format == GZPX_FORMAT_BGZF ? 18 : 20;
    for (size_t b = 0; b < nb; b++)
        if (sizes[b] < hdr_len + 8 || offsets[b] + sizes[b] > in_len) return GZPX_ERR_INVALID_ARG;
    int rc = dctx_reserve(c, nb);
    if (rc != GZPX_OK) return rc;
    if (!stream) stream = c->stream;
    HIP_TRY(hipMemcpyAsync(c->d_offsets, offsets, nb * 8, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(c->d_sizes, sizes, nb * 4, hipMemcpyHostToDevice, stream));
    launch_inflate(hdr_len, d_in, c->d_offsets, c->d_sizes, (uint32_t)nb, c->d_blk, c->d_out_off, d_out, out_cap,
                   c->d_crc, c->cc, c->debug, c->ev[0], c->ev[1], stream);
    c->last_nb = nb;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_blk, c->d_blk, nb * sizeof(DBlockHost), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(c->h_crc, c->d_crc, nb * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(c->h_total, c->d_out_off + nb, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (size_t b = 0; b < nb; b++) {  // first failing block, in stream order (src/par/decompress.rs:162-186)
        const DBlockHost &d = c->h_blk[b];
        int err = GZPX_OK;
        if (d.status == 1) err = GZPX_ERR_BAD_DATA;
        else if (d.status == 2) err = GZPX_ERR_INSUFFICIENT_SPACE;
        else if (c->h_crc[b] != d.crc) err = GZPX_ERR_INVALID_CHECK;
        if (err != GZPX_OK) {
            if (info) {
                info->block = b;
                info->found = c->h_crc[b];
                info->expected = d.crc;
            }
            return err;
        }
    }
    *out_len = (size_t)*c->h_total;
    return GZPX_OK;
}

}  // namespace

extern "C" {

int gzpx_dctx_create(int device, int format, gzpx_dctx **out) {
    if (!out || (format != GZPX_FORMAT_BGZF && format != GZPX_FORMAT_MGZIP)) return GZPX_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GZPX_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return GZPX_ERR_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) return GZPX_ERR_DEVICE;
    gzpx_dctx *c = new (std::nothrow) gzpx_dctx();
    if (!c) return GZPX_ERR_DEVICE;
    c->device = device;
    c->format = format;
    for (unsigned l = 0; l < 10; l++) c->cc.pow64[l] = x2k(9 + l);
    c->cc.pow_tile = x2k(19);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev[0]) != hipSuccess || hipEventCreate(&c->ev[1]) != hipSuccess ||
        hipHostMalloc((void **)&c->h_total, 64, hipHostMallocDefault) != hipSuccess) {
        gzpx_dctx_destroy(c);
        return GZPX_ERR_DEVICE;
    }
    *out = c;
    return GZPX_OK;
}

void gzpx_dctx_destroy(gzpx_dctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    dctx_free_tables(c);
    if (c->h_total) (void)hipHostFree(c->h_total);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->ev[0]) (void)hipEventDestroy(c->ev[0]);
    if (c->ev[1]) (void)hipEventDestroy(c->ev[1]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int gzpx_scan_blocks(int format, const uint8_t *in, size_t in_len, uint64_t *offsets, uint32_t *sizes,
                     size_t max_blocks, size_t *n_blocks, size_t *consumed) {
    if ((!in && in_len) || !n_blocks || !consumed) return GZPX_ERR_INVALID_ARG;
    const size_t hdr = format == GZPX_FORMAT_BGZF ? 18 : 20;  // BlockFormatSpec::HEADER_SIZE
    size_t pos = 0, nb = 0;
    *n_blocks = 0;
    *consumed = 0;
    while (in_len - pos >= hdr) {  // read_exact(header) succeeds
        const uint8_t *h = in + pos;
        if ((h[3] & 4) != 4) return GZPX_ERR_INVALID_HEADER;  // "Extra field flag not set"
        if (format == GZPX_FORMAT_BGZF ? (h[12] != 'B' || h[13] != 'C') : (h[12] != 'I' || h[13] != 'G'))
            return GZPX_ERR_INVALID_HEADER;  // "Bad SID"
        const size_t size = format == GZPX_FORMAT_BGZF
                                ? (size_t)(h[16] | (h[17] << 8)) + 1
                                : (size_t)h[16] | ((size_t)h[17] << 8) | ((size_t)h[18] << 16) | ((size_t)h[19] << 24);
        if (size < hdr + 8) return GZPX_ERR_INVALID_HEADER;
        if (in_len - pos < size) break;  // read_exact(remainder) would wait for more input
        if (offsets && sizes) {
            if (nb >= max_blocks) break;
            offsets[nb] = pos;
            sizes[nb] = (uint32_t)size;
        }
        nb++;
        pos += size;
    }
    *n_blocks = nb;
    *consumed = pos;
    return GZPX_OK;
}

int gzpx_decompress_blocks_device(gzpx_dctx *c, const void *d_in, size_t in_len, const uint64_t *offsets,
                                  const uint32_t *sizes, size_t n_blocks, void *d_out, size_t out_cap,
                                  size_t *out_len, gzpx_check_info *info, void *hip_stream) {
    if (!c) return GZPX_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) return GZPX_ERR_DEVICE;
    return decompress_device_locked(c, (const uint8_t *)d_in, in_len, offsets, sizes, n_blocks, (uint8_t *)d_out,
                                    out_cap, out_len, info, (hipStream_t)hip_stream);
}

int gzpx_decompress_blocks(gzpx_dctx *c, const uint8_t *in, size_t in_len, const uint64_t *offsets,
                           const uint32_t *sizes, size_t n_blocks, uint8_t *out, size_t out_cap,
                           size_t *out_len, gzpx_check_info *info) {
    if (!c || (!in && in_len) || (!out && out_cap) || !out_len) return GZPX_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) return GZPX_ERR_DEVICE;
    if (in_len + 16 > c->d_in_cap) {
        if (c->d_in) (void)hipFree(c->d_in);
        c->d_in = nullptr;
        c->d_in_cap = 0;
        HIP_TRY(hipMalloc((void **)&c->d_in, in_len + in_len / 8 + 4096));
        c->d_in_cap = in_len + in_len / 8 + 4096;
    }
    if (out_cap + 16 > c->d_out_cap) {
        if (c->d_out) (void)hipFree(c->d_out);
        c->d_out = nullptr;
        c->d_out_cap = 0;
        HIP_TRY(hipMalloc((void **)&c->d_out, out_cap + out_cap / 8 + 4096));
        c->d_out_cap = out_cap + out_cap / 8 + 4096;
    }
    if (in_len) HIP_TRY(hipMemcpyAsync(c->d_in, in, in_len, hipMemcpyHostToDevice, c->stream));
    size_t produced = 0;
    const int rc = decompress_device_locked(c, c->d_in, in_len, offsets, sizes, n_blocks, c->d_out, out_cap,
                                            &produced, info, c->stream);
    if (rc != GZPX_OK) return rc;
    if (produced) HIP_TRY(hipMemcpyAsync(out, c->d_out, produced, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *out_len = produced;
    return GZPX_OK;
}

struct gzpx_decompressor {
    gzpx_dctx *ctx;
    std::vector<uint8_t> framed;
};

gzpx_decompressor *gzpx_alloc_decompressor(void) {
    gzpx_decompressor *d = new (std::nothrow) gzpx_decompressor();
    if (d) d->ctx = nullptr;
    return d;
}

int gzpx_deflate_decompress(gzpx_decompressor *d, const void *in, size_t n, void *out, size_t cap, size_t *actual) {
    if (!d || (!in && n) || (!out && cap)) return GZPX_ERR_INVALID_ARG;
    if (!d->ctx) {
        const int rc = gzpx_dctx_create(0, GZPX_FORMAT_MGZIP, &d->ctx);
        if (rc != GZPX_OK) return rc;
    }
    // the kernels work on framed members: wrap the raw stream with a header and a footer whose
    // ISIZE is the caller's capacity (libdeflate semantics: at most `cap` bytes may come out)
    d->framed.assign(20 + n + 8, 0);
    memcpy(d->framed.data() + 20, in, n);
    uint8_t *f = d->framed.data() + 20 + n;
    const uint32_t isz = (uint32_t)cap;
    f[4] = (uint8_t)isz;
    f[5] = (uint8_t)(isz >> 8);
    f[6] = (uint8_t)(isz >> 16);
    f[7] = (uint8_t)(isz >> 24);
    const uint64_t off = 0;
    const uint32_t size = (uint32_t)d->framed.size();
    std::vector<uint8_t> tmp(cap ? cap : 1);
    size_t produced = 0;
    gzpx_check_info info = {0, 0, 0};
    gzpx_dctx *c = d->ctx;
    int rc = gzpx_decompress_blocks(c, d->framed.data(), d->framed.size(), &off, &size, 1, tmp.data(), cap,
                                    &produced, &info);
    if (rc == GZPX_ERR_INVALID_CHECK) rc = GZPX_OK;  // a raw stream carries no checksum
    if (rc != GZPX_OK) return rc;
    const size_t got = cap ? c->h_blk[0].produced : 0;
    if (cap) {
        // on the InvalidCheck path nothing was copied back: fetch the bytes now
        if (hipMemcpy(tmp.data(), c->d_out, got, hipMemcpyDeviceToHost) != hipSuccess) return GZPX_ERR_DEVICE;
        memcpy(out, tmp.data(), got);
    }
    if (actual) *actual = got;
    return GZPX_OK;
}

void gzpx_free_decompressor(gzpx_decompressor *d) {
    if (!d) return;
    if (d->ctx) gzpx_dctx_destroy(d->ctx);
    delete d;
}

}  // extern "C"

// ---------------------------------------------------------------- measurement / debug hooks
int gzpx_ctx_set_profiling(gzpx_ctx *ctx, int on) {
    if (!ctx) return GZPX_ERR_INVALID_ARG;
    ctx->profiling = on != 0;
    return GZPX_OK;
}

int gzpx_ctx_last_stage_ms(const gzpx_ctx *ctx, float ms[GZPX_N_STAGES]) {
    if (!ctx || !ms) return GZPX_ERR_INVALID_ARG;
    for (int i = 0; i < GZPX_N_STAGES; i++) ms[i] = ctx->stage_ms[i];
    return GZPX_OK;
}

const char *gzpx_stage_name(int stage) {
    static const char *names[GZPX_N_STAGES] = {"k_init_meta", "k_candidates", "k_match", "k_parse", "k_hist",
                                               "k_huffman",   "k_crc32",      "k_scan",  "k_emit"};
    return (stage >= 0 && stage < GZPX_N_STAGES) ? names[stage] : "?";
}

int gzpx_debug_tokens(gzpx_ctx *ctx, size_t block, uint32_t *tokens, size_t max_tokens,
                      size_t *n_tokens, uint32_t *sub_first_token, size_t *n_sub) {
    if (!ctx || !n_tokens || block >= ctx->last_nb) return GZPX_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (hipSetDevice(ctx->cfg.device) != hipSuccess) return GZPX_ERR_DEVICE;
    const BlockMeta &m = ctx->h_meta[block];
    const uint32_t max_sub = ctx->dcfg.max_sub;
    *n_tokens = m.ntok;
    if (n_sub) *n_sub = m.nsub;
    if (sub_first_token && m.nsub) {
        HIP_TRY(hipMemcpy(ctx->h_sub, ctx->scratch.sub + block * (size_t)max_sub,
                          (size_t)m.nsub * sizeof(SubMeta), hipMemcpyDeviceToHost));
        for (uint32_t s = 0; s < m.nsub && s < max_sub; s++) sub_first_token[s] = ctx->h_sub[s].tok_begin;
    }
    const size_t ncopy = m.ntok < max_tokens ? m.ntok : max_tokens;
    if (tokens && ncopy)
        HIP_TRY(hipMemcpy(tokens, ctx->scratch.tok + block * (size_t)ctx->dcfg.stride, ncopy * 4,
                          hipMemcpyDeviceToHost));
    return GZPX_OK;
}

int gzpx_debug_set_flags(gzpx_ctx *ctx, uint32_t flags) {
    if (!ctx) return GZPX_ERR_INVALID_ARG;
    ctx->dcfg.debug = flags;
    return GZPX_OK;
}

int gzpx_debug_phase_cycles(const gzpx_ctx *ctx, uint64_t cycles[8]) {
    if (!ctx || !cycles) return GZPX_ERR_INVALID_ARG;
    for (int k = 0; k < 8; k++) cycles[k] = 0;
    for (uint32_t b = 0; b < ctx->last_nb; b++)
        for (int k = 0; k < 8; k++) cycles[k] += ctx->h_meta[b].phase_cycles[k];
    return GZPX_OK;
}

void *gzpx_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void gzpx_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int gzpx_dctx_last_inflate_ms(gzpx_dctx *ctx, float *ms) {
    if (!ctx || !ms) return GZPX_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    *ms = 0.0f;
    if (!ctx->last_nb) return GZPX_OK;
    return hipEventElapsedTime(ms, ctx->ev[0], ctx->ev[1]) == hipSuccess ? GZPX_OK : GZPX_ERR_DEVICE;
}

int gzpx_debug_inflate(gzpx_dctx *ctx, int enable, uint64_t sums[8]) {
    if (!ctx) return GZPX_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->debug = enable != 0;
    if (sums) {
        for (int k = 0; k < 8; k++) sums[k] = 0;
        for (size_t b = 0; b < ctx->last_nb; b++)
            for (int k = 0; k < 8; k++) sums[k] += ctx->h_blk[b].cyc[k];
    }
    return GZPX_OK;
}

int gzpx_debug_cand_cycles(const gzpx_ctx *ctx, uint64_t cycles[4]) {
    if (!ctx || !cycles) return GZPX_ERR_INVALID_ARG;
    for (int k = 0; k < 4; k++) cycles[k] = 0;
    for (uint32_t b = 0; b < ctx->last_nb; b++)
        for (int k = 0; k < 4; k++) cycles[k] += ctx->h_meta[b].cand_cycles[k];
    return GZPX_OK;
}

const char *gzpx_strerror(int code) {
    switch (code) {
        case GZPX_OK: return "ok";
        case GZPX_ERR_INVALID_ARG: return "invalid argument";
        case GZPX_ERR_BUFFER_SIZE: return "buffer size must be >= 32768 (GzpError::BufferSize)";
        case GZPX_ERR_COMPRESSION_LEVEL: return "invalid compression level (GzpError::LibDeflaterCompressionLvl)";
        case GZPX_ERR_INSUFFICIENT_SPACE: return "insufficient output space (GzpError::LibDeflaterCompress)";
        case GZPX_ERR_BLOCK_SIZE_EXCEEDED: return "compressed block >= 65536 bytes (GzpError::BlockSizeExceeded)";
        case GZPX_ERR_DEVICE: return "HIP runtime error";
        case GZPX_ERR_NO_DEVICE: return "no HIP device (no CPU fallback exists)";
        case GZPX_ERR_UNSUPPORTED: return "configuration valid in gzp but not built yet";
        case GZPX_ERR_NUM_THREADS: return "number of threads must be > 0 (GzpError::NumThreads)";
        case GZPX_ERR_IO: return "the wrapped writer failed (GzpError::Io)";
        case GZPX_ERR_CHANNEL: return "compression pipeline already closed (GzpError::ChannelSend)";
        case GZPX_ERR_INVALID_HEADER: return "invalid block header (GzpError::InvalidHeader)";
        case GZPX_ERR_INVALID_CHECK: return "checksum mismatch (GzpError::InvalidCheck)";
        case GZPX_ERR_BAD_DATA: return "invalid DEFLATE stream (GzpError::LibDelfaterDecompress(BadData))";
        default: return "unknown error";
    }
}

const char *gzpx_device_name(const gzpx_ctx *ctx) { return ctx ? ctx->devname : ""; }
const char *gzpx_version(void) { return "gzpx 0.1 (gfx950)"; }

}  // extern "C"
