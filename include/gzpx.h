/*
 * gzpx.h -- C ABI of the MI355X-native per-block encoder behind gzp's ParCompress<Bgzf/Mgzip>.
 *
 * This is the drop-in boundary: the entry points are what gzp's FFI for this path binds today
 * (libdeflater -> libdeflate-sys), plus one slab-level call that lets the orchestration layer
 * hand thousands of blocks to the GPU at once.  Plain pointers and sizes only.
 *
 * Reference interfaces replaced (paths relative to the gzp tree, v2.0.1):
 *
 *   gzpx_alloc_compressor      libdeflater::Compressor::new            src/deflate.rs:596-599 (Bgzf::create_compressor)
 *                              = libdeflate_alloc_compressor            /opt/conda/include/libdeflate.h:62-69
 *   gzpx_deflate_compress      Compressor::deflate_compress            src/bgzf.rs:214-216, src/mgzip.rs:201-203
 *                              = libdeflate_deflate_compress            libdeflate.h:88-91
 *   gzpx_deflate_compress_bound  Compressor::deflate_compress_bound    libdeflate.h:93-95
 *   gzpx_free_compressor       Drop for Compressor                     libdeflate.h:117
 *   gzpx_crc32                 libdeflater::Crc::update / sum          src/bgzf.rs:224-225, src/check.rs:62,70
 *                              = libdeflate_crc32                       libdeflate.h:343-344
 *   gzpx_encode_block          FormatSpec::encode for Bgzf / Mgzip     src/lib.rs:351-358, src/deflate.rs:613-626, 463-472
 *                              (= bgzf::compress src/bgzf.rs:204-237 + BGZF_EOF src/bgzf.rs:24-38)
 *   gzpx_compress_slab*        the worker loop of ParCompress::run     src/par/compress.rs:279-294, applied to every
 *                              block of a slab cut by ParCompress::write / flush_last (src/par/compress.rs:413-463, 332-362)
 *   error codes                GzpError variants on this path          src/lib.rs:114-163
 *
 * Results are byte-identical to the reference's libdeflate path at the same level (see
 * DESIGN.md for the one libdeflate-version-dependent rule selected by `compat`).
 */
#ifndef GZPX_H
#define GZPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (GzpError classes reachable on the path, src/lib.rs:114-163) ---- */
#define GZPX_OK 0
#define GZPX_ERR_INVALID_ARG 1         /* null pointer, non-multiple slab, ...                       */
#define GZPX_ERR_BUFFER_SIZE 2         /* GzpError::BufferSize: buffer_size < DICT_SIZE (32768)       */
#define GZPX_ERR_COMPRESSION_LEVEL 3   /* GzpError::LibDeflaterCompressionLvl                         */
#define GZPX_ERR_INSUFFICIENT_SPACE 4  /* GzpError::LibDeflaterCompress(InsufficientSpace)            */
#define GZPX_ERR_BLOCK_SIZE_EXCEEDED 5 /* GzpError::BlockSizeExceeded(c, 65536), src/bgzf.rs:218-223  */
#define GZPX_ERR_DEVICE 6              /* HIP runtime error (the Io-like class)                       */
#define GZPX_ERR_NO_DEVICE 7           /* no MI355X / HIP device: there is NO CPU fallback            */
#define GZPX_ERR_UNSUPPORTED 8         /* valid in the reference, not built yet (blocks > 64 MiB) */
#define GZPX_ERR_NUM_THREADS 9         /* GzpError::NumThreads(0), src/par/compress.rs:84-90          */
#define GZPX_ERR_IO 10                 /* GzpError::Io: the wrapped writer failed                     */
#define GZPX_ERR_CHANNEL 11            /* GzpError::ChannelSend/Receive: pipeline already closed      */
#define GZPX_ERR_INVALID_HEADER 12     /* GzpError::InvalidHeader (src/deflate.rs:555-565, 405-415)    */
#define GZPX_ERR_INVALID_CHECK 13      /* GzpError::InvalidCheck{found, expected}                      */
#define GZPX_ERR_BAD_DATA 14           /* GzpError::LibDelfaterDecompress(BadData)                     */
#define GZPX_ERR_BUSY 15               /* submit: every slab slot of the context is in flight          */

/* how a slab is cut (the `mode` argument of gzpx_compress_slab*) */
#define GZPX_SLAB_FULL_BLOCKS 0 /* write(): only whole buffer_size blocks, in_len a non-zero multiple   */
#define GZPX_SLAB_LAST 1        /* flush_last(true): final piece may be short/empty; BGZF EOF appended  */
#define GZPX_SLAB_FLUSH 2       /* flush_last(false): final piece may be short/empty; no EOF marker     */

#define GZPX_FORMAT_BGZF 0
#define GZPX_FORMAT_MGZIP 1

/* libdeflate behaviour pinned by Cargo.lock is 1.24; the image's binary oracle is 1.10.  The two
 * differ (for levels 1-4) only in how a Huffman code with no used symbol is emitted and in
 * min_len for scans shorter than 512 bytes; levels 5-9 are pinned against 1.10 only (no 1.24 binary or
 * source here to compare the lazy parsers with), with the same two rules applied. */
#define GZPX_COMPAT_LIBDEFLATE_1_24 0
#define GZPX_COMPAT_LIBDEFLATE_1_10 1

typedef struct gzpx_config {
    int device;                /* HIP device ordinal                                              */
    int format;                /* GZPX_FORMAT_*                                                   */
    int level;                 /* flate2::Compression level, 0..12 as libdeflate accepts (src/deflate.rs:596-599); 10..12: the near-optimal
                                * parser as in libdeflate 1.10 -- later versions changed it, so at these levels `compat` is
                                * ignored, the 1.10 rules apply throughout (gzpx_ctx_active_compat) and the stream equals the
                                * 1.10 binary's, not 1.24's */
    int compat;                /* GZPX_COMPAT_*                                                   */
    size_t buffer_size;        /* ParCompressBuilder::buffer_size (65280 default for BGZF)        */
    size_t max_slab_bytes;     /* largest slab a single gzpx_compress_slab* call will be given    */
} gzpx_config;

typedef struct gzpx_ctx gzpx_ctx;

/* Fills *cfg with the reference's defaults for `format` (Bgzf: buffer_size 65280, level 3 ->
 * callers set the level they use; compat 1.24; device 0; max_slab_bytes 1 GiB). */
void gzpx_config_default(gzpx_config *cfg, int format);

/* Validates like ParCompressBuilder (src/par/compress.rs:68-74) + CompressionLvl::new and
 * allocates device scratch.  Fails with GZPX_ERR_NO_DEVICE when no GPU is present. */
int gzpx_ctx_create(const gzpx_config *cfg, gzpx_ctx **out);
void gzpx_ctx_destroy(gzpx_ctx *ctx);

/* The GZPX_COMPAT_* rules this context really runs: cfg->compat at levels 0-9; at levels 10-12 always
 * GZPX_COMPAT_LIBDEFLATE_1_10 -- the near-optimal parser built here is libdeflate 1.10's (the one binary there is to
 * pin it on), so its stream is the 1.10 binary's bit for bit and never a mix of two versions' rules. */
int gzpx_ctx_active_compat(const gzpx_ctx *ctx);

/* Upper bound of the bytes gzpx_compress_slab* can produce for in_len input bytes. */
size_t gzpx_slab_bound(const gzpx_ctx *ctx, size_t in_len);

/*
 * Compress one slab held in HOST memory.  The slab is cut into buffer_size blocks exactly as
 * ParCompress::write does; `mode` is one of GZPX_SLAB_*: with FULL_BLOCKS, in_len must be a
 * non-zero multiple of buffer_size (the caller keeps the remainder, as write() does); with LAST
 * or FLUSH the final piece may be short or empty (flush_last), and LAST appends the BGZF EOF
 * marker after it.
 * out receives the framed blocks back to back; block_sizes[i] (optional) the framed size of
 * block i.  On GZPX_ERR_BLOCK_SIZE_EXCEEDED, *n_blocks holds the index of the failing block.
 */
int gzpx_compress_slab(gzpx_ctx *ctx, const uint8_t *in, size_t in_len, int mode, uint8_t *out,
                       size_t out_cap, size_t *out_len, uint32_t *block_sizes, size_t max_blocks,
                       size_t *n_blocks);

/* For the hip_stream / after_stream arguments below: "the caller has already synchronized -- no dependency":
 * nothing is recorded on any stream (NULL, the legacy default stream, waits behind ALL blocking streams of the
 * device and may not be recorded on while a stream capture is running). */
#define GZPX_STREAM_NONE ((void *)(intptr_t)-1)

/* Same, with the slab and the output already resident in DEVICE memory (d_in, d_out are device
 * pointers).  hip_stream (a hipStream_t; NULL = the legacy default stream, which is also PyTorch's
 * current stream unless told otherwise): the slab is read only after everything enqueued on that stream
 * so far has completed -- the context's own streams are non-blocking, so the dependency is always made
 * explicit with an event, also for NULL; GZPX_STREAM_NONE: no dependency at all, the slab is ready now.
 * Synchronous with respect to the host on return. */
int gzpx_compress_slab_device(gzpx_ctx *ctx, const void *d_in, size_t in_len, int mode,
                              void *d_out, size_t out_cap, size_t *out_len, uint32_t *block_sizes,
                              size_t max_blocks, size_t *n_blocks, void *hip_stream);

/*
 * Asynchronous form ("async variant with stream/event handles for H2D || kernel || D2H
 * overlap").  A context owns a copy-in stream, a compute stream and a copy-out stream and up to
 * GZPX_SLOTS slabs in flight, each with its own device staging buffers:
 *
 *   gzpx_compress_slab_submit   enqueues the copy-in of `in` (page-locked memory makes it a DMA
 *                               transfer: gzpx_host_alloc) and every kernel of the slab, and returns
 *                               without waiting for the device, at every level.  GZPX_ERR_BUSY when all
 *                               slots are taken.  `in` and `out` must stay valid until the wait.
 *   gzpx_compress_slab_wait     blocks until that slab's kernels are done, copies exactly the
 *                               produced bytes to `out`, reports like gzpx_compress_slab, frees the slot.
 *
 * Submitting slab k+1 before waiting for slab k overlaps k+1's copy-in with k's kernels and k's
 * copy-out with k+1's kernels; results come back in submission order if waited for in that order.
 * The _device form works on device pointers (no copies); `after_stream` as above.
 * gzpx_compress_slab_event hands out the hipEvent_t that fires when the slab's kernels and result
 * records are complete, for callers that chain their own streams (valid until the wait).
 */
#define GZPX_SLOTS 3
int gzpx_compress_slab_submit(gzpx_ctx *ctx, const uint8_t *in, size_t in_len, int mode, uint8_t *out,
                              size_t out_cap, uint64_t *ticket);
int gzpx_compress_slab_submit_device(gzpx_ctx *ctx, const void *d_in, size_t in_len, int mode, void *d_out,
                                     size_t out_cap, void *after_stream, uint64_t *ticket);
int gzpx_compress_slab_wait(gzpx_ctx *ctx, uint64_t ticket, size_t *out_len, uint32_t *block_sizes,
                            size_t max_blocks, size_t *n_blocks);
int gzpx_compress_slab_event(gzpx_ctx *ctx, uint64_t ticket, void **hip_event);

/*
 * Multi-device form: one context per entry of devices[]; a slab is cut into
 * that many contiguous block ranges (balanced to one block; only the range holding the slab's end
 * takes `mode`), every device compresses its range concurrently, and once the shard sizes are known
 * each device copies its shard straight to its offset in `out` -- the in-order write-out without any
 * payload crossing between GPUs.  Same result, byte for byte, as gzpx_compress_slab on one device.
 * (One process per GPU + an RCCL gather of the shards: gzp_amd/shard.py, bench.py --gpus N.)
 */
typedef struct gzpx_multi gzpx_multi;
int gzpx_multi_create(const gzpx_config *cfg, const int *devices, size_t n_devices, gzpx_multi **out);
void gzpx_multi_destroy(gzpx_multi *m);
size_t gzpx_multi_devices(const gzpx_multi *m);
int gzpx_multi_compress_slab(gzpx_multi *m, const uint8_t *in, size_t in_len, int mode, uint8_t *out,
                             size_t out_cap, size_t *out_len, uint32_t *block_sizes, size_t max_blocks,
                             size_t *n_blocks);

/* The same with every range already resident on its own device -- north_star's "shard of the input slab
 * across 8 GPUs with a gather of compressed blocks over xGMI for in-order write-out": d_in[g] is a device
 * pointer ON devices[g] to range g of the slab (gzpx_multi_shard tells offset and length of range g for a
 * slab of in_len bytes), every device compresses its range into its own staging, and each shard is then
 * copied once, device to device (hipMemcpyPeerAsync, all peers at once), into its stream offset of d_out on
 * devices[root].  No payload passes through host memory; only the 16-byte result records and the
 * per-block sizes do.  Byte for byte the stream of gzpx_compress_slab.  Replaces the writer thread's
 * in-order collection, src/par/compress.rs:305-310.
 * Caller contract: the ranges d_in[g] are complete and d_out is IDLE on entry (no work of the caller still reads
 * or writes it on any stream of devices[root]) -- the call orders its kernels and peer copies among themselves
 * and returns when all of them are done, but takes no stream or event of the caller to wait behind. */
int gzpx_multi_shard(const gzpx_multi *m, size_t in_len, size_t g, size_t *offset, size_t *len);
int gzpx_multi_compress_slab_device(gzpx_multi *m, const void *const *d_in, size_t in_len, int mode, size_t root,
                                    void *d_out, size_t out_cap, size_t *out_len, uint32_t *block_sizes,
                                    size_t max_blocks, size_t *n_blocks);

/* FormatSpec::encode: one framed block (is_last => BGZF_EOF appended for BGZF). */
int gzpx_encode_block(gzpx_ctx *ctx, const uint8_t *in, size_t n, int is_last, uint8_t *out,
                      size_t out_cap, size_t *out_len);

/* ---- libdeflate-shaped per-block ABI (what libdeflater binds) ---- */
typedef struct gzpx_compressor gzpx_compressor;
gzpx_compressor *gzpx_alloc_compressor(int level); /* NULL: bad level / no device / unsupported */
size_t gzpx_deflate_compress(gzpx_compressor *c, const void *in, size_t n, void *out, size_t cap);
size_t gzpx_deflate_compress_bound(gzpx_compressor *c, size_t n);
void gzpx_free_compressor(gzpx_compressor *c);
/* compat / device knobs for the handle above (before first use) */
int gzpx_compressor_set_compat(gzpx_compressor *c, int compat);
uint32_t gzpx_crc32(uint32_t crc, const void *buf, size_t n);
/* libdeflate_crc32's signature cannot report a failure: on a device error gzpx_crc32 returns `crc`
 * unchanged and records the status for the calling thread; gzpx_crc32_checked returns it directly. */
int gzpx_crc32_checked(uint32_t crc, const void *buf, size_t n, uint32_t *out);
int gzpx_last_status(void);

/* The checks of gzp's Gzip / Zlib formats (src/check.rs:85-164) as helpers; the encoders of those formats are not
 * built (zlib-ng's output cannot be pinned in this image), so nothing inside the library calls these.
 *   gzpx_crc32_combine    Crc32::combine (flate2::Crc::combine, src/check.rs:160-163): the CRC-32 of A || B from crc(A),
 *                         crc(B) and |B|.  Arithmetic only.  (Crc32::update is gzpx_crc32 above.)
 *   gzpx_adler32          Adler32::update (libz_ng_sys::adler32, src/check.rs:112-119): start with 1.  On the device;
 *                         errors as gzpx_crc32 (gzpx_last_status), gzpx_adler32_checked returns them.
 *   gzpx_adler32_combine  Adler32::combine (adler32_combine, src/check.rs:121-127).  Arithmetic only. */
uint32_t gzpx_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2);
uint32_t gzpx_adler32(uint32_t adler, const void *buf, size_t n);
int gzpx_adler32_checked(uint32_t adler, const void *buf, size_t n, uint32_t *out);
uint32_t gzpx_adler32_combine(uint32_t adler1, uint32_t adler2, uint64_t len2);

/* ---- ParCompress<Bgzf/Mgzip> twin: Write + ZWriter::finish over device lanes (C++ class
 * gzp::ParCompress in gzp_amd/csrc/gzpx_par.hpp; src/par/compress.rs:33-469) ---- */
typedef struct gzpx_par gzpx_par;
typedef int (*gzpx_write_fn)(void *user, const uint8_t *data, size_t n); /* 0 = ok (the `W: Write`) */
typedef struct gzpx_par_config {
    int format;          /* GZPX_FORMAT_*                                               */
    int level;           /* ParCompressBuilder::compression_level                       */
    int compat;          /* GZPX_COMPAT_*                                               */
    int device;          /* HIP device                                                  */
    size_t buffer_size;  /* ParCompressBuilder::buffer_size (>= 32768)                  */
    size_t num_threads;  /* ParCompressBuilder::num_threads (> 0); > 1 adds copy helpers */
    size_t batch_blocks; /* blocks per slab handed to the device (0 = default 1024; a slab is capped at 128 MiB) */
} gzpx_par_config;
int gzpx_par_create(const gzpx_par_config *cfg, gzpx_write_fn write_fn, void *user, gzpx_par **out);
/* the same with ParCompressBuilder::pin_threads(Some(first_core)) (src/par/compress.rs:99-107): the
 * twin's device thread and copy helpers are pinned to consecutive cores the process may run on */
int gzpx_par_create_pinned(const gzpx_par_config *cfg, size_t first_core, gzpx_write_fn write_fn, void *user,
                           gzpx_par **out);
int gzpx_par_write(gzpx_par *p, const uint8_t *buf, size_t n);
/* The loop of the reference's own benchmark (benches/bench.rs:36-45: read 64 KiB, write_all it) on the
 * native side of the boundary: n bytes through write() calls of `chunk` bytes each.  What a Rust caller
 * of gzpx_par_write sees without this binding's per-call cost; same stream as one write() of n bytes. */
int gzpx_par_write_chunked(gzpx_par *p, const uint8_t *buf, size_t n, size_t chunk);
/* In-place form of write() for producers that can fill memory they are handed (read(2) into the
 * slab, a decoder's output): reserve returns room (>= buffer_size bytes) inside the page-locked slab
 * that is being filled, commit appends the first n bytes of it to the stream.  Same cut rule as
 * write(), no copy on the way to the device. */
int gzpx_par_reserve(gzpx_par *p, uint8_t **ptr, size_t *cap);
int gzpx_par_commit(gzpx_par *p, size_t n);
int gzpx_par_flush(gzpx_par *p);
int gzpx_par_finish(gzpx_par *p);
void gzpx_par_destroy(gzpx_par *p);
const char *gzpx_par_last_error(const gzpx_par *p);

/* ---- block index side-product (README.md:161 "Return an auto-generated index for BGZF / Mgzip
 * formats"): where every block written so far starts in the compressed and in the uncompressed
 * stream, in stream order; complete after gzpx_par_finish.  gzpx_gzi_write serialises it in
 * htslib's .gzi layout (u64 count, then (compressed, uncompressed) u64 pairs of every block but
 * the first, little-endian). */
typedef struct gzpx_index_entry {
    uint64_t compressed_offset;
    uint64_t uncompressed_offset;
} gzpx_index_entry;
int gzpx_par_index(gzpx_par *p, gzpx_index_entry *entries, size_t max_entries, size_t *n_entries);
size_t gzpx_gzi_size(size_t n_entries);
int gzpx_gzi_write(const gzpx_index_entry *entries, size_t n_entries, uint8_t *out, size_t out_cap,
                   size_t *out_len);

/* ---- ParDecompress<Bgzf/Mgzip> (src/par/decompress.rs:132-337; BlockFormatSpec src/lib.rs:411-448) ----
 *   gzpx_scan_blocks            the reader thread's header walk: check_header + get_block_size
 *                               (src/deflate.rs:555-570 / 405-422) over a host buffer
 *   gzpx_decompress_blocks*     the worker loop (src/par/decompress.rs:162-186) for every block of a
 *                               slab: get_footer_values, decode_block = libdeflate_deflate_decompress
 *                               into ISIZE bytes, LibDeflateCrc check (src/check.rs:38-82)
 *   gzpx_*_decompressor         libdeflater::Decompressor (libdeflate.h: libdeflate_alloc_decompressor,
 *                               libdeflate_deflate_decompress, libdeflate_free_decompressor)
 */
typedef struct gzpx_dctx gzpx_dctx;
typedef struct gzpx_check_info {
    size_t block;            /* index of the block that failed (any error)          */
    uint32_t found, expected; /* InvalidCheck { found, expected }                    */
} gzpx_check_info;
int gzpx_dctx_create(int device, int format, gzpx_dctx **out);
void gzpx_dctx_destroy(gzpx_dctx *ctx);
/* Walks the block headers in in[0..in_len).  offsets[i] / sizes[i]: start and total size of block
 * i; *consumed: bytes covered by complete blocks (a trailing partial block is left to the caller,
 * as read_exact would block on it).  GZPX_ERR_INVALID_HEADER for a header that fails check_header. */
int gzpx_scan_blocks(int format, const uint8_t *in, size_t in_len, uint64_t *offsets, uint32_t *sizes,
                     size_t max_blocks, size_t *n_blocks, size_t *consumed);
int gzpx_decompress_blocks(gzpx_dctx *ctx, const uint8_t *in, size_t in_len, const uint64_t *offsets,
                           const uint32_t *sizes, size_t n_blocks, uint8_t *out, size_t out_cap,
                           size_t *out_len, gzpx_check_info *info);
/* asynchronous form (same slots / streams scheme as gzpx_compress_slab_submit): the copy-in, the
 * kernels and the copy-out of the inflated bytes (their count is the sum of the footers' ISIZE fields,
 * known at submit time) are all enqueued by submit; wait blocks for them and reports the first
 * failing block in stream order.  offsets / sizes are copied by submit. */
int gzpx_decompress_blocks_submit(gzpx_dctx *ctx, const uint8_t *in, size_t in_len, const uint64_t *offsets,
                                  const uint32_t *sizes, size_t n_blocks, uint8_t *out, size_t out_cap,
                                  uint64_t *ticket);
int gzpx_decompress_blocks_wait(gzpx_dctx *ctx, uint64_t ticket, size_t *out_len, gzpx_check_info *info);
/* d_in / d_out are device pointers; offsets / sizes stay host arrays */
int gzpx_decompress_blocks_device(gzpx_dctx *ctx, const void *d_in, size_t in_len, const uint64_t *offsets,
                                  const uint32_t *sizes, size_t n_blocks, void *d_out, size_t out_cap,
                                  size_t *out_len, gzpx_check_info *info, void *hip_stream);
typedef struct gzpx_decompressor gzpx_decompressor;
gzpx_decompressor *gzpx_alloc_decompressor(void);
/* 0 = ok (short output allowed, *actual = bytes produced), GZPX_ERR_BAD_DATA, GZPX_ERR_INSUFFICIENT_SPACE */
int gzpx_deflate_decompress(gzpx_decompressor *d, const void *in, size_t n, void *out, size_t cap, size_t *actual);
void gzpx_free_decompressor(gzpx_decompressor *d);

/* ParDecompress twin: `Read` over a block stream (C++ class gzp::ParDecompress, gzpx_par.hpp) */
typedef struct gzpx_pard gzpx_pard;
typedef long (*gzpx_read_fn)(void *user, uint8_t *buf, size_t cap); /* bytes read, 0 = EOF, < 0 = error */
int gzpx_pard_create(int format, int device, size_t batch_bytes, gzpx_read_fn read_fn, void *user,
                     gzpx_pard **out);
int gzpx_pard_read(gzpx_pard *p, uint8_t *buf, size_t n, size_t *got);
/* std::io::BufRead's fill_buf / consume for the same stream: *ptr = the inflated bytes where they lie (the current slab's
 * page-locked buffer, valid until the next fill_buf / read that follows a consume of all of them), *len = how many
 * (0 at the end of the stream).  gzp's ParDecompress is `Read` only (src/par/decompress.rs:241-352); a binding that
 * wants no copy between the slab and its own buffer implements BufRead over these two. */
int gzpx_pard_fill_buf(gzpx_pard *p, const uint8_t **ptr, size_t *len);
int gzpx_pard_consume(gzpx_pard *p, size_t n);
void gzpx_pard_destroy(gzpx_pard *p);
const char *gzpx_pard_last_error(const gzpx_pard *p);

/* Page-locked host memory for slab staging.  A caller that fills its slabs in such buffers (the
 * twin does) turns the library's copies to and from the device into DMA transfers that overlap
 * with the other lane's kernels: 12 GiB/s host to host instead of 6 with pageable memory. */
void *gzpx_host_alloc(size_t bytes);
void gzpx_host_free(void *p);

/* ---- workload support: the synthetic FASTQ stream of BASELINE configs[3] (32 GiB sharded over 8
 * GPUs), generated in HBM.  Fills d_out[0..n) with bytes [stream_offset, stream_offset + n) of the
 * stream with this seed (oracle/synth_fastq.c states the stream on the CPU). ---- */
int gzpx_synth_fastq_device(void *d_out, uint64_t stream_offset, uint64_t n, uint64_t seed, void *hip_stream);
/* the printable-ASCII noise of BASELINE configs[2] (byte i = 0x20 + (splitmix64 output i >> 56) % 95) */
int gzpx_synth_ascii_device(void *d_out, uint64_t stream_offset, uint64_t n, uint64_t seed, void *hip_stream);

/* ---- measurement hooks (HIP events on the launching stream; bench.py roofline leg) ---- */
#define GZPX_N_STAGES 9
/* stage order: init_meta, candidates, match, parse, hist, huffman, crc32, scan, emit */
/* on: 0 off, 1 HIP events around every stage, 2 around the dominant stage (2: match) only -- two markers in the
 * stream instead of thirteen, for a timed region that wants the kernel's duration without paying for the rest */
int gzpx_ctx_set_profiling(gzpx_ctx *ctx, int on);
int gzpx_ctx_last_stage_ms(const gzpx_ctx *ctx, float ms[GZPX_N_STAGES]);
const char *gzpx_stage_name(int stage);
/* the kernel(s) behind a stage for THIS context: stage 2 is k_mparse (level 1, blocks <= 64 KiB: match
 * on demand; k_match / k_parse then only see the blocks it hands back), k_match (level 1, larger
 * blocks), k_match_hc + k_parse_hc (levels 2-4) or k_match_hc + k_parse_lazy (levels 5-9) */
const char *gzpx_ctx_stage_kernel(const gzpx_ctx *ctx, int stage);

/* ---- test hooks: intermediate products of the last slab call (device -> host copies) ---- */
int gzpx_debug_tokens(gzpx_ctx *ctx, size_t block, uint32_t *tokens, size_t max_tokens,
                      size_t *n_tokens, uint32_t *sub_first_token, size_t *n_sub);

/* Diagnostics switches (0 in production): bit 0 = k_candidates takes its order-independent
 * fallback (cand_block_safe) on every block instead of the atomic-chain form; bit 1 = level 1 through the
 * dense k_match / k_parse pair instead of the match-on-demand kernel k_mparse; bit 2 = k_mparse
 * hands every block back to the dense pair (exercises the redo list). */
int gzpx_debug_set_flags(gzpx_ctx *ctx, uint32_t flags);
/* Level 1: how many blocks of the last batch k_mparse handed back to the dense kernels. */
int gzpx_debug_redo_count(gzpx_ctx *ctx, uint32_t *count);

/* HIP-event duration of the inflate kernels (k_inflate_seg + k_lzcopy + k_inflate over the redo list, or
 * k_inflate alone) in the last decompress launch of this context */
int gzpx_dctx_last_inflate_ms(gzpx_dctx *ctx, float *ms);
/* GZPX_INFLATE_SEG: the same split in two, ms[0] = k_inflate_seg (Huffman decode), ms[1] = k_lzcopy + k_inflate over the
 * hand-backs (zeros on the other route) */
int gzpx_dctx_last_inflate_stage_ms(gzpx_dctx *ctx, float ms[2]);
/* Which kernels inflate (decode_block, src/par/decompress.rs:162-186): GZPX_INFLATE_SEG (default) = the decode /
 * LZ-copy pair, members they cannot take handed to k_inflate; GZPX_INFLATE_WAVE = k_inflate (one wave per member,
 * window in HBM) for every member.  Same bytes, same error classes either way. */
#define GZPX_INFLATE_SEG 0
#define GZPX_INFLATE_WAVE 1
int gzpx_dctx_set_route(gzpx_dctx *ctx, int route);
/* How many members of the last launch the decode / copy pair handed to k_inflate (diagnostics). */
int gzpx_dctx_last_redo_count(gzpx_dctx *ctx, uint32_t *count);
/* inflate: switch the instrumented kernels on (1, or 2 on the GZPX_INFLATE_SEG route for k_lzcopy's clocks instead
 * of k_inflate_seg's) / off (0); sums[] = per-member counters of the last instrumented launch summed over its members.
 * enable = 1, GZPX_INFLATE_SEG: [0] cycles of k_inflate_seg, [1] headers + tables, [2] pass 1, [3] pass 2, [4] pass 3,
 * [5] spans, [6] pass-2 iterations, [7] symbol steps of passes 1 and 3.  enable = 2: [0] cycles of k_lzcopy, [1] tile
 * staged in, [2] chunk set-up, [3] polling, [4] CRC + tile written out, [5] polling iterations, [6] of them without
 * progress, [7] matches.  GZPX_INFLATE_WAVE: [0] cycles, [1] headers + tables, [2] round set-up, [3] stores + copies,
 * [4] rounds, [5] literals, [6] matches, [7] flushes */
int gzpx_debug_inflate(gzpx_dctx *ctx, int enable, uint64_t sums[8]);

const char *gzpx_strerror(int code);
const char *gzpx_device_name(const gzpx_ctx *ctx);
const char *gzpx_version(void);
/* Identifies the sources this library was built from (gzp_amd/build.py: source_id()); "unknown" for builds made
 * another way.  profiles/pmc_traffic.json carries the id of the build its counters were collected with. */
const char *gzpx_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* GZPX_H */
