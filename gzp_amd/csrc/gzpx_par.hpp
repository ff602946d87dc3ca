// gzpx_par.hpp -- C++ twin of gzp's user-facing writer / reader API for the block formats, on top of
// the GPU slab encoder (include/gzpx.h).  Rust is not available in the build image, so the host
// side above the C ABI is restated in C++ with the reference's names, argument meaning and error
// behaviour; INTEGRATION.md shows the Rust binding that replaces this file inside gzp.
//
// Reference (paths relative to the gzp tree):
//   ParCompressBuilder          src/par/compress.rs:33-204 (from_writer :110-138, from_borrowed_writer :162-194)
//   ParCompress (Write/ZWriter) src/par/compress.rs:221-469
//   ParDecompressBuilder        src/par/decompress.rs:17-109
//   ParDecompress (Read)        src/par/decompress.rs:112-352
//   ZBuilder                    src/lib.rs:181-275
//   GzpError                    src/lib.rs:114-163
//   Compression                 flate2::Compression (re-exported src/lib.rs:81)
//   Bgzf / Mgzip                src/deflate.rs:506-653, 357-498
//
// Orchestration kept from the reference: the caller thread only buffers and cuts (strict `>` rule,
// src/par/compress.rs:415), sends an ORDER TOKEN to the writer queue first and the work item second
// (src/par/compress.rs:424-457), both queues are bounded (back-pressure), and a single writer thread
// emits results in submission order (src/par/compress.rs:305-310).  What changes is the unit of
// work and the worker: a work item is a *slab* of `batch_blocks` blocks in page-locked memory
// instead of one block, and the N compressor threads become one device thread that keeps up to
// GZPX_SLOTS slabs in flight through the asynchronous slab ABI -- the copy-in of slab k+1, the
// kernels of slab k and the copy-out of slab k-1 run side by side on three HIP streams.
// ParDecompress mirrors it: a reader thread walks the block headers and fills page-locked slabs
// (src/par/decompress.rs:195-209), a device thread inflates them GZPX_SLOTS at a time
// (:162-186), and read() drains a bounded queue of inflated slabs in stream order.
#pragma once

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/gzpx.h"

namespace gzp {

constexpr size_t BUFSIZE = 64 * (1 << 10) * 2;  // src/lib.rs:105
constexpr size_t DICT_SIZE = 32768;             // src/lib.rs:108

// GzpError variants reachable on this path (src/lib.rs:114-163)
enum class GzpErrorKind {
    BufferSize,                 // BufferSize(got, min)
    NumThreads,                 // NumThreads(0)
    LibDeflaterCompressionLvl,  // invalid level
    LibDeflaterCompress,        // InsufficientSpace
    BlockSizeExceeded,          // BlockSizeExceeded(c, 65536)
    Io,                         // error returned by the wrapped writer
    ChannelSend,                // write()/finish() after the pipeline died
    ChannelReceive,
    InvalidHeader,              // InvalidHeader(&str)
    InvalidCheck,               // InvalidCheck { found, expected }
    LibDeflaterDecompress,      // LibDelfaterDecompress(BadData | InsufficientSpace)
    Device,                     // HIP failure / no device (no CPU fallback exists)
    Unsupported,                // valid in gzp, not built yet
};

class GzpError : public std::runtime_error {
  public:
    GzpError(GzpErrorKind k, const std::string &msg) : std::runtime_error(msg), kind(k) {}
    GzpErrorKind kind;
};

GzpError error_from_code(int gzpx_code, size_t block = 0);

// flate2::Compression
struct Compression {
    int lvl;
    explicit Compression(int l = 6) : lvl(l) {}
    static Compression none() { return Compression(0); }
    static Compression fast() { return Compression(1); }
    static Compression best() { return Compression(9); }
    int level() const { return lvl; }
};

struct Bgzf {
    static constexpr size_t DEFAULT_BUFSIZE = 65280;  // BGZF_BLOCK_SIZE, src/deflate.rs:583
    static constexpr int FORMAT = GZPX_FORMAT_BGZF;
};
struct Mgzip {
    static constexpr size_t DEFAULT_BUFSIZE = BUFSIZE;  // trait default, src/lib.rs:330
    static constexpr int FORMAT = GZPX_FORMAT_MGZIP;
};

// The wrapped `W: Write`: returns false on an I/O error (message in *err).
using WriteFn = std::function<bool(const uint8_t *data, size_t n, std::string *err)>;

// One entry of the block index (README.md:161 "auto-generated index for BGZF / Mgzip formats"):
// where block i starts in the compressed stream and in the uncompressed stream.
struct IndexEntry {
    uint64_t compressed_offset;
    uint64_t uncompressed_offset;
};

struct ParConfig {
    int format = GZPX_FORMAT_BGZF;
    size_t buffer_size = Bgzf::DEFAULT_BUFSIZE;
    size_t num_threads = 0;  // 0 = "all cores" default of the builder; only >= 1 after validation
    Compression compression_level = Compression(3);  // src/par/compress.rs:54-62
    std::optional<size_t> pin_threads;               // first core of the twin's threads (src/par/compress.rs:99-107)
    // GPU-side knobs (no counterpart in the reference)
    int device = 0;
    int compat = GZPX_COMPAT_LIBDEFLATE_1_24;
    size_t batch_blocks = 1024;  // blocks per slab handed to the device (clamped to kMaxSlabBytes)
};

// Helper threads that split one large memcpy (a big write() into a page-locked slab) between them:
// one core copies ~12 GB/s, the copy engines take ~50.
class CopyPool {
  public:
    explicit CopyPool(size_t helpers, std::optional<size_t> pin_at = std::nullopt);
    ~CopyPool();
    void copy(uint8_t *dst, const uint8_t *src, size_t n);

  private:
    struct Task {
        uint8_t *dst;
        const uint8_t *src;
        size_t n;
    };
    void main();
    std::mutex mu_;
    std::condition_variable cv_task_, cv_done_;
    std::deque<Task> tasks_;
    size_t pending_ = 0;
    bool stop_ = false;
    std::vector<std::thread> threads_;
};

// ZWriter (src/lib.rs:166-170) + std::io::Write
class ParCompress {
  public:
    // slabs are capped so that the page-locked staging stays bounded whatever buffer_size is asked for
    static constexpr size_t kMaxSlabBytes = (size_t)128 << 20;

    ParCompress(const ParConfig &cfg, WriteFn writer);
    ~ParCompress();  // Drop: finish() if not finished (src/par/compress.rs:391-402)
    ParCompress(const ParCompress &) = delete;
    ParCompress &operator=(const ParCompress &) = delete;

    size_t write(const uint8_t *buf, size_t n);  // src/par/compress.rs:413-463
    void write_all(const uint8_t *buf, size_t n) { write(buf, n); }
    void flush();   // src/par/compress.rs:466-468 -> flush_last(false)
    void finish();  // src/par/compress.rs:377-388 -> flush_last(true), join

    // In-place variant of write() for producers that can fill memory they are handed (a file read,
    // a decoder): reserve() returns room inside the current page-locked slab (at least one byte),
    // commit(n) appends the first n bytes of it to the stream -- the same cut rule as write(), with
    // no copy at all.
    std::pair<uint8_t *, size_t> reserve();
    void commit(size_t n);

    // (compressed offset, uncompressed offset) of every block written so far, in stream order.
    // Complete after finish().
    std::vector<IndexEntry> index() const;

    size_t effective_batch_blocks() const { return batch_blocks_; }

  private:
    struct Pinned {
        uint8_t *p = nullptr;
        size_t cap = 0;
        size_t len = 0;
    };
    struct Done {
        Pinned out;
        std::vector<uint32_t> block_sizes;  // framed size of every block of the slab
        size_t in_len = 0;
    };
    struct Job {
        Pinned input;  // whole blocks (or the final short piece)
        int mode = GZPX_SLAB_FULL_BLOCKS;
        std::promise<Done> result;
    };
    struct InFlight {
        std::unique_ptr<Job> job;
        Pinned out;
        uint64_t ticket = 0;
    };
    Pinned take_buffer(size_t cap);
    void give_buffer(Pinned b);
    void flush_last(bool is_last);
    void dispatch(Pinned input, int mode);
    void after_append();
    // a slab is filled up to one block past the batch: the strict `>` cut rule then leaves that tail
    size_t fill_room() const { return batch_bytes_ + cfg_.buffer_size - fill_.len; }
    void device_main();
    void writer_main();
    void raise_pipeline_error();
    void complete(InFlight &f);

    ParConfig cfg_;
    WriteFn writer_;
    size_t batch_blocks_ = 0;
    size_t batch_bytes_ = 0;
    size_t buf_cap_ = 0;  // capacity of every staging buffer of the pool
    Pinned fill_;  // the slab being filled by write() / reserve()
    bool finished_ = false;
    std::unique_ptr<CopyPool> copier_;

    // bounded queues (flume::bounded(2N) in the reference)
    std::mutex mu_;
    std::condition_variable cv_work_, cv_order_, cv_space_;
    std::deque<std::unique_ptr<Job>> work_q_;
    std::deque<std::future<Done>> order_q_;
    std::mutex pool_mu_;
    std::vector<Pinned> pool_;       // free buffers
    std::vector<uint8_t *> pinned_;  // every buffer ever allocated (freed by the destructor)
    size_t q_cap_ = 0;
    bool closed_ = false;
    bool failed_ = false;
    std::exception_ptr error_;

    mutable std::mutex index_mu_;
    std::vector<IndexEntry> index_;
    uint64_t coff_ = 0, uoff_ = 0;

    std::thread device_thread_;
    std::thread writer_thread_;
    gzpx_ctx *ctx_ = nullptr;
};

// ParCompressBuilder<F> (src/par/compress.rs:33-204)
template <class F>
class ParCompressBuilder {
  public:
    ParCompressBuilder() {
        cfg_.format = F::FORMAT;
        cfg_.buffer_size = F::DEFAULT_BUFSIZE;
        cfg_.num_threads = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 1;
    }
    ParCompressBuilder &buffer_size(size_t n) {
        if (n < DICT_SIZE)
            throw GzpError(GzpErrorKind::BufferSize, "Invalid buffer size " + std::to_string(n) +
                                                         ", must be >= " + std::to_string(DICT_SIZE));
        cfg_.buffer_size = n;
        return *this;
    }
    ParCompressBuilder &num_threads(size_t n) {
        if (n == 0) throw GzpError(GzpErrorKind::NumThreads, "Invalid number of threads 0");
        cfg_.num_threads = n;
        return *this;
    }
    ParCompressBuilder &compression_level(Compression c) {
        cfg_.compression_level = c;
        return *this;
    }
    ParCompressBuilder &pin_threads(std::optional<size_t> p) {
        cfg_.pin_threads = p;
        return *this;
    }
    ParCompressBuilder &device(int d) {
        cfg_.device = d;
        return *this;
    }
    ParCompressBuilder &compat(int c) {
        cfg_.compat = c;
        return *this;
    }
    ParCompressBuilder &batch_blocks(size_t b) {
        cfg_.batch_blocks = b ? b : 1;
        return *this;
    }
    // from_writer: the ParCompress owns the writer (a closure here)
    std::unique_ptr<ParCompress> from_writer(WriteFn w) const {
        return std::make_unique<ParCompress>(cfg_, std::move(w));
    }
    // from_borrowed_writer (src/par/compress.rs:162-194): the caller keeps the writer and lends it
    // for the lifetime of the ParCompress -- which must be finished (or dropped) before `w` goes
    // away, exactly the contract the reference states for its scoped-thread variant.  W needs
    // `bool write(const uint8_t *, size_t, std::string *err)`.
    template <class W>
    std::unique_ptr<ParCompress> from_borrowed_writer(W &w) const {
        W *borrowed = &w;
        return std::make_unique<ParCompress>(
            cfg_, [borrowed](const uint8_t *d, size_t n, std::string *err) { return borrowed->write(d, n, err); });
    }
    const ParConfig &config() const { return cfg_; }

  private:
    ParConfig cfg_;
};

// ZBuilder<F, W> (src/lib.rs:181-275): the one-stop builder.  In the reference num_threads <= 1 selects
// the single-threaded SyncZ; the GPU path has no such variant, so every setting builds a ParCompress.
template <class F>
class ZBuilder {
  public:
    ZBuilder &buffer_size(size_t n) {
        b_.buffer_size(n);
        return *this;
    }
    ZBuilder &num_threads(size_t n) {
        threads_ = n;
        return *this;
    }
    ZBuilder &compression_level(Compression c) {
        b_.compression_level(c);
        return *this;
    }
    ZBuilder &pin_threads(std::optional<size_t> p) {
        b_.pin_threads(p);
        return *this;
    }
    std::unique_ptr<ParCompress> from_writer(WriteFn w) {
        b_.num_threads(threads_ ? threads_ : 1);
        return b_.from_writer(std::move(w));
    }

  private:
    ParCompressBuilder<F> b_;
    size_t threads_ = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 1;
};

// The wrapped `R: Read`: returns the number of bytes read into buf (0 = end of stream), -1 on error.
using ReadFn = std::function<long(uint8_t *buf, size_t cap, std::string *err)>;

struct ParDecompressConfig {
    int format = GZPX_FORMAT_BGZF;
    size_t num_threads = 0;  // accepted for API parity (src/par/decompress.rs:43-50)
    int device = 0;
    size_t batch_bytes = (size_t)16 << 20;  // compressed bytes handed to the GPU per slab (round 6: 16 MiB, not 64 -- three
                                            // slabs in flight overlap better: 2.2 GiB of text host to host 8.9 -> 20.6 GiB/s)
};

// ParDecompress (src/par/decompress.rs:112-352): `Read` over a block-compressed stream.
//   reader thread  header walk (check_header + get_block_size, :195-209) over page-locked slabs of
//                  input; a slab = the whole blocks among ~batch_bytes read ahead
//   device thread  decode_block + per-block CRC check (:162-186) for every block of a slab in one
//                  launch, up to GZPX_SLOTS slabs in flight
//   read()         drains the inflated slabs in stream order from a bounded queue (:260-337); an
//                  error travels in the queue and surfaces at the position it belongs to
class ParDecompress {
  public:
    ParDecompress(const ParDecompressConfig &cfg, ReadFn reader);
    ~ParDecompress();
    ParDecompress(const ParDecompress &) = delete;
    ParDecompress &operator=(const ParDecompress &) = delete;
    size_t read(uint8_t *buf, size_t n);  // 0 = end of stream
    // std::io::BufRead's pair (round 6): the inflated bytes where they lie -- the current slab's page-locked buffer --
    // instead of a copy into the caller's; *len = 0 at the end of the stream.  `read` is fill_buf + memcpy + consume.
    const uint8_t *fill_buf(size_t *len);
    void consume(size_t n);
    void finish();                        // src/par/decompress.rs:222-238

  private:
    struct Staging {
        uint8_t *p = nullptr;
        size_t cap = 0, len = 0;
    };
    struct Slab {  // one unit of work, reader -> device -> read()
        Staging in, out;
        std::vector<uint64_t> offs;
        std::vector<uint32_t> sizes;
        size_t used = 0, total = 0;
        std::exception_ptr error;  // set instead of data: surfaces in read() at this position
        bool end = false;          // end-of-stream marker
        uint64_t ticket = 0;
    };
    using SlabPtr = std::unique_ptr<Slab>;
    void reserve(Staging &s, size_t cap, size_t keep);
    void reader_main();
    void device_main();
    bool push(std::deque<SlabPtr> &q, SlabPtr s, std::condition_variable &cv);
    SlabPtr pop(std::deque<SlabPtr> &q, std::condition_variable &cv, bool wait);
    SlabPtr recycle();
    void stop_threads();

    ParDecompressConfig cfg_;
    ReadFn reader_;
    gzpx_dctx *ctx_ = nullptr;
    std::mutex mu_;
    std::condition_variable cv_in_, cv_out_, cv_space_;
    std::deque<SlabPtr> in_q_, out_q_, free_;
    size_t q_cap_ = 4;
    bool stop_ = false;
    std::thread reader_thread_, device_thread_;
    SlabPtr cur_;  // the slab read() is draining
    size_t out_pos_ = 0;
    bool done_ = false;
    std::exception_ptr sticky_;
    std::vector<uint8_t *> pinned_;
};

// ParDecompressBuilder<F> (src/par/decompress.rs:17-109)
template <class F>
class ParDecompressBuilder {
  public:
    ParDecompressBuilder() { cfg_.format = F::FORMAT; }
    ParDecompressBuilder &num_threads(size_t n) {
        if (n == 0) throw GzpError(GzpErrorKind::NumThreads, "Invalid number of threads 0");
        cfg_.num_threads = n;
        return *this;
    }
    ParDecompressBuilder &device(int d) {
        cfg_.device = d;
        return *this;
    }
    ParDecompressBuilder &batch_bytes(size_t b) {
        if (b) cfg_.batch_bytes = b;
        return *this;
    }
    std::unique_ptr<ParDecompress> from_reader(ReadFn r) const {
        return std::make_unique<ParDecompress>(cfg_, std::move(r));
    }

  private:
    ParDecompressConfig cfg_;
};

}  // namespace gzp
