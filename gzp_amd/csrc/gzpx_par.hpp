// gzpx_par.hpp -- C++ twin of gzp's user-facing writer API for the block formats, on top of the
// GPU slab encoder (include/gzpx.h).  Rust is not available in the build image, so the host side
// above the C ABI is restated in C++ with the reference's names, argument meaning and error
// behaviour; INTEGRATION.md shows the ~40-line Rust binding that replaces this file inside gzp.
//
// Reference (paths relative to the gzp tree):
//   ParCompressBuilder          src/par/compress.rs:33-204
//   ParCompress (Write/ZWriter) src/par/compress.rs:221-469
//   ZBuilder                    src/lib.rs:181-275
//   GzpError                    src/lib.rs:114-163
//   Compression                 flate2::Compression (re-exported src/lib.rs:81)
//   Bgzf / Mgzip                src/deflate.rs:506-653, 357-498
//
// Orchestration kept from the reference: the caller thread only buffers and cuts
// (strict `>` rule, src/par/compress.rs:415), sends an ORDER TOKEN to the writer queue first and
// the work item second (src/par/compress.rs:424-457), both queues are bounded (back-pressure),
// workers own one compressor each (src/par/compress.rs:278), and a single writer thread emits
// results in submission order (src/par/compress.rs:305-310).  What changes is the unit of work:
// a worker is a *device lane* (its own gzpx_ctx = its own HIP stream and device buffers) and a
// work item is a slab of `batch_blocks` blocks instead of one block, so that thousands of blocks
// go to the GPU per launch while lane A's copies overlap lane B's kernels.
#pragma once

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

struct ParConfig {
    int format = GZPX_FORMAT_BGZF;
    size_t buffer_size = Bgzf::DEFAULT_BUFSIZE;
    size_t num_threads = 0;  // 0 = "all cores" default of the builder; only >= 1 after validation
    Compression compression_level = Compression(3);  // src/par/compress.rs:54-62
    std::optional<size_t> pin_threads;               // accepted, no effect (device lanes)
    // GPU-side knobs (no counterpart in the reference)
    int device = 0;
    int compat = GZPX_COMPAT_LIBDEFLATE_1_24;
    size_t batch_blocks = 1024;  // blocks per slab handed to one device lane
    std::string library;         // reserved
};

// ZWriter (src/lib.rs:166-170) + std::io::Write
class ParCompress {
  public:
    ParCompress(const ParConfig &cfg, WriteFn writer);
    ~ParCompress();  // Drop: finish() if not finished (src/par/compress.rs:391-402)
    ParCompress(const ParCompress &) = delete;
    ParCompress &operator=(const ParCompress &) = delete;

    size_t write(const uint8_t *buf, size_t n);  // src/par/compress.rs:413-463
    void write_all(const uint8_t *buf, size_t n) { write(buf, n); }
    void flush();   // src/par/compress.rs:466-468 -> flush_last(false)
    void finish();  // src/par/compress.rs:377-388 -> flush_last(true), join

  private:
    // Slabs travel in page-locked host memory (a small pool, reused), so that the copies to and
    // from the device are DMA transfers that overlap with the other lane's kernels.
    struct Pinned {
        uint8_t *p = nullptr;
        size_t cap = 0;
        size_t len = 0;
    };
    struct Job {
        Pinned input;  // whole blocks (or the final short piece)
        int mode = GZPX_SLAB_FULL_BLOCKS;
        std::promise<Pinned> result;
    };
    Pinned take_buffer(size_t cap);
    void give_buffer(Pinned b);
    void flush_last(bool is_last);
    void dispatch(Pinned input, int mode);
    void worker_main(size_t lane);
    void writer_main();
    void raise_pipeline_error();

    ParConfig cfg_;
    WriteFn writer_;
    std::vector<uint8_t> buffer_;
    size_t batch_bytes_;
    bool finished_ = false;

    // bounded queues (flume::bounded(2N) in the reference)
    std::mutex mu_;
    std::condition_variable cv_work_, cv_order_, cv_space_;
    std::deque<std::unique_ptr<Job>> work_q_;
    std::deque<std::future<Pinned>> order_q_;
    std::mutex pool_mu_;
    std::vector<Pinned> pool_;         // free buffers
    std::vector<uint8_t *> pinned_;    // every buffer ever allocated (freed by the destructor)
    size_t q_cap_;
    bool closed_ = false;
    bool failed_ = false;
    std::exception_ptr error_;

    std::vector<std::thread> workers_;
    std::thread writer_thread_;
    std::vector<gzpx_ctx *> ctxs_;
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
    std::unique_ptr<ParCompress> from_writer(WriteFn w) const {
        return std::make_unique<ParCompress>(cfg_, std::move(w));
    }
    const ParConfig &config() const { return cfg_; }

  private:
    ParConfig cfg_;
};

// The wrapped `R: Read`: returns the number of bytes read into buf (0 = end of stream), -1 on error.
using ReadFn = std::function<long(uint8_t *buf, size_t cap, std::string *err)>;

struct ParDecompressConfig {
    int format = GZPX_FORMAT_BGZF;
    size_t num_threads = 0;  // accepted for API parity (src/par/decompress.rs:43-50)
    int device = 0;
    size_t batch_bytes = (size_t)64 << 20;  // compressed bytes handed to the GPU per slab
};

// ParDecompress (src/par/decompress.rs:112-352): `Read` over a block-compressed stream.  The
// reader part (header walk: check_header + get_block_size, src/par/decompress.rs:195-209) runs on
// the calling thread over a slab of input, the worker part (decode_block + per-block CRC check,
// :162-186) is one GPU launch over every block of the slab; blocks come out in stream order.
class ParDecompress {
  public:
    ParDecompress(const ParDecompressConfig &cfg, ReadFn reader);
    ~ParDecompress();
    ParDecompress(const ParDecompress &) = delete;
    ParDecompress &operator=(const ParDecompress &) = delete;
    size_t read(uint8_t *buf, size_t n);  // 0 = end of stream
    void finish();                        // src/par/decompress.rs:222-238

  private:
    bool fill();
    ParDecompressConfig cfg_;
    ReadFn reader_;
    gzpx_dctx *ctx_ = nullptr;
    // page-locked staging (grown on demand, never value-initialised): compressed bytes not yet
    // decoded (may end in a partial block) and decoded bytes not yet handed out
    struct Staging {
        uint8_t *p = nullptr;
        size_t cap = 0, len = 0;
    };
    void reserve(Staging &s, size_t cap);
    Staging in_, out_;
    size_t out_pos_ = 0;
    bool eof_ = false;
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
    std::unique_ptr<ParDecompress> from_reader(ReadFn r) const {
        return std::make_unique<ParDecompress>(cfg_, std::move(r));
    }

  private:
    ParDecompressConfig cfg_;
};

}  // namespace gzp
