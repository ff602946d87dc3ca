// gzpx_par.cpp -- ParCompress twin (see gzpx_par.hpp) + its C ABI (gzpx_par_* in include/gzpx.h).
#include "gzpx_par.hpp"

#include <hip/hip_runtime.h>

#include <cstring>

namespace gzp {

GzpError error_from_code(int code, size_t block) {
    const std::string msg = std::string(gzpx_strerror(code));
    switch (code) {
        case GZPX_ERR_BUFFER_SIZE: return GzpError(GzpErrorKind::BufferSize, msg);
        case GZPX_ERR_COMPRESSION_LEVEL: return GzpError(GzpErrorKind::LibDeflaterCompressionLvl, msg);
        case GZPX_ERR_INSUFFICIENT_SPACE: return GzpError(GzpErrorKind::LibDeflaterCompress, msg);
        case GZPX_ERR_BLOCK_SIZE_EXCEEDED:
            return GzpError(GzpErrorKind::BlockSizeExceeded, msg + " (block " + std::to_string(block) + ")");
        case GZPX_ERR_UNSUPPORTED: return GzpError(GzpErrorKind::Unsupported, msg);
        case GZPX_ERR_INVALID_HEADER: return GzpError(GzpErrorKind::InvalidHeader, msg);
        case GZPX_ERR_INVALID_CHECK: return GzpError(GzpErrorKind::InvalidCheck, msg);
        case GZPX_ERR_BAD_DATA: return GzpError(GzpErrorKind::LibDeflaterDecompress, msg);
        default: return GzpError(GzpErrorKind::Device, msg);
    }
}

ParCompress::ParCompress(const ParConfig &cfg, WriteFn writer) : cfg_(cfg), writer_(std::move(writer)) {
    if (cfg_.buffer_size < DICT_SIZE) throw error_from_code(GZPX_ERR_BUFFER_SIZE);
    if (cfg_.num_threads == 0) cfg_.num_threads = 1;
    const size_t lanes = cfg_.num_threads < 2 ? 1 : 2;  // device lanes: one copies while one computes
    if (cfg_.batch_blocks == 0) cfg_.batch_blocks = 1;
    batch_bytes_ = cfg_.batch_blocks * cfg_.buffer_size;
    q_cap_ = 2 * lanes;  // bounded(num_threads * 2), src/par/compress.rs:111-112
    for (size_t i = 0; i < lanes; i++) {
        gzpx_config c;
        gzpx_config_default(&c, cfg_.format);
        c.device = cfg_.device;
        c.level = cfg_.compression_level.level();
        c.compat = cfg_.compat;
        c.buffer_size = cfg_.buffer_size;
        c.max_slab_bytes = batch_bytes_;
        gzpx_ctx *ctx = nullptr;
        const int rc = gzpx_ctx_create(&c, &ctx);  // Bgzf::create_compressor, once per worker
        if (rc != GZPX_OK) {
            for (gzpx_ctx *x : ctxs_) gzpx_ctx_destroy(x);
            ctxs_.clear();
            throw error_from_code(rc);
        }
        ctxs_.push_back(ctx);
    }
    buffer_.reserve(batch_bytes_ + cfg_.buffer_size);
    for (size_t i = 0; i < lanes; i++) workers_.emplace_back([this, i] { worker_main(i); });
    writer_thread_ = std::thread([this] { writer_main(); });
}

ParCompress::~ParCompress() {
    if (!finished_) {
        try {
            finish();
        } catch (...) {
            // Drop cannot report; the reference unwraps here (src/par/compress.rs:398)
        }
    }
    for (gzpx_ctx *x : ctxs_) gzpx_ctx_destroy(x);
    for (uint8_t *p : pinned_) (void)hipHostFree(p);  // (the threads have been joined by finish())
}

void ParCompress::raise_pipeline_error() {
    std::exception_ptr e;
    {
        std::lock_guard<std::mutex> lk(mu_);
        e = error_;
    }
    if (e) std::rethrow_exception(e);
    throw GzpError(GzpErrorKind::ChannelSend, "compression pipeline is closed");
}

ParCompress::Pinned ParCompress::take_buffer(size_t cap) {
    {
        std::lock_guard<std::mutex> lk(pool_mu_);
        for (size_t i = 0; i < pool_.size(); i++) {
            if (pool_[i].cap >= cap) {
                Pinned b = pool_[i];
                pool_.erase(pool_.begin() + (ptrdiff_t)i);
                b.len = 0;
                return b;
            }
        }
    }
    Pinned b;
    b.cap = cap < 4096 ? 4096 : cap;
    if (hipHostMalloc((void **)&b.p, b.cap, hipHostMallocDefault) != hipSuccess)
        throw GzpError(GzpErrorKind::Device, "hipHostMalloc failed");
    {
        std::lock_guard<std::mutex> lk(pool_mu_);
        pinned_.push_back(b.p);
    }
    return b;
}

void ParCompress::give_buffer(Pinned b) {
    if (!b.p) return;
    std::lock_guard<std::mutex> lk(pool_mu_);
    pool_.push_back(b);
}

void ParCompress::dispatch(Pinned input, int mode) {
    auto job = std::make_unique<Job>();
    job->input = input;
    job->mode = mode;
    std::unique_lock<std::mutex> lk(mu_);
    cv_space_.wait(lk, [&] { return failed_ || closed_ || (work_q_.size() < q_cap_ && order_q_.size() < q_cap_); });
    if (failed_ || closed_) {
        lk.unlock();
        give_buffer(input);
        raise_pipeline_error();
    }
    order_q_.push_back(job->result.get_future());  // order token FIRST (src/par/compress.rs:424-440)
    work_q_.push_back(std::move(job));             // then the work item (:441-457)
    cv_order_.notify_one();
    cv_work_.notify_one();
}

size_t ParCompress::write(const uint8_t *buf, size_t n) {
    if (finished_) throw GzpError(GzpErrorKind::ChannelSend, "write after finish");
    // `while buffer.len() > buffer_size` (strict): full blocks leave only while at least one byte
    // stays behind.  Blocks are handed over batch_blocks at a time; the cut points are the same.
    // Slabs are cut straight from the caller's bytes (behind whatever is still buffered), so a large
    // write is copied once, not shuffled through the buffer.
    const size_t bs = cfg_.buffer_size;
    size_t off = 0;  // bytes of buf already handed over or buffered
    while (buffer_.size() + (n - off) > batch_bytes_) {
        const size_t have = buffer_.size() + (n - off);
        size_t blocks = (have - 1) / bs;
        if (blocks > cfg_.batch_blocks) blocks = cfg_.batch_blocks;
        const size_t take = blocks * bs;
        Pinned slab = take_buffer(batch_bytes_);
        const size_t from_buffer = buffer_.size() < take ? buffer_.size() : take;
        memcpy(slab.p, buffer_.data(), from_buffer);
        buffer_.erase(buffer_.begin(), buffer_.begin() + (ptrdiff_t)from_buffer);  // (at most one slab's worth)
        memcpy(slab.p + from_buffer, buf + off, take - from_buffer);
        off += take - from_buffer;
        slab.len = take;
        dispatch(slab, GZPX_SLAB_FULL_BLOCKS);
    }
    buffer_.insert(buffer_.end(), buf + off, buf + n);
    return n;
}

void ParCompress::flush_last(bool is_last) {
    // everything buffered goes out cut at buffer_size; the final piece may be short and -- if the
    // buffer is empty -- is an empty block (src/par/compress.rs:333-341 runs at least once)
    const size_t bs = cfg_.buffer_size;
    size_t pos = 0;
    while (buffer_.size() - pos > batch_bytes_) {
        const size_t take = cfg_.batch_blocks * bs;
        Pinned slab = take_buffer(batch_bytes_);
        memcpy(slab.p, buffer_.data() + pos, take);
        slab.len = take;
        pos += take;
        dispatch(slab, GZPX_SLAB_FULL_BLOCKS);
    }
    Pinned rest = take_buffer(batch_bytes_);
    rest.len = buffer_.size() - pos;
    memcpy(rest.p, buffer_.data() + pos, rest.len);
    buffer_.clear();
    dispatch(rest, is_last ? GZPX_SLAB_LAST : GZPX_SLAB_FLUSH);
}

void ParCompress::flush() {
    if (finished_) throw GzpError(GzpErrorKind::ChannelSend, "flush after finish");
    flush_last(false);
}

void ParCompress::finish() {
    if (finished_) return;
    finished_ = true;
    std::exception_ptr first;
    try {
        flush_last(true);
    } catch (...) {
        first = std::current_exception();
    }
    {
        std::lock_guard<std::mutex> lk(mu_);
        closed_ = true;  // drop(tx_compressor), drop(tx_writer)
    }
    cv_work_.notify_all();
    cv_order_.notify_all();
    cv_space_.notify_all();
    for (auto &t : workers_) t.join();
    if (writer_thread_.joinable()) writer_thread_.join();
    std::exception_ptr e;
    {
        std::lock_guard<std::mutex> lk(mu_);
        e = error_;
    }
    if (e) std::rethrow_exception(e);  // the pipeline's own error wins (Io preserved)
    if (first) std::rethrow_exception(first);
}

void ParCompress::worker_main(size_t lane) {
    gzpx_ctx *ctx = ctxs_[lane];
    for (;;) {
        std::unique_ptr<Job> job;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_work_.wait(lk, [&] { return !work_q_.empty() || closed_ || failed_; });
            if (work_q_.empty()) return;
            job = std::move(work_q_.front());
            work_q_.pop_front();
            cv_space_.notify_all();
        }
        try {
            const size_t n = job->input.len;
            const int mode = job->mode;
            Pinned out = take_buffer(gzpx_slab_bound(ctx, batch_bytes_));
            size_t out_len = 0, nb = 0;
            const int rc = gzpx_compress_slab(ctx, job->input.p, n, mode, out.p, out.cap, &out_len, nullptr, 0, &nb);
            give_buffer(job->input);
            job->input = Pinned();
            if (rc != GZPX_OK) {
                give_buffer(out);
                throw error_from_code(rc, nb);
            }
            out.len = out_len;
            job->result.set_value(out);
        } catch (...) {
            give_buffer(job->input);
            job->input = Pinned();
            job->result.set_exception(std::current_exception());
        }
    }
}

void ParCompress::writer_main() {
    for (;;) {
        std::future<Pinned> fut;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_order_.wait(lk, [&] { return !order_q_.empty() || closed_; });
            if (order_q_.empty()) return;
            fut = std::move(order_q_.front());
            order_q_.pop_front();
            cv_space_.notify_all();
        }
        try {
            Pinned chunk = fut.get();  // blocks until THAT slab is done -> in order
            std::string err;
            bool ok;
            {
                std::lock_guard<std::mutex> lk(mu_);
                ok = !failed_;
            }
            const bool wrote = !ok || writer_(chunk.p, chunk.len, &err);
            give_buffer(chunk);
            if (!wrote) throw GzpError(GzpErrorKind::Io, err.empty() ? "write failed" : err);
        } catch (...) {
            std::lock_guard<std::mutex> lk(mu_);
            if (!failed_) {
                failed_ = true;
                error_ = std::current_exception();
            }
            cv_space_.notify_all();
            cv_work_.notify_all();
        }
    }
}

// ---------------------------------------------------------------- ParDecompress
ParDecompress::ParDecompress(const ParDecompressConfig &cfg, ReadFn reader)
    : cfg_(cfg), reader_(std::move(reader)) {
    const int rc = gzpx_dctx_create(cfg_.device, cfg_.format, &ctx_);  // create_decompressor
    if (rc != GZPX_OK) throw error_from_code(rc);
}

ParDecompress::~ParDecompress() {
    if (ctx_) gzpx_dctx_destroy(ctx_);
    if (in_.p) (void)hipHostFree(in_.p);
    if (out_.p) (void)hipHostFree(out_.p);
}

void ParDecompress::reserve(Staging &s, size_t cap) {
    if (cap <= s.cap) return;
    cap += cap / 4;
    uint8_t *np = nullptr;
    if (hipHostMalloc((void **)&np, cap, hipHostMallocDefault) != hipSuccess)
        throw GzpError(GzpErrorKind::Device, "hipHostMalloc failed");
    if (s.len) memcpy(np, s.p, s.len);
    if (s.p) (void)hipHostFree(s.p);
    s.p = np;
    s.cap = cap;
}

bool ParDecompress::fill() {
    const size_t hdr = cfg_.format == GZPX_FORMAT_BGZF ? 18 : 20;
    for (;;) {
        // top the slab up from the reader
        reserve(in_, cfg_.batch_bytes + (1u << 20));
        while (!eof_ && in_.len < cfg_.batch_bytes) {
            const size_t want = in_.cap - in_.len;
            std::string err;
            const long got = reader_(in_.p + in_.len, want, &err);
            if (got < 0) throw GzpError(GzpErrorKind::Io, err.empty() ? "read failed" : err);
            in_.len += (size_t)got;
            if (got == 0) eof_ = true;
        }
        size_t nb = 0, used = 0;
        int rc = gzpx_scan_blocks(cfg_.format, in_.p, in_.len, nullptr, nullptr, 0, &nb, &used);
        if (rc != GZPX_OK) throw error_from_code(rc);
        if (nb == 0) {
            if (!eof_) {  // one block larger than the slab: keep reading
                cfg_.batch_bytes *= 2;
                continue;
            }
            // EOF: a failed header read ends the stream silently (src/par/decompress.rs:207-209);
            // a header followed by a short body is read_exact's UnexpectedEof (:201-202)
            if (in_.len >= hdr) throw GzpError(GzpErrorKind::Io, "failed to fill whole buffer");
            return false;
        }
        std::vector<uint64_t> offs(nb);
        std::vector<uint32_t> sizes(nb);
        rc = gzpx_scan_blocks(cfg_.format, in_.p, in_.len, offs.data(), sizes.data(), nb, &nb, &used);
        if (rc != GZPX_OK) throw error_from_code(rc);
        size_t total = 0;
        for (size_t b = 0; b < nb; b++) {
            const uint8_t *f = in_.p + offs[b] + sizes[b] - 4;  // ISIZE
            total += (size_t)f[0] | ((size_t)f[1] << 8) | ((size_t)f[2] << 16) | ((size_t)f[3] << 24);
        }
        out_.len = 0;
        reserve(out_, total ? total : 1);
        size_t got = 0;
        gzpx_check_info info = {0, 0, 0};
        rc = gzpx_decompress_blocks(ctx_, in_.p, used, offs.data(), sizes.data(), nb, out_.p, total, &got, &info);
        if (rc == GZPX_ERR_INVALID_CHECK)
            throw GzpError(GzpErrorKind::InvalidCheck, "Invalid check value: found " + std::to_string(info.found) +
                                                           ", expected " + std::to_string(info.expected));
        if (rc != GZPX_OK) throw error_from_code(rc, info.block);
        out_.len = got;
        out_pos_ = 0;
        memmove(in_.p, in_.p + used, in_.len - used);  // the partial block at the end, if any
        in_.len -= used;
        if (got) return true;  // slabs of empty blocks only (e.g. the EOF marker): look further
        if (eof_ && in_.len == 0) return false;
    }
}

size_t ParDecompress::read(uint8_t *buf, size_t n) {
    if (out_pos_ == out_.len) {
        out_.len = 0;
        out_pos_ = 0;
        if (!fill()) return 0;
    }
    const size_t take = out_.len - out_pos_ < n ? out_.len - out_pos_ : n;
    memcpy(buf, out_.p + out_pos_, take);
    out_pos_ += take;
    return take;
}

void ParDecompress::finish() {}

}  // namespace gzp

// ---------------------------------------------------------------- C ABI of the twin
struct gzpx_par {
    std::unique_ptr<gzp::ParCompress> pc;
    std::string last_error;
    int last_kind = 0;
};

static int kind_to_code(gzp::GzpErrorKind k) {
    using K = gzp::GzpErrorKind;
    switch (k) {
        case K::BufferSize: return GZPX_ERR_BUFFER_SIZE;
        case K::NumThreads: return GZPX_ERR_NUM_THREADS;
        case K::LibDeflaterCompressionLvl: return GZPX_ERR_COMPRESSION_LEVEL;
        case K::LibDeflaterCompress: return GZPX_ERR_INSUFFICIENT_SPACE;
        case K::BlockSizeExceeded: return GZPX_ERR_BLOCK_SIZE_EXCEEDED;
        case K::Io: return GZPX_ERR_IO;
        case K::ChannelSend:
        case K::ChannelReceive: return GZPX_ERR_CHANNEL;
        case K::Unsupported: return GZPX_ERR_UNSUPPORTED;
        case K::InvalidHeader: return GZPX_ERR_INVALID_HEADER;
        case K::InvalidCheck: return GZPX_ERR_INVALID_CHECK;
        case K::LibDeflaterDecompress: return GZPX_ERR_BAD_DATA;
        default: return GZPX_ERR_DEVICE;
    }
}

template <class Fn>
static int guarded(gzpx_par *p, Fn fn) {
    try {
        fn();
        return GZPX_OK;
    } catch (const gzp::GzpError &e) {
        if (p) p->last_error = e.what();
        return kind_to_code(e.kind);
    } catch (const std::exception &e) {
        if (p) p->last_error = e.what();
        return GZPX_ERR_DEVICE;
    }
}

struct gzpx_pard {
    std::unique_ptr<gzp::ParDecompress> pd;
    std::string last_error;
};

template <class Fn>
static int guarded_d(gzpx_pard *p, Fn fn) {
    try {
        fn();
        return GZPX_OK;
    } catch (const gzp::GzpError &e) {
        if (p) p->last_error = e.what();
        return kind_to_code(e.kind);
    } catch (const std::exception &e) {
        if (p) p->last_error = e.what();
        return GZPX_ERR_DEVICE;
    }
}

extern "C" {

int gzpx_par_create(const gzpx_par_config *cfg, gzpx_write_fn write_fn, void *user, gzpx_par **out) {
    if (!cfg || !write_fn || !out) return GZPX_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->num_threads == 0) return GZPX_ERR_NUM_THREADS;  // src/par/compress.rs:84-90
    auto *p = new gzpx_par();
    const int rc = guarded(p, [&] {
        gzp::ParConfig c;
        c.format = cfg->format;
        c.buffer_size = cfg->buffer_size;
        c.num_threads = cfg->num_threads;
        c.compression_level = gzp::Compression(cfg->level);
        c.device = cfg->device;
        c.compat = cfg->compat;
        c.batch_blocks = cfg->batch_blocks ? cfg->batch_blocks : 1024;
        p->pc = std::make_unique<gzp::ParCompress>(
            c, [write_fn, user](const uint8_t *d, size_t n, std::string *err) {
                const int r = write_fn(user, d, n);
                if (r != 0 && err) *err = "writer callback returned " + std::to_string(r);
                return r == 0;
            });
    });
    if (rc != GZPX_OK) {
        delete p;
        return rc;
    }
    *out = p;
    return GZPX_OK;
}

int gzpx_par_write(gzpx_par *p, const uint8_t *buf, size_t n) {
    if (!p || (!buf && n)) return GZPX_ERR_INVALID_ARG;
    return guarded(p, [&] { p->pc->write(buf, n); });
}

int gzpx_par_flush(gzpx_par *p) {
    if (!p) return GZPX_ERR_INVALID_ARG;
    return guarded(p, [&] { p->pc->flush(); });
}

int gzpx_par_finish(gzpx_par *p) {
    if (!p) return GZPX_ERR_INVALID_ARG;
    return guarded(p, [&] { p->pc->finish(); });
}

void gzpx_par_destroy(gzpx_par *p) {
    if (!p) return;
    p->pc.reset();  // Drop: finishes if needed
    delete p;
}

const char *gzpx_par_last_error(const gzpx_par *p) { return p ? p->last_error.c_str() : ""; }

int gzpx_pard_create(int format, int device, size_t batch_bytes, gzpx_read_fn read_fn, void *user,
                     gzpx_pard **out) {
    if (!read_fn || !out) return GZPX_ERR_INVALID_ARG;
    *out = nullptr;
    auto *p = new gzpx_pard();
    const int rc = guarded_d(p, [&] {
        gzp::ParDecompressConfig c;
        c.format = format;
        c.device = device;
        if (batch_bytes) c.batch_bytes = batch_bytes;
        p->pd = std::make_unique<gzp::ParDecompress>(c, [read_fn, user](uint8_t *b, size_t cap, std::string *err) {
            const long r = read_fn(user, b, cap);
            if (r < 0 && err) *err = "reader callback returned " + std::to_string(r);
            return r;
        });
    });
    if (rc != GZPX_OK) {
        delete p;
        return rc;
    }
    *out = p;
    return GZPX_OK;
}

int gzpx_pard_read(gzpx_pard *p, uint8_t *buf, size_t n, size_t *got) {
    if (!p || (!buf && n) || !got) return GZPX_ERR_INVALID_ARG;
    *got = 0;
    return guarded_d(p, [&] { *got = p->pd->read(buf, n); });
}

void gzpx_pard_destroy(gzpx_pard *p) { delete p; }
const char *gzpx_pard_last_error(const gzpx_pard *p) { return p ? p->last_error.c_str() : ""; }

}  // extern "C"
