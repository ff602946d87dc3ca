// gzpx_par.cpp -- ParCompress / ParDecompress twins (see gzpx_par.hpp) + their C ABI (gzpx_par_* /
// gzpx_pard_* in include/gzpx.h).
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include "gzpx_par.hpp"

#include <hip/hip_runtime.h>

#include <pthread.h>
#include <sched.h>

#include <cstring>

namespace gzp {

// pin_threads(Some(n)) in the reference pins worker i to the (n + i)-th core the process may run on
// (core_affinity::get_core_ids, src/par/compress.rs:260-276); the twin's workers are the device
// thread (i = 0) and the copy helpers (i = 1 ..).  A core that does not exist is skipped silently,
// as there.
static void pin_current_thread(const std::optional<size_t> &pin_at, size_t i) {
    if (!pin_at) return;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    size_t want = *pin_at + i, seen = 0;
    for (int c = 0; c < CPU_SETSIZE; c++) {
        if (!CPU_ISSET(c, &allowed)) continue;
        if (seen++ == want) {
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(c, &one);
            (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
            return;
        }
    }
}

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

// ---------------------------------------------------------------- CopyPool
CopyPool::CopyPool(size_t helpers, std::optional<size_t> pin_at) {
    for (size_t i = 0; i < helpers; i++)
        threads_.emplace_back([this, pin_at, i] {
            pin_current_thread(pin_at, 1 + i);
            main();
        });
}

// Copy into a staging slab: the bytes are read next by the copy engine, never again by this CPU, so the
// bulk goes out with streaming (non-temporal) stores -- no read-for-ownership of the destination
// lines, no cache pollution.  (glibc's memcpy only does that for copies of several MiB; the write()
// calls of the reference's own benchmark are 64 KiB.)
static void stage_copy(uint8_t *dst, const uint8_t *src, size_t n) {
#if defined(__SSE2__)
    if (n >= 4096) {
        const size_t head = (size_t)(-(uintptr_t)dst & 63u);  // up to the destination's next cache line
        memcpy(dst, src, head);
        dst += head;
        src += head;
        n -= head;
        const size_t body = n & ~(size_t)63;
        for (size_t i = 0; i < body; i += 64) {
            const __m128i a = _mm_loadu_si128((const __m128i *)(src + i));
            const __m128i b = _mm_loadu_si128((const __m128i *)(src + i + 16));
            const __m128i c = _mm_loadu_si128((const __m128i *)(src + i + 32));
            const __m128i d = _mm_loadu_si128((const __m128i *)(src + i + 48));
            _mm_stream_si128((__m128i *)(dst + i), a);
            _mm_stream_si128((__m128i *)(dst + i + 16), b);
            _mm_stream_si128((__m128i *)(dst + i + 32), c);
            _mm_stream_si128((__m128i *)(dst + i + 48), d);
        }
        _mm_sfence();  // the stores are globally visible before the slab is handed to the device thread
        memcpy(dst + body, src + body, n - body);
        return;
    }
#endif
    memcpy(dst, src, n);
}

CopyPool::~CopyPool() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    cv_task_.notify_all();
    for (auto &t : threads_) t.join();
}

void CopyPool::main() {
    for (;;) {
        Task t;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_task_.wait(lk, [&] { return stop_ || !tasks_.empty(); });
            if (tasks_.empty()) return;
            t = tasks_.front();
            tasks_.pop_front();
        }
        stage_copy(t.dst, t.src, t.n);
        {
            std::lock_guard<std::mutex> lk(mu_);
            pending_--;
        }
        cv_done_.notify_all();
    }
}

void CopyPool::copy(uint8_t *dst, const uint8_t *src, size_t n) {
    const size_t parts = threads_.size() + 1;
    if (n < ((size_t)1 << 20) || parts == 1) {
        stage_copy(dst, src, n);
        return;
    }
    const size_t piece = ((n / parts) + 4095) & ~(size_t)4095;
    size_t off = piece < n ? piece : n;  // the caller's own share is the first piece
    {
        std::lock_guard<std::mutex> lk(mu_);
        for (size_t o = off; o < n; o += piece) {
            tasks_.push_back(Task{dst + o, src + o, n - o < piece ? n - o : piece});
            pending_++;
        }
    }
    cv_task_.notify_all();
    stage_copy(dst, src, off);
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return pending_ == 0; });
}

// ---------------------------------------------------------------- ParCompress
ParCompress::ParCompress(const ParConfig &cfg, WriteFn writer) : cfg_(cfg), writer_(std::move(writer)) {
    if (cfg_.buffer_size < DICT_SIZE) throw error_from_code(GZPX_ERR_BUFFER_SIZE);
    if (cfg_.num_threads == 0) cfg_.num_threads = 1;
    // a slab is batch_blocks blocks, but never more than kMaxSlabBytes of page-locked memory (1 MiB
    // or 16 MiB Mgzip blocks would otherwise ask for gigabytes per slab) and never less than one block
    const size_t by_budget = kMaxSlabBytes / cfg_.buffer_size;
    batch_blocks_ = cfg_.batch_blocks ? cfg_.batch_blocks : 1;
    if (batch_blocks_ > by_budget) batch_blocks_ = by_budget;
    if (batch_blocks_ == 0) batch_blocks_ = 1;
    batch_bytes_ = batch_blocks_ * cfg_.buffer_size;
    q_cap_ = 2;  // bounded(num_threads * 2) in the reference (src/par/compress.rs:111-112); here: slabs
    gzpx_config c;
    gzpx_config_default(&c, cfg_.format);
    c.device = cfg_.device;
    c.level = cfg_.compression_level.level();
    c.compat = cfg_.compat;
    c.buffer_size = cfg_.buffer_size;
    c.max_slab_bytes = batch_bytes_;
    const int rc = gzpx_ctx_create(&c, &ctx_);  // Bgzf::create_compressor
    if (rc != GZPX_OK) throw error_from_code(rc);
    (void)hipSetDevice(cfg_.device);
    try {
        // The whole staging pool is page-locked now (the analogue of the reference spinning up its
        // thread pool in from_writer), not slab by slab inside the first writes: locking a 64 MiB
        // slab costs as much as compressing three of them.  Inputs: the slab being filled, the
        // queued ones, the ones in flight; outputs: in flight + waiting for the writer.
        const size_t in_cap = batch_bytes_ + cfg_.buffer_size, out_cap = gzpx_slab_bound(ctx_, batch_bytes_);
        buf_cap_ = in_cap > out_cap ? in_cap : out_cap;  // one size: any pool buffer serves either way
        std::vector<Pinned> warm;
        for (size_t i = 0; i < 1 + 2 * (q_cap_ + GZPX_SLOTS); i++) warm.push_back(take_buffer(buf_cap_));
        for (const Pinned &b : warm) give_buffer(b);
        fill_ = take_buffer(buf_cap_);
    } catch (...) {
        for (uint8_t *p : pinned_) (void)hipHostFree(p);
        gzpx_ctx_destroy(ctx_);
        throw;
    }
    const size_t helpers = cfg_.num_threads > 1 ? (cfg_.num_threads - 1 < 3 ? cfg_.num_threads - 1 : 3) : 0;
    copier_ = std::make_unique<CopyPool>(helpers, cfg_.pin_threads);
    device_thread_ = std::thread([this] { device_main(); });
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
    copier_.reset();
    if (ctx_) gzpx_ctx_destroy(ctx_);
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
    // (a slab stays in the order queue while it is in flight: that queue is GZPX_SLOTS deeper)
    cv_space_.wait(lk, [&] {
        return failed_ || closed_ || (work_q_.size() < q_cap_ && order_q_.size() < q_cap_ + GZPX_SLOTS);
    });
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

// `while buffer.len() > buffer_size` (strict, src/par/compress.rs:415): full blocks leave only while
// at least one byte stays behind.  Blocks are handed over batch_blocks at a time -- the cut points
// are the same multiples of buffer_size -- and what stays behind (at most one block) moves to the
// front of the next slab.
void ParCompress::after_append() {
    while (fill_.len > batch_bytes_) {
        Pinned next = take_buffer(buf_cap_);
        next.len = fill_.len - batch_bytes_;
        memcpy(next.p, fill_.p + batch_bytes_, next.len);
        Pinned full = fill_;
        full.len = batch_bytes_;
        fill_ = next;
        dispatch(full, GZPX_SLAB_FULL_BLOCKS);
    }
}

size_t ParCompress::write(const uint8_t *buf, size_t n) {
    if (finished_) throw GzpError(GzpErrorKind::ChannelSend, "write after finish");
    size_t off = 0;
    while (off < n) {
        const size_t room = fill_room();  // >= buffer_size: after_append keeps len <= batch_bytes_
        const size_t take = n - off < room ? n - off : room;
        copier_->copy(fill_.p + fill_.len, buf + off, take);
        fill_.len += take;
        off += take;
        after_append();
    }
    return n;
}

std::pair<uint8_t *, size_t> ParCompress::reserve() {
    if (finished_) throw GzpError(GzpErrorKind::ChannelSend, "reserve after finish");
    return {fill_.p + fill_.len, fill_room()};
}

void ParCompress::commit(size_t n) {
    if (finished_) throw GzpError(GzpErrorKind::ChannelSend, "commit after finish");
    if (n > fill_room()) throw GzpError(GzpErrorKind::LibDeflaterCompress, "commit past the reserved room");
    fill_.len += n;
    after_append();
}

void ParCompress::flush_last(bool is_last) {
    // everything buffered goes out cut at buffer_size; the final piece may be short and -- if the
    // buffer is empty -- is an empty block (src/par/compress.rs:333-341 runs at least once)
    Pinned rest = fill_;
    fill_ = Pinned();
    if (!is_last) fill_ = take_buffer(buf_cap_);
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
    if (device_thread_.joinable()) device_thread_.join();
    if (writer_thread_.joinable()) writer_thread_.join();
    std::exception_ptr e;
    {
        std::lock_guard<std::mutex> lk(mu_);
        e = error_;
    }
    if (e) std::rethrow_exception(e);  // the pipeline's own error wins (Io preserved)
    if (first) std::rethrow_exception(first);
}

void ParCompress::complete(InFlight &f) {
    const size_t n = f.job->input.len, bs = cfg_.buffer_size;
    const size_t nb = n == 0 ? 1 : (n + bs - 1) / bs;
    bool waited = false;
    try {
        Done d;
        d.block_sizes.resize(nb);
        d.in_len = n;
        size_t out_len = 0, blk = 0;
        waited = true;
        const int rc = gzpx_compress_slab_wait(ctx_, f.ticket, &out_len, d.block_sizes.data(), nb, &blk);
        give_buffer(f.job->input);
        f.job->input = Pinned();
        if (rc != GZPX_OK) {
            give_buffer(f.out);
            throw error_from_code(rc, blk);
        }
        f.out.len = out_len;
        d.out = f.out;
        f.job->result.set_value(std::move(d));
    } catch (...) {
        // (an exception before the wait -- bad_alloc -- must not leave the ticket's slot taken: every
        // later submit would find the context busy)
        if (!waited) (void)gzpx_compress_slab_wait(ctx_, f.ticket, nullptr, nullptr, 0, nullptr);
        f.job->result.set_exception(std::current_exception());
    }
}

// The device thread: the N compressor threads of the reference (src/par/compress.rs:267-296) folded
// into one submitter that keeps up to GZPX_SLOTS slabs in flight.  With nothing queued it completes
// the oldest slab instead of idling, so a slow producer still sees its blocks written promptly.
void ParCompress::device_main() {
    pin_current_thread(cfg_.pin_threads, 0);
    (void)hipSetDevice(cfg_.device);
    std::deque<InFlight> fl;
    for (;;) {
        std::unique_ptr<Job> job;
        {
            std::unique_lock<std::mutex> lk(mu_);
            if (fl.empty()) cv_work_.wait(lk, [&] { return !work_q_.empty() || closed_ || failed_; });
            if (!work_q_.empty()) {
                job = std::move(work_q_.front());
                work_q_.pop_front();
                cv_space_.notify_all();
            } else if (fl.empty()) {
                return;  // closed (or failed) and drained
            }
        }
        if (!job) {
            complete(fl.front());
            fl.pop_front();
            continue;
        }
        if (fl.size() == GZPX_SLOTS) {
            complete(fl.front());
            fl.pop_front();
        }
        InFlight f;
        try {
            f.out = take_buffer(buf_cap_);
            const int rc = gzpx_compress_slab_submit(ctx_, job->input.p, job->input.len, job->mode, f.out.p,
                                                     f.out.cap, &f.ticket);
            if (rc != GZPX_OK) {
                give_buffer(f.out);
                throw error_from_code(rc);
            }
            f.job = std::move(job);
            fl.push_back(std::move(f));
        } catch (...) {
            give_buffer(job->input);
            job->input = Pinned();
            job->result.set_exception(std::current_exception());
        }
    }
}

void ParCompress::writer_main() {
    for (;;) {
        std::future<Done> fut;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_order_.wait(lk, [&] { return !order_q_.empty() || closed_; });
            if (order_q_.empty()) return;
            fut = std::move(order_q_.front());
            order_q_.pop_front();
            cv_space_.notify_all();
        }
        try {
            Done d = fut.get();  // blocks until THAT slab is done -> in order
            std::string err;
            bool ok;
            {
                std::lock_guard<std::mutex> lk(mu_);
                ok = !failed_;
            }
            const bool wrote = !ok || writer_(d.out.p, d.out.len, &err);
            give_buffer(d.out);
            if (!wrote) throw GzpError(GzpErrorKind::Io, err.empty() ? "write failed" : err);
            if (ok) {  // the index side-product: where every block of the slab starts
                std::lock_guard<std::mutex> lk(index_mu_);
                const size_t nb = d.block_sizes.size(), bs = cfg_.buffer_size;
                for (size_t i = 0; i < nb; i++) {
                    index_.push_back(IndexEntry{coff_, uoff_});
                    coff_ += d.block_sizes[i];
                    uoff_ += i + 1 < nb ? bs : d.in_len - (nb - 1) * bs;
                }
            }
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

std::vector<IndexEntry> ParCompress::index() const {
    std::lock_guard<std::mutex> lk(index_mu_);
    return index_;
}

// ---------------------------------------------------------------- ParDecompress
ParDecompress::ParDecompress(const ParDecompressConfig &cfg, ReadFn reader)
    : cfg_(cfg), reader_(std::move(reader)) {
    const int rc = gzpx_dctx_create(cfg_.device, cfg_.format, &ctx_);  // create_decompressor
    if (rc != GZPX_OK) throw error_from_code(rc);
    reader_thread_ = std::thread([this] { reader_main(); });
    device_thread_ = std::thread([this] { device_main(); });
}

void ParDecompress::stop_threads() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    cv_in_.notify_all();
    cv_out_.notify_all();
    cv_space_.notify_all();
    if (reader_thread_.joinable()) reader_thread_.join();
    if (device_thread_.joinable()) device_thread_.join();
}

ParDecompress::~ParDecompress() {
    stop_threads();
    if (ctx_) gzpx_dctx_destroy(ctx_);
    for (uint8_t *p : pinned_) (void)hipHostFree(p);
}

void ParDecompress::finish() { stop_threads(); }

// grow a page-locked staging buffer, keeping its first `keep` bytes
void ParDecompress::reserve(Staging &s, size_t cap, size_t keep) {
    if (cap <= s.cap) return;
    cap += cap / 4;
    uint8_t *np = nullptr;
    if (hipHostMalloc((void **)&np, cap, hipHostMallocDefault) != hipSuccess)
        throw GzpError(GzpErrorKind::Device, "hipHostMalloc failed");
    if (keep) memcpy(np, s.p, keep);
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (s.p) {
            for (auto &q : pinned_)
                if (q == s.p) q = np;
            (void)hipHostFree(s.p);
        } else {
            pinned_.push_back(np);
        }
    }
    s.p = np;
    s.cap = cap;
}

bool ParDecompress::push(std::deque<SlabPtr> &q, SlabPtr s, std::condition_variable &cv) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_space_.wait(lk, [&] { return stop_ || q.size() < q_cap_; });
    if (stop_) return false;
    q.push_back(std::move(s));
    cv.notify_all();
    return true;
}

ParDecompress::SlabPtr ParDecompress::pop(std::deque<SlabPtr> &q, std::condition_variable &cv, bool wait) {
    std::unique_lock<std::mutex> lk(mu_);
    if (wait) cv.wait(lk, [&] { return stop_ || !q.empty(); });
    if (q.empty()) return nullptr;
    SlabPtr s = std::move(q.front());
    q.pop_front();
    cv_space_.notify_all();
    return s;
}

ParDecompress::SlabPtr ParDecompress::recycle() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (!free_.empty()) {
            SlabPtr s = std::move(free_.front());
            free_.pop_front();
            s->in.len = s->out.len = 0;
            s->used = s->total = 0;
            s->error = nullptr;
            s->end = false;
            return s;
        }
    }
    return std::make_unique<Slab>();
}

// The reader thread of ParDecompress::run (src/par/decompress.rs:188-219): read ahead, walk the block
// headers, hand whole blocks on.
void ParDecompress::reader_main() {
    (void)hipSetDevice(cfg_.device);
    const size_t hdr = cfg_.format == GZPX_FORMAT_BGZF ? 18 : 20;
    std::vector<uint8_t> carry;  // the partial block at the end of the previous slab
    bool eof = false;
    for (;;) {
        SlabPtr s = recycle();
        try {
            reserve(s->in, cfg_.batch_bytes + ((size_t)1 << 20) + carry.size(), 0);
            memcpy(s->in.p, carry.data(), carry.size());
            s->in.len = carry.size();
            carry.clear();
            size_t nb = 0, used = 0;
            for (;;) {
                while (!eof && s->in.len < cfg_.batch_bytes) {  // top the slab up from the reader
                    {
                        std::lock_guard<std::mutex> lk(mu_);
                        if (stop_) return;
                    }
                    std::string err;
                    const long got = reader_(s->in.p + s->in.len, s->in.cap - s->in.len, &err);
                    if (got < 0) throw GzpError(GzpErrorKind::Io, err.empty() ? "read failed" : err);
                    s->in.len += (size_t)got;
                    if (got == 0) eof = true;
                }
                const int rc = gzpx_scan_blocks(cfg_.format, s->in.p, s->in.len, nullptr, nullptr, 0, &nb, &used);
                if (rc != GZPX_OK) throw error_from_code(rc);
                if (nb || eof) break;
                cfg_.batch_bytes *= 2;  // one block larger than the slab: keep reading
                reserve(s->in, cfg_.batch_bytes + ((size_t)1 << 20), s->in.len);
            }
            if (nb == 0) {
                // EOF: a failed header read ends the stream silently (src/par/decompress.rs:207-209);
                // a header followed by a short body is read_exact's UnexpectedEof (:201-202)
                if (s->in.len >= hdr) throw GzpError(GzpErrorKind::Io, "failed to fill whole buffer");
                s->end = true;
                (void)push(in_q_, std::move(s), cv_in_);
                return;
            }
            s->offs.resize(nb);
            s->sizes.resize(nb);
            const int rc = gzpx_scan_blocks(cfg_.format, s->in.p, s->in.len, s->offs.data(), s->sizes.data(), nb, &nb,
                                            &used);
            if (rc != GZPX_OK) throw error_from_code(rc);
            size_t total = 0;
            for (size_t b = 0; b < nb; b++) {
                const uint8_t *f = s->in.p + s->offs[b] + s->sizes[b] - 4;  // ISIZE (get_footer_values)
                total += (size_t)f[0] | ((size_t)f[1] << 8) | ((size_t)f[2] << 16) | ((size_t)f[3] << 24);
            }
            s->used = used;
            s->total = total;
            carry.assign(s->in.p + used, s->in.p + s->in.len);
            if (!push(in_q_, std::move(s), cv_in_)) return;
        } catch (...) {
            s->error = std::current_exception();
            (void)push(in_q_, std::move(s), cv_in_);
            return;
        }
    }
}

// The inflate workers (src/par/decompress.rs:162-186) as one device thread with GZPX_SLOTS slabs in
// flight; results leave in submission order.
void ParDecompress::device_main() {
    (void)hipSetDevice(cfg_.device);
    std::deque<SlabPtr> fl;
    auto complete_oldest = [&]() -> bool {  // false: the stream ends here (error or shutdown)
        SlabPtr s = std::move(fl.front());
        fl.pop_front();
        size_t got = 0;
        gzpx_check_info info = {0, 0, 0};
        const int rc = gzpx_decompress_blocks_wait(ctx_, s->ticket, &got, &info);
        if (rc == GZPX_ERR_INVALID_CHECK)
            s->error = std::make_exception_ptr(
                GzpError(GzpErrorKind::InvalidCheck, "Invalid check value: found " + std::to_string(info.found) +
                                                         ", expected " + std::to_string(info.expected)));
        else if (rc != GZPX_OK)
            s->error = std::make_exception_ptr(error_from_code(rc, info.block));
        s->out.len = got;
        const bool bad = (bool)s->error;
        if (!push(out_q_, std::move(s), cv_out_)) return false;
        return !bad;
    };
    for (;;) {
        SlabPtr s = pop(in_q_, cv_in_, fl.empty());
        if (!s) {
            if (fl.empty()) return;  // shutdown
            if (!complete_oldest()) break;
            continue;
        }
        if (s->error || s->end) {  // everything before it first, then the marker itself
            bool ok = true;
            while (ok && !fl.empty()) ok = complete_oldest();
            if (ok) (void)push(out_q_, std::move(s), cv_out_);
            break;
        }
        if (fl.size() == GZPX_SLOTS && !complete_oldest()) break;
        try {
            reserve(s->out, s->total ? s->total : 1, 0);
            const int rc = gzpx_decompress_blocks_submit(ctx_, s->in.p, s->used, s->offs.data(), s->sizes.data(),
                                                         s->offs.size(), s->out.p, s->total, &s->ticket);
            if (rc != GZPX_OK) throw error_from_code(rc);
            fl.push_back(std::move(s));
        } catch (...) {
            s->error = std::current_exception();
            bool ok = true;
            while (ok && !fl.empty()) ok = complete_oldest();
            if (ok) (void)push(out_q_, std::move(s), cv_out_);
            break;
        }
    }
    // slabs still in flight hold device work that references their buffers: wait for it
    while (!fl.empty()) {
        size_t got = 0;
        (void)gzpx_decompress_blocks_wait(ctx_, fl.front()->ticket, &got, nullptr);
        fl.pop_front();
    }
}

const uint8_t *ParDecompress::fill_buf(size_t *len) {
    *len = 0;
    if (sticky_) std::rethrow_exception(sticky_);
    while (!cur_ || out_pos_ == cur_->out.len) {
        if (cur_) {
            std::lock_guard<std::mutex> lk(mu_);
            free_.push_back(std::move(cur_));
            cur_ = nullptr;
        }
        if (done_) return nullptr;
        SlabPtr s = pop(out_q_, cv_out_, true);
        if (!s || s->end) {
            done_ = true;
            return nullptr;
        }
        if (s->error) {
            sticky_ = s->error;
            done_ = true;
            stop_threads();
            std::rethrow_exception(sticky_);
        }
        cur_ = std::move(s);  // (a slab of empty blocks only -- e.g. the EOF marker -- is skipped by the loop)
        out_pos_ = 0;
    }
    *len = cur_->out.len - out_pos_;
    return cur_->out.p + out_pos_;
}

void ParDecompress::consume(size_t n) {
    if (cur_) out_pos_ += n < cur_->out.len - out_pos_ ? n : cur_->out.len - out_pos_;
}

size_t ParDecompress::read(uint8_t *buf, size_t n) {
    size_t have = 0;
    const uint8_t *p = fill_buf(&have);
    if (!p || !have) return 0;
    const size_t take = have < n ? have : n;
    memcpy(buf, p, take);
    consume(take);
    return take;
}

}  // namespace gzp

// ---------------------------------------------------------------- C ABI of the twins
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

static int par_create(const gzpx_par_config *cfg, std::optional<size_t> pin, gzpx_write_fn write_fn, void *user,
                      gzpx_par **out);

int gzpx_par_create(const gzpx_par_config *cfg, gzpx_write_fn write_fn, void *user, gzpx_par **out) {
    return par_create(cfg, std::nullopt, write_fn, user, out);
}

int gzpx_par_create_pinned(const gzpx_par_config *cfg, size_t first_core, gzpx_write_fn write_fn, void *user,
                           gzpx_par **out) {
    return par_create(cfg, first_core, write_fn, user, out);
}

static int par_create(const gzpx_par_config *cfg, std::optional<size_t> pin, gzpx_write_fn write_fn, void *user,
                      gzpx_par **out) {
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
        c.pin_threads = pin;
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

int gzpx_par_write_chunked(gzpx_par *p, const uint8_t *buf, size_t n, size_t chunk) {
    if (!p || (!buf && n) || chunk == 0) return GZPX_ERR_INVALID_ARG;
    return guarded(p, [&] {
        for (size_t lo = 0; lo < n; lo += chunk) p->pc->write(buf + lo, n - lo < chunk ? n - lo : chunk);
    });
}

int gzpx_par_reserve(gzpx_par *p, uint8_t **ptr, size_t *cap) {
    if (!p || !ptr || !cap) return GZPX_ERR_INVALID_ARG;
    return guarded(p, [&] {
        const auto r = p->pc->reserve();
        *ptr = r.first;
        *cap = r.second;
    });
}

int gzpx_par_commit(gzpx_par *p, size_t n) {
    if (!p) return GZPX_ERR_INVALID_ARG;
    return guarded(p, [&] { p->pc->commit(n); });
}

int gzpx_par_flush(gzpx_par *p) {
    if (!p) return GZPX_ERR_INVALID_ARG;
    return guarded(p, [&] { p->pc->flush(); });
}

int gzpx_par_finish(gzpx_par *p) {
    if (!p) return GZPX_ERR_INVALID_ARG;
    return guarded(p, [&] { p->pc->finish(); });
}

int gzpx_par_index(gzpx_par *p, gzpx_index_entry *entries, size_t max_entries, size_t *n_entries) {
    if (!p || !n_entries) return GZPX_ERR_INVALID_ARG;
    return guarded(p, [&] {
        const std::vector<gzp::IndexEntry> idx = p->pc->index();
        *n_entries = idx.size();
        if (entries)
            for (size_t i = 0; i < idx.size() && i < max_entries; i++)
                entries[i] = gzpx_index_entry{idx[i].compressed_offset, idx[i].uncompressed_offset};
    });
}

void gzpx_par_destroy(gzpx_par *p) {
    if (!p) return;
    p->pc.reset();  // Drop: finishes if needed
    delete p;
}

const char *gzpx_par_last_error(const gzpx_par *p) { return p ? p->last_error.c_str() : ""; }

size_t gzpx_gzi_size(size_t n_entries) { return 8 + 16 * (n_entries ? n_entries - 1 : 0); }

int gzpx_gzi_write(const gzpx_index_entry *entries, size_t n_entries, uint8_t *out, size_t out_cap,
                   size_t *out_len) {
    if ((!entries && n_entries) || !out || !out_len) return GZPX_ERR_INVALID_ARG;
    const size_t need = gzpx_gzi_size(n_entries);
    if (need > out_cap) return GZPX_ERR_INSUFFICIENT_SPACE;
    auto put64 = [](uint8_t *q, uint64_t v) {
        for (int k = 0; k < 8; k++) q[k] = (uint8_t)(v >> (8 * k));
    };
    put64(out, n_entries ? n_entries - 1 : 0);  // the first block (0, 0) is implicit
    for (size_t i = 1; i < n_entries; i++) {
        put64(out + 8 + 16 * (i - 1), entries[i].compressed_offset);
        put64(out + 16 + 16 * (i - 1), entries[i].uncompressed_offset);
    }
    *out_len = need;
    return GZPX_OK;
}

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

int gzpx_pard_fill_buf(gzpx_pard *p, const uint8_t **ptr, size_t *len) {
    if (!p || !ptr || !len) return GZPX_ERR_INVALID_ARG;
    *ptr = nullptr;
    *len = 0;
    return guarded_d(p, [&] { *ptr = p->pd->fill_buf(len); });
}

int gzpx_pard_consume(gzpx_pard *p, size_t n) {
    if (!p) return GZPX_ERR_INVALID_ARG;
    return guarded_d(p, [&] { p->pd->consume(n); });
}

void gzpx_pard_destroy(gzpx_pard *p) { delete p; }
const char *gzpx_pard_last_error(const gzpx_pard *p) { return p ? p->last_error.c_str() : ""; }

}  // extern "C"
