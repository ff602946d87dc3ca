// gzpx_api.cpp -- the C ABI of include/gzpx.h over the HIP pipeline of gzpx_kernels.hip.
//
// Host-side mirror of what a gzp worker thread does per block (src/par/compress.rs:279-294:
// create_compressor once, encode per block) lifted to whole slabs: one call = every block of the
// slab through the kernel pipeline, blocks cut and ordered exactly like ParCompress::write /
// flush_last (src/par/compress.rs:413-463, 332-362).  No CPU fallback exists on purpose: without
// a HIP device every entry point fails with GZPX_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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

constexpr int kProfPairs = 64;  // launch groups of one batch that measurement mode can bracket

constexpr int kSlots = 3;  // slabs of one context that may be in flight (copy in / kernels / copy out)

// One slab in flight.  Host-buffer jobs own a pair of device staging buffers per slot, so the H2D
// copy of slab k+1 and the D2H copy of slab k-1 run (on their own streams) while the kernels of
// slab k occupy the compute stream.
struct Slot {
    int state = 0;  // 0 free, 1 submitted, 2 a thread is inside wait()
    uint64_t gen = 0;
    uint8_t *d_in = nullptr, *d_out = nullptr;  // staging of host-buffer jobs (grown on demand)
    size_t d_in_cap = 0, d_out_cap = 0;
    hipEvent_t ev_h2d = nullptr, ev_kernels = nullptr, ev_d2h = nullptr;
    hipEvent_t ev_dom_b = nullptr, ev_dom_e = nullptr;  // profiling mode 2: around the dominant stage of this job
    bool dom_timed = false;                             // ... recorded for this job (read back in wait)
    SlabResult *d_results = nullptr, *h_results = nullptr;  // one per batch of the slab (h_: pinned)
    size_t results_cap = 0;
    uint32_t *h_sizes = nullptr;  // pinned: framed size of every block of the slab
    size_t sizes_cap = 0;
    // the job
    uint8_t *job_d_out = nullptr;  // where the kernels wrote
    size_t job_out_cap = 0;
    uint8_t *host_out = nullptr;  // null: device job (output stays in HBM)
    size_t host_out_cap = 0;
    uint64_t total_nb = 0;
    uint32_t n_batches = 0;
};

}  // namespace

struct gzpx_ctx {
    gzpx_config cfg;
    Config dcfg;
    CrcConsts crc_consts;
    hipStream_t stream = nullptr;  // compute: every kernel of this context, in submission order
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    hipStream_t s_side = nullptr;  // k_crc32 next to k_candidates (needs only the input)
    hipEvent_t ev_meta = nullptr, ev_crc = nullptr;    // fork / join of the side stream
    hipEvent_t ev_crc_t0 = nullptr, ev_crc_t1 = nullptr;  // timing of k_crc32 when profiling
    hipEvent_t ev_dep = nullptr;  // "the caller's stream got this far" (device jobs)
    hipEvent_t prof_ev[2 * 64] = {nullptr};  // measurement mode: begin / end of launch groups
    int prof_stage[64] = {0};
    int prof_b[64] = {0}, prof_e[64] = {0};  // a group's begin / end event (indices into prof_ev)
    int prof_n = 0;
    uint32_t batch_blocks = 0;
    Scratch scratch = {};
    Slot slots[kSlots];
    uint64_t next_gen = 1;
    BlockMeta *h_meta = nullptr;  // pinned; CRC-only contexts and the debug hooks
    SubMeta *h_sub = nullptr;     // pinned, max_sub entries (debug hooks)
    int profiling = 0;  // 0 off, 1 every stage, 2 the dominant stage only (two markers per batch instead of thirteen)
    bool crc_only = false;
    float stage_ms[GZPX_N_STAGES] = {0};
    uint32_t last_nb = 0;
    char devname[256] = {0};
    std::mutex mu;
    std::condition_variable cv_slot;
};

namespace {

// (GZPX_TRACE in the environment: the failing runtime call is named on stderr)
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            if (getenv("GZPX_TRACE")) fprintf(stderr, "gzpx: %s:%d %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return GZPX_ERR_DEVICE;                                                                     \
        }                                                                                               \
    } while (0)

thread_local int t_last_status = GZPX_OK;  // libdeflate-shaped calls have no status channel

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
    return (size_t)c.stride * (2 + 1 + 2 + 4 + (c.level >= 2 ? 2 : 0) + (c.lazy ? 6 : 0)) + c.stride / 8 +
           (size_t)c.max_sub * (sizeof(SubMeta) + (kHistStride + kCodeWords + kHdrWords) * 4) +
           sizeof(BlockMeta) + 8 + 4;
}

int alloc_scratch(gzpx_ctx *ctx) {
    const size_t nb = ctx->batch_blocks;
    const Config &c = ctx->dcfg;
    Scratch &s = ctx->scratch;
    HIP_TRY(hipMalloc((void **)&s.meta, nb * sizeof(BlockMeta)));
    if (ctx->crc_only) {  // gzpx_crc32's private context: k_init_meta + k_crc32 only
        HIP_TRY(hipHostMalloc((void **)&ctx->h_meta, nb * sizeof(BlockMeta), hipHostMallocDefault));
        return GZPX_OK;
    }
    HIP_TRY(hipHostMalloc((void **)&ctx->h_meta, sizeof(BlockMeta), hipHostMallocDefault));
    HIP_TRY(hipMalloc((void **)&s.sub, nb * (size_t)c.max_sub * sizeof(SubMeta)));
    HIP_TRY(hipMalloc((void **)&s.cand, nb * (size_t)c.stride * sizeof(uint16_t)));
    HIP_TRY(hipMalloc((void **)&s.len8, nb * (size_t)c.stride));
    HIP_TRY(hipMalloc((void **)&s.which, nb * (size_t)(c.stride / 32) * 4));
    HIP_TRY(hipMalloc((void **)&s.alt, nb * (size_t)c.stride * sizeof(uint16_t)));
    HIP_TRY(hipMalloc((void **)&s.tok, nb * (size_t)c.stride * 4));
    HIP_TRY(hipMalloc((void **)&s.redo, (nb + 1 + 8) * sizeof(uint32_t)));
    HIP_TRY(hipMemset(s.redo, 0, (nb + 1 + 8) * sizeof(uint32_t)));
    s.claim = s.redo + nb + 1;
    if (c.level >= 2) {  // hc_matchfinder levels: hash4 chain links + per-block parse state
        HIP_TRY(hipMalloc((void **)&s.d4, nb * (size_t)c.stride * sizeof(uint16_t)));
        HIP_TRY(hipMalloc((void **)&s.hc, nb * sizeof(HcState)));
        HIP_TRY(hipMalloc((void **)&s.pending, 64));
    }
    if (c.lazy) {  // levels 5-9: the matches of the half / quarter depth searches
        HIP_TRY(hipMalloc((void **)&s.lz_len, nb * 2 * (size_t)c.stride));
        HIP_TRY(hipMalloc((void **)&s.lz_dist, nb * 2 * (size_t)c.stride * sizeof(uint16_t)));
    }
    if (c.level >= 10) {
        // levels 10-12: one lane per block in flight, each with its trees / cache / path nodes (gzpx_nearopt.hip);
        // as many as half the free HBM holds, 96 GiB at most (a BGZF block: 7.5 MiB); the lanes walk the batch's blocks
        const size_t per_lane = no_lane_bytes() + no_cache_bytes() + no_nodes_bytes(c.block_size);
        // (never more than half of what is free right now; if an allocation still fails -- fragmentation, another
        // context growing meanwhile -- the lane count is halved and tried again: fewer lanes are slower, not wrong)
        size_t free_b = 0, total_b = 0, budget = (size_t)12 << 30;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            if (free_b / 2 > budget) budget = free_b / 2 < ((size_t)96 << 30) ? free_b / 2 : ((size_t)96 << 30);  // (288 GB of HBM: up to 96 GiB of it)
            if (free_b / 2 < budget) budget = free_b / 2;
        }
        size_t lanes = budget / per_lane;
        if (lanes > nb) lanes = nb;
        if (lanes < 1) lanes = 1;
        for (;;) {
            const bool ok = hipMalloc(&s.no_state, lanes * no_lane_bytes()) == hipSuccess &&
                            hipMalloc((void **)&s.no_cache, lanes * no_cache_bytes()) == hipSuccess &&
                            hipMalloc((void **)&s.no_nodes, lanes * no_nodes_bytes(c.block_size)) == hipSuccess;
            if (ok) break;
            (void)hipGetLastError();
            if (s.no_state) (void)hipFree(s.no_state);
            if (s.no_cache) (void)hipFree(s.no_cache);
            if (s.no_nodes) (void)hipFree(s.no_nodes);
            s.no_state = nullptr;
            s.no_cache = nullptr;
            s.no_nodes = nullptr;
            if (lanes == 1) return GZPX_ERR_DEVICE;
            lanes = (lanes + 1) / 2;
        }
        s.no_lanes = (uint32_t)lanes;
        // libdeflate's default_litlen_costs[]: int(-log2((1 - p) / max(j, 1)) * 16), int(-log2(p / 29) * 16)
        uint8_t tables[3 * 258];
        static const double probs[3] = {0.25, 0.5, 0.75};
        for (int k = 0; k < 3; k++) {
            for (int j = 0; j <= 256; j++) tables[258 * k + j] = (uint8_t)(int)(-std::log2((1.0 - probs[k]) / (double)(j ? j : 1)) * 16);
            tables[258 * k + 257] = (uint8_t)(int)(-std::log2(probs[k] / 29.0) * 16);
        }
        uint8_t *d_tables = nullptr;
        HIP_TRY(hipMalloc((void **)&d_tables, sizeof(tables)));
        hipError_t te = hipMemcpy(d_tables, tables, sizeof(tables), hipMemcpyHostToDevice);
        if (te == hipSuccess) {
            launch_near_optimal_tables(s.no_state, s.no_lanes, d_tables, nullptr);
            te = hipDeviceSynchronize();
        }
        (void)hipFree(d_tables);  // (on every path)
        HIP_TRY(te);
    }
    HIP_TRY(hipMalloc((void **)&s.hist, nb * (size_t)c.max_sub * kHistStride * 4));
    HIP_TRY(hipMalloc((void **)&s.codes, nb * (size_t)c.max_sub * kCodeWords * 4));
    HIP_TRY(hipMalloc((void **)&s.hdr, nb * (size_t)c.max_sub * kHdrWords * 4));
    HIP_TRY(hipMalloc((void **)&s.out_off, (nb + 1) * sizeof(uint64_t)));
    HIP_TRY(hipMalloc((void **)&s.sizes, nb * sizeof(uint32_t)));
    HIP_TRY(hipHostMalloc((void **)&ctx->h_sub, (size_t)c.max_sub * sizeof(SubMeta), hipHostMallocDefault));
    return GZPX_OK;
}

void free_slot(Slot &sl) {
    if (sl.d_in) (void)hipFree(sl.d_in);
    if (sl.d_out) (void)hipFree(sl.d_out);
    if (sl.d_results) (void)hipFree(sl.d_results);
    if (sl.h_results) (void)hipHostFree(sl.h_results);
    if (sl.h_sizes) (void)hipHostFree(sl.h_sizes);
    if (sl.ev_h2d) (void)hipEventDestroy(sl.ev_h2d);
    if (sl.ev_kernels) (void)hipEventDestroy(sl.ev_kernels);
    if (sl.ev_d2h) (void)hipEventDestroy(sl.ev_d2h);
    if (sl.ev_dom_b) (void)hipEventDestroy(sl.ev_dom_b);
    if (sl.ev_dom_e) (void)hipEventDestroy(sl.ev_dom_e);
    sl = Slot();
}

void free_scratch(gzpx_ctx *ctx) {
    Scratch &s = ctx->scratch;
    if (s.meta) (void)hipFree(s.meta);
    if (s.sub) (void)hipFree(s.sub);
    if (ctx->h_sub) (void)hipHostFree(ctx->h_sub);
    if (s.cand) (void)hipFree(s.cand);
    if (s.tok) (void)hipFree(s.tok);
    if (s.redo) (void)hipFree(s.redo);
    if (s.no_state) (void)hipFree(s.no_state);
    if (s.no_cache) (void)hipFree(s.no_cache);
    if (s.no_nodes) (void)hipFree(s.no_nodes);
    if (s.len8) (void)hipFree(s.len8);
    if (s.which) (void)hipFree(s.which);
    if (s.alt) (void)hipFree(s.alt);
    if (s.d4) (void)hipFree(s.d4);
    if (s.hc) (void)hipFree(s.hc);
    if (s.pending) (void)hipFree(s.pending);
    if (s.lz_len) (void)hipFree(s.lz_len);
    if (s.lz_dist) (void)hipFree(s.lz_dist);
    if (s.hist) (void)hipFree(s.hist);
    if (s.codes) (void)hipFree(s.codes);
    if (s.hdr) (void)hipFree(s.hdr);
    if (s.out_off) (void)hipFree(s.out_off);
    if (s.sizes) (void)hipFree(s.sizes);
    if (ctx->h_meta) (void)hipHostFree(ctx->h_meta);
    for (Slot &sl : ctx->slots) free_slot(sl);
    s = Scratch{};
}

// HIP-event pairs around groups of launches (measurement mode only): a stage may consist of several
// launches on several streams, its time is the sum of its pairs.
// (A group that begins where the previous group of the same stream ended shares that event: every
// event record is a marker the stream has to process, and twenty of them cost the 3.7 ms step of the
// bench slab 0.08 ms.)
struct ProfPairs {
    gzpx_ctx *ctx;
    bool on;
    hipEvent_t dom_b = nullptr, dom_e = nullptr;  // mode 2, asynchronous: the job's own pair, read when it is waited for
    int used = 0;              // events handed out
    int last_ev = -1;          // the event recorded last ...
    hipStream_t last_st = nullptr;  // ... on this stream, with nothing enqueued behind it yet
    int begin(int stage, hipStream_t st) {
        if (!on || ctx->prof_n >= kProfPairs || used + 2 > 2 * kProfPairs) return -1;
        if (ctx->profiling == 2 && stage != 2) return -1;
        if (dom_b) {
            (void)hipEventRecord(dom_b, st);
            return -2;
        }
        const int i = ctx->prof_n++;
        ctx->prof_stage[i] = stage;
        if (last_ev >= 0 && last_st == st) {
            ctx->prof_b[i] = last_ev;
        } else {
            ctx->prof_b[i] = used++;
            (void)hipEventRecord(ctx->prof_ev[ctx->prof_b[i]], st);
        }
        last_ev = -1;
        return i;
    }
    void end(int i, hipStream_t st) {
        if (i == -2) (void)hipEventRecord(dom_e, st);
        if (i < 0) return;
        ctx->prof_e[i] = used++;
        (void)hipEventRecord(ctx->prof_ev[ctx->prof_e[i]], st);
        last_ev = ctx->prof_e[i];
        last_st = st;
    }
    // something that is not part of a measured group went onto `st`: the next group records its own begin
    void touch(hipStream_t st) {
        if (last_st == st) last_ev = -1;
    }
};

// One batch of blocks through the pipeline, enqueued on `stream`.  Nothing here waits for the device.
// `prev` / `result`: the batch before this one of the same slab (device, may be null) and this
// batch's own record; output offsets continue from prev->total.
int enqueue_batch(gzpx_ctx *ctx, const uint8_t *d_in, size_t in_len, uint32_t nb, int is_last,
                  uint8_t *d_out, size_t out_cap, hipStream_t stream, const SlabResult *prev,
                  SlabResult *result, hipEvent_t dom_b = nullptr, hipEvent_t dom_e = nullptr) {
    const Config &c = ctx->dcfg;
    const Scratch &s = ctx->scratch;
    ProfPairs pp{ctx, ctx->profiling != 0};
    pp.dom_b = dom_b;
    pp.dom_e = dom_e;
    ctx->prof_n = 0;
    // fork: the CRC of every block on the low-priority side stream, beside the match kernels.  Where it
    // runs is a question of whom it takes issue slots from (per 550 MiB step, round 3): beside k_mparse
    // that kernel goes from 2.01 to 2.07 ms and k_hist + k_huffman, alone now, from 0.41 to 0.33 -- 3.50
    // against 3.53 ms; forked behind k_mparse (in front of the dense pair) 3.56.  (Round 2, beside the
    // one-workgroup-per-block k_candidates: that kernel slowed down by the CRC's time.)
    auto fork_crc = [&]() -> int {
        HIP_TRY(hipEventRecord(ctx->ev_meta, stream));
        HIP_TRY(hipStreamWaitEvent(ctx->s_side, ctx->ev_meta, 0));
        pp.touch(stream);
        const int tc = pp.begin(6, ctx->s_side);
        launch_crc32(c, d_in, in_len, nb, s, ctx->crc_consts, ctx->s_side);
        pp.end(tc, ctx->s_side);
        HIP_TRY(hipEventRecord(ctx->ev_crc, ctx->s_side));
        return GZPX_OK;
    };
    // Levels 2-12 fork BEHIND their match / parse kernels, beside k_hist / k_huffman (round 6, tools/gpu_r6_fork_ab.sh: the
    // one-workgroup-per-CU matchers lose more to the CRC's waves than the small kernels do -- configs[2] 61.2-62.0 ->
    // 60.7 ms, text at levels 2-9 0.04 ms each).
    // (Config.debug bits 8-9, experiments: 1 = fork behind all match / parse kernels, 2 = behind the first of them,
    // 3 = beside them whatever the level)
    const uint32_t fork_sel = (c.debug >> 8) & 3u;
    const uint32_t fork_at = fork_sel == 0 ? (c.level <= 1 ? 1u : 0u) : fork_sel == 1 ? 0u : fork_sel == 2 ? 2u : 1u;
    // (the slab is cut into blocks -- BlockMeta -- by the first k_candidates launch; level 0: k_init_meta)
    int t = pp.begin(c.level == 0 ? 0 : 1, stream);
    launch_candidates(c, d_in, in_len, nb, is_last, s, stream);
    pp.end(t, stream);
    if (fork_at == 1) {
        const int rc = fork_crc();
        if (rc != GZPX_OK) return rc;
    }
    if (c.level <= 1) {  // (at level 0 every block is a passthrough block: the kernels return at once)
        // (Tried in round 3: the batch cut into 2..8 block ranges, k_hist / k_huffman of range k on a
        // side stream beside the matching of range k + 1.  4.27 -> 4.83 / 5.90 / 7.44 ms per step: the
        // 24 six-KiB workgroups k_huffman puts on a CU take the LDS that k_candidates / k_mparse need
        // whole, so the big kernels lose CUs to the small one instead of sharing them.)
        t = pp.begin(2, stream);
        launch_match(c, d_in, in_len, nb, s, stream);  // k_mparse, and k_match over the blocks it handed back
        pp.end(t, stream);
        if (fork_at == 2) {
            const int rc = fork_crc();
            if (rc != GZPX_OK) return rc;
        }
        t = pp.begin(3, stream);
        launch_parse(c, d_in, in_len, nb, s, stream);
        pp.end(t, stream);
    } else {
        t = pp.begin(2, stream);
        if (c.level >= 10) {  // levels 10-12: the near-optimal parser, one lane per block (gzpx_nearopt.hip)
            launch_near_optimal(c, d_in, nb, s, stream);
        } else if (c.lazy) {  // levels 5-9: every match variant once, then the serial-per-block lazy parse
            launch_lazy(c, d_in, nb, s, stream);
        } else {
            // levels 2-4: every match once, then the greedy parse; a block whose new sub-block needs
            // another min_len parses on inside k_parse_hc, so nothing is read back and submit returns
            // without waiting for the device (round 2 ran these rounds from the host, a word per round)
            launch_hc(c, d_in, nb, s, stream);
        }
        pp.end(t, stream);
    }
    if (fork_at == 0 || (fork_at == 2 && c.level > 1)) {
        const int rc = fork_crc();
        if (rc != GZPX_OK) return rc;
    }
    t = pp.begin(4, stream);
    launch_hist(c, nb, s, stream);
    pp.end(t, stream);
    t = pp.begin(5, stream);
    launch_huffman(c, nb, s, stream);
    pp.end(t, stream);
    t = pp.begin(7, stream);
    launch_scan(nb, s, prev, result, stream);
    pp.end(t, stream);
    // join behind k_scan (which does not read the CRCs): k_emit writes them into the footers
    HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_crc, 0));
    pp.touch(stream);  // (the wait for the side stream is nobody's stage time)
    t = pp.begin(8, stream);
    launch_emit(c, d_in, in_len, nb, s, d_out, out_cap, stream);
    pp.end(t, stream);
    HIP_TRY(hipGetLastError());
    if (pp.on && !dom_b) {  // measurement mode: one host wait per batch
        HIP_TRY(hipStreamSynchronize(stream));
        for (int i = 0; i < ctx->prof_n; i++) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, ctx->prof_ev[ctx->prof_b[i]], ctx->prof_ev[ctx->prof_e[i]]));
            ctx->stage_ms[ctx->prof_stage[i]] += ms;
        }
    }
    ctx->last_nb = nb;
    return GZPX_OK;
}

int slot_reserve(Slot &sl, size_t n_batches, size_t n_blocks) {
    if (n_batches > sl.results_cap) {
        if (sl.d_results) (void)hipFree(sl.d_results);
        if (sl.h_results) (void)hipHostFree(sl.h_results);
        sl.d_results = sl.h_results = nullptr;
        sl.results_cap = 0;
        const size_t cap = n_batches + 8;
        HIP_TRY(hipMalloc((void **)&sl.d_results, cap * sizeof(SlabResult)));
        HIP_TRY(hipHostMalloc((void **)&sl.h_results, cap * sizeof(SlabResult), hipHostMallocDefault));
        sl.results_cap = cap;
    }
    if (n_blocks > sl.sizes_cap) {
        if (sl.h_sizes) (void)hipHostFree(sl.h_sizes);
        sl.h_sizes = nullptr;
        sl.sizes_cap = 0;
        const size_t cap = n_blocks + n_blocks / 4 + 64;
        HIP_TRY(hipHostMalloc((void **)&sl.h_sizes, cap * sizeof(uint32_t), hipHostMallocDefault));
        sl.sizes_cap = cap;
    }
    return GZPX_OK;
}

int slot_staging(Slot &sl, size_t in_len, size_t out_need) {
    if (in_len + 16 > sl.d_in_cap) {
        if (sl.d_in) (void)hipFree(sl.d_in);
        sl.d_in = nullptr;
        sl.d_in_cap = 0;
        const size_t cap = in_len + in_len / 8 + 4096;
        HIP_TRY(hipMalloc((void **)&sl.d_in, cap));
        sl.d_in_cap = cap;
    }
    if (out_need > sl.d_out_cap) {
        if (sl.d_out) (void)hipFree(sl.d_out);
        sl.d_out = nullptr;
        sl.d_out_cap = 0;
        const size_t cap = out_need + out_need / 8 + 4096;
        HIP_TRY(hipMalloc((void **)&sl.d_out, cap));
        sl.d_out_cap = cap;
    }
    return GZPX_OK;
}

// The side stream runs at the lowest priority the device offers: its workgroups are only meant to
// fill what the main stream's kernels leave free.
hipError_t create_side_stream(hipStream_t *s) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, least);
}

int check_slab_args(const gzpx_ctx *ctx, const void *in, size_t in_len, int mode, const void *out) {
    const size_t bs = ctx->cfg.buffer_size;
    if (mode != GZPX_SLAB_FULL_BLOCKS && mode != GZPX_SLAB_LAST && mode != GZPX_SLAB_FLUSH)
        return GZPX_ERR_INVALID_ARG;
    if (mode == GZPX_SLAB_FULL_BLOCKS && (in_len == 0 || in_len % bs != 0)) return GZPX_ERR_INVALID_ARG;
    if ((in_len && !in) || !out) return GZPX_ERR_INVALID_ARG;
    return GZPX_OK;
}

// Enqueue one slab (ctx->mu held).  host_in / host_out non-null: a host-buffer job that goes
// through the slot's staging buffers; otherwise d_in / d_out are the caller's device buffers.
int submit_enqueue(gzpx_ctx *ctx, const uint8_t *host_in, const uint8_t *d_in, size_t in_len, int mode,
                   uint8_t *host_out, uint8_t *d_out, size_t out_cap, hipStream_t after, bool block_for_slot,
                   std::unique_lock<std::mutex> &lk, uint64_t *ticket);

// A submit that fails may already have put copies and kernels on the streams (the slot stays free):
// nothing of them may still be running when the caller gets its buffers back or the next submit reuses
// the slot's staging.
int submit_locked(gzpx_ctx *ctx, const uint8_t *host_in, const uint8_t *d_in, size_t in_len, int mode,
                  uint8_t *host_out, uint8_t *d_out, size_t out_cap, hipStream_t after, bool block_for_slot,
                  std::unique_lock<std::mutex> &lk, uint64_t *ticket) {
    const int rc = submit_enqueue(ctx, host_in, d_in, in_len, mode, host_out, d_out, out_cap, after, block_for_slot,
                                  lk, ticket);
    if (rc != GZPX_OK && rc != GZPX_ERR_BUSY && rc != GZPX_ERR_INVALID_ARG) {
        (void)hipStreamSynchronize(ctx->s_h2d);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->s_side);
    }
    return rc;
}

int submit_enqueue(gzpx_ctx *ctx, const uint8_t *host_in, const uint8_t *d_in, size_t in_len, int mode,
                   uint8_t *host_out, uint8_t *d_out, size_t out_cap, hipStream_t after, bool block_for_slot,
                   std::unique_lock<std::mutex> &lk, uint64_t *ticket) {
    if (ctx->crc_only) return GZPX_ERR_INVALID_ARG;
    const size_t caller_cap = out_cap;  // (host jobs: out_cap becomes the staging buffer's below)
    int si = -1;
    for (;;) {
        for (int i = 0; i < kSlots; i++)
            if (ctx->slots[i].state == 0) {
                si = i;
                break;
            }
        if (si >= 0) break;
        if (!block_for_slot) return GZPX_ERR_BUSY;
        ctx->cv_slot.wait(lk);
    }
    Slot &sl = ctx->slots[si];
    if (!sl.ev_kernels) {
        HIP_TRY(hipEventCreateWithFlags(&sl.ev_h2d, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.ev_kernels, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.ev_d2h, hipEventDisableTiming));
        HIP_TRY(hipEventCreate(&sl.ev_dom_b));
        HIP_TRY(hipEventCreate(&sl.ev_dom_e));
    }
    const size_t bs = ctx->cfg.buffer_size;
    const uint64_t total_nb = blocks_of(ctx, in_len);
    const uint64_t n_batches = (total_nb + ctx->batch_blocks - 1) / ctx->batch_blocks;
    int rc = slot_reserve(sl, (size_t)n_batches, (size_t)total_nb);
    if (rc != GZPX_OK) return rc;
    hipStream_t stream = ctx->stream;
    if (host_out) {
        rc = slot_staging(sl, in_len, gzpx_slab_bound(ctx, in_len));
        if (rc != GZPX_OK) return rc;
        if (in_len) {
            HIP_TRY(hipMemcpyAsync(sl.d_in, host_in, in_len, hipMemcpyHostToDevice, ctx->s_h2d));
            HIP_TRY(hipEventRecord(sl.ev_h2d, ctx->s_h2d));
            HIP_TRY(hipStreamWaitEvent(stream, sl.ev_h2d, 0));
        }
        d_in = sl.d_in;
        d_out = sl.d_out;
        out_cap = sl.d_out_cap;
    } else if (after != stream && after != (hipStream_t)GZPX_STREAM_NONE) {  // the slab is ready once `after` has reached this point; NULL is the
        // legacy default stream (PyTorch's current stream unless told otherwise) -- the context's
        // streams are non-blocking, so nothing orders them behind it implicitly
        HIP_TRY(hipEventRecord(ctx->ev_dep, after));
        HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_dep, 0));
    }
    memset(ctx->stage_ms, 0, sizeof(ctx->stage_ms));
    // profiling mode 2 on a one-batch slab: the job carries its own pair of events around the dominant stage and
    // stays asynchronous (the pair is read when the ticket is waited for); every other measurement waits per batch
    sl.dom_timed = ctx->profiling == 2 && n_batches == 1;
    const int is_last = mode == GZPX_SLAB_LAST;
    for (uint64_t bi = 0; bi < n_batches; bi++) {
        const uint64_t b0 = bi * ctx->batch_blocks;
        const uint32_t nb = (uint32_t)((total_nb - b0 < ctx->batch_blocks) ? total_nb - b0 : ctx->batch_blocks);
        const size_t in_begin = (size_t)b0 * bs;
        size_t in_batch = in_len > in_begin ? in_len - in_begin : 0;
        if (in_batch > (size_t)nb * bs) in_batch = (size_t)nb * bs;
        const int last_batch = (b0 + nb == total_nb) ? is_last : 0;
        rc = enqueue_batch(ctx, d_in + in_begin, in_batch, nb, last_batch, d_out, out_cap, stream,
                           bi ? sl.d_results + (bi - 1) : nullptr, sl.d_results + bi,
                           sl.dom_timed ? sl.ev_dom_b : nullptr, sl.dom_timed ? sl.ev_dom_e : nullptr);
        if (rc != GZPX_OK) {
            (void)hipStreamSynchronize(stream);
            return rc;
        }
        // the per-block sizes of this batch, before the next batch reuses the scratch
        HIP_TRY(hipMemcpyAsync(sl.h_sizes + b0, ctx->scratch.sizes, nb * sizeof(uint32_t), hipMemcpyDeviceToHost,
                               stream));
    }
    HIP_TRY(hipMemcpyAsync(sl.h_results, sl.d_results, (size_t)n_batches * sizeof(SlabResult),
                           hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipEventRecord(sl.ev_kernels, stream));
    sl.job_d_out = d_out;
    sl.job_out_cap = out_cap;
    sl.host_out = host_out;
    sl.host_out_cap = host_out ? caller_cap : 0;
    sl.total_nb = total_nb;
    sl.n_batches = (uint32_t)n_batches;
    sl.state = 1;
    sl.gen = ctx->next_gen++;
    *ticket = (sl.gen << 8) | (uint64_t)si;
    return GZPX_OK;
}

// Completing a ticket, in three phases (the multi-device entry runs them device by device so that the
// shards' copies overlap): wait for the kernels and read the result record; start the copy-out of a
// host job; wait for it and free the slot.
struct Completion {
    int rc = GZPX_OK;
    size_t produced = 0, blocks_done = 0;
};

int claim_ticket(gzpx_ctx *ctx, uint64_t ticket, Slot **slot) {
    const int si = (int)(ticket & 0xFF);
    if (si >= kSlots) return GZPX_ERR_INVALID_ARG;
    Slot &sl = ctx->slots[si];
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (sl.state != 1 || sl.gen != (ticket >> 8)) return GZPX_ERR_INVALID_ARG;
    sl.state = 2;
    *slot = &sl;
    return GZPX_OK;
}

Completion kernels_done(gzpx_ctx *ctx, Slot &sl) {
    Completion c;
    c.blocks_done = (size_t)sl.total_nb;
    if (hipSetDevice(ctx->cfg.device) != hipSuccess || hipEventSynchronize(sl.ev_kernels) != hipSuccess) {
        c.rc = GZPX_ERR_DEVICE;
        return c;
    }
    if (sl.dom_timed) {  // the stage times of the job that has just been waited for
        float ms = 0;
        std::lock_guard<std::mutex> lk(ctx->mu);  // (a submit on another thread clears stage_ms under this lock)
        memset(ctx->stage_ms, 0, sizeof(ctx->stage_ms));
        if (hipEventElapsedTime(&ms, sl.ev_dom_b, sl.ev_dom_e) == hipSuccess) ctx->stage_ms[2] = ms;
        sl.dom_timed = false;
    }
    for (uint32_t bi = 0; bi < sl.n_batches && c.rc == GZPX_OK; bi++) {
        const SlabResult &r = sl.h_results[bi];
        if (r.fail_block != 0xFFFFFFFFu) {
            c.blocks_done = (size_t)bi * ctx->batch_blocks + r.fail_block;
            c.rc = r.fail_status == kStatusBlockSizeExceeded ? GZPX_ERR_BLOCK_SIZE_EXCEEDED : GZPX_ERR_DEVICE;
        }
    }
    if (c.rc == GZPX_OK) {
        c.produced = (size_t)sl.h_results[sl.n_batches - 1].total;
        if (c.produced > sl.job_out_cap) c.rc = GZPX_ERR_INSUFFICIENT_SPACE;
    }
    return c;
}

// the kernels are done (the host has seen their event): the copy needs no stream dependency
int start_copy_out(gzpx_ctx *ctx, Slot &sl, uint8_t *host_out, size_t produced) {
    if (!produced) return GZPX_OK;
    if (hipSetDevice(ctx->cfg.device) != hipSuccess ||
        hipMemcpyAsync(host_out, sl.job_d_out, produced, hipMemcpyDeviceToHost, ctx->s_d2h) != hipSuccess ||
        hipEventRecord(sl.ev_d2h, ctx->s_d2h) != hipSuccess)
        return GZPX_ERR_DEVICE;
    return GZPX_OK;
}

int finish_copy_out(gzpx_ctx *ctx, Slot &sl, size_t produced) {
    if (!produced) return GZPX_OK;
    if (hipSetDevice(ctx->cfg.device) != hipSuccess || hipEventSynchronize(sl.ev_d2h) != hipSuccess) return GZPX_ERR_DEVICE;
    return GZPX_OK;
}

void release_slot(gzpx_ctx *ctx, Slot &sl) {
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        sl.state = 0;
    }
    ctx->cv_slot.notify_all();
}

// Complete a ticket: wait for its kernels, copy the stream out (host jobs), report.
int wait_ticket(gzpx_ctx *ctx, uint64_t ticket, size_t *out_len, uint32_t *block_sizes, size_t max_blocks,
                size_t *n_blocks) {
    Slot *slp = nullptr;
    int rc = claim_ticket(ctx, ticket, &slp);
    if (rc != GZPX_OK) return rc;
    Slot &sl = *slp;
    const size_t host_out_cap = sl.host_out_cap;  // (the slot is ours from the claim on: no other thread writes it)
    Completion c = kernels_done(ctx, sl);
    rc = c.rc;
    if (rc == GZPX_OK && sl.host_out && c.produced > host_out_cap) rc = GZPX_ERR_INSUFFICIENT_SPACE;
    if (rc == GZPX_OK && block_sizes) {
        if (max_blocks < sl.total_nb) rc = GZPX_ERR_INVALID_ARG;
        else memcpy(block_sizes, sl.h_sizes, (size_t)sl.total_nb * sizeof(uint32_t));
    }
    if (rc == GZPX_OK && sl.host_out) {
        rc = start_copy_out(ctx, sl, sl.host_out, c.produced);
        if (rc == GZPX_OK) rc = finish_copy_out(ctx, sl, c.produced);
    }
    if (out_len) *out_len = rc == GZPX_OK ? c.produced : 0;
    if (n_blocks) *n_blocks = c.blocks_done;
    release_slot(ctx, sl);
    return rc;
}

int ctx_create(const gzpx_config *cfg, bool crc_only, gzpx_ctx **out);

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

int gzpx_ctx_create(const gzpx_config *cfg, gzpx_ctx **out) { return ctx_create(cfg, false, out); }

}  // extern "C"

namespace {

int ctx_create(const gzpx_config *cfg, bool crc_only, gzpx_ctx **out) {
    if (!cfg || !out) return GZPX_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->format != GZPX_FORMAT_BGZF && cfg->format != GZPX_FORMAT_MGZIP) return GZPX_ERR_INVALID_ARG;
    if (cfg->buffer_size < kDictSize) return GZPX_ERR_BUFFER_SIZE;  // src/par/compress.rs:68-74
    if (cfg->level < 0 || cfg->level > 12) return GZPX_ERR_COMPRESSION_LEVEL;
    if (cfg->compat != GZPX_COMPAT_LIBDEFLATE_1_24 && cfg->compat != GZPX_COMPAT_LIBDEFLATE_1_10)
        return GZPX_ERR_INVALID_ARG;
    if (cfg->buffer_size > kMaxBlockSize) return GZPX_ERR_UNSUPPORTED;  // > 16 MiB blocks: not built
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GZPX_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return GZPX_ERR_INVALID_ARG;
    if (hipSetDevice(cfg->device) != hipSuccess) return GZPX_ERR_DEVICE;

    gzpx_ctx *ctx = new (std::nothrow) gzpx_ctx();
    if (!ctx) return GZPX_ERR_DEVICE;
    ctx->cfg = *cfg;
    ctx->crc_only = crc_only;
    ctx->dcfg.format = (uint32_t)cfg->format;
    ctx->dcfg.level = (uint32_t)cfg->level;
    // Levels 10-12 are libdeflate 1.10's near-optimal parser (1.24 changed it; there is nothing here to pin a port of
    // the later parser on): the 1.10 rules go with it WHATEVER the caller asked for, so the stream is exactly the
    // 1.10 binary's and never a hybrid of two versions.  gzpx_ctx_active_compat() tells which rules a context runs.
    ctx->dcfg.compat = (uint32_t)(cfg->level >= 10 ? GZPX_COMPAT_LIBDEFLATE_1_10 : cfg->compat);
    if (cfg->level >= 10 && cfg->compat != GZPX_COMPAT_LIBDEFLATE_1_10 && getenv("GZPX_TRACE")) {
        static std::atomic<bool> told{false};  // (once per process; ADVICE round 4: the override used to be silent)
        if (!told.exchange(true))
            fprintf(stderr, "gzpx: level %d runs libdeflate 1.10's near-optimal parser; compat %d was asked for, the 1.10 "
                            "rules are in force (gzpx_ctx_active_compat)\n", cfg->level, cfg->compat);
    }
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
        // libdeflate_alloc_compressor: max_search_depth / nice_match_length of levels 2-9
        // (levels 10-12, deflate_compress_near_optimal: + num_optim_passes 2 / 3 / 4)
        static const uint32_t depth[13] = {0, 0, 6, 12, 16, 16, 35, 100, 300, 600, 35, 70, 150};
        static const uint32_t nice[13] = {0, 0, 10, 14, 30, 30, 65, 130, 258, 258, 75, 150, 258};
        ctx->dcfg.hc_depth = depth[cfg->level];
        ctx->dcfg.hc_nice = nice[cfg->level];
        ctx->dcfg.lazy = cfg->level >= 10 ? 0u : cfg->level >= 8 ? 2u : cfg->level >= 5 ? 1u : 0u;
        ctx->dcfg.no_passes = cfg->level >= 10 ? (uint32_t)cfg->level - 8u : 0u;
    }
    for (unsigned l = 0; l < 10; l++) ctx->crc_consts.pow64[l] = x2k(9 + l);
    ctx->crc_consts.pow_tile = x2k(19);  // x^(8 * 65536) = x^(2^19)
    ctx->crc_consts.pow_small = x2k(17);  // x^(8 * 16384)
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) {
        snprintf(ctx->devname, sizeof(ctx->devname), "%.150s (%.60s, %d CUs)", prop.name, prop.gcnArchName,
                 prop.multiProcessorCount);
        ctx->dcfg.n_cu = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 0u;
    }
    const uint64_t want = blocks_of(ctx, cfg->max_slab_bytes ? cfg->max_slab_bytes : 1);
    ctx->batch_blocks = (uint32_t)(want < kMaxBatchBlocks ? want : kMaxBatchBlocks);
    if (!crc_only) {
        const size_t fit = kMaxScratchBytes / scratch_bytes_per_block(ctx->dcfg);
        if (ctx->batch_blocks > fit) ctx->batch_blocks = (uint32_t)fit;
    }
    if (ctx->batch_blocks == 0) ctx->batch_blocks = 1;
    int rc = GZPX_OK;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->s_h2d, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->s_d2h, hipStreamNonBlocking) != hipSuccess ||
        create_side_stream(&ctx->s_side) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_meta, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_crc, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&ctx->ev_crc_t0) != hipSuccess || hipEventCreate(&ctx->ev_crc_t1) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_dep, hipEventDisableTiming) != hipSuccess)
        rc = GZPX_ERR_DEVICE;
    if (rc == GZPX_OK) rc = alloc_scratch(ctx);
    if (rc == GZPX_OK) {
        for (int i = 0; i < 2 * kProfPairs; i++)
            if (hipEventCreate(&ctx->prof_ev[i]) != hipSuccess) rc = GZPX_ERR_DEVICE;
    }
    if (rc != GZPX_OK) {
        gzpx_ctx_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return GZPX_OK;
}

}  // namespace

extern "C" {

void gzpx_ctx_destroy(gzpx_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->s_h2d) (void)hipStreamSynchronize(ctx->s_h2d);
    if (ctx->s_d2h) (void)hipStreamSynchronize(ctx->s_d2h);
    free_scratch(ctx);
    for (hipEvent_t e : ctx->prof_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->ev_dep) (void)hipEventDestroy(ctx->ev_dep);
    for (hipEvent_t e : {ctx->ev_meta, ctx->ev_crc, ctx->ev_crc_t0, ctx->ev_crc_t1})
        if (e) (void)hipEventDestroy(e);
    if (ctx->s_side) {
        (void)hipStreamSynchronize(ctx->s_side);
        (void)hipStreamDestroy(ctx->s_side);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->s_h2d) (void)hipStreamDestroy(ctx->s_h2d);
    if (ctx->s_d2h) (void)hipStreamDestroy(ctx->s_d2h);
    delete ctx;
}

int gzpx_ctx_active_compat(const gzpx_ctx *ctx) { return ctx ? (int)ctx->dcfg.compat : GZPX_ERR_INVALID_ARG; }

size_t gzpx_slab_bound(const gzpx_ctx *ctx, size_t in_len) {
    if (!ctx) return 0;
    return (size_t)blocks_of(ctx, in_len) * framed_bound_per_block(ctx) + 28 + 64;
}

int gzpx_compress_slab_submit(gzpx_ctx *ctx, const uint8_t *in, size_t in_len, int mode, uint8_t *out,
                              size_t out_cap, uint64_t *ticket) {
    if (!ctx || !ticket) return GZPX_ERR_INVALID_ARG;
    int rc = check_slab_args(ctx, in, in_len, mode, out);
    if (rc != GZPX_OK) return rc;
    std::unique_lock<std::mutex> lk(ctx->mu);
    if (hipSetDevice(ctx->cfg.device) != hipSuccess) return GZPX_ERR_DEVICE;
    return submit_locked(ctx, in, nullptr, in_len, mode, out, nullptr, out_cap, nullptr, false, lk, ticket);
}

int gzpx_compress_slab_submit_device(gzpx_ctx *ctx, const void *d_in, size_t in_len, int mode, void *d_out,
                                     size_t out_cap, void *after_stream, uint64_t *ticket) {
    if (!ctx || !ticket) return GZPX_ERR_INVALID_ARG;
    int rc = check_slab_args(ctx, d_in, in_len, mode, d_out);
    if (rc != GZPX_OK) return rc;
    std::unique_lock<std::mutex> lk(ctx->mu);
    if (hipSetDevice(ctx->cfg.device) != hipSuccess) return GZPX_ERR_DEVICE;
    return submit_locked(ctx, nullptr, (const uint8_t *)d_in, in_len, mode, nullptr, (uint8_t *)d_out, out_cap,
                         (hipStream_t)after_stream, false, lk, ticket);
}

int gzpx_compress_slab_wait(gzpx_ctx *ctx, uint64_t ticket, size_t *out_len, uint32_t *block_sizes,
                            size_t max_blocks, size_t *n_blocks) {
    if (!ctx) return GZPX_ERR_INVALID_ARG;
    const int si = (int)(ticket & 0xFF);
    if (si >= kSlots) return GZPX_ERR_INVALID_ARG;
    return wait_ticket(ctx, ticket, out_len, block_sizes, max_blocks, n_blocks);
}

int gzpx_compress_slab_event(gzpx_ctx *ctx, uint64_t ticket, void **hip_event) {
    if (!ctx || !hip_event) return GZPX_ERR_INVALID_ARG;
    const int si = (int)(ticket & 0xFF);
    if (si >= kSlots) return GZPX_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    Slot &sl = ctx->slots[si];
    if (sl.state != 1 || sl.gen != (ticket >> 8)) return GZPX_ERR_INVALID_ARG;
    *hip_event = (void *)sl.ev_kernels;
    return GZPX_OK;
}

int gzpx_compress_slab_device(gzpx_ctx *ctx, const void *d_in, size_t in_len, int mode,
                              void *d_out, size_t out_cap, size_t *out_len, uint32_t *block_sizes,
                              size_t max_blocks, size_t *n_blocks, void *hip_stream) {
    if (!ctx || !out_len) return GZPX_ERR_INVALID_ARG;
    int rc = check_slab_args(ctx, d_in, in_len, mode, d_out);
    if (rc != GZPX_OK) return rc;
    if (block_sizes && max_blocks < blocks_of(ctx, in_len)) return GZPX_ERR_INVALID_ARG;
    uint64_t ticket = 0;
    {
        std::unique_lock<std::mutex> lk(ctx->mu);
        if (hipSetDevice(ctx->cfg.device) != hipSuccess) return GZPX_ERR_DEVICE;
        rc = submit_locked(ctx, nullptr, (const uint8_t *)d_in, in_len, mode, nullptr, (uint8_t *)d_out, out_cap,
                           (hipStream_t)hip_stream, true, lk, &ticket);
    }
    if (rc != GZPX_OK) return rc;
    return wait_ticket(ctx, ticket, out_len, block_sizes, max_blocks, n_blocks);
}

int gzpx_compress_slab(gzpx_ctx *ctx, const uint8_t *in, size_t in_len, int mode, uint8_t *out,
                       size_t out_cap, size_t *out_len, uint32_t *block_sizes, size_t max_blocks,
                       size_t *n_blocks) {
    if (!ctx || !out_len) return GZPX_ERR_INVALID_ARG;
    int rc = check_slab_args(ctx, in, in_len, mode, out);
    if (rc != GZPX_OK) return rc;
    if (block_sizes && max_blocks < blocks_of(ctx, in_len)) return GZPX_ERR_INVALID_ARG;
    uint64_t ticket = 0;
    {
        std::unique_lock<std::mutex> lk(ctx->mu);
        if (hipSetDevice(ctx->cfg.device) != hipSuccess) return GZPX_ERR_DEVICE;
        rc = submit_locked(ctx, in, nullptr, in_len, mode, out, nullptr, out_cap, nullptr, true, lk, &ticket);
    }
    if (rc != GZPX_OK) return rc;
    return wait_ticket(ctx, ticket, out_len, block_sizes, max_blocks, n_blocks);
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

// ---------------------------------------------------------------- multi-device slab call
// SURVEY 8(b) "multi-device variant", 8(e): blocks share no state, so a slab shards into contiguous
// block ranges, one per device, balanced to one block; only the range that holds the slab's end is
// cut with the caller's mode (short final piece / EOF marker).  Every device runs the single-device
// pipeline on its range -- no data-path exchange between them -- and the in-order write-out is done
// on the host side of the boundary: once the shard sizes are known (16 bytes per device), every
// device copies its shard straight to its offset in `out`, all copies in flight together, each
// over its own PCIe link.  (The one-process-per-GPU form of the same split, with the shards
// gathered over RCCL, is gzp_amd/shard.py + bench.py --gpus N.)
struct gzpx_multi {
    std::vector<gzpx_ctx *> ctxs;
    size_t buffer_size = 0;
    std::mutex mu;
    // device-resident form: every device's own output staging (its shard before the gather)
    std::vector<uint8_t *> d_stage;
    std::vector<size_t> d_stage_cap;
};

namespace {
// the block range of device g of G when a slab of in_len bytes is cut (contiguous, balanced to one block)
struct MultiPart {
    uint64_t first = 0, nb = 0;
    size_t lo = 0, n = 0;
};
void multi_part_of(size_t in_len, size_t bs, size_t G, size_t g, MultiPart &p) {
    const uint64_t total_nb = in_len == 0 ? 1 : (in_len + bs - 1) / bs;
    p = MultiPart();
    for (size_t k = 0; k <= g; k++) {
        p.first += p.nb;
        p.nb = total_nb / G + (k < total_nb % G ? 1 : 0);
    }
    p.lo = (size_t)(p.first * bs < in_len ? p.first * bs : in_len);
    const size_t hi = (size_t)((p.first + p.nb) * bs < in_len ? (p.first + p.nb) * bs : in_len);
    p.n = hi - p.lo;
}
}  // namespace

int gzpx_multi_create(const gzpx_config *cfg, const int *devices, size_t n_devices, gzpx_multi **out) {
    if (!cfg || !devices || !n_devices || !out) return GZPX_ERR_INVALID_ARG;
    *out = nullptr;
    gzpx_multi *m = new (std::nothrow) gzpx_multi();
    if (!m) return GZPX_ERR_DEVICE;
    m->buffer_size = cfg->buffer_size;
    for (size_t g = 0; g < n_devices; g++) {
        gzpx_config c = *cfg;
        c.device = devices[g];
        // each device sees 1/n of the largest slab (+ one block of imbalance)
        if (c.max_slab_bytes) c.max_slab_bytes = c.max_slab_bytes / n_devices + 2 * c.buffer_size;
        gzpx_ctx *ctx = nullptr;
        const int rc = gzpx_ctx_create(&c, &ctx);
        if (rc != GZPX_OK) {
            gzpx_multi_destroy(m);
            return rc;
        }
        m->ctxs.push_back(ctx);
    }
    // device-to-device copies of the gather: direct over xGMI where the devices can reach each other
    for (size_t a = 0; a < n_devices; a++)
        for (size_t b = 0; b < n_devices; b++) {
            int can = 0;
            if (devices[a] != devices[b] && hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can &&
                hipSetDevice(devices[a]) == hipSuccess)
                (void)hipDeviceEnablePeerAccess(devices[b], 0);  // (already enabled: an error we ignore)
        }
    (void)hipGetLastError();
    *out = m;
    return GZPX_OK;
}

void gzpx_multi_destroy(gzpx_multi *m) {
    if (!m) return;
    for (size_t g = 0; g < m->d_stage.size(); g++)
        if (m->d_stage[g] && g < m->ctxs.size() && hipSetDevice(m->ctxs[g]->cfg.device) == hipSuccess)
            (void)hipFree(m->d_stage[g]);
    for (gzpx_ctx *c : m->ctxs) gzpx_ctx_destroy(c);
    delete m;
}

int gzpx_multi_shard(const gzpx_multi *m, size_t in_len, size_t g, size_t *offset, size_t *len) {
    if (!m || g >= m->ctxs.size() || !offset || !len) return GZPX_ERR_INVALID_ARG;
    MultiPart p;
    multi_part_of(in_len, m->buffer_size, m->ctxs.size(), g, p);
    *offset = p.lo;
    *len = p.n;
    return GZPX_OK;
}

// The device-resident form: north_star's write-out.  Range g of the slab already lives on device g
// (d_in[g]); every device compresses its range into its own staging buffer, the host learns the shard
// sizes (a 16-byte record per device) and each shard then travels ONCE, device to device, into its
// stream offset of `d_out` on devices[root] -- hipMemcpyPeerAsync on the owning device's copy stream,
// all peers at once (every peer has its own xGMI link to the root); the root's own shard is a local
// copy.  No payload byte touches host memory.  In-order property: src/par/compress.rs:305-310.
int gzpx_multi_compress_slab_device(gzpx_multi *m, const void *const *d_in, size_t in_len, int mode, size_t root,
                                    void *d_out, size_t out_cap, size_t *out_len, uint32_t *block_sizes,
                                    size_t max_blocks, size_t *n_blocks) {
    if (!m || !out_len || m->ctxs.empty() || !d_in || !d_out || root >= m->ctxs.size()) return GZPX_ERR_INVALID_ARG;
    if (mode != GZPX_SLAB_FULL_BLOCKS && mode != GZPX_SLAB_LAST && mode != GZPX_SLAB_FLUSH) return GZPX_ERR_INVALID_ARG;
    const size_t bs = m->buffer_size, G = m->ctxs.size();
    if (mode == GZPX_SLAB_FULL_BLOCKS && (in_len == 0 || in_len % bs != 0)) return GZPX_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> guard(m->mu);
    const uint64_t total_nb = in_len == 0 ? 1 : (in_len + bs - 1) / bs;
    if (block_sizes && max_blocks < total_nb) return GZPX_ERR_INVALID_ARG;
    struct Part {
        MultiPart r;
        uint64_t ticket = 0;
        Slot *slot = nullptr;
        Completion c;
        bool submitted = false;
    };
    std::vector<Part> parts(G);
    m->d_stage.resize(G, nullptr);
    m->d_stage_cap.resize(G, 0);
    int rc = GZPX_OK;
    // 1. every device: the kernels of its range, output into its own staging (devices work concurrently)
    for (size_t g = 0; g < G && rc == GZPX_OK; g++) {
        Part &p = parts[g];
        multi_part_of(in_len, bs, G, g, p.r);
        if (p.r.nb == 0) continue;
        if (p.r.n && !d_in[g]) {
            rc = GZPX_ERR_INVALID_ARG;
            break;
        }
        const bool owns_end = p.r.first + p.r.nb == total_nb;
        gzpx_ctx *ctx = m->ctxs[g];
        std::unique_lock<std::mutex> lk(ctx->mu);
        if (hipSetDevice(ctx->cfg.device) != hipSuccess) {
            rc = GZPX_ERR_DEVICE;
            break;
        }
        const size_t need = gzpx_slab_bound(ctx, p.r.n);
        if (need > m->d_stage_cap[g]) {
            if (m->d_stage[g]) (void)hipFree(m->d_stage[g]);
            m->d_stage[g] = nullptr;
            m->d_stage_cap[g] = 0;
            if (hipMalloc((void **)&m->d_stage[g], need + need / 8 + 4096) != hipSuccess) {
                rc = GZPX_ERR_DEVICE;
                break;
            }
            m->d_stage_cap[g] = need + need / 8 + 4096;
        }
        rc = submit_locked(ctx, nullptr, (const uint8_t *)d_in[g], p.r.n, owns_end ? mode : GZPX_SLAB_FULL_BLOCKS, nullptr,
                           m->d_stage[g], m->d_stage_cap[g], (hipStream_t)GZPX_STREAM_NONE, true, lk, &p.ticket);  // (the caller's contract: the ranges are ready)
        p.submitted = rc == GZPX_OK;
    }
    // 2. shard sizes -> stream offsets
    size_t fail_block = (size_t)total_nb;
    for (size_t g = 0; g < G; g++) {
        Part &p = parts[g];
        if (!p.submitted) continue;
        if (claim_ticket(m->ctxs[g], p.ticket, &p.slot) != GZPX_OK) {
            if (rc == GZPX_OK) rc = GZPX_ERR_DEVICE;
            p.submitted = false;
            continue;
        }
        p.c = kernels_done(m->ctxs[g], *p.slot);
        if (p.c.rc != GZPX_OK && rc == GZPX_OK) {  // the first failing block in stream order
            rc = p.c.rc;
            fail_block = (size_t)p.r.first + p.c.blocks_done;
        }
    }
    size_t total = 0;
    std::vector<size_t> offs(G, 0);
    for (size_t g = 0; g < G; g++) {
        offs[g] = total;
        if (parts[g].submitted) total += parts[g].c.produced;
    }
    if (rc == GZPX_OK && total > out_cap) rc = GZPX_ERR_INSUFFICIENT_SPACE;
    // 3. the ordered gather: every shard straight into its place on the root device, all at once
    const int root_dev = m->ctxs[root]->cfg.device;
    for (size_t g = 0; g < G && rc == GZPX_OK; g++) {
        Part &p = parts[g];
        if (!p.submitted || !p.c.produced) continue;
        gzpx_ctx *ctx = m->ctxs[g];
        if (hipSetDevice(ctx->cfg.device) != hipSuccess ||
            hipMemcpyPeerAsync((uint8_t *)d_out + offs[g], root_dev, m->d_stage[g], ctx->cfg.device, p.c.produced,
                               ctx->s_d2h) != hipSuccess ||
            hipEventRecord(p.slot->ev_d2h, ctx->s_d2h) != hipSuccess)
            rc = GZPX_ERR_DEVICE;
    }
    for (size_t g = 0; g < G; g++) {
        Part &p = parts[g];
        if (!p.submitted) continue;
        const int r2 = finish_copy_out(m->ctxs[g], *p.slot, p.c.produced);
        if (rc == GZPX_OK) rc = r2;
        if (rc == GZPX_OK && block_sizes)
            memcpy(block_sizes + p.r.first, p.slot->h_sizes, (size_t)p.r.nb * sizeof(uint32_t));
        release_slot(m->ctxs[g], *p.slot);
    }
    *out_len = rc == GZPX_OK ? total : 0;
    if (n_blocks) *n_blocks = rc == GZPX_OK ? (size_t)total_nb : fail_block;
    return rc;
}

size_t gzpx_multi_devices(const gzpx_multi *m) { return m ? m->ctxs.size() : 0; }

int gzpx_multi_compress_slab(gzpx_multi *m, const uint8_t *in, size_t in_len, int mode, uint8_t *out,
                             size_t out_cap, size_t *out_len, uint32_t *block_sizes, size_t max_blocks,
                             size_t *n_blocks) {
    if (!m || !out_len || m->ctxs.empty()) return GZPX_ERR_INVALID_ARG;
    int rc = check_slab_args(m->ctxs[0], in, in_len, mode, out);
    if (rc != GZPX_OK) return rc;
    std::lock_guard<std::mutex> guard(m->mu);
    const size_t bs = m->buffer_size, G = m->ctxs.size();
    const uint64_t total_nb = in_len == 0 ? 1 : (in_len + bs - 1) / bs;
    if (block_sizes && max_blocks < total_nb) return GZPX_ERR_INVALID_ARG;
    struct Part {
        uint64_t first = 0, nb = 0;
        size_t lo = 0, n = 0;
        uint64_t ticket = 0;
        Slot *slot = nullptr;
        Completion c;
        bool submitted = false;
    };
    std::vector<Part> parts(G);
    uint64_t first = 0;
    for (size_t g = 0; g < G; g++) {  // contiguous ranges, balanced to within one block
        Part &p = parts[g];
        p.first = first;
        p.nb = total_nb / G + (g < total_nb % G ? 1 : 0);
        first += p.nb;
        p.lo = (size_t)(p.first * bs < in_len ? p.first * bs : in_len);
        const size_t hi = (size_t)((p.first + p.nb) * bs < in_len ? (p.first + p.nb) * bs : in_len);
        p.n = hi - p.lo;
    }
    // 1. every device: copy-in + kernels of its range (devices work concurrently)
    for (size_t g = 0; g < G && rc == GZPX_OK; g++) {
        Part &p = parts[g];
        if (p.nb == 0) continue;
        const bool owns_end = p.first + p.nb == total_nb;
        gzpx_ctx *ctx = m->ctxs[g];
        std::unique_lock<std::mutex> lk(ctx->mu);
        if (hipSetDevice(ctx->cfg.device) != hipSuccess) {
            rc = GZPX_ERR_DEVICE;
            break;
        }
        // (`out` is only a placeholder here: the real destination is known after the sizes are)
        rc = submit_locked(ctx, in + p.lo, nullptr, p.n, owns_end ? mode : GZPX_SLAB_FULL_BLOCKS, out, nullptr, 0, nullptr,
                           true, lk, &p.ticket);
        p.submitted = rc == GZPX_OK;
    }
    // 2. sizes -> offsets; 3. all copies started; 4. all copies finished
    size_t fail_block = (size_t)total_nb;
    for (size_t g = 0; g < G; g++) {
        Part &p = parts[g];
        if (!p.submitted) continue;
        if (claim_ticket(m->ctxs[g], p.ticket, &p.slot) != GZPX_OK) {
            if (rc == GZPX_OK) rc = GZPX_ERR_DEVICE;
            p.submitted = false;
            continue;
        }
        p.c = kernels_done(m->ctxs[g], *p.slot);
        if (p.c.rc != GZPX_OK && rc == GZPX_OK) {  // the first failing block in stream order
            rc = p.c.rc;
            fail_block = (size_t)p.first + p.c.blocks_done;
        }
    }
    size_t total = 0;
    std::vector<size_t> offs(G, 0);
    for (size_t g = 0; g < G; g++) {
        offs[g] = total;
        if (parts[g].submitted) total += parts[g].c.produced;
    }
    if (rc == GZPX_OK && total > out_cap) rc = GZPX_ERR_INSUFFICIENT_SPACE;
    for (size_t g = 0; g < G && rc == GZPX_OK; g++)
        if (parts[g].submitted) rc = start_copy_out(m->ctxs[g], *parts[g].slot, out + offs[g], parts[g].c.produced);
    for (size_t g = 0; g < G; g++) {
        Part &p = parts[g];
        if (!p.submitted) continue;
        const int r2 = finish_copy_out(m->ctxs[g], *p.slot, p.c.produced);
        if (rc == GZPX_OK) rc = r2;
        if (rc == GZPX_OK && block_sizes)
            memcpy(block_sizes + p.first, p.slot->h_sizes, (size_t)p.nb * sizeof(uint32_t));
        release_slot(m->ctxs[g], *p.slot);
    }
    *out_len = rc == GZPX_OK ? total : 0;
    if (n_blocks) *n_blocks = rc == GZPX_OK ? (size_t)total_nb : fail_block;
    return rc;
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

size_t gzpx_deflate_compress_bound(gzpx_compressor *c, size_t n) {
    // libdeflate_deflate_compress_bound: worst case = stored blocks of MIN_BLOCK_LENGTH (5000)
    // bytes, 5 bytes of header each.  v1.10 (probed on the image's binary) adds 1 + 8 bytes of
    // slack (OUTPUT_END_PADDING); the pinned v1.24 does not.
    size_t max_blocks = (n + 4999) / 5000;
    if (max_blocks < 1) max_blocks = 1;
    const size_t slack = (c && c->compat == GZPX_COMPAT_LIBDEFLATE_1_10) ? 9 : 0;
    return 5 * max_blocks + n + slack;
}

void gzpx_free_compressor(gzpx_compressor *c) {
    if (!c) return;
    if (c->ctx) gzpx_ctx_destroy(c->ctx);
    delete c;
}

int gzpx_crc32_checked(uint32_t crc, const void *buf, size_t n, uint32_t *out) {
    // libdeflate_crc32 semantics: crc32(crc, buf) = combine(crc, crc32(0, buf), n)
    static std::mutex mu;
    static gzpx_ctx *ctx = nullptr;
    if (!out || (!buf && n)) return GZPX_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(mu);
    if (!ctx) {
        gzpx_config cfg;
        gzpx_config_default(&cfg, GZPX_FORMAT_MGZIP);
        cfg.level = 1;
        cfg.buffer_size = kTile;
        cfg.max_slab_bytes = (size_t)64 << 20;
        const int rc = ctx_create(&cfg, true, &ctx);  // k_init_meta + k_crc32 only: no compressor scratch
        if (rc != GZPX_OK) return rc;
    }
    *out = crc;
    if (n == 0) return GZPX_OK;
    std::lock_guard<std::mutex> lock2(ctx->mu);
    if (hipSetDevice(ctx->cfg.device) != hipSuccess) return GZPX_ERR_DEVICE;
    Slot &sl = ctx->slots[0];  // staging only
    const uint8_t *p = (const uint8_t *)buf;
    const size_t slab_max = (size_t)ctx->batch_blocks * kTile;
    while (n) {
        const size_t take = n < slab_max ? n : slab_max;
        int rc = slot_staging(sl, take, 0);
        if (rc != GZPX_OK) return rc;
        const uint32_t nb = (uint32_t)((take + kTile - 1) / kTile);
        HIP_TRY(hipMemcpyAsync(sl.d_in, p, take, hipMemcpyHostToDevice, ctx->stream));
        launch_init_meta(ctx->dcfg, take, nb, 0, ctx->scratch, ctx->stream);
        launch_crc32(ctx->dcfg, sl.d_in, take, nb, ctx->scratch, ctx->crc_consts, ctx->stream);
        HIP_TRY(hipMemcpyAsync(ctx->h_meta, ctx->scratch.meta, nb * sizeof(BlockMeta), hipMemcpyDeviceToHost,
                               ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (uint32_t b = 0; b < nb; b++) crc = crc32_combine(crc, ctx->h_meta[b].crc, ctx->h_meta[b].n);
        p += take;
        n -= take;
    }
    *out = crc;
    return GZPX_OK;
}

uint32_t gzpx_crc32(uint32_t crc, const void *buf, size_t n) {
    // libdeflate_crc32's signature has no error channel: a device failure leaves `crc` unchanged
    // and is reported through gzpx_last_status() (thread-local); gzpx_crc32_checked returns it.
    uint32_t out = crc;
    t_last_status = gzpx_crc32_checked(crc, buf, n, &out);
    return t_last_status == GZPX_OK ? out : crc;
}

int gzpx_last_status(void) { return t_last_status; }

// ---- Crc32::combine, Adler32::update / combine (src/check.rs:85-164): see gzpx_check.hip
uint32_t gzpx_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2) { return crc32_combine(crc1, crc2, len2); }

uint32_t gzpx_adler32_combine(uint32_t adler1, uint32_t adler2, uint64_t len2) {
    // zlib's adler32_combine (zlib 1.2.11 / zlib-ng 2.x: the same arithmetic), restated from its published algorithm
    const uint32_t base = 65521u;
    const uint32_t rem = (uint32_t)(len2 % base);
    uint32_t sum1 = adler1 & 0xFFFFu;
    uint32_t sum2 = (uint32_t)(((uint64_t)rem * sum1) % base);
    sum1 += (adler2 & 0xFFFFu) + base - 1;
    sum2 += ((adler1 >> 16) & 0xFFFFu) + ((adler2 >> 16) & 0xFFFFu) + base - rem;
    if (sum1 >= base) sum1 -= base;
    if (sum1 >= base) sum1 -= base;
    if (sum2 >= (base << 1)) sum2 -= (base << 1);
    if (sum2 >= base) sum2 -= base;
    return sum1 | (sum2 << 16);
}

int gzpx_adler32_checked(uint32_t adler, const void *buf, size_t n, uint32_t *out) {
    static std::mutex mu;
    static uint8_t *d_in = nullptr;
    static uint32_t *d_out3 = nullptr, *h_out3 = nullptr;
    static hipStream_t stream = nullptr;
    constexpr size_t kChunk = (size_t)64 << 20, kTiles = kChunk / 65536;
    if (!out || (!buf && n)) return GZPX_ERR_INVALID_ARG;
    *out = adler;
    if (n == 0) return GZPX_OK;
    std::lock_guard<std::mutex> lock(mu);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GZPX_ERR_NO_DEVICE;
    if (hipSetDevice(0) != hipSuccess) return GZPX_ERR_DEVICE;
    if (!stream) {
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIP_TRY(hipMalloc((void **)&d_in, kChunk));
        HIP_TRY(hipMalloc((void **)&d_out3, kTiles * 12));
        HIP_TRY(hipHostMalloc((void **)&h_out3, kTiles * 12, hipHostMallocDefault));
    }
    const uint32_t base = 65521u;
    uint32_t a = adler & 0xFFFFu, b = (adler >> 16) & 0xFFFFu;
    const uint8_t *p = (const uint8_t *)buf;
    while (n) {
        const size_t take = n < kChunk ? n : kChunk;
        const size_t tiles = (take + 65535) / 65536;
        HIP_TRY(hipMemcpyAsync(d_in, p, take, hipMemcpyHostToDevice, stream));
        launch_adler32(d_in, take, d_out3, stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_out3, d_out3, tiles * 12, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        for (size_t t = 0; t < tiles; t++) {  // a' = a + s1, b' = b + len a + s2
            const uint32_t s1 = h_out3[3 * t], s2 = h_out3[3 * t + 1], len = h_out3[3 * t + 2];
            b = (uint32_t)((b + (uint64_t)(len % base) * a + s2) % base);
            a = (a + s1) % base;
        }
        p += take;
        n -= take;
    }
    *out = a | (b << 16);
    return GZPX_OK;
}

uint32_t gzpx_adler32(uint32_t adler, const void *buf, size_t n) {
    // zlib's adler32 signature has no error channel: a device failure leaves `adler` unchanged and is reported through
    // gzpx_last_status() (thread-local), as with gzpx_crc32
    uint32_t out = adler;
    t_last_status = gzpx_adler32_checked(adler, buf, n, &out);
    return t_last_status == GZPX_OK ? out : adler;
}

// ---------------------------------------------------------------- ParDecompress side
namespace {

// One slab of blocks being inflated.  Everything a slab needs lives in its slot (block tables,
// staging), so up to kSlots slabs are in flight: copy-in of one, kernels of another, copy-out of
// a third, each on its own stream.
struct DSlot {
    int state = 0;  // 0 free, 1 submitted, 2 a thread is inside wait()
    uint64_t gen = 0;
    size_t cap_blocks = 0;
    uint64_t *d_offsets = nullptr, *d_out_off = nullptr;
    uint32_t *d_sizes = nullptr, *d_crc = nullptr;
    DBlockHost *d_blk = nullptr;
    DBlockHost *h_blk = nullptr;  // pinned
    uint32_t *h_crc = nullptr;    // pinned
    uint64_t *h_offsets = nullptr;  // pinned copies of the caller's arrays (the caller's may be pageable
    uint32_t *h_sizes = nullptr;    //  and must not be referenced after submit returns)
    uint64_t *h_total = nullptr;  // pinned
    uint32_t *h_summary = nullptr;  // pinned: k_dsummary's record (the first failing member, its checksums)
    bool have_blk = false;          // h_blk holds this launch's records (debug launches only)
    uint8_t *d_in = nullptr, *d_out = nullptr;
    size_t d_in_cap = 0, d_out_cap = 0;
    InflateScratch sc;  // match records / tile table of k_inflate_seg + k_lzcopy; sc.redo lives with the block tables
    size_t mlist_cap = 0, tfirst_cap = 0;
    hipEvent_t ev_h2d = nullptr, ev_kernels = nullptr, ev_done = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // around the inflate kernels (timing enabled)
    hipEvent_t ev_tm = nullptr;                   // behind k_inflate_seg (the decode / copy route)
    size_t nb = 0;
};

void dslot_free_tables(DSlot &c) {
    if (c.d_offsets) (void)hipFree(c.d_offsets);
    if (c.d_out_off) (void)hipFree(c.d_out_off);
    if (c.d_sizes) (void)hipFree(c.d_sizes);
    if (c.d_crc) (void)hipFree(c.d_crc);
    if (c.d_blk) (void)hipFree(c.d_blk);
    if (c.h_blk) (void)hipHostFree(c.h_blk);
    if (c.h_crc) (void)hipHostFree(c.h_crc);
    if (c.h_offsets) (void)hipHostFree(c.h_offsets);
    if (c.h_sizes) (void)hipHostFree(c.h_sizes);
    if (c.sc.redo) (void)hipFree(c.sc.redo);
    c.sc.redo = nullptr;
    c.d_offsets = c.d_out_off = nullptr;
    c.d_sizes = c.d_crc = nullptr;
    c.d_blk = c.h_blk = nullptr;
    c.h_crc = c.h_sizes = nullptr;
    c.h_offsets = nullptr;
    c.cap_blocks = 0;
}

int dslot_reserve(DSlot &c, size_t nb) {
    if (!c.ev_kernels) {
        HIP_TRY(hipEventCreateWithFlags(&c.ev_h2d, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c.ev_kernels, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c.ev_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreate(&c.ev_t0));
        HIP_TRY(hipEventCreate(&c.ev_t1));
        HIP_TRY(hipEventCreate(&c.ev_tm));
        HIP_TRY(hipHostMalloc((void **)&c.h_total, 64, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void **)&c.h_summary, 64, hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&c.sc.summary, 64));
        HIP_TRY(hipMemset(c.sc.summary, 0, 64));  // (word 15: k_inflate_seg's first-block hint, kept from launch to launch)
    }
    if (nb <= c.cap_blocks) return GZPX_OK;
    dslot_free_tables(c);
    const size_t cap = nb + nb / 4 + 64;
    HIP_TRY(hipMalloc((void **)&c.d_offsets, cap * 8));
    HIP_TRY(hipMalloc((void **)&c.d_out_off, (cap + 1) * 8));
    HIP_TRY(hipMalloc((void **)&c.d_sizes, cap * 4));
    HIP_TRY(hipMalloc((void **)&c.d_crc, cap * 4));
    HIP_TRY(hipMalloc((void **)&c.d_blk, cap * sizeof(DBlockHost)));
    HIP_TRY(hipMalloc((void **)&c.sc.redo, (cap + 2) * 4));
    HIP_TRY(hipHostMalloc((void **)&c.h_blk, cap * sizeof(DBlockHost), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&c.h_crc, cap * 4, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&c.h_offsets, cap * 8, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&c.h_sizes, cap * 4, hipHostMallocDefault));
    c.cap_blocks = cap;
    return GZPX_OK;
}

}  // namespace

struct gzpx_dctx {
    int device = 0;
    int format = 0;
    CrcConsts cc;
    hipStream_t stream = nullptr, s_h2d = nullptr, s_d2h = nullptr;
    hipEvent_t ev_dep = nullptr;
    DSlot slots[kSlots];
    uint64_t next_gen = 1;
    int n_cu = 0;
    int debug = 0;  // 1: instrumented k_inflate_seg / k_inflate, 2: instrumented k_lzcopy
    int route = kInflateRouteSeg;  // GZPX_INFLATE_ROUTE=wave / gzpx_dctx_set_route: k_inflate for every member
    int last_slot = -1;  // the slot of the last completed launch (timing / debug counters)
    size_t last_nb = 0;
    std::mutex mu;
    std::condition_variable cv_slot;
};

namespace {

int dsubmit_enqueue(gzpx_dctx *c, const uint8_t *host_in, const uint8_t *d_in, size_t in_len,
                    const uint64_t *offsets, const uint32_t *sizes, size_t nb, uint8_t *host_out, uint8_t *d_out,
                    size_t out_cap, hipStream_t after, bool block_for_slot, std::unique_lock<std::mutex> &lk,
                    uint64_t *ticket);

// As on the compress side (submit_locked): a submit that fails may already have put copies and kernels on the
// streams -- the copy-in still reads the caller's `in`, the copy-out may still write the caller's `out` -- and
// the slot stays free; nothing of them may still be running when the caller gets its buffers back.
int dsubmit_locked(gzpx_dctx *c, const uint8_t *host_in, const uint8_t *d_in, size_t in_len,
                   const uint64_t *offsets, const uint32_t *sizes, size_t nb, uint8_t *host_out, uint8_t *d_out,
                   size_t out_cap, hipStream_t after, bool block_for_slot, std::unique_lock<std::mutex> &lk,
                   uint64_t *ticket) {
    const int rc = dsubmit_enqueue(c, host_in, d_in, in_len, offsets, sizes, nb, host_out, d_out, out_cap, after,
                                   block_for_slot, lk, ticket);
    if (rc != GZPX_OK && rc != GZPX_ERR_BUSY && rc != GZPX_ERR_INVALID_ARG) {
        (void)hipStreamSynchronize(c->s_h2d);
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(c->s_d2h);
    }
    return rc;
}

int dsubmit_enqueue(gzpx_dctx *c, const uint8_t *host_in, const uint8_t *d_in, size_t in_len,
                    const uint64_t *offsets, const uint32_t *sizes, size_t nb, uint8_t *host_out, uint8_t *d_out,
                    size_t out_cap, hipStream_t after, bool block_for_slot, std::unique_lock<std::mutex> &lk,
                    uint64_t *ticket) {
    if (nb && (!offsets || !sizes || (!host_in && !d_in))) return GZPX_ERR_INVALID_ARG;
    if (nb > 0xFFFFFFFFull) return GZPX_ERR_INVALID_ARG;
    const uint32_t hdr_len = c->format == GZPX_FORMAT_BGZF ? 18 : 20;
    for (size_t b = 0; b < nb; b++)
        if (sizes[b] < hdr_len + 8 || offsets[b] > in_len || sizes[b] > in_len - offsets[b]) return GZPX_ERR_INVALID_ARG;
    int si = -1;
    for (;;) {
        for (int i = 0; i < kSlots; i++)
            if (c->slots[i].state == 0) {
                si = i;
                break;
            }
        if (si >= 0) break;
        if (!block_for_slot) return GZPX_ERR_BUSY;
        c->cv_slot.wait(lk);
    }
    DSlot &sl = c->slots[si];
    int rc = dslot_reserve(sl, nb ? nb : 1);
    if (rc != GZPX_OK) return rc;
    sl.nb = nb;
    hipStream_t stream = c->stream;
    if (nb) {
        if (host_in) {  // staging
            if (in_len + 16 > sl.d_in_cap) {
                if (sl.d_in) (void)hipFree(sl.d_in);
                sl.d_in = nullptr;
                sl.d_in_cap = 0;
                HIP_TRY(hipMalloc((void **)&sl.d_in, in_len + in_len / 8 + 4096));
                sl.d_in_cap = in_len + in_len / 8 + 4096;
            }
            if (out_cap + 16 > sl.d_out_cap) {
                if (sl.d_out) (void)hipFree(sl.d_out);
                sl.d_out = nullptr;
                sl.d_out_cap = 0;
                HIP_TRY(hipMalloc((void **)&sl.d_out, out_cap + out_cap / 8 + 4096));
                sl.d_out_cap = out_cap + out_cap / 8 + 4096;
            }
            HIP_TRY(hipMemcpyAsync(sl.d_in, host_in, in_len, hipMemcpyHostToDevice, c->s_h2d));
            d_in = sl.d_in;
            d_out = sl.d_out;
        } else if (after != stream && after != (hipStream_t)GZPX_STREAM_NONE) {  // (NULL = the legacy default stream, as on the compress side)
            HIP_TRY(hipEventRecord(c->ev_dep, after));
            HIP_TRY(hipStreamWaitEvent(stream, c->ev_dep, 0));
        }
        memcpy(sl.h_offsets, offsets, nb * 8);
        memcpy(sl.h_sizes, sizes, nb * 4);
        HIP_TRY(hipMemcpyAsync(sl.d_offsets, sl.h_offsets, nb * 8, hipMemcpyHostToDevice, c->s_h2d));
        HIP_TRY(hipMemcpyAsync(sl.d_sizes, sl.h_sizes, nb * 4, hipMemcpyHostToDevice, c->s_h2d));
        HIP_TRY(hipEventRecord(sl.ev_h2d, c->s_h2d));
        HIP_TRY(hipStreamWaitEvent(stream, sl.ev_h2d, 0));
        sl.sc.n_cu = c->n_cu;
        {
            uint64_t csum = 0;
            for (size_t b = 0; b < nb; b++) csum += sizes[b];
            sl.sc.big_members = nb && csum / nb >= 131072u ? 1 : 0;
        }
        if (c->route == kInflateRouteSeg) {  // scratch of the decode / copy pair, sized by what the caller can take
            const size_t need_m = inflate_mlist_bytes(out_cap, nb), need_t = inflate_tfirst_bytes(out_cap, nb);
            if (need_m > sl.mlist_cap) {
                if (sl.sc.mlist) (void)hipFree(sl.sc.mlist);
                sl.sc.mlist = nullptr;
                sl.mlist_cap = 0;
                HIP_TRY(hipMalloc(&sl.sc.mlist, need_m + need_m / 8));
                sl.mlist_cap = need_m + need_m / 8;
            }
            if (need_t > sl.tfirst_cap) {
                if (sl.sc.tfirst) (void)hipFree(sl.sc.tfirst);
                sl.sc.tfirst = nullptr;
                sl.tfirst_cap = 0;
                HIP_TRY(hipMalloc((void **)&sl.sc.tfirst, need_t + need_t / 8));
                sl.tfirst_cap = need_t + need_t / 8;
            }
        }
        launch_inflate(hdr_len, d_in, sl.d_offsets, sl.d_sizes, (uint32_t)nb, sl.d_blk, sl.d_out_off, d_out, out_cap,
                       sl.d_crc, c->cc, c->debug, sl.ev_t0, sl.ev_t1, stream, sl.sc, c->route, sl.ev_tm);
        HIP_TRY(hipGetLastError());
        // what the host needs of the members' records is k_dsummary's 48 bytes; the records themselves only for the
        // debug counters
        HIP_TRY(hipMemcpyAsync(sl.h_summary, sl.sc.summary, 48, hipMemcpyDeviceToHost, stream));
        sl.have_blk = c->debug != 0;
        if (c->debug) HIP_TRY(hipMemcpyAsync(sl.h_blk, sl.d_blk, nb * sizeof(DBlockHost), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(sl.h_total, sl.d_out_off + nb, 8, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipEventRecord(sl.ev_kernels, stream));
        if (host_out) {
            // Blocks land back to back from offset 0 and their sizes are the footers' ISIZE fields,
            // so the bytes to bring back are known now: the copy-out is enqueued behind the kernels
            // and needs no host round trip.
            uint64_t total = 0;
            for (size_t b = 0; b < nb; b++) {
                const uint8_t *f = host_in + offsets[b] + sizes[b] - 4;
                total += (uint64_t)f[0] | ((uint64_t)f[1] << 8) | ((uint64_t)f[2] << 16) | ((uint64_t)f[3] << 24);
            }
            if (total > out_cap) total = out_cap;  // (the kernels flag the blocks that do not fit)
            HIP_TRY(hipStreamWaitEvent(c->s_d2h, sl.ev_kernels, 0));
            if (total) HIP_TRY(hipMemcpyAsync(host_out, sl.d_out, (size_t)total, hipMemcpyDeviceToHost, c->s_d2h));
            HIP_TRY(hipEventRecord(sl.ev_done, c->s_d2h));
        } else {
            HIP_TRY(hipEventRecord(sl.ev_done, stream));
        }
    }
    sl.state = 1;
    sl.gen = c->next_gen++;
    *ticket = (sl.gen << 8) | (uint64_t)si;
    return GZPX_OK;
}

// `short_ok`: the libdeflate-shaped call offers its capacity as ISIZE and accepts fewer bytes (libdeflate
// with actual_out_nbytes_ret); a framed member that inflates to fewer bytes than its footer says is
// BadData (libdeflate SHORT_OUTPUT through decode_block, src/par/decompress.rs:162-186).
int dwait_ticket(gzpx_dctx *c, uint64_t ticket, size_t *out_len, gzpx_check_info *info, bool short_ok = false) {
    const int si = (int)(ticket & 0xFF);
    if (si >= kSlots) return GZPX_ERR_INVALID_ARG;
    DSlot &sl = c->slots[si];
    {
        std::lock_guard<std::mutex> lk(c->mu);
        if (sl.state != 1 || sl.gen != (ticket >> 8)) return GZPX_ERR_INVALID_ARG;
        sl.state = 2;
    }
    int rc = GZPX_OK;
    size_t produced = 0;
    if (sl.nb) {
        if (hipSetDevice(c->device) != hipSuccess || hipEventSynchronize(sl.ev_done) != hipSuccess) {
            rc = GZPX_ERR_DEVICE;
        } else {
            // the first failing block in stream order (src/par/decompress.rs:162-186), found on the device (k_dsummary)
            const uint32_t *q = sl.h_summary + (short_ok ? 4 : 0);
            if (q[0] != 0xFFFFFFFFu) {
                const uint32_t st = q[1];
                rc = st == 1 ? GZPX_ERR_BAD_DATA
                     : st == 2 ? GZPX_ERR_INSUFFICIENT_SPACE
                     : (st != 0 && !short_ok) ? GZPX_ERR_BAD_DATA  // 3: fewer bytes than ISIZE
                                              : GZPX_ERR_INVALID_CHECK;
                if (info) {
                    info->block = q[0];
                    info->found = q[2];
                    info->expected = q[3];
                }
            }
            if (rc == GZPX_OK) produced = (size_t)*sl.h_total;
        }
    }
    if (out_len) *out_len = produced;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        sl.state = 0;
        c->last_slot = si;
        c->last_nb = sl.nb;
    }
    c->cv_slot.notify_all();
    return rc;
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
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
    }
    if (const char *e = getenv("GZPX_INFLATE_ROUTE"))
        if (!strcmp(e, "wave")) c->route = kInflateRouteWave;
    for (unsigned l = 0; l < 10; l++) c->cc.pow64[l] = x2k(9 + l);
    c->cc.pow_tile = x2k(19);
    c->cc.pow_small = x2k(17);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->s_h2d, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->s_d2h, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_dep, hipEventDisableTiming) != hipSuccess) {
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
    if (c->s_h2d) (void)hipStreamSynchronize(c->s_h2d);
    if (c->s_d2h) (void)hipStreamSynchronize(c->s_d2h);
    for (DSlot &sl : c->slots) {
        dslot_free_tables(sl);
        if (sl.h_total) (void)hipHostFree(sl.h_total);
        if (sl.h_summary) (void)hipHostFree(sl.h_summary);
        if (sl.sc.summary) (void)hipFree(sl.sc.summary);
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.sc.mlist) (void)hipFree(sl.sc.mlist);
        if (sl.sc.tfirst) (void)hipFree(sl.sc.tfirst);
        for (hipEvent_t e : {sl.ev_h2d, sl.ev_kernels, sl.ev_done, sl.ev_t0, sl.ev_t1, sl.ev_tm})
            if (e) (void)hipEventDestroy(e);
    }
    if (c->ev_dep) (void)hipEventDestroy(c->ev_dep);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->s_h2d) (void)hipStreamDestroy(c->s_h2d);
    if (c->s_d2h) (void)hipStreamDestroy(c->s_d2h);
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

int gzpx_decompress_blocks_submit(gzpx_dctx *c, const uint8_t *in, size_t in_len, const uint64_t *offsets,
                                  const uint32_t *sizes, size_t n_blocks, uint8_t *out, size_t out_cap,
                                  uint64_t *ticket) {
    if (!c || (!in && in_len) || (!out && out_cap) || !ticket) return GZPX_ERR_INVALID_ARG;
    std::unique_lock<std::mutex> lk(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) return GZPX_ERR_DEVICE;
    return dsubmit_locked(c, in, nullptr, in_len, offsets, sizes, n_blocks, out, nullptr, out_cap, nullptr, false,
                          lk, ticket);
}

int gzpx_decompress_blocks_wait(gzpx_dctx *c, uint64_t ticket, size_t *out_len, gzpx_check_info *info) {
    if (!c) return GZPX_ERR_INVALID_ARG;
    return dwait_ticket(c, ticket, out_len, info);
}

int gzpx_decompress_blocks_device(gzpx_dctx *c, const void *d_in, size_t in_len, const uint64_t *offsets,
                                  const uint32_t *sizes, size_t n_blocks, void *d_out, size_t out_cap,
                                  size_t *out_len, gzpx_check_info *info, void *hip_stream) {
    if (!c || !out_len) return GZPX_ERR_INVALID_ARG;
    *out_len = 0;
    if (n_blocks == 0) return GZPX_OK;
    uint64_t ticket = 0;
    int rc;
    {
        std::unique_lock<std::mutex> lk(c->mu);
        if (hipSetDevice(c->device) != hipSuccess) return GZPX_ERR_DEVICE;
        rc = dsubmit_locked(c, nullptr, (const uint8_t *)d_in, in_len, offsets, sizes, n_blocks, nullptr,
                            (uint8_t *)d_out, out_cap, (hipStream_t)hip_stream, true, lk, &ticket);
    }
    if (rc != GZPX_OK) return rc;
    return dwait_ticket(c, ticket, out_len, info);
}

int gzpx_decompress_blocks(gzpx_dctx *c, const uint8_t *in, size_t in_len, const uint64_t *offsets,
                           const uint32_t *sizes, size_t n_blocks, uint8_t *out, size_t out_cap,
                           size_t *out_len, gzpx_check_info *info) {
    if (!c || (!in && in_len) || (!out && out_cap) || !out_len) return GZPX_ERR_INVALID_ARG;
    *out_len = 0;
    if (n_blocks == 0) return GZPX_OK;
    uint64_t ticket = 0;
    int rc;
    {
        std::unique_lock<std::mutex> lk(c->mu);
        if (hipSetDevice(c->device) != hipSuccess) return GZPX_ERR_DEVICE;
        uint8_t dummy = 0;
        rc = dsubmit_locked(c, in, nullptr, in_len, offsets, sizes, n_blocks, out ? out : &dummy, nullptr, out_cap,
                            nullptr, true, lk, &ticket);
    }
    if (rc != GZPX_OK) return rc;
    return dwait_ticket(c, ticket, out_len, info);
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
    if (cap > 0xFFFFFFFFull || n > 0xFFFFFFFFull - 64) return GZPX_ERR_UNSUPPORTED;  // one member: 32-bit sizes
    if (!d->ctx) {
        const int rc = gzpx_dctx_create(0, GZPX_FORMAT_MGZIP, &d->ctx);
        if (rc != GZPX_OK) return rc;
    }
    // the kernels work on framed members: wrap the raw stream with a header and a footer whose
    // ISIZE is the caller's capacity (libdeflate semantics: at most `cap` bytes may come out).  With
    // cap == 0 one byte of room is offered, so that a stream with output is told apart from an
    // empty one (libdeflate: INSUFFICIENT_SPACE).
    const size_t room = cap ? cap : 1;
    d->framed.assign(20 + n + 8, 0);
    memcpy(d->framed.data() + 20, in, n);
    uint8_t *f = d->framed.data() + 20 + n;
    const uint32_t isz = (uint32_t)room;
    f[4] = (uint8_t)isz;
    f[5] = (uint8_t)(isz >> 8);
    f[6] = (uint8_t)(isz >> 16);
    f[7] = (uint8_t)(isz >> 24);
    const uint64_t off = 0;
    const uint32_t size = (uint32_t)d->framed.size();
    std::vector<uint8_t> tmp(room);
    gzpx_dctx *c = d->ctx;
    uint64_t ticket = 0;
    int rc;
    size_t got = 0;
    {
        std::unique_lock<std::mutex> lk(c->mu);
        if (hipSetDevice(c->device) != hipSuccess) return GZPX_ERR_DEVICE;
        rc = dsubmit_locked(c, d->framed.data(), nullptr, d->framed.size(), &off, &size, 1, tmp.data(), nullptr, room,
                            nullptr, true, lk, &ticket);
    }
    if (rc != GZPX_OK) return rc;
    const DSlot &sl = c->slots[ticket & 0xFF];
    gzpx_check_info info = {0, 0, 0};
    rc = dwait_ticket(c, ticket, nullptr, &info, true);
    if (rc == GZPX_ERR_INVALID_CHECK) rc = GZPX_OK;  // a raw stream carries no checksum
    if (rc != GZPX_OK) return rc;
    {
        std::lock_guard<std::mutex> lk(c->mu);  // (this handle is single-threaded like libdeflate's: the
        got = sl.h_summary[8];                  //  slot has not been reused since the wait)
    }
    if (got > cap) return GZPX_ERR_INSUFFICIENT_SPACE;  // cap == 0 and the stream has output
    if (got) memcpy(out, tmp.data(), got);
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
    ctx->profiling = on == 2 ? 2 : on != 0 ? 1 : 0;
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

const char *gzpx_ctx_stage_kernel(const gzpx_ctx *ctx, int stage) {
    // the kernels behind stages 2 and 3 depend on the level (and, at level 1, on the block size)
    if (ctx && stage == 2) {
        const Config &c = ctx->dcfg;
        if (c.level <= 1) return (c.block_size <= kTile && !(c.debug & 2u)) ? "k_mparse" : "k_match";
        if (c.level >= 10) return "k_near_optimal";
        return c.lazy ? "k_match_hc+k_parse_lazy" : "k_match_hc+k_parse_hc";
    }
    if (ctx && stage == 3 && ctx->dcfg.level > 1) return "-";
    return gzpx_stage_name(stage);
}

int gzpx_debug_tokens(gzpx_ctx *ctx, size_t block, uint32_t *tokens, size_t max_tokens,
                      size_t *n_tokens, uint32_t *sub_first_token, size_t *n_sub) {
    if (!ctx || !n_tokens || block >= ctx->last_nb || ctx->crc_only) return GZPX_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (hipSetDevice(ctx->cfg.device) != hipSuccess) return GZPX_ERR_DEVICE;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(ctx->h_meta, ctx->scratch.meta + block, sizeof(BlockMeta), hipMemcpyDeviceToHost));
    const BlockMeta m = ctx->h_meta[0];
    const uint32_t max_sub = ctx->dcfg.max_sub;
    *n_tokens = m.ntok;
    if (n_sub) *n_sub = m.nsub;
    if (sub_first_token && m.nsub) {
        HIP_TRY(hipMemcpy(ctx->h_sub, ctx->scratch.sub + block * (size_t)max_sub,
                          (size_t)(m.nsub < max_sub ? m.nsub : max_sub) * sizeof(SubMeta), hipMemcpyDeviceToHost));
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

int gzpx_debug_redo_count(gzpx_ctx *ctx, uint32_t *count) {
    if (!ctx || !count || ctx->crc_only) return GZPX_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (hipSetDevice(ctx->cfg.device) != hipSuccess) return GZPX_ERR_DEVICE;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(count, ctx->scratch.redo, sizeof(uint32_t), hipMemcpyDeviceToHost));
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
    if (!ctx->last_nb || ctx->last_slot < 0) return GZPX_OK;
    const DSlot &sl = ctx->slots[ctx->last_slot];
    return hipEventElapsedTime(ms, sl.ev_t0, sl.ev_t1) == hipSuccess ? GZPX_OK : GZPX_ERR_DEVICE;
}

int gzpx_dctx_last_inflate_stage_ms(gzpx_dctx *ctx, float ms[2]) {
    if (!ctx || !ms) return GZPX_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    ms[0] = ms[1] = 0.0f;
    if (!ctx->last_nb || ctx->last_slot < 0 || ctx->route != kInflateRouteSeg) return GZPX_OK;
    const DSlot &sl = ctx->slots[ctx->last_slot];
    if (hipEventElapsedTime(&ms[0], sl.ev_t0, sl.ev_tm) != hipSuccess || hipEventElapsedTime(&ms[1], sl.ev_tm, sl.ev_t1) != hipSuccess)
        return GZPX_ERR_DEVICE;
    return GZPX_OK;
}

int gzpx_dctx_set_route(gzpx_dctx *ctx, int route) {
    if (!ctx || (route != GZPX_INFLATE_SEG && route != GZPX_INFLATE_WAVE)) return GZPX_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->route = route == GZPX_INFLATE_WAVE ? kInflateRouteWave : kInflateRouteSeg;
    return GZPX_OK;
}

int gzpx_dctx_last_redo_count(gzpx_dctx *ctx, uint32_t *count) {
    if (!ctx || !count) return GZPX_ERR_INVALID_ARG;
    *count = 0;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (ctx->last_slot < 0 || !ctx->slots[ctx->last_slot].sc.redo || ctx->route != kInflateRouteSeg) return GZPX_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return GZPX_ERR_DEVICE;
    if (hipMemcpy(count, ctx->slots[ctx->last_slot].sc.redo, 4, hipMemcpyDeviceToHost) != hipSuccess) return GZPX_ERR_DEVICE;
    return GZPX_OK;
}

int gzpx_debug_inflate(gzpx_dctx *ctx, int enable, uint64_t sums[8]) {
    if (!ctx) return GZPX_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->debug = enable < 0 ? 0 : enable > 2 ? 1 : enable;
    if (sums) {
        for (int k = 0; k < 8; k++) sums[k] = 0;
        if (ctx->last_slot >= 0 && ctx->slots[ctx->last_slot].have_blk)
            for (size_t b = 0; b < ctx->last_nb; b++)
                for (int k = 0; k < 8; k++) sums[k] += ctx->slots[ctx->last_slot].h_blk[b].cyc[k];
    }
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
        case GZPX_ERR_BUSY: return "every slab slot of the context is in flight: wait for one first";
        default: return "unknown error";
    }
}

const char *gzpx_device_name(const gzpx_ctx *ctx) { return ctx ? ctx->devname : ""; }
const char *gzpx_version(void) { return "gzpx 0.1 (gfx950)"; }
#ifndef GZPX_BUILD_ID
#define GZPX_BUILD_ID "unknown"
#endif
const char *gzpx_build_id(void) { return GZPX_BUILD_ID; }

}  // extern "C"
