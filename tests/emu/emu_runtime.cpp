// tests/emu/emu_runtime.cpp -- fiber scheduler + host-runtime stubs of the CPU SIMT emulator.
// TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h in this directory).
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <mutex>
#include <vector>

namespace emu {

ThreadCtx *cur = nullptr;

namespace {

constexpr size_t kStackBytes = 256 * 1024;
constexpr int kMaxThreads = 1024;

struct Fiber {
    void *sp = nullptr;
    char *stack = nullptr;
    bool done = true;
    ThreadCtx ctx;
};

struct WaveState {
    int live = 0;     // lanes that have not returned from the kernel
    int arrived = 0;  // lanes waiting at the current collective
    unsigned gen = 0;
    uint64_t in[64];
    int src[64];
    uint64_t out[64];
    int kind = -1;  // 0 ballot, 1 exchange
    unsigned long long ballot = 0;
    bool present[64];
};

struct GroupState {
    int live = 0;
    int arrived = 0;
    unsigned gen = 0;
    int acc_or = 0, acc_count = 0;
    int res_or = 0, res_count = 0;
};

Fiber fibers[kMaxThreads];
WaveState waves[kMaxThreads / 64];
GroupState group;
void *sched_sp = nullptr;
int cur_fiber = -1;
int n_threads = 0;
unsigned long progress = 0;
const std::function<void()> *cur_body = nullptr;

extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

void yield_to_scheduler() {
    Fiber &f = fibers[cur_fiber];
    emu_switch(&f.sp, sched_sp);
    cur = &fibers[cur_fiber].ctx;
}

void complete_wave(WaveState &w) {
    if (w.kind == 0) {
        unsigned long long m = 0;
        for (int l = 0; l < 64; l++)
            if (w.present[l] && w.in[l]) m |= 1ull << l;
        w.ballot = m;
    } else {
        for (int l = 0; l < 64; l++)
            if (w.present[l]) {
                int s = w.src[l] & 63;
                w.out[l] = w.present[s] ? w.in[s] : w.in[l];
            }
    }
    for (int l = 0; l < 64; l++) w.present[l] = false;
    w.arrived = 0;
    w.gen++;
    progress++;
}

void complete_group() {
    group.res_or = group.acc_or;
    group.res_count = group.acc_count;
    group.acc_or = 0;
    group.acc_count = 0;
    group.arrived = 0;
    group.gen++;
    progress++;
}

void fiber_exit_bookkeeping(int fid) {
    WaveState &w = waves[fid / 64];
    w.live--;
    group.live--;
    if (w.live > 0 && w.arrived == w.live) complete_wave(w);
    if (group.live > 0 && group.arrived == group.live) complete_group();
    progress++;
}

extern "C" void emu_fiber_main() {
    (*cur_body)();
    int fid = cur_fiber;
    fibers[fid].done = true;
    fiber_exit_bookkeeping(fid);
    void *dummy;
    emu_switch(&dummy, sched_sp);
    abort();  // never resumed
}

void prepare_fiber(Fiber &f) {
    if (!f.stack) {
        f.stack = (char *)mmap(nullptr, kStackBytes, PROT_READ | PROT_WRITE,
                               MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (f.stack == (char *)MAP_FAILED) {
            perror("emu: mmap stack");
            abort();
        }
    }
    uintptr_t top = ((uintptr_t)f.stack + kStackBytes) & ~(uintptr_t)15;
    uint64_t *sp = (uint64_t *)top;
    *--sp = 0;                           // fake return address of emu_fiber_main
    *--sp = (uint64_t)&emu_fiber_main;   // `ret` target of the first switch
    for (int i = 0; i < 6; i++) *--sp = 0;  // rbp rbx r12 r13 r14 r15
    f.sp = sp;
    f.done = false;
}

int wave_collective(int kind, uint64_t v, int src) {
    int fid = cur_fiber;
    WaveState &w = waves[fid / 64];
    int lane = fid & 63;
    if (w.arrived > 0 && w.kind != kind) {
        fprintf(stderr, "emu: lanes of one wave reached different collectives (divergent code)\n");
        abort();
    }
    w.kind = kind;
    w.in[lane] = v;
    w.src[lane] = src;
    w.present[lane] = true;
    w.arrived++;
    unsigned my_gen = w.gen;
    if (w.arrived == w.live)
        complete_wave(w);
    else
        while (w.gen == my_gen) yield_to_scheduler();
    return lane;
}

}  // namespace

unsigned long long wave_ballot(int pred) {
    int fid = cur_fiber;
    wave_collective(0, pred ? 1 : 0, 0);
    return waves[fid / 64].ballot;
}

uint64_t wave_exchange(uint64_t v, int src_lane, int, int) {
    int fid = cur_fiber;
    int lane = wave_collective(1, v, src_lane);
    return waves[fid / 64].out[lane];
}

static void group_barrier(int pred_or, int pred_count) {
    group.acc_or |= pred_or;
    group.acc_count += pred_count;
    group.arrived++;
    unsigned my_gen = group.gen;
    if (group.arrived == group.live)
        complete_group();
    else
        while (group.gen == my_gen) yield_to_scheduler();
}

void sync_threads() { group_barrier(0, 0); }
int sync_threads_or(int pred) {
    group_barrier(pred != 0, 0);
    return group.res_or;
}
int sync_threads_count(int pred) {
    group_barrier(0, pred != 0);
    return group.res_count;
}
void set_lds_poison(bool) {}
static unsigned long spin_yields = 0;
void yield_now() {
    spin_yields++;
    yield_to_scheduler();
}

void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
    // one emulated "device": kernels from different host threads (device lanes) run one at a time
    static std::mutex launch_mu;
    std::lock_guard<std::mutex> launch_lock(launch_mu);
    n_threads = (int)(block.x * block.y * block.z);
    if (n_threads <= 0 || n_threads > kMaxThreads) {
        fprintf(stderr, "emu: unsupported block size %d\n", n_threads);
        abort();
    }
    cur_body = &body;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                int n_waves = (n_threads + 63) / 64;
                for (int w = 0; w < n_waves; w++) {
                    waves[w] = WaveState();
                    for (int l = 0; l < 64; l++) waves[w].present[l] = false;
                }
                group = GroupState();
                group.live = n_threads;
                for (int t = 0; t < n_threads; t++) {
                    Fiber &f = fibers[t];
                    prepare_fiber(f);
                    f.ctx.flat = (unsigned)t;
                    f.ctx.tid = {(unsigned)t % block.x, ((unsigned)t / block.x) % block.y,
                                 (unsigned)t / (block.x * block.y)};
                    f.ctx.bid = {bx, by, bz};
                    f.ctx.bdim = block;
                    f.ctx.gdim = grid;
                    waves[t / 64].live++;
                }
                int remaining = n_threads;
                while (remaining > 0) {
                    unsigned long before = progress;
                    for (int t = 0; t < n_threads; t++) {
                        Fiber &f = fibers[t];
                        if (f.done) continue;
                        cur_fiber = t;
                        cur = &f.ctx;
                        emu_switch(&sched_sp, f.sp);
                        if (f.done) remaining--;
                    }
                    static unsigned idle_passes = 0;
                    if (progress != before) idle_passes = 0;
                    if (progress == before && remaining > 0 && spin_yields) {
                        // only spinners ran: legal while another fiber still has to reach its
                        // hand-off, so allow many passes before calling it a deadlock
                        spin_yields = 0;
                        if (++idle_passes < 100000) continue;
                    }
                    spin_yields = 0;
                    if (progress == before && remaining > 0) {
                        fprintf(stderr,
                                "emu: deadlock in block (%u,%u,%u): %d threads blocked "
                                "(divergent barrier/collective?)\n", bx, by, bz, remaining);
                        abort();
                    }
                }
            }
    cur = nullptr;
    cur_fiber = -1;
}

}  // namespace emu

// ---------------------------------------------------------------- host runtime stubs
static double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

hipError_t hipMalloc(void **p, size_t n) {
    *p = malloc(n ? n : 1);
    if (*p) memset(*p, 0xA5, n);  // "device memory" starts as garbage
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p) {
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
    *p = malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p) {
    free(p);
    return hipSuccess;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
    memmove(d, s, n);
    return hipSuccess;
}
// Failure injection for the error-path tests: the n-th asynchronous copy from now fails (0 = off).  The
// stream-synchronise counter tells whether the library drained its streams behind the failure.
static long inject_memcpy_async = 0;
static long stream_syncs = 0;
extern "C" void emu_fail_nth_memcpy_async(long nth) { inject_memcpy_async = nth; }
extern "C" long emu_stream_sync_count() { return stream_syncs; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
    if (inject_memcpy_async > 0 && --inject_memcpy_async == 0) return hipErrorUnknown;
    memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemset(void *d, int v, size_t n) {
    memset(d, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) {
    memset(d, v, n);
    return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t *s) {
    *s = nullptr;
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
    *s = nullptr;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) {
    stream_syncs++;
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) {
    *e = new emu_event{0.0};
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->t = now_ms();
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = (float)(b->t - a->t);
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int *d) {
    *d = 0;
    return hipSuccess;
}
hipError_t hipGetDeviceCount(int *n) {
    *n = 1;
    return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "gzpx CPU SIMT emulator");
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "emu");
    p->multiProcessorCount = 3;  // (persistent kernels: three workgroups share the blocks)
    return hipSuccess;
}
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipError(emu)"; }
