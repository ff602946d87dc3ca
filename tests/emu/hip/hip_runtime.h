// tests/emu/hip/hip_runtime.h -- a tiny SIMT emulator for kernel-LOGIC tests on a CPU.
//
// TEST INFRASTRUCTURE ONLY.  The product (gzp_amd/lib/libgzpx.so) is built by hipcc for gfx950
// and never sees this file.  tests/emu/build_emu.py compiles the very same sources
// (gzp_amd/csrc/*.hip, *.cpp) with g++ and `-I tests/emu`, so that `#include
// <hip/hip_runtime.h>` resolves here; the resulting tests/emu/libgzpx_emu.so lets the
// `-m "not gpu"` suite run every kernel's integer logic against the oracle without a GPU.
// It checks indexing / control flow / arithmetic -- not timing, LDS limits or memory-model
// behaviour (those are what the `-m gpu` tests on the real library are for).
//
// Model: one workgroup at a time; every thread is a fiber (own stack, cooperative switch);
// __syncthreads() and the wave-wide collectives (__ballot, __shfl, ...) are rendezvous points
// for the 64 lanes of a wave / all threads of the workgroup.  Collectives must be reached by
// all live lanes of the wave (convergent code) -- a divergent collective is reported as a
// deadlock instead of silently mis-executing.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define GZPX_EMU 1

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu {
    unsigned x, y, z;
};

struct uint2 {
    unsigned x, y;
};
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
struct uint4 {
    unsigned x, y, z, w;
};
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

namespace emu {
struct ThreadCtx {
    uint3_emu tid;
    uint3_emu bid;
    dim3 bdim;
    dim3 gdim;
    unsigned flat;  // flat thread id in the workgroup
};
extern ThreadCtx *cur;  // the running fiber's context
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
void sync_threads();
int sync_threads_or(int pred);
int sync_threads_count(int pred);
unsigned long long wave_ballot(int pred);
uint64_t wave_exchange(uint64_t v, int src_lane, int mode, int width);  // mode 0: idx
void set_lds_poison(bool on);
void yield_now();  // let the other fibers run (spin-wait loops)
}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::cur->bid)
#define blockDim (emu::cur->bdim)
#define gridDim (emu::cur->gdim)
#define warpSize 64

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__ __restrict
#endif
#define __constant__ static const

// ---------------------------------------------------------------- barriers / collectives
static inline void __syncthreads() { emu::sync_threads(); }
static inline void __threadfence() {}        // (fibers are cooperative: program order is memory order)
static inline void __threadfence_block() {}
static inline int __syncthreads_or(int p) { return emu::sync_threads_or(p); }
static inline int __syncthreads_count(int p) { return emu::sync_threads_count(p); }
static inline unsigned long long __ballot(int p) { return emu::wave_ballot(p); }
static inline int __any(int p) { return emu::wave_ballot(p) != 0; }
static inline int __all(int p) { return emu::wave_ballot(!p) == 0; }

template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
    uint64_t raw = 0;
    static_assert(sizeof(T) <= 8, "shfl size");
    memcpy(&raw, &v, sizeof(T));
    raw = emu::wave_exchange(raw, src, 0, width);
    T r;
    memcpy(&r, &raw, sizeof(T));
    return r;
}
template <typename T>
static inline T __shfl_up(T v, unsigned delta, int width = 64) {
    (void)width;
    int lane = (int)(emu::cur->flat & 63);
    int src = lane - (int)delta;
    return __shfl(v, src < 0 ? lane : src);
}
template <typename T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    (void)width;
    int lane = (int)(emu::cur->flat & 63);
    int src = lane + (int)delta;
    return __shfl(v, src > 63 ? lane : src);
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width;
    int lane = (int)(emu::cur->flat & 63);
    return __shfl(v, (lane ^ mask) & 63);
}
static inline long long clock64() { return 0; }
static inline unsigned __lane_id() { return emu::cur->flat & 63; }

// ---------------------------------------------------------------- integer intrinsics
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline unsigned __brev(unsigned v) {
    unsigned r = 0;
    for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i);
    return r;
}
static inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned sel) {
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (unsigned)(v >> (8 * (sel & 3)));
}
static inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sel) {
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (unsigned)(v >> (sel & 31));
}
// wave-level ordering points: a rendezvous of the wave's live lanes in the emulator
static inline void __builtin_amdgcn_wave_barrier() { (void)emu::wave_ballot(0); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static inline void __builtin_amdgcn_s_sleep(int) { emu::yield_now(); }
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
template <typename T>
static inline T __hip_atomic_load(const T *p, int, int) { return *(const volatile T *)p; }
template <typename T, typename U>
static inline void __hip_atomic_store(T *p, U v, int, int) { *(volatile T *)p = (T)v; }
static inline unsigned __builtin_amdgcn_readfirstlane(unsigned v) { return __shfl(v, 0); }
static inline int __builtin_amdgcn_readlane(int v, int lane) { return __shfl(v, lane); }
static inline int __builtin_amdgcn_ds_bpermute(int addr, int v) { return __shfl(v, (addr >> 2) & 63); }
// DPP controls used by the kernels: row_shr:n (0x110 + n), row_bcast15 (0x142), row_bcast31 (0x143);
// lanes without a valid source, or in a row that row_mask disables, keep `old`
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask,
                                              bool bound_ctrl) {
    (void)bank_mask;
    (void)bound_ctrl;
    const int lane = (int)(emu::cur->flat & 63);
    int from = lane;
    bool valid = false;
    if (ctrl > 0x110 && ctrl <= 0x11F) {
        const int n = ctrl - 0x110;
        valid = (lane & 15) >= n;
        from = lane - n;
    } else if (ctrl == 0x138) {  // wave_shr:1
        valid = lane >= 1;
        from = lane - 1;
    } else if (ctrl == 0x142) {
        valid = lane >= 16;
        from = (lane & ~15) - 1;
    } else if (ctrl == 0x143) {
        valid = lane >= 32;
        from = 31;
    }
    if (!((row_mask >> (lane >> 4)) & 1)) valid = false;
    const int got = __shfl(src, valid ? from : lane);
    return valid ? got : old;
}
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
static inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }

// ---------------------------------------------------------------- atomics (fibers are cooperative)
template <typename T>
static inline T atomicAdd(T *p, T v) {
    T o = *p;
    *p = o + v;
    return o;
}
template <typename T>
static inline T atomicOr(T *p, T v) {
    T o = *p;
    *p = o | v;
    return o;
}
template <typename T>
static inline T atomicAnd(T *p, T v) {
    T o = *p;
    *p = o & v;
    return o;
}
template <typename T>
static inline T atomicMax(T *p, T v) {
    T o = *p;
    if (v > o) *p = v;
    return o;
}
template <typename T>
static inline T atomicMin(T *p, T v) {
    T o = *p;
    if (v < o) *p = v;
    return o;
}
template <typename T>
static inline T atomicExch(T *p, T v) {
    T o = *p;
    *p = v;
    return o;
}

// ---------------------------------------------------------------- host runtime subset
typedef int hipError_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
#define hipErrorOutOfMemory 2
#define hipErrorUnknown 999
#define hipErrorNoDevice 100
typedef struct emu_stream *hipStream_t;
typedef struct emu_event {
    double t;
} *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost,
                     hipMemcpyDeviceToDevice, hipMemcpyDefault };
#define hipStreamNonBlocking 1
#define hipHostMallocDefault 0
#define hipEventDefault 0
#define hipEventDisableTiming 2

struct hipDeviceProp_t {
    char name[256];
    int multiProcessorCount;
    size_t totalGlobalMem;
    char gcnArchName[256];
};

hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags = 0);
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st = 0);
static inline hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t st = 0) {
    return hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, st);
}
static inline hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 0; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st = 0);
hipError_t hipStreamCreate(hipStream_t *s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
static inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int) { return hipStreamCreateWithFlags(s, flags); }
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = 0);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipGetDeviceCount(int *n);
static inline hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) { *free_b = (size_t)1 << 30; *total_b = (size_t)1 << 30; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
const char *hipGetErrorString(hipError_t e);

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); })
