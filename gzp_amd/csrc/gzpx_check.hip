// gzpx_check.hip -- the checks of gzp's streaming formats as helpers behind the C ABI (src/check.rs:85-164):
// Adler32::update (zlib's adler32 over a buffer) on the device, Adler32::combine / Crc32::combine as plain arithmetic.
// (Crc32::update is gzpx_crc32, gzpx_api.cpp.)  They have no caller inside this library: gzp uses them for its Gzip /
// Zlib formats, whose encoder (zlib-ng) has no binary in this image to pin its output on; the checks themselves are
// pinned on Python's zlib module (tests/test_checks.py, tests/test_gpu_checks.py).
//
// Adler-32 (RFC 1950): a = 1 + sum d_i, b = sum of the running a, both mod 65521.  For a segment of n bytes entered
// with (a0, b0):  a = a0 + s1,  b = b0 + n a0 + s2  with  s1 = sum d_i,  s2 = sum (n - i + 1) d_i  (i = 1..n) -- so a
// buffer is cut into pieces, every piece gives (s1, s2, n) on its own, and two neighbours X, Y combine to
// (s1x + s1y, s2x + |Y| s1x + s2y, |X| + |Y|).  One workgroup per 64 KiB tile, 256 bytes per thread, a log-tree over
// the 256 threads; the tiles of a buffer are combined on the host (a few dozen multiplications per MiB).
#include <hip/hip_runtime.h>

#include "../../include/gzpx.h"

namespace gzpx {

constexpr uint32_t kAdlerBase = 65521u;
constexpr uint32_t kAdlerTile = 65536u, kAdlerThreads = 256u, kAdlerSeg = kAdlerTile / kAdlerThreads;

__global__ __launch_bounds__(kAdlerThreads) void k_adler32(const uint8_t *__restrict__ in, uint64_t n, uint32_t *__restrict__ out3) {
    __shared__ uint32_t s1s[kAdlerThreads], s2s[kAdlerThreads], ns[kAdlerThreads];
    const uint32_t tid = threadIdx.x;
    const uint64_t tile0 = (uint64_t)blockIdx.x * kAdlerTile;
    const uint64_t lo = tile0 + (uint64_t)tid * kAdlerSeg;
    const uint32_t len = lo >= n ? 0u : (uint32_t)(n - lo < kAdlerSeg ? n - lo : kAdlerSeg);
    uint32_t s1 = 0, s2 = 0;
    for (uint32_t i = 0; i < len; i++) {  // (256 bytes: sums stay below 2^24)
        s1 += in[lo + i];
        s2 += s1;
    }
    s1s[tid] = s1 % kAdlerBase;
    s2s[tid] = s2 % kAdlerBase;
    ns[tid] = len;
    __syncthreads();
    for (uint32_t stride = 1; stride < kAdlerThreads; stride <<= 1) {
        const uint32_t left = 2 * stride * tid;
        const bool act = left + stride < kAdlerThreads;
        uint32_t a = 0, b = 0, m = 0;
        if (act) {
            const uint32_t ny = ns[left + stride];
            a = (s1s[left] + s1s[left + stride]) % kAdlerBase;
            b = (uint32_t)((s2s[left] + (uint64_t)(ny % kAdlerBase) * s1s[left] + s2s[left + stride]) % kAdlerBase);
            m = ns[left] + ny;
        }
        __syncthreads();
        if (act) {
            s1s[left] = a;
            s2s[left] = b;
            ns[left] = m;
        }
        __syncthreads();
    }
    if (tid == 0) {
        out3[3 * blockIdx.x + 0] = s1s[0];
        out3[3 * blockIdx.x + 1] = s2s[0];
        out3[3 * blockIdx.x + 2] = ns[0];
    }
}

void launch_adler32(const uint8_t *d_in, uint64_t n, uint32_t *d_out3, hipStream_t stream) {
    const uint32_t tiles = (uint32_t)((n + kAdlerTile - 1) / kAdlerTile);
    if (tiles) hipLaunchKernelGGL(k_adler32, dim3(tiles), dim3(kAdlerThreads), 0, stream, d_in, n, d_out3);
}

}  // namespace gzpx
