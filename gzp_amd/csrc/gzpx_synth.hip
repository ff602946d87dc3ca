// gzpx_synth.hip -- on-device generator of the synthetic FASTQ workload of BASELINE.json configs[3]
// ("8xMI355X input shard ... 32 GiB synthetic FASTQ"; SURVEY.md 8(d) "Config 4 input").  Measurement
// support, not part of the encode path: bench.py --workload fastq and the config-4 tests fill each
// rank's 4 GiB share of the stream directly in HBM instead of generating and copying it from the host.
//
// The stream is defined page by page (64 KiB pages, each from its own splitmix64 state
// seed ^ page * 0xD6E8FEB86659FD93) so that any byte range can be produced independently: one thread
// per page, records written until the page is full (the last one is cut at the page end).
//   record k of page c, read number r = c * 512 + k + 1:
//     "@SRR" %07d(seed % 10^7) "." r " " r "/1\n"
//     L = 100 + next() % 51 bases, 5 per next() (12 bits each: low 10 bits zero -> 'N', else
//     "ACGT"[bits 10..11]), "\n+\n", L qualities from a 4-state chain over "F:,#" (start 'F', 21 steps
//     per next(), 3 bits each through the `step` table below), "\n".
// The tests compare the kernel byte for byte with an independent CPU statement of the same rules.
#include <hip/hip_runtime.h>

#include "../../include/gzpx.h"

namespace {

constexpr uint32_t kPage = 65536;

__device__ __forceinline__ uint64_t next64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// `p` walks the page (0..kPage); bytes whose stream position falls into [lo, hi) are stored
struct Emitter {
    uint8_t *out;
    uint64_t page_base, lo, hi;
    uint32_t p;
    __device__ __forceinline__ void put(uint32_t b) {
        if (p < kPage) {
            const uint64_t pos = page_base + p;
            if (pos >= lo && pos < hi) out[pos - lo] = (uint8_t)b;
        }
        p++;
    }
    __device__ void dec(uint64_t v) {
        uint32_t nd = 1;
        for (uint64_t t = v; t >= 10; t /= 10) nd++;
        uint64_t div = 1;
        for (uint32_t i = 1; i < nd; i++) div *= 10;
        for (uint32_t i = 0; i < nd; i++, div /= 10) put('0' + (uint32_t)((v / div) % 10));
    }
};

__global__ __launch_bounds__(64) void k_synth_fastq(uint8_t *__restrict__ out, uint64_t lo, uint64_t n, uint64_t seed,
                                                    uint64_t first_page, uint64_t n_pages) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pages) return;
    const uint64_t c = first_page + i;
    uint64_t s = seed ^ (c * 0xD6E8FEB86659FD93ull);
    Emitter e{out, c * kPage, lo, lo + n, 0};
    const uint32_t run = (uint32_t)(seed % 10000000ull);
    // next state of the quality chain by (state, 3 random bits), 2 bits per entry
    const uint32_t step[4] = {0x4000u, 0x9540u, 0xEA94u, 0xFFFAu};
    for (uint32_t k = 0; e.p < kPage; k++) {
        const uint64_t readno = c * 512 + k + 1;
        e.put('@');
        e.put('S');
        e.put('R');
        e.put('R');
        for (uint32_t d = 1000000; d; d /= 10) e.put('0' + (run / d) % 10);
        e.put('.');
        e.dec(readno);
        e.put(' ');
        e.dec(readno);
        e.put('/');
        e.put('1');
        e.put('\n');
        const uint32_t L = 100 + (uint32_t)(next64(s) % 51);
        for (uint32_t b = 0; b < L; b += 5) {
            uint64_t v = next64(s);
            for (uint32_t j = 0; j < 5 && b + j < L; j++, v >>= 12)
                e.put((v & 1023) == 0 ? 'N' : (0x54474341u >> (8 * ((v >> 10) & 3))) & 0xFFu);  // "ACGT"
        }
        e.put('\n');
        e.put('+');
        e.put('\n');
        uint32_t st = 0;
        for (uint32_t b = 0; b < L; b += 21) {
            uint64_t v = next64(s);
            for (uint32_t j = 0; j < 21 && b + j < L; j++, v >>= 3) {
                st = (step[st] >> (2 * (v & 7))) & 3u;
                e.put((0x232C3A46u >> (8 * st)) & 0xFFu);  // "F:,#"
            }
        }
        e.put('\n');
    }
}

// BASELINE configs[2] input ("4 GiB /dev/urandom-seeded ASCII"): byte i of the stream is
// 0x20 + (splitmix64 output i >> 56) % 95 -- gzp_amd/synth.py ascii_random states the same on the host.
__global__ __launch_bounds__(256) void k_synth_ascii(uint8_t *__restrict__ out, uint64_t lo, uint64_t n, uint64_t seed) {
    const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;  // 4 bytes per thread
    if (i0 >= n) return;
    uint32_t v[4];
    for (uint32_t k = 0; k < 4; k++) {
        uint64_t z = seed + (lo + i0 + k + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        v[k] = 0x20u + (uint32_t)(z >> 56) % 95u;
    }
    if (i0 + 4 <= n && (((uintptr_t)(out + i0)) & 3u) == 0) {
        *(uint32_t *)(out + i0) = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
    } else {
        for (uint32_t k = 0; k < 4 && i0 + k < n; k++) out[i0 + k] = (uint8_t)v[k];
    }
}

}  // namespace

extern "C" int gzpx_synth_ascii_device(void *d_out, uint64_t stream_offset, uint64_t n, uint64_t seed, void *hip_stream) {
    if (!d_out && n) return GZPX_ERR_INVALID_ARG;
    if (n == 0) return GZPX_OK;
    const uint64_t threads = (n + 3) / 4;
    hipLaunchKernelGGL(k_synth_ascii, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                       (uint8_t *)d_out, stream_offset, n, seed);
    return hipGetLastError() == hipSuccess ? GZPX_OK : GZPX_ERR_DEVICE;
}

extern "C" int gzpx_synth_fastq_device(void *d_out, uint64_t stream_offset, uint64_t n, uint64_t seed,
                                       void *hip_stream) {
    if (!d_out && n) return GZPX_ERR_INVALID_ARG;
    if (n == 0) return GZPX_OK;
    const uint64_t first_page = stream_offset / kPage;
    const uint64_t n_pages = (stream_offset + n - 1) / kPage - first_page + 1;
    hipLaunchKernelGGL(k_synth_fastq, dim3((unsigned)((n_pages + 63) / 64)), dim3(64), 0, (hipStream_t)hip_stream,
                       (uint8_t *)d_out, stream_offset, n, seed, first_page, n_pages);
    return hipGetLastError() == hipSuccess ? GZPX_OK : GZPX_ERR_DEVICE;
}
