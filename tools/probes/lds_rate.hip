// Probe: cost of LDS ops issued by ONE wave with random (hashed) addresses in a 128 KiB table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(64) void probe(unsigned *out, long long *cyc, unsigned iters) {
    __shared__ unsigned tab[32768];
    unsigned lane = threadIdx.x;
    for (unsigned i = lane; i < 32768; i += 64) tab[i] = 0;
    __syncthreads();
    unsigned x = lane * 2654435761u + 12345u, acc = 0;
    long long t0 = clock64();
    for (unsigned it = 0; it < iters; it++) {
        unsigned a[8], r[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            x = x * 1664525u + 1013904223u;
            a[k] = MODE >= 10 ? ((it * 8 + k) * 64 + lane) & 32767u : (x >> 17);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (MODE % 10 == 0) r[k] = atomicMax(&tab[a[k]], (it * 8 + k) << 16);
            if (MODE % 10 == 1) { atomicMax(&tab[a[k]], (it * 8 + k) << 16); r[k] = 0; }
            if (MODE % 10 == 2) r[k] = tab[a[k]];
            if (MODE % 10 == 3) { tab[a[k]] = it + k; r[k] = 0; }
            if (MODE % 10 == 4) r[k] = atomicAdd(&tab[a[k]], 1u);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc += r[k];
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + lane] = acc + tab[lane];
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name) {
    unsigned *d_out; long long *d_cyc;
    hipMalloc((void **)&d_out, 256 * 64 * 4);
    hipMalloc((void **)&d_cyc, 256 * 8);
    const unsigned iters = 2000;
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(64), 0, 0, d_out, d_cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> c(256);
    hipMemcpy(c.data(), d_cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : c) avg += v; avg /= 256;
    printf("%-34s %8.1f cycles per wave-instruction (8 in flight)\n", name, avg / (iters * 8.0));
    hipFree(d_out); hipFree(d_cyc);
}

int main() {
    run<0>("ds_max_rtn_u32 random");
    run<1>("ds_max_u32 (no return) random");
    run<2>("ds_read_b32 random");
    run<3>("ds_write_b32 random");
    run<4>("ds_add_rtn_u32 random");
    run<10>("ds_max_rtn_u32 linear");
    run<11>("ds_max_u32 linear");
    run<12>("ds_read_b32 linear");
    run<13>("ds_write_b32 linear");
    return 0;
}
