// Probe: ds_mskor_rtn_b32 (MEM = (MEM & ~mask) | data, returns the old dword) on one half of an LDS dword -- (a) does the
// LDS apply the same-address operations of ONE wave instruction in ascending lane order (as it does for ds_max_rtn_u32,
// lds_atomic_order.hip), so that the returned half is the lane's predecessor in its 16-bit bucket?  (b) does the other
// half of the dword stay intact?  What a 2^16-bucket table of 16-bit entries in 128 KiB of LDS would need (a one-pass
// hash4 k_candidates).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned mskor_rtn(unsigned *p, unsigned mask, unsigned data) {
    unsigned old;
    const unsigned addr = (unsigned)(size_t)p;  // LDS byte address
    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(mask), "v"(data) : "memory");
    return old;
}

__global__ void probe(const unsigned *bucket_of_lane, unsigned *ret_out, unsigned *tab_out, unsigned npat) {
    __shared__ unsigned tab[64];
    unsigned lane = threadIdx.x;
    for (unsigned p = 0; p < npat; p++) {
        tab[lane] = 0;
        __syncthreads();
        const unsigned bk = bucket_of_lane[p * 64 + lane];  // 16-bit bucket index 0..127: dword bk >> 1, half bk & 1
        const unsigned sh = 16u * (bk & 1u);
        const unsigned old = mskor_rtn(&tab[bk >> 1], 0xFFFFu << sh, (lane + 1) << sh);
        __syncthreads();
        ret_out[p * 64 + lane] = (old >> sh) & 0xFFFFu;
        tab_out[p * 64 + lane] = tab[lane];
        __syncthreads();
    }
}

int main() {
    const unsigned npat = 4000;
    std::vector<unsigned> bk(npat * 64);
    srand(2);
    for (unsigned p = 0; p < npat; p++) {
        unsigned nb = p == 0 ? 1 : p == 1 ? 2 : 1 + rand() % 128;
        for (unsigned l = 0; l < 64; l++) bk[p * 64 + l] = (p == 1) ? (l & 1) : rand() % nb;
    }
    unsigned *d_b, *d_r, *d_t;
    hipMalloc((void **)&d_b, bk.size() * 4);
    hipMalloc((void **)&d_r, bk.size() * 4);
    hipMalloc((void **)&d_t, bk.size() * 4);
    hipMemcpy(d_b, bk.data(), bk.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_b, d_r, d_t, npat);
    std::vector<unsigned> ret(bk.size()), tab(bk.size());
    hipMemcpy(ret.data(), d_r, ret.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(tab.data(), d_t, tab.size() * 4, hipMemcpyDeviceToHost);
    unsigned asc = 0, other = 0, tab_ok = 0;
    for (unsigned p = 0; p < npat; p++) {
        bool is_asc = true, t_ok = true;
        unsigned final_tab[64] = {0};
        for (unsigned l = 0; l < 64; l++) {
            const unsigned b = bk[p * 64 + l];
            int prev = -1;
            for (int j = (int)l - 1; j >= 0; j--) if (bk[p * 64 + j] == b) { prev = j; break; }
            if (ret[p * 64 + l] != (prev < 0 ? 0u : (unsigned)(prev + 1))) is_asc = false;
            const unsigned sh = 16u * (b & 1u);
            final_tab[b >> 1] = (final_tab[b >> 1] & ~(0xFFFFu << sh)) | ((l + 1) << sh);  // the highest lane of a bucket stays
        }
        for (unsigned w = 0; w < 64; w++) if (tab[p * 64 + w] != final_tab[w]) t_ok = false;
        asc += is_asc; other += !is_asc; tab_ok += t_ok;
        if (p < 2) {
            printf("pattern %u returned:", p);
            for (unsigned l = 0; l < 64; l++) printf(" %u", ret[p * 64 + l]);
            printf("\n");
        }
    }
    printf("patterns=%u ascending=%u other=%u final-table-as-expected=%u\n", npat, asc, other, tab_ok);
    return 0;
}
