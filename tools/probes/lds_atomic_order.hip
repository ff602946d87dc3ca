// Probe: in which order does the LDS apply same-address atomics issued by ONE wave instruction?
// For every pattern, lanes of a group atomicMax ascending values into one LDS word and keep the
// returned old value; "ascending" means every lane saw exactly its predecessor in the group.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void probe(const unsigned *group_of_lane, unsigned *ret_out, unsigned npat) {
    __shared__ unsigned tab[64];
    unsigned lane = threadIdx.x;
    for (unsigned p = 0; p < npat; p++) {
        tab[lane] = 0;
        __syncthreads();
        unsigned g = group_of_lane[p * 64 + lane];
        unsigned old = atomicMax(&tab[g], (lane + 1) << 16);
        __syncthreads();
        ret_out[p * 64 + lane] = old;
        __syncthreads();
    }
}

int main() {
    const unsigned npat = 2000;
    std::vector<unsigned> groups(npat * 64);
    srand(1);
    for (unsigned p = 0; p < npat; p++) {
        unsigned ng = p == 0 ? 1 : p == 1 ? 2 : 1 + rand() % 64;
        for (unsigned l = 0; l < 64; l++) groups[p * 64 + l] = (p == 1) ? (l & 1) : rand() % ng;
    }
    unsigned *d_g, *d_r;
    hipMalloc((void **)&d_g, groups.size() * 4);
    hipMalloc((void **)&d_r, groups.size() * 4);
    hipMemcpy(d_g, groups.data(), groups.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_g, d_r, npat);
    std::vector<unsigned> ret(groups.size());
    hipMemcpy(ret.data(), d_r, ret.size() * 4, hipMemcpyDeviceToHost);
    unsigned asc = 0, desc = 0, other = 0;
    for (unsigned p = 0; p < npat; p++) {
        bool is_asc = true, is_desc = true;
        for (unsigned l = 0; l < 64; l++) {
            unsigned g = groups[p * 64 + l];
            int prev = -1, next = -1;
            for (int j = (int)l - 1; j >= 0; j--) if (groups[p * 64 + j] == g) { prev = j; break; }
            for (unsigned j = l + 1; j < 64; j++) if (groups[p * 64 + j] == g) { next = (int)j; break; }
            unsigned want_asc = prev < 0 ? 0 : (unsigned)(prev + 1) << 16;
            unsigned r = ret[p * 64 + l];
            if (r != want_asc) is_asc = false;
            // descending: the highest lane goes first (sees 0), everyone else sees a larger value
            (void)next;
            if (!(r == 0 ? next < 0 : r > ((l + 1) << 16))) is_desc = false;
        }
        if (is_asc) asc++; else if (is_desc) desc++; else other++;
        if (p < 2) {
            printf("pattern %u returned(lane>>16):", p);
            for (unsigned l = 0; l < 64; l++) printf(" %u", ret[p * 64 + l] >> 16);
            printf("\n");
        }
    }
    printf("patterns=%u ascending=%u descending=%u other=%u\n", npat, asc, desc, other);
    return 0;
}
