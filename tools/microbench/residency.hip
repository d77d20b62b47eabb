// How many workgroups of a given size / LDS footprint does a gfx950 CU hold at once?  Each block spins
// for ~20 us of wall clock and records its interval and CU; the host reports the peak overlap per CU.
// build: hipcc --offload-arch=gfx950 -O2 -o residency residency.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

template <int NV>
__global__ void k_spin(unsigned long long *rec, int ticks) {
    extern __shared__ unsigned s_dyn[];
    // raise the kernel's VGPR allocation without using the registers
    if (NV == 32) asm volatile("v_mov_b32 v31, 0" ::: "v31");
    if (NV == 56) asm volatile("v_mov_b32 v55, 0" ::: "v55");
    if (NV == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if (NV == 72) asm volatile("v_mov_b32 v71, 0" ::: "v71");
    if (NV == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) s_dyn[0] = 1;
    while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec[3 * blockIdx.x + 0] = t0;
        rec[3 * blockIdx.x + 1] = wall_clock64();
        rec[3 * blockIdx.x + 2] = ((unsigned long long)(xcc & 0xff) << 32) | hwid;
    }
}

int main() {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
    const int cus = p.multiProcessorCount, blocks = cus * 12;
    unsigned long long *d;
    if (hipMalloc(&d, (size_t)blocks * 24) != hipSuccess) return 1;
    std::vector<unsigned long long> h((size_t)blocks * 3);
    for (int nv : {8, 32, 56, 64, 72, 128})
    for (int threads : {256, 512})
        for (int lds : {1024, 32768, 40960}) {
            switch (nv) {
#define CASE(N) case N: (void)hipFuncSetAttribute((const void *)k_spin<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                        hipLaunchKernelGGL(k_spin<N>, dim3(blocks), dim3(threads), lds, 0, d, 2000); break;
                CASE(8) CASE(32) CASE(56) CASE(64) CASE(72) CASE(128)
#undef CASE
            }
            if (hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
            std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
            for (int b = 0; b < blocks; ++b) {
                const unsigned long long hw = h[3 * b + 2];
                const unsigned long long key = ((hw >> 32) << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 0xf);
                ev[key].push_back({h[3 * b], 1});
                ev[key].push_back({h[3 * b + 1], -1});
            }
            int lo = 1 << 30, hi = 0;
            for (auto &kv : ev) {
                std::sort(kv.second.begin(), kv.second.end());
                int c = 0, best = 0;
                for (auto &e : kv.second) { c += e.second; best = std::max(best, c); }
                lo = std::min(lo, best); hi = std::max(hi, best);
            }
            printf("VGPRs %3d threads %4d  LDS %6d B : CUs seen %zu, peak resident workgroups per CU %d..%d\n", nv, threads, lds, ev.size(), lo, hi);
        }
    return 0;
}
