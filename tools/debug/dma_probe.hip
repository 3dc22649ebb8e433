// dma_probe.hip — does global_load_lds_dwordx4 / _dword on gfx950 take 4-byte-aligned (not 16-byte-aligned) global addresses, with
// inactive lanes skipped, and does the M0 recipe of the programming guide work inside compiler-scheduled code?  (shade.hip stages
// the light lists of the NEXT work item with it: 156-byte cluster records, the 128 bytes of indices start at +28.)
//   hipcc -O3 --offload-arch=gfx950 tools/debug/dma_probe.hip -o tools/debug/bin/dma_probe && tools/debug/bin/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>

__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// records of 156 bytes: count at +24, 32 indices at +28.  Block of 256 threads copies `n_cl` records' indices (x4: 8 lanes per record) and counts.
__global__ void k_probe(const unsigned char* __restrict__ rec, int n_cl, uint32_t* __restrict__ out_idx, uint32_t* __restrict__ out_cnt) {
    extern __shared__ uint4 lds[];
    uint32_t* raw = reinterpret_cast<uint32_t*>(lds) + 64;          // not at LDS offset 0 on purpose
    uint32_t* cnt = raw + 96 * 32;
    for (int i = threadIdx.x; i < 96 * 33; i += 256) raw[i] = 0xDEADBEEFu;
    __syncthreads();
    const uint32_t raw_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)raw;
    const uint32_t cnt_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)cnt;
    const int w = threadIdx.x >> 6;
    for (int r = 0; r < 3; r++) {
        const int c = (threadIdx.x >> 3) + 32 * r, part = threadIdx.x & 7;
        if (c < n_cl) dma16(rec + (size_t)c * 156 + 28 + 16 * part, __builtin_amdgcn_readfirstlane(raw_b + (32 * r + 8 * w) * 128));
    }
    if ((int)threadIdx.x < n_cl) dma4(rec + (size_t)threadIdx.x * 156 + 24, __builtin_amdgcn_readfirstlane(cnt_b + w * 256));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 96 * 32; i += 256) out_idx[i] = raw[i];
    for (int i = threadIdx.x; i < 96; i += 256) out_cnt[i] = cnt[i];
}

int main() {
    const int N = 96;
    std::vector<unsigned char> h(N * 156 + 64);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char)(i * 131 + 7);
    unsigned char* d; uint32_t *oi, *oc;
    hipMalloc(&d, h.size()); hipMalloc(&oi, N * 32 * 4); hipMalloc(&oc, N * 4);
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    int bad = 0;
    for (int n_cl : {8, 24, 40, 80, 96}) {
        hipLaunchKernelGGL(k_probe, dim3(4), dim3(256), 64 * 4 + 96 * 33 * 4 + 64, 0, d, n_cl, oi, oc);
        std::vector<uint32_t> hi(N * 32), hc(N);
        hipMemcpy(hi.data(), oi, N * 32 * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hc.data(), oc, N * 4, hipMemcpyDeviceToHost);
        int e = 0;
        for (int c = 0; c < N; c++) {
            uint32_t want_c; memcpy(&want_c, &h[c * 156 + 24], 4);
            if (hc[c] != (c < n_cl ? want_c : 0xDEADBEEFu)) e++;
            for (int j = 0; j < 32; j++) {
                uint32_t want; memcpy(&want, &h[c * 156 + 28 + 4 * j], 4);
                if (hi[c * 32 + j] != (c < n_cl ? want : 0xDEADBEEFu)) e++;
            }
        }
        printf("n_cl %d: %d mismatches (hipGetLastError %d)\n", n_cl, e, (int)hipDeviceSynchronize());
        bad += e;
    }
    printf(bad ? "DMA PROBE FAILED\n" : "DMA PROBE OK: unaligned x4 + dword LDS-DMA, inactive lanes skipped\n");
    return bad != 0;
}
