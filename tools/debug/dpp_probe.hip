// probe: semantics and issue cost of wave-wide DPP shifts on gfx950 (v_mov_b32_dpp wave_shl:1 / wave_shr:1)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_sem(const float* in, float* l1, float* r1) {
    int vi = __builtin_bit_cast(int, in[threadIdx.x]);
    l1[threadIdx.x] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(-1, vi, 0x130, 0xf, 0xf, false));
    r1[threadIdx.x] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(-1, vi, 0x138, 0xf, 0xf, false));
}
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters) {
    int a = threadIdx.x, b = threadIdx.x * 3, c = threadIdx.x * 5, d = threadIdx.x * 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (MODE == 0) {   // wave_shl:1
                a = __builtin_amdgcn_update_dpp(0, a, 0x130, 0xf, 0xf, true); b = __builtin_amdgcn_update_dpp(0, b, 0x130, 0xf, 0xf, true);
                c = __builtin_amdgcn_update_dpp(0, c, 0x130, 0xf, 0xf, true); d = __builtin_amdgcn_update_dpp(0, d, 0x130, 0xf, 0xf, true);
            } else if (MODE == 1) {   // row_shl:1
                a = __builtin_amdgcn_update_dpp(0, a, 0x101, 0xf, 0xf, true); b = __builtin_amdgcn_update_dpp(0, b, 0x101, 0xf, 0xf, true);
                c = __builtin_amdgcn_update_dpp(0, c, 0x101, 0xf, 0xf, true); d = __builtin_amdgcn_update_dpp(0, d, 0x101, 0xf, 0xf, true);
            } else {   // plain integer add
                asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = __builtin_bit_cast(float, a ^ b ^ c ^ d);
}
int main() {
    float *in, *l1, *r1, *out;
    hipMalloc(&in, 256); hipMalloc(&l1, 256); hipMalloc(&r1, 256); hipMalloc(&out, 4 * 256 * 4096);
    std::vector<float> h(64); for (int i = 0; i < 64; i++) h[i] = (float)(i + 100);
    hipMemcpy(in, h.data(), 256, hipMemcpyHostToDevice);
    k_sem<<<1, 64>>>(in, l1, r1);
    std::vector<float> a(64), b(64);
    hipMemcpy(a.data(), l1, 256, hipMemcpyDeviceToHost); hipMemcpy(b.data(), r1, 256, hipMemcpyDeviceToHost);
    printf("wave_shl:1 lane0..3 = %g %g %g %g  lane 15,16,31,32 = %g %g %g %g lane 62,63 = %g %g\n", a[0], a[1], a[2], a[3], a[15], a[16], a[31], a[32], a[62], a[63]);
    printf("wave_shr:1 lane0..3 = %g %g %g %g  lane 15,16,31,32 = %g %g %g %g lane 62,63 = %g %g\n", b[0], b[1], b[2], b[3], b[15], b[16], b[31], b[32], b[62], b[63]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 5;   // 5 blocks of 4 waves per CU = 5 waves per SIMD
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) k_rate<0><<<blocks, 256>>>(out, iters); else if (mode == 1) k_rate<1><<<blocks, 256>>>(out, iters); else k_rate<2><<<blocks, 256>>>(out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("mode %d (%s): %.3f ms, %.2f ns per wave-instruction per SIMD\n", mode, mode == 0 ? "wave_shl:1" : mode == 1 ? "row_shl:1" : "v_add_u32",
                                 ms, ms * 1e6 / ((double)iters * 64 * 5));
        }
    }
    return 0;
}
