// Micro-benchmark: VALU issue rate on gfx950 for plain vs packed fp32 (drives the shade kernel's design).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    f2 A = {a, a}, B = {b, b};
    float y0 = 1.0f + 1e-7f * threadIdx.x, y1 = 1.0f - 1e-7f * threadIdx.x, z0 = 1e-9f * threadIdx.x, z1 = -1e-9f * threadIdx.x;
    f2 q0 = {y0, y1}, q1 = {z0, z1};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {   // 8 independent v_fma_f32
            x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b; x4 = x4 * a + b; x5 = x5 * a + b; x6 = x6 * a + b; x7 = x7 * a + b;
        } else if (MODE == 1) {   // 8 independent v_pk_fma_f32
            p0 = p0 * A + B; p1 = p1 * A + B; p2 = p2 * A + B; p3 = p3 * A + B; p4 = p4 * A + B; p5 = p5 * A + B; p6 = p6 * A + B; p7 = p7 * A + B;
        } else if (MODE == 2) {   // 8 v_rcp_f32
            x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
            x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7);
        } else if (MODE == 4) {   // 8 v_fma_f32 with three VGPR operands
            x0 = x0 * y0 + z0; x1 = x1 * y1 + z1; x2 = x2 * y0 + z1; x3 = x3 * y1 + z0; x4 = x4 * y0 + z0; x5 = x5 * y1 + z1; x6 = x6 * y0 + z1; x7 = x7 * y1 + z0;
        } else if (MODE == 5) {   // 8 v_mul_f32 with two VGPR operands
            x0 = x0 * y0; x1 = x1 * y1; x2 = x2 * y0; x3 = x3 * y1; x4 = x4 * y0; x5 = x5 * y1; x6 = x6 * y0; x7 = x7 * y1;
        } else if (MODE == 6) {   // 8 v_pk_fma_f32 with three VGPR-pair operands
            p0 = p0 * q0 + q1; p1 = p1 * q1 + q0; p2 = p2 * q0 + q1; p3 = p3 * q1 + q0; p4 = p4 * q0 + q1; p5 = p5 * q1 + q0; p6 = p6 * q0 + q1; p7 = p7 * q1 + q0;
        } else {   // 8 v_max_f32
            x0 = fmaxf(x0, a); x1 = fmaxf(x1, b); x2 = fmaxf(x2, a); x3 = fmaxf(x3, b); x4 = fmaxf(x4, a); x5 = fmaxf(x5, b); x6 = fmaxf(x6, a); x7 = fmaxf(x7, b);
            a += 1.0f; b += 1.0f;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}
template <int MODE>
void run(const char* name, int waves_per_simd, float* out) {
    int iters = 20000;
    int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts_per_simd = (double)iters * 8 * waves_per_simd;
    printf("%-14s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, waves_per_simd, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4, 8}) { run<0>("v_fma_f32", w, out); run<1>("v_pk_fma_f32", w, out); run<2>("v_rcp_f32", w, out); run<3>("v_max_f32", w, out); run<4>("v_fma 3vgpr", w, out); run<5>("v_mul 2vgpr", w, out); run<6>("v_pk_fma 3vgpr", w, out); }
    return 0;
}
