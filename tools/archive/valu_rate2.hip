// Micro-benchmark 2: per-opcode VALU issue cost on gfx950 via inline asm (8 independent chains per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP8(INS) \
    asm volatile(INS " %0, %0, %8\n\t" INS " %1, %1, %9\n\t" INS " %2, %2, %8\n\t" INS " %3, %3, %9\n\t" \
                 INS " %4, %4, %8\n\t" INS " %5, %5, %9\n\t" INS " %6, %6, %8\n\t" INS " %7, %7, %9" \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y0), "v"(y1));
#define OP8U(INS) \
    asm volatile(INS " %0, %0\n\t" INS " %1, %1\n\t" INS " %2, %2\n\t" INS " %3, %3\n\t" INS " %4, %4\n\t" INS " %5, %5\n\t" INS " %6, %6\n\t" INS " %7, %7" \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
template <int MODE>
__global__ void k(float* out, int iters) {
    float x0 = threadIdx.x + 1.5f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float y0 = 1.0f + 1e-7f * threadIdx.x, y1 = 1.0f - 1e-7f * threadIdx.x;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { OP8("v_mul_f32") }
        else if (MODE == 1) { OP8("v_max_f32") }
        else if (MODE == 2) { OP8("v_add_f32") }
        else if (MODE == 3) { OP8U("v_rcp_f32") }
        else if (MODE == 4) { OP8U("v_rsq_f32") }
        else if (MODE == 5) { OP8U("v_floor_f32") }
        else if (MODE == 6) { OP8U("v_cvt_f32_i32") }
        else { OP8U("v_mov_b32") }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int MODE>
void run(const char* name, int w, float* out) {
    int iters = 20000, blocks = 256 * w;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * 8 * w;
    printf("%-14s waves/SIMD %d: %.2f cycles/instr/SIMD @2.4GHz\n", name, w, ms * 1e6 / n * 2.4);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {2, 4, 8}) { run<0>("v_mul_f32", w, out); run<1>("v_max_f32", w, out); run<2>("v_add_f32", w, out); run<3>("v_rcp_f32", w, out);
                              run<4>("v_rsq_f32", w, out); run<5>("v_floor_f32", w, out); run<6>("v_cvt_f32_i32", w, out); run<7>("v_mov_b32", w, out); }
    return 0;
}
