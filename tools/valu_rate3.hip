// Micro-benchmark 3: VALU issue rate on gfx950 measured in SHADER-CLOCK CYCLES, not wall time x an assumed clock.
//
// Every wave brackets its instruction loop with s_memtime (clock64: the shader-core cycle counter) and
// s_memrealtime (wall_clock64: the constant 100 MHz counter).  For each opcode and each occupancy (waves per SIMD):
//   per-wave   = cycles one wave needs per instruction of its own stream            (issue cost seen by a wave)
//   aggregate  = SIMD-level throughput: cycles between the first start and the last end of the waves that shared a
//                SIMD, divided by ALL instructions they issued                      (what a roofline must use)
//   clock_GHz  = shader cycles / real time during the loop                         (the DVFS state it ran at)
// If a SIMD retired a wave64 fp32 op every 2 cycles with several waves resident, `aggregate` would drop to ~2 at
// occupancy >= 2 while `per-wave` stays ~4.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate3.hip -o /tmp/valu_rate3 && /tmp/valu_rate3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <map>

typedef float f2 __attribute__((ext_vector_type(2)));

#define OP8(INS) \
    asm volatile(INS " %0, %0, %8\n\t" INS " %1, %1, %9\n\t" INS " %2, %2, %8\n\t" INS " %3, %3, %9\n\t" \
                 INS " %4, %4, %8\n\t" INS " %5, %5, %9\n\t" INS " %6, %6, %8\n\t" INS " %7, %7, %9" \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y0), "v"(y1));
#define OP8S(INS) /* second source an SGPR */ \
    asm volatile(INS " %0, %0, %8\n\t" INS " %1, %1, %9\n\t" INS " %2, %2, %8\n\t" INS " %3, %3, %9\n\t" \
                 INS " %4, %4, %8\n\t" INS " %5, %5, %9\n\t" INS " %6, %6, %8\n\t" INS " %7, %7, %9" \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(sa), "s"(sb));
#define FMA8 /* v_fma_f32 d, d, y0, y1: three VGPR sources */ \
    asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %9, %8\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %9, %8\n\t" \
                 "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %9, %8\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %9, %8" \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y0), "v"(y1));
#define PK8(INS3) /* packed: 64-bit register pairs */ \
    asm volatile(INS3 " %0, %0, %8, %9\n\t" INS3 " %1, %1, %9, %8\n\t" INS3 " %2, %2, %8, %9\n\t" INS3 " %3, %3, %9, %8\n\t" \
                 INS3 " %4, %4, %8, %9\n\t" INS3 " %5, %5, %9, %8\n\t" INS3 " %6, %6, %8, %9\n\t" INS3 " %7, %7, %9, %8" \
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q0), "v"(q1));
#define PK8B(INS2) \
    asm volatile(INS2 " %0, %0, %8\n\t" INS2 " %1, %1, %9\n\t" INS2 " %2, %2, %8\n\t" INS2 " %3, %3, %9\n\t" \
                 INS2 " %4, %4, %8\n\t" INS2 " %5, %5, %9\n\t" INS2 " %6, %6, %8\n\t" INS2 " %7, %7, %9" \
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q0), "v"(q1));
#define OP8U(INS) \
    asm volatile(INS " %0, %0\n\t" INS " %1, %1\n\t" INS " %2, %2\n\t" INS " %3, %3\n\t" INS " %4, %4\n\t" INS " %5, %5\n\t" INS " %6, %6\n\t" INS " %7, %7" \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));

#define MIX8 /* v_fma_mix_f32 d, half(y0.lo), y1, d: the fp16 -> fp32 convert rides in the instruction */ \
    asm volatile("v_fma_mix_f32 %0, %8, %9, %0 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %8, %9, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" \
                 "v_fma_mix_f32 %2, %8, %9, %2 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, %8, %9, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" \
                 "v_fma_mix_f32 %4, %8, %9, %4 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %5, %8, %9, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" \
                 "v_fma_mix_f32 %6, %8, %9, %6 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %7, %8, %9, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]" \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y0), "v"(y1));
#define CUBE8 /* v_cubeid_f32 d, d, y0, y1 */ \
    asm volatile("v_cubeid_f32 %0, %0, %8, %9\n\tv_cubesc_f32 %1, %1, %9, %8\n\tv_cubetc_f32 %2, %2, %8, %9\n\tv_cubema_f32 %3, %3, %9, %8\n\t" \
                 "v_cubeid_f32 %4, %4, %8, %9\n\tv_cubesc_f32 %5, %5, %9, %8\n\tv_cubetc_f32 %6, %6, %8, %9\n\tv_cubema_f32 %7, %7, %9, %8" \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y0), "v"(y1));

struct Stamp { uint64_t c0, c1, r0, r1; uint32_t hw_id, xcc; };

template <int MODE>
__global__ void k(float* out, Stamp* st, int iters) {
    float x0 = threadIdx.x + 1.5f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float y0 = 1.0f + 1e-7f * threadIdx.x, y1 = 1.0f - 1e-7f * threadIdx.x;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    f2 q0 = {y0, y1}, q1 = {y1 * 1e-9f, y0 * 1e-9f};
    const float sa = 1.0f + 1e-7f * (float)(blockIdx.x & 1), sb = 1.0f - 1e-7f * (float)(blockIdx.x & 1);
    const uint64_t r0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { OP8("v_mul_f32") }
        else if (MODE == 1) { OP8("v_add_f32") }
        else if (MODE == 2) { FMA8 }
        else if (MODE == 3) { OP8S("v_mul_f32") }
        else if (MODE == 4) { PK8("v_pk_fma_f32") }
        else if (MODE == 5) { PK8B("v_pk_mul_f32") }
        else if (MODE == 6) { PK8B("v_pk_add_f32") }
        else if (MODE == 7) { OP8U("v_rcp_f32") }
        else if (MODE == 8) { OP8U("v_rsq_f32") }
        else if (MODE == 9) { OP8("v_max_f32") }
        else if (MODE == 11) { MIX8 }
        else if (MODE == 12) { OP8U("v_cvt_f32_f16") }
        else if (MODE == 13) { OP8("v_fmac_f32") }
        else if (MODE == 14) { CUBE8 }
        else if (MODE == 15) { OP8U("v_log_f32") }
        else if (MODE == 16) { OP8U("v_floor_f32") }
        else { OP8U("v_mov_b32") }
    }
    const uint64_t c1 = clock64(), r1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                                 p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    if ((threadIdx.x & 63) == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        st[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = Stamp{c0, c1, r0, r1, hw, xcc};
    }
}

template <int MODE>
void run(const char* name, int w, float* out, Stamp* st_dev) {
    const int iters = 20000, blocks = 256 * w, waves = blocks * 4;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, st_dev, 100);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, st_dev, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> st(waves);
    (void)hipMemcpy(st.data(), st_dev, waves * sizeof(Stamp), hipMemcpyDeviceToHost);
    const double n = (double)iters * 8;
    // per wave
    std::vector<double> per_wave;
    double clk = 0;
    for (auto& s : st) {
        per_wave.push_back((double)(s.c1 - s.c0) / n);
        clk += (double)(s.c1 - s.c0) / ((double)(s.r1 - s.r0) * 10.0);   // 100 MHz real-time counter -> cycles per ns
    }
    std::sort(per_wave.begin(), per_wave.end());
    // group by SIMD: XCC id + HW_ID's se/sh/cu/simd fields.  gfx9 HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ...
    std::map<uint32_t, std::vector<const Stamp*>> groups;
    for (auto& s : st) groups[(s.xcc & 0xF) << 16 | (s.hw_id & 0xFF30)].push_back(&s);
    std::vector<double> agg;
    size_t max_group = 0;
    for (auto& g : groups) {
        uint64_t lo = ~0ull, hi = 0;
        for (auto* s : g.second) { lo = std::min(lo, s->c0); hi = std::max(hi, s->c1); }
        agg.push_back((double)(hi - lo) / (n * g.second.size()));
        max_group = std::max(max_group, g.second.size());
    }
    std::sort(agg.begin(), agg.end());
    printf("%-14s waves/SIMD %d | per-wave cycles/instr: median %.2f (min %.2f max %.2f) | SIMD aggregate cycles/instr: median %.2f (min %.2f max %.2f; %zu SIMDs seen, <= %zu waves each) | clock %.2f GHz | wall %.3f ms = %.2f cycles/instr/SIMD at that clock\n",
           name, w, per_wave[per_wave.size() / 2], per_wave.front(), per_wave.back(), agg[agg.size() / 2], agg.front(), agg.back(), groups.size(), max_group,
           clk / waves, ms, ms * 1e6 * (clk / waves) / (n * w));
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    Stamp* st; (void)hipMalloc(&st, 256 * 8 * 4 * sizeof(Stamp));
    for (int w : {1, 2, 4, 5, 8}) {
        run<0>("v_mul_f32", w, out, st); run<1>("v_add_f32", w, out, st); run<2>("v_fma_f32 3vgpr", w, out, st); run<3>("v_mul_f32 sgpr", w, out, st);
        run<4>("v_pk_fma_f32", w, out, st); run<5>("v_pk_mul_f32", w, out, st); run<6>("v_pk_add_f32", w, out, st);
        run<7>("v_rcp_f32", w, out, st); run<8>("v_rsq_f32", w, out, st); run<9>("v_max_f32", w, out, st); run<10>("v_mov_b32", w, out, st);
        // round 3: the instructions the prefilter / bloom / shade trims were choosing between
        run<11>("v_fma_mix_f32", w, out, st); run<12>("v_cvt_f32_f16", w, out, st); run<13>("v_fmac_f32", w, out, st); run<14>("v_cube*_f32", w, out, st);
        run<15>("v_log_f32", w, out, st); run<16>("v_floor_f32", w, out, st);
        printf("\n");
    }
    return 0;
}
