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

#define OP8T(INS) /* three sources: d, d, y0, y1 */ \
    asm volatile(INS " %0, %0, %8, %9\n\t" INS " %1, %1, %9, %8\n\t" INS " %2, %2, %8, %9\n\t" INS " %3, %3, %9, %8\n\t" \
                 INS " %4, %4, %8, %9\n\t" INS " %5, %5, %9, %8\n\t" INS " %6, %6, %8, %9\n\t" INS " %7, %7, %9, %8" \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y0), "v"(y1));
#define RFL8 /* v_readfirstlane_b32 into eight SGPRs */ \
    { int s0, s1, s2, s3, s4, s5, s6, s7; \
      asm volatile("v_readfirstlane_b32 %0, %8\n\tv_readfirstlane_b32 %1, %9\n\tv_readfirstlane_b32 %2, %10\n\tv_readfirstlane_b32 %3, %11\n\t" \
                   "v_readfirstlane_b32 %4, %12\n\tv_readfirstlane_b32 %5, %13\n\tv_readfirstlane_b32 %6, %14\n\tv_readfirstlane_b32 %7, %15" \
                   : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7) \
                   : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7)); \
      asm volatile("" :: "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7)); }

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
        else if (MODE == 17) { OP8U("v_cvt_rpi_i32_f32") }
        else if (MODE == 18) { OP8U("v_cvt_flr_i32_f32") }
        else if (MODE == 19) { OP8U("v_cvt_f32_ubyte0") }
        else if (MODE == 20) { OP8U("v_cvt_i32_f32") }
        else if (MODE == 21) { OP8T("v_mad_u32_u24") }
        else if (MODE == 22) { OP8("v_mul_lo_u32") }
        else if (MODE == 23) { OP8("v_mul_u32_u24") }
        else if (MODE == 24) { OP8T("v_lshl_add_u32") }
        else if (MODE == 25) { OP8("v_add_u32") }
        else if (MODE == 26) { OP8("v_lshlrev_b32") }
        else if (MODE == 27) { OP8("v_and_b32") }
        else if (MODE == 28) { RFL8 }
        else if (MODE == 29) { OP8U("v_fract_f32") }
        else if (MODE == 30) { OP8("v_cvt_pkrtz_f16_f32") }
        else if (MODE == 31) { OP8T("v_add3_u32") }
        else if (MODE == 32) { OP8T("v_med3_f32") }
        else if (MODE == 33) { OP8U("v_cvt_f32_i32") }
        else if (MODE == 34) { OP8U("v_exp_f32") }
        else if (MODE == 35) { OP8U("v_sqrt_f32") }
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

// what v_cvt_rpi_i32_f32 computes: (int)floorf(x + 0.5f) with the fp32 add's rounding, or the exact floor(x + 0.5)?
__global__ void k_rpi(const float* x, int* rpi, int* flr, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int a, b;
    const float y = x[i] + 0.5f;
    asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(a) : "v"(x[i]));
    asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(b) : "v"(y));
    rpi[i] = a; flr[i] = b;
}
static void rpi_probe() {
    std::vector<float> x;
    for (int k = -2; k < 20; k++) for (int j = -40; j <= 40; j++) {
        const float base = ldexpf(1.0f, k);
        float v = base - 0.5f;
        for (int t = 0; t < abs(j); t++) v = nextafterf(v, j < 0 ? -1e30f : 1e30f);
        x.push_back(v);
        float u = (float)(k + 3) + 0.5f;
        for (int t = 0; t < abs(j); t++) u = nextafterf(u, j < 0 ? -1e30f : 1e30f);
        x.push_back(u);
    }
    uint32_t r = 12345u;
    for (int i = 0; i < 1 << 20; i++) { r = r * 1664525u + 1013904223u; x.push_back((float)(r >> 8) * (1.0f / 16777216.0f) * 140000.0f); }
    const int n = (int)x.size();
    float* dx; int *da, *db;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&da, n * 4); (void)hipMalloc(&db, n * 4);
    (void)hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_rpi, dim3((n + 255) / 256), dim3(256), 0, 0, dx, da, db, n);
    std::vector<int> a(n), b(n);
    (void)hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    long vs_float = 0, vs_exact = 0, flr_bad = 0, shown = 0;
    for (int i = 0; i < n; i++) {
        const int f = (int)floorf(x[i] + 0.5f), e = (int)floor((double)x[i] + 0.5);
        vs_float += a[i] != f; vs_exact += a[i] != e; flr_bad += b[i] != f;
        if (f != e && shown < 6) { printf("  x = %.9g: floorf(x + 0.5f) = %d, exact floor(x + 0.5) = %d, v_cvt_rpi = %d, v_add + v_cvt_flr = %d\n", x[i], f, e, a[i], b[i]); shown++; }
    }
    printf("v_cvt_rpi_i32_f32 on %d inputs: %ld differ from (int)floorf(x + 0.5f), %ld differ from the exact floor(x + 0.5); v_add_f32 0.5 + v_cvt_flr_i32_f32: %ld differ from (int)floorf(x + 0.5f)\n\n",
           n, vs_float, vs_exact, flr_bad);
}

int main(int argc, char** argv) {
    const bool only_new = argc > 1;   // any argument: only the round-3b rows (integer / convert instructions), at 5 and 8 waves per SIMD
    rpi_probe();
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    Stamp* st; (void)hipMalloc(&st, 256 * 8 * 4 * sizeof(Stamp));
    for (int w : {1, 2, 4, 5, 8}) {
        if (only_new && w != 5 && w != 8) continue;
        run<17>("v_cvt_rpi_i32", w, out, st); run<18>("v_cvt_flr_i32", w, out, st); run<19>("v_cvt_f32_ubyte0", w, out, st); run<20>("v_cvt_i32_f32", w, out, st);
        run<33>("v_cvt_f32_i32", w, out, st); run<21>("v_mad_u32_u24", w, out, st); run<22>("v_mul_lo_u32", w, out, st); run<23>("v_mul_u32_u24", w, out, st);
        run<24>("v_lshl_add_u32", w, out, st); run<25>("v_add_u32", w, out, st); run<26>("v_lshlrev_b32", w, out, st); run<27>("v_and_b32", w, out, st);
        run<28>("v_readfirstlane", w, out, st); run<29>("v_fract_f32", w, out, st); run<30>("v_cvt_pkrtz_f16", w, out, st); run<31>("v_add3_u32", w, out, st);
        run<32>("v_med3_f32", w, out, st); run<34>("v_exp_f32", w, out, st); run<35>("v_sqrt_f32", w, out, st);
        if (only_new) { printf("\n"); continue; }
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
