// pbr_internal.hpp — host-side context shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include "../../include/pbr_hip.h"

struct pbr_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    void* scratch = nullptr;          // device scratch (SH partials, ...)
    size_t scratch_bytes = 0;
    std::string err;
    std::vector<float> host_tmp;      // host staging (prefilter sample tables)
    // pbr_prefilter_env: the sample tables of the last (size, mips, sky mips), kept on the device between calls
    void* pf_dev = nullptr;
    uint32_t pf_key[3] = {0, 0, 0};
    uint32_t pf_count[16] = {};
    float pf_wsum[16] = {};
    // RCCL (loaded lazily with dlopen so a 1-GPU run never needs librccl)
    void* rccl_lib = nullptr;
    void* comm = nullptr;             // frame communicator: halo exchange (ncclSend / ncclRecv)
    void* comm_hist = nullptr;        // ncclCommSplit of it: the histogram all-reduce (may run on the side stream)
    int world = 1;
    int rank = 0;
    // high-priority side stream of pbr_ctx_side_begin / _end / _join
    hipStream_t side_stream = nullptr, main_saved = nullptr;
    hipEvent_t ev_side_fork = nullptr, ev_side_join = nullptr;
    bool on_side = false, side_pending = false, main_was_null = false;
    std::vector<uint32_t> side_cu_mask;   // pbr_ctx_set_cu_masks: the side stream's CUs (empty: all, high priority)
    bool bloom_shader_order = false;      // pbr_ctx_set_bloom_shader_order: large 2x-up bloom levels in the shader's operation order (bit-exact)
    int cu_count = 0;                     // compute units of the device (the shade sizes its blocks by the resident-block count)
};

namespace pbr {

inline pbr_status fail(pbr_ctx* ctx, pbr_status code, const char* what) {
    if (ctx) ctx->err = what;
    return code;
}
inline pbr_status hip_fail(pbr_ctx* ctx, hipError_t e, const char* where) {
    if (ctx) {
        ctx->err = std::string(where) + ": " + hipGetErrorString(e);
    }
    return PBR_ERR_HIP;
}
// after a kernel launch
inline pbr_status launched(pbr_ctx* ctx, const char* where) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(ctx, e, where);
    return PBR_OK;
}

constexpr size_t SCRATCH_BYTES = 1u << 20;

// Tuning / A-B switches of the launch code.  The PRODUCT library takes every one at its built-in value and never reads the
// environment: a drop-in must not change its kernel choice on an environment variable.  A build with -DPBR_DEBUG_KNOBS
// (`make knobs` -> libpbr_hip_knobs.so; used by the tests that force a kernel choice and by the sweep scripts through
// PBR_HIP_LIB) reads PBR_<NAME> once per call site.
inline const char* knob_text(const char* name) {
#ifdef PBR_DEBUG_KNOBS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
inline int knob_int(const char* name, int dflt) { const char* v = knob_text(name); return v ? atoi(v) : dflt; }
inline float knob_float(const char* name, float dflt) { const char* v = knob_text(name); return v ? (float)atof(v) : dflt; }
inline bool knob_set(const char* name) { return knob_text(name) != nullptr; }

}  // namespace pbr

#define PBR_REQUIRE(ctx, cond, msg) \
    do { if (!(cond)) return pbr::fail((ctx), PBR_ERR_INVALID, msg); } while (0)
#define PBR_HIP(ctx, call) \
    do { hipError_t e__ = (call); if (e__ != hipSuccess) return pbr::hip_fail((ctx), e__, #call); } while (0)
