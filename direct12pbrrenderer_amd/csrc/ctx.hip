// ctx.hip — context, layout helpers and the RCCL histogram all-reduce of the C ABI.
#include "pbr_internal.hpp"
#include "pbr_device.hpp"
#include <dlfcn.h>
// RCCL: declarations only — the library is bound with dlopen (section "RCCL"), so a one-GPU process never needs librccl.  A build box
// without the development header still builds: the handful of types, enum values and signatures the calls need are restated below
// (RCCL >= 2.18 ABI) and the static_asserts that pin them to the header are compiled out.  -DPBR_NO_RCCL_HEADER forces that branch:
// tests/test_runtime_cpu.py compiles this file host-only with it, so the branch stays buildable on boxes that do have the header.
#if __has_include(<rccl/rccl.h>) && !defined(PBR_NO_RCCL_HEADER)
#include <rccl/rccl.h>
#define PBR_HAVE_RCCL_HEADER 1
#else
#define PBR_HAVE_RCCL_HEADER 0
extern "C" {
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint32 = 3 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef struct ncclConfig_v21700 ncclConfig_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId*);
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int);
ncclResult_t ncclCommSplit(ncclComm_t, int, int, ncclComm_t*, ncclConfig_t*);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
ncclResult_t ncclSend(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclRecv(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
}
#endif
#include <link.h>
#include <unistd.h>
#include <cstring>
#include <vector>

extern "C" {

const char* pbr_version(void) { return "pbr_hip 0.1 (gfx950)"; }

size_t pbr_cube_texels(uint32_t size, uint32_t mips) { return pbr::cube_mip_offset(size, mips); }
size_t pbr_cube_mip_offset(uint32_t size, uint32_t mip) { return pbr::cube_mip_offset(size, mip); }
size_t pbr_env_padded_mip_offset(uint32_t size, uint32_t mip) { return pbr::env_padded_mip_offset(size, mip); }
size_t pbr_env_padded_texels(uint32_t size, uint32_t mips) { return pbr::env_padded_mip_offset(size, mips); }
size_t pbr_bloom_level_offset(uint32_t w, uint32_t h, uint32_t level) {
    size_t off = 0;
    for (uint32_t l = 0; l < level; l++) off += (size_t)(w >> l) * (h >> l);
    return off;
}
size_t pbr_bloom_chain_texels(uint32_t w, uint32_t h) { return pbr_bloom_level_offset(w, h, PBR_BLOOM_MIPS); }

// A process that also hosts PyTorch must run ONE HIP runtime: torch ships its own libamdhip64 (same soname as
// /opt/rocm's), and whichever copy is mapped first serves both.  If this library was loaded before torch, torch ends
// up on the system runtime it was not built against and fails later in obscure ways (hipStreamCreate errors).  The
// loader cannot be told to prefer "the copy torch will bring", so the mismatch is detected and reported instead.
struct RuntimeScan { std::string hip_dir, torch_dir; };
static int scan_cb(struct dl_phdr_info* info, size_t, void* data) {
    RuntimeScan* s = (RuntimeScan*)data;
    const char* name = info->dlpi_name;
    if (!name || !*name) return 0;
    const char* base = std::strrchr(name, '/');
    const std::string dir = base ? std::string(name, (size_t)(base - name)) : std::string();
    base = base ? base + 1 : name;
    if (std::strncmp(base, "libamdhip64.so", 14) == 0 && s->hip_dir.empty()) s->hip_dir = dir;
    if (std::strncmp(base, "libtorch_hip.so", 15) == 0) s->torch_dir = dir;
    return 0;
}
static char g_runtime_err[512];
static bool dir_has_hip_runtime(const std::string& dir) {
    // torch wheels bundle libamdhip64.so next to libtorch_hip.so; a torch built against the system ROCm does not
    for (const char* n : {"/libamdhip64.so", "/libamdhip64.so.7", "/libamdhip64.so.6"})
        if (access((dir + n).c_str(), F_OK) == 0) return true;
    return false;
}
// the rule on its own (no process state), so that it can be checked without arranging a second runtime
extern "C" int pbr_runtime_mismatch_dirs(const char* hip_dir, const char* torch_dir) {
    if (!hip_dir || !torch_dir || !*hip_dir || !*torch_dir) return 0;   // no torch, or no HIP runtime mapped yet
    if (std::strcmp(hip_dir, torch_dir) == 0) return 0;                  // torch's own copy is the one in use
    return dir_has_hip_runtime(torch_dir) ? 1 : 0;                       // torch without a bundled runtime uses the system one too
}
static bool runtime_mismatch() {
    RuntimeScan s;
    dl_iterate_phdr(scan_cb, &s);
    if (!pbr_runtime_mismatch_dirs(s.hip_dir.c_str(), s.torch_dir.c_str())) return false;
    snprintf(g_runtime_err, sizeof(g_runtime_err),
             "two ROCm installations in one process: the HIP runtime in use is %s/libamdhip64 but PyTorch (%s) ships its own; "
             "load torch BEFORE libpbr_hip.so / libpbr_host.so (direct12pbrrenderer_amd._lib.load() does)", s.hip_dir.c_str(), s.torch_dir.c_str());
    return true;
}

const char* pbr_runtime_error(void) { return runtime_mismatch() ? g_runtime_err : nullptr; }

pbr_status pbr_ctx_create(int hip_device, pbr_ctx** out) {
    if (!out) return PBR_ERR_INVALID;
    *out = nullptr;
    if (runtime_mismatch()) { fprintf(stderr, "pbr_ctx_create: %s\n", g_runtime_err); return PBR_ERR_UNSUPPORTED; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || hip_device < 0 || hip_device >= n) return PBR_ERR_HIP;
    if (hipSetDevice(hip_device) != hipSuccess) return PBR_ERR_HIP;
    pbr_ctx* c = new (std::nothrow) pbr_ctx();
    if (!c) return PBR_ERR_NOMEM;
    c->device = hip_device;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return PBR_ERR_HIP; }
    c->stream = c->own_stream;
    if (hipMalloc(&c->scratch, pbr::SCRATCH_BYTES) != hipSuccess) {
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return PBR_ERR_NOMEM;
    }
    c->scratch_bytes = pbr::SCRATCH_BYTES;
    if (hipDeviceGetAttribute(&c->cu_count, hipDeviceAttributeMultiprocessorCount, hip_device) != hipSuccess || c->cu_count < 1) c->cu_count = 256;   // MI355X
    *out = c;
    return PBR_OK;
}

static void comm_teardown(pbr_ctx* ctx);

void pbr_ctx_destroy(pbr_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    comm_teardown(ctx);
    if (ctx->side_stream) { (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); }
    if (ctx->ev_side_fork) (void)hipEventDestroy(ctx->ev_side_fork);
    if (ctx->ev_side_join) (void)hipEventDestroy(ctx->ev_side_join);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->pf_dev) (void)hipFree(ctx->pf_dev);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

pbr_status pbr_ctx_set_stream(pbr_ctx* ctx, void* hip_stream) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, !ctx->on_side, "pbr_ctx_set_stream: on the side stream (pbr_ctx_side_end first)");
    ctx->stream = (hipStream_t)hip_stream;   // NULL is HIP's default (null) stream, a legal target
    return PBR_OK;
}

pbr_status pbr_ctx_use_own_stream(pbr_ctx* ctx) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, !ctx->on_side, "pbr_ctx_use_own_stream: on the side stream (pbr_ctx_side_end first)");
    ctx->stream = ctx->own_stream;
    return PBR_OK;
}

pbr_status pbr_ctx_set_bloom_shader_order(pbr_ctx* ctx, int on) {
    if (!ctx) return PBR_ERR_INVALID;
    ctx->bloom_shader_order = on != 0;
    return PBR_OK;
}

void* pbr_ctx_get_stream(const pbr_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

const char* pbr_last_error(const pbr_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

pbr_status pbr_sync(pbr_ctx* ctx) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // work forked to the side stream (pbr_ctx_side_begin .. _end) and not joined yet belongs to "everything enqueued so far"
    if (ctx->side_stream && (ctx->side_pending || ctx->on_side) && ctx->stream != ctx->side_stream)
        PBR_HIP(ctx, hipStreamSynchronize(ctx->side_stream));
    return PBR_OK;
}

// ------------------------------------------------------------------------------------------- RCCL
// Types, enum values and signatures come from <rccl/rccl.h> (included at the top of this file); the library itself is still
// bound with dlopen so that a one-GPU process never needs librccl.  Every dlsym'd pointer is typed with decltype(&ncclXxx):
// a signature drift between the header and this file fails the build instead of the first 8-GPU run.
#if PBR_HAVE_RCCL_HEADER   // pin the header to what this file (and its restated fallback declarations) assume
static_assert(sizeof(ncclUniqueId) == 128 && NCCL_UNIQUE_ID_BYTES == 128, "pbr_comm_unique_id / pbr_comm_init hand the id over as 128 opaque bytes (pbr_hip.h)");
static_assert(ncclInt8 == 0 && ncclUint32 == 3 && ncclSum == 0 && ncclSuccess == 0, "RCCL enum values this file was written against");
#endif
static_assert(sizeof(ncclComm_t) == sizeof(void*) && sizeof(ncclUniqueId) == 128, "pbr_ctx keeps the communicators as opaque pointers, the id travels as 128 bytes");

#define RCCL_FN(lib, name) ((lib) ? reinterpret_cast<decltype(&name)>(dlsym((lib), #name)) : nullptr)

static void* open_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    // a copy that is already mapped (PyTorch brings its own librccl.so.1) is reused: one RCCL per process
    for (const char* n : names) {
        void* h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (h) return h;
    }
    for (const char* n : names) {
        void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) return h;
    }
    return nullptr;
}

pbr_status pbr_comm_unique_id(void* out_128_bytes) {
    if (!out_128_bytes) return PBR_ERR_INVALID;
    void* lib = open_rccl();
    if (!lib) return PBR_ERR_COMM;
    auto f = RCCL_FN(lib, ncclGetUniqueId);
    if (!f) return PBR_ERR_COMM;
    ncclUniqueId id;
    if (f(&id) != ncclSuccess) return PBR_ERR_COMM;
    std::memcpy(out_128_bytes, &id, sizeof(id));
    return PBR_OK;
}

static void comm_reset(pbr_ctx* ctx) { ctx->comm = ctx->comm_hist = nullptr; }

// Destroys whatever communicators the context holds.  world / rank keep what the caller asked for: after a failed
// pbr_comm_init(world > 1) BOTH collectives refuse ("no communicator") — a context is never half in multi-GPU mode
// (one collective working beside one refusing), and never silently single-GPU when the caller asked for more.
static void comm_teardown(pbr_ctx* ctx) {
    if (ctx->rccl_lib) {
        auto d = RCCL_FN(ctx->rccl_lib, ncclCommDestroy);
        if (d && ctx->comm_hist) (void)d((ncclComm_t)ctx->comm_hist);
        if (d && ctx->comm) (void)d((ncclComm_t)ctx->comm);
    }
    comm_reset(ctx);
}

pbr_status pbr_comm_init(pbr_ctx* ctx, int world, int rank, const void* unique_id_128_bytes) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world, "pbr_comm_init: bad world/rank");
    // a call that is REFUSED (bad arguments, a second init of a context that has a communicator) changes nothing — world and rank
    // included (they used to be overwritten before the "already has a communicator" check).  A call that is accepted and
    // then FAILS inside RCCL keeps the world / rank the caller asked for, with no communicator: both collectives refuse from then on
    // (comm_teardown), the context never falls back to single-GPU behaviour silently.
    PBR_REQUIRE(ctx, world == 1 || unique_id_128_bytes != nullptr, "pbr_comm_init: null unique id");
    // (any init, also a world-1 one with no id: it would set world / rank to 1 / 0 and leave the old communicators attached,
    //  and pbr_allreduce_hist would go on reducing over them — ADVICE r05)
    PBR_REQUIRE(ctx, ctx->comm == nullptr && ctx->comm_hist == nullptr, "pbr_comm_init: the context already has a communicator");
    ctx->world = world;
    ctx->rank = rank;
    // world 1 needs no communicator; with a unique id one is created all the same (a 1-rank RCCL communicator is
    // legal) so that the RCCL entry points can be exercised on a single GPU
    if (world == 1 && unique_id_128_bytes == nullptr) return PBR_OK;
    if (!ctx->rccl_lib) ctx->rccl_lib = open_rccl();
    if (!ctx->rccl_lib) return pbr::fail(ctx, PBR_ERR_COMM, "pbr_comm_init: librccl not found");
    auto f = RCCL_FN(ctx->rccl_lib, ncclCommInitRank);
    if (!f) return pbr::fail(ctx, PBR_ERR_COMM, "pbr_comm_init: ncclCommInitRank missing");
    ncclUniqueId id;
    std::memcpy(&id, unique_id_128_bytes, sizeof(id));
    PBR_HIP(ctx, hipSetDevice(ctx->device));
    ncclComm_t comm = nullptr, comm_hist = nullptr;
    ncclResult_t r = f(&comm, world, id, rank);
    if (r != ncclSuccess || !comm) { comm_reset(ctx); return pbr::fail(ctx, PBR_ERR_COMM, "ncclCommInitRank failed"); }
    ctx->comm = comm;
    // A second communicator over the same ranks for the histogram all-reduce: the halo exchange (frame stream) and the
    // all-reduce (side stream when the frame's tail is overlapped) may then be in flight at the same time without
    // relying on every rank enqueueing them in the same host order — RCCL orders operations per communicator.
    // If it cannot be made the frame communicator goes as well (no half-initialised state: a halo exchange that works
    // beside an all-reduce that refuses would hang the other ranks at their first average).
    auto split = RCCL_FN(ctx->rccl_lib, ncclCommSplit);
    if (!split) { comm_teardown(ctx); return pbr::fail(ctx, PBR_ERR_COMM, "pbr_comm_init: ncclCommSplit missing (RCCL too old)"); }
    r = split(comm, 0, rank, &comm_hist, nullptr);
    if (r != ncclSuccess || !comm_hist) { comm_teardown(ctx); return pbr::fail(ctx, PBR_ERR_COMM, "ncclCommSplit failed"); }
    ctx->comm_hist = comm_hist;
    return PBR_OK;
}

pbr_status pbr_allreduce_hist(pbr_ctx* ctx, uint32_t* hist256) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, hist256 != nullptr, "pbr_allreduce_hist: null histogram");
    if (ctx->world <= 1 && !ctx->comm_hist) return PBR_OK;   // single GPU: the local histogram is the global one
    if (!ctx->comm_hist) return pbr::fail(ctx, PBR_ERR_COMM, "pbr_allreduce_hist: world > 1 but no communicator");
    auto f = RCCL_FN(ctx->rccl_lib, ncclAllReduce);
    if (!f) return pbr::fail(ctx, PBR_ERR_COMM, "ncclAllReduce missing");
    const ncclResult_t r = f(hist256, hist256, PBR_HISTOGRAM_BINS, ncclUint32, ncclSum, (ncclComm_t)ctx->comm_hist, ctx->stream);
    if (r != ncclSuccess) return pbr::fail(ctx, PBR_ERR_COMM, "ncclAllReduce failed");
    return PBR_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------- HBM streaming-read probe
// The "measured HBM-read roofline" bench.py reports next to the 8 TB/s nominal figure: every lane streams 16-byte
// loads with a grid stride (one 1 KiB segment per wave and trip, 4 loads in flight per lane), folds them with xor so
// the loads cannot be dropped, and a block writes one word.  Run on a buffer several times the 256 MiB Infinity Cache.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_membench_read(const u32x4* __restrict__ buf, size_t n16, uint32_t* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    u32x4 a = {0, 0, 0, 0};
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 v0 = __builtin_nontemporal_load(buf + i), v1 = __builtin_nontemporal_load(buf + i + stride);
        const u32x4 v2 = __builtin_nontemporal_load(buf + i + 2 * stride), v3 = __builtin_nontemporal_load(buf + i + 3 * stride);
        a.x ^= v0.x ^ v1.x ^ v2.x ^ v3.x; a.y ^= v0.y ^ v1.y ^ v2.y ^ v3.y;
        a.z ^= v0.z ^ v1.z ^ v2.z ^ v3.z; a.w ^= v0.w ^ v1.w ^ v2.w ^ v3.w;
    }
    for (; i < n16; i += stride) { const u32x4 v = buf[i]; a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; }
    uint32_t r = a.x ^ a.y ^ a.z ^ a.w;
    for (int o = 32; o > 0; o >>= 1) r ^= __shfl_xor(r, o);
    if ((threadIdx.x & 63) == 0) atomicXor(&sink[blockIdx.x], r);
}

extern "C" pbr_status pbr_membench_read(pbr_ctx* ctx, const void* buf, size_t bytes, uint32_t* sink, uint32_t blocks) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, buf && sink && bytes >= 16 && ((uintptr_t)buf & 15u) == 0 && blocks >= 1 && blocks <= 65535, "pbr_membench_read: bad arguments");
    hipLaunchKernelGGL(k_membench_read, dim3(blocks), dim3(256), 0, ctx->stream, (const u32x4*)buf, bytes / 16, sink);
    return pbr::launched(ctx, "k_membench_read");
}

// ------------------------------------------------------------------------------------------- VALU issue-rate probe
// (the loop of tools/valu_rate3.hip as a library entry: bench.py measures the issue rate and the sustained shader clock on the
//  box it runs on.)  Eight independent accumulators per lane, so a wave never waits for its own previous result.
typedef float vb_f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k_valubench(uint64_t* __restrict__ stamps, int iters) {
    float x0 = threadIdx.x + 1.5f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const float y0 = 1.0f + 1e-7f * threadIdx.x, y1 = 1.0f - 1e-7f * threadIdx.x;
    vb_f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    const vb_f2 q0 = {y0, y1}, q1 = {y1 * 1e-9f, y0 * 1e-9f};
    const uint64_t r0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (OP == 0)
            asm volatile("v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %9\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %9\n\t"
                         "v_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %9\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %9"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y0), "v"(y1));
        else if (OP == 1)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %9, %8\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %9, %8\n\t"
                         "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %9, %8\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %9, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y0), "v"(y1));
        else if (OP == 2)
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %9, %8\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %9, %8\n\t"
                         "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %9, %8\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %9, %8"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q0), "v"(q1));
        else
            asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\t"
                         "v_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_rcp_f32 %6, %6\n\tv_rcp_f32 %7, %7"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    }
    const uint64_t c1 = clock64(), r1 = wall_clock64();
    // keep every accumulator alive without a store the timed loop could be blamed for
    asm volatile("" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7));
    if ((threadIdx.x & 63) == 0) {
        uint64_t* s = stamps + 4 * ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6));
        s[0] = c0; s[1] = c1; s[2] = r0; s[3] = r1;
    }
}

extern "C" pbr_status pbr_valubench(pbr_ctx* ctx, uint32_t op, uint32_t blocks, uint32_t iters, uint64_t* stamps) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, stamps && op <= 3 && blocks >= 1 && blocks <= 65535 && iters >= 1 && iters <= (1u << 24), "pbr_valubench: bad arguments");
    const dim3 g(blocks), b(256);
    if (op == 0) hipLaunchKernelGGL(k_valubench<0>, g, b, 0, ctx->stream, stamps, (int)iters);
    else if (op == 1) hipLaunchKernelGGL(k_valubench<1>, g, b, 0, ctx->stream, stamps, (int)iters);
    else if (op == 2) hipLaunchKernelGGL(k_valubench<2>, g, b, 0, ctx->stream, stamps, (int)iters);
    else hipLaunchKernelGGL(k_valubench<3>, g, b, 0, ctx->stream, stamps, (int)iters);
    return pbr::launched(ctx, "k_valubench");
}

// ------------------------------------------------------------------------------------------- halo exchange
// Level-1 strips of the bloom pyramid between neighbouring tiles (SURVEY 8e option 2).  One pack launch gathers every
// outgoing rectangle of the plane into a contiguous staging area, one ncclGroup sends / receives all strips, one
// unpack launch scatters what arrived: three enqueues per frame however many neighbours a tile has.
constexpr int HALO_MAX_PEERS = 16;
struct HaloRects {
    int n;
    int x[HALO_MAX_PEERS], y[HALO_MAX_PEERS], w[HALO_MAX_PEERS], h[HALO_MAX_PEERS];
    uint32_t off[HALO_MAX_PEERS];   // texel offset of the rectangle in the staging area
};
// grid (ceil(max_texels / 256), n): block row r copies rectangle r; TO_STAGING: plane -> staging, else staging -> plane
template <bool TO_STAGING>
__global__ __launch_bounds__(256) void k_halo_copy(uint2* __restrict__ plane, int pitch, uint2* __restrict__ staging, HaloRects rc) {
    const int r = blockIdx.y;
    const int w = rc.w[r], n = w * rc.h[r];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int yy = i / w, xx = i - yy * w;
        const size_t p = (size_t)(rc.y[r] + yy) * pitch + (rc.x[r] + xx);
        if (TO_STAGING) staging[rc.off[r] + i] = plane[p];
        else plane[p] = staging[rc.off[r] + i];
    }
}

extern "C" {

size_t pbr_halo_staging_bytes(const pbr_halo_peer* peers, uint32_t n_peers) {
    size_t t = 0;
    for (uint32_t i = 0; peers && i < n_peers; i++) t += (size_t)peers[i].send[2] * peers[i].send[3] + (size_t)peers[i].recv[2] * peers[i].recv[3];
    return t * 8;
}

static pbr_status halo_exchange_on(pbr_ctx* ctx, hipStream_t stream, pbr_half* plane, uint32_t pitch, uint32_t rows,
                                   const pbr_halo_peer* peers, uint32_t n_peers, void* staging, size_t staging_bytes) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, plane && pitch >= 1 && rows >= 1, "pbr_halo_exchange: null plane");
    if (n_peers == 0) return PBR_OK;
    PBR_REQUIRE(ctx, peers && n_peers <= (uint32_t)HALO_MAX_PEERS, "pbr_halo_exchange: bad peer list");
    PBR_REQUIRE(ctx, staging && staging_bytes >= pbr_halo_staging_bytes(peers, n_peers), "pbr_halo_exchange: staging area too small");
    HaloRects snd{}, rcv{};
    uint32_t off = 0;
    int max_s = 0, max_r = 0;
    int peer_of_s[HALO_MAX_PEERS], peer_of_r[HALO_MAX_PEERS];
    for (uint32_t i = 0; i < n_peers; i++) {
        const pbr_halo_peer& p = peers[i];
        PBR_REQUIRE(ctx, p.send[0] + p.send[2] <= pitch && p.send[1] + p.send[3] <= rows && p.recv[0] + p.recv[2] <= pitch && p.recv[1] + p.recv[3] <= rows,
                    "pbr_halo_exchange: rectangle outside the plane");
        PBR_REQUIRE(ctx, p.rank >= 0 && p.rank < ctx->world && (p.rank != ctx->rank || ctx->world == 1), "pbr_halo_exchange: bad peer rank");
        if (p.send[2] && p.send[3]) {
            const int k = snd.n++;
            snd.x[k] = (int)p.send[0]; snd.y[k] = (int)p.send[1]; snd.w[k] = (int)p.send[2]; snd.h[k] = (int)p.send[3]; snd.off[k] = off;
            off += p.send[2] * p.send[3];
            peer_of_s[k] = p.rank;
            if (snd.w[k] * snd.h[k] > max_s) max_s = snd.w[k] * snd.h[k];
        }
    }
    for (uint32_t i = 0; i < n_peers; i++) {
        const pbr_halo_peer& p = peers[i];
        if (p.recv[2] && p.recv[3]) {
            const int k = rcv.n++;
            rcv.x[k] = (int)p.recv[0]; rcv.y[k] = (int)p.recv[1]; rcv.w[k] = (int)p.recv[2]; rcv.h[k] = (int)p.recv[3]; rcv.off[k] = off;
            off += p.recv[2] * p.recv[3];
            peer_of_r[k] = p.rank;
            if (rcv.w[k] * rcv.h[k] > max_r) max_r = rcv.w[k] * rcv.h[k];
        }
    }
    if (!ctx->comm) return pbr::fail(ctx, PBR_ERR_COMM, "pbr_halo_exchange: no communicator (pbr_comm_init)");
    auto f_send = RCCL_FN(ctx->rccl_lib, ncclSend);
    auto f_recv = RCCL_FN(ctx->rccl_lib, ncclRecv);
    auto f_gs = RCCL_FN(ctx->rccl_lib, ncclGroupStart);
    auto f_ge = RCCL_FN(ctx->rccl_lib, ncclGroupEnd);
    if (!f_send || !f_recv || !f_gs || !f_ge) return pbr::fail(ctx, PBR_ERR_COMM, "pbr_halo_exchange: ncclSend/ncclRecv/ncclGroup* missing");
    uint2* st = (uint2*)staging;
    if (snd.n) {
        const int bx = (max_s + 255) / 256 > 256 ? 256 : (max_s + 255) / 256;
        hipLaunchKernelGGL(k_halo_copy<true>, dim3(bx, snd.n), dim3(256), 0, stream, (uint2*)plane, (int)pitch, st, snd);
        pbr_status r = pbr::launched(ctx, "k_halo_copy<pack>");
        if (r) return r;
    }
    if (f_gs() != ncclSuccess) return pbr::fail(ctx, PBR_ERR_COMM, "ncclGroupStart failed");
    ncclResult_t rc = ncclSuccess;
    for (int k = 0; k < snd.n && rc == ncclSuccess; k++) rc = f_send(st + snd.off[k], (size_t)snd.w[k] * snd.h[k] * 8, ncclInt8, peer_of_s[k], (ncclComm_t)ctx->comm, stream);
    for (int k = 0; k < rcv.n && rc == ncclSuccess; k++) rc = f_recv(st + rcv.off[k], (size_t)rcv.w[k] * rcv.h[k] * 8, ncclInt8, peer_of_r[k], (ncclComm_t)ctx->comm, stream);
    const ncclResult_t ge = f_ge();
    if (rc != ncclSuccess || ge != ncclSuccess) return pbr::fail(ctx, PBR_ERR_COMM, "ncclSend/ncclRecv group failed");
    if (rcv.n) {
        const int bx = (max_r + 255) / 256 > 256 ? 256 : (max_r + 255) / 256;
        hipLaunchKernelGGL(k_halo_copy<false>, dim3(bx, rcv.n), dim3(256), 0, stream, (uint2*)plane, (int)pitch, st, rcv);
        pbr_status r = pbr::launched(ctx, "k_halo_copy<unpack>");
        if (r) return r;
    }
    return PBR_OK;
}

pbr_status pbr_halo_exchange(pbr_ctx* ctx, pbr_half* plane, uint32_t pitch, uint32_t rows,
                             const pbr_halo_peer* peers, uint32_t n_peers, void* staging, size_t staging_bytes) {
    if (!ctx) return PBR_ERR_INVALID;
    return halo_exchange_on(ctx, ctx->stream, plane, pitch, rows, peers, n_peers, staging, staging_bytes);
}

// ---- side stream: a second, HIGH-PRIORITY stream of the context ------------------------------------------------------
// pbr_ctx_side_begin : the side stream waits for everything enqueued so far; calls made until _end enqueue THERE;
// pbr_ctx_side_end   : back to the context's stream — what follows runs concurrently with the side stream's work;
// pbr_ctx_side_join  : the context's stream waits for the side stream.
// The overlapped multi-GPU frame puts the tile's border ring (shade, level-1 strips, halo exchange) on the side stream
// and shades the core on the main one: the dispatcher serves the high-priority queue first, the core's blocks fill
// whatever the ring leaves free, and the strips travel while the core is still being shaded.
static pbr_status ensure_side(pbr_ctx* ctx) {
    if (ctx->side_stream) return PBR_OK;
    PBR_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->side_cu_mask.empty()) {   // pbr_ctx_set_cu_masks: the side stream owns a set of CUs (no priority: it does not compete)
        PBR_HIP(ctx, hipExtStreamCreateWithCUMask(&ctx->side_stream, (uint32_t)ctx->side_cu_mask.size(), ctx->side_cu_mask.data()));
    } else {
        int lo = 0, hi = 0;
        PBR_HIP(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));   // hi = numerically lowest = greatest priority
        PBR_HIP(ctx, hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, hi));
    }
    if (ctx->ev_side_fork) return PBR_OK;
    PBR_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_side_fork, hipEventDisableTiming));
    PBR_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_side_join, hipEventDisableTiming));
    return PBR_OK;
}

#ifdef PBR_DEBUG_KNOBS   // knobs build only since round 6 (include/pbr_hip.h, last section): tools/cu_partition.py and its test load libpbr_hip_knobs.so
pbr_status pbr_ctx_set_cu_masks(pbr_ctx* ctx, const uint32_t* main_mask, const uint32_t* side_mask, uint32_t words) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, !ctx->on_side && !ctx->side_pending, "pbr_ctx_set_cu_masks: side-stream work pending (pbr_ctx_side_end / _join first)");
    PBR_REQUIRE(ctx, (main_mask == nullptr && side_mask == nullptr) || (words >= 1 && words <= 64), "pbr_ctx_set_cu_masks: 1 .. 64 mask words");
    auto any = [&](const uint32_t* m) { for (uint32_t i = 0; m && i < words; i++) if (m[i]) return true; return m == nullptr; };
    PBR_REQUIRE(ctx, any(main_mask) && any(side_mask), "pbr_ctx_set_cu_masks: an empty CU mask");
    PBR_HIP(ctx, hipSetDevice(ctx->device));
    PBR_HIP(ctx, hipDeviceSynchronize());
    const bool on_own = ctx->stream == ctx->own_stream;
    hipStream_t fresh = nullptr;
    if (main_mask) PBR_HIP(ctx, hipExtStreamCreateWithCUMask(&fresh, words, main_mask));
    else PBR_HIP(ctx, hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
    (void)hipStreamDestroy(ctx->own_stream);
    ctx->own_stream = fresh;
    if (on_own) ctx->stream = fresh;
    if (ctx->side_stream) { (void)hipStreamDestroy(ctx->side_stream); ctx->side_stream = nullptr; }
    ctx->side_cu_mask.assign(side_mask ? side_mask : nullptr, side_mask ? side_mask + words : nullptr);
    return ensure_side(ctx);
}
#endif

pbr_status pbr_ctx_side_begin(pbr_ctx* ctx) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, ctx->main_saved == nullptr && !ctx->on_side, "pbr_ctx_side_begin: already on the side stream");
    pbr_status r = ensure_side(ctx);
    if (r) return r;
    PBR_HIP(ctx, hipEventRecord(ctx->ev_side_fork, ctx->stream));
    PBR_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_side_fork, 0));
    ctx->main_saved = ctx->stream;
    ctx->main_was_null = ctx->stream == nullptr;
    ctx->stream = ctx->side_stream;
    ctx->on_side = true;
    return PBR_OK;
}

pbr_status pbr_ctx_side_end(pbr_ctx* ctx) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, ctx->on_side, "pbr_ctx_side_end: not on the side stream");
    PBR_HIP(ctx, hipEventRecord(ctx->ev_side_join, ctx->side_stream));
    ctx->stream = ctx->main_saved;
    ctx->main_saved = nullptr;
    ctx->on_side = false;
    ctx->side_pending = true;
    return PBR_OK;
}

pbr_status pbr_ctx_side_join(pbr_ctx* ctx) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, !ctx->on_side, "pbr_ctx_side_join: still on the side stream (pbr_ctx_side_end first)");
    if (!ctx->side_pending) return PBR_OK;
    ctx->side_pending = false;
    PBR_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side_join, 0));
    return PBR_OK;
}

// The two halves of pbr_halo_exchange as separate calls, for transports other than the context's RCCL communicator
// (tests, torch.distributed): pack = every send rectangle -> staging; unpack = staging -> every recv rectangle.  The
// staging layout is the one pbr_halo_exchange uses: all send rectangles in peer order, then all recv rectangles.
pbr_status pbr_halo_pack(pbr_ctx* ctx, pbr_half* plane, uint32_t pitch, uint32_t rows,
                         const pbr_halo_peer* peers, uint32_t n_peers, void* staging, size_t staging_bytes, int unpack) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, plane && pitch >= 1 && rows >= 1, "pbr_halo_pack: null plane");
    if (n_peers == 0) return PBR_OK;
    PBR_REQUIRE(ctx, peers && n_peers <= (uint32_t)HALO_MAX_PEERS, "pbr_halo_pack: bad peer list");
    PBR_REQUIRE(ctx, staging && staging_bytes >= pbr_halo_staging_bytes(peers, n_peers), "pbr_halo_pack: staging area too small");
    HaloRects rc{};
    uint32_t off = 0;
    int mx = 0;
    for (int pass = 0; pass < 2; pass++) {
        for (uint32_t i = 0; i < n_peers; i++) {
            const uint32_t* q = pass == 0 ? peers[i].send : peers[i].recv;
            PBR_REQUIRE(ctx, q[0] + q[2] <= pitch && q[1] + q[3] <= rows, "pbr_halo_pack: rectangle outside the plane");
            if (!(q[2] && q[3])) continue;
            if ((pass == 1) == (unpack != 0)) {
                const int k = rc.n++;
                rc.x[k] = (int)q[0]; rc.y[k] = (int)q[1]; rc.w[k] = (int)q[2]; rc.h[k] = (int)q[3]; rc.off[k] = off;
                if (rc.w[k] * rc.h[k] > mx) mx = rc.w[k] * rc.h[k];
            }
            off += q[2] * q[3];
        }
    }
    if (!rc.n) return PBR_OK;
    const int bx = (mx + 255) / 256 > 256 ? 256 : (mx + 255) / 256;
    if (unpack) hipLaunchKernelGGL(k_halo_copy<false>, dim3(bx, rc.n), dim3(256), 0, ctx->stream, (uint2*)plane, (int)pitch, (uint2*)staging, rc);
    else hipLaunchKernelGGL(k_halo_copy<true>, dim3(bx, rc.n), dim3(256), 0, ctx->stream, (uint2*)plane, (int)pitch, (uint2*)staging, rc);
    return pbr::launched(ctx, "k_halo_copy");
}

}  // extern "C"
