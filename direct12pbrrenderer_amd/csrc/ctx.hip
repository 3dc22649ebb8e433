// ctx.hip — context, layout helpers and the RCCL histogram all-reduce of the C ABI.
#include "pbr_internal.hpp"
#include "pbr_device.hpp"
#include <dlfcn.h>
#include <cstring>

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

pbr_status pbr_ctx_create(int hip_device, pbr_ctx** out) {
    if (!out) return PBR_ERR_INVALID;
    *out = nullptr;
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
    *out = c;
    return PBR_OK;
}

typedef int (*nccl_destroy_fn)(void*);

void pbr_ctx_destroy(pbr_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm && ctx->rccl_lib) {
        nccl_destroy_fn d = (nccl_destroy_fn)dlsym(ctx->rccl_lib, "ncclCommDestroy");
        if (d) d(ctx->comm);
    }
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

pbr_status pbr_ctx_set_stream(pbr_ctx* ctx, void* hip_stream) {
    if (!ctx) return PBR_ERR_INVALID;
    ctx->stream = (hipStream_t)hip_stream;   // NULL is HIP's default (null) stream, a legal target
    return PBR_OK;
}

pbr_status pbr_ctx_use_own_stream(pbr_ctx* ctx) {
    if (!ctx) return PBR_ERR_INVALID;
    ctx->stream = ctx->own_stream;
    return PBR_OK;
}

const char* pbr_last_error(const pbr_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

pbr_status pbr_sync(pbr_ctx* ctx) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PBR_OK;
}

// ------------------------------------------------------------------------------------------- RCCL
// ncclUniqueId is 128 opaque bytes passed BY VALUE to ncclCommInitRank (rccl.h).
struct nccl_uid { char internal[128]; };
typedef int (*nccl_get_uid_fn)(nccl_uid*);
typedef int (*nccl_init_rank_fn)(void**, int, nccl_uid, int);
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
enum { NCCL_UINT32 = 3, NCCL_SUM = 0 };

static void* open_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
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
    nccl_get_uid_fn f = (nccl_get_uid_fn)dlsym(lib, "ncclGetUniqueId");
    if (!f) return PBR_ERR_COMM;
    nccl_uid id;
    if (f(&id) != 0) return PBR_ERR_COMM;
    std::memcpy(out_128_bytes, &id, sizeof(id));
    return PBR_OK;
}

pbr_status pbr_comm_init(pbr_ctx* ctx, int world, int rank, const void* unique_id_128_bytes) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world, "pbr_comm_init: bad world/rank");
    ctx->world = world;
    ctx->rank = rank;
    if (world == 1) return PBR_OK;
    PBR_REQUIRE(ctx, unique_id_128_bytes != nullptr, "pbr_comm_init: null unique id");
    if (!ctx->rccl_lib) ctx->rccl_lib = open_rccl();
    if (!ctx->rccl_lib) return pbr::fail(ctx, PBR_ERR_COMM, "pbr_comm_init: librccl not found");
    nccl_init_rank_fn f = (nccl_init_rank_fn)dlsym(ctx->rccl_lib, "ncclCommInitRank");
    if (!f) return pbr::fail(ctx, PBR_ERR_COMM, "pbr_comm_init: ncclCommInitRank missing");
    nccl_uid id;
    std::memcpy(&id, unique_id_128_bytes, sizeof(id));
    PBR_HIP(ctx, hipSetDevice(ctx->device));
    int r = f(&ctx->comm, world, id, rank);
    if (r != 0) { ctx->comm = nullptr; return pbr::fail(ctx, PBR_ERR_COMM, "ncclCommInitRank failed"); }
    return PBR_OK;
}

pbr_status pbr_allreduce_hist(pbr_ctx* ctx, uint32_t* hist256) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, hist256 != nullptr, "pbr_allreduce_hist: null histogram");
    if (ctx->world <= 1) return PBR_OK;   // single GPU: the local histogram is the global one
    if (!ctx->comm) return pbr::fail(ctx, PBR_ERR_COMM, "pbr_allreduce_hist: world > 1 but no communicator");
    nccl_allreduce_fn f = (nccl_allreduce_fn)dlsym(ctx->rccl_lib, "ncclAllReduce");
    if (!f) return pbr::fail(ctx, PBR_ERR_COMM, "ncclAllReduce missing");
    int r = f(hist256, hist256, PBR_HISTOGRAM_BINS, NCCL_UINT32, NCCL_SUM, ctx->comm, ctx->stream);
    if (r != 0) return pbr::fail(ctx, PBR_ERR_COMM, "ncclAllReduce failed");
    return PBR_OK;
}

}  // extern "C"
