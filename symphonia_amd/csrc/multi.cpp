// Multi-GPU at the C level: what a Rust / C++ host needs to do BASELINE config 4's "batch split / gather" without Python --
// shard arithmetic, and one-to-all scatter / all-to-one gather of chain-major shards over RCCL (xGMI) with ncclSend / ncclRecv
// inside one group.  Chains (one channel of one stream) are independent: the split is by whole streams, the data
// path of the synthesis itself never communicates; these two calls exist for callers whose batch starts and ends on one rank.
//
// RCCL is resolved at first use (dlopen of librccl.so: a process that never touches these entry points does not need the
// library), or replaced by a caller-supplied transport (symaccel_multi_set_transport): MPI, shared memory, or the in-process
// mailbox of tests/test_multi_c.py.
#include <dlfcn.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "symaccel_internal.h"

namespace symaccel {

namespace {

// the RCCL entry points used, with their C signatures (rccl.h: ncclResult_t == int, ncclDataType_t ncclInt8 == 0)
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(void *id) = nullptr;
    int (*CommInitRank)(void **comm, int nranks, symaccel_unique_id id, int rank) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream) = nullptr;
    int (*Recv)(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
// The caller-supplied transport: published with release / read with acquire, so that rank threads started after
// symaccel_multi_set_transport see a complete table (setting it while an exchange is in flight is still the caller's bug).
symaccel_transport g_transport_slots[2]{};
std::atomic<const symaccel_transport *> g_transport_ptr{nullptr};  // null: RCCL
#define g_have_transport (g_transport_ptr.load(std::memory_order_acquire) != nullptr)
// (every use loads the pointer ONCE into a local: a concurrent symaccel_multi_set_transport(NULL) between a check and a second load
// would be a null dereference)
inline const symaccel_transport *transport_now() { return g_transport_ptr.load(std::memory_order_acquire); }

void load_rccl() {
    const char *names[] = {std::getenv("SYMACCEL_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        if (!n || !*n) continue;
        void *h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        Rccl r;
        r.handle = h;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(h, "ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(h, "ncclRecv"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv) {
            g_rccl = r;
            return;
        }
        dlclose(h);
    }
}

const Rccl *rccl() {
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.handle ? &g_rccl : nullptr;
}

int rccl_fail(symaccel_ctx *ctx, int rc, const char *where) {
    if (ctx) {
        const Rccl *r = rccl();
        ctx->last_error = std::string(where) + ": " + ((r && r->GetErrorString) ? r->GetErrorString(rc) : "RCCL error");
    }
    return SYMACCEL_ERR_DEVICE;
}

// one peer-to-peer transfer through the transport in force
int xfer_send(symaccel_ctx *ctx, const void *buf, size_t bytes, int peer, void *comm, hipStream_t stream) {
    if (const symaccel_transport *t = transport_now()) return t->send(buf, bytes, peer, comm, (void *)stream) == 0 ? SYMACCEL_OK : SYMACCEL_ERR_DEVICE;
    const int rc = rccl()->Send(buf, bytes, 0, peer, comm, stream);
    return rc == 0 ? SYMACCEL_OK : rccl_fail(ctx, rc, "ncclSend");
}
int xfer_recv(symaccel_ctx *ctx, void *buf, size_t bytes, int peer, void *comm, hipStream_t stream) {
    if (const symaccel_transport *t = transport_now()) return t->recv(buf, bytes, peer, comm, (void *)stream) == 0 ? SYMACCEL_OK : SYMACCEL_ERR_DEVICE;
    const int rc = rccl()->Recv(buf, bytes, 0, peer, comm, stream);
    return rc == 0 ? SYMACCEL_OK : rccl_fail(ctx, rc, "ncclRecv");
}
int group_start(symaccel_ctx *ctx) {
    if (const symaccel_transport *t = transport_now()) return t->group_start ? (t->group_start() == 0 ? SYMACCEL_OK : SYMACCEL_ERR_DEVICE) : SYMACCEL_OK;
    const int rc = rccl()->GroupStart();
    return rc == 0 ? SYMACCEL_OK : rccl_fail(ctx, rc, "ncclGroupStart");
}
int group_end(symaccel_ctx *ctx) {
    if (const symaccel_transport *t = transport_now()) return t->group_end ? (t->group_end() == 0 ? SYMACCEL_OK : SYMACCEL_ERR_DEVICE) : SYMACCEL_OK;
    const int rc = rccl()->GroupEnd();
    return rc == 0 ? SYMACCEL_OK : rccl_fail(ctx, rc, "ncclGroupEnd");
}

bool have_backend() { return g_have_transport || rccl() != nullptr; }

struct Slice {
    size_t first, count;  // in streams
};
Slice slice_of(size_t n_streams, int world, int rank) {
    const size_t base = n_streams / (size_t)world, extra = n_streams % (size_t)world;
    const size_t r = (size_t)rank;
    return Slice{r * base + (r < extra ? r : extra), base + (r < extra ? 1u : 0u)};
}

// scatter (to_root == false) or gather (to_root == true) of the ranks' stream slices; `d_all` is read / written on the root only
int exchange(symaccel_ctx *ctx, void *comm, int world, int rank, int root, void *d_all, void *d_mine, size_t n_streams,
             size_t bytes_per_stream, bool to_root) {
    if (!ctx || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world) return SYMACCEL_ERR_INVALID_ARG;
    if (n_streams == 0 || bytes_per_stream == 0) return SYMACCEL_OK;
    const Slice mine = slice_of(n_streams, world, rank);
    if ((mine.count && !d_mine) || (rank == root && !d_all)) return SYMACCEL_ERR_INVALID_ARG;
    // (`comm` is RCCL's communicator; a caller-supplied transport gets it passed through untouched and may not need one)
    if (world > 1 && !comm && !g_have_transport) return SYMACCEL_ERR_INVALID_ARG;
    if (world > 1 && !have_backend()) {
        ctx->last_error = "librccl.so not found (set SYMACCEL_RCCL_LIB or install a transport)";
        return SYMACCEL_ERR_UNSUPPORTED;
    }
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    char *all = static_cast<char *>(d_all);
    if (rank == root && mine.count) {
        // the root's own slice never leaves the device
        void *a = all + mine.first * bytes_per_stream;
        if (to_root) SYM_GPU(ctx, hipMemcpyAsync(a, d_mine, mine.count * bytes_per_stream, hipMemcpyDeviceToDevice, ctx->stream));
        else SYM_GPU(ctx, hipMemcpyAsync(d_mine, a, mine.count * bytes_per_stream, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (world == 1) return SYMACCEL_OK;
    // one group: the root posts world - 1 transfers (each over its own xGMI link), every other rank one
    SYM_TRY(group_start(ctx));
    int st = SYMACCEL_OK;
    if (rank == root) {
        for (int p = 0; p < world && st == SYMACCEL_OK; ++p) {
            if (p == root) continue;
            const Slice s = slice_of(n_streams, world, p);
            if (!s.count) continue;
            void *a = all + s.first * bytes_per_stream;
            st = to_root ? xfer_recv(ctx, a, s.count * bytes_per_stream, p, comm, ctx->stream) : xfer_send(ctx, a, s.count * bytes_per_stream, p, comm, ctx->stream);
        }
    } else if (mine.count) {
        st = to_root ? xfer_send(ctx, d_mine, mine.count * bytes_per_stream, root, comm, ctx->stream) : xfer_recv(ctx, d_mine, mine.count * bytes_per_stream, root, comm, ctx->stream);
    }
    const int ge = group_end(ctx);  // (always closed, also after a failed post)
    return st != SYMACCEL_OK ? st : ge;
}

// chunk c of n of a slice of `count` streams: [first, first + len) in the slice's own numbering
Slice chunk_of(size_t count, int n_chunks, int c) {
    const size_t a = count * (size_t)c / (size_t)n_chunks, b = count * (size_t)(c + 1) / (size_t)n_chunks;
    return Slice{a, b - a};
}

// One direction of one chunk on `stream`: the root posts world - 1 transfers in one group, every other rank one; the root's own
// piece is a device copy on the same stream.
int exchange_chunk(symaccel_ctx *ctx, void *comm, int world, int rank, int root, char *all, char *mine_buf, size_t n_streams,
                   size_t bytes_per_stream, int n_chunks, int c, bool to_root, hipStream_t stream) {
    const Slice mine = slice_of(n_streams, world, rank);
    const Slice mc = chunk_of(mine.count, n_chunks, c);
    if (rank == root && mc.count) {
        char *a = all + (mine.first + mc.first) * bytes_per_stream, *m = mine_buf + mc.first * bytes_per_stream;
        if (to_root) SYM_GPU(ctx, hipMemcpyAsync(a, m, mc.count * bytes_per_stream, hipMemcpyDeviceToDevice, stream));
        else SYM_GPU(ctx, hipMemcpyAsync(m, a, mc.count * bytes_per_stream, hipMemcpyDeviceToDevice, stream));
    }
    if (world == 1) return SYMACCEL_OK;
    SYM_TRY(group_start(ctx));
    int st = SYMACCEL_OK;
    if (rank == root) {
        for (int p = 0; p < world && st == SYMACCEL_OK; ++p) {
            if (p == root) continue;
            const Slice s = slice_of(n_streams, world, p);
            const Slice pc = chunk_of(s.count, n_chunks, c);
            if (!pc.count) continue;
            char *a = all + (s.first + pc.first) * bytes_per_stream;
            st = to_root ? xfer_recv(ctx, a, pc.count * bytes_per_stream, p, comm, stream) : xfer_send(ctx, a, pc.count * bytes_per_stream, p, comm, stream);
        }
    } else if (mc.count) {
        char *m = mine_buf + mc.first * bytes_per_stream;
        st = to_root ? xfer_send(ctx, m, mc.count * bytes_per_stream, root, comm, stream) : xfer_recv(ctx, m, mc.count * bytes_per_stream, root, comm, stream);
    }
    const int ge = group_end(ctx);
    return st != SYMACCEL_OK ? st : ge;
}

}  // namespace

}  // namespace symaccel

using namespace symaccel;

extern "C" {

int symaccel_shard_range(size_t n_streams, int world, int rank, size_t *first, size_t *count) {
    if (world < 1 || rank < 0 || rank >= world || !first || !count) return SYMACCEL_ERR_INVALID_ARG;
    const Slice s = slice_of(n_streams, world, rank);
    *first = s.first;
    *count = s.count;
    return SYMACCEL_OK;
}

int symaccel_multi_set_transport(const symaccel_transport *t) {
    if (!t) {
        g_transport_ptr.store(nullptr, std::memory_order_release);
        return SYMACCEL_OK;
    }
    if (!t->send || !t->recv) return SYMACCEL_ERR_INVALID_ARG;
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    // write the slot that is NOT published, then publish it
    symaccel_transport *slot = g_transport_ptr.load(std::memory_order_acquire) == &g_transport_slots[0] ? &g_transport_slots[1] : &g_transport_slots[0];
    *slot = *t;
    g_transport_ptr.store(slot, std::memory_order_release);
    return SYMACCEL_OK;
}

int symaccel_comm_unique_id(symaccel_unique_id *id) {
    if (!id) return SYMACCEL_ERR_INVALID_ARG;
    const Rccl *r = rccl();
    if (!r) return SYMACCEL_ERR_UNSUPPORTED;
    return r->GetUniqueId(id) == 0 ? SYMACCEL_OK : SYMACCEL_ERR_DEVICE;
}

int symaccel_comm_init(symaccel_ctx *ctx, const symaccel_unique_id *id, int world, int rank, void **comm) {
    if (!ctx || !id || !comm || world < 1 || rank < 0 || rank >= world) return SYMACCEL_ERR_INVALID_ARG;
    *comm = nullptr;
    const Rccl *r = rccl();
    if (!r) {
        ctx->last_error = "librccl.so not found (set SYMACCEL_RCCL_LIB)";
        return SYMACCEL_ERR_UNSUPPORTED;
    }
    DeviceGuard dev(ctx);  // a communicator belongs to the device that is current when it is created
    if (!dev.ok()) return dev.status();
    const int rc = r->CommInitRank(comm, world, *id, rank);
    return rc == 0 ? SYMACCEL_OK : rccl_fail(ctx, rc, "ncclCommInitRank");
}

int symaccel_comm_destroy(void *comm) {
    if (!comm) return SYMACCEL_OK;
    const Rccl *r = rccl();
    if (!r) return SYMACCEL_ERR_UNSUPPORTED;
    return r->CommDestroy(comm) == 0 ? SYMACCEL_OK : SYMACCEL_ERR_DEVICE;
}

int symaccel_exchange_pipelined(symaccel_ctx *ctx, void *comm, int world, int rank, int root, const void *d_all_in, void *d_mine_in,
                                size_t in_bytes_per_stream, void *d_all_out, void *d_mine_out, size_t out_bytes_per_stream,
                                size_t n_streams, int n_chunks, symaccel_step_fn step, void *user) {
    if (!ctx || !step || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || n_chunks < 1) return SYMACCEL_ERR_INVALID_ARG;
    if (n_streams == 0) return SYMACCEL_OK;
    if (in_bytes_per_stream == 0 || out_bytes_per_stream == 0) return SYMACCEL_ERR_INVALID_ARG;
    const Slice mine = slice_of(n_streams, world, rank);
    if ((mine.count && (!d_mine_in || !d_mine_out)) || (rank == root && (!d_all_in || !d_all_out))) return SYMACCEL_ERR_INVALID_ARG;
    if (world > 1 && !comm && !g_have_transport) return SYMACCEL_ERR_INVALID_ARG;
    if (world > 1 && !have_backend()) {
        ctx->last_error = "librccl.so not found (set SYMACCEL_RCCL_LIB or install a transport)";
        return SYMACCEL_ERR_UNSUPPORTED;
    }
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    // The transfers run on a second stream of the context (the one the pinned host paths use for their uploads), the steps on the
    // context's own: chunk c + 1 travels to the ranks while chunk c is processed and chunk c - 1's result travels back.  Every
    // rank posts the same sequence on its transfer stream -- S0, S1, G0, S2, G1, ... -- so the sends and receives pair up.
    if (!ctx->stage_in) SYM_GPU(ctx, hipStreamCreate(&ctx->stage_in));
    for (hipEvent_t &e : ctx->stage_events)
        if (!e) SYM_GPU(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipStream_t xs = ctx->stage_in;
    hipEvent_t *ev_in = &ctx->stage_events[0], *ev_k = &ctx->stage_events[2], ev_done = ctx->stage_events[4], ev_start = ctx->stage_events[5];
    char *all_in = static_cast<char *>(const_cast<void *>(d_all_in)), *all_out = static_cast<char *>(d_all_out);
    char *mine_in = static_cast<char *>(d_mine_in), *mine_out = static_cast<char *>(d_mine_out);
    // the transfer stream starts behind whatever the caller queued on the context's stream (the buffers' producers)
    SYM_GPU(ctx, hipEventRecord(ev_start, ctx->stream));
    SYM_GPU(ctx, hipStreamWaitEvent(xs, ev_start, 0));
    // (from here on every failure is an assignment to `st` and a break: each exit goes through the ev_done join below)
    auto gpu = [&](hipError_t e, const char *what) -> int { return e == hipSuccess ? SYMACCEL_OK : ctx_fail(ctx, e, what); };
    auto scatter = [&](int c) -> int {
        const int s = exchange_chunk(ctx, comm, world, rank, root, all_in, mine_in, n_streams, in_bytes_per_stream, n_chunks, c, false, xs);
        return s != SYMACCEL_OK ? s : gpu(hipEventRecord(ev_in[c & 1], xs), "hipEventRecord(scatter)");
    };
    int st = scatter(0);
    for (int c = 0; c < n_chunks && st == SYMACCEL_OK; ++c) {
        if (c + 1 < n_chunks) st = scatter(c + 1);
        if (st != SYMACCEL_OK) break;
        if ((st = gpu(hipStreamWaitEvent(ctx->stream, ev_in[c & 1], 0), "hipStreamWaitEvent(scatter)")) != SYMACCEL_OK) break;
        const Slice mc = chunk_of(mine.count, n_chunks, c);
        if (mc.count && step(user, mc.first, mc.count) != 0) {
            ctx->last_error = "symaccel_exchange_pipelined: the step callback failed";
            st = SYMACCEL_ERR_DEVICE;
            break;
        }
        if ((st = gpu(hipEventRecord(ev_k[c & 1], ctx->stream), "hipEventRecord(step)")) != SYMACCEL_OK) break;
        if ((st = gpu(hipStreamWaitEvent(xs, ev_k[c & 1], 0), "hipStreamWaitEvent(step)")) != SYMACCEL_OK) break;
        st = exchange_chunk(ctx, comm, world, rank, root, all_out, mine_out, n_streams, out_bytes_per_stream, n_chunks, c, true, xs);
    }
    // whoever synchronises the context's stream afterwards has the gathered result (also after an error: nothing stays in flight
    // behind the caller's back)
    (void)hipEventRecord(ev_done, xs);
    (void)hipStreamWaitEvent(ctx->stream, ev_done, 0);
    return st;
}

int symaccel_scatter_streams(symaccel_ctx *ctx, void *comm, int world, int rank, int root, const void *d_all, void *d_mine,
                             size_t n_streams, size_t bytes_per_stream) {
    return exchange(ctx, comm, world, rank, root, const_cast<void *>(d_all), d_mine, n_streams, bytes_per_stream, false);
}

int symaccel_gather_streams(symaccel_ctx *ctx, void *comm, int world, int rank, int root, const void *d_mine, void *d_all,
                            size_t n_streams, size_t bytes_per_stream) {
    return exchange(ctx, comm, world, rank, root, d_all, const_cast<void *>(d_mine), n_streams, bytes_per_stream, true);
}

}  // extern "C"
