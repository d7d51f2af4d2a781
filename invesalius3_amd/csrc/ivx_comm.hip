// ivx_comm.hip -- the Z-slab communicator of the C ABI: RCCL over xGMI, no PyTorch.
//
// One process per GPU.  What the sharded path exchanges are boundary planes between Z-neighbours
// (ncclSend / ncclRecv inside one group, on the stream the kernels run on), a few bytes of votes / ranges
// (ncclAllReduce) and, for the projections, rows or images (ncclAllGather / ncclBroadcast).  librccl.so is opened with
// dlopen the first time a communicator is made, so single-GPU processes never load it.
//
// Rendezvous: rank 0 calls ivx_comm_unique_id and hands the 128 bytes to the other ranks out of band (bench.py /
// invesalius3_amd.comm use a file next to the launcher); every rank then calls ivx_comm_init on ITS device.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

#include "ivx_internal.h"

namespace {
using namespace ivx;

struct Rccl {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return IVX_OK;
    void *h = nullptr;
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    IVX_REQUIRE(h, IVX_EHIP, "ivx_comm: cannot load librccl.so (%s)", dlerror());
#define IVX_SYM(field, sym)                                                              \
    g_rccl.field = (decltype(g_rccl.field))dlsym(h, #sym);                                 \
    IVX_REQUIRE(g_rccl.field, IVX_EHIP, "ivx_comm: librccl.so lacks " #sym)
    IVX_SYM(GetUniqueId, ncclGetUniqueId);
    IVX_SYM(CommInitRank, ncclCommInitRank);
    IVX_SYM(CommDestroy, ncclCommDestroy);
    IVX_SYM(Send, ncclSend);
    IVX_SYM(Recv, ncclRecv);
    IVX_SYM(AllReduce, ncclAllReduce);
    IVX_SYM(AllGather, ncclAllGather);
    IVX_SYM(Broadcast, ncclBroadcast);
    IVX_SYM(GroupStart, ncclGroupStart);
    IVX_SYM(GroupEnd, ncclGroupEnd);
    IVX_SYM(GetErrorString, ncclGetErrorString);
#undef IVX_SYM
    g_rccl.h = h;
    return IVX_OK;
}

#define IVX_NCCL(expr)                                                                                      \
    do {                                                                                                    \
        ncclResult_t r__ = (expr);                                                                          \
        if (r__ != ncclSuccess) {                                                                           \
            ivx::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(r__));        \
            return IVX_EHIP;                                                                                \
        }                                                                                                   \
    } while (0)

// inside a GroupStart / GroupEnd pair: close the group before bailing out, or every later RCCL call of this thread
// would be swallowed by the group that was left open
#define IVX_NCCL_G(expr)                                                                                    \
    do {                                                                                                    \
        ncclResult_t r__ = (expr);                                                                          \
        if (r__ != ncclSuccess) {                                                                           \
            ivx::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(r__));        \
            g_rccl.GroupEnd();                                                                              \
            return IVX_EHIP;                                                                                \
        }                                                                                                   \
    } while (0)

struct Comm {
    ncclComm_t nc;
    int rank, world;
};

bool nccl_type(int dtype, ncclDataType_t *t) {
    switch (dtype) {
    case IVX_U8: *t = ncclUint8; return true;
    case IVX_I8: *t = ncclInt8; return true;
    case IVX_I32: *t = ncclInt32; return true;
    case IVX_I64: *t = ncclInt64; return true;
    case IVX_F32: *t = ncclFloat32; return true;
    case IVX_F64: *t = ncclFloat64; return true;
    default: return false; // RCCL reduces no 16-bit integers: widen first
    }
}
} // namespace

extern "C" int ivx_comm_unique_id(uint8_t id[128]) {
    const int rc = rccl_load();
    if (rc != IVX_OK) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    IVX_NCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, 128);
    return IVX_OK;
}

extern "C" int ivx_comm_init(const uint8_t id[128], int rank, int world, void **comm) {
    IVX_REQUIRE(comm && world >= 1 && rank >= 0 && rank < world, IVX_EINVAL, "ivx_comm_init: rank %d of %d", rank, world);
    const int rc = rccl_load();
    if (rc != IVX_OK) return rc;
    ncclUniqueId u;
    memcpy(&u, id, 128);
    Comm *c = new Comm{nullptr, rank, world};
    ncclResult_t r = g_rccl.CommInitRank(&c->nc, world, u, rank);
    if (r != ncclSuccess) {
        ivx::set_error("ncclCommInitRank(rank %d of %d) -> %s", rank, world, g_rccl.GetErrorString(r));
        delete c;
        return IVX_EHIP;
    }
    *comm = c;
    return IVX_OK;
}

extern "C" int ivx_comm_destroy(void *comm) {
    if (!comm) return IVX_OK;
    Comm *c = (Comm *)comm;
    if (c->nc) IVX_NCCL(g_rccl.CommDestroy(c->nc));
    delete c;
    return IVX_OK;
}

extern "C" int ivx_comm_rank(const void *comm, int *rank, int *world) {
    IVX_REQUIRE(comm, IVX_EINVAL, "ivx_comm: null communicator");
    *rank = ((const Comm *)comm)->rank;
    *world = ((const Comm *)comm)->world;
    return IVX_OK;
}

// Halo exchange with the two Z-neighbours in one group: to_down -> rank-1, to_up -> rank+1, and the mirror receives.
// Pointers that do not apply (rank 0 has no lower neighbour, ...) are ignored; any may be NULL to skip that leg, but
// both sides of a link must agree (a send to rank+1 needs rank+1's receive from down).
extern "C" int ivx_comm_exchange(void *comm, const void *to_down, void *from_down, const void *to_up, void *from_up,
                                 size_t nbytes, void *stream) {
    IVX_REQUIRE(comm, IVX_EINVAL, "ivx_comm: null communicator");
    Comm *c = (Comm *)comm;
    if (c->world == 1 || nbytes == 0) return IVX_OK;
    hipStream_t st = S(stream);
    IVX_NCCL(g_rccl.GroupStart());
    if (c->rank > 0) {
        if (to_down) IVX_NCCL_G(g_rccl.Send(to_down, nbytes, ncclUint8, c->rank - 1, c->nc, st));
        if (from_down) IVX_NCCL_G(g_rccl.Recv(from_down, nbytes, ncclUint8, c->rank - 1, c->nc, st));
    }
    if (c->rank < c->world - 1) {
        if (to_up) IVX_NCCL_G(g_rccl.Send(to_up, nbytes, ncclUint8, c->rank + 1, c->nc, st));
        if (from_up) IVX_NCCL_G(g_rccl.Recv(from_up, nbytes, ncclUint8, c->rank + 1, c->nc, st));
    }
    IVX_NCCL(g_rccl.GroupEnd());
    return IVX_OK;
}

// the same exchange followed, on the same stream, by an in-place all-reduce (sum) of `nvote` int32 words: the
// region-growing round's planes and its "did anybody gain anything" vote are enqueued by ONE call, nothing waits in between
extern "C" int ivx_comm_exchange_vote(void *comm, const void *to_down, void *from_down, const void *to_up, void *from_up,
                                      size_t nbytes, int32_t *vote, int nvote, void *stream) {
    IVX_REQUIRE(comm, IVX_EINVAL, "ivx_comm: null communicator");
    Comm *c = (Comm *)comm;
    if (c->world == 1) return IVX_OK;
    hipStream_t st = S(stream);
    IVX_NCCL(g_rccl.GroupStart());
    if (nbytes) {
        if (c->rank > 0) {
            if (to_down) IVX_NCCL_G(g_rccl.Send(to_down, nbytes, ncclUint8, c->rank - 1, c->nc, st));
            if (from_down) IVX_NCCL_G(g_rccl.Recv(from_down, nbytes, ncclUint8, c->rank - 1, c->nc, st));
        }
        if (c->rank < c->world - 1) {
            if (to_up) IVX_NCCL_G(g_rccl.Send(to_up, nbytes, ncclUint8, c->rank + 1, c->nc, st));
            if (from_up) IVX_NCCL_G(g_rccl.Recv(from_up, nbytes, ncclUint8, c->rank + 1, c->nc, st));
        }
    }
    IVX_NCCL(g_rccl.GroupEnd());
    // the vote follows on the same stream, outside the point-to-point group: fusing a collective into a send / recv group is
    // not something every RCCL release accepts, and this path cannot be exercised on a one-GPU box
    if (vote && nvote > 0) IVX_NCCL(g_rccl.AllReduce(vote, vote, (size_t)nvote, ncclInt32, ncclSum, c->nc, st));
    return IVX_OK;
}

// in place; op 0 sum, 1 max, 2 min; dtype IVX_I32 / IVX_I64 / IVX_F32 / IVX_F64 / IVX_U8 / IVX_I8
extern "C" int ivx_comm_allreduce(void *comm, void *buf, size_t count, int dtype, int op, void *stream) {
    IVX_REQUIRE(comm, IVX_EINVAL, "ivx_comm: null communicator");
    Comm *c = (Comm *)comm;
    ncclDataType_t t;
    IVX_REQUIRE(nccl_type(dtype, &t), IVX_EINVAL, "ivx_comm_allreduce: dtype %d (RCCL reduces no 16-bit integers)", dtype);
    IVX_REQUIRE(op >= 0 && op <= 2, IVX_EINVAL, "ivx_comm_allreduce: op %d", op);
    if (c->world == 1 || count == 0) return IVX_OK;
    const ncclRedOp_t ops[3] = {ncclSum, ncclMax, ncclMin};
    IVX_NCCL(g_rccl.AllReduce(buf, buf, count, t, ops[op], c->nc, S(stream)));
    return IVX_OK;
}

// recv holds world * nbytes; rank r's block lands at recv + r * nbytes
extern "C" int ivx_comm_allgather(void *comm, const void *send, void *recv, size_t nbytes, void *stream) {
    IVX_REQUIRE(comm, IVX_EINVAL, "ivx_comm: null communicator");
    Comm *c = (Comm *)comm;
    if (nbytes == 0) return IVX_OK;
    if (c->world == 1) {
        if (send != recv) IVX_HIP(hipMemcpyAsync(recv, send, nbytes, hipMemcpyDeviceToDevice, S(stream)));
        return IVX_OK;
    }
    IVX_NCCL(g_rccl.AllGather(send, recv, nbytes, ncclUint8, c->nc, S(stream)));
    return IVX_OK;
}

extern "C" int ivx_comm_bcast(void *comm, void *buf, size_t nbytes, int root, void *stream) {
    IVX_REQUIRE(comm, IVX_EINVAL, "ivx_comm: null communicator");
    Comm *c = (Comm *)comm;
    IVX_REQUIRE(root >= 0 && root < c->world, IVX_EINVAL, "ivx_comm_bcast: root %d", root);
    if (c->world == 1 || nbytes == 0) return IVX_OK;
    IVX_NCCL(g_rccl.Broadcast(buf, buf, nbytes, ncclUint8, root, c->nc, S(stream)));
    return IVX_OK;
}

extern "C" int ivx_comm_send(void *comm, const void *buf, size_t nbytes, int peer, void *stream) {
    IVX_REQUIRE(comm, IVX_EINVAL, "ivx_comm: null communicator");
    Comm *c = (Comm *)comm;
    IVX_REQUIRE(peer >= 0 && peer < c->world && peer != c->rank, IVX_EINVAL, "ivx_comm_send: peer %d", peer);
    IVX_NCCL(g_rccl.Send(buf, nbytes, ncclUint8, peer, c->nc, S(stream)));
    return IVX_OK;
}

extern "C" int ivx_comm_recv(void *comm, void *buf, size_t nbytes, int peer, void *stream) {
    IVX_REQUIRE(comm, IVX_EINVAL, "ivx_comm: null communicator");
    Comm *c = (Comm *)comm;
    IVX_REQUIRE(peer >= 0 && peer < c->world && peer != c->rank, IVX_EINVAL, "ivx_comm_recv: peer %d", peer);
    IVX_NCCL(g_rccl.Recv(buf, nbytes, ncclUint8, peer, c->nc, S(stream)));
    return IVX_OK;
}

// ---- self-test: every entry point once, on small buffers, against the analytic answer -----------------------------------
// A multi-GPU job calls this right after ivx_comm_init (bench.py does, tests/test_gpu_slab.py does at world 1): a broken
// link, a mismatched RCCL or a wrong rank order then fails in the first second with the NAME of the collective instead of
// hanging in the first region-growing round.  At world 1 the data-path entry points return before touching RCCL (there
// is no neighbour), so the test drives RCCL directly there: all-reduce / all-gather / broadcast on the one-rank
// communicator and a send + receive to itself inside one group -- the same symbols, types and group discipline.
namespace {
__global__ void k_comm_fill(int32_t *p, int n, int base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = base + i;
}
} // namespace

extern "C" int ivx_comm_selftest(void *comm, void *stream) {
    IVX_REQUIRE(comm, IVX_EINVAL, "ivx_comm: null communicator");
    Comm *c = (Comm *)comm;
    hipStream_t st = S(stream);
    const int R = c->rank, W = c->world;
    constexpr int N = 1024; // int32 words per leg
    int32_t *d = nullptr;
    const size_t words = (size_t)N * (6 + (size_t)W);
    IVX_HIP(hipMalloc(&d, words * 4));
    struct Free {
        void *p;
        ~Free() { (void)hipFree(p); }
    } guard{d};
    std::vector<int32_t> h(words);
    int32_t *to_dn = d, *to_up = d + N, *fr_dn = d + 2 * N, *fr_up = d + 3 * N, *red = d + 4 * N, *one = d + 5 * N, *gat = d + 6 * N;
    auto fetch = [&]() -> int {
        IVX_HIP(hipStreamSynchronize(st));
        IVX_HIP(hipMemcpy(h.data(), d, words * 4, hipMemcpyDeviceToHost));
        return IVX_OK;
    };
    int rc;
    IVX_HIP(hipMemsetAsync(d, 0xff, words * 4, st));
    hipLaunchKernelGGL(k_comm_fill, dim3(N / 256), dim3(256), 0, st, to_dn, N, 1000 * R + 100000);
    hipLaunchKernelGGL(k_comm_fill, dim3(N / 256), dim3(256), 0, st, to_up, N, 1000 * R + 200000);
    hipLaunchKernelGGL(k_comm_fill, dim3(N / 256), dim3(256), 0, st, red, N, R);
    hipLaunchKernelGGL(k_comm_fill, dim3(N / 256), dim3(256), 0, st, one, N, 7 * R);
    IVX_LAUNCH_CHECK();
    // 1. halo exchange + vote
    if ((rc = ivx_comm_exchange(comm, to_dn, fr_dn, to_up, fr_up, (size_t)N * 4, stream))) return rc;
    if ((rc = fetch())) return rc;
    for (int i = 0; i < N; i++) {
        if (R > 0) IVX_REQUIRE(h[2 * N + i] == 1000 * (R - 1) + 200000 + i, IVX_EHIP, "ivx_comm_exchange: rank %d got %d from below at %d", R, h[2 * N + i], i);
        if (R < W - 1) IVX_REQUIRE(h[3 * N + i] == 1000 * (R + 1) + 100000 + i, IVX_EHIP, "ivx_comm_exchange: rank %d got %d from above at %d", R, h[3 * N + i], i);
    }
    IVX_HIP(hipMemsetAsync(fr_dn, 0xff, (size_t)2 * N * 4, st));
    if ((rc = ivx_comm_exchange_vote(comm, to_dn, fr_dn, to_up, fr_up, (size_t)N * 4, red, 8, stream))) return rc;
    if ((rc = fetch())) return rc;
    for (int i = 0; i < 8; i++) {
        const int want = W == 1 ? R + i : W * (W - 1) / 2 + W * i;
        IVX_REQUIRE(h[4 * N + i] == want, IVX_EHIP, "ivx_comm_exchange_vote: vote word %d is %d, expected %d", i, h[4 * N + i], want);
    }
    for (int i = 0; i < N; i++) {
        if (R > 0) IVX_REQUIRE(h[2 * N + i] == 1000 * (R - 1) + 200000 + i, IVX_EHIP, "ivx_comm_exchange_vote: plane from below differs at %d", i);
        if (R < W - 1) IVX_REQUIRE(h[3 * N + i] == 1000 * (R + 1) + 100000 + i, IVX_EHIP, "ivx_comm_exchange_vote: plane from above differs at %d", i);
    }
    // 2. all-reduce (sum, max, min), all-gather, broadcast through the public entry points
    for (int op = 0; op < 3; op++) {
        hipLaunchKernelGGL(k_comm_fill, dim3(N / 256), dim3(256), 0, st, red, N, R);
        IVX_LAUNCH_CHECK();
        if ((rc = ivx_comm_allreduce(comm, red, N, IVX_I32, op, stream))) return rc;
        if ((rc = fetch())) return rc;
        for (int i = 0; i < N; i++) {
            const int want = op == 0 ? W * (W - 1) / 2 + W * i : (op == 1 ? W - 1 + i : i);
            IVX_REQUIRE(h[4 * N + i] == want, IVX_EHIP, "ivx_comm_allreduce(op %d): word %d is %d, expected %d", op, i, h[4 * N + i], want);
        }
    }
    if ((rc = ivx_comm_allgather(comm, one, gat, (size_t)N * 4, stream))) return rc;
    if ((rc = fetch())) return rc;
    for (int r = 0; r < W; r++)
        for (int i = 0; i < N; i++)
            IVX_REQUIRE(h[(size_t)(6 + r) * N + i] == 7 * r + i, IVX_EHIP, "ivx_comm_allgather: block %d word %d is %d", r, i, h[(size_t)(6 + r) * N + i]);
    const int root = W - 1;
    if ((rc = ivx_comm_bcast(comm, one, (size_t)N * 4, root, stream))) return rc;
    if ((rc = fetch())) return rc;
    for (int i = 0; i < N; i++) IVX_REQUIRE(h[5 * N + i] == 7 * root + i, IVX_EHIP, "ivx_comm_bcast: word %d is %d", i, h[5 * N + i]);
    // 3. the point-to-point chain of the front-to-back ray hand-over: rank r receives from r-1, then sends to r+1
    hipLaunchKernelGGL(k_comm_fill, dim3(N / 256), dim3(256), 0, st, to_up, N, 31 * R);
    IVX_LAUNCH_CHECK();
    if (R > 0 && (rc = ivx_comm_recv(comm, fr_dn, (size_t)N * 4, R - 1, stream))) return rc;
    if (R < W - 1 && (rc = ivx_comm_send(comm, to_up, (size_t)N * 4, R + 1, stream))) return rc;
    if ((rc = fetch())) return rc;
    if (R > 0)
        for (int i = 0; i < N; i++) IVX_REQUIRE(h[2 * N + i] == 31 * (R - 1) + i, IVX_EHIP, "ivx_comm_send/recv: word %d is %d", i, h[2 * N + i]);
    if (W > 1) return IVX_OK;
    // 4. world 1: nothing above went through RCCL -- drive it directly on the one-rank communicator
    hipLaunchKernelGGL(k_comm_fill, dim3(N / 256), dim3(256), 0, st, red, N, 5);
    hipLaunchKernelGGL(k_comm_fill, dim3(N / 256), dim3(256), 0, st, one, N, 9);
    IVX_LAUNCH_CHECK();
    IVX_HIP(hipMemsetAsync(fr_dn, 0xff, (size_t)N * 4, st));
    IVX_NCCL(g_rccl.AllReduce(red, red, (size_t)N, ncclInt32, ncclSum, c->nc, st));
    IVX_NCCL(g_rccl.AllGather(one, gat, (size_t)N * 4, ncclUint8, c->nc, st));
    IVX_NCCL(g_rccl.Broadcast(one, one, (size_t)N * 4, ncclUint8, 0, c->nc, st));
    IVX_NCCL(g_rccl.GroupStart());
    IVX_NCCL_G(g_rccl.Send(to_up, (size_t)N * 4, ncclUint8, 0, c->nc, st));
    IVX_NCCL_G(g_rccl.Recv(fr_dn, (size_t)N * 4, ncclUint8, 0, c->nc, st));
    IVX_NCCL(g_rccl.GroupEnd());
    if ((rc = fetch())) return rc;
    for (int i = 0; i < N; i++) {
        IVX_REQUIRE(h[4 * N + i] == 5 + i, IVX_EHIP, "RCCL all-reduce on one rank: word %d is %d", i, h[4 * N + i]);
        IVX_REQUIRE(h[6 * N + i] == 9 + i && h[5 * N + i] == 9 + i, IVX_EHIP, "RCCL all-gather / broadcast on one rank: word %d", i);
        IVX_REQUIRE(h[2 * N + i] == 31 * R + i, IVX_EHIP, "RCCL send + recv to self in one group: word %d is %d", i, h[2 * N + i]);
    }
    return IVX_OK;
}
