// libporeseg_comm.so: the multi-GPU entry points of include/poreseg_comm.h -- thin, on purpose: the data path of the
// segmenter has no collective (events, files and trace pieces are independent), the one exchange is the boundary gather, and
// that is ncclAllGather over xGMI.  Built by pypore_amd/csrc/Makefile (target comm), loaded by pypore_amd/_comm.py.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "poreseg_comm.h"

static_assert(sizeof(ncclUniqueId) == PS_COMM_ID_BYTES, "PS_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

struct ps_comm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0, device = 0;
};

namespace {
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define COMM_TRY_NCCL(expr)                                                                                    \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return fail(PS_COMM_ERR_RCCL, "%s: %s", #expr, ncclGetErrorString(r_));         \
    } while (0)
#define COMM_TRY_HIP(expr)                                                                                     \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return fail(PS_COMM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));            \
    } while (0)
}  // namespace

extern "C" {

const char *ps_comm_last_error(void) { return g_err; }

int ps_comm_unique_id(char id[PS_COMM_ID_BYTES])
{
    if (!id) return fail(PS_COMM_ERR_ARG, "id is NULL");
    ncclUniqueId u;
    COMM_TRY_NCCL(ncclGetUniqueId(&u));
    std::memcpy(id, &u, PS_COMM_ID_BYTES);
    return 0;
}

int ps_comm_init_rank(int world, int rank, const char id[PS_COMM_ID_BYTES], int device, ps_comm **out)
{
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return fail(PS_COMM_ERR_ARG, "world %d rank %d", world, rank);
    *out = nullptr;
    ncclUniqueId u;
    std::memcpy(&u, id, PS_COMM_ID_BYTES);
    COMM_TRY_HIP(hipSetDevice(device));
    ps_comm *c = new (std::nothrow) ps_comm;
    if (!c) return fail(PS_COMM_ERR_ARG, "out of memory");
    c->world = world; c->rank = rank; c->device = device;
    ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { delete c; return fail(PS_COMM_ERR_RCCL, "ncclCommInitRank: %s", ncclGetErrorString(r)); }
    *out = c;
    return 0;
}

int ps_comm_init_all(int ndev, const int *devices, ps_comm **out)
{
    if (!out || ndev < 1) return fail(PS_COMM_ERR_ARG, "ndev %d", ndev);
    std::vector<int> devs(static_cast<size_t>(ndev));
    for (int i = 0; i < ndev; ++i) devs[i] = devices ? devices[i] : i;
    std::vector<ncclComm_t> comms(static_cast<size_t>(ndev), nullptr);
    for (int i = 0; i < ndev; ++i) out[i] = nullptr;
    COMM_TRY_NCCL(ncclCommInitAll(comms.data(), ndev, devs.data()));
    for (int i = 0; i < ndev; ++i) {
        ps_comm *c = new (std::nothrow) ps_comm;
        if (!c) {
            // nothing half-made is left behind: the ranks wrapped so far and every communicator go back, out[] is all NULL
            for (int k = 0; k < i; ++k) { out[k]->comm = nullptr; delete out[k]; out[k] = nullptr; }
            for (int k = 0; k < ndev; ++k) { (void)hipSetDevice(devs[k]); (void)ncclCommDestroy(comms[k]); }
            return fail(PS_COMM_ERR_ARG, "out of memory");
        }
        c->comm = comms[i]; c->world = ndev; c->rank = i; c->device = devs[i];
        out[i] = c;
    }
    return 0;
}

int ps_comm_world(const ps_comm *comm) { return comm ? comm->world : 0; }
int ps_comm_rank(const ps_comm *comm) { return comm ? comm->rank : -1; }

void ps_comm_destroy(ps_comm *comm)
{
    if (!comm) return;
    if (comm->comm) { (void)hipSetDevice(comm->device); (void)ncclCommDestroy(comm->comm); }
    delete comm;
}

int ps_gather_bounds(ps_comm *comm, const int32_t *d_send, int32_t *d_recv, int64_t capacity, void *stream)
{
    if (!comm || !d_send || !d_recv || capacity < PS_GATHER_HEADER) return fail(PS_COMM_ERR_ARG, "capacity %lld", static_cast<long long>(capacity));
    COMM_TRY_HIP(hipSetDevice(comm->device));
    COMM_TRY_NCCL(ncclAllGather(d_send, d_recv, static_cast<size_t>(capacity), ncclInt32, comm->comm, static_cast<hipStream_t>(stream)));
    return 0;
}

int ps_gather_bounds_all(ps_comm *const *comms, int ndev, const int32_t *const *d_send, int32_t *const *d_recv,
                         int64_t capacity, void *const *streams)
{
    if (!comms || !d_send || !d_recv || ndev < 1 || capacity < PS_GATHER_HEADER) return fail(PS_COMM_ERR_ARG, "ndev %d capacity %lld", ndev, static_cast<long long>(capacity));
    for (int i = 0; i < ndev; ++i)
        if (!comms[i] || !d_send[i] || !d_recv[i]) return fail(PS_COMM_ERR_ARG, "rank %d: NULL argument", i);
    COMM_TRY_NCCL(ncclGroupStart());
    for (int i = 0; i < ndev; ++i) {
        ncclResult_t r = ncclAllGather(d_send[i], d_recv[i], static_cast<size_t>(capacity), ncclInt32, comms[i]->comm,
                                       static_cast<hipStream_t>(streams ? streams[i] : nullptr));
        if (r != ncclSuccess) { (void)ncclGroupEnd(); return fail(PS_COMM_ERR_RCCL, "ncclAllGather (rank %d): %s", i, ncclGetErrorString(r)); }
    }
    COMM_TRY_NCCL(ncclGroupEnd());
    return 0;
}

}  // extern "C"
