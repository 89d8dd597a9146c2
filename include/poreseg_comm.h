/* poreseg_comm.h -- the multi-GPU entry points of the segmenter's C ABI (ps_comm_init_all, ps_gather_bounds).
 *
 * What they replace: nothing in the reference -- PyPore walks files and events in one Python loop (DataTypes.py:956-988,
 * Experiment.parse; File.parse :589-602).  The MI355X form shards that loop over the GPUs of a node (events / files /
 * pieces of one trace are independent: no collective on the data path) and needs ONE exchange at the end: every rank's
 * segment boundaries to every rank.  That exchange is a fixed-shape all-gather over RCCL (xGMI): each rank contributes a slot
 * of `capacity` int32 elements -- element 0 its count, the boundaries from element PS_GATHER_HEADER on (the layout of
 * pypore_amd/dist.py's BoundaryGather, which sends ps_segment_batch's output buffer as it is).
 *
 * A separate small library (libporeseg_comm.so, links librccl): a host that shards with its own launcher and has no
 * torch.distributed -- the C / C++ host the ABI is meant for -- gets the gather from here; the Python host of this repository
 * uses torch.distributed by default (the same ncclAllGather underneath) and this library with
 * dist.BoundaryGather(..., backend="library").  All functions return 0 or a negative PS_COMM_ERR_* code;
 * ps_comm_last_error() has the text.  Not thread-safe per communicator (like a ps_ctx: one call at a time). */
#ifndef PORESEG_COMM_H
#define PORESEG_COMM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PS_COMM_ID_BYTES 128        /* sizeof(ncclUniqueId) */
#define PS_GATHER_HEADER 4          /* elements in front of the payload: [0] = count, [1..3] reserved (16 bytes: the payload keeps its alignment) */
#define PS_COMM_ERR_ARG (-1)
#define PS_COMM_ERR_RCCL (-2)
#define PS_COMM_ERR_HIP (-3)

typedef struct ps_comm ps_comm;

/* One process per GPU (torchrun, mpirun, the driver's launcher): rank 0 makes an id, hands it to the others out of band
 * (a file, MPI_Bcast, torch.distributed.broadcast_object_list), every rank joins on its own device. */
int ps_comm_unique_id(char id[PS_COMM_ID_BYTES]);
int ps_comm_init_rank(int world, int rank, const char id[PS_COMM_ID_BYTES], int device, ps_comm **out);
/* One process that drives `ndev` GPUs (devices == NULL: 0 .. ndev-1): out[0 .. ndev-1], rank i on devices[i]. */
int ps_comm_init_all(int ndev, const int *devices, ps_comm **out);
int ps_comm_world(const ps_comm *comm);
int ps_comm_rank(const ps_comm *comm);
void ps_comm_destroy(ps_comm *comm);

/* The boundary gather: d_recv[r * capacity .. (r + 1) * capacity) = rank r's d_send[0 .. capacity), on `stream` (a
 * hipStream_t; NULL = the device's null stream), asynchronously -- synchronise the stream before reading.  d_send and
 * d_recv are device pointers on the communicator's device; d_recv holds world * capacity elements.  A rank whose
 * boundaries do not fit writes its true count all the same: every rank sees it and falls back together (dist.py). */
int ps_gather_bounds(ps_comm *comm, const int32_t *d_send, int32_t *d_recv, int64_t capacity, void *stream);
/* The same for the communicators of ps_comm_init_all, as one group call (a single thread cannot issue them one by one:
 * each would wait for the others). */
int ps_gather_bounds_all(ps_comm *const *comms, int ndev, const int32_t *const *d_send, int32_t *const *d_recv,
                         int64_t capacity, void *const *streams);

const char *ps_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
