/*
 * ppp_gather.h -- the one exchange step of the multi-GPU path: the variable-length gather of the
 * ranks' finished W_r x 6 blocks to the root, as ONE RCCL group of direct send / recv pairs (xGMI is point to point:
 * every non-root rank writes to the root over its own link; no ring, no reduction).
 *
 * Host-only and free of HIP / RCCL types so that the call pattern -- offsets, counts, root != 0, ranks with no rows,
 * and the rule that ncclGroupEnd is reached whatever happens after ncclGroupStart -- is testable without a GPU
 * (tests/test_host_logic.py::test_gather_exchange_call_pattern drives it with recording stand-ins).
 */
#pragma once
#include <stddef.h>

struct PppGatherOps {
    int (*group_start)();
    int (*group_end)();
    int (*send)(const void *buf, size_t count, int dtype, int peer, void *comm, void *stream);
    int (*recv)(void *buf, size_t count, int dtype, int peer, void *comm, void *stream);
    /* the root's own block: a device-to-device copy on `stream`; returns 0 on success */
    int (*local_copy)(void *dst, const void *src, size_t bytes, void *stream);
};

enum { PPP_GATHER_OK = 0, PPP_GATHER_COPY_FAILED = 1, PPP_GATHER_GROUP_START_FAILED = 2, PPP_GATHER_XFER_FAILED = 3 };

/* counts_rows[r] = rows (6 floats each) of rank r; recv is the root's buffer of sum(counts_rows) rows, blocks in rank order.
   *nccl_result receives the first non-zero ncclResult.  Nothing is enqueued for a rank without rows. */
static inline int ppp_gather_exchange(const PppGatherOps &ops, int rank, int nranks, int root, const size_t *counts_rows,
                                      const float *send, float *recv, void *comm, void *stream, int *nccl_result)
{
    const int kFloat = 7; /* ncclFloat32 */
    if (nccl_result) *nccl_result = 0;
    /* the root's own block is a plain copy: outside the group, so that nothing between ncclGroupStart and ncclGroupEnd
       can return early (an open group hangs the first collective that follows) */
    if (rank == root && counts_rows[root]) {
        size_t off = 0;
        for (int r = 0; r < root; ++r) off += counts_rows[r];
        if (ops.local_copy(recv + 6 * off, send, counts_rows[root] * 24, stream) != 0) return PPP_GATHER_COPY_FAILED;
    }
    int e = ops.group_start();
    if (e) { if (nccl_result) *nccl_result = e; return PPP_GATHER_GROUP_START_FAILED; }
    if (rank == root) {
        size_t off = 0;
        for (int r = 0; r < nranks; ++r) {
            if (r != root && counts_rows[r] && !e) e = ops.recv(recv + 6 * off, counts_rows[r] * 6, kFloat, r, comm, stream);
            off += counts_rows[r];
        }
    } else if (counts_rows[rank]) e = ops.send(send, counts_rows[rank] * 6, kFloat, root, comm, stream);
    const int e2 = ops.group_end(); /* always reached */
    if (e || e2) { if (nccl_result) *nccl_result = e ? e : e2; return PPP_GATHER_XFER_FAILED; }
    return PPP_GATHER_OK;
}

/* Pre-flight of the group path on ONE rank (no second GPU needed): the rank sends its block to itself and receives it, both
   inside one ncclGroupStart / ncclGroupEnd -- the same calls, datatype, communicator and stream a multi-rank gather hands to
   librccl, with real transfers behind them (ppp_gather_waypoints with nranks == 1 and a one-rank communicator;
   tests/test_gpu_parity.py::test_gather_rehearsal_through_the_real_rccl). */
static inline int ppp_gather_self_loop(const PppGatherOps &ops, size_t rows, const float *send, float *recv, void *comm, void *stream,
                                       int *nccl_result)
{
    const int kFloat = 7; /* ncclFloat32 */
    if (nccl_result) *nccl_result = 0;
    if (!rows) return PPP_GATHER_OK;
    int e = ops.group_start();
    if (e) { if (nccl_result) *nccl_result = e; return PPP_GATHER_GROUP_START_FAILED; }
    e = ops.send(send, rows * 6, kFloat, 0, comm, stream);
    if (!e) e = ops.recv(recv, rows * 6, kFloat, 0, comm, stream);
    const int e2 = ops.group_end(); /* always reached */
    if (e || e2) { if (nccl_result) *nccl_result = e ? e : e2; return PPP_GATHER_XFER_FAILED; }
    return PPP_GATHER_OK;
}
