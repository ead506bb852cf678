/*
 * ppp_queue.cpp -- a planner queue over the C ABI of include/ppp_hip.h: workpieces in, lists out, a few engine handles behind it
 * taking turns.  A pass is three dependent launches that leave most of the chip idle; passes on DIFFERENT handles (HIP streams)
 * overlap: 1 M-point workpieces 0.065 -> 0.039 ms each on three handles, never-seen clouds handed over without
 * a wait (ppp_set_cloud_device_async) included.  Host code only: everything it does a caller could do with the handles itself.
 */
#include "../../include/ppp_hip.h"

#include <new>
#include <string>
#include <vector>

struct ppp_queue_s {
    struct Lane {
        ppp_handle h = nullptr;
        long long ticket = -1; /* the job this lane holds (its pass enqueued, or its result not yet overwritten) */
        bool waited = false;   /* ... and whether ppp_sync has returned for it */
        int status = PPP_OK;
    };
    std::vector<Lane> lanes;
    long long next = 0;
    std::string err;
};

namespace {

int queue_create(int device, int lanes, const ppp_params *params, ppp_queue *out)
{
    if (!out || lanes < 0 || lanes > 16) return PPP_ERR_ARG;
    *out = nullptr;
    /* two: every workpiece of a queue is a new cloud -- a conversion pass, the host's planning, three launches enqueued one by one --, and a
       third lane then only adds contention (50 against 62 us per 1 M-point cloud; one lane: 88).  Replays of RESIDENT clouds, which bring none
       of that, do best on three handles (bench.py); four and more collide on the runtime's hardware queues (profiles/r04m_handles_taking_turns.txt) */
    if (lanes == 0) lanes = 2;
    ppp_queue q = new (std::nothrow) ppp_queue_s();
    if (!q) return PPP_ERR_HIP;
    q->lanes.resize((size_t)lanes);
    for (auto &l : q->lanes) {
        int rc = ppp_create(device, &l.h);
        if (rc == PPP_OK && params) rc = ppp_set_params(l.h, params);
        if (rc == PPP_OK) rc = ppp_set_side_by_side(l.h, lanes);
        if (rc != PPP_OK) {
            for (auto &m : q->lanes) if (m.h) ppp_destroy(m.h);
            delete q;
            return rc;
        }
    }
    *out = q;
    return PPP_OK;
}

void queue_destroy(ppp_queue q)
{
    if (!q) return;
    for (auto &l : q->lanes) if (l.h) ppp_destroy(l.h);
    delete q;
}

int lane_wait(ppp_queue q, ppp_queue_s::Lane &l)
{
    if (l.ticket >= 0 && !l.waited) {
        l.status = ppp_sync(l.h);
        if (l.status != PPP_OK) q->err = ppp_last_error(l.h);
        l.waited = true;
    }
    return l.status;
}

int queue_submit(ppp_queue q, const float *xyz_dev, size_t n, size_t stride_bytes, const float *viewpoint, long long *ticket)
{
    if (!q || !ticket) return PPP_ERR_ARG;
    *ticket = -1;
    ppp_queue_s::Lane &l = q->lanes[(size_t)(q->next % (long long)q->lanes.size())];
    (void)lane_wait(q, l); /* the lane's earlier job has finished; whatever it ended with was its caller's to read (ppp_queue_wait) */
    /* no wait for the cloud's bounds where the lane's plan is one of an earlier cloud of this size (ppp_set_plan_reuse), the pass
       right behind the conversion pass in the lane's stream */
    int rc = ppp_set_cloud_device_async(l.h, xyz_dev, n, stride_bytes, viewpoint);
    if (rc == PPP_OK) rc = ppp_run_async(l.h);
    if (rc != PPP_OK) { q->err = ppp_last_error(l.h); l.ticket = -1; return rc; }
    l.ticket = q->next; l.waited = false; l.status = PPP_OK;
    *ticket = q->next++;
    return PPP_OK;
}

int queue_wait(ppp_queue q, long long ticket, size_t *W, const float **list_dev)
{
    if (!q || ticket < 0) return PPP_ERR_ARG;
    if (W) *W = 0;
    if (list_dev) *list_dev = nullptr;
    ppp_queue_s::Lane &l = q->lanes[(size_t)(ticket % (long long)q->lanes.size())];
    if (l.ticket != ticket) { q->err = "the lane of this ticket holds a later workpiece already (its list is gone): wait before `lanes` more are submitted"; return PPP_ERR_ARG; }
    int rc = lane_wait(q, l);
    if (rc != PPP_OK) return rc;
    rc = ppp_get_waypoints_device(l.h, list_dev, W);
    if (rc != PPP_OK) q->err = ppp_last_error(l.h);
    return rc;
}

} // namespace

/* the boundary lets no C++ exception through */
extern "C" {
int ppp_queue_create(int device, int lanes, const ppp_params *params, ppp_queue *out)
{
    try { return queue_create(device, lanes, params, out); } catch (...) { if (out) *out = nullptr; return PPP_ERR_HIP; }
}
void ppp_queue_destroy(ppp_queue q) { try { queue_destroy(q); } catch (...) {} }
const char *ppp_queue_last_error(ppp_queue q) { return q ? q->err.c_str() : "no queue"; }
int ppp_queue_submit(ppp_queue q, const float *xyz_dev, size_t n, size_t stride_bytes, const float *viewpoint, long long *ticket)
{
    try { return queue_submit(q, xyz_dev, n, stride_bytes, viewpoint, ticket); } catch (...) { return PPP_ERR_HIP; }
}
int ppp_queue_wait(ppp_queue q, long long ticket, size_t *W, const float **list_dev)
{
    try { return queue_wait(q, ticket, W, list_dev); } catch (...) { return PPP_ERR_HIP; }
}
int ppp_queue_lanes(ppp_queue q) { return q ? (int)q->lanes.size() : 0; }
ppp_handle ppp_queue_lane(ppp_queue q, int i) { return (q && i >= 0 && i < (int)q->lanes.size()) ? q->lanes[(size_t)i].h : nullptr; }
} /* extern "C" */
