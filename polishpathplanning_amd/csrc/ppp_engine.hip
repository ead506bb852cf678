/*
 * ppp_engine.hip -- host side of the C ABI declared in include/ppp_hip.h.
 *
 * One handle = one device, one HIP stream, one resident cloud.  All device memory is
 * allocated when the cloud / parameters are set (the "plan"); the two hot calls
 * ppp_gen_path_async / ppp_get_path_async only enqueue kernels -- no allocation, no host
 * synchronisation -- so a caller may overlap handles, capture them, or time them.
 * There is no CPU fallback: without a HIP device every entry point fails loudly.
 */
#include "ppp_kernels.h"
#ifdef PPP_SINGLE_TU /* diagnostic builds (in-kernel stamps share one g_stamps array): the window kernels in this translation unit */
#include "ppp_window.h"
#else
#include "ppp_window_decl.h" /* their kernels are ppp_window.hip's */
#endif
#include "ppp_preproc.h"
#include "ppp_sort.h"
#include "ppp_align.h"
#include "ppp_gather.h"
#include <atomic>
#include <cerrno>
#include <fcntl.h>
#include <thread>
#include <unistd.h>
/* LDS slots of a slab's sort workgroup, as a multiple of the mean slab population (rounded up to a power of two): only
   clouds beyond the 8192-slab cap see it (mean > 1024), where 1.6 keeps the workgroup at 24 KiB of LDS -- twice as many
   slabs in flight, cfg 5's sort 125 -> 110 us -- and a slab denser than that goes through the arena pass */
#ifndef PPP_NN_HINT_SPACINGS
#define PPP_NN_HINT_SPACINGS 3.0 /* first bound of a waypoint's nearest-neighbour search, in mean point spacings (a hint: nothing found -> unbounded repeat) */
#endif
#ifndef PPP_SLAB_PTS
#define PPP_SLAB_PTS 832 /* mean points per x-slab (measured, see make_plan) */
#endif
#ifndef PPP_PPT8_FROM
#define PPP_PPT8_FROM 500000 /* points from which a scatter workgroup takes 8 points per thread instead of 4: half the per-(workgroup, slab) reservations (1 M points: scatter 19.2 -> 16.2 us; 250 k points are better off with 4) */
#endif
#ifndef WIN_SLICE_WAVES_PER_CU
#define WIN_SLICE_WAVES_PER_CU 16 /* waves of the slice kernel a CU holds (win_pick_threads shares them between the workgroups the LDS admits) */
#endif
#ifndef PPP_WIN_PPT8_FROM
#define PPP_WIN_PPT8_FROM 1200000 /* the window path's binning launch: what it costs is its workgroups' reservations (one global atomic per workgroup and
                                     non-empty window) -- 1 M points / 256 windows: 122 workgroups of 8 points per thread 20.1 us, 244 of 4 18.8 us, 488 22.7-23.8, 977 33 us */
#endif
#ifndef PPP_PPT16_FROM
#define PPP_PPT16_FROM 1500000 /* ... and 16: a workgroup's run in a slab grows to ~7 points = most of a 128-byte line (2 M points: scatter 36.4 -> 32.4 us) */
#endif
#ifndef PPP_BATCH_SPLIT_FROM
#define PPP_BATCH_SPLIT_FROM 8 /* members from which a batch is launched as two halves side by side (8 x 1 M points: 0.549 -> 0.533 ms; 16: 1.07 -> 1.04; 64 x 250 k: 0.877 -> 0.844) */
#endif
#ifndef PPP_MM_GRID_MAX
#define PPP_MM_GRID_MAX 192 /* workgroups of the bounds + histogram pass: every one flushes its LDS histogram with an atomic per non-empty slab, which is what grows with the grid (1 M points: 192 is 1.5 us ahead of 256; 128 .. 160 the same) */
#endif
#ifndef PPP_SLAB_CAP_FACTOR
#define PPP_SLAB_CAP_FACTOR 1.6
#endif
#include "../../include/ppp_hip.h"

#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#define PPP_VERSION_STR "polishpathplanning_amd 0.1 (gfx950)"

namespace {

struct KTimer {
    std::string name;
    std::vector<hipEvent_t> e0, e1; /* one pair per launch of this kernel in a pass */
    int used = 0;
};

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n)
    {
        if (n <= cap && p) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        if (n == 0) n = 1;
        hipError_t e = hipMalloc((void **)&p, n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

} // namespace

struct ppp_handle_s {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    ppp_params P;
    float vp[3] = {0, 0, 0};
    size_t n = 0;
    bool have_cloud = false, planned = false, index_built = false, gen_done = false, path_done = false;
    /* host copy of the knots of the last pass (slice tables + node arrays), fetched whole by the first ppp_get_nodes after a pass:
       the planner classes ask slice by slice (a Spline view per slice: 2 calls x 256 slices), and a synchronous copy of a few
       bytes costs ~20 us on this runtime -- 60 ms of GenPath() for 0.07 ms of planning before this cache */
    unsigned long long gen_serial = 0, hn_serial = ~0ull;
    std::vector<int> hn_off;    /* S + 1 offsets into ... */
    std::vector<float> hn_xyz;  /* ... three planes (x | y | z) of hn_off[S] floats */
    DevBuf<int> pack_tab;       /* device: node_start as the host validated it, then the offsets */
    DevBuf<float> pack_out;
    int max_lds = 65536;
    int num_cus = 256;

    /* plan */
    int B = 1, slab_cap = 4096, S_cap = 1, capb = 2048, W_cap = 1, node_cap = 1;
    int knot_cap = 2048, stage_cap = POSE_STAGE_CAP, tab_slabs = 8, pose_threads = POSE_T, cnt_est = 1; /* launch geometry of k_pose (make_plan) */
    float pose_pad = 8.f;
    float h_mn[3] = {0, 0, 0}, h_mx[3] = {0, 0, 0};
    int h_nvalid = 0;
    /* slice-range handles (SURVEY.md 8e case ii) */
    bool ranged = false;          /* plans a strict sub-range of the slices: getPath stops after a12 */
    /* trans2center ran (Alignment = true): TransAlign, its inverse, and a second handle holding the cloud carried back by
       the inverse with its own slab index (path_translation_alg.cpp:171-174 searches and estimates normals there) */
    bool aligned = false;
    float TA[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}, invTA[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    ppp_handle back = nullptr;
    int sb = 0, se = 0;           /* the range, resolved against the walk */
    float incl_lo = -INFINITY, incl_hi = INFINITY;
    int n_range = 0;              /* expected number of indexed points */
    bool list_final = false;      /* wp_out holds a finished WayPointsList */

    DevBuf<float> X, Y, Z;
    /* slice-range handles: the points of [incl_lo, incl_hi] in cloud order with their cloud indices (built by make_plan):
       the hot path streams these instead of the whole cloud */
    DevBuf<float> Xp, Yp, Zp;
    DevBuf<int> part_idx;
    int n_part = 0;
    bool use_part = false;
    /* ppp_set_cloud_part: the resident cloud IS a part (every point with x in [part_lo, part_hi], cloud order); the whole
       cloud's bounds and point count came with it, part_idx (optional) holds the points' cloud indices */
    bool part_given = false;
    bool part_has_idx = false;
    float part_lo = 0.f, part_hi = 0.f;
    DevBuf<float4> unsorted4, sorted4;
    DevBuf<int> slab_cnt, slab_start, slab_cursor, coarse_cursor, slab_ytab;
    bool two_pass_scatter = false; /* large clouds: coarse bins first (see k_slab_scatter) */
    DevBuf<float> slab_xmin, slab_xmax;
    DevBuf<DevMeta> meta;
    DevBuf<float> px, lo, hi;
    DevBuf<float> node_x, node_y, node_z;
    /* dynamic adjustment (allocated when Dynamic_adjustment is on or ppp_area2cloud is used) */
    DevBuf<float4> normals4, dyn_bnd_pts, dyn_adj_pts, dyn_first_ab, dyn_first_snap;
    DevBuf<double> dyn_first_node;
    DevBuf<float> ell_cs;
    DevBuf<double> dyn_bnd_knots;
    DevBuf<int> dyn_bnd_n;
    int dyn_maxNB = 1, dyn_maxNA = 1;
    bool dyn_keep_all = false;
    bool normals_valid = false;
    DevBuf<int> node_start, node_cnt, band_cnt;
    DevBuf<int> wp_cnt, wp_off, tail, slice_wpcnt;
    DevBuf<float4> wp_xyz, wp_normal;
    DevBuf<int> wp_nn;
    DevBuf<float> wp_pre, wp_smooth, wp_out;
    DevBuf<MinMaxPart> mm_part;
    DevBuf<int> big_slabs, big_slices; /* work lists of the LDS-overflow fallback kernels */
    DevBuf<char> arena;                /* their global scratch, allocated on first need */
    bool big_path = false;             /* launch the fallback kernels (set by the plan or after an overflow) */
    int mm_grid = 1, mm_grid_used = 1, sm_tiles = 1;
    DevBuf<char> scratch; /* API staging */
    /* window path (ppp_window.h): three launches, every point binned once into the window of its slice */
    bool win_allowed = true;    /* ppp_set_fast_path */
    bool win_disabled = false;  /* a pass was handed back (overflow / reach / stale plan): this cloud + parameters stay on the slab path */
    bool win_path = false;      /* the current plan runs the window path */
    bool win_staged = false;    /* the binning launch writes through LDS in window order (large clouds, ppp_window.h) */
    bool stage_compact = true;  /* wp_xyz / wp_nn / wp_normal hold the list order (a window pass leaves them in per-slice slots) */
    float win_pad = 4.f;
    int win_NBc_thr = 0; /* y-buckets per class in launches of several workgroups per CU: the most that cost no workgroup its place in the LDS */
    int win_capw = 0, win_cap_el = 0, win_NB = 0, win_NBc = 0, win_stride = 1, win_threads = 256, win_ppt = 4, win_gs = 1;
    int win_rec_lds = 0; /* waypoint records parked in the slice workgroup's LDS (0: in global slots) */
    int win_nkept = 0, win_first_kept = 0, win_el_expect = 0;
    float win_px0 = 0.f;
    DevBuf<float> win_px;
    DevBuf<int> win_cnt;
    DevBuf<float4> win_pts;
    DevBuf<MinMaxPart> win_part;
    DevBuf<float4> wps_xyz, wps_normal, wps_rec;
    DevBuf<int> fin_ticket; /* arrivals of the window finish launch's workgroups (the last one publishes the meta block) */
    DevBuf<int> wps_nn;
    DevBuf<float> wps_pre;

    DevMeta hmeta;
    DevMeta *hmeta_pinned = nullptr; /* the hot calls end with an async copy of the device meta into it */
    /* a new cloud's plan without the host in the middle: k_ingest_minmax's last workgroup reduces the bounds and walks the slices,
       k_win_census_auto counts the windows, both write their results to pinned memory (PlanAuto + plane table + census) */
    DevBuf<int> plan_ticket;         /* [2], zero between launches */
    DevBuf<PlanAuto> plan_auto;
    bool auto_valid = false;         /* the pinned census belongs to the cloud just set, with auto_S slices and auto_pad */
    /* Plan reuse: a planner that is fed one scan after the other plans clouds of one size with one set of parameters.  The first
       plan takes its window capacities from a census of that cloud; a later cloud with the same point count, parameters, slice
       count and pad inherits them (+4 %) and skips the census launch -- the pass itself detects a window that does not fit
       (WIN_FLAG_OVERFLOW), and the plan is then made again from a census of its own */
    bool plan_reuse = true;          /* ppp_set_plan_reuse */
    bool inh_valid = false;          /* the members below describe a census-based window plan of this handle */
    int inh_S = 0, inh_n = 0, inh_max_w = 0, inh_max_el = 0;
    float inh_pad = 0.f;
    ppp_params inh_P;
    bool auto_px_only = false;       /* the cloud just set brought walk + pad along (auto_S, auto_pad, pinned plane table), but no census */
    bool plan_inherited = false;     /* the current window plan's capacities are inherited */
    int auto_S = 0;
    float auto_pad = 0.f;
    bool slab_cnt_used = true;       /* a slab-path pass has been enqueued since the slab histogram was last cleared by the plan */
    char *pin = nullptr;             /* pinned staging for the small copies of the plan (bounds partials, plane table, census) */
    size_t pin_bytes = 0;
    hipError_t ensure_pin(size_t bytes)
    {
        if (bytes <= pin_bytes) return hipSuccess;
        if (pin) { (void)hipHostFree(pin); pin = nullptr; pin_bytes = 0; }
        const size_t want = std::max<size_t>(bytes, 128 * 1024);
        hipError_t e = hipHostMalloc((void **)&pin, want, hipHostMallocDefault);
        if (e == hipSuccess) pin_bytes = want;
        return e;
    }
    char *pcd_stage[2] = {nullptr, nullptr}; /* ppp_set_cloud_pcd: two pinned pieces ... */
    size_t pcd_stage_bytes = 0;
    hipEvent_t pcd_ev[2] = {nullptr, nullptr}; /* ... and the event behind each one's copy */
    bool meta_in_flight = false;
    /* A cloud set while the handle holds a window plan of an earlier cloud of the same size and parameters does not wait for its
       bounds: the conversion pass is enqueued, the plan stays, and the pass of the new cloud may be enqueued right
       behind it -- the device checks walk length, pad, bounds and capacities against the record that pass leaves, and hands a
       pass back whose plan does not fit.  plan_deferred: that record has not been read yet (resolve_deferred does, at the first
       call that is not one of the three enqueue-only entry points). */
    int side_by_side = 1; /* handles the caller runs side by side on this device (ppp_set_side_by_side): from two on the slice workgroups of small windows stay at 512 threads */
    bool plan_deferred = false, deferred_census = false;
    bool rec_current = false;   /* plan_auto holds the record of the resident cloud (it came through k_ingest_minmax and was not altered since) */
    bool plan_walk_ok = false;  /* the window plan's S, pad and plane table are the device's own, bit for bit (plan_window: census that came with the cloud, or inherited) */
    bool meta_fresh = false; /* hmeta is the device's block as of now: nothing was launched on this handle since it was fetched (every launch clears it) */
    bool chain_calls = false;       /* GenPath is followed by getPath in the same enqueue: its meta copy is skipped */
    float *out2 = nullptr;          /* batched form: the emitting launch also writes the list here (at most out2_cap rows) */
    int out2_cap = 0;
    float *last_out2 = nullptr;     /* where the last batch put this handle's list: a re-run after an LDS overflow writes there too */
    int last_out2_cap = 0;
    std::shared_ptr<struct BatchMetas> bmetas; /* batched launches publish every member's meta block in one pinned array ... */
    size_t bslot = 0;                          /* ... this handle's is entry bslot */
    bool meta_from_batch = false;
    int internal = 0;               /* > 0 while GenPath / getPath are enqueued on behalf of a batch or a re-run (keeps last_out2) */
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    unsigned epoch = 0;                 /* bumped whenever the launch sequence of this handle changes */
    unsigned graph_epoch_seen = ~0u;    /* ppp_run_async: the plan epoch of the last call (the first call of a plan runs eagerly) */
    hipStream_t pending_stream = nullptr; /* a batch graph launched on another handle's stream carries this handle's work */
    struct BatchGraph *batches[2] = {nullptr, nullptr}; /* cached batch graphs (lead handle only): two, so a caller can
                                                           alternate between two destination buffers (double buffering) */
    int batch_next = 0;                                 /* slot the next new graph replaces */
    bool timing = false;
    std::vector<KTimer> timers;

    void drop_graph()
    {
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (graph) (void)hipGraphDestroy(graph);
        graph_exec = nullptr; graph = nullptr;
        ++epoch;
    }
    void drop_batch();
    ~ppp_handle_s()
    {
        (void)hipSetDevice(device);
        X.release(); Y.release(); Z.release(); Xp.release(); Yp.release(); Zp.release(); part_idx.release(); unsorted4.release(); sorted4.release();
        slab_cnt.release(); slab_start.release(); slab_cursor.release(); coarse_cursor.release(); slab_ytab.release(); slab_xmin.release(); slab_xmax.release();
        meta.release(); px.release(); lo.release(); hi.release(); node_x.release(); node_y.release(); node_z.release();
        normals4.release(); dyn_bnd_pts.release(); dyn_adj_pts.release(); dyn_first_ab.release(); dyn_first_snap.release(); dyn_first_node.release(); ell_cs.release(); dyn_bnd_knots.release(); dyn_bnd_n.release(); plan_ticket.release(); plan_auto.release();
        node_start.release(); node_cnt.release(); band_cnt.release(); wp_cnt.release(); wp_off.release(); tail.release(); slice_wpcnt.release();
        wp_xyz.release(); wp_normal.release(); wp_nn.release(); wp_pre.release(); wp_smooth.release(); wp_out.release();
        mm_part.release(); big_slabs.release(); big_slices.release(); arena.release(); scratch.release(); pack_tab.release(); pack_out.release();
        win_px.release(); win_cnt.release(); win_pts.release(); win_part.release(); wps_xyz.release(); wps_normal.release(); wps_nn.release(); wps_pre.release(); wps_rec.release();
        drop_graph();
        drop_batch();
        if (hmeta_pinned) (void)hipHostFree(hmeta_pinned);
        for (int b = 0; b < 2; ++b) { if (pcd_stage[b]) (void)hipHostFree(pcd_stage[b]); if (pcd_ev[b]) (void)hipEventDestroy(pcd_ev[b]); }
        if (pin) (void)hipHostFree(pin);
        for (auto &t : timers) { for (auto e : t.e0) (void)hipEventDestroy(e); for (auto e : t.e1) (void)hipEventDestroy(e); }
        if (stream) (void)hipStreamDestroy(stream);
        if (back) { delete back; back = nullptr; }
    }
};

/* the meta blocks of a batch on the host (pinned), shared by the batch graph and the member handles that read them */
struct BatchMetas {
    DevMeta *pinned = nullptr;
    ~BatchMetas() { if (pinned) (void)hipHostFree(pinned); }
};
struct BatchGraph {
    std::vector<ppp_handle> hs;
    std::vector<unsigned> epochs;
    float *dst = nullptr;
    std::vector<size_t> off, cap;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    hipEvent_t fork = nullptr;
    std::vector<hipEvent_t> join;
    /* batched form (one launch per stage over all members): the members' records and meta blocks */
    bool batched = false, eager = false; /* eager: launched directly every time (kernel timing), no graph */
    int maxB = 1, max_slab_cap = 2048, max_capb = 1024; /* launch geometry over all members */
    int pose_threads = 256, slice_thr = SLICE_KD_T;
    size_t pose_lds = 0;
    int gx_mm = 1, gx_scat = 1, gx_sort = 1, gx_slice = 1, gx_pose = 1, gx_smooth = 1;
    bool full_slabs = false, ppt8 = false;
    DevBuf<BatchMember> members;
    bool win = false;              /* every member runs the window path: the three k_win_*_b launches */
    DevBuf<WinArgs> wmembers;
    int win_ppt = 4, win_threads = 256, gx_wfin = 1;
    bool win_staged = false;
    size_t win_lds = 0, win_scat_lds = 0, win_fin_lds = 0;
    DevBuf<DevMeta> metas;
    std::shared_ptr<BatchMetas> hmetas;
    ~BatchGraph()
    {
        if (ge) (void)hipGraphExecDestroy(ge);
        if (g) (void)hipGraphDestroy(g);
        if (fork) (void)hipEventDestroy(fork);
        for (auto e : join) if (e) (void)hipEventDestroy(e);
        members.release(); wmembers.release(); metas.release();
    }
};
void ppp_handle_s::drop_batch()
{
    for (auto &b : batches) { delete b; b = nullptr; }
}

namespace {

int fail(ppp_handle h, int code, const std::string &msg)
{
    if (h) h->err = msg;
    return code;
}
#define HIPCHK(h, expr)                                                                               \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return fail(h, PPP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)

DevParams dev_params(const ppp_handle h)
{
    DevParams D;
    memset(&D, 0, sizeof(D));
    D.tool_radius = h->P.tool_radius; D.path_resolution = h->P.path_resolution; D.rpy_resolution = h->P.rpy_resolution;
    D.trim = h->P.trim; D.ee_length = h->P.ee_length; D.normal_radius = h->P.normal_radius;
    memcpy(D.handeye, h->P.handeye, sizeof(D.handeye));
    memcpy(D.viewpoint, h->vp, sizeof(D.viewpoint));
    D.change_range = h->P.change_range; D.pairing = h->P.pairing; D.walk = h->P.walk; D.drop_ends = h->P.drop_ends;
    D.smooth = h->P.smooth; D.smooth_max_sweeps = h->P.smooth_max_sweeps;
    D.slice_begin = h->P.slice_begin; D.slice_end = h->P.slice_end; D.ranged = h->ranged ? 1 : 0;
    D.incl_lo = h->incl_lo; D.incl_hi = h->incl_hi;
    D.knots_on_plane = h->P.dynamic_adjustment ? 0 : 1;
    D.bounds_given = (h->use_part || h->part_given) ? 1 : 0; D.g_nvalid = h->h_nvalid;
    for (int d = 0; d < 3; ++d) { D.g_mn[d] = h->h_mn[d]; D.g_mx[d] = h->h_mx[d]; }
    {   /* mean spacing of a sheet-like cloud from its bounding rectangle; only a search hint, never a cut-off */
        const double area = ((double)h->h_mx[0] - h->h_mn[0]) * ((double)h->h_mx[1] - h->h_mn[1]);
        const double spacing = (area > 0 && h->h_nvalid > 0) ? std::sqrt(area / h->h_nvalid) : 1.0;
        const double r = std::max(1.0, PPP_NN_HINT_SPACINGS * spacing);
        D.nn_hint2 = (float)(r * r);
    }
    return D;
}

KTimer *timer_for(ppp_handle h, const char *name)
{
    KTimer *t = nullptr;
    for (auto &x : h->timers) if (x.name == name) { t = &x; break; }
    if (!t) { h->timers.emplace_back(); t = &h->timers.back(); t->name = name; }
    if ((size_t)t->used >= t->e0.size()) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        t->e0.push_back(a); t->e1.push_back(b);
    }
    return t;
}

/* synchronous copy on the handle's own stream.  Never hipMemcpy: that runs on the legacy stream, which HIP refuses
   (and which poisons the other thread's capture) while ANY thread is capturing a graph -- distinct handles are driven
   from concurrent host threads. */
hipError_t copy_sync(ppp_handle h, void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, h->stream);
    return e == hipSuccess ? hipStreamSynchronize(h->stream) : e;
}

/* launch helper: optional hipEvent bracket on the handle's stream */
#define LAUNCH(h, name, kern, grid, block, shmem, ...)                                                \
    do {                                                                                              \
        KTimer *_t = (h)->timing ? timer_for((h), name) : nullptr;                                    \
        if (_t) (void)hipEventRecord(_t->e0[_t->used], (h)->stream);                                  \
        (void)hipGetLastError(); /* the check below must not pick up an older, unrelated error */     \
        (h)->meta_fresh = false;                                                                      \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), (shmem), (h)->stream, __VA_ARGS__);         \
        if (_t) { (void)hipEventRecord(_t->e1[_t->used], (h)->stream); _t->used++; }                  \
        hipError_t _le = hipGetLastError();                                                           \
        if (_le != hipSuccess) return fail((h), PPP_ERR_HIP, std::string(name) + ": " + hipGetErrorString(_le)); \
    } while (0)

int validate_params(ppp_handle h, const ppp_params *p)
{
    if (!(p->tool_radius > 0) || (int)(p->tool_radius * 2) <= 0) return fail(h, PPP_ERR_ARG, "Tool_Radius must give an integer step >= 1");
    if (!(p->path_resolution > 0)) return fail(h, PPP_ERR_ARG, "PathResolution must be > 0");
    if (!(p->trim >= 0)) return fail(h, PPP_ERR_ARG, "trim must be >= 0");
    if (p->pairing != PPP_PAIR_KD && p->pairing != PPP_PAIR_BRUTE) return fail(h, PPP_ERR_ARG, "pairing");
    if (p->walk < 0 || p->walk > 4) return fail(h, PPP_ERR_ARG, "walk");
    if (!(p->normal_radius > 0)) return fail(h, PPP_ERR_ARG, "normal_radius");
    if (p->smooth_max_sweeps < 1 || p->smooth_max_sweeps > SM_MAXS) return fail(h, PPP_ERR_ARG, "smooth_max_sweeps must be in [1, 512]");
    if (p->alignment) return fail(h, PPP_ERR_ARG, "ppp_params.alignment must be 0: Alignment / Smooth / RemoveOutlier are calls on the resident cloud (ppp_trans2center, ppp_smooth_mls, ppp_remove_outlier), not plan parameters");
    if (p->slice_begin < 0 || (p->slice_end > 0 && p->slice_end < p->slice_begin)) return fail(h, PPP_ERR_ARG, "slice_begin / slice_end");
    if ((p->slice_begin > 0 || p->slice_end > 0) && !(p->range_margin >= 2 * p->normal_radius))
        return fail(h, PPP_ERR_ARG, "range_margin must be at least twice the normal radius");
    if ((p->slice_begin > 0 || p->slice_end > 0) && p->dynamic_adjustment)
        return fail(h, PPP_ERR_UNSUPPORTED, "Dynamic_adjustment chains slice s to slice s-1: it does not shard by slice range (SURVEY.md 8e)");
    if (p->dynamic_adjustment) {
        if (p->walk != PPP_WALK_CENTER_INT && p->walk != PPP_WALK_SDIR_INT && p->walk != PPP_WALK_V1_CONTACT)
            return fail(h, PPP_ERR_UNSUPPORTED, "Dynamic_adjustment exists for the connect / connect1 / main planners only (walks center_int, sdir_int, v1_contact); SectPath and slicing_method have none");
        if (p->curvature_k < 3 || p->curvature_k > 64) return fail(h, PPP_ERR_ARG, "curvature_k must be in [3, 64]");
        if (!(p->depth > 0) || !(p->adjust_threshold >= 0) || !(p->toolthickness > 0)) return fail(h, PPP_ERR_ARG, "depth / Adjust_Threshold / toolthickness");
    }
    return PPP_OK;
}

/* Launch-geometry / search-radius overrides of the tuning scripts (tools/_run_*.sh): read only by builds made with -DPPP_TUNING
   (make variant NAME=tune DEFS=-DPPP_TUNING); the product library ignores them, so a stray variable cannot change a plan. */
static inline const char *tuning_env(const char *name)
{
#ifdef PPP_TUNING
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

DynParams dyn_params(const ppp_handle h)
{
    DynParams D;
    D.tool_radius = h->P.tool_radius; D.depth = h->P.depth; D.toolthickness = h->P.toolthickness;
    D.adjust_threshold = h->P.adjust_threshold; D.k = h->P.curvature_k;
    /* first radius of the k-NN gather: k points of a sheet of the cloud's mean areal density, +25 % */
    double area = ((double)h->h_mx[0] - h->h_mn[0]) * ((double)h->h_mx[1] - h->h_mn[1]);
    double rho = (area > 0 && h->h_nvalid > 0) ? (double)h->h_nvalid / area : 1.0;
    D.r0 = (float)std::max(0.5, 1.25 * std::sqrt((double)D.k / (3.14159265358979 * rho)));
    if (const char *ev = tuning_env("PPP_DYN_R0F")) D.r0 = (float)std::max(0.5, atof(ev) * std::sqrt((double)D.k / (3.14159265358979 * rho))); /* tuning runs only */
    D.r1 = (float)std::max(0.25, 2.0 * std::sqrt(1.0 / (3.14159265358979 * rho)));
    return D;
}

int ensure_dynamic_buffers(ppp_handle h)
{
    HIPCHK(h, h->normals4.ensure(std::max<size_t>(h->n, 1)));
    HIPCHK(h, h->dyn_bnd_pts.ensure(2 * (size_t)h->dyn_maxNB)); HIPCHK(h, h->dyn_adj_pts.ensure(2 * (size_t)h->dyn_maxNA));
    /* every boundary of a pass is kept (a slot per slice) for ppp_get_boundary -- the curves the reference's viewer paints green --
       unless that is more than 1 GiB: then the chains' two current ones only */
    const size_t slot_doubles = 3 * ((size_t)h->dyn_maxNB + 2), nslots = (size_t)std::max(h->S_cap, 2);
    h->dyn_keep_all = nslots * slot_doubles * sizeof(double) <= ((size_t)1 << 30);
    HIPCHK(h, h->dyn_bnd_knots.ensure((h->dyn_keep_all ? nslots : 2) * slot_doubles)); HIPCHK(h, h->dyn_bnd_n.ensure(4 + nslots));
    const size_t nfirst = (size_t)std::max(h->S_cap, 1) * h->dyn_maxNA;
    if (nfirst > ((size_t)1 << 27)) /* 56 bytes a node: 7.5 GB */
        return fail(h, PPP_ERR_CAPACITY, "dynamic adjustment: slices x nodes per slice beyond 2^27");
    HIPCHK(h, h->dyn_first_ab.ensure(nfirst)); HIPCHK(h, h->dyn_first_node.ensure(3 * nfirst)); HIPCHK(h, h->dyn_first_snap.ensure(nfirst));
    if (!h->ell_cs.p) {
        /* cos / sin of the 721 ellipse angles, computed as the reference does (float angle, pcl::deg2rad,
           std::cos(float)) with the host libm so the device uses the very same values */
        std::vector<float> cs(2 * DYN_ELL);
        int a = 0;
        for (float angle(0.0); angle <= 360.0 && a < DYN_ELL; angle += 0.5, ++a) {
            const float rad = angle * 0.017453293f;
            cs[2 * a] = std::cos(rad); cs[2 * a + 1] = std::sin(rad);
        }
        HIPCHK(h, h->ell_cs.ensure(2 * DYN_ELL));
        HIPCHK(h, hipMemcpyAsync(h->ell_cs.p, cs.data(), sizeof(float) * 2 * DYN_ELL, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return PPP_OK;
}

/* whole-cloud normals into normals4 (cloud index order); needs the slab index */
int enqueue_normals(ppp_handle h)
{
    const size_t n = h->n;
    DevParams D = dev_params(h);
    HIPCHK(h, hipMemsetAsync(h->normals4.p, 0xff, std::max<size_t>(n, 1) * 16, h->stream)); /* dropped points: NaN */
    if (n)
        LAUNCH(h, "k_normals_all", k_normals_all, (unsigned)((n + 255) / 256), 256, 0, h->meta.p, D, h->sorted4.p, h->slab_start.p,
               h->slab_xmin.p, h->slab_xmax.p, h->slab_ytab.p, -1, h->normals4.p);
    return PPP_OK;
}

/* pinned layout of the plan results of a new cloud */
constexpr size_t PIN_REC0 = 0, PIN_REC1 = 64, PIN_PX = 128, PIN_CENSUS = PIN_PX + sizeof(float) * WIN_AUTO_SCAP,
                 PIN_AUTO_BYTES = PIN_CENSUS + sizeof(int) * 3 * WIN_AUTO_SCAP;

/* Threads of a slice workgroup for a launch of `wgs` of them.  The kernel holds 116 VGPRs, i.e. 16 waves per CU.  While a
   launch has fewer workgroups than the device has room for, a workgroup is as wide as its work can use (a left point per
   thread in the pairing, 4 .. 8 lanes per waypoint in the pose stage): the launch ends with its slowest workgroup.  A launch
   of several rounds of workgroups (batches, cfg 5) is about throughput: as many workgroups per CU as the LDS allows, the 16
   waves shared between them (measured, 64 x 250 k points: 256 threads 0.42 ms, 320 .. 512 threads 0.58 .. 0.64 ms). */
size_t win_slice_lds_for(const ppp_handle h, int NBc)
{   /* (the checking workgroup of the launch keeps the walk and its scratch there) */
    return std::max(win_slice_lds_bytes(h->win_capw, h->win_cap_el, WIN_CLASSES * NBc), sizeof(float) * ((size_t)h->S_cap + 2048) + 64);
}
/* several workgroups per CU, at once or one after the other? */
bool win_throughput_launch(const ppp_handle h, long long wgs) { return wgs > (long long)h->num_cus; }

int win_pick_threads(const ppp_handle h, long long wgs)
{
    const int capw = h->win_capw, cap_el = h->win_cap_el;
    int tmin = std::max(128, 64 * ((capw + 64 * WIN_EMAX - 1) / (64 * WIN_EMAX)));
    tmin = std::max(tmin, 64 * ((cap_el + 64 * WIN_CE - 1) / (64 * WIN_CE)));
    if (tmin > 1024) return 0;
    const int wide = std::min(1024, std::max(256, 64 * (int)std::ceil(std::max(1.05 * (double)h->win_el_expect, 4.0 * (double)h->cnt_est) / 64.0)));
    const size_t lds = win_slice_lds_for(h, win_throughput_launch(h, wgs) ? h->win_NBc_thr : h->win_NBc) + 1024;
    const int by_lds = (int)std::max<size_t>(1, (size_t)h->max_lds / lds);
    const long long per_cu = (wgs + h->num_cus - 1) / std::max(1, h->num_cus);
    int T = wide;
    if (win_throughput_launch(h, wgs)) {
        const int conc = (int)std::min<long long>(by_lds, per_cu);
        int waves_cu = WIN_SLICE_WAVES_PER_CU;
        if (const char *ev = tuning_env("PPP_WIN_WAVES_CU")) waves_cu = std::max(4, atoi(ev)); /* tuning runs only */
        T = std::min(wide, 64 * std::max(1, waves_cu / conc));
    }
    /* Several handles side by side (ppp_set_side_by_side): the launches of neighbouring passes share the CUs, and two 1024-thread slice
       workgroups fill a CU's 2048 thread slots -- no room for a binning or finish workgroup of another pass.  512-thread workgroups
       leave it (1 M points / 256 slices, three handles: 0.0377 -> 0.0286 ms per step; alone a pass is 5 % slower: 0.065 -> 0.068).
       Only where a window is small enough for six staged points per thread: at 2 M points / 256 slices (704 threads at least) and at
       10 M / 1024 narrower workgroups were slower side by side too. */
    if (h->side_by_side >= 2 && !win_throughput_launch(h, wgs) && capw <= 3072) T = std::min(T, 512);
    T = std::max(T, tmin);
    if (const char *ev = tuning_env("PPP_WIN_T")) { /* tuning runs only */
        const int tv = atoi(ev);
        if (tv >= tmin && tv <= 1024 && tv % 64 == 0) T = tv;
    }
    return T;
}

/* The window path (ppp_window.h) for this plan, when it applies: kd pairing without dynamic adjustment or alignment, windows
   [Px - pad, Px + pad] that do not overlap (tool steps of about 2 pad + 2 mm and more) and that fit a workgroup's LDS.  Everything
   else -- and every pass the window path hands back -- runs on the slab index.  Sizes come from the cached bounds, as the slab
   grid's do; the device re-derives bounds and walk in every pass and checks them against this plan. */
int plan_window(ppp_handle h, int S, double per)
{
    h->win_path = false;
    h->plan_walk_ok = false;
    const int step = (int)(h->P.tool_radius * 2);
    if (!h->win_allowed || h->win_disabled || getenv("PPP_NO_WINDOW_PATH")) return PPP_OK;
    if (h->P.dynamic_adjustment || h->aligned || h->big_path) return PPP_OK; /* (both pairings: kd and v1's brute-force greedy) */
    if (h->h_nvalid <= 0 || S < 1 || S > WIN_S_MAX || step < 1) return PPP_OK;
    const double rx = (double)h->h_mx[0] - h->h_mn[0], ry = (double)h->h_mx[1] - h->h_mn[1];
    const double area = rx * ry;
    const double spacing = area > 0 ? std::sqrt(area / h->h_nvalid) : 1.0;
    /* the band reaches 2 + |Px - trunc(Px)| < 3 mm from the plane; the nearest point of a waypoint lies within a point spacing or
       so of the plane, its normal neighbourhood a radius further */
    const float pad = (float)std::max(3.0, (double)h->P.normal_radius + std::max(1.5, spacing));
    std::vector<float> px((size_t)S);
    ppp_slice_walk(h->P.walk, h->h_mn[0], h->h_mx[0], h->P.tool_radius, px.data(), S);
    for (int s = 0; s < S; ++s) {
        if (!(std::fabs((double)px[s]) < 1.0e7)) return PPP_OK;
        if (s + 1 < S && !((double)px[s + 1] - (double)px[s] > 2.0 * pad + 1.0e-2)) return PPP_OK; /* windows would overlap */
    }
    /* exact populations of the windows of this handle's slices and of their band sides: one pass over the x coordinates at plan time */
    const int n_src = h->use_part ? h->n_part : (int)h->n;
    HIPCHK(h, h->win_px.ensure((size_t)S)); HIPCHK(h, h->win_cnt.ensure(std::max<size_t>(3, WIN_CNT_STRIDE) * (size_t)S));
    /* a cloud that has just been set brought its census along (refresh_bounds_and_plan): taken when the device's walk, slice count
       and pad are this plan's, bit for bit */
    const bool walk_ok = !h->use_part && h->auto_S == S && S <= WIN_AUTO_SCAP && h->pin &&
                         memcmp(&h->auto_pad, &pad, sizeof(float)) == 0 && memcmp(h->pin + PIN_PX, px.data(), sizeof(float) * (size_t)S) == 0;
    const bool from_auto = h->auto_valid && walk_ok;
    /* ... or it brought the walk only, and the capacities of an earlier cloud's plan apply (refresh_bounds_and_plan) */
    const bool inherit = !from_auto && h->auto_px_only && walk_ok && h->inh_valid && h->inh_S == S && memcmp(&h->inh_pad, &pad, sizeof(float)) == 0 &&
                         h->sb == 0 && h->se == S;
    h->auto_valid = false; h->auto_px_only = false;
    h->plan_inherited = false;
    h->plan_walk_ok = from_auto || inherit;
    int *census = nullptr;
    std::vector<int> inherited;
    if (from_auto) census = (int *)(h->pin + PIN_CENSUS);
    else if (inherit) { /* every window as full as the fullest of the earlier cloud, + 4 % */
        inherited.assign(3 * (size_t)S, 0);
        for (int s2 = 0; s2 < S; ++s2) {
            inherited[s2] = h->inh_max_w + h->inh_max_w / 25 + 8;
            inherited[(size_t)S + s2] = inherited[2 * (size_t)S + s2] = h->inh_max_el + h->inh_max_el / 25 + 8;
        }
        census = inherited.data();
        h->plan_inherited = true;
    } else {
    HIPCHK(h, h->ensure_pin(sizeof(float) * 4 * (size_t)S));
    float *px_pin = (float *)h->pin;
    census = (int *)(px_pin + S);
    memcpy(px_pin, px.data(), sizeof(float) * (size_t)S);
    memset(census, 0, sizeof(int) * 3 * (size_t)S);
    HIPCHK(h, hipMemcpyAsync(h->win_px.p, px_pin, sizeof(float) * (size_t)S, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(h->win_cnt.p, 0, sizeof(int) * 3 * (size_t)S, h->stream));
    if (n_src > 0) {
        const float *cx = h->use_part ? h->Xp.p : h->X.p;
        if (S <= 4096) /* the counters fit a workgroup's LDS: a few hundred workgroups, each flushing its non-zero counters once */
            LAUNCH(h, "k_win_census", k_win_census<true>, std::max(1, std::min((n_src + 4095) / 4096, 512)), 256, sizeof(int) * 3 * (size_t)S, cx, n_src,
                   h->win_px.p, S, px[0], 1.0f / (float)step, pad, h->win_cnt.p, h->win_cnt.p + S, h->win_cnt.p + 2 * (size_t)S);
        else
            LAUNCH(h, "k_win_census", k_win_census<false>, std::max(1, std::min((n_src + 255) / 256, 4096)), 256, 0, cx, n_src,
                   h->win_px.p, S, px[0], 1.0f / (float)step, pad, h->win_cnt.p, h->win_cnt.p + S, h->win_cnt.p + 2 * (size_t)S);
        HIPCHK(h, hipMemcpyAsync(census, h->win_cnt.p, sizeof(int) * 3 * (size_t)S, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    int max_w = 0, max_el = 0;
    for (int s2 = h->sb; s2 < h->se; ++s2) {
        max_w = std::max(max_w, census[s2]); max_el = std::max(max_el, census[(size_t)S + s2]);
        if (h->P.pairing == PPP_PAIR_BRUTE) max_el = std::max(max_el, census[2 * (size_t)S + s2]); /* its per-side tables hold either side */
    }
    const double expect = std::max(1, max_w);
    int NBc = 16; /* y-buckets per class: about two per point of a class where the LDS allows (most buckets then hold one point or none) */
    while (NBc < expect / 5.0 && NBc < 4096) NBc <<= 1;
    /* capacities = this cloud's own maxima (rounded up to 64 points); what gives way under LDS pressure is the bucket table */
    const size_t budget = (size_t)h->max_lds - 2048;
    int capw = 64 * ((max_w + 63) / 64 + 0), cap_el = 64 * ((max_el + 63) / 64);
    capw = std::max(capw, 64); cap_el = std::max(cap_el, 64);
    while (win_slice_lds_bytes(capw, cap_el, WIN_CLASSES * NBc) > budget && NBc > 32) NBc >>= 1;
    if (win_slice_lds_bytes(capw, cap_el, WIN_CLASSES * NBc) > budget) capw = 0;
    if (!capw) return PPP_OK; /* the windows of this cloud do not fit a workgroup's LDS */
    /* ... and for launches with several workgroups per CU half as many (about one per point: the table's LDS is worth more as
       room for another workgroup; measured on 64 x 250 k points: 256 buckets per class and 4 workgroups per CU 0.42 ms, 512 and
       3 workgroups 0.62 ms, 128 and 5 workgroups 0.49 ms) */
    h->win_NBc_thr = std::max(16, NBc / 2);
    while (h->win_NBc_thr * 2 <= NBc && h->win_NBc_thr < expect / 10.0) h->win_NBc_thr <<= 1;
    if (const char *ev = tuning_env("PPP_WIN_NBC_THR")) { const int v = atoi(ev); if (v >= 16 && v <= NBc && (v & (v - 1)) == 0) h->win_NBc_thr = v; } /* tuning runs only */
    const int NB = WIN_CLASSES * NBc;
    h->win_el_expect = max_el;
    h->win_capw = capw; h->win_cap_el = cap_el; h->win_NB = NB; h->win_NBc = NBc;
    int T = win_pick_threads(h, std::max(1, h->se - h->sb));
    if (!T) return PPP_OK;
    /* points per thread of the binning launch: 8 from 1.2 million points on (16 was slower at 10 M points: 107 against 100 us) */
    h->win_ppt = n_src > PPP_WIN_PPT8_FROM ? 8 : 4;
    /* large clouds leave the binning launch through LDS in window order (write amplification 2.1 -> ~1.3 at 10 M points), with
       as many points per thread as the stage has room for */
    {
        int from = PPP_PPT16_FROM;
        if (const char *ev = tuning_env("PPP_WIN_STAGE_FROM")) from = atoi(ev); /* tuning runs only */
        h->win_staged = n_src > from && !tuning_env("PPP_WIN_NO_STAGE");
    }
    if (h->win_staged) {
        h->win_ppt = 8;
        if (win_scatter_lds_bytes(S, 8, WSC_T, true) + 2048 > (size_t)h->max_lds) h->win_ppt = 4;
        if (win_scatter_lds_bytes(S, h->win_ppt, WSC_T, true) + 2048 > (size_t)h->max_lds) { h->win_staged = false; h->win_ppt = 8; }
    }
    if (const char *ev = tuning_env("PPP_WIN_PPT")) { const int pv = atoi(ev); if ((pv == 2 || pv == 4 || pv == 8) && !h->win_staged) h->win_ppt = pv; } /* tuning runs only */
    h->win_gs = std::max(1, (n_src + h->win_ppt * WSC_T - 1) / (h->win_ppt * WSC_T));
    if (h->win_staged) h->win_gs = std::min(h->win_gs, std::max(1, h->num_cus)); /* the staged form loops over its chunks: a workgroup per CU (its LDS admits no second) */
    h->win_pad = pad; h->win_capw = capw; h->win_cap_el = cap_el; h->win_NB = NB; h->win_NBc = NBc; h->win_threads = T;
    h->win_stride = std::max(1, (int)per);
    /* the waypoints' records between their searches and their pose (13 words each) wait in the pairing scratch the knots leave free
       -- seven word rows in the 4 bytes per left point of `cz`, six in the candidate histogram's -- when a slice's waypoints fit a row
       (every BASELINE configuration: a left point per 1.1 mm of y, a waypoint per 7); else in global slots */
    h->win_rec_lds = (h->win_stride <= cap_el / 7) ? cap_el / 7 : 0;
#ifdef WIN_REC_GLOBAL /* (A/B builds) */
    h->win_rec_lds = 0;
#endif
    h->win_first_kept = h->P.drop_ends ? 1 : 0;
    h->win_nkept = std::max(0, h->P.drop_ends ? S - 2 : S);
    h->win_px0 = px[0];
    HIPCHK(h, h->win_part.ensure((size_t)std::max(h->win_gs, (n_src + 2 * WSC_T - 1) / (2 * WSC_T)))); /* (a batch may bin with fewer points per thread) */
    HIPCHK(h, h->win_pts.ensure((size_t)S * (size_t)capw));
    {   /* the knot arrays hold a cap_el segment per slice on this path */
        const double need = (double)S * (double)cap_el;
        if (need > 1.0e9) return PPP_OK;
        h->node_cap = std::max(h->node_cap, (int)need);
        HIPCHK(h, h->node_x.ensure(h->node_cap)); HIPCHK(h, h->node_y.ensure(h->node_cap)); HIPCHK(h, h->node_z.ensure(h->node_cap));
    }
    const size_t slots = (size_t)std::max(1, h->win_nkept) * (size_t)h->win_stride;
    HIPCHK(h, h->wps_xyz.ensure(slots)); HIPCHK(h, h->wps_normal.ensure(slots)); HIPCHK(h, h->wps_nn.ensure(slots)); HIPCHK(h, h->wps_pre.ensure(6 * slots)); HIPCHK(h, h->wps_rec.ensure(4 * slots));
    if (!from_auto && !inherit) /* the windows' counters: every pass leaves them cleared again (in stream order ahead of the first pass: no wait); the census that came with the cloud has cleared its own */
        HIPCHK(h, hipMemsetAsync(h->win_cnt.p, 0, sizeof(int) * (size_t)S * std::max<size_t>(3, WIN_CNT_STRIDE), h->stream));
    h->win_path = true;
    if (!inherit && h->sb == 0 && h->se == S && !h->use_part) { /* what a later cloud of this size may inherit */
        h->inh_valid = true; h->inh_S = S; h->inh_n = n_src; h->inh_max_w = max_w; h->inh_max_el = max_el; h->inh_pad = pad; h->inh_P = h->P;
    }
    if (getenv("PPP_WIN_DEBUG"))
        fprintf(stderr, "[ppp] window plan: S %d [%d,%d) pad %.2f capw %d cap_el %d NBc %d (throughput %d) threads %d ppt %d lds %zu B max window %d max left side %d census %s\n",
                S, h->sb, h->se, pad, capw, cap_el, NBc, h->win_NBc_thr, T, h->win_ppt, win_slice_lds_for(h, NBc), max_w, max_el,
                from_auto ? "came with the cloud" : (inherit ? "inherited from an earlier cloud of this size" : "at plan time"));
    return PPP_OK;
}

/* the arguments of the three window launches for this handle */
WinArgs win_args(const ppp_handle h)
{
    WinArgs A;
    memset(&A, 0, sizeof(A));
    A.m = h->meta.p; A.P = dev_params(h);
    A.X = h->use_part ? h->Xp.p : h->X.p; A.Y = h->use_part ? h->Yp.p : h->Y.p; A.Z = h->use_part ? h->Zp.p : h->Z.p;
    A.idmap = h->use_part ? h->part_idx.p : ((h->part_given && h->part_has_idx) ? h->part_idx.p : nullptr);
    A.n = h->use_part ? h->n_part : (int)h->n;
    A.plan_px = h->win_px.p;
    A.S = h->S_cap; A.sb = h->sb; A.se = h->se; A.first_kept = h->win_first_kept; A.nkept = h->win_nkept;
    A.pad = h->win_pad; A.px0 = h->win_px0; A.inv_step = 1.0f / (float)std::max(1, (int)(h->P.tool_radius * 2));
    A.y0 = h->h_mn[1];
    { const float yr = h->h_mx[1] - h->h_mn[1]; A.yscale = yr > 0.f ? (float)h->win_NBc / yr : 0.f; }
    for (int d = 0; d < 3; ++d) { A.plan_mn[d] = h->h_mn[d]; A.plan_mx[d] = h->h_mx[d]; }
    A.plan_nvalid = h->h_nvalid;
    A.capw = h->win_capw; A.cap_el = h->win_cap_el; A.NB = h->win_NB; A.NBc = h->win_NBc; A.stride = h->win_stride; A.rec_lds = h->win_rec_lds;
    A.W_cap = h->W_cap; A.node_cap = h->node_cap;
    A.g_scatter = h->win_gs; A.g_slice = std::max(0, h->se - h->sb); A.g_finish = h->sm_tiles;
    A.finish = h->ranged ? 0 : 1;
    A.win_cnt = h->win_cnt.p; A.win_pts = h->win_pts.p; A.win_part = h->win_part.p;
    A.px = h->px.p; A.lo = h->lo.p; A.hi = h->hi.p;
    A.node_x = h->node_x.p; A.node_y = h->node_y.p; A.node_z = h->node_z.p;
    A.node_start = h->node_start.p; A.node_cnt = h->node_cnt.p; A.band_cnt = h->band_cnt.p;
    A.wp_cnt = h->wp_cnt.p; A.wp_off = h->wp_off.p; A.tail = h->tail.p;
    A.wps_xyz = h->wps_xyz.p; A.wps_normal = h->wps_normal.p; A.wps_nn = h->wps_nn.p; A.wps_pre = h->wps_pre.p; A.wps_rec = h->wps_rec.p;
    A.wp_pre = h->wp_pre.p; A.wp_smooth = h->wp_smooth.p; A.wp_out = h->wp_out.p; A.out2 = h->out2; A.out2_cap = h->out2_cap;
    A.plan_rec = (h->rec_current && h->plan_walk_ok) ? h->plan_auto.p : nullptr;
    A.meta_host = h->hmeta_pinned; A.fin_ticket = h->fin_ticket.p; /* (a member of a batch of several publishes into the batch's pinned array: upload_members_win) */
    return A;
}
size_t win_slice_lds(const ppp_handle h) { return win_slice_lds_for(h, h->win_NBc); }
/* the same arguments with the bucket table of a many-workgroups-per-CU launch */
void win_args_throughput(const ppp_handle h, WinArgs &A)
{
    A.NBc = h->win_NBc_thr; A.NB = WIN_CLASSES * A.NBc;
    const float yr = h->h_mx[1] - h->h_mn[1];
    A.yscale = yr > 0.f ? (float)A.NBc / yr : 0.f;
}

/* GenPath on the window path: bounds + binning, then the per-slice kernel (which also does getPath's per-waypoint half) */
int enqueue_window_gen(ppp_handle h)
{
    WinArgs A = win_args(h);
    const bool thr = win_throughput_launch(h, A.g_slice);
    if (thr) win_args_throughput(h, A);
    const size_t scat_lds = win_scatter_lds_bytes(A.S, h->win_ppt, WSC_T, h->win_staged);
    if (h->win_staged && h->win_ppt == 8) LAUNCH(h, "k_win_scatter", (k_win_scatter<8, true>), A.g_scatter, WSC_T, scat_lds, A);
    else if (h->win_staged) LAUNCH(h, "k_win_scatter", (k_win_scatter<4, true>), A.g_scatter, WSC_T, scat_lds, A);
    else if (h->win_ppt == 8) LAUNCH(h, "k_win_scatter", (k_win_scatter<8, false>), A.g_scatter, WSC_T, scat_lds, A);
    else if (h->win_ppt == 2) LAUNCH(h, "k_win_scatter", (k_win_scatter<2, false>), A.g_scatter, WSC_T, scat_lds, A);
    else LAUNCH(h, "k_win_scatter", (k_win_scatter<4, false>), A.g_scatter, WSC_T, scat_lds, A);
    const int T = h->win_threads;
    const size_t lds = win_slice_lds_for(h, A.NBc);
    const int extra = tuning_env("PPP_WIN_NO_VERIFY") ? 0 : 1; /* (tuning runs only: what the checking workgroup costs the launch) */
    if (T <= 256) LAUNCH(h, "k_win_slice", k_win_slice<256>, A.g_slice + extra, T, lds, A);
    else if (T <= 512) LAUNCH(h, "k_win_slice", k_win_slice<512>, A.g_slice + extra, T, lds, A);
    else if (T <= 768) LAUNCH(h, "k_win_slice", k_win_slice<768>, A.g_slice + extra, T, lds, A);
    else LAUNCH(h, "k_win_slice", k_win_slice<1024>, A.g_slice + extra, T, lds, A);
    h->stage_compact = false;
    return PPP_OK;
}
/* the rest of getPath: offsets, compaction, postion_smooth / reduceRPY / flange */
int enqueue_window_finish(ppp_handle h)
{
    const WinArgs A = win_args(h);
    LAUNCH(h, "k_win_finish", k_win_finish, A.g_finish, SMF_T, sizeof(int) * ((size_t)A.nkept + 2), A);
    return PPP_OK;
}

/* sizes every workspace from the resident cloud + parameters; no kernel of the hot path allocates */
int make_plan(ppp_handle h)
{
    if (!h->have_cloud) return fail(h, PPP_ERR_ARG, "no cloud set");
    /* from here on the handle's members are rewritten step by step: a re-plan that fails half way (a slice range wider
       than the part this handle holds, an allocation) must leave no old plan, no finished-looking results and no
       captured graph behind for a later ppp_run_async to replay against the new members */
    h->planned = false; h->index_built = false; h->gen_done = false; h->meta_fresh = false; h->path_done = false; h->list_final = false;
    h->drop_graph();
    const int n = (int)h->n;
    /* exact slice count from the cached bounds (the device recomputes the same walk) */
    int S = h->h_nvalid ? ppp_slice_walk(h->P.walk, h->h_mn[0], h->h_mx[0], h->P.tool_radius, nullptr, 0) : 0;
    if (S >= PPP_WALK_HARD_MAX) return fail(h, PPP_ERR_CAPACITY, "slice walk does not terminate");
    /* slice range -> the x interval whose points this handle indexes */
    h->sb = std::min(std::max(0, h->P.slice_begin), S);
    h->se = (h->P.slice_end <= 0 || h->P.slice_end > S) ? S : h->P.slice_end;
    h->ranged = S > 0 && (h->sb > 0 || h->se < S);
    h->incl_lo = -INFINITY; h->incl_hi = INFINITY;
    h->n_range = h->h_nvalid;
    if (h->ranged) {
        if (h->sb >= h->se) { h->incl_lo = INFINITY; h->incl_hi = -INFINITY; h->n_range = 0; } /* empty range: index nothing */
        else {
            std::vector<float> px((size_t)S);
            ppp_slice_walk(h->P.walk, h->h_mn[0], h->h_mx[0], h->P.tool_radius, px.data(), S);
            /* Path_set is ascending in x; band of slice s = [int(px) - 2, int(px) + 2] (rangedX_index) */
            h->incl_lo = (float)((int)px[h->sb] - 2) - h->P.range_margin;
            h->incl_hi = (float)((int)px[h->se - 1] + 2) + h->P.range_margin;
            const double range = (double)h->h_mx[0] - (double)h->h_mn[0];
            const double part = std::min((double)h->incl_hi, (double)h->h_mx[0]) - std::max((double)h->incl_lo, (double)h->h_mn[0]);
            h->n_range = range > 0 ? (int)std::min((double)h->h_nvalid, std::max(0.0, part / range) * h->h_nvalid * 1.05 + 64) : h->h_nvalid;
        }
    }
    h->use_part = false; h->n_part = 0;
    if (h->part_given) {
        if (!h->ranged && !(h->part_lo == -INFINITY && h->part_hi == INFINITY))
            return fail(h, PPP_ERR_ARG, "a part-only cloud (ppp_set_cloud_part) needs a slice range: set slice_begin / slice_end");
        if (h->sb < h->se && !(h->incl_lo >= h->part_lo && h->incl_hi <= h->part_hi))
            return fail(h, PPP_ERR_ARG, "the slice range with its margin reaches beyond the part this handle was given (ppp_range_interval tells what it needs)");
        h->n_range = n;
    } else if (h->ranged && h->sb < h->se && n > 0) {
        /* the range's own points, once per plan: the hot path then streams n_part instead of n points (the bounds, the
           walk and the slab grid stay the whole cloud's: they come from the values cached with the cloud) */
        const int nblocks = (n + VOX_CHUNK - 1) / VOX_CHUNK;
        DevBuf<int> bcnt;
        DevBuf<VoxStats> st;
        hipError_t e = bcnt.ensure(nblocks);
        if (e == hipSuccess) e = st.ensure(1);
        if (e != hipSuccess) { bcnt.release(); st.release(); return fail(h, PPP_ERR_HIP, std::string("range part: ") + hipGetErrorString(e)); }
        VoxStats hst{0};
        auto run = [&]() -> int {
            LAUNCH(h, "k_part_count", k_part_count, nblocks, 256, 0, h->X.p, n, h->incl_lo, h->incl_hi, bcnt.p);
            LAUNCH(h, "k_vox_scan", k_vox_scan, 1, 1024, 0, bcnt.p, nblocks, st.p);
            HIPCHK(h, hipMemcpyAsync(&hst, st.p, sizeof(VoxStats), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            const size_t np = (size_t)std::max(hst.n_out, 1);
            HIPCHK(h, h->Xp.ensure(np)); HIPCHK(h, h->Yp.ensure(np)); HIPCHK(h, h->Zp.ensure(np)); HIPCHK(h, h->part_idx.ensure(np));
            LAUNCH(h, "k_part_compact", k_part_compact, nblocks, 256, 0, h->X.p, h->Y.p, h->Z.p, n, h->incl_lo, h->incl_hi, bcnt.p, h->Xp.p,
                   h->Yp.p, h->Zp.p, h->part_idx.p);
            HIPCHK(h, hipStreamSynchronize(h->stream));
            return PPP_OK;
        };
        const int rcp = run();
        bcnt.release(); st.release();
        if (rcp != PPP_OK) return rcp;
        h->n_part = hst.n_out;
        h->n_range = hst.n_out;
        h->use_part = true;
    }
    /* x-slabs: the histogram must fit LDS.  A slice-range handle
       keeps the WHOLE cloud's slab grid and just leaves the slabs outside its interval empty: its slabs then hold
       the same points in the same order as a whole-cloud handle's, so every sum over neighbours (normals) adds
       in the same order and the sharded list is bit-identical to the unsharded one. */
    /* 832 points per slab on average (measured: 640 .. 960 within 5 %, best here; from 1024 on the sort needs the
       larger LDS block and loses occupancy) -- a slab 2.4 times denser than the mean still sorts in LDS */
    const int SLAB_PTS = PPP_SLAB_PTS;
    int B = (h->h_nvalid + SLAB_PTS - 1) / SLAB_PTS;
    B = std::max(1, std::min(B, 8192));
    h->B = B;
    {
        double mean = (double)h->h_nvalid / B;
        /* LDS slots of a slab's sort workgroup: PPP_SLAB_CAP_FACTOR x the mean population in steps of 128 (12 B a slot: 1408
           slots = 16.5 KiB let eight 256-thread workgroups share a CU); a denser slab goes through the arena pass */
        h->slab_cap = (int)std::min(4096.0, std::max(1024.0, 128.0 * std::ceil(PPP_SLAB_CAP_FACTOR * mean / 128.0)));
    }
    HIPCHK(h, h->big_slabs.ensure(B));
    h->mm_grid = std::max(1, std::min(((h->use_part ? h->n_part : n) / 4 + 255) / 256, 2048)); /* 8 workgroups per CU keep enough loads in flight */
    HIPCHK(h, h->mm_part.ensure(h->mm_grid));
    h->S_cap = std::max(1, S);
    HIPCHK(h, h->big_slices.ensure(h->S_cap));
    /* band capacity: expected points in a 4 mm band, x2 margin, in [1024, 4096] */
    double range = (double)h->h_mx[0] - (double)h->h_mn[0];
    double expect = range > 0 ? (double)h->h_nvalid * 4.0 / range : (double)h->h_nvalid;
    /* (twice the mean band, in steps of 128 points: at 36 B a point every step decides how many slice workgroups share a CU) */
    int capb = (int)std::min(4096.0, std::max(1024.0, 128.0 * std::ceil(2.0 * expect / 128.0)));
    h->capb = capb;
    /* slabs or bands that will hardly fit LDS: run the arena passes from the start.  (Anything else that overflows turns them
       on by itself -- the pass re-runs once --; launching them for nothing costs two empty launches per pass, 13 us of cfg 5's
       600.) */
    if (1.2 * expect > 4096 || 1.2 * (double)h->h_nvalid / B > 4096) h->big_path = true;
    if (h->big_path) HIPCHK(h, h->arena.ensure((size_t)64 * (size_t)std::max(n, 1) + (1u << 20)));
    /* waypoints: every kept slice samples at most (yrange - 2 trim)/res + 1 points */
    double yr = (double)h->h_mx[1] - (double)h->h_mn[1];
    double per = std::max(0.0, (yr - 2 * h->P.trim)) / h->P.path_resolution + 2.0;
    {   /* k_pose: what a workgroup stages in LDS.  Knots: a slice has at most one per left band point (half the band, twice
           over).  Points: the slabs that overlap [Px - pad, Px + pad] span at most 2 pad + 2 slab widths; pad covers the
           nearest point (within a grid spacing of the plane) plus the normal's radius -- more when the dynamic adjustment
           moves knots off their plane.  Both are capacities of a fast path: what does not fit is read from global memory. */
        {   /* knots: about half the band's points are left points, at most one knot each -- 1.5 x that, a power of two */
            int kc = 256;
            while (kc < capb && kc < 0.75 * expect) kc <<= 1;
            h->knot_cap = kc;
        }
        h->pose_pad = h->P.dynamic_adjustment ? 8.f : std::max(5.f, 2.f * h->P.normal_radius);
        /* points: an interval of 2 pad overlaps at most floor(2 pad / w) + 2 slabs of width w; mean population + 5 %.  Every
           KiB counts: at 250 k points this is what lets three pose workgroups share a CU's LDS instead of two */
        const double slab_w = range > 0 ? range / B : 0.0;
        const double slabs = slab_w > 0 ? std::floor(2.0 * h->pose_pad / slab_w) + 2.0 : 1.0;
        const double want = 1.05 * slabs * ((double)h->h_nvalid / B) + 64;
        h->stage_cap = (int)std::min<double>(POSE_STAGE_CAP, std::max(512.0, 128.0 * std::ceil(want / 128.0)));
        h->tab_slabs = (int)std::min(16.0, slabs);
        /* all of it must fit beside the kernel's static LDS: knots give way first (a slice with more reads them from memory) */
        while (pose_lds_bytes(h->knot_cap, h->stage_cap, h->tab_slabs) > (size_t)h->max_lds - 8192 && h->knot_cap > 256) h->knot_cap >>= 1;
        while (pose_lds_bytes(h->knot_cap, h->stage_cap, h->tab_slabs) > (size_t)h->max_lds - 8192 && h->stage_cap > 512) h->stage_cap -= 128;
        h->cnt_est = (int)std::min(1.0e6, per);
        /* whole waves for every waypoint's lanes; the kernel is instantiated per 256 threads of budget */
        h->pose_threads = std::min(POSE_T, std::max(256, 64 * ((h->cnt_est * pose_lanes(h->cnt_est) + 63) / 64)));
    }
    double wc = per * (double)h->S_cap;
    if (wc > 2.0e8) return fail(h, PPP_ERR_CAPACITY, "waypoint bound too large");
    h->W_cap = std::max(1, (int)wc);
    /* nodes: one per left-side band point at most; bands may overlap when step < 5 */
    int step = (int)(h->P.tool_radius * 2);
    double overlap = step >= 5 ? 1.0 : 5.0 / std::max(1, step) + 1.0;
    h->node_cap = std::max(16, (int)std::min(2.0e9, overlap * (double)n + 16));

    HIPCHK(h, h->unsorted4.ensure(n)); HIPCHK(h, h->sorted4.ensure(n));
    const int *slab_cnt_before = h->slab_cnt.p;
    HIPCHK(h, h->slab_cnt.ensure(B)); HIPCHK(h, h->slab_start.ensure(B + 1)); HIPCHK(h, h->slab_cursor.ensure(B));
    HIPCHK(h, h->coarse_cursor.ensure((B >> SCAT_COARSE_SHIFT) + 2));
    HIPCHK(h, h->slab_ytab.ensure((size_t)B * (YTB + 1)));
    /* one scatter pass leaves runs of chunk / B points: below ~4 points per run the second pass pays for itself */
    h->two_pass_scatter = B >= 4096 && h->n_range >= 3000000;
    if (h->slab_cnt_used || h->slab_cnt.p != slab_cnt_before) { /* every run leaves it cleared again: cleared here after a slab-path pass (one that broke off may not have) and when new */
        HIPCHK(h, hipMemsetAsync(h->slab_cnt.p, 0, sizeof(int) * (size_t)B, h->stream));
        h->slab_cnt_used = false;
    }
    HIPCHK(h, h->slab_xmin.ensure(B)); HIPCHK(h, h->slab_xmax.ensure(B));
    HIPCHK(h, h->px.ensure(h->S_cap)); HIPCHK(h, h->lo.ensure(h->S_cap)); HIPCHK(h, h->hi.ensure(h->S_cap));
    if (h->P.dynamic_adjustment) {
        /* compute_boundary samples every Tool_Radius/4 between miny+2 and maxy-2; dynamic_adjust_path every ~5 mm */
        h->dyn_maxNB = (int)std::min(4.0e6, std::max(0.0, yr - 4) / (h->P.tool_radius / 4) + 4);
        h->dyn_maxNA = (int)std::min(4.0e6, yr / 5 + 4);
        h->node_cap = (int)std::min(2.0e9, (double)h->node_cap + (double)h->S_cap * h->dyn_maxNA);
        int rc = ensure_dynamic_buffers(h);
        if (rc) return rc;
    }
    HIPCHK(h, h->node_x.ensure(h->node_cap)); HIPCHK(h, h->node_y.ensure(h->node_cap)); HIPCHK(h, h->node_z.ensure(h->node_cap));
    HIPCHK(h, h->node_start.ensure(h->S_cap)); HIPCHK(h, h->node_cnt.ensure(h->S_cap)); HIPCHK(h, h->band_cnt.ensure(h->S_cap));
    HIPCHK(h, h->wp_cnt.ensure(h->S_cap)); HIPCHK(h, h->wp_off.ensure(h->S_cap + 1)); HIPCHK(h, h->tail.ensure(h->S_cap));
    HIPCHK(h, h->slice_wpcnt.ensure(h->S_cap));
    HIPCHK(h, h->wp_xyz.ensure(h->W_cap)); HIPCHK(h, h->wp_normal.ensure(h->W_cap)); HIPCHK(h, h->wp_nn.ensure(h->W_cap));
    HIPCHK(h, h->wp_pre.ensure(6 * (size_t)h->W_cap)); HIPCHK(h, h->wp_smooth.ensure(6 * (size_t)h->W_cap));
    HIPCHK(h, h->wp_out.ensure(6 * (size_t)h->W_cap));
    h->sm_tiles = smooth_tiles(h->W_cap);
    { int rcw = plan_window(h, S, per); if (rcw) return rcw; }
    h->stage_compact = true;
    h->planned = true;
    h->index_built = false; h->gen_done = false; h->meta_fresh = false; h->path_done = false; h->list_final = false;
    h->drop_graph(); /* buffer addresses and launch geometry are baked into the captured graph */
    return PPP_OK;
}

/* Threads of a k_slice_kd workgroup.  One workgroup per slice with the band in LDS: 1024 threads finish a slice soonest
   (one round of nearest-neighbour queries for bands of up to 2048 points), and that is what counts while the slices of a
   launch fit the GPU in one go.  With several times more slices than CUs (batches of workpieces) two 512-thread
   workgroups per CU get more slices through -- if two bands fit the CU's LDS. */
int slice_threads(const ppp_handle h, long long slices_in_launch)
{
    const bool two_fit = 2 * (slice_kd_bytes(h->capb) + 1024) <= (size_t)h->max_lds;
    return (two_fit && slices_in_launch >= 2LL * h->num_cus) ? 512 : SLICE_KD_T;
}

/* a2 + a3 + the x-slab index (generalised slice binning) */
int enqueue_index(ppp_handle h)
{
    /* a slice-range handle streams its own part of the cloud (make_plan), everything else the whole cloud */
    const int n = h->use_part ? h->n_part : (int)h->n;
    const float *sX = h->use_part ? h->Xp.p : h->X.p, *sY = h->use_part ? h->Yp.p : h->Y.p, *sZ = h->use_part ? h->Zp.p : h->Z.p;
    const int *idmap = h->use_part ? h->part_idx.p : ((h->part_given && h->part_has_idx) ? h->part_idx.p : nullptr);
    DevParams D = dev_params(h);
    h->slab_cnt_used = true;
    D.keep_run_state = (h->win_path && h->gen_done) ? 1 : 0; /* an API mirror asks for the slab index behind a finished window pass */
    size_t hist_lds = sizeof(int) * (size_t)h->B;
    /* slab grid from the bounds cached when the cloud was set (identical to what k_minmax finds) */
    const float slab_x0 = h->h_mn[0];
    const float xr = h->h_mx[0] - h->h_mn[0];
    const float slab_invw = (h->h_nvalid && xr > 0.f) ? (float)h->B / xr : 0.f;
    {   /* a2 and the slab histogram in ONE pass over the cloud (the slab grid comes from the bounds cached with the
           cloud).  At most PPP_MM_GRID_MAX workgroups: each flushes its LDS histogram with one global atomic per non-empty slab, and
           that flush, not the streaming, is what grows with the grid. */
        const int gf = std::max(1, std::min(h->mm_grid, PPP_MM_GRID_MAX));
        LAUNCH(h, "k_minmax", k_minmax<true>, gf, MM_T, hist_lds, sX, sY, sZ, n, h->mm_part.p, slab_x0, slab_invw, h->B, h->slab_cnt.p,
               h->incl_lo, h->incl_hi, h->slab_cursor.p);
        h->mm_grid_used = gf;
    }
    /* points per scatter workgroup: every workgroup reserves its share of each slab with one global atomic per
       non-empty (workgroup, slab) pair, so larger clouds use 8 instead of 4 points per thread */
    const bool ppt8 = n > PPP_PPT8_FROM;
    const bool ppt16 = n > PPP_PPT16_FROM && !h->two_pass_scatter;
    const int chunk = (ppt16 ? 16 : (ppt8 ? 8 : 4)) * SCAT_T;
    const int gs = std::max(1, (n + chunk - 1) / chunk);
    if (h->two_pass_scatter)
        LAUNCH(h, "k_setup", k_setup, 1, SETUP_T, 0, h->meta.p, D, h->mm_part.p, h->mm_grid_used, h->px.p, h->lo.p, h->hi.p, h->S_cap, h->B,
               h->slab_cnt.p, slab_x0, slab_invw, h->slab_start.p, h->slab_cursor.p, h->coarse_cursor.p);
    if (!h->two_pass_scatter) {
        /* one level: the set-up rides in the scatter's launch as its last workgroup (k_scatter_setup) */
        ScatGrid G;
        G.x0 = slab_x0; G.invw = slab_invw; G.xlo = h->incl_lo; G.xhi = h->incl_hi; G.B = h->B;
        if (ppt16)
            LAUNCH(h, "k_scatter_setup", k_scatter_setup<16>, gs + 1, SCAT_T, 2 * hist_lds, sX, sY, sZ, n, G, h->slab_cnt.p, h->slab_cursor.p,
                   h->unsorted4.p, idmap, gs, h->meta.p, D, h->mm_part.p, h->mm_grid_used, h->px.p, h->lo.p, h->hi.p, h->S_cap, h->slab_start.p);
        else if (ppt8)
            LAUNCH(h, "k_scatter_setup", k_scatter_setup<8>, gs + 1, SCAT_T, 2 * hist_lds, sX, sY, sZ, n, G, h->slab_cnt.p, h->slab_cursor.p,
                   h->unsorted4.p, idmap, gs, h->meta.p, D, h->mm_part.p, h->mm_grid_used, h->px.p, h->lo.p, h->hi.p, h->S_cap, h->slab_start.p);
        else
            LAUNCH(h, "k_scatter_setup", k_scatter_setup<4>, gs + 1, SCAT_T, 2 * hist_lds, sX, sY, sZ, n, G, h->slab_cnt.p, h->slab_cursor.p,
                   h->unsorted4.p, idmap, gs, h->meta.p, D, h->mm_part.p, h->mm_grid_used, h->px.p, h->lo.p, h->hi.p, h->S_cap, h->slab_start.p);
    } else {
        /* coarse bins into sorted4 (free until k_slab_sort writes it), then from there into the slabs */
        LAUNCH(h, "k_slab_scatter", (k_slab_scatter<1, 8>), gs, SCAT_T, hist_lds, sX, sY, sZ, (const float4 *)nullptr, n, h->meta.p,
               h->coarse_cursor.p, h->sorted4.p, idmap);
        const int gs2 = std::max(1, (n + 4 * SCAT_T - 1) / (4 * SCAT_T));
        LAUNCH(h, "k_slab_scatter2", (k_slab_scatter<2, 4>), gs2, SCAT_T, hist_lds, sX, sY, sZ, (const float4 *)h->sorted4.p, n,
               h->meta.p, h->slab_cursor.p, h->unsorted4.p, (const int *)nullptr);
    }
    size_t sort_lds = (size_t)h->slab_cap * 12 + 16;
    /* threads per slab: 256 while a slab holds the planned 832 points on average (more slabs in flight per CU: cfg 2 sorts in
       15.3 us against 17.0), SORT_T for the fuller slabs of clouds beyond the 8192-slab cap (cfg 5: 125 us against 157) */
    const int sort_threads = (h->B > 0 && h->h_nvalid / h->B > 1000) ? SORT_T : 256;
    /* a slice-range handle sorts the slabs of its interval only (the others are empty and are never looked at) */
    int first_slab = 0, nslabs = h->B;
    if (h->use_part && slab_invw > 0.f) {
        auto slab_of_host = [&](float x) { int b = (int)((x - slab_x0) * slab_invw); b = b < 0 ? 0 : b; return b >= h->B ? h->B - 1 : b; };
        first_slab = slab_of_host(h->incl_lo);
        nslabs = slab_of_host(h->incl_hi) - first_slab + 1;
    }
    LAUNCH(h, "k_slab_sort", k_slab_sort<false>, nslabs, sort_threads, sort_lds, h->unsorted4.p, h->slab_start.p, h->sorted4.p,
           h->slab_xmin.p, h->slab_xmax.p, h->meta.p, h->slab_cap, h->big_slabs.p, h->arena.p, (unsigned long long)h->arena.cap, h->slab_ytab.p,
           first_slab, h->slab_cnt.p);
    if (h->big_path)
        LAUNCH(h, "k_slab_sort_arena", k_slab_sort<true>, h->B, SORT_T, 0, h->unsorted4.p, h->slab_start.p, h->sorted4.p,
               h->slab_xmin.p, h->slab_xmax.p, h->meta.p, h->slab_cap, h->big_slabs.p, h->arena.p, (unsigned long long)h->arena.cap, h->slab_ytab.p, 0, (int *)nullptr);
    h->index_built = true;
    return PPP_OK;
}

/* centre slice of the centre-out walk = number of slices left of it (path_dynamic_alg.cpp:310-313) */
int host_centre_index(const ppp_handle h)
{
    const int step = (int)(h->P.tool_radius * 2);
    const int imin = (int)h->h_mn[0], imax = (int)h->h_mx[0];
    const int c = (imax + imin) / 2;
    return (step > 0 && imax > c - step && c - step > imin) ? (c - imin - 1) / step : 0;
}

/* GenPath with Adjust = true: whole-cloud normals, then the slice-to-slice chains */
int enqueue_dynamic(ppp_handle h)
{
    if (h->dyn_maxNB > 4096 || h->dyn_maxNA > 4096)
        return fail(h, PPP_ERR_CAPACITY, "dynamic adjustment: more than 4096 boundary or path samples per slice");
    int rc = enqueue_normals(h);
    if (rc) return rc;
    const DynParams D = dyn_params(h);
    DynBuffers Bf{h->dyn_bnd_pts.p, h->dyn_bnd_knots.p, h->dyn_bnd_n.p, h->dyn_bnd_n.p + 2, h->dyn_bnd_n.p + 4, h->dyn_keep_all ? 1 : 0,
                  h->dyn_adj_pts.p, h->dyn_maxNB, h->dyn_maxNA, h->dyn_first_ab.p, h->dyn_first_node.p, h->dyn_first_snap.p};
    HIPCHK(h, hipMemsetAsync(h->dyn_bnd_n.p, 0, (4 + (size_t)std::max(h->S_cap, 2)) * sizeof(int), h->stream));
    const int S = h->S_cap, walk = h->P.walk;
    const int centre = walk == PPP_WALK_CENTER_INT ? host_centre_index(h) : 0;
    const int nchains = walk == PPP_WALK_CENTER_INT ? 2 : 1;
    const int steps = walk == PPP_WALK_CENTER_INT ? std::max(centre, S - 1 - centre) : S - 1;
    const dim3 gb((Bf.maxNB + DYN_WAVES - 1) / DYN_WAVES + 1, nchains), ga((Bf.maxNA + DYN_WAVES - 1) / DYN_WAVES, nchains);
    const size_t lds_b = dyn_boundary_pts_lds(Bf.maxNA), lds_a = dyn_adjust_pts_lds(Bf.maxNB);
    /* every node's first Area2Cloud and snap, for all slices at once: they do not depend on the chains */
    LAUNCH(h, "k_dyn_first_eval", k_dyn_first_eval, dim3(ga.x, S), 64 * DYN_WAVES, 0, h->meta.p, D, walk, centre, h->sorted4.p,
           h->slab_start.p, h->slab_xmin.p, h->slab_xmax.p, h->normals4.p, h->ell_cs.p, h->slab_ytab.p, h->node_x.p, h->node_y.p,
           h->node_z.p, h->node_start.p, h->node_cnt.p, Bf);
    /* two launches per step: each begins with the fit of what the launch before it sampled */
    for (int t = 0; t < steps; ++t) {
        LAUNCH(h, "k_dyn_boundary_pts", k_dyn_boundary_pts, gb, 64 * DYN_WAVES, lds_b, h->meta.p, D, walk, t, centre, h->sorted4.p,
               h->slab_start.p, h->slab_xmin.p, h->slab_xmax.p, h->normals4.p, h->ell_cs.p, h->slab_ytab.p, h->node_x.p, h->node_y.p,
               h->node_z.p, h->node_cap, h->node_start.p, h->node_cnt.p, Bf);
        LAUNCH(h, "k_dyn_adjust_pts", k_dyn_adjust_pts, ga, 64 * DYN_WAVES, lds_a, h->meta.p, D, walk, t, centre, h->sorted4.p,
               h->slab_start.p, h->slab_xmin.p, h->slab_xmax.p, h->normals4.p, h->ell_cs.p, h->slab_ytab.p, S, Bf);
    }
    if (steps > 0)
        LAUNCH(h, "k_dyn_adjust_fit", k_dyn_adjust_fit, nchains, 256, dyn_scratch_bytes(Bf.maxNA), h->meta.p, walk, steps - 1, centre, Bf,
               h->node_x.p, h->node_y.p, h->node_z.p, h->node_cap, h->node_start.p, h->node_cnt.p);
    return PPP_OK;
}

/* work enqueued for this handle by a batch graph runs on the lead handle's stream */
int settle_streams(ppp_handle h)
{
    if (h->pending_stream) {
        HIPCHK(h, hipStreamSynchronize(h->pending_stream));
        h->pending_stream = nullptr;
    }
    return PPP_OK;
}
int resolve_deferred(ppp_handle h);
/* every entry point but the three that only enqueue a pass: nothing pending on another stream, and the plan is this cloud's */
int settle(ppp_handle h)
{
    int rc = settle_streams(h);
    if (rc == PPP_OK && h->plan_deferred) rc = resolve_deferred(h);
    return rc;
}
/* ppp_run_async / ppp_gen_path_async / ppp_get_path_async: a pass may be enqueued on the plan a cloud was set under (window
   path, same size and parameters) before that cloud's bounds have come back */
int settle_enqueue_only(ppp_handle h)
{
    if (h->plan_deferred && h->planned && h->win_path) return settle_streams(h);
    return settle(h);
}

int fetch_meta(ppp_handle h)
{
    { int rc = settle(h); if (rc) return rc; }
    if (h->meta_in_flight) { /* GenPath / getPath already enqueued the copy behind their last kernel */
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->hmeta = (h->meta_from_batch && h->bmetas) ? h->bmetas->pinned[h->bslot] : *h->hmeta_pinned;
        h->meta_in_flight = false;
        h->meta_fresh = true;
        return PPP_OK;
    }
    /* (a planner class asks per slice -- a Spline view per slice, two questions each: a copy and a wait per question were
       11 ms of GenPath() at 256 slices) */
    if (h->meta_fresh) return PPP_OK;
    HIPCHK(h, hipMemcpyAsync(&h->hmeta, h->meta.p, sizeof(DevMeta), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->meta_fresh = true;
    return PPP_OK;
}

int enqueue_meta_copy(ppp_handle h)
{
    HIPCHK(h, hipMemcpyAsync(h->hmeta_pinned, h->meta.p, sizeof(DevMeta), hipMemcpyDeviceToHost, h->stream));
    h->meta_in_flight = true;
    h->meta_from_batch = false;
    return PPP_OK;
}

int map_dev_err(ppp_handle h)
{
    if (!h->big_path && (h->hmeta.big_slabs > 0 || h->hmeta.big_slices > 0))
        return fail(h, PPP_ERR_CAPACITY, "a slab or band exceeds the LDS capacity: re-run (the arena passes are now enabled)");
    if (h->win_path && h->hmeta.win_flag)
        return fail(h, PPP_ERR_CAPACITY, "the window path handed this pass back (window overflow / reach / stale plan): re-run (the slab-index path is now selected)");
    switch (h->hmeta.err) {
    case DERR_NONE: return PPP_OK;
    case DERR_SLICE: {
        char buf[160];
        snprintf(buf, sizeof(buf), "slice %d has an empty side or fewer than 3 spline nodes (the reference aborts here)", h->hmeta.err_slice);
        return fail(h, PPP_ERR_SLICE, buf);
    }
    case DERR_CAPACITY: return fail(h, PPP_ERR_CAPACITY, "device capacity exceeded (slab / band / node / waypoint buffer)");
    case DERR_DOMAIN: return fail(h, PPP_ERR_DOMAIN, "spline evaluated outside its knots");
    case DERR_QUERY: return fail(h, PPP_ERR_DOMAIN, "non-finite waypoint handed to the nearest-neighbour query");
    case DERR_MARGIN: return fail(h, PPP_ERR_CAPACITY, "a waypoint's nearest neighbour or its normal neighbourhood reaches beyond the indexed slice range: raise range_margin");
    }
    return fail(h, PPP_ERR_HIP, "unknown device error");
}

/* An overflow of the LDS-resident fast path is not an error of the input: turn the arena passes on
   and run the same calls again, once. */
int rerun_with_arena(ppp_handle h)
{
    const bool had_path = h->path_done;
    if (h->win_path && h->hmeta.win_flag && getenv("PPP_WIN_DEBUG"))
        fprintf(stderr, "[ppp] window pass handed back: flags %d (1 overflow, 2 reach, 4 stale plan)%s\n", h->hmeta.win_flag, h->plan_inherited ? ", capacities were inherited" : "");
    if (h->win_path && h->hmeta.win_flag && h->plan_inherited && !(h->hmeta.win_flag & ~WIN_FLAG_OVERFLOW)) {
        /* capacities inherited from an earlier cloud did not hold for this one: plan it from a census of its own, on the window path */
        h->inh_valid = false;
        int rcp = make_plan(h);
        if (rcp) return rcp;
    } else if (h->win_path && h->hmeta.win_flag) {
        /* the window path's capacities or reach did not hold for this cloud: the same cloud and parameters on the slab index
           from now on (a new cloud or new parameters try the window path again) */
        h->win_disabled = true;
        int rcp = make_plan(h);
        if (rcp) return rcp;
    } else {
        h->big_path = true;
        h->drop_graph();
        HIPCHK(h, h->arena.ensure((size_t)64 * (size_t)std::max<size_t>(h->n, 1) + (1u << 20)));
    }
    ++h->internal;
    int rc = ppp_gen_path_async(h);
    /* a member of a batch: the re-planned list must land where the batch put the first one (the caller's gather buffer) */
    h->out2 = h->last_out2; h->out2_cap = h->last_out2_cap;
    if (rc == PPP_OK && had_path) rc = ppp_get_path_async(h);
    h->out2 = nullptr; h->out2_cap = 0;
    --h->internal;
    if (rc) return rc;
    return fetch_meta(h);
}

bool overflowed_fast_path(ppp_handle h)
{
    /* (a GenPath on its own leaves no error behind: the slices parked for the arena pass are simply not planned yet) */
    if (h->win_path && h->hmeta.win_flag) return true; /* the window path handed the pass back */
    return !h->big_path && (h->hmeta.big_slabs > 0 || h->hmeta.big_slices > 0);
}

int ensure_ready(ppp_handle h, bool need_gen, bool need_path)
{
    if (!h) return PPP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->have_cloud) return fail(h, PPP_ERR_ARG, "no cloud set");
    if (need_gen && !h->gen_done) return fail(h, PPP_ERR_ARG, "call ppp_gen_path_async first");
    if (need_path && !h->path_done) return fail(h, PPP_ERR_ARG, "call ppp_get_path_async first");
    int rc = fetch_meta(h);
    if (rc) return rc;
    /* a pass handed back by the window path runs on the slab index, whose LDS fast path may overflow in turn (arena passes) */
    for (int tries = 0; tries < 2 && overflowed_fast_path(h); ++tries) { rc = rerun_with_arena(h); if (rc) return rc; }
    return PPP_OK;
}

int ensure_index(ppp_handle h)
{
    if (!h) return PPP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->have_cloud) return fail(h, PPP_ERR_ARG, "no cloud set");
    { int rcs = settle(h); if (rcs) return rcs; }
    if (!h->planned) { int rc = make_plan(h); if (rc) return rc; }
    if (!h->index_built) { int rc = enqueue_index(h); if (rc) return rc; }
    return PPP_OK;
}

int slice_lds_ok(ppp_handle h, int capb)
{
    return std::max(slice_lds_bytes(capb), slice_kd_bytes(capb)) + 1024 <= (size_t)h->max_lds;
}

/* workgroups of the bounds pass over n points (k_minmax<false> / k_ingest_minmax) */
int bounds_grid(size_t n) { return std::max(1, std::min(((int)n / 4 + 255) / 256, 2048)); }
/* ... of the conversion pass of a new cloud (k_ingest_minmax): about one workgroup per CU, eight points per thread.  Its workgroups
   end on ONE ticket counter, and same-address atomics serialise at ~11 ns each: the 977 workgroups bounds_grid gives a million
   points spent 10 us of a 23 us launch queueing there */
int ingest_grid(size_t n) { return std::max(1, std::min(((int)n / 8 + MM_T - 1) / MM_T, 256)); }

/* may the window path apply to the next plan, as far as the parameters say (plan_window decides with the bounds in hand)? */
bool window_params_ok(const ppp_handle h)
{
    return h->win_allowed && !h->win_disabled && !getenv("PPP_NO_WINDOW_PATH") && !h->P.dynamic_adjustment &&
           !h->aligned && !h->big_path && (int)(h->P.tool_radius * 2) >= 1;
}

/* cache the bounds of the resident cloud for the plan (sizing only; the hot path recomputes them on device), then plan.
   raw != nullptr: the cloud has just arrived and is converted in the same pass (k_ingest_minmax); the bounds come back reduced, in
   pinned memory, and where the window path may apply its census follows in the same stream: two launches, one wait, no copy or
   fill command (each of those costs the host 10-20 us here; this path was 150 us for 30 us of kernels) */
/* what k_ingest_minmax (and the census behind it) left in pinned memory for the host: bounds, count, the device's walk */
int adopt_ingest_record(ppp_handle h, bool census, bool reuse)
{
    const PlanAuto *rec0 = (const PlanAuto *)(h->pin + PIN_REC0), *rec1 = (const PlanAuto *)(h->pin + PIN_REC1);
    if (rec0->S == -2) return fail(h, PPP_ERR_HIP, "the bounds of the new cloud did not arrive");
    h->h_nvalid = rec0->fin.cnt;
    for (int d = 0; d < 3; ++d) { h->h_mn[d] = rec0->fin.mn[d]; h->h_mx[d] = rec0->fin.mx[d]; }
    if (census && !reuse && rec1->census == 1) { h->auto_valid = true; h->auto_S = rec1->S; h->auto_pad = rec1->pad; }
    if (reuse && rec0->S >= 1 && rec0->S <= WIN_AUTO_SCAP) { h->auto_px_only = true; h->auto_S = rec0->S; h->auto_pad = rec0->pad; }
    return PPP_OK;
}

int refresh_bounds_and_plan(ppp_handle h, const char *raw = nullptr, size_t stride_bytes = 0, bool may_defer = false)
{
    const size_t n = h->n;
    h->auto_valid = false; h->auto_px_only = false;
    h->win_disabled = false;
    h->big_path = false; /* (the plan turns the arena passes on again where this cloud needs them) */
    {
        const int g = bounds_grid(n);
        HIPCHK(h, h->mm_part.ensure(g));
        (void)hipGetLastError();
        if (raw) {
            HIPCHK(h, h->ensure_pin(PIN_AUTO_BYTES));
            if (!h->plan_ticket.p) {
                HIPCHK(h, h->plan_ticket.ensure(2)); HIPCHK(h, h->plan_auto.ensure(1));
                HIPCHK(h, hipMemsetAsync(h->plan_ticket.p, 0, 2 * sizeof(int), h->stream));
            }
            const bool census = window_params_ok(h) && !h->part_given && (h->P.walk >= 0 && h->P.walk <= 4);
            /* capacities of an earlier cloud of this size and these parameters: no census launch (plan_window decides with the walk in hand) */
            const bool reuse = census && h->plan_reuse && h->inh_valid && h->inh_n == (int)n && memcmp(&h->inh_P, &h->P, sizeof(ppp_params)) == 0;
            if (census && (h->win_px.cap < WIN_AUTO_SCAP || h->win_cnt.cap < 3 * (size_t)WIN_AUTO_SCAP)) {
                HIPCHK(h, h->win_px.ensure(WIN_AUTO_SCAP)); HIPCHK(h, h->win_cnt.ensure(3 * (size_t)WIN_AUTO_SCAP));
                HIPCHK(h, hipMemsetAsync(h->win_cnt.p, 0, sizeof(int) * 3 * (size_t)WIN_AUTO_SCAP, h->stream)); /* every census and every pass leaves them cleared */
            }
            PlanAutoArgs PA;
            PA.ticket = h->plan_ticket.p; PA.dev = h->plan_auto.p; PA.host = (PlanAuto *)(h->pin + PIN_REC0);
            PA.walk = census ? h->P.walk : -1; PA.tool_radius = h->P.tool_radius; PA.normal_radius = h->P.normal_radius;
            PA.px = h->win_px.p; PA.px_cap = census ? WIN_AUTO_SCAP : 0;
            PA.px_host = reuse ? (float *)(h->pin + PIN_PX) : nullptr;
            PlanAuto *rec0 = (PlanAuto *)(h->pin + PIN_REC0), *rec1 = (PlanAuto *)(h->pin + PIN_REC1);
            rec0->S = -2; rec1->census = 0; rec1->S = -2;
            hipLaunchKernelGGL(k_ingest_minmax, dim3(ingest_grid(n)), dim3(MM_T), 0, h->stream, raw, stride_bytes, (int)n, h->P.change_range, h->X.p, h->Y.p,
                               h->Z.p, h->mm_part.p, PA);
            HIPCHK(h, hipGetLastError());
            if (census && !reuse) {
                const int gc = std::max(1, std::min(((int)n + 4095) / 4096, 512));
                hipLaunchKernelGGL(k_win_census_auto, dim3(gc), dim3(256), sizeof(int) * 3 * (size_t)WIN_AUTO_SCAP, h->stream, h->X.p, (int)n,
                                   h->win_px.p, h->plan_auto.p, 1.0f / (float)(int)(h->P.tool_radius * 2), h->win_cnt.p, h->plan_ticket.p + 1, rec1,
                                   (float *)(h->pin + PIN_PX), (int *)(h->pin + PIN_CENSUS));
                HIPCHK(h, hipGetLastError());
            }
            h->rec_current = true;
            /* The caller lets go of the raw points only when it is told to (ppp_set_cloud_device_async, or they are the handle's own:
               ppp_set_cloud_pcd), and the handle's plan is a window plan for a cloud of this size and these parameters, made from
               the device's own walk: no wait.  The plan stays and a pass of the new cloud may follow the conversion pass in the stream at once
               (settle_enqueue_only); walk length, pad, bounds and every capacity are checked on the device against the record
               this launch leaves (win_verify_body), and the first call that needs the host's view of the cloud reads it
               (resolve_deferred). */
            if (may_defer && reuse && h->planned && h->win_path && h->plan_walk_ok && !h->ranged && !h->use_part && h->sb == 0 && h->se == h->S_cap &&
                h->inh_S == h->S_cap && !getenv("PPP_NO_DEFERRED_PLAN")) {
                h->plan_deferred = true; h->deferred_census = census;
                h->have_cloud = true;
                h->index_built = false; h->gen_done = false; h->meta_fresh = false; h->path_done = false; h->list_final = false;
                h->normals_valid = false;
                return PPP_OK;
            }
#ifdef PPP_TUNING
            const auto t_enq = std::chrono::steady_clock::now();
#endif
            HIPCHK(h, hipStreamSynchronize(h->stream));
#ifdef PPP_TUNING
            if (getenv("PPP_COLD_DEBUG")) {
                const auto t_syn = std::chrono::steady_clock::now();
                fprintf(stderr, "[ppp cold] wait for ingest%s: %.1f us\n", census ? " + census" : "", std::chrono::duration<double, std::micro>(t_syn - t_enq).count());
            }
#endif
            { int rca = adopt_ingest_record(h, census, reuse); if (rca) return rca; }
        } else {
            h->rec_current = false; /* (the cloud was altered on the device: the conversion pass's record is another cloud's) */
            hipLaunchKernelGGL(k_minmax<false>, dim3(g), dim3(MM_T), 0, h->stream, h->X.p, h->Y.p, h->Z.p, (int)n, h->mm_part.p, 0.f, 0.f, 0,
                               (int *)nullptr, 0.f, 0.f, (int *)nullptr);
            HIPCHK(h, hipGetLastError());
            /* (through pinned memory: a copy into pageable memory is staged by the runtime, ~10 us more on this critical path) */
            HIPCHK(h, h->ensure_pin(sizeof(MinMaxPart) * (size_t)g));
            MinMaxPart *parts = (MinMaxPart *)h->pin;
            HIPCHK(h, hipMemcpyAsync(parts, h->mm_part.p, sizeof(MinMaxPart) * g, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            h->h_nvalid = 0;
            for (int d = 0; d < 3; ++d) { h->h_mn[d] = INFINITY; h->h_mx[d] = -INFINITY; }
            for (int q = 0; q < g; ++q) {
                const MinMaxPart &r = parts[q];
                h->h_nvalid += r.cnt;
                for (int d = 0; d < 3; ++d) { h->h_mn[d] = std::min(h->h_mn[d], r.mn[d]); h->h_mx[d] = std::max(h->h_mx[d], r.mx[d]); }
            }
        }
        if (!h->h_nvalid) for (int d = 0; d < 3; ++d) { h->h_mn[d] = 3.402823466e+38f; h->h_mx[d] = -3.402823466e+38f; }
    }
    h->have_cloud = true;
    h->planned = false; h->index_built = false; h->gen_done = false; h->meta_fresh = false; h->path_done = false;
    h->normals_valid = false;
#ifdef PPP_TUNING
    if (getenv("PPP_COLD_DEBUG")) {
        const auto t_a = std::chrono::steady_clock::now();
        const int rcp = make_plan(h);
        fprintf(stderr, "[ppp cold] make_plan: %.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_a).count());
        return rcp;
    }
#endif
    return make_plan(h);
}

/* The cloud was set on a plan of an earlier cloud (refresh_bounds_and_plan): now the host's view of it.  Waits for the stream
   (the conversion pass and whatever pass was enqueued behind it), takes the record, and plans the cloud exactly as a waiting
   ppp_set_cloud* would have.  A pass that already ran keeps its results when that plan asks for the launches and buffers the
   pass used (the rule for a cloud of the same kind: capacities are inherited either way); else it runs again on the right plan.
   What the device found wrong with the pass itself (WIN_FLAG_*) is in its meta block and handled where every pass's is. */
int resolve_deferred(ppp_handle h)
{
    if (!h->plan_deferred) return PPP_OK;
    h->plan_deferred = false;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const bool ran = h->gen_done, had_path = h->path_done, was_final = h->list_final;
    auto signature = [](ppp_handle q) {
        WinArgs A = win_args(q);
        memset(&A.P, 0, sizeof(A.P)); /* the parameters are the same by the rule of refresh_bounds_and_plan; what of them follows the bounds is a hint */
        memcpy(A.P.viewpoint, q->vp, sizeof(q->vp)); /* (the cloud's own: it decides the normals' sign) */
        A.y0 = A.yscale = A.px0 = 0.f; /* bucket mapping and lattice origin: any monotone mapping sorts alike, the origin is read from the table */
        for (int d = 0; d < 3; ++d) A.plan_mn[d] = A.plan_mx[d] = 0.f;
        A.plan_nvalid = 0;             /* (compared on the device against the record) */
        A.out2 = nullptr; A.out2_cap = 0;
        return A;
    };
    const WinArgs before = signature(h);
    const bool was_window = h->win_path;
    /* (a captured pass does not outlive this: replaying it for the next cloud was measured against enqueueing that cloud's three
       launches one by one while its conversion pass runs -- 96 against 88 us for a never-seen cloud, 54 against 50 us per cloud
       through two lanes of the planner queue) */
    { int rca = adopt_ingest_record(h, h->deferred_census, true); if (rca) return rca; }
    int rc = make_plan(h);
    if (rc) return rc;
    const WinArgs after = signature(h);
    const bool same = was_window && h->win_path && memcmp(&before, &after, sizeof(WinArgs)) == 0;
    if (!ran) return PPP_OK;
    if (same) {
        h->gen_done = true; h->path_done = had_path; h->list_final = was_final; /* (make_plan withdrew them) */
        return PPP_OK;
    }
    if (getenv("PPP_WIN_DEBUG")) fprintf(stderr, "[ppp] the pass enqueued ahead of this cloud's bounds ran on another plan than the cloud's own: repeated\n");
    ++h->internal;
    rc = ppp_gen_path_async(h);
    h->out2 = h->last_out2; h->out2_cap = h->last_out2_cap; /* (the caller's output buffer of the first attempt) */
    if (rc == PPP_OK && had_path) rc = ppp_get_path_async(h);
    h->out2 = nullptr; h->out2_cap = 0;
    --h->internal;
    return rc;
}

int set_cloud_common(ppp_handle h, const char *raw_dev, size_t n, size_t stride_bytes, const float *viewpoint, bool may_defer = false)
{
    if (n > 0x7fffffffu / 8) return fail(h, PPP_ERR_CAPACITY, "cloud too large");
    h->n = n;
    h->part_given = false; h->part_has_idx = false;
    h->aligned = false; /* a new cloud: TransAlign = identity (path_slicing_alg.cpp:25) */
    if (h->back) { delete h->back; h->back = nullptr; }
    {
        float vp_new[3] = {0.f, 0.f, 0.f};
        if (viewpoint) memcpy(vp_new, viewpoint, 12);
        /* the viewpoint travels by value in the launches' arguments: a captured pass of the earlier cloud must not be replayed for
           this one (a plan made ahead of the bounds keeps its graph otherwise) */
        if (memcmp(vp_new, h->vp, 12) != 0) h->drop_graph();
        memcpy(h->vp, vp_new, 12);
    }
    HIPCHK(h, h->X.ensure(n)); HIPCHK(h, h->Y.ensure(n)); HIPCHK(h, h->Z.ensure(n));
    return n ? refresh_bounds_and_plan(h, raw_dev, stride_bytes, may_defer) : refresh_bounds_and_plan(h);
}


} // namespace

extern "C" {

void ppp_default_params(ppp_params *p)
{   /* config.txt:1-13, Path_Generate_Algorithm.h:43-48 */
    memset(p, 0, sizeof(*p));
    p->tool_radius = 12; p->path_resolution = 7; p->rpy_resolution = 7; p->ee_length = 0.3f;
    p->change_range = 1; p->pairing = PPP_PAIR_KD; p->walk = PPP_WALK_CENTER_INT;
    p->trim = 10; p->drop_ends = 1; p->smooth = 1;
    const float he[6] = {-0.764091f, 0.025886f, 0.663790f, -3.1270175f, -0.040124f, -1.6063578f};
    memcpy(p->handeye, he, sizeof(he));
    p->normal_radius = 2.5f;
    p->smooth_max_sweeps = 32;
    p->alignment = 0; p->dynamic_adjustment = 0; /* config.txt says true: the planner classes pass it through */
    p->depth = 0.01; p->adjust_threshold = 1; p->toolthickness = 10; p->curvature_k = 50;
    p->slice_begin = 0; p->slice_end = 0; p->range_margin = 24.f;
}

const char *ppp_version(void) { return PPP_VERSION_STR; }

int ppp_create(int device_id, ppp_handle *out)
{
    if (!out) return PPP_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return PPP_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= count) return PPP_ERR_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return PPP_ERR_HIP;
    ppp_handle h = new ppp_handle_s();
    h->device = device_id;
    ppp_default_params(&h->P);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return PPP_ERR_HIP; }
    if (h->meta.ensure(1) != hipSuccess) { delete h; return PPP_ERR_HIP; }
    /* a handle that finishes a gathered list before it ever ran a pass (ppp_finish_path_async) reads win_flag, S and the bounds
       from this block: hipMalloc does not clear it */
    if (hipMemset(h->meta.p, 0, sizeof(DevMeta)) != hipSuccess) { delete h; return PPP_ERR_HIP; }
    if (h->fin_ticket.ensure(1 + WIN_FIN_GROUPS) != hipSuccess || hipMemset(h->fin_ticket.p, 0, sizeof(int) * (1 + WIN_FIN_GROUPS)) != hipSuccess) { delete h; return PPP_ERR_HIP; }
    if (hipHostMalloc((void **)&h->hmeta_pinned, sizeof(DevMeta), hipHostMallocDefault) != hipSuccess) { delete h; return PPP_ERR_HIP; }
    int lds = 0;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id) == hipSuccess && lds > 0) h->max_lds = lds;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
        if (strncmp(prop.gcnArchName, "gfx950", 6) == 0) h->max_lds = std::max(h->max_lds, 160 * 1024); /* CDNA4: one workgroup may own the CU's whole LDS */
        if (prop.multiProcessorCount > 0) h->num_cus = prop.multiProcessorCount;
    }
    /* kernels with > 64 KiB of dynamic LDS opt in explicitly */
    (void)hipFuncSetAttribute((const void *)k_slice, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 1024);
    (void)hipFuncSetAttribute((const void *)k_slice_kd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 1024);
    (void)hipFuncSetAttribute((const void *)k_minmax<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
    (void)hipFuncSetAttribute((const void *)k_slab_scatter<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
    (void)hipFuncSetAttribute((const void *)k_slab_scatter<0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
    (void)hipFuncSetAttribute((const void *)k_slab_scatter<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
    (void)hipFuncSetAttribute((const void *)k_slab_scatter<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
    (void)hipFuncSetAttribute((const void *)k_minmax_b, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
    (void)hipFuncSetAttribute((const void *)k_slab_scatter_b<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
    (void)hipFuncSetAttribute((const void *)k_slab_scatter_b<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
    (void)hipFuncSetAttribute((const void *)k_scatter_setup<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
    (void)hipFuncSetAttribute((const void *)k_scatter_setup<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
    (void)hipFuncSetAttribute((const void *)k_scatter_setup<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
    (void)hipFuncSetAttribute((const void *)k_slice_kd_b, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 1024);
    (void)hipFuncSetAttribute((const void *)k_pose_b<256>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 8192); /* (8 KiB: the kernel's static LDS, the y-bucket rows) */
    (void)hipFuncSetAttribute((const void *)k_pose_b<512>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 8192);
    (void)hipFuncSetAttribute((const void *)k_pose_b<768>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 8192);
    (void)hipFuncSetAttribute((const void *)k_pose_b<POSE_T>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 8192);
    (void)hipFuncSetAttribute((const void *)k_pose<false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 8192);
    (void)hipFuncSetAttribute((const void *)k_pose<false, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 8192);
    (void)hipFuncSetAttribute((const void *)k_pose<false, 768>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 8192);
    (void)hipFuncSetAttribute((const void *)k_pose<false, POSE_T>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 8192);
    (void)hipFuncSetAttribute((const void *)k_pose<true, POSE_T>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 8192);
    (void)hipFuncSetAttribute((const void *)k_dyn_boundary_pts, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    (void)hipFuncSetAttribute((const void *)k_dyn_adjust_pts, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    (void)hipFuncSetAttribute((const void *)k_dyn_adjust_fit, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)k_band_indices, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 1024);
    (void)hipFuncSetAttribute((const void *)k_win_slice<256>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 2048);
    (void)hipFuncSetAttribute((const void *)k_win_slice<512>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 2048);
    (void)hipFuncSetAttribute((const void *)k_win_slice<768>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 2048);
    (void)hipFuncSetAttribute((const void *)k_win_slice<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 2048);
    (void)hipFuncSetAttribute((const void *)k_win_slice_b<256>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 2048);
    (void)hipFuncSetAttribute((const void *)k_win_slice_b<512>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 2048);
    (void)hipFuncSetAttribute((const void *)k_win_slice_b<768>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 2048);
    (void)hipFuncSetAttribute((const void *)k_win_slice_b<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 2048);
    (void)hipFuncSetAttribute((const void *)k_win_scatter<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * WIN_S_MAX);
    (void)hipFuncSetAttribute((const void *)k_win_scatter<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * WIN_S_MAX);
    (void)hipFuncSetAttribute((const void *)k_win_scatter_b<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * WIN_S_MAX);
    (void)hipFuncSetAttribute((const void *)k_win_scatter_b<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * WIN_S_MAX);
    (void)hipFuncSetAttribute((const void *)k_win_scatter<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 2048);
    (void)hipFuncSetAttribute((const void *)k_win_scatter<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 2048);
    (void)hipFuncSetAttribute((const void *)k_win_scatter_b<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 2048);
    (void)hipFuncSetAttribute((const void *)k_win_scatter_b<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, h->max_lds - 2048);
    /* none of the opt-ins above is fatal (the launches check for themselves): leave no stale error behind for the next HIP
       user of this thread (a framework that reads hipGetLastError after its own calls would trip over it) */
    (void)hipGetLastError();
    *out = h;
    return PPP_OK;
}

int ppp_destroy(ppp_handle h)
{
    if (!h) return PPP_ERR_ARG;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    delete h;
    return PPP_OK;
}

const char *ppp_last_error(ppp_handle h) { return h ? h->err.c_str() : "null handle"; }

int ppp_set_params(ppp_handle h, const ppp_params *p)
{
    if (!h || !p) return PPP_ERR_ARG;
    int rc = validate_params(h, p);
    if (rc) return rc;
    bool rescale = h->have_cloud && (p->change_range != h->P.change_range);
    if (rescale) return fail(h, PPP_ERR_ARG, "ChangeRange changed after the cloud was set: set the cloud again");
    const ppp_params before = h->P;
    h->P = *p;
    h->win_disabled = false;
    if (h->have_cloud) {
        HIPCHK(h, hipSetDevice(h->device));
        int rcs = settle(h);
        if (!rcs) rcs = make_plan(h);
        if (rcs) h->P = before; /* the rejected parameters do not stay: the next call plans the accepted ones again (planned is false) */
        return rcs;
    }
    return PPP_OK;
}

int ppp_set_cloud(ppp_handle h, const float *xyz_host, size_t n, size_t stride_bytes, const float *viewpoint)
{
    if (!h || (!xyz_host && n) || stride_bytes < 12 || (stride_bytes & 3)) return fail(h, PPP_ERR_ARG, "bad cloud arguments");
    HIPCHK(h, hipSetDevice(h->device));
    { int rcs = settle(h); if (rcs) return rcs; }
    size_t bytes = n * stride_bytes;
    HIPCHK(h, h->scratch.ensure(bytes));
    if (bytes) HIPCHK(h, hipMemcpyAsync(h->scratch.p, xyz_host, bytes, hipMemcpyHostToDevice, h->stream));
    return set_cloud_common(h, h->scratch.p, n, stride_bytes, viewpoint);
}

/* ---- a PCD file straight into HBM ---- */
namespace {
constexpr size_t PCD_PIECE = (size_t)8 << 20; /* bytes of one piece in flight (two pinned buffers of this size per handle) */
constexpr int PCD_READERS = 4;                /* threads that fill a piece from the page cache (one memcpy stream does ~5 GB/s) */

/* fills buf[0, len) from the file at `pos`, with PCD_READERS preads side by side; false on a short read */
bool read_piece(int fd, char *buf, size_t len, long long pos)
{
    std::atomic<bool> ok{true};
    auto part = [&](size_t a, size_t b) {
        while (a < b) {
            const ssize_t r = pread(fd, buf + a, b - a, (off_t)(pos + (long long)a));
            if (r <= 0) { if (r < 0 && errno == EINTR) continue; ok = false; return; }
            a += (size_t)r;
        }
    };
    const int nt = len >= ((size_t)1 << 20) ? PCD_READERS : 1;
    const size_t per = ((len + nt - 1) / nt + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    th.reserve((size_t)nt);
    for (int t = 1; t < nt; ++t) {
        const size_t a = std::min(len, per * t), b = std::min(len, per * (t + 1));
        if (a >= b) continue;
        try { th.emplace_back(part, a, b); } catch (...) { part(a, b); } /* (no thread to be had: read it here) */
    }
    part(0, std::min(len, per));
    for (auto &t : th) t.join();
    return ok;
}
} // namespace

int ppp_set_cloud_pcd(ppp_handle h, const char *path, size_t *n_out, float viewpoint_out[7])
{
    if (!h || !path) return fail(h, PPP_ERR_ARG, "bad cloud arguments");
    if (n_out) *n_out = 0;
    ppp_pcd_layout L;
    int rc = ppp_pcd_probe(path, &L);
    if (rc != PPP_OK) return fail(h, rc, std::string("not a readable PCD file: ") + path);
    if (viewpoint_out) memcpy(viewpoint_out, L.viewpoint, sizeof(L.viewpoint));
    const bool direct = L.data_kind == 1 && L.xyz_float32 && L.y_offset == L.x_offset + 4 && L.z_offset == L.x_offset + 8 &&
                        (L.x_offset & 3) == 0 && (L.record_bytes & 3) == 0 && L.record_bytes >= 12 && L.points > 0;
    if (!direct) { /* ascii, compressed, F8 / integer coordinates, x y z apart: through the host */
        float *xyz = nullptr;
        size_t n = 0;
        float vp[7];
        rc = ppp_load_pcd(path, &xyz, &n, vp);
        if (rc != PPP_OK) return fail(h, rc, std::string("not a readable PCD file: ") + path);
        rc = ppp_set_cloud(h, xyz, n, 12, vp);
        ppp_free(xyz);
        if (rc == PPP_OK && n_out) *n_out = n;
        return rc;
    }
    HIPCHK(h, hipSetDevice(h->device));
    { int rcs = settle(h); if (rcs) return rcs; }
    const size_t bytes = L.points * L.record_bytes;
    HIPCHK(h, h->scratch.ensure(bytes));
    const size_t piece = std::min(PCD_PIECE, (bytes + 4095) & ~(size_t)4095);
    if (h->pcd_stage_bytes < piece) {
        for (int b = 0; b < 2; ++b) { if (h->pcd_stage[b]) (void)hipHostFree(h->pcd_stage[b]); h->pcd_stage[b] = nullptr; }
        h->pcd_stage_bytes = 0;
        for (int b = 0; b < 2; ++b) HIPCHK(h, hipHostMalloc((void **)&h->pcd_stage[b], piece, hipHostMallocDefault));
        h->pcd_stage_bytes = piece;
    }
    for (int b = 0; b < 2; ++b) if (!h->pcd_ev[b]) HIPCHK(h, hipEventCreateWithFlags(&h->pcd_ev[b], hipEventDisableTiming));
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return fail(h, PPP_ERR_IO, std::string("cannot open ") + path);
    bool io_ok = true;
    hipError_t he = hipSuccess;
    size_t done = 0;
    for (int k = 0; done < bytes && io_ok && he == hipSuccess; ++k) {
        const int b = k & 1;
        const size_t len = std::min(h->pcd_stage_bytes, bytes - done);
        if (k >= 2) he = hipEventSynchronize(h->pcd_ev[b]); /* the copy that last read this buffer */
        if (he != hipSuccess) break;
        io_ok = read_piece(fd, h->pcd_stage[b], len, L.data_offset + (long long)done);
        if (!io_ok) break;
        he = hipMemcpyAsync(h->scratch.p + done, h->pcd_stage[b], len, hipMemcpyHostToDevice, h->stream);
        if (he == hipSuccess) he = hipEventRecord(h->pcd_ev[b], h->stream);
        done += len;
    }
    close(fd);
    if (!io_ok || he != hipSuccess) {
        (void)hipStreamSynchronize(h->stream); /* nothing may still read the pinned buffers */
        if (he != hipSuccess) return fail(h, PPP_ERR_HIP, std::string("PCD upload: ") + hipGetErrorString(he));
        return fail(h, PPP_ERR_IO, std::string("short read: ") + path);
    }
    rc = set_cloud_common(h, h->scratch.p + L.x_offset, L.points, L.record_bytes, L.viewpoint, true); /* (the records are in the handle's own buffer: no caller to hand them back to) */
    if (rc == PPP_OK && n_out) *n_out = L.points;
    return rc;
}

int ppp_set_cloud_device(ppp_handle h, const float *xyz_dev, size_t n, size_t stride_bytes, const float *viewpoint)
{
    if (!h || (!xyz_dev && n) || stride_bytes < 12 || (stride_bytes & 3)) return fail(h, PPP_ERR_ARG, "bad cloud arguments");
    HIPCHK(h, hipSetDevice(h->device));
    { int rcs = settle(h); if (rcs) return rcs; }
    return set_cloud_common(h, (const char *)xyz_dev, n, stride_bytes, viewpoint);
}

int ppp_set_cloud_device_async(ppp_handle h, const float *xyz_dev, size_t n, size_t stride_bytes, const float *viewpoint)
{
    if (!h || (!xyz_dev && n) || stride_bytes < 12 || (stride_bytes & 3)) return fail(h, PPP_ERR_ARG, "bad cloud arguments");
    HIPCHK(h, hipSetDevice(h->device));
    { int rcs = settle(h); if (rcs) return rcs; }
    return set_cloud_common(h, (const char *)xyz_dev, n, stride_bytes, viewpoint, true);
}

int ppp_range_interval(const ppp_params *p, float min_x, float max_x, float *lo, float *hi, int *num_slices)
{
    if (!p || !lo || !hi) return PPP_ERR_ARG;
    const int S = ppp_slice_walk(p->walk, min_x, max_x, p->tool_radius, nullptr, 0);
    if (num_slices) *num_slices = S;
    if (S <= 0 || S >= PPP_WALK_HARD_MAX) return PPP_ERR_ARG;
    const int sb = std::min(std::max(0, p->slice_begin), S);
    const int se = (p->slice_end <= 0 || p->slice_end > S) ? S : p->slice_end;
    if (sb >= se) { *lo = INFINITY; *hi = -INFINITY; return PPP_OK; } /* an empty range needs no point */
    if (sb == 0 && se == S) { *lo = -INFINITY; *hi = INFINITY; return PPP_OK; }
    std::vector<float> px((size_t)S);
    ppp_slice_walk(p->walk, min_x, max_x, p->tool_radius, px.data(), S);
    *lo = (float)((int)px[sb] - 2) - p->range_margin;       /* as make_plan: band of slice s = [int(px) - 2, int(px) + 2] */
    *hi = (float)((int)px[se - 1] + 2) + p->range_margin;
    return PPP_OK;
}

int ppp_set_cloud_part(ppp_handle h, const float *xyz_host, size_t n_part, size_t stride_bytes, const float *viewpoint,
                       const int *cloud_index, const float mn[3], const float mx[3], size_t n_valid_total, float part_lo, float part_hi)
{
    if (!h || (!xyz_host && n_part) || stride_bytes < 12 || (stride_bytes & 3) || !mn || !mx) return fail(h, PPP_ERR_ARG, "bad cloud arguments");
    if (!(part_lo <= part_hi) || n_valid_total < n_part || n_valid_total > 0x7fffffffu / 8) return fail(h, PPP_ERR_ARG, "bad part interval / point count");
    if (n_valid_total > 0)
        for (int d = 0; d < 3; ++d)
            if (!std::isfinite(mn[d]) || !std::isfinite(mx[d]) || !(mn[d] <= mx[d]))
                return fail(h, PPP_ERR_ARG, "the whole cloud's bounds must be finite and ordered (mn <= mx): they define the slice walk and the slab grid");
    HIPCHK(h, hipSetDevice(h->device));
    { int rcs = settle(h); if (rcs) return rcs; }
    const size_t bytes = n_part * stride_bytes;
    HIPCHK(h, h->scratch.ensure(bytes));
    if (bytes) HIPCHK(h, hipMemcpyAsync(h->scratch.p, xyz_host, bytes, hipMemcpyHostToDevice, h->stream));
    h->n = n_part;
    h->aligned = false;
    if (h->back) { delete h->back; h->back = nullptr; }
    if (viewpoint) memcpy(h->vp, viewpoint, 12); else h->vp[0] = h->vp[1] = h->vp[2] = 0.f;
    HIPCHK(h, h->X.ensure(n_part)); HIPCHK(h, h->Y.ensure(n_part)); HIPCHK(h, h->Z.ensure(n_part));
    if (n_part) {
        (void)hipGetLastError();
        hipLaunchKernelGGL(k_ingest, dim3((unsigned)((n_part + 255) / 256)), dim3(256), 0, h->stream, h->scratch.p, stride_bytes, (int)n_part,
                           h->P.change_range, h->X.p, h->Y.p, h->Z.p);
        HIPCHK(h, hipGetLastError());
    }
    h->part_has_idx = cloud_index != nullptr;
    if (cloud_index) {
        HIPCHK(h, h->part_idx.ensure(std::max<size_t>(n_part, 1)));
        if (n_part) HIPCHK(h, hipMemcpyAsync(h->part_idx.p, cloud_index, n_part * sizeof(int), hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    /* the whole cloud's bounds and count, in the planner's units (what refresh_bounds_and_plan measures on a whole cloud) */
    for (int d = 0; d < 3; ++d) { h->h_mn[d] = mn[d]; h->h_mx[d] = mx[d]; }
    h->h_nvalid = (int)n_valid_total;
    if (!h->h_nvalid) for (int d = 0; d < 3; ++d) { h->h_mn[d] = 3.402823466e+38f; h->h_mx[d] = -3.402823466e+38f; }
    h->part_given = true; h->part_lo = part_lo; h->part_hi = part_hi;
    h->win_disabled = false;
    h->big_path = false;
    h->have_cloud = true;
    h->rec_current = false; /* (no conversion pass of the window plan's kind: the bounds came with the call) */
    h->planned = false; h->index_built = false; h->gen_done = false; h->meta_fresh = false; h->path_done = false;
    h->normals_valid = false;
    h->drop_graph();
    return make_plan(h);
}

namespace {

/* the slab index of a handle, complete: built, its meta block read back, the arena passes run if a slab overflowed */
int index_ready(ppp_handle h, bool strict = true)
{
    int rc = ensure_index(h);
    if (rc) return rc;
    rc = fetch_meta(h);
    if (rc) return rc;
    if (overflowed_fast_path(h)) {
        h->big_path = true; h->drop_graph();
        HIPCHK(h, h->arena.ensure((size_t)64 * std::max<size_t>(h->n, 1) + (1u << 20)));
        rc = enqueue_index(h); if (rc) return rc;
        rc = fetch_meta(h); if (rc) return rc;
    }
    /* strict: any deferred device error is the caller's (preprocessing replaces the cloud).  The single-call mirrors only
       read the index: an earlier GenPath's slice error is not theirs, a slab that fits nowhere is */
    if (!strict && h->hmeta.err != DERR_CAPACITY) return PPP_OK;
    return map_dev_err(h);
}

/* path_translation_alg.cpp:171-174: the cloud carried back by invTransAlign, with its own slab index (same point
   indices).  The cloud is fixed between runs, so this happens once per cloud change, not per getPath. */
int rebuild_back(ppp_handle h)
{
    if (!h->back) {
        int rc = ppp_create(h->device, &h->back);
        if (rc) return fail(h, rc, "sensor-frame handle");
    }
    ppp_handle b = h->back;
    b->P = h->P;
    b->P.tool_radius = 1.0e6; b->P.dynamic_adjustment = 0; b->P.slice_begin = 0; b->P.slice_end = 0; /* one slice: this handle only ever serves its index */
    b->n = h->n;
    memcpy(b->vp, h->vp, sizeof(b->vp));
    HIPCHK(h, b->X.ensure(h->n)); HIPCHK(h, b->Y.ensure(h->n)); HIPCHK(h, b->Z.ensure(h->n));
    if (h->n) {
        Mat34 M;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) M.m[r][c] = h->invTA[r][c];
        LAUNCH(h, "k_transform_se3", k_transform_se3, (unsigned)((h->n + 255) / 256), 256, 0, h->X.p, h->Y.p, h->Z.p, (int)h->n, M, b->X.p, b->Y.p, b->Z.p);
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    b->big_path = false;
    int rc = refresh_bounds_and_plan(b);
    if (rc == PPP_OK) rc = index_ready(b);
    if (rc == PPP_OK) { hipError_t e = hipStreamSynchronize(b->stream); if (e != hipSuccess) rc = PPP_ERR_HIP; }
    if (rc != PPP_OK) return fail(h, rc, std::string("sensor-frame index: ") + b->err);
    return PPP_OK;
}

/* the resident cloud was replaced or moved: bounds, plan, and the sensor-frame copy when the cloud is aligned */
int cloud_changed(ppp_handle h)
{
    int rc = refresh_bounds_and_plan(h);
    if (rc == PPP_OK && h->aligned) rc = rebuild_back(h);
    return rc;
}

} // namespace

int ppp_trans2center(ppp_handle h, float *trans_align16, float *centroid3, float *covariance9)
{
    if (!h) return PPP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    { int rcs = settle(h); if (rcs) return rcs; }
    if (!h->have_cloud) return fail(h, PPP_ERR_ARG, "no cloud set");
    if (h->ranged || h->part_given) return fail(h, PPP_ERR_ARG, "preprocess the cloud on a whole-cloud handle");
    if (h->aligned) return fail(h, PPP_ERR_ARG, "the cloud is aligned already (TransAlign would be overwritten): set the cloud again");
    const int n = (int)h->n;
    if (n == 0 || h->h_nvalid == 0) return fail(h, PPP_ERR_ARG, "no finite point to align");
    const size_t stride = ((size_t)n + 3) & ~(size_t)3;
    DevBuf<float> V, sums;
    auto cleanup = [&]() { V.release(); sums.release(); };
    hipError_t e = V.ensure(6 * stride);
    if (e == hipSuccess) e = sums.ensure(8);
    if (e != hipSuccess) { cleanup(); return fail(h, PPP_ERR_HIP, std::string("trans2center buffers: ") + hipGetErrorString(e)); }
    float hs[6] = {0, 0, 0, 0, 0, 0}, c[3] = {0, 0, 0};
    const int hcnt = h->h_nvalid; /* the finite points, counted with the bounds */
    const unsigned gb = (unsigned)((n + 255) / 256);
    auto phase1 = [&]() -> int { /* pcl::compute3DCentroid: three running float sums, / float(count) */
        LAUNCH(h, "k_seq_prep_centroid", k_seq_prep_centroid, gb, 256, 0, h->X.p, h->Y.p, h->Z.p, n, stride, V.p);
        LAUNCH(h, "k_seq_sum", k_seq_sum, 3, 64 * SEQ_WAVES, 0, V.p, stride, n, sums.p);
        HIPCHK(h, hipMemcpyAsync(hs, sums.p, 3 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return PPP_OK;
    };
    auto phase2 = [&]() -> int { /* pcl::computeCovarianceMatrix: six running float sums of float products */
        LAUNCH(h, "k_seq_prep_cov", k_seq_prep_cov, gb, 256, 0, h->X.p, h->Y.p, h->Z.p, n, c[0], c[1], c[2], stride, V.p);
        LAUNCH(h, "k_seq_sum", k_seq_sum, 6, 64 * SEQ_WAVES, 0, V.p, stride, n, sums.p);
        HIPCHK(h, hipMemcpyAsync(hs, sums.p, 6 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return PPP_OK;
    };
    int rc = phase1();
    if (rc == PPP_OK && hcnt <= 0) rc = fail(h, PPP_ERR_ARG, "no finite point to align");
    if (rc == PPP_OK) {
        for (int d = 0; d < 3; ++d) c[d] = hs[d] / static_cast<float>(hcnt);
        rc = phase2();
    }
    if (rc != PPP_OK) { cleanup(); return rc; }
    float cov[3][3];
    cov[1][1] = hs[0]; cov[1][2] = hs[1]; cov[2][2] = hs[2]; cov[0][0] = hs[3]; cov[0][1] = hs[4]; cov[0][2] = hs[5];
    cov[1][0] = cov[0][1]; cov[2][0] = cov[0][2]; cov[2][1] = cov[1][2];
    if (centroid3) memcpy(centroid3, c, sizeof(c));
    if (covariance9) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) covariance9[3 * i + j] = cov[i][j];
    ppp_align::EigenSolver3f es;
    es.compute(cov);
    if (es.complex_pair || !es.converged) {
        cleanup();
        return fail(h, PPP_ERR_DOMAIN, "trans2center: the float Schur form of the covariance keeps a complex pair (two equal extents) or did not converge");
    }
    ppp_align::trans_align(es, c, h->TA);
    ppp_align::inverse4(h->TA, h->invTA);
    if (trans_align16) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) trans_align16[4 * i + j] = h->TA[i][j];
    auto phase3 = [&]() -> int { /* pcl::transformPointCloud(*cloud, *cloud, TransAlign) */
        Mat34 M;
        for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 4; ++cc) M.m[r][cc] = h->TA[r][cc];
        LAUNCH(h, "k_transform_se3", k_transform_se3, gb, 256, 0, h->X.p, h->Y.p, h->Z.p, n, M, h->X.p, h->Y.p, h->Z.p);
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return PPP_OK;
    };
    rc = phase3();
    cleanup();
    if (rc != PPP_OK) return rc;
    h->aligned = true;
    h->drop_graph();
    return cloud_changed(h);
}

int ppp_remove_outlier(ppp_handle h, int mean_k, double stddev_mul, size_t *n_kept, double *threshold)
{
    if (!h) return PPP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    { int rcs = settle(h); if (rcs) return rcs; }
    if (!h->have_cloud) return fail(h, PPP_ERR_ARG, "no cloud set");
    if (mean_k < 1 || mean_k > 63) return fail(h, PPP_ERR_ARG, "mean_k must be in [1, 63]");
    if (h->ranged || h->part_given) return fail(h, PPP_ERR_ARG, "preprocess the cloud on a whole-cloud handle");
    int rc = ensure_index(h);
    if (rc) return rc;
    rc = fetch_meta(h);
    if (rc) return rc;
    if (overflowed_fast_path(h)) { h->big_path = true; h->drop_graph(); HIPCHK(h, h->arena.ensure((size_t)64 * std::max<size_t>(h->n, 1) + (1u << 20))); rc = enqueue_index(h); if (rc) return rc; rc = fetch_meta(h); if (rc) return rc; }
    rc = map_dev_err(h);
    if (rc) return rc;
    const int n = (int)h->n, ns = h->hmeta.n_sorted;
    if (ns < mean_k + 1) return fail(h, PPP_ERR_ARG, "fewer finite points than mean_k + 1 (PCL reads past its neighbour vectors here)");
    /* first radius of the k-NN gather: mean_k + 1 points of a sheet of the cloud's mean areal density, +25 % */
    const double area = ((double)h->h_mx[0] - h->h_mn[0]) * ((double)h->h_mx[1] - h->h_mn[1]);
    const double rho = (area > 0 && h->h_nvalid > 0) ? (double)h->h_nvalid / area : 1.0;
    const float r0 = (float)std::max(0.5, 1.25 * std::sqrt((double)(mean_k + 1) / (3.14159265358979 * rho)));
    const int nblocks = (n + SOR_CHUNK - 1) / SOR_CHUNK, nparts = std::max(1, std::min(1024, (n + 255) / 256));
    DevBuf<float> dist, X2, Y2, Z2;
    DevBuf<double> part;
    DevBuf<int> bcnt;
    DevBuf<SorStats> st;
    auto cleanup = [&]() { dist.release(); X2.release(); Y2.release(); Z2.release(); part.release(); bcnt.release(); st.release(); };
    hipError_t e = dist.ensure(n);
    if (e == hipSuccess) e = X2.ensure(n);
    if (e == hipSuccess) e = Y2.ensure(n);
    if (e == hipSuccess) e = Z2.ensure(n);
    if (e == hipSuccess) e = part.ensure(2 * (size_t)nparts);
    if (e == hipSuccess) e = bcnt.ensure(nblocks);
    if (e == hipSuccess) e = st.ensure(1);
    if (e != hipSuccess) { cleanup(); return fail(h, PPP_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    SorStats hst;
    auto run = [&]() -> int {
        HIPCHK(h, hipMemsetAsync(dist.p, 0, sizeof(float) * (size_t)n, h->stream)); /* non-finite points: distance 0 */
        LAUNCH(h, "k_sor_dist", k_sor_dist, (unsigned)((ns + DYN_WAVES - 1) / DYN_WAVES), 64 * DYN_WAVES, 0, h->meta.p, h->sorted4.p, h->slab_start.p,
               h->slab_xmin.p, h->slab_xmax.p, mean_k, r0, dist.p);
        LAUNCH(h, "k_sor_partial", k_sor_partial, nparts, 256, 0, dist.p, n, part.p);
        LAUNCH(h, "k_sor_threshold", k_sor_threshold, 1, 256, 0, h->meta.p, part.p, nparts, stddev_mul, st.p);
        LAUNCH(h, "k_sor_count", k_sor_count, nblocks, 256, 0, dist.p, n, st.p, bcnt.p);
        LAUNCH(h, "k_sor_scan", k_sor_scan, 1, 1024, 0, bcnt.p, nblocks, st.p);
        LAUNCH(h, "k_sor_compact", k_sor_compact, nblocks, 256, 0, dist.p, n, st.p, bcnt.p, h->X.p, h->Y.p, h->Z.p, X2.p, Y2.p, Z2.p);
        HIPCHK(h, hipMemcpyAsync(&hst, st.p, sizeof(SorStats), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return PPP_OK;
    };
    rc = run();
    if (rc == PPP_OK) { h->meta_in_flight = false; h->meta_fresh = false; rc = fetch_meta(h); }
    if (rc == PPP_OK) rc = map_dev_err(h);
    if (rc != PPP_OK) { cleanup(); return rc; }
    /* the filtered cloud replaces the resident one (sor.filter(*cloud)) */
    std::swap(h->X, X2); std::swap(h->Y, Y2); std::swap(h->Z, Z2);
    h->n = (size_t)hst.n_kept;
    if (n_kept) *n_kept = h->n;
    if (threshold) *threshold = hst.threshold;
    cleanup();
    h->drop_graph();
    return cloud_changed(h);
}

int ppp_voxel_down(ppp_handle h, float lx, float ly, float lz, size_t *n_out, int *overflow)
{
    if (!h) return PPP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    { int rcs = settle(h); if (rcs) return rcs; }
    if (!h->have_cloud) return fail(h, PPP_ERR_ARG, "no cloud set");
    if (h->ranged || h->part_given) return fail(h, PPP_ERR_ARG, "preprocess the cloud on a whole-cloud handle");
    if (!(lx > 0.f) || !(ly > 0.f) || !(lz > 0.f) || !std::isfinite(lx) || !std::isfinite(ly) || !std::isfinite(lz))
        return fail(h, PPP_ERR_ARG, "leaf sizes must be positive and finite");
    if (overflow) *overflow = 0;
    if (n_out) *n_out = h->n;
    const int n = (int)h->n;
    if (n == 0) return PPP_OK;
    /* voxel_grid.hpp applyFilter: inverse_leaf_size_ = 1 / leaf_size_ (float), the index-overflow test on the float extents,
       min_b_ / max_b_ / div_b_ / divb_mul_ */
    const float inv[3] = {1.0f / lx, 1.0f / ly, 1.0f / lz};
    VoxGrid g;
    long long cells = 1, dxyz = 1;
    int div_b[3];
    if (h->h_nvalid > 0) {
        for (int d = 0; d < 3; ++d) {
            dxyz *= (long long)((h->h_mx[d] - h->h_mn[d]) * inv[d]) + 1;
            const int min_b = (int)std::floor(h->h_mn[d] * inv[d]), max_b = (int)std::floor(h->h_mx[d] * inv[d]);
            div_b[d] = max_b - min_b + 1;
            cells *= div_b[d];
            g.inv[d] = inv[d];
            g.min_b[d] = (float)min_b;
            if (dxyz > 0x7fffffffLL || cells > 0x7fffffffLL || dxyz <= 0 || cells <= 0) {
                /* "Leaf size is too small for the input dataset. Integer indices would overflow.": output = input */
                if (overflow) *overflow = 1;
                return PPP_OK;
            }
        }
        g.mul[0] = 1; g.mul[1] = div_b[0]; g.mul[2] = div_b[0] * div_b[1];
    } else {
        for (int d = 0; d < 3; ++d) { g.inv[d] = inv[d]; g.min_b[d] = 0.f; g.mul[d] = 0; }
    }
    g.none = (unsigned)cells;
    int end_bit = 1;
    while (end_bit < 32 && (cells >> end_bit)) ++end_bit;
    const int nblocks = (n + VOX_CHUNK - 1) / VOX_CHUNK;
    DevBuf<unsigned> key, key2;
    DevBuf<int> idx, idx2, bcnt;
    DevBuf<char> tmp;
    DevBuf<float4> pts;
    DevBuf<float> X2, Y2, Z2;
    DevBuf<VoxStats> st;
    auto cleanup = [&]() { key.release(); key2.release(); idx.release(); idx2.release(); bcnt.release(); tmp.release(); pts.release();
                           X2.release(); Y2.release(); Z2.release(); st.release(); };
    size_t tmp_bytes = 0;
    hipError_t e = ppp_sort_pairs_u32(nullptr, &tmp_bytes, nullptr, nullptr, nullptr, nullptr, (size_t)n, end_bit, h->stream);
    if (e == hipSuccess) e = key.ensure(n);
    if (e == hipSuccess) e = key2.ensure(n);
    if (e == hipSuccess) e = idx.ensure(n);
    if (e == hipSuccess) e = idx2.ensure(n);
    if (e == hipSuccess) e = bcnt.ensure(nblocks);
    if (e == hipSuccess) e = tmp.ensure(tmp_bytes);
    if (e == hipSuccess) e = pts.ensure(n);
    if (e == hipSuccess) e = X2.ensure(n);
    if (e == hipSuccess) e = Y2.ensure(n);
    if (e == hipSuccess) e = Z2.ensure(n);
    if (e == hipSuccess) e = st.ensure(1);
    if (e != hipSuccess) { cleanup(); return fail(h, PPP_ERR_HIP, std::string("voxel_down buffers: ") + hipGetErrorString(e)); }
    VoxStats hst{0};
    auto run = [&]() -> int {
        const unsigned gb = (unsigned)((n + 255) / 256);
        LAUNCH(h, "k_vox_key", k_vox_key, gb, 256, 0, h->X.p, h->Y.p, h->Z.p, n, g, key.p, idx.p);
        HIPCHK(h, ppp_sort_pairs_u32(tmp.p, &tmp_bytes, key.p, key2.p, idx.p, idx2.p, (size_t)n, end_bit, h->stream));
        LAUNCH(h, "k_vox_count", k_vox_count, nblocks, 256, 0, key2.p, n, g.none, bcnt.p);
        LAUNCH(h, "k_vox_scan", k_vox_scan, 1, 1024, 0, bcnt.p, nblocks, st.p);
        LAUNCH(h, "k_vox_gather", k_vox_gather, gb, 256, 0, h->X.p, h->Y.p, h->Z.p, idx2.p, n, pts.p);
        LAUNCH(h, "k_vox_reduce", k_vox_reduce, nblocks, 256, 0, key2.p, pts.p, n, g.none, bcnt.p, X2.p, Y2.p, Z2.p);
        HIPCHK(h, hipMemcpyAsync(&hst, st.p, sizeof(VoxStats), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return PPP_OK;
    };
    int rc = run();
    if (rc != PPP_OK) { cleanup(); return rc; }
    std::swap(h->X, X2); std::swap(h->Y, Y2); std::swap(h->Z, Z2);
    h->n = (size_t)hst.n_out;
    if (n_out) *n_out = h->n;
    cleanup();
    h->drop_graph();
    return cloud_changed(h);
}

int ppp_smooth_mls(ppp_handle h, double search_radius, int order, size_t *n_out)
{
    if (!h) return PPP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    { int rcs = settle(h); if (rcs) return rcs; }
    if (!h->have_cloud) return fail(h, PPP_ERR_ARG, "no cloud set");
    if (h->ranged || h->part_given) return fail(h, PPP_ERR_ARG, "preprocess the cloud on a whole-cloud handle");
    if (!(search_radius > 0) || !std::isfinite(search_radius)) return fail(h, PPP_ERR_ARG, "search radius must be positive"); /* mls.hpp: "Invalid search radius" */
    if (order < 0 || order > 3) return fail(h, PPP_ERR_ARG, "polynomial order must be in [0, 3]");
    int rc = ensure_index(h);
    if (rc) return rc;
    rc = fetch_meta(h);
    if (rc) return rc;
    if (overflowed_fast_path(h)) { h->big_path = true; h->drop_graph(); HIPCHK(h, h->arena.ensure((size_t)64 * std::max<size_t>(h->n, 1) + (1u << 20))); rc = enqueue_index(h); if (rc) return rc; rc = fetch_meta(h); if (rc) return rc; }
    rc = map_dev_err(h);
    if (rc) return rc;
    const int n = (int)h->n, ns = h->hmeta.n_sorted;
    if (n_out) *n_out = h->n;
    if (n == 0) return PPP_OK;
    const int nblocks = (n + VOX_CHUNK - 1) / VOX_CHUNK;
    DevBuf<float4> rec;
    DevBuf<float> X2, Y2, Z2;
    DevBuf<int> bcnt;
    DevBuf<VoxStats> st;
    auto cleanup = [&]() { rec.release(); X2.release(); Y2.release(); Z2.release(); bcnt.release(); st.release(); };
    hipError_t e = rec.ensure(n);
    if (e == hipSuccess) e = X2.ensure(n);
    if (e == hipSuccess) e = Y2.ensure(n);
    if (e == hipSuccess) e = Z2.ensure(n);
    if (e == hipSuccess) e = bcnt.ensure(nblocks);
    if (e == hipSuccess) e = st.ensure(1);
    if (e != hipSuccess) { cleanup(); return fail(h, PPP_ERR_HIP, std::string("smooth buffers: ") + hipGetErrorString(e)); }
    VoxStats hst{0};
    auto run = [&]() -> int {
        HIPCHK(h, hipMemsetAsync(rec.p, 0, sizeof(float4) * (size_t)n, h->stream));
        const unsigned gb = (unsigned)((std::max(ns, 1) + 255) / 256);
        const float rf = (float)search_radius;
        const double sq = search_radius * search_radius;
        if (order == 3) LAUNCH(h, "k_mls<3>", k_mls<3>, gb, 256, 0, h->meta.p, h->sorted4.p, h->slab_start.p, h->slab_xmin.p, h->slab_xmax.p, rf, sq, rec.p);
        else if (order == 2) LAUNCH(h, "k_mls<2>", k_mls<2>, gb, 256, 0, h->meta.p, h->sorted4.p, h->slab_start.p, h->slab_xmin.p, h->slab_xmax.p, rf, sq, rec.p);
        else LAUNCH(h, "k_mls<1>", k_mls<1>, gb, 256, 0, h->meta.p, h->sorted4.p, h->slab_start.p, h->slab_xmin.p, h->slab_xmax.p, rf, sq, rec.p);
        LAUNCH(h, "k_flag_count", k_flag_count, nblocks, 256, 0, rec.p, n, bcnt.p);
        LAUNCH(h, "k_vox_scan", k_vox_scan, 1, 1024, 0, bcnt.p, nblocks, st.p);
        LAUNCH(h, "k_flag_compact", k_flag_compact, nblocks, 256, 0, rec.p, n, bcnt.p, X2.p, Y2.p, Z2.p);
        HIPCHK(h, hipMemcpyAsync(&hst, st.p, sizeof(VoxStats), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return PPP_OK;
    };
    rc = run();
    if (rc == PPP_OK) { h->meta_in_flight = false; h->meta_fresh = false; rc = fetch_meta(h); }
    if (rc == PPP_OK) rc = map_dev_err(h);
    if (rc != PPP_OK) { cleanup(); return rc; }
    std::swap(h->X, X2); std::swap(h->Y, Y2); std::swap(h->Z, Z2);
    h->n = (size_t)hst.n_out;
    if (n_out) *n_out = h->n;
    cleanup();
    h->drop_graph();
    return cloud_changed(h);
}

int ppp_get_cloud(ppp_handle h, float *xyz, size_t cap, size_t *n)
{
    if (!h) return PPP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    { int rcs = settle(h); if (rcs) return rcs; }
    if (!h->have_cloud) return fail(h, PPP_ERR_ARG, "no cloud set");
    if (n) *n = h->n;
    const size_t k = std::min(cap, h->n);
    if (xyz && k) {
        std::vector<float> t(3 * k);
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, copy_sync(h, t.data(), h->X.p, 4 * k, hipMemcpyDeviceToHost));
        HIPCHK(h, copy_sync(h, t.data() + k, h->Y.p, 4 * k, hipMemcpyDeviceToHost));
        HIPCHK(h, copy_sync(h, t.data() + 2 * k, h->Z.p, 4 * k, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < k; ++i) { xyz[3 * i] = t[i]; xyz[3 * i + 1] = t[k + i]; xyz[3 * i + 2] = t[2 * k + i]; }
    }
    return PPP_OK;
}

int ppp_num_points(ppp_handle h, size_t *n)
{
    if (!h || !n) return PPP_ERR_ARG;
    *n = h->n;
    return PPP_OK;
}

int ppp_gen_path_async(ppp_handle h)
{
    if (!h) return PPP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    { int rcs = settle_enqueue_only(h); if (rcs) return rcs; }
    if (!h->have_cloud) return fail(h, PPP_ERR_ARG, "no cloud set");
    if (!h->internal) { h->last_out2 = nullptr; h->last_out2_cap = 0; } /* a plain call: the list stays in the handle */
    if (!h->planned) { int rc = make_plan(h); if (rc) return rc; }
    if (h->win_path) { /* bounds + binning into the slices' windows, then everything per slice in one launch (ppp_window.h) */
        int rcw = enqueue_window_gen(h);
        if (rcw) return rcw;
        h->gen_done = true; ++h->gen_serial;
        h->path_done = false;
        if (h->chain_calls) return PPP_OK;
        return enqueue_meta_copy(h);
    }
    if (!slice_lds_ok(h, h->capb)) return fail(h, PPP_ERR_CAPACITY, "band capacity exceeds the LDS of this device");
    int rc = enqueue_index(h);
    if (rc) return rc;
    /* the pairing kernel leaves every slice's waypoint count for k_pose -- unless the dynamic adjustment re-fits the knots after it */
    int *cnt_out = (h->P.pairing == PPP_PAIR_KD && !h->P.dynamic_adjustment) ? h->slice_wpcnt.p : nullptr;
    if (h->P.pairing == PPP_PAIR_KD) {
        LAUNCH(h, "k_slice_kd", k_slice_kd<false>, h->S_cap, slice_threads(h, h->S_cap), slice_kd_bytes(h->capb), h->sorted4.p, h->slab_start.p, h->meta.p,
               h->px.p, h->lo.p, h->hi.p, h->capb, h->node_x.p, h->node_y.p, h->node_z.p, h->node_cap, h->node_start.p, h->node_cnt.p,
               h->band_cnt.p, h->big_slices.p, h->arena.p, (unsigned long long)h->arena.cap, h->P.trim, h->P.path_resolution, h->W_cap, cnt_out);
        if (h->big_path)
            LAUNCH(h, "k_slice_kd_arena", k_slice_kd<true>, h->S_cap, SLICE_KD_T, 0, h->sorted4.p, h->slab_start.p, h->meta.p, h->px.p,
                   h->lo.p, h->hi.p, h->capb, h->node_x.p, h->node_y.p, h->node_z.p, h->node_cap, h->node_start.p, h->node_cnt.p,
                   h->band_cnt.p, h->big_slices.p, h->arena.p, (unsigned long long)h->arena.cap, h->P.trim, h->P.path_resolution, h->W_cap, cnt_out);
    } else {
        LAUNCH(h, "k_slice", k_slice, h->S_cap, K_SLICE_T, slice_lds_bytes(h->capb), h->sorted4.p, h->slab_start.p, h->meta.p, h->px.p,
               h->lo.p, h->hi.p, h->P.pairing, h->capb, h->node_x.p, h->node_y.p, h->node_z.p, h->node_cap, h->node_start.p,
               h->node_cnt.p, h->band_cnt.p, h->big_slices.p);
        if (h->big_path)
            LAUNCH(h, "k_slice_brute_arena", k_slice_brute_arena, h->S_cap, 1024, 0, h->sorted4.p, h->slab_start.p, h->meta.p, h->px.p,
                   h->lo.p, h->hi.p, h->P.pairing, (int)h->n, h->node_x.p, h->node_y.p, h->node_z.p, h->node_cap, h->node_start.p,
                   h->node_cnt.p, h->band_cnt.p, h->big_slices.p, h->arena.p, (unsigned long long)h->arena.cap);
    }
    if (h->P.dynamic_adjustment) {
        int rc2 = enqueue_dynamic(h);
        if (rc2) return rc2;
    }
    h->gen_done = true; ++h->gen_serial;
    h->path_done = false;
    if (h->chain_calls) return PPP_OK; /* getPath follows in the same enqueue and ends with the copy */
    return enqueue_meta_copy(h);
}

/* getPath's second half: postion_smooth, reduceRPY, TransFlangeposition (path_translation_alg.cpp:212-214) */
int enqueue_finish(ppp_handle h, const DevParams &D)
{
    /* postion_smooth solved directly (one 65-tap filter per waypoint), then reduceRPY, the flange offset and the copy into
       the caller's buffer of the batched form: ONE launch */
    LAUNCH(h, "k_smooth_solve", k_smooth_solve, h->sm_tiles, SMF_T, 0, h->meta.p, D, h->W_cap, h->wp_pre.p, h->wp_smooth.p, h->wp_out.p,
           h->tail.p, h->out2, h->out2_cap);
    return enqueue_meta_copy(h);
}

int ppp_get_path_async(ppp_handle h)
{
    if (!h) return PPP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    { int rcs = settle_enqueue_only(h); if (rcs) return rcs; }
    if (!h->gen_done) return fail(h, PPP_ERR_ARG, "call ppp_gen_path_async first");
    if (h->win_path) { /* the per-waypoint half ran with the slices: offsets, compaction and getPath's list-wide second half */
        int rcw = enqueue_window_finish(h);
        if (rcw) return rcw;
        h->path_done = true;
        h->list_final = !h->ranged;
        h->meta_in_flight = true; h->meta_from_batch = false; /* the finish launch's last workgroup leaves the meta block in hmeta_pinned */
        return PPP_OK;
    }
    DevParams D = dev_params(h);
    int nk = std::max(1, h->S_cap);
    PoseBack PB;
    memset(&PB, 0, sizeof(PB));
    const int *cnt_in = (h->P.pairing == PPP_PAIR_KD && !h->P.dynamic_adjustment) ? h->slice_wpcnt.p : nullptr;
    if (h->aligned) {
        if (h->ranged) return fail(h, PPP_ERR_UNSUPPORTED, "Alignment with a slice range");
        if (!h->back || !h->back->index_built) return fail(h, PPP_ERR_ARG, "aligned cloud without its sensor-frame index");
        PB.sorted4 = h->back->sorted4.p; PB.slab_start = h->back->slab_start.p; PB.slab_xmin = h->back->slab_xmin.p; PB.slab_xmax = h->back->slab_xmax.p;
        PB.m = h->back->meta.p; PB.ytab = h->back->slab_ytab.p;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) PB.inv[r][c] = h->invTA[r][c];
        LAUNCH(h, "k_pose<aligned>", (k_pose<true, POSE_T>), nk, h->pose_threads, pose_lds_bytes(h->knot_cap, h->stage_cap, h->tab_slabs), h->meta.p, D, h->sorted4.p, h->slab_start.p, h->slab_xmin.p,
               h->slab_xmax.p, h->px.p, h->node_x.p, h->node_y.p, h->node_z.p, h->node_start.p, h->node_cnt.p, h->wp_cnt.p, h->wp_off.p,
               h->tail.p, h->W_cap, h->big_path ? 1 : 0, h->knot_cap, h->stage_cap, h->tab_slabs, h->pose_pad,
               h->wp_xyz.p, h->wp_nn.p, h->wp_normal.p, h->wp_pre.p, PB, h->slab_ytab.p, cnt_in);
    } else
    {
#define PPP_POSE_ARGS h->meta.p, D, h->sorted4.p, h->slab_start.p, h->slab_xmin.p, \
           h->slab_xmax.p, h->px.p, h->node_x.p, h->node_y.p, h->node_z.p, h->node_start.p, h->node_cnt.p, h->wp_cnt.p, h->wp_off.p, \
           h->tail.p, h->W_cap, h->big_path ? 1 : 0, h->knot_cap, h->stage_cap, h->tab_slabs, h->pose_pad, \
           h->wp_xyz.p, h->wp_nn.p, h->wp_normal.p, h->wp_pre.p, PB, h->slab_ytab.p, cnt_in
        if (h->pose_threads <= 256) LAUNCH(h, "k_pose", (k_pose<false, 256>), nk, h->pose_threads, pose_lds_bytes(h->knot_cap, h->stage_cap, h->tab_slabs), PPP_POSE_ARGS);
        else if (h->pose_threads <= 512) LAUNCH(h, "k_pose", (k_pose<false, 512>), nk, h->pose_threads, pose_lds_bytes(h->knot_cap, h->stage_cap, h->tab_slabs), PPP_POSE_ARGS);
        else if (h->pose_threads <= 768) LAUNCH(h, "k_pose", (k_pose<false, 768>), nk, h->pose_threads, pose_lds_bytes(h->knot_cap, h->stage_cap, h->tab_slabs), PPP_POSE_ARGS);
        else LAUNCH(h, "k_pose", (k_pose<false, POSE_T>), nk, h->pose_threads, pose_lds_bytes(h->knot_cap, h->stage_cap, h->tab_slabs), PPP_POSE_ARGS);
#undef PPP_POSE_ARGS
    }
    h->path_done = true;
    h->list_final = false;
    h->stage_compact = true;
    /* a slice-range handle stops here: postion_smooth couples the slices of different handles */
    if (!h->ranged) {
        int rc = enqueue_finish(h, D); /* publishes the meta block itself */
        if (rc) return rc;
        h->list_final = true;
        return PPP_OK;
    }
    return enqueue_meta_copy(h);
}

int ppp_finish_path_async(ppp_handle h, const float *pre6_dev, size_t W, const int *counts, size_t nkept)
{
    if (!h) return PPP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    { int rcs = settle(h); if (rcs) return rcs; }
    if (!h->have_cloud) return fail(h, PPP_ERR_ARG, "no cloud set");
    if (!h->planned) { int rc = make_plan(h); if (rc) return rc; }
    if ((W && !pre6_dev) || (nkept && !counts)) return fail(h, PPP_ERR_ARG, "bad arguments");
    if (W > (size_t)h->W_cap || nkept > (size_t)h->S_cap) return fail(h, PPP_ERR_CAPACITY, "the list is larger than this handle's plan (other cloud or parameters?)");
    if (!h->index_built && !h->win_path) { int rc = enqueue_index(h); if (rc) return rc; } /* the meta block is initialised by k_setup (k_count_given sets what the finish reads) */
    DevParams D = dev_params(h);
    if (nkept) HIPCHK(h, hipMemcpyAsync(h->wp_cnt.p, counts, nkept * sizeof(int), hipMemcpyHostToDevice, h->stream));
    LAUNCH(h, "k_count_given", k_count_given, 1, 1024, 0, h->meta.p, D, (int)nkept, (int)W, h->wp_cnt.p, h->wp_off.p, h->tail.p, h->W_cap);
    if (W) LAUNCH(h, "k_load_pre", k_load_pre, (unsigned)((W + 255) / 256), 256, 0, h->meta.p, pre6_dev, h->wp_pre.p);
    int rc = enqueue_finish(h, D);
    if (rc) return rc;
    h->gen_done = true; ++h->gen_serial; h->path_done = true; h->list_final = true;
    h->stage_compact = true; /* (no per-waypoint stage lists belong to a list finished from gathered blocks) */
    return PPP_OK;
}

int ppp_run_async(ppp_handle h)
{
    if (!h) return PPP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    { int rcs = settle_enqueue_only(h); if (rcs) return rcs; }
    if (!h->have_cloud) return fail(h, PPP_ERR_ARG, "no cloud set");
    if (!h->planned) { int rc = make_plan(h); if (rc) return rc; }
    h->last_out2 = nullptr; h->last_out2_cap = 0;
    if (h->timing) {
        int rc = ppp_gen_path_async(h);
        return rc ? rc : ppp_get_path_async(h);
    }
    if (!h->graph_exec && h->graph_epoch_seen != h->epoch) {
        /* the first pass of a plan (a new cloud, new parameters) is enqueued directly: six launches cost the host less than
           capturing and instantiating a graph does (~0.1 ms), and a planner fed with a new cloud every time never replays.
           The second call of the same plan captures. */
        h->graph_epoch_seen = h->epoch;
        h->chain_calls = true;
        ++h->internal;
        int rc = ppp_gen_path_async(h);
        h->chain_calls = false;
        if (rc == PPP_OK) rc = ppp_get_path_async(h);
        --h->internal;
        return rc;
    }
    if (!h->graph_exec) {
        HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        h->chain_calls = true;
        ++h->internal;
        int rc = ppp_gen_path_async(h);
        h->chain_calls = false;
        if (rc == PPP_OK) rc = ppp_get_path_async(h);
        --h->internal;
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(h->stream, &g);
        if (rc != PPP_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) return fail(h, PPP_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        h->graph = g;
        e = hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0);
        if (e != hipSuccess) { h->drop_graph(); return fail(h, PPP_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
    }
    h->meta_fresh = false;
    HIPCHK(h, hipGraphLaunch(h->graph_exec, h->stream));
    if (!h->win_path) h->index_built = true;
    h->stage_compact = !h->win_path;
    h->gen_done = true; ++h->gen_serial; h->path_done = true;
    h->list_final = !h->ranged;
    h->meta_in_flight = true; /* the captured sequence ends with the meta copy */
    h->meta_from_batch = false;
    return PPP_OK;
}

extern "C++" {
namespace {

/* can this handle's pass run as rows of the batched launches?  (kd pairing on the LDS fast path, whole cloud, one-level
   scatter; anything else -- dynamic adjustment, alignment, slice ranges, arena passes -- keeps its own launch sequence) */
bool batch_eligible(const ppp_handle h)
{
    return h->P.pairing == PPP_PAIR_KD && !h->P.dynamic_adjustment && !h->aligned && !h->ranged && !h->big_path && !h->two_pass_scatter &&
           slice_lds_ok(h, h->capb);
}

#define LAUNCHB(lead, strm, name, kern, grid, block, shmem, ...)                                      \
    do {                                                                                              \
        KTimer *_t = (lead)->timing ? timer_for((lead), name) : nullptr;                              \
        if (_t) (void)hipEventRecord(_t->e0[_t->used], (strm));                                       \
        (void)hipGetLastError();                                                                      \
        hipLaunchKernelGGL(kern, grid, dim3(block), (shmem), (strm), __VA_ARGS__);                    \
        if (_t) { (void)hipEventRecord(_t->e1[_t->used], (strm)); _t->used++; }                       \
        hipError_t _le = hipGetLastError();                                                           \
        if (_le != hipSuccess) return fail((lead), PPP_ERR_HIP, std::string(name) + ": " + hipGetErrorString(_le)); \
    } while (0)

/* the members' records (host -> device, synchronous, outside any capture) and the launch geometry of the batch */
int upload_members(ppp_handle lead, BatchGraph *bg, float *dst_dev, const size_t *offset_rows, const size_t *cap_rows)
{
    const size_t count = bg->hs.size();
    std::vector<BatchMember> mem(count);
    int &maxB = bg->maxB, &max_slab_cap = bg->max_slab_cap, &max_capb = bg->max_capb;
    int &gx_mm = bg->gx_mm, &gx_scat = bg->gx_scat, &gx_sort = bg->gx_sort, &gx_slice = bg->gx_slice, &gx_pose = bg->gx_pose, &gx_smooth = bg->gx_smooth;
    bool &full_slabs = bg->full_slabs, &ppt8 = bg->ppt8;
    int max_n = 0;
    long long slices_total = 0;
    bg->pose_threads = 256; bg->pose_lds = 0;
    for (size_t i = 0; i < count; ++i) max_n = std::max(max_n, (int)bg->hs[i]->n);
    ppt8 = max_n > PPP_PPT8_FROM;
    const int chunk = (ppt8 ? 8 : 4) * SCAT_T;
    /* the bounds + histogram pass: about 2048 workgroups over the whole batch (each flushes its LDS histogram with one
       atomic per non-empty slab) */
    const int mm_share = std::max(4, (int)(2048 / count));
    for (size_t i = 0; i < count; ++i) {
        ppp_handle h = bg->hs[i];
        BatchMember &M = mem[i];
        memset(&M, 0, sizeof(M));
        M.m = h->meta.p; M.P = dev_params(h);
        M.X = h->X.p; M.Y = h->Y.p; M.Z = h->Z.p; M.n = (int)h->n;
        M.mm_part = h->mm_part.p;
        const float xr = h->h_mx[0] - h->h_mn[0];
        M.slab_x0 = h->h_mn[0]; M.slab_invw = (h->h_nvalid && xr > 0.f) ? (float)h->B / xr : 0.f; /* as enqueue_index */
        M.incl_lo = h->incl_lo; M.incl_hi = h->incl_hi;
        M.B = h->B; M.S_cap = h->S_cap; M.slab_cap = h->slab_cap; M.capb = h->capb; M.node_cap = h->node_cap; M.W_cap = h->W_cap;
        M.knot_cap = h->knot_cap; M.stage_cap = h->stage_cap; M.tab_slabs = h->tab_slabs; M.pose_pad = h->pose_pad;
        bg->pose_threads = std::max(bg->pose_threads, h->pose_threads);
        bg->pose_lds = std::max(bg->pose_lds, pose_lds_bytes(h->knot_cap, h->stage_cap, h->tab_slabs));
        slices_total += h->S_cap;
        M.out2 = dst_dev ? dst_dev + 6 * offset_rows[i] : nullptr;
        M.out2_cap = dst_dev ? (int)std::min<size_t>(cap_rows[i], 0x7fffffff) : 0;
        M.g_minmax = std::max(1, std::min(std::min(h->mm_grid, PPP_MM_GRID_MAX), mm_share));
        M.g_scatter = std::max(1, ((int)h->n + chunk - 1) / chunk);
        M.g_sort = h->B; M.g_slice = h->S_cap; M.g_pose = std::max(1, h->S_cap); M.g_smooth = h->sm_tiles;
        h->slab_cnt_used = true;
        M.slab_cnt = h->slab_cnt.p; M.slab_start = h->slab_start.p; M.slab_cursor = h->slab_cursor.p; M.coarse_cursor = h->coarse_cursor.p;
        M.px = h->px.p; M.lo = h->lo.p; M.hi = h->hi.p;
        M.unsorted4 = h->unsorted4.p; M.sorted4 = h->sorted4.p; M.slab_xmin = h->slab_xmin.p; M.slab_xmax = h->slab_xmax.p;
        M.big_slabs = h->big_slabs.p; M.big_slices = h->big_slices.p;
        M.node_x = h->node_x.p; M.node_y = h->node_y.p; M.node_z = h->node_z.p;
        M.node_start = h->node_start.p; M.node_cnt = h->node_cnt.p; M.band_cnt = h->band_cnt.p;
        M.wp_cnt = h->wp_cnt.p; M.wp_off = h->wp_off.p; M.tail = h->tail.p;
        M.wp_xyz = h->wp_xyz.p; M.wp_normal = h->wp_normal.p; M.wp_nn = h->wp_nn.p;
        M.wp_pre = h->wp_pre.p; M.wp_smooth = h->wp_smooth.p; M.wp_out = h->wp_out.p; M.ytab = h->slab_ytab.p; M.slice_wpcnt = h->slice_wpcnt.p;
        h->mm_grid_used = M.g_minmax;
        maxB = std::max(maxB, h->B); max_slab_cap = std::max(max_slab_cap, h->slab_cap); max_capb = std::max(max_capb, h->capb);
        full_slabs = full_slabs || (h->B > 0 && h->h_nvalid / h->B > 1000);
        gx_mm = std::max(gx_mm, M.g_minmax); gx_scat = std::max(gx_scat, M.g_scatter); gx_sort = std::max(gx_sort, M.g_sort);
        gx_slice = std::max(gx_slice, M.g_slice); gx_pose = std::max(gx_pose, M.g_pose); gx_smooth = std::max(gx_smooth, M.g_smooth);
    }
    {   /* two 512-thread slice workgroups per CU when every member's band leaves room for two (see slice_threads) */
        bool two_fit = true;
        for (size_t i = 0; i < count; ++i) two_fit = two_fit && 2 * (slice_kd_bytes(bg->hs[i]->capb) + 1024) <= (size_t)lead->max_lds;
        bg->slice_thr = (two_fit && slices_total >= 2LL * lead->num_cus) ? 512 : SLICE_KD_T;
    }
    HIPCHK(lead, copy_sync(lead, bg->members.p, mem.data(), sizeof(BatchMember) * count, hipMemcpyHostToDevice));
    return PPP_OK;
}

/* the window path's records (ppp_window.h): every member's three launches become three launches for the batch */
int upload_members_win(ppp_handle lead, BatchGraph *bg, float *dst_dev, const size_t *offset_rows, const size_t *cap_rows)
{
    const size_t count = bg->hs.size();
    std::vector<WinArgs> mem(count);
    bg->win_ppt = 4; bg->win_threads = 256; bg->win_lds = 0; bg->win_scat_lds = 0; bg->win_fin_lds = 0;
    bg->gx_scat = 1; bg->gx_slice = 1; bg->gx_wfin = 1;
    long long slices_total = 0;
    /* one form of the binning launch for the batch: staged when any member is, with the points per thread all staged members have room for */
    bg->win_staged = false;
    for (size_t i = 0; i < count; ++i) bg->win_staged = bg->win_staged || bg->hs[i]->win_staged;
    for (size_t i = 0; i < count; ++i) bg->win_ppt = std::max(bg->win_ppt, bg->hs[i]->win_ppt);
    if (bg->win_staged) {
        int smax = 1;
        for (size_t i = 0; i < count; ++i) smax = std::max(smax, bg->hs[i]->S_cap);
        bg->win_ppt = 8;
        if (win_scatter_lds_bytes(smax, 8, WSC_T, true) + 2048 > (size_t)lead->max_lds) bg->win_ppt = 4;
        if (win_scatter_lds_bytes(smax, bg->win_ppt, WSC_T, true) + 2048 > (size_t)lead->max_lds) { bg->win_staged = false; bg->win_ppt = 8; }
    }
    for (size_t i = 0; i < count; ++i) {
        ppp_handle h = bg->hs[i];
        h->out2 = dst_dev ? dst_dev + 6 * offset_rows[i] : nullptr;
        h->out2_cap = dst_dev ? (int)std::min<size_t>(cap_rows[i], 0x7fffffff) : 0;
        WinArgs &A = mem[i];
        A = win_args(h);
        /* a member publishes its meta block into the batch's pinned array; a batch of ONE -- a stream of steps on one workpiece -- does
           not publish at all (the arrival counters are a microsecond at the end of every step): the block is fetched when somebody asks */
        A.meta_host = count > 1 ? bg->hmetas->pinned + i : nullptr;
        h->out2 = nullptr; h->out2_cap = 0;
        A.g_scatter = std::max(1, (A.n + bg->win_ppt * WSC_T - 1) / (bg->win_ppt * WSC_T)); /* (the members' partials are sized for 4 points per thread) */
        if (bg->win_staged) A.g_scatter = std::min(A.g_scatter, std::max(1, lead->num_cus)); /* (the staged form loops over its chunks) */
        slices_total += A.g_slice;
        /* (NB / NBc / yscale of the launch kind are set below, once the launch's total of slices is known) */
        bg->win_scat_lds = std::max(bg->win_scat_lds, win_scatter_lds_bytes(A.S, bg->win_ppt, WSC_T, bg->win_staged));
        bg->win_fin_lds = std::max(bg->win_fin_lds, sizeof(int) * ((size_t)A.nkept + 2));
        bg->gx_scat = std::max(bg->gx_scat, A.g_scatter); bg->gx_slice = std::max(bg->gx_slice, A.g_slice + 1); bg->gx_wfin = std::max(bg->gx_wfin, A.g_finish);
    }
    /* one thread count for the launch: what the widest member needs, for the launch's total of slice workgroups */
    bg->win_threads = 128;
    for (size_t i = 0; i < count; ++i) {
        if (win_throughput_launch(bg->hs[i], slices_total)) win_args_throughput(bg->hs[i], mem[i]);
        bg->win_lds = std::max(bg->win_lds, win_slice_lds_for(bg->hs[i], mem[i].NBc));
        const int t = win_pick_threads(bg->hs[i], slices_total);
        if (!t) return fail(lead, PPP_ERR_CAPACITY, "window path: a member's windows do not fit a workgroup");
        bg->win_threads = std::max(bg->win_threads, t);
    }
    if (getenv("PPP_WIN_DEBUG"))
        fprintf(stderr, "[ppp] window batch: %zu members, %lld slices, threads %d, NBc %d, lds %zu B, ppt %d\n", count, slices_total, bg->win_threads, mem[0].NBc, bg->win_lds, bg->win_ppt);
    HIPCHK(lead, copy_sync(lead, bg->wmembers.p, mem.data(), sizeof(WinArgs) * count, hipMemcpyHostToDevice));
    return PPP_OK;
}

/* one launch per stage over all members (blockIdx.y = member); ends with ONE copy of all meta blocks */
/* the stage launches over members [first, first + n) of a batch, on `strm` (timers: the lead's, eager runs only) */
static int enqueue_batched_stages(ppp_handle lead, BatchGraph *bg, hipStream_t strm, size_t first, size_t n)
{
    const int maxB = bg->maxB, max_slab_cap = bg->max_slab_cap, max_capb = bg->max_capb;
    const int gx_mm = bg->gx_mm, gx_scat = bg->gx_scat, gx_sort = bg->gx_sort, gx_slice = bg->gx_slice, gx_pose = bg->gx_pose, gx_smooth = bg->gx_smooth;
    const bool full_slabs = bg->full_slabs, ppt8 = bg->ppt8;
    const unsigned gy = (unsigned)n;
    if (bg->win) { /* the window path: bounds + binning, the per-slice kernel, the finish -- three launches for the whole batch */
        const WinArgs *wm = bg->wmembers.p + first;
        if (bg->win_staged && bg->win_ppt == 8) LAUNCHB(lead, strm, "k_win_scatter_b", (k_win_scatter_b<8, true>), dim3(bg->gx_scat, gy), WSC_T, bg->win_scat_lds, wm);
        else if (bg->win_staged) LAUNCHB(lead, strm, "k_win_scatter_b", (k_win_scatter_b<4, true>), dim3(bg->gx_scat, gy), WSC_T, bg->win_scat_lds, wm);
        else if (bg->win_ppt == 8) LAUNCHB(lead, strm, "k_win_scatter_b", (k_win_scatter_b<8, false>), dim3(bg->gx_scat, gy), WSC_T, bg->win_scat_lds, wm);
        else LAUNCHB(lead, strm, "k_win_scatter_b", (k_win_scatter_b<4, false>), dim3(bg->gx_scat, gy), WSC_T, bg->win_scat_lds, wm);
        const int T = bg->win_threads;
        if (T <= 256) LAUNCHB(lead, strm, "k_win_slice_b", k_win_slice_b<256>, dim3(bg->gx_slice, gy), T, bg->win_lds, wm);
        else if (T <= 512) LAUNCHB(lead, strm, "k_win_slice_b", k_win_slice_b<512>, dim3(bg->gx_slice, gy), T, bg->win_lds, wm);
        else if (T <= 768) LAUNCHB(lead, strm, "k_win_slice_b", k_win_slice_b<768>, dim3(bg->gx_slice, gy), T, bg->win_lds, wm);
        else LAUNCHB(lead, strm, "k_win_slice_b", k_win_slice_b<1024>, dim3(bg->gx_slice, gy), T, bg->win_lds, wm);
        LAUNCHB(lead, strm, "k_win_finish_b", k_win_finish_b, dim3(bg->gx_wfin, gy), SMF_T, bg->win_fin_lds, wm);
        return PPP_OK;
    }
    const BatchMember *mem = bg->members.p + first;
    const size_t hist_lds = sizeof(int) * (size_t)maxB;
    LAUNCHB(lead, strm, "k_minmax_b", k_minmax_b, dim3(gx_mm, gy), MM_T, hist_lds, mem);
    /* (the set-up of every member rides in the scatter launch as that member's last workgroup) */
    if (ppt8) LAUNCHB(lead, strm, "k_slab_scatter_b", k_slab_scatter_b<8>, dim3(gx_scat + 1, gy), SCAT_T, 2 * hist_lds, mem);
    else LAUNCHB(lead, strm, "k_slab_scatter_b", k_slab_scatter_b<4>, dim3(gx_scat + 1, gy), SCAT_T, 2 * hist_lds, mem);
    LAUNCHB(lead, strm, "k_slab_sort_b", k_slab_sort_b, dim3(gx_sort, gy), full_slabs ? SORT_T : 256, (size_t)max_slab_cap * 12 + 16, mem);
    LAUNCHB(lead, strm, "k_slice_kd_b", k_slice_kd_b, dim3(gx_slice, gy), bg->slice_thr, slice_kd_bytes(max_capb), mem);
    if (bg->pose_threads <= 256) LAUNCHB(lead, strm, "k_pose_b", k_pose_b<256>, dim3(gx_pose, gy), bg->pose_threads, bg->pose_lds, mem);
    else if (bg->pose_threads <= 512) LAUNCHB(lead, strm, "k_pose_b", k_pose_b<512>, dim3(gx_pose, gy), bg->pose_threads, bg->pose_lds, mem);
    else if (bg->pose_threads <= 768) LAUNCHB(lead, strm, "k_pose_b", k_pose_b<768>, dim3(gx_pose, gy), bg->pose_threads, bg->pose_lds, mem);
    else LAUNCHB(lead, strm, "k_pose_b", k_pose_b<POSE_T>, dim3(gx_pose, gy), bg->pose_threads, bg->pose_lds, mem);
    LAUNCHB(lead, strm, "k_smooth_solve_b", k_smooth_solve_b, dim3(gx_smooth, gy), SMF_T, 0, mem);
    return PPP_OK;
}

int enqueue_batched(ppp_handle lead, BatchGraph *bg)
{
    const size_t count = bg->hs.size();
    /* A large batch goes as two halves side by side (the second on a member's own, otherwise idle stream, forked from and
       joined to the lead's): while one half's launch drains -- its last workgroups on a mostly idle device -- the other
       half's next stage is already running (64 x 250 k points: 0.87 -> 0.81 ms; four parts: 0.98).  Not when every launch is
       timed (eager): the events would serialise the halves anyway. */
    /* (the window path's three launches gain less from it: 16 members 0.150 ms as halves against 0.143 ms as one, 64 members 0.414 against 0.420) */
    const bool split = !bg->eager && bg->fork && count >= (size_t)(bg->win ? 4 * PPP_BATCH_SPLIT_FROM : PPP_BATCH_SPLIT_FROM);
    if (!split) {
        int rc = enqueue_batched_stages(lead, bg, lead->stream, 0, count);
        if (rc) return rc;
    } else {
        const size_t half = count / 2;
        hipStream_t side = bg->hs[half]->stream;
        HIPCHK(lead, hipEventRecord(bg->fork, lead->stream));
        HIPCHK(lead, hipStreamWaitEvent(side, bg->fork, 0));
        int rc = enqueue_batched_stages(lead, bg, lead->stream, 0, half);
        if (rc == PPP_OK) rc = enqueue_batched_stages(lead, bg, side, half, count - half);
        if (rc) return rc;
        HIPCHK(lead, hipEventRecord(bg->join[0], side));
        HIPCHK(lead, hipStreamWaitEvent(lead->stream, bg->join[0], 0));
    }
    if (bg->win) return PPP_OK; /* every member's finish launch leaves its meta block in pinned memory itself (win_publish_meta) */
    if (count == 1) /* nothing to collect, and nothing to publish per step either: the one meta block is fetched when somebody asks
                       (ppp_sync_batch, a getter) -- in a stream of steps on one workpiece a 200-byte copy behind every pass is a
                       fourth launch (a blit kernel, 4.4 us of a 68 us step) whose result only the last pass's reader looks at */
        return PPP_OK;
    if (bg->win) LAUNCHB(lead, lead->stream, "k_collect_meta", k_collect_meta_win, dim3((unsigned)count), 64, 0, bg->wmembers.p, (int)count, bg->metas.p);
    else LAUNCHB(lead, lead->stream, "k_collect_meta", k_collect_meta, dim3((unsigned)count), 64, 0, bg->members.p, (int)count, bg->metas.p);
    HIPCHK(lead, hipMemcpyAsync(bg->hmetas->pinned, bg->metas.p, sizeof(DevMeta) * count, hipMemcpyDeviceToHost, lead->stream));
    return PPP_OK;
}

} // namespace
} // extern "C++"

int ppp_run_batch_async(ppp_handle *hs, size_t count, float *dst_dev, const size_t *offset_rows, const size_t *cap_rows)
{
    if (!hs || !count || !hs[0]) return PPP_ERR_ARG;
    ppp_handle lead = hs[0];
    if (dst_dev && (!offset_rows || !cap_rows)) return fail(lead, PPP_ERR_ARG, "offset_rows / cap_rows are needed with a destination");
    if (count > 65535) return fail(lead, PPP_ERR_CAPACITY, "a batch holds at most 65535 workpieces (one grid row each)");
    HIPCHK(lead, hipSetDevice(lead->device));
    bool plain = false, batched = true, allwin = true;
    for (size_t i = 0; i < count; ++i) {
        ppp_handle h = hs[i];
        if (!h) return fail(lead, PPP_ERR_ARG, "null handle in the batch");
        if (h->device != lead->device) return fail(lead, PPP_ERR_ARG, "a batch lives on one device");
        for (size_t j = 0; j < i; ++j) if (hs[j] == h) return fail(lead, PPP_ERR_ARG, "a handle appears twice in the batch");
        if (!h->have_cloud) return fail(lead, PPP_ERR_ARG, "a handle of the batch has no cloud");
        int rc = settle(h);
        if (rc) return rc;
        if (!h->planned) { rc = make_plan(h); if (rc) { lead->err = h->err; return rc; } }
        plain = plain || h->timing;
        batched = batched && batch_eligible(h);
        allwin = allwin && h->win_path;
    }
    /* A member's list must not depend on the company it is planned in: the two paths differ in the last bits of the normals'
       float sums, so a batch runs as rows of batched launches only when ALL members are on the window path or NONE is
       (mixed batches take one graph branch per member, each on its own path). */
    if (allwin) batched = true;
    else for (size_t i = 0; i < count; ++i) if (hs[i]->win_path) batched = false;
    auto remember_dst = [&](size_t i) { /* where a re-run after an LDS overflow must put the list (rerun_with_arena) */
        hs[i]->last_out2 = dst_dev ? dst_dev + 6 * offset_rows[i] : nullptr;
        hs[i]->last_out2_cap = dst_dev ? (int)std::min<size_t>(cap_rows[i], 0x7fffffff) : 0;
    };
    /* kernel timing brackets every launch with events: the batched launches run eagerly on the lead's stream (timers of the
       lead), members that need their own launch sequence run as plain per-handle calls */
    const bool eager = plain && batched;
    if (plain && !batched) {
        for (size_t i = 0; i < count; ++i) {
            ++hs[i]->internal;
            int rc = ppp_gen_path_async(hs[i]);
            if (dst_dev) { hs[i]->out2 = dst_dev + 6 * offset_rows[i]; hs[i]->out2_cap = (int)std::min<size_t>(cap_rows[i], 0x7fffffff); }
            if (rc == PPP_OK) rc = ppp_get_path_async(hs[i]);
            hs[i]->out2 = nullptr; hs[i]->out2_cap = 0;
            --hs[i]->internal;
            remember_dst(i);
            if (rc) { lead->err = hs[i]->err; return rc; }
        }
        return PPP_OK;
    }
    BatchGraph *bg = nullptr;
    for (BatchGraph *cand : lead->batches) {
        bool same = cand && cand->hs.size() == count && cand->dst == dst_dev && cand->batched == batched && cand->eager == eager && cand->win == allwin;
        for (size_t i = 0; same && i < count; ++i)
            same = cand->hs[i] == hs[i] && cand->epochs[i] == hs[i]->epoch && (!dst_dev || (cand->off[i] == offset_rows[i] && cand->cap[i] == cap_rows[i]));
        if (same) { bg = cand; break; }
    }
    const bool fresh = bg != nullptr;
    int slot = -1;
    if (!fresh) {
        slot = lead->batch_next;
        lead->batch_next ^= 1;
        if (lead->batches[slot]) HIPCHK(lead, hipStreamSynchronize(lead->stream)); /* its last launch may still be running: it owns the records that launch reads */
        delete lead->batches[slot];
        bg = new BatchGraph();
        lead->batches[slot] = bg;
        bg->hs.assign(hs, hs + count);
        bg->dst = dst_dev;
        bg->batched = batched;
        bg->win = allwin;
        bg->eager = eager;
        if (dst_dev) { bg->off.assign(offset_rows, offset_rows + count); bg->cap.assign(cap_rows, cap_rows + count); }
        auto discard = [&]() { delete lead->batches[slot]; lead->batches[slot] = nullptr; };
        int rc = PPP_OK;
        const char *where = "";
        hipError_t e = hipSuccess;
        if (batched) {
            /* ONE launch per stage over all members.  Records and meta array first, outside the capture. */
            hipError_t ea = bg->members.ensure(count);
            if (ea == hipSuccess) ea = bg->wmembers.ensure(count);
            if (ea == hipSuccess) ea = bg->metas.ensure(count);
            bg->hmetas = std::make_shared<BatchMetas>();
            if (ea == hipSuccess) ea = hipHostMalloc((void **)&bg->hmetas->pinned, sizeof(DevMeta) * count, hipHostMallocDefault);
            if (ea != hipSuccess) { discard(); return fail(lead, PPP_ERR_HIP, std::string("batch buffers: ") + hipGetErrorString(ea)); }
            rc = bg->win ? upload_members_win(lead, bg, dst_dev, offset_rows, cap_rows) : upload_members(lead, bg, dst_dev, offset_rows, cap_rows);
            if (rc != PPP_OK) { discard(); return rc; }
            if (!eager && count >= PPP_BATCH_SPLIT_FROM) { /* the two halves of a large batch run side by side (enqueue_batched) */
                bg->join.resize(1, nullptr);
                hipError_t ee = hipEventCreateWithFlags(&bg->fork, hipEventDisableTiming);
                if (ee == hipSuccess) ee = hipEventCreateWithFlags(&bg->join[0], hipEventDisableTiming);
                if (ee != hipSuccess) { discard(); return fail(lead, PPP_ERR_HIP, std::string("batch events: ") + hipGetErrorString(ee)); }
            }
            if (!eager) {
                HIPCHK(lead, hipStreamBeginCapture(lead->stream, hipStreamCaptureModeThreadLocal));
                rc = enqueue_batched(lead, bg);
            }
        } else {
            HIPCHK(lead, hipEventCreateWithFlags(&bg->fork, hipEventDisableTiming));
            bg->join.resize(count, nullptr);
            for (size_t i = 1; i < count; ++i) HIPCHK(lead, hipEventCreateWithFlags(&bg->join[i], hipEventDisableTiming));
            /* one capture: the lead stream forks into every other handle's stream and joins them again */
            HIPCHK(lead, hipStreamBeginCapture(lead->stream, hipStreamCaptureModeThreadLocal));
            e = hipEventRecord(bg->fork, lead->stream);
            if (e != hipSuccess) where = "fork record";
            for (size_t i = 1; i < count && e == hipSuccess; ++i) {
                e = hipStreamWaitEvent(hs[i]->stream, bg->fork, 0);
                if (e != hipSuccess) where = "fork wait";
            }
            for (size_t i = 0; i < count && e == hipSuccess && rc == PPP_OK; ++i) {
                ppp_handle h = hs[i];
                h->chain_calls = true;
                ++h->internal;
                rc = ppp_gen_path_async(h);
                h->chain_calls = false;
                if (dst_dev) { h->out2 = dst_dev + 6 * offset_rows[i]; h->out2_cap = (int)std::min<size_t>(cap_rows[i], 0x7fffffff); }
                if (rc == PPP_OK) rc = ppp_get_path_async(h); /* the emitting launch also writes the list to its place in dst_dev */
                h->out2 = nullptr; h->out2_cap = 0;
                --h->internal;
                if (rc != PPP_OK) lead->err = "batch member " + std::to_string(i) + ": " + h->err;
                if (i && e == hipSuccess && rc == PPP_OK) {
                    e = hipEventRecord(bg->join[i], h->stream);
                    if (e != hipSuccess) where = "join record";
                    if (e == hipSuccess) { e = hipStreamWaitEvent(lead->stream, bg->join[i], 0); if (e != hipSuccess) where = "join wait"; }
                }
            }
        }
        if (!eager) {
            hipGraph_t g = nullptr;
            hipError_t e2 = hipStreamEndCapture(lead->stream, &g);
            if (rc != PPP_OK || e != hipSuccess || e2 != hipSuccess) {
                if (g) (void)hipGraphDestroy(g);
                discard();
                if (rc != PPP_OK) return rc;
                return fail(lead, PPP_ERR_HIP, std::string("batch capture (") + where + "): " + hipGetErrorString(e != hipSuccess ? e : e2));
            }
            bg->g = g;
            e = hipGraphInstantiate(&bg->ge, bg->g, nullptr, nullptr, 0);
            if (e != hipSuccess) { discard(); return fail(lead, PPP_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
        }
        bg->epochs.resize(count);
        for (size_t i = 0; i < count; ++i) bg->epochs[i] = hs[i]->epoch;
    }
    if (bg->eager) { int rc = enqueue_batched(lead, bg); if (rc) return rc; }
    else HIPCHK(lead, hipGraphLaunch(bg->ge, lead->stream));
    for (size_t i = 0; i < count; ++i) {
        ppp_handle h = hs[i];
        if (!(bg->batched && bg->win) && !(!bg->batched && h->win_path)) h->index_built = true;
        h->stage_compact = !((bg->batched && bg->win) || (!bg->batched && h->win_path));
        h->gen_done = true; ++h->gen_serial; h->path_done = true;
        h->list_final = !h->ranged;
        h->meta_fresh = false;
        h->meta_in_flight = !(bg->batched && count == 1); /* a batch of one does not publish its meta block: fetched on demand */
        h->meta_from_batch = bg->batched && count > 1;
        if (bg->batched && count > 1) { h->bmetas = bg->hmetas; h->bslot = i; }
        h->pending_stream = (i == 0) ? nullptr : lead->stream;
        remember_dst(i);
    }
    return PPP_OK;
}

extern "C++" {
namespace {
/* librccl, looked up on first use: the engine does not link it (a process that already carries a framework's RCCL
   gets that one back from dlopen) */
struct Rccl {
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
    int (*send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    bool ok = false;
};
const Rccl &rccl()
{
    static Rccl r = [] {
        Rccl x;
        /* PPP_RCCL_LIB: another RCCL build (or the recording stand-in of tests/test_gpu_parity.py) instead of the system's */
        const char *over = getenv("PPP_RCCL_LIB");
        void *lib = (over && *over) ? dlopen(over, RTLD_NOW | RTLD_GLOBAL) : nullptr;
        if (!lib && !(over && *over)) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib && !(over && *over)) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return x;
        x.group_start = (int (*)())dlsym(lib, "ncclGroupStart");
        x.group_end = (int (*)())dlsym(lib, "ncclGroupEnd");
        x.send = (int (*)(const void *, size_t, int, int, void *, hipStream_t))dlsym(lib, "ncclSend");
        x.recv = (int (*)(void *, size_t, int, int, void *, hipStream_t))dlsym(lib, "ncclRecv");
        x.ok = x.group_start && x.group_end && x.send && x.recv;
        return x;
    }();
    return r;
}
} // namespace
} // extern "C++"

int ppp_gather_waypoints(ppp_handle h, void *nccl_comm, int rank, int nranks, int root, const size_t *counts_rows, float *recv_dev)
{
    if (!h) return PPP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    { int rcs = settle(h); if (rcs) return rcs; }
    if (nranks < 1 || rank < 0 || rank >= nranks || root < 0 || root >= nranks || !counts_rows) return fail(h, PPP_ERR_ARG, "bad rank / root / counts");
    if (!h->path_done || !h->list_final) return fail(h, PPP_ERR_ARG, "no finished list on this handle (call ppp_get_path_async / ppp_run_async first)");
    if (counts_rows[rank] > (size_t)h->W_cap) return fail(h, PPP_ERR_CAPACITY, "counts_rows[rank] exceeds this handle's list capacity");
    if (rank == root && !recv_dev) return fail(h, PPP_ERR_ARG, "the root needs a receive buffer");
    /* one rank WITH a communicator (a one-rank group: the pre-flight of this exchange on one GPU): the block travels through librccl's
       send / recv group -- to itself -- instead of the plain copy a lone rank without a communicator gets */
    const bool rehearse = nranks == 1 && nccl_comm != nullptr;
    if (nranks == 1 && !rehearse) { /* nothing to exchange: the list goes to the receive buffer */
        if (counts_rows[0]) HIPCHK(h, hipMemcpyAsync(recv_dev, h->wp_out.p, counts_rows[0] * 24, hipMemcpyDeviceToDevice, h->stream));
        return PPP_OK;
    }
    if (!nccl_comm) return fail(h, PPP_ERR_ARG, "nccl_comm is NULL");
    const Rccl &R = rccl();
    if (!R.ok) return fail(h, PPP_ERR_UNSUPPORTED, "librccl.so not found (ncclGroupStart / ncclSend / ncclRecv)");
    PppGatherOps ops;
    ops.group_start = R.group_start; ops.group_end = R.group_end;
    ops.send = [](const void *buf, size_t count, int dtype, int peer, void *comm, void *stream) -> int { return rccl().send(buf, count, dtype, peer, comm, (hipStream_t)stream); };
    ops.recv = [](void *buf, size_t count, int dtype, int peer, void *comm, void *stream) -> int { return rccl().recv(buf, count, dtype, peer, comm, (hipStream_t)stream); };
    ops.local_copy = [](void *dst, const void *src, size_t bytes, void *stream) -> int {
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? 0 : 1;
    };
    int nres = 0;
    const int gres = rehearse ? ppp_gather_self_loop(ops, counts_rows[0], h->wp_out.p, recv_dev, nccl_comm, (void *)h->stream, &nres)
                              : ppp_gather_exchange(ops, rank, nranks, root, counts_rows, h->wp_out.p, recv_dev, nccl_comm, (void *)h->stream, &nres);
    switch (gres) {
    case PPP_GATHER_OK: break;
    case PPP_GATHER_COPY_FAILED: return fail(h, PPP_ERR_HIP, "copy of the root's own block failed");
    case PPP_GATHER_GROUP_START_FAILED: return fail(h, PPP_ERR_HIP, "ncclGroupStart failed (ncclResult " + std::to_string(nres) + ")");
    default: return fail(h, PPP_ERR_HIP, "RCCL send / recv failed (ncclResult " + std::to_string(nres) + ")");
    }
    return PPP_OK;
}

int ppp_get_stream(ppp_handle h, void **stream)
{
    if (!h || !stream) return PPP_ERR_ARG;
    *stream = (void *)h->stream;
    return PPP_OK;
}

int ppp_sync_batch(ppp_handle *hs, size_t count, size_t *failed)
{
    if (!hs || !count) return PPP_ERR_ARG;
    /* the members' meta blocks travel side by side: every handle that has none on its way gets its copy enqueued first (a copy and
       a wait per handle, one after the other, cost three handles 90 us at the end of a loop) */
    for (size_t i = 0; i < count; ++i) {
        ppp_handle h = hs[i];
        if (h && h->have_cloud && !h->plan_deferred && !h->pending_stream && !h->meta_in_flight && !h->meta_fresh && hipSetDevice(h->device) == hipSuccess)
            (void)enqueue_meta_copy(h);
    }
    for (size_t i = 0; i < count; ++i) {
        int rc = ppp_sync(hs[i]);
        if (rc) { if (failed) *failed = i; return rc; }
    }
    return PPP_OK;
}

int ppp_sync(ppp_handle h)
{
    if (!h) return PPP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = fetch_meta(h);
    if (rc) return rc;
    for (int tries = 0; tries < 2 && overflowed_fast_path(h); ++tries) {
        rc = rerun_with_arena(h);
        if (rc) return rc;
    }
    return map_dev_err(h);
}

int ppp_failed_slice(ppp_handle h) { return (h && h->hmeta.err != DERR_NONE && h->hmeta.err_slice != 0x7fffffff) ? h->hmeta.err_slice : -1; }

int ppp_num_slices(ppp_handle h, int *S)
{
    int rc = ensure_ready(h, true, false);
    if (rc) return rc;
    *S = h->hmeta.S;
    return PPP_OK;
}

int ppp_num_waypoints(ppp_handle h, size_t *W)
{
    int rc = ensure_ready(h, true, true);
    if (rc) return rc;
    rc = map_dev_err(h);
    if (rc) return rc;
    *W = (size_t)h->hmeta.W;
    return PPP_OK;
}

static const char *NOT_FINAL_MSG = "slice-range handle: the list ends before postion_smooth; gather the blocks and call ppp_finish_path_async";

int ppp_get_waypoints(ppp_handle h, float *out6, size_t cap, size_t *W)
{
    int rc = ensure_ready(h, true, true);
    if (rc) return rc;
    if (!h->list_final) return fail(h, PPP_ERR_ARG, NOT_FINAL_MSG);
    rc = map_dev_err(h);
    if (rc) return rc;
    size_t w = (size_t)h->hmeta.W;
    if (W) *W = w;
    if (out6 && cap) {
        size_t k = std::min(cap, w);
        if (k) HIPCHK(h, copy_sync(h, out6, h->wp_out.p, k * 24, hipMemcpyDeviceToHost));
    }
    return PPP_OK;
}

int ppp_get_waypoints_device(ppp_handle h, const float **dptr, size_t *W)
{
    int rc = ensure_ready(h, true, true);
    if (rc) return rc;
    if (!h->list_final) return fail(h, PPP_ERR_ARG, NOT_FINAL_MSG);
    rc = map_dev_err(h);
    if (rc) return rc;
    if (dptr) *dptr = h->wp_out.p;
    if (W) *W = (size_t)h->hmeta.W;
    return PPP_OK;
}

int ppp_copy_waypoints_to_device(ppp_handle h, float *dst_dev, size_t cap, size_t *W)
{
    int rc = ensure_ready(h, true, true);
    if (rc) return rc;
    if (!h->list_final) return fail(h, PPP_ERR_ARG, NOT_FINAL_MSG);
    rc = map_dev_err(h);
    if (rc) return rc;
    size_t w = (size_t)h->hmeta.W;
    if (W) *W = w;
    size_t k = std::min(cap, w);
    if (k && dst_dev) {
        HIPCHK(h, hipMemcpyAsync(dst_dev, h->wp_out.p, k * 24, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return PPP_OK;
}

int ppp_get_tail_index(ppp_handle h, int *tail, size_t cap, size_t *n)
{
    int rc = ensure_ready(h, true, true);
    if (rc) return rc;
    if (!h->list_final) return fail(h, PPP_ERR_ARG, NOT_FINAL_MSG);
    rc = map_dev_err(h);
    if (rc) return rc;
    size_t nk = (size_t)h->hmeta.nkept;
    if (n) *n = nk;
    if (tail && cap) {
        size_t k = std::min(cap, nk);
        if (k) HIPCHK(h, copy_sync(h, tail, h->tail.p, k * sizeof(int), hipMemcpyDeviceToHost));
    }
    return PPP_OK;
}

int ppp_get_waypoint_counts(ppp_handle h, int *counts, size_t cap, size_t *nkept)
{
    int rc = ensure_ready(h, true, true);
    if (rc) return rc;
    rc = map_dev_err(h);
    if (rc) return rc;
    size_t nk = (size_t)h->hmeta.nkept;
    if (nkept) *nkept = nk;
    if (counts && cap) {
        size_t k = std::min(cap, nk);
        if (k) HIPCHK(h, copy_sync(h, counts, h->wp_cnt.p, k * sizeof(int), hipMemcpyDeviceToHost));
    }
    return PPP_OK;
}

int ppp_copy_stage_to_device(ppp_handle h, int stage, float *dst_dev, size_t cap, size_t *W)
{
    int rc = ensure_ready(h, true, true);
    if (rc) return rc;
    rc = map_dev_err(h);
    if (rc) return rc;
    const float *src = stage == PPP_STAGE_WP_PRESMOOTH ? h->wp_pre.p : stage == PPP_STAGE_WP_SMOOTHED ? h->wp_smooth.p : nullptr;
    if (!src) return fail(h, PPP_ERR_ARG, "stage must be PPP_STAGE_WP_PRESMOOTH or PPP_STAGE_WP_SMOOTHED");
    if (stage == PPP_STAGE_WP_SMOOTHED && !h->list_final) return fail(h, PPP_ERR_ARG, NOT_FINAL_MSG);
    size_t w = (size_t)h->hmeta.W;
    if (W) *W = w;
    size_t k = std::min(cap, w);
    if (k && dst_dev) {
        HIPCHK(h, hipMemcpyAsync(dst_dev, src, k * 24, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return PPP_OK;
}

/* bounds and slice tables of the resident cloud: left by the last window pass (its checking workgroup), else by the slab index */
static int bounds_ready(ppp_handle h)
{
    if (h && h->have_cloud && h->planned && h->win_path && h->gen_done) {
        HIPCHK(h, hipSetDevice(h->device));
        return PPP_OK;
    }
    return ensure_index(h);
}

int ppp_minmax(ppp_handle h, float mn[3], float mx[3])
{
    int rc = bounds_ready(h);
    if (rc) return rc;
    rc = fetch_meta(h);
    if (rc) return rc;
    for (int d = 0; d < 3; ++d) { mn[d] = h->hmeta.mn[d]; mx[d] = h->hmeta.mx[d]; }
    return PPP_OK;
}

int ppp_get_slice_positions(ppp_handle h, float *px, size_t cap, size_t *S)
{
    int rc = bounds_ready(h);
    if (rc) return rc;
    rc = fetch_meta(h);
    if (rc) return rc;
    size_t s = (size_t)h->hmeta.S;
    if (S) *S = s;
    if (px && cap) {
        size_t k = std::min(cap, s);
        if (k) HIPCHK(h, copy_sync(h, px, h->px.p, k * sizeof(float), hipMemcpyDeviceToHost));
    }
    return PPP_OK;
}

static int band_indices(ppp_handle h, float lo, float hi, int *out, size_t cap, size_t *n)
{
    int capb = 4096;
    if (!slice_lds_ok(h, capb)) return fail(h, PPP_ERR_CAPACITY, "LDS too small");
    HIPCHK(h, h->scratch.ensure(sizeof(int) * (size_t)capb));
    LAUNCH(h, "k_band_indices", k_band_indices, 1, 256, slice_lds_bytes(capb), h->sorted4.p, h->slab_start.p, h->meta.p, lo, hi,
           capb, (int *)h->scratch.p, capb);
    h->meta_in_flight = false; h->meta_fresh = false; /* the kernel above wrote meta: take a fresh copy */
    int rc = fetch_meta(h);
    if (rc) return rc;
    const int *src = (const int *)h->scratch.p;
    if (h->hmeta.api_flag) {
        /* the band does not fit LDS: same answer from global scratch sized by the count just learned */
        const size_t cnt = (size_t)h->hmeta.api_cnt;
        const int NB = next_pow2((int)std::min<size_t>(cnt, 1u << 20));
        const size_t bytes = cnt * 4 + 16 + cnt * 8 * 2 + ((size_t)NB + 1) * 4 + 64;
        HIPCHK(h, h->scratch.ensure(bytes));
        int *dout = (int *)h->scratch.p;
        u64 *tmp = (u64 *)(h->scratch.p + ((cnt * 4 + 15) & ~(size_t)15));
        u64 *srt = tmp + cnt;
        int *hist = (int *)(srt + cnt);
        LAUNCH(h, "k_band_indices_big", k_band_indices_big, 1, 256, 0, h->sorted4.p, h->slab_start.p, h->meta.p, lo, hi, (int)h->n, tmp,
               srt, hist, NB, (int)cnt, dout);
        h->meta_in_flight = false; h->meta_fresh = false;
        rc = fetch_meta(h);
        if (rc) return rc;
        if (h->hmeta.api_flag) return fail(h, PPP_ERR_CAPACITY, "rangedX_index: band changed size between passes");
        src = dout;
    }
    if (n) *n = (size_t)h->hmeta.api_cnt;
    if (out && cap) {
        size_t k = std::min(cap, (size_t)h->hmeta.api_cnt);
        if (k) HIPCHK(h, copy_sync(h, out, src, k * sizeof(int), hipMemcpyDeviceToHost));
    }
    return PPP_OK;
}

int ppp_ranged_x_index(ppp_handle h, int position, int *out, size_t cap, size_t *n)
{
    int rc = index_ready(h, false); /* complete index: slabs beyond the LDS capacity go through the arena pass first */
    if (rc) return rc;
    return band_indices(h, (float)(-2 + position), (float)(2 + position), out, cap, n);
}

int ppp_get_slice_indices(ppp_handle h, int s, int *out, size_t cap, size_t *n)
{
    int rc = index_ready(h, false); /* complete index: slabs beyond the LDS capacity go through the arena pass first */
    if (rc) return rc;
    rc = fetch_meta(h);
    if (rc) return rc;
    if (s < 0 || s >= h->hmeta.S) return fail(h, PPP_ERR_ARG, "slice out of range");
    float lohi[2];
    HIPCHK(h, copy_sync(h, &lohi[0], h->lo.p + s, 4, hipMemcpyDeviceToHost));
    HIPCHK(h, copy_sync(h, &lohi[1], h->hi.p + s, 4, hipMemcpyDeviceToHost));
    return band_indices(h, lohi[0], lohi[1], out, cap, n);
}

int ppp_get_nodes(ppp_handle h, int s, double *y, double *x, double *z, size_t cap, size_t *m)
{
    int rc = ensure_ready(h, true, false);
    if (rc) return rc;
    if (s < 0 || s >= h->hmeta.S) return fail(h, PPP_ERR_ARG, "slice out of range");
    if (h->hn_serial != h->gen_serial) { /* first question about this pass: every slice's knots, packed on the device, in one copy */
        const size_t S = (size_t)h->hmeta.S;
        std::vector<int> tab(2 * S + 1);
        int *start = tab.data(), *off = tab.data() + S;
        std::vector<int> cnt(S);
        HIPCHK(h, copy_sync(h, start, h->node_start.p, S * 4, hipMemcpyDeviceToHost));
        HIPCHK(h, copy_sync(h, cnt.data(), h->node_cnt.p, S * 4, hipMemcpyDeviceToHost));
        size_t total = 0;
        for (size_t i = 0; i < S; ++i) {
            /* (a slice-range handle fills the rows of its own slices only: whatever the others hold reads as "no knots") */
            if (start[i] < 0 || cnt[i] < 0 || (size_t)start[i] + (size_t)cnt[i] > (size_t)h->node_cap) { start[i] = 0; cnt[i] = 0; }
            off[i] = (int)total;
            total += (size_t)cnt[i];
        }
        if (total > 0x7fffffffu / 3) return fail(h, PPP_ERR_CAPACITY, "more knots than one copy holds");
        off[S] = (int)total;
        h->hn_off.assign(off, off + S + 1);
        h->hn_xyz.resize(3 * total);
        if (total) {
            HIPCHK(h, h->pack_tab.ensure(2 * S + 1));
            HIPCHK(h, h->pack_out.ensure(3 * total));
            HIPCHK(h, copy_sync(h, h->pack_tab.p, tab.data(), (2 * S + 1) * 4, hipMemcpyHostToDevice));
            LAUNCH(h, "k_nodes_pack", k_nodes_pack, (unsigned)S, 256, 0, h->pack_tab.p, h->pack_tab.p + S, (int)S, (int)total, h->node_x.p, h->node_y.p, h->node_z.p,
                   h->pack_out.p);
            HIPCHK(h, copy_sync(h, h->hn_xyz.data(), h->pack_out.p, 3 * total * 4, hipMemcpyDeviceToHost));
        }
        h->hn_serial = h->gen_serial;
    }
    const size_t st = (size_t)h->hn_off[(size_t)s], cnt = (size_t)(h->hn_off[(size_t)s + 1] - h->hn_off[(size_t)s]), total = (size_t)h->hn_off.back();
    if (m) *m = cnt;
    const size_t k = std::min(cap, cnt);
    for (size_t i = 0; i < k; ++i) {
        if (x) x[i] = (double)h->hn_xyz[st + i];
        if (y) y[i] = (double)h->hn_xyz[total + st + i];
        if (z) z[i] = (double)h->hn_xyz[2 * total + st + i];
    }
    return PPP_OK;
}

int ppp_get_boundary(ppp_handle h, int s, double *y, double *x, double *z, size_t cap, size_t *m, int *step)
{
    int rc = ensure_ready(h, true, false);
    if (rc) return rc;
    if (s < 0 || s >= h->hmeta.S) return fail(h, PPP_ERR_ARG, "slice out of range");
    if (m) *m = 0;
    if (step) *step = -1;
    if (!h->P.dynamic_adjustment) return PPP_OK;
    /* the step of its chain slice s is adjusted in (the chains of enqueue_dynamic) */
    const int walk = h->P.walk;
    int t = -1;
    if (walk == PPP_WALK_CENTER_INT) {
        const int centre = host_centre_index(h);
        t = s < centre ? centre - 1 - s : (s > centre ? s - centre - 1 : -1);
    } else t = s - 1;
    if (step) *step = t;
    if (t < 0) return PPP_OK;
    if (!h->dyn_keep_all || !h->dyn_bnd_n.p) return fail(h, PPP_ERR_CAPACITY, "the boundaries of this pass were not kept (more than 1 GiB of them)");
    int cnt = 0;
    HIPCHK(h, copy_sync(h, &cnt, h->dyn_bnd_n.p + 4 + s, 4, hipMemcpyDeviceToHost));
    if (cnt < 0 || cnt > h->dyn_maxNB + 2) return fail(h, PPP_ERR_HIP, "boundary table corrupt");
    if (m) *m = (size_t)cnt;
    const size_t k = std::min(cap, (size_t)cnt), row = (size_t)h->dyn_maxNB + 2;
    const double *base = h->dyn_bnd_knots.p + (size_t)s * 3 * row;
    if (k && y) HIPCHK(h, copy_sync(h, y, base, k * 8, hipMemcpyDeviceToHost));
    if (k && x) HIPCHK(h, copy_sync(h, x, base + row, k * 8, hipMemcpyDeviceToHost));
    if (k && z) HIPCHK(h, copy_sync(h, z, base + 2 * row, k * 8, hipMemcpyDeviceToHost));
    return PPP_OK;
}

int ppp_eval_spline(ppp_handle h, int s, const double *y, size_t k, double *xyz)
{
    int rc = ensure_ready(h, true, false);
    if (rc) return rc;
    if (s < 0 || s >= h->hmeta.S || !y || !xyz) return fail(h, PPP_ERR_ARG, "bad arguments");
    if (!k) return PPP_OK;
    HIPCHK(h, h->scratch.ensure(k * 32));
    double *dq = (double *)h->scratch.p, *dout = dq + k;
    HIPCHK(h, hipMemcpyAsync(dq, y, k * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(&h->meta.p->api_flag, 0, 4, h->stream));
    LAUNCH(h, "k_eval_api", k_eval_api, (unsigned)((k + 127) / 128), 128, 0, h->meta.p, h->node_x.p, h->node_y.p, h->node_z.p,
           h->node_start.p, h->node_cnt.p, s, dq, (int)k, dout);
    HIPCHK(h, hipMemcpyAsync(xyz, dout, k * 24, hipMemcpyDeviceToHost, h->stream));
    h->meta_in_flight = false; h->meta_fresh = false;
    rc = fetch_meta(h);
    if (rc) return rc;
    if (h->hmeta.api_flag == DERR_DOMAIN) return fail(h, PPP_ERR_DOMAIN, "y outside [miny, bigy] (GSL_EDOM)");
    return PPP_OK;
}

int ppp_insert_point(ppp_handle h, const int *indices, size_t n, float plane_x, double *y, double *x, double *z, size_t cap,
                     size_t *m)
{
    if (!h || (!indices && n)) return PPP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->have_cloud) return fail(h, PPP_ERR_ARG, "no cloud set");
    if (h->part_given) return fail(h, PPP_ERR_UNSUPPORTED, "cloud indices address the whole cloud: this handle holds a part (ppp_set_cloud_part)");
    int capb = 4096;
    if (n > (size_t)capb) return fail(h, PPP_ERR_CAPACITY, "insert_point: more than 4096 indices in one band");
    if (!slice_lds_ok(h, capb)) return fail(h, PPP_ERR_CAPACITY, "LDS too small");
    HIPCHK(h, h->scratch.ensure((size_t)capb * 12));
    int *didx = (int *)h->scratch.p;
    float *dy = (float *)(didx + capb), *dz = dy + capb;
    if (n) HIPCHK(h, hipMemcpyAsync(didx, indices, n * 4, hipMemcpyHostToDevice, h->stream));
    LAUNCH(h, "k_insert_api", k_insert_api, 1, 256, slice_lds_bytes(capb), h->X.p, h->Y.p, h->Z.p, (int)h->n, didx, (int)n, plane_x,
           h->P.pairing, capb, h->meta.p, dy, dz, capb);
    h->meta_in_flight = false; h->meta_fresh = false;
    int rc = fetch_meta(h);
    if (rc) return rc;
    if (h->hmeta.api_flag == DERR_SLICE) return fail(h, PPP_ERR_SLICE, "insert_point: empty right side (the reference crashes here)");
    if (h->hmeta.api_flag) return fail(h, PPP_ERR_CAPACITY, "insert_point capacity");
    size_t cnt = (size_t)h->hmeta.api_cnt;
    if (m) *m = cnt;
    size_t k = std::min(cap, cnt);
    if (k) {
        std::vector<float> fy(k), fz(k);
        HIPCHK(h, copy_sync(h, fy.data(), dy, k * 4, hipMemcpyDeviceToHost));
        HIPCHK(h, copy_sync(h, fz.data(), dz, k * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < k; ++i) {
            if (y) y[i] = (double)fy[i];
            if (x) x[i] = (double)plane_x;
            if (z) z[i] = (double)fz[i];
        }
    }
    return PPP_OK;
}

int ppp_normals_at(ppp_handle h, const int *idx, size_t k, float *out4)
{
    int rc = index_ready(h, false); /* complete index: slabs beyond the LDS capacity go through the arena pass first */
    if (rc) return rc;
    if (h->part_given) return fail(h, PPP_ERR_UNSUPPORTED, "cloud indices address the whole cloud: this handle holds a part (ppp_set_cloud_part)");
    if (!k) return PPP_OK;
    if (!idx || !out4) return fail(h, PPP_ERR_ARG, "bad arguments");
    HIPCHK(h, h->scratch.ensure(k * 20));
    int *didx = (int *)h->scratch.p;
    float *dout = (float *)(didx + k);
    HIPCHK(h, hipMemcpyAsync(didx, idx, k * 4, hipMemcpyHostToDevice, h->stream));
    DevParams D = dev_params(h);
    LAUNCH(h, "k_normals_api", k_normals_api, (unsigned)((k + 63) / 64), 64, 0, h->meta.p, D, h->sorted4.p, h->slab_start.p,
           h->slab_xmin.p, h->slab_xmax.p, h->X.p, h->Y.p, h->Z.p, (int)h->n, didx, (int)k, dout);
    HIPCHK(h, hipMemcpyAsync(out4, dout, k * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PPP_OK;
}

int ppp_estimate_normals(ppp_handle h, float *out4)
{
    int rc = index_ready(h, false); /* complete index: slabs beyond the LDS capacity go through the arena pass first */
    if (rc) return rc;
    if (h->part_given) return fail(h, PPP_ERR_UNSUPPORTED, "the normal field is indexed by the whole cloud: this handle holds a part (ppp_set_cloud_part)");
    if (!h->n) return PPP_OK;
    if (!out4) return fail(h, PPP_ERR_ARG, "bad arguments");
    rc = fetch_meta(h);
    if (rc) return rc;
    const size_t n = h->n;
    HIPCHK(h, h->scratch.ensure(n * 16));
    /* dropped (non-finite) points never enter the index: they keep the NaN fill */
    HIPCHK(h, hipMemsetAsync(h->scratch.p, 0xff, n * 16, h->stream));
    DevParams D = dev_params(h);
    const int nsorted = h->hmeta.n_sorted;
    if (nsorted > 0)
        LAUNCH(h, "k_normals_all", k_normals_all, (unsigned)((nsorted + 255) / 256), 256, 0, h->meta.p, D, h->sorted4.p, h->slab_start.p,
               h->slab_xmin.p, h->slab_xmax.p, h->slab_ytab.p, nsorted, (float4 *)h->scratch.p);
    HIPCHK(h, hipMemcpyAsync(out4, h->scratch.p, n * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PPP_OK;
}

int ppp_area2cloud(ppp_handle h, const double *pts_xyz, size_t k, int key, float *out3)
{
    int rc = index_ready(h, false); /* complete index: slabs beyond the LDS capacity go through the arena pass first */
    if (rc) return rc;
    if (!k) return PPP_OK;
    if (!pts_xyz || !out3 || (key != 0 && key != 1)) return fail(h, PPP_ERR_ARG, "bad arguments");
    if (h->P.curvature_k < 3 || h->P.curvature_k > 64) return fail(h, PPP_ERR_ARG, "curvature_k must be in [3, 64]");
    rc = ensure_dynamic_buffers(h);
    if (rc) return rc;
    rc = enqueue_normals(h);
    if (rc) return rc;
    HIPCHK(h, h->scratch.ensure(k * 36 + 64));
    double *dq = (double *)h->scratch.p;
    float *dout = (float *)(dq + 3 * k);
    HIPCHK(h, hipMemcpyAsync(dq, pts_xyz, k * 24, hipMemcpyHostToDevice, h->stream));
    LAUNCH(h, "k_area2cloud_api", k_area2cloud_api, (unsigned)((k + DYN_WAVES - 1) / DYN_WAVES), 64 * DYN_WAVES, 0, h->meta.p, dyn_params(h),
           h->sorted4.p, h->slab_start.p, h->slab_xmin.p, h->slab_xmax.p, h->normals4.p, h->ell_cs.p, h->slab_ytab.p, dq, (int)k, key, dout);
    HIPCHK(h, hipMemcpyAsync(out3, dout, k * 12, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PPP_OK;
}

int ppp_nearest(ppp_handle h, const float *q_xyz, size_t k, int *idx)
{
    int rc = index_ready(h, false); /* complete index: slabs beyond the LDS capacity go through the arena pass first */
    if (rc) return rc;
    if (!k) return PPP_OK;
    if (!q_xyz || !idx) return fail(h, PPP_ERR_ARG, "bad arguments");
    HIPCHK(h, h->scratch.ensure(k * 16));
    float *dq = (float *)h->scratch.p;
    int *dout = (int *)(dq + 3 * k);
    HIPCHK(h, hipMemcpyAsync(dq, q_xyz, k * 12, hipMemcpyHostToDevice, h->stream));
    LAUNCH(h, "k_nearest_api", k_nearest_api, (unsigned)((k + 63) / 64), 64, 0, h->meta.p, h->sorted4.p, h->slab_start.p, h->slab_xmin.p,
           h->slab_xmax.p, dq, (int)k, dout);
    HIPCHK(h, hipMemcpyAsync(idx, dout, k * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PPP_OK;
}

/* a window pass leaves the per-waypoint stage lists in per-slice slots: into list order when somebody asks */
static int ensure_stage_lists(ppp_handle h)
{
    if (h->stage_compact || !h->win_path) return PPP_OK;
    const WinArgs A = win_args(h);
    if (A.nkept > 0) LAUNCH(h, "k_win_gather_stage", k_win_gather_stage, A.nkept, 256, 0, A, h->wp_xyz.p, h->wp_nn.p, h->wp_normal.p);
    h->stage_compact = true;
    return PPP_OK;
}

int ppp_get_stage(ppp_handle h, int stage, void *out, size_t cap_bytes, size_t *count)
{
    int rc = ensure_ready(h, true, true);
    if (rc) return rc;
    rc = map_dev_err(h);
    if (rc) return rc;
    if (stage == PPP_STAGE_WP_XYZ || stage == PPP_STAGE_WP_NN || stage == PPP_STAGE_WP_NORMAL) { rc = ensure_stage_lists(h); if (rc) return rc; }
    size_t W = (size_t)h->hmeta.W;
    if (count) *count = W;
    if (!out || !cap_bytes || !W) return PPP_OK;
    const void *src = nullptr;
    size_t elem = 0;
    std::vector<float> tmp;
    switch (stage) {
    case PPP_STAGE_WP_XYZ: {
        std::vector<float4> t4(W);
        HIPCHK(h, copy_sync(h, t4.data(), h->wp_xyz.p, W * 16, hipMemcpyDeviceToHost));
        tmp.resize(3 * W);
        for (size_t i = 0; i < W; ++i) { tmp[3 * i] = t4[i].x; tmp[3 * i + 1] = t4[i].y; tmp[3 * i + 2] = t4[i].z; }
        memcpy(out, tmp.data(), std::min(cap_bytes, W * 12));
        return PPP_OK;
    }
    case PPP_STAGE_WP_NN: src = h->wp_nn.p; elem = 4; break;
    case PPP_STAGE_WP_NORMAL: src = h->wp_normal.p; elem = 16; break;
    case PPP_STAGE_WP_PRESMOOTH: src = h->wp_pre.p; elem = 24; break;
    case PPP_STAGE_WP_SMOOTHED: src = h->wp_smooth.p; elem = 24; break;
    default: return fail(h, PPP_ERR_ARG, "unknown stage");
    }
    HIPCHK(h, copy_sync(h, out, src, std::min(cap_bytes, W * elem), hipMemcpyDeviceToHost));
    return PPP_OK;
}

/* ---- class Spline on caller-supplied knots (include/Spline.h:7-51) ---- */
struct ppp_spline_s {
    int device = 0;
    hipStream_t stream = nullptr;
    size_t n = 0;
    double miny = 0, bigy = 0;
    DevBuf<double> knots;   /* y | x | z, n each */
    DevBuf<double> scratch; /* queries, results */
    DevBuf<int> flag;
};

static int spline_fit(ppp_spline sp, size_t n, const double *y, const double *x, const double *z)
{
    if (!sp || !y || !x || !z) return PPP_ERR_ARG;
    if (n < 3 || n > 0x7fffffffu / 8) return PPP_ERR_ARG; /* gsl_spline_alloc: "insufficient number of points for interpolation type" */
    for (size_t i = 1; i < n; ++i) if (!(y[i - 1] < y[i])) return PPP_ERR_ARG; /* gsl_interp_init: "x values must be strictly increasing" */
    if (hipSetDevice(sp->device) != hipSuccess) return PPP_ERR_HIP;
    if (sp->knots.ensure(3 * n) != hipSuccess) return PPP_ERR_HIP;
    hipError_t e = hipMemcpyAsync(sp->knots.p, y, n * 8, hipMemcpyHostToDevice, sp->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(sp->knots.p + n, x, n * 8, hipMemcpyHostToDevice, sp->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(sp->knots.p + 2 * n, z, n * 8, hipMemcpyHostToDevice, sp->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(sp->stream); /* the caller's arrays are free again on return, as with GSL */
    if (e != hipSuccess) return PPP_ERR_HIP;
    sp->n = n; sp->miny = y[0]; sp->bigy = y[n - 1];
    return PPP_OK;
}

int ppp_spline_create(int device_id, size_t n, const double *y, const double *x, const double *z, ppp_spline *out)
{
    if (!out) return PPP_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return PPP_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= count) return PPP_ERR_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return PPP_ERR_HIP;
    ppp_spline sp = new ppp_spline_s();
    sp->device = device_id;
    if (hipStreamCreateWithFlags(&sp->stream, hipStreamNonBlocking) != hipSuccess || sp->flag.ensure(1) != hipSuccess) { (void)ppp_spline_destroy(sp); return PPP_ERR_HIP; }
    int rc = spline_fit(sp, n, y, x, z);
    if (rc != PPP_OK) { (void)ppp_spline_destroy(sp); return rc; }
    *out = sp;
    return PPP_OK;
}

int ppp_spline_restart(ppp_spline sp, size_t n, const double *y, const double *x, const double *z) { return spline_fit(sp, n, y, x, z); }

int ppp_spline_eval(ppp_spline sp, const double *y, size_t k, double *xyz)
{
    if (!sp || sp->n < 3 || (k && (!y || !xyz))) return PPP_ERR_ARG;
    if (!k) return PPP_OK;
    if (k > 0x7fffffffu / 8) return PPP_ERR_CAPACITY; /* the kernel indexes with int, the scratch holds 4 k doubles */
    if (hipSetDevice(sp->device) != hipSuccess) return PPP_ERR_HIP;
    if (sp->scratch.ensure(4 * k) != hipSuccess) return PPP_ERR_HIP;
    double *dq = sp->scratch.p, *dout = dq + k;
    int flag = 0;
    hipError_t e = hipMemcpyAsync(dq, y, k * 8, hipMemcpyHostToDevice, sp->stream);
    if (e == hipSuccess) e = hipMemsetAsync(sp->flag.p, 0, sizeof(int), sp->stream);
    if (e != hipSuccess) return PPP_ERR_HIP;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_spline_eval_d, dim3((unsigned)((k + 127) / 128)), dim3(128), 0, sp->stream, sp->knots.p, sp->knots.p + sp->n,
                       sp->knots.p + 2 * sp->n, (int)sp->n, dq, (int)k, dout, sp->flag.p);
    if (hipGetLastError() != hipSuccess) return PPP_ERR_HIP;
    e = hipMemcpyAsync(xyz, dout, k * 24, hipMemcpyDeviceToHost, sp->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&flag, sp->flag.p, sizeof(int), hipMemcpyDeviceToHost, sp->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(sp->stream);
    if (e != hipSuccess) return PPP_ERR_HIP;
    return flag == DERR_DOMAIN ? PPP_ERR_DOMAIN : PPP_OK;
}

int ppp_spline_range(ppp_spline sp, double *miny, double *bigy, size_t *n)
{
    if (!sp) return PPP_ERR_ARG;
    if (miny) *miny = sp->miny;
    if (bigy) *bigy = sp->bigy;
    if (n) *n = sp->n;
    return PPP_OK;
}

int ppp_spline_destroy(ppp_spline sp)
{
    if (!sp) return PPP_ERR_ARG;
    (void)hipSetDevice(sp->device);
    if (sp->stream) { (void)hipStreamSynchronize(sp->stream); }
    sp->knots.release(); sp->scratch.release(); sp->flag.release();
    if (sp->stream) (void)hipStreamDestroy(sp->stream);
    delete sp;
    return PPP_OK;
}

#ifdef PPP_STAMPS
/* diagnostic build only: 16 x 16 accumulated s_memtime deltas, then reset */
extern "C" int ppp_dbg_stamps(unsigned long long *out)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 256) != hipSuccess) return PPP_ERR_HIP;
    static const unsigned long long zero[256] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero, sizeof(zero)) == hipSuccess ? PPP_OK : PPP_ERR_HIP;
}
#endif

int ppp_smooth_sweeps(ppp_handle h, int *sweeps)
{
    int rc = ensure_ready(h, true, true);
    if (rc) return rc;
    *sweeps = h->hmeta.sweeps;
    return PPP_OK;
}

int ppp_set_plan_reuse(ppp_handle h, int on)
{
    if (!h) return PPP_ERR_ARG;
    if (h->plan_deferred) { HIPCHK(h, hipSetDevice(h->device)); int rcs = settle(h); if (rcs) return rcs; } /* (a cloud set under the old setting) */
    h->plan_reuse = on != 0;
    if (!h->plan_reuse) h->inh_valid = false;
    return PPP_OK;
}

int ppp_set_side_by_side(ppp_handle h, int handles)
{
    if (!h || handles < 1) return PPP_ERR_ARG;
    if (handles == h->side_by_side) return PPP_OK;
    const bool changes = (handles >= 2) != (h->side_by_side >= 2);
    h->side_by_side = handles;
    if (changes && h->have_cloud) {
        HIPCHK(h, hipSetDevice(h->device));
        int rcs = settle(h);
        if (rcs) return rcs;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return make_plan(h);
    }
    return PPP_OK;
}

int ppp_set_fast_path(ppp_handle h, int on)
{
    if (!h) return PPP_ERR_ARG;
    const bool want = on != 0;
    if (want == h->win_allowed) return PPP_OK;
    h->win_allowed = want;
    if (h->have_cloud) {
        HIPCHK(h, hipSetDevice(h->device));
        int rcs = settle(h);
        if (rcs) return rcs;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return make_plan(h);
    }
    return PPP_OK;
}

int ppp_get_fast_path(ppp_handle h, int *active)
{
    if (!h || !active) return PPP_ERR_ARG;
    if (h->have_cloud) { HIPCHK(h, hipSetDevice(h->device)); int rcs = settle(h); if (rcs) return rcs; }
    if (h->have_cloud && !h->planned) { int rc = make_plan(h); if (rc) return rc; }
    *active = (h->have_cloud && h->win_path) ? 1 : 0;
    return PPP_OK;
}

int ppp_enable_timing(ppp_handle h, int on)
{
    if (!h) return PPP_ERR_ARG;
    h->timing = on != 0;
    for (auto &t : h->timers) t.used = 0;
    return PPP_OK;
}

int ppp_get_kernel_times(ppp_handle h, char *names, float *ms, int *launches, size_t cap, size_t *n)
{
    if (!h) return PPP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    size_t k = 0;
    for (auto &t : h->timers) {
        if (!t.used) continue;
        if (k < cap) {
            float v = 0.f;
            for (int q = 0; q < t.used; ++q) {
                float one = 0.f;
                if (hipEventElapsedTime(&one, t.e0[q], t.e1[q]) == hipSuccess) v += one;
            }
            if (ms) ms[k] = v; /* sum over this kernel's launches since the last reset */
            if (launches) launches[k] = t.used;
            if (names) { strncpy(names + 48 * k, t.name.c_str(), 47); names[48 * k + 47] = 0; }
        }
        ++k;
    }
    for (auto &t : h->timers) t.used = 0; /* the next pass starts a fresh sum */
    if (n) *n = k;
    return PPP_OK;
}

} /* extern "C" */
