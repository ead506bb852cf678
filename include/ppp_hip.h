/*
 * ppp_hip.h -- C ABI of the MI355X (gfx950) polishing-path engine.
 *
 * This is the drop-in boundary: the reference's planner classes
 * (include/Path_Generate.h, include/Path_Generate_Algorithm.h,
 * include/robot_path.h of tsai0507/PolishPathPlanning) keep their public
 * methods and forward the arithmetic to these entry points; see
 * INTEGRATION.md for the binding a maintainer adds.  Plain C types only, an
 * opaque handle, caller-allocated outputs (two-call size query: pass cap = 0
 * to learn the count), `int` status everywhere (0 = ok, < 0 = error; nothing
 * throws or aborts across the boundary).  One handle = one device + one HIP
 * stream; distinct handles may be used from distinct threads.
 *
 * Each entry point cites the reference code it replaces (file:line relative
 * to the reference tree).
 */
#ifndef PPP_HIP_H
#define PPP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ppp_handle_s *ppp_handle;

/* status codes */
enum {
    PPP_OK = 0,
    PPP_ERR_ARG = -1,          /* bad argument / call order                                  */
    PPP_ERR_HIP = -2,          /* a HIP runtime call failed (ppp_last_error has the text)     */
    PPP_ERR_NO_DEVICE = -3,    /* no gfx950 device: the engine has no CPU fallback            */
    PPP_ERR_SLICE = -4,        /* a slice has an empty side or < 3 nodes: the reference aborts
                                  there; ppp_failed_slice() names it     */
    PPP_ERR_CAPACITY = -5,     /* an internal capacity was exceeded (reported, never silent)  */
    PPP_ERR_DOMAIN = -6,       /* spline evaluated outside [miny, bigy] (GSL_EDOM)            */
    PPP_ERR_UNSUPPORTED = -7,  /* option outside the hot-path scope (e.g. Alignment=true)     */
    PPP_ERR_IO = -8
};

/* insert_point flavour */
enum { PPP_PAIR_KD = 0,      /* src/Path_Alg/path_slicing_alg.cpp:164-237 (connect, connect1, contour) */
       PPP_PAIR_BRUTE = 1    /* src/Path_Generation.cpp:107-206 (./main)                                */ };

/* slice-position walk */
enum { PPP_WALK_SECTPATH = 0,   /* SectPath::GenPath, path_slicing_alg.cpp:308-330        */
       PPP_WALK_CENTER_INT = 1, /* path_generater::GenPath + thread_worker,
                                   path_dynamic_alg.cpp:308-372 (connect)                  */
       PPP_WALK_SDIR_INT = 2,   /* dynamic_alg_sdir.cpp:349-374 (connect1)                */
       PPP_WALK_V1_CONTACT = 3, /* Contact_Path_Generation, Path_Generation.cpp:711-725   */
       PPP_WALK_V1_SLICING = 4  /* slicing_method, Path_Generation.cpp:295-304            */ };

/* All config.txt keys (config.txt:1-13) + the compile-time constants of the reference. */
typedef struct ppp_params {
    double tool_radius;       /* Tool_Radius                                                  */
    double path_resolution;   /* PathResolution                                               */
    double rpy_resolution;    /* RPYresolution                                                */
    float  ee_length;         /* End effector length (m)                                      */
    int    change_range;      /* ChangeRange                                                  */
    int    pairing;           /* PPP_PAIR_*                                                   */
    int    walk;              /* PPP_WALK_*                                                   */
    double trim;              /* 10: path_translation_alg.cpp:158-159; 5: contour_alg.cpp:496 */
    int    drop_ends;         /* 1: path_translation_alg.cpp:149-150                          */
    int    smooth;            /* 1: postion_smooth() applied (path_translation_alg.cpp:212)   */
    float  handeye[6];        /* HANDEYEx..rz (Path_Generate_Algorithm.h:43-48)               */
    float  normal_radius;     /* 2.5 (path_slicing_alg.cpp:147)                               */
    int    smooth_max_sweeps; /* unused by the engine (postion_smooth is solved directly, ppp_kernels.h a13); the
                                 oracle's sequential sweep takes it as its cap     */
    int    alignment;         /* must be 0: Alignment / Smooth / RemoveOutlier are the ppp_trans2center / ppp_smooth_mls / ppp_remove_outlier calls */
    int    dynamic_adjustment;/* Dynamic_adjustment (config.txt:13): path_dynamic_alg.cpp:77-306 for the connect /
                                 connect1 walks, Path_Generation.cpp:362-634 for PPP_WALK_V1_CONTACT            */
    double depth;             /* depth            (config.txt:5)                              */
    double adjust_threshold;  /* Adjust_Threshold (config.txt:3)                              */
    double toolthickness;     /* toolthickness    (config.txt:4)                              */
    int    curvature_k;       /* neighbours of compute_transform: 50 (path_dynamic_alg.cpp:87) */
    /* Slice-range sharding of ONE cloud over several GPUs (SURVEY.md 8e case ii): this handle plans the slices
       [slice_begin, slice_end) of the walk only (slice_end <= 0: up to the last one).  Every handle still takes the
       whole cloud (the walk needs its bounds) but indexes only the points within range_margin mm of its slices.
       getPath then stops after HandEyeTransform (a12): postion_smooth couples neighbouring slices
       (path_translation_alg.cpp:117-140), so it runs once on the gathered list -- ppp_finish_path_async. */
    int    slice_begin, slice_end;
    float  range_margin;      /* mm kept beyond the first/last band of the range (default 24) */
} ppp_params;

void ppp_default_params(ppp_params *p);

/* ---- lifetime ---- */
int ppp_create(int device_id, ppp_handle *out);
int ppp_destroy(ppp_handle h);
const char *ppp_last_error(ppp_handle h);
const char *ppp_version(void);

int ppp_set_params(ppp_handle h, const ppp_params *p);

/* Replaces the constructors' load + scale loop (path_slicing_alg.cpp:10-25,
 * Path_Generation.cpp:8-34, path_dynamic_alg.cpp:12-28): takes the points as read from
 * the PCD (xyz every `stride_bytes`: 32 = pcl::PointXYZRGB, 16, or 12 packed), applies
 * the x1000 when change_range is set, keeps the cloud resident in HBM.
 * viewpoint = PCD VIEWPOINT translation (cloud.sensor_origin_), NULL = origin. */
int ppp_set_cloud(ppp_handle h, const float *xyz_host, size_t n, size_t stride_bytes, const float *viewpoint);
/* same, the buffer already lives on this handle's device */
int ppp_set_cloud_device(ppp_handle h, const float *xyz_dev, size_t n, size_t stride_bytes, const float *viewpoint);
/* same, without waiting for the conversion pass where the handle can do without (see ppp_set_plan_reuse: it holds a window plan
 * of an earlier cloud of this size and these parameters): returns as soon as that pass is enqueued on the handle's stream, and a
 * pass of the new cloud may be enqueued right behind it.  xyz_dev must stay as it is until a call that waits has returned
 * (ppp_sync, any getter) -- or, for a caller that orders its own work after the handle's stream (ppp_get_stream), until that
 * stream has passed this point.  Where the handle cannot do without the bounds this is ppp_set_cloud_device. */
int ppp_set_cloud_device_async(ppp_handle h, const float *xyz_dev, size_t n, size_t stride_bytes, const float *viewpoint);
/* The constructors' first line and their loop in one call (pcl::io::loadPCDFile<PointXYZRGB>(name, *cloud) + the x1000 loop:
 * path_slicing_alg.cpp:10-25, path_dynamic_alg.cpp:12-28, Path_Generation.cpp:8-34): the file's records go straight to HBM.
 * A `DATA binary` file whose x, y, z are consecutive float32 fields is streamed in pieces through two pinned buffers that
 * belong to the handle (several readers fill one while the other is on its way to the device; the conversion kernel then picks
 * x, y, z out of the records: no host copy of the cloud is ever made); every other flavour (ascii, binary_compressed, F8
 * or integer coordinates) goes through ppp_load_pcd + ppp_set_cloud.  *n = the points of the file, viewpoint_out (7 floats,
 * may be NULL) its VIEWPOINT line; the translation part becomes the handle's viewpoint as in ppp_set_cloud. */
int ppp_set_cloud_pcd(ppp_handle h, const char *path, size_t *n, float viewpoint_out[7]);
/* Slice-range sharding without the whole cloud on every GPU (SURVEY.md 8e case ii, pre-partitioned by x).
 * ppp_range_interval: the x interval [lo, hi] -- planner units: the file's values x 1000 (as floats) when ChangeRange -- whose
 * points a handle with p->slice_begin / slice_end / range_margin indexes, for a cloud with the given x bounds (host arithmetic
 * only: the slice walk of path_slicing_alg.cpp:308-330 etc.; +-INFINITY for the whole walk, lo > hi for an empty range);
 * *num_slices (optional) receives S.
 * ppp_set_cloud_part: like ppp_set_cloud, but xyz holds only the n_part points of the cloud whose planner-unit x lies in
 * [part_lo, part_hi], in the cloud's own order (index ties then break as in the whole cloud); cloud_index (optional) = their
 * indices in the whole cloud, so that every index the engine reports is the cloud's; mn / mx / n_valid_total = pcl::getMinMax3D
 * and the number of finite points of the WHOLE cloud in planner units (ranks agree on them with one all-reduce of 3 minima,
 * 3 maxima and a count).  The handle must be given a slice range whose interval lies inside [part_lo, part_hi]; it then plans
 * exactly what a whole-cloud handle with that range plans (same slab grid, bit-identical list).  Calls that address points by
 * whole-cloud index or replace the cloud (ppp_insert_point, ppp_normals_at, ppp_estimate_normals, the preprocessing) are refused. */
int ppp_range_interval(const ppp_params *p, float min_x, float max_x, float *lo, float *hi, int *num_slices);
int ppp_set_cloud_part(ppp_handle h, const float *xyz_host, size_t n_part, size_t stride_bytes, const float *viewpoint,
                       const int *cloud_index, const float mn[3], const float mx[3], size_t n_valid_total, float part_lo, float part_hi);
int ppp_num_points(ppp_handle h, size_t *n);
/* the resident cloud (after the x1000 and any preprocessing) as n x 3 packed floats in index order */
int ppp_get_cloud(ppp_handle h, float *xyz, size_t cap, size_t *n);

/* ---- preprocessing of the constructors, on the resident cloud ---- */
/* SectPath::remove_outlier() (path_slicing_alg.cpp:101-108): pcl::StatisticalOutlierRemoval, setMeanK(mean_k = 50),
 * setStddevMulThresh(stddev_mul = 1.0), sor.filter(*cloud): the filtered cloud replaces the resident one (same order,
 * new indices), the plan is rebuilt.  n_kept / threshold (mean + mul * stddev of the per-point mean neighbour
 * distances) are optional outputs.  mean_k <= 63. */
int ppp_remove_outlier(ppp_handle h, int mean_k, double stddev_mul, size_t *n_kept, double *threshold);
/* path_generater::voxel_down(x, y, z) (Path_Generation.cpp:53-59): pcl::VoxelGrid, setLeafSize(x, y, z), sor.filter(*cloud)
 * with the class defaults: one point per occupied voxel (float centroid of its points), in ascending voxel id; replaces
 * the resident cloud, the plan is rebuilt.  Leaf sizes in the cloud's units (mm after ChangeRange).  *overflow = 1 and
 * the cloud is left as it is where PCL warns "Leaf size is too small ... Integer indices would overflow". */
int ppp_voxel_down(ppp_handle h, float leaf_x, float leaf_y, float leaf_z, size_t *n_out, int *overflow);
/* SectPath::trans2center() (path_slicing_alg.cpp:82-99; v1 Path_Generation.cpp:60-92): centroid and covariance as
 * pcl::compute3DCentroid / pcl::computeCovarianceMatrix accumulate them (float, point after point -- reproduced bit for
 * bit on the device), Eigen::EigenSolver<Matrix3f> eigenvectors (unsorted, as they come), TransAlign = [V^T | -V^T c],
 * pcl::transformPointCloud on the resident cloud.  From then on ppp_get_path applies invTransAlign to the sampled
 * points and looks up the nearest point and its normal in the cloud carried back by the inverse
 * (path_translation_alg.cpp:146-174), until ppp_set_cloud.  Optional outputs: TransAlign (row-major 4 x 4), the
 * centroid, the 3 x 3 accumulated covariance.  PPP_ERR_DOMAIN when the float Schur form keeps a complex pair. */
int ppp_trans2center(ppp_handle h, float *trans_align16, float *centroid3, float *covariance9);
/* SectPath::smooth() (path_slicing_alg.cpp:111-139; v1 Path_Generation.cpp:340-360): pcl::MovingLeastSquares with
 * setPolynomialOrder(order = 3), setSearchRadius(search_radius = 15), SIMPLE projection, no upsampling; the projected
 * points replace the resident cloud (points with fewer than 3 neighbours in the radius, and non-finite points, are
 * not in the output), the plan is rebuilt.  Radius in the cloud's units (mm after ChangeRange); order 0..3 (0 and 1:
 * projection on the local plane only).  The "smooth_<name>" PCD side file of the reference is the caller's to write. */
int ppp_smooth_mls(ppp_handle h, double search_radius, int order, size_t *n_out);

/* ---- whole hot path, asynchronous on the handle's stream ---- */
/* GenPath(): getMinMax3D + slice walk + rangedX_index + insert_point + Spline for every
 * slice (path_slicing_alg.cpp:290-342, path_dynamic_alg.cpp:337-372 with Adjust=false,
 * Path_Generation.cpp:282-304,659-727 without the dynamic adjustment). */
int ppp_gen_path_async(ppp_handle h);
/* getPath(): path_translation_alg.cpp:144-214 (sampling, normals, pose, HandEye, smoothing,
 * reduceRPY, TransFlangeposition); the list stays in HBM. */
int ppp_get_path_async(ppp_handle h);
/* GenPath() followed by getPath() as ONE enqueue: the kernel sequence is captured into a hipGraph
 * the first time and replayed afterwards (one host call per workpiece instead of ~10 launches;
 * this is what keeps a batch of small workpieces from being host-launch-bound).  Falls back to
 * the two plain calls while kernel timing is enabled. */
int ppp_run_async(ppp_handle h);
/* Batched form (SURVEY.md 8b / BASELINE config 3: many small workpieces on ONE GPU): GenPath + getPath of
 * `count` handles of the same device as ONE hipGraph whose branches (one per handle) run side by side --
 * one host call per batch instead of one per workpiece.  When dst_dev is not NULL every branch's last
 * kernel also writes its WayPointsList to dst_dev + 6 * offset_rows[i] (at most cap_rows[i] rows; a longer
 * list is an error of that handle), so the batch lands in one caller-owned device buffer (the RCCL send
 * buffer) without a host round trip.  Graphs are cached in hs[0] (two: a caller may alternate between two
 * destinations) and rebuilt when the handle list, a handle's plan or the destination changes.  Follow with
 * ppp_sync_batch (or any per-handle call, which waits). */
int ppp_run_batch_async(ppp_handle *hs, size_t count, float *dst_dev, const size_t *offset_rows, const size_t *cap_rows);
/* waits for the batch, returns the first handle's error (index in *failed when not NULL) */
int ppp_sync_batch(ppp_handle *hs, size_t count, size_t *failed);
/* Multi-GPU exchange of the finished lists: variable-length gather to `root` over RCCL.  Rank r
 * contributes the counts_rows[r] rows of its WayPointsList; on root, recv_dev (device memory, sum(counts) x 6 floats)
 * receives the blocks back to back in rank order.  One ncclGroup of direct ncclSend / ncclRecv pairs -- point to point
 * over xGMI, no ring -- on the handle's stream (asynchronous; ppp_sync afterwards).  nccl_comm is the caller's
 * ncclComm_t (one rank per process / GPU); librccl is looked up at the first call (dlopen), the engine itself does not
 * link it.  Frameworks that own the communicator (torch.distributed) use their own collective on the list's device
 * pointer instead -- polishpathplanning_amd/robot_path.py.
 * nranks == 1: with nccl_comm == NULL the list is copied to recv_dev; with a (one-rank) communicator it goes through librccl's
 * group -- ncclSend to self + ncclRecv from self -- which is the pre-flight of this exchange on one GPU. */
int ppp_gather_waypoints(ppp_handle h, void *nccl_comm, int rank, int nranks, int root, const size_t *counts_rows, float *recv_dev);
/* the handle's HIP stream (hipStream_t), so a framework can order its own work after the planner's on the GPU
 * (e.g. torch.cuda.ExternalStream + wait_stream before the RCCL gather) instead of waiting on the host */
int ppp_get_stream(ppp_handle h, void **stream);
/* waits for the stream, then reports deferred device-side errors */
int ppp_sync(ppp_handle h);
int ppp_failed_slice(ppp_handle h);

/* ---- results (each call synchronises the handle's stream) ---- */
int ppp_num_slices(ppp_handle h, int *S);
int ppp_num_waypoints(ppp_handle h, size_t *W);
/* WayPointsList: W x 6 floats (x y z roll pitch yaw), what getPath writes to pathFile
 * (path_translation_alg.cpp:216-228) */
int ppp_get_waypoints(ppp_handle h, float *out6, size_t cap, size_t *W);
/* device pointer of the same list (valid until the next ppp_get_path_async / destroy) */
int ppp_get_waypoints_device(ppp_handle h, const float **dptr, size_t *W);
/* copies the list into a caller-owned DEVICE buffer (e.g. a framework tensor used as the
 * send buffer of the RCCL gather); asynchronous on the handle's stream after the count is known */
int ppp_copy_waypoints_to_device(ppp_handle h, float *dst_dev, size_t cap, size_t *W);
/* waypoints of every kept slice in list order (the sizes of the reference's per-slice vectors,
 * path_translation_alg.cpp:156-169); zero for slices outside this handle's slice range */
int ppp_get_waypoint_counts(ppp_handle h, int *counts, size_t cap, size_t *nkept);
/* copies a W x 6 stage list (PPP_STAGE_WP_PRESMOOTH / PPP_STAGE_WP_SMOOTHED) into a caller-owned DEVICE buffer */
int ppp_copy_stage_to_device(ppp_handle h, int stage, float *dst_dev, size_t cap, size_t *W);
/* Second half of getPath on a list assembled elsewhere: postion_smooth, reduceRPY, TransFlangeposition
 * (path_translation_alg.cpp:212-214) over `W` pre-smoothing waypoints in DEVICE memory (the blocks of
 * the slice-range handles concatenated in slice order) with `counts[nkept]` waypoints per kept slice.
 * Needs a handle planned for the same cloud and parameters (any slice range).  The result is read
 * with ppp_get_waypoints / ppp_copy_waypoints_to_device / ppp_get_tail_index as after getPath. */
int ppp_finish_path_async(ppp_handle h, const float *pre6_dev, size_t W, const int *counts, size_t nkept);
/* TailIndex (path_translation_alg.cpp:177,210) */
int ppp_get_tail_index(ppp_handle h, int *tail, size_t cap, size_t *n);

/* pcl::getMinMax3D (path_slicing_alg.cpp:303) */
int ppp_minmax(ppp_handle h, float mn[3], float mx[3]);
/* plane x of every slice in Path_set order */
int ppp_get_slice_positions(ppp_handle h, float *px, size_t cap, size_t *S);
/* rangedX_index result of slice s, ascending cloud indices */
int ppp_get_slice_indices(ppp_handle h, int s, int *out, size_t cap, size_t *n);
/* Spline knots of slice s (ascending y): what OnePath / path_track feed to Spline() */
int ppp_get_nodes(ppp_handle h, int s, double *y, double *x, double *z, size_t cap, size_t *m);
/* Dynamic adjustment only: the boundary spline slice s was adjusted against -- compute_boundary(pre_path, boundary, key) of
   thread_worker (path_dynamic_alg.cpp:183-235, 320-322; v1: Path_Generation.cpp:585-592), the curve drawpath(*boundary, 0,255,0)
   paints: knots in ascending y with the two end knots 20 mm out.  *m = 0 where compute_boundary returned 0 at that step (fewer than
   3 boundary points: nothing is painted, the slice is adjusted against the chain's previous boundary) and for the slice a chain starts
   from.  *step = the slice's step in its chain (0 = next to the start slice; -1 = the start slice itself, or no dynamic adjustment):
   the order thread_worker adjusts -- and paints -- in.  Evaluate with ppp_spline_create / ppp_spline_eval below. */
int ppp_get_boundary(ppp_handle h, int s, double *y, double *x, double *z, size_t cap, size_t *m, int *step);
/* Spline::point(y) of slice s (include/Spline.h:22-25): xyz[3*i..] */
int ppp_eval_spline(ppp_handle h, int s, const double *y, size_t k, double *xyz);

/* ---- class Spline on caller-supplied knots (include/Spline.h:7-51) ----
 * The planner's own splines live with their slice (ppp_get_nodes / ppp_eval_spline above); these entry points serve callers
 * that construct a Spline themselves, as the reference's OnePath (path_slicing_alg.cpp:240-267), path_track
 * (Path_Generation.cpp:659-687) and dynamic_adjust_path (path_dynamic_alg.cpp:297-303, Spline::restart) do.
 * Knots and evaluation are double, as in GSL; the object owns its device copy of the knots and a HIP stream. */
typedef struct ppp_spline_s *ppp_spline;
/* Spline(int number, const double* point_y, const double* point_x, const double* point_z) (Spline.h:10-20): two
 * gsl_interp_steffen splines y -> x and y -> z.  Where GSL raises GSL_EINVAL and aborts -- fewer than 3 knots
 * (gsl_spline_alloc, steffen min_size 3) or y not strictly increasing (gsl_interp_init) -- PPP_ERR_ARG comes back. */
int ppp_spline_create(int device_id, size_t n, const double *y, const double *x, const double *z, ppp_spline *out);
/* Spline::restart (Spline.h:30-42): re-fit the same object on new knots */
int ppp_spline_restart(ppp_spline sp, size_t n, const double *y, const double *x, const double *z);
/* Spline::point(y) for k values (Spline.h:22-25): xyz[3*i..] = (splineYX(y), y, splineYZ(y)); a y outside [miny, bigy]
 * gives NaNs and PPP_ERR_DOMAIN (GSL_EDOM; GSL's default handler aborts there) */
int ppp_spline_eval(ppp_spline sp, const double *y, size_t k, double *xyz);
/* miny() / bigy() (Spline.h:27-28) and the knot count */
int ppp_spline_range(ppp_spline sp, double *miny, double *bigy, size_t *n);
int ppp_spline_destroy(ppp_spline sp);

/* ---- single-call mirrors of the reference's public methods ---- */
/* rangedX_index(int position) (path_slicing_alg.cpp:152-162, Path_Generation.cpp:94-104) */
int ppp_ranged_x_index(ppp_handle h, int position, int *out, size_t cap, size_t *n);
/* insert_point(indices, PlanePoint) (path_slicing_alg.cpp:164-237 / Path_Generation.cpp:107-206);
 * returns the MAP flattened in key order */
int ppp_insert_point(ppp_handle h, const int *indices, size_t n, float plane_x,
                     double *y, double *x, double *z, size_t cap, size_t *m);
/* estimate_normal() (path_slicing_alg.cpp:141-150) evaluated at the given cloud indices:
 * out4 = nx ny nz curvature */
int ppp_normals_at(ppp_handle h, const int *idx, size_t k, float *out4);
/* estimate_normal() over the WHOLE cloud (path_slicing_alg.cpp:141-150, Path_Generation.cpp:323-333):
 * out4 = n x 4 floats (nx ny nz curvature) in cloud index order; NaN where PCL gives NaN
 * (< 3 neighbours, dropped points).  SURVEY.md 8f rank 2. */
int ppp_estimate_normals(ppp_handle h, float *out4);
/* Area2Cloud(point, flag, key) of the dynamic adjustment (path_dynamic_alg.cpp:110-180) for k points
 * (xyz doubles, mm): key 0 = left boundary point (min x of the contact ellipse), 1 = right (max x);
 * out3 = k x 3 floats (NaN where the reference gets NaN) */
int ppp_area2cloud(ppp_handle h, const double *pts_xyz, size_t k, int key, float *out3);
/* kdtree.nearestKSearch(q, 1) on the whole cloud for k query points */
int ppp_nearest(ppp_handle h, const float *q_xyz, size_t k, int *idx);

/* ---- intermediate stages of getPath, for stage-by-stage parity tests ---- */
enum { PPP_STAGE_WP_XYZ = 0,      /* W x 3 float: sampled points, mm (path_translation_alg.cpp:156-169) */
       PPP_STAGE_WP_NN = 1,       /* W int: nearest cloud index (:189)                                  */
       PPP_STAGE_WP_NORMAL = 2,   /* W x 4 float: normal + curvature (:190)                             */
       PPP_STAGE_WP_PRESMOOTH = 3,/* W x 6 float: after HandEyeTransform (:208)                         */
       PPP_STAGE_WP_SMOOTHED = 4  /* W x 6 float: after postion_smooth (:212)                           */ };
int ppp_get_stage(ppp_handle h, int stage, void *out, size_t cap_bytes, size_t *count);
/* sweeps of postion_smooth: always 0 -- the engine solves the sweeps' fixed point directly (one launch) */
int ppp_smooth_sweeps(ppp_handle h, int *sweeps);

/* ---- host-side file formats of the reference (no device work) ---- */
/* pcl::io::loadPCDFile<PointXYZRGB> (path_slicing_alg.cpp:10, Path_Generation.cpp:8): PCD v0.7,
 * DATA ascii | binary | binary_compressed (LZF, field-major), fields matched by name, x y z as F4 or F8.
 * *xyz receives n x 3 packed floats allocated by the library (release with ppp_free); viewpoint = the 7
 * VIEWPOINT numbers (tx ty tz qw qx qy qz). */
int ppp_load_pcd(const char *path, float **xyz, size_t *n, float viewpoint[7]);
/* the header alone: how the file stores its points (what ppp_set_cloud_pcd decides on) */
typedef struct ppp_pcd_layout {
    int data_kind;                    /* 0 = ascii, 1 = binary, 2 = binary_compressed */
    size_t points;                    /* POINTS (or WIDTH x HEIGHT) */
    size_t record_bytes;              /* bytes of one point's record: the sum of SIZE x COUNT over FIELDS */
    int x_offset, y_offset, z_offset; /* where x, y, z sit in it */
    int xyz_float32;                  /* 1 when all three are F 4 */
    long long data_offset;            /* the byte after the DATA line */
    float viewpoint[7];
} ppp_pcd_layout;
int ppp_pcd_probe(const char *path, ppp_pcd_layout *layout);
/* binary: 0 = ascii (pcl::io::savePCDFileASCII, path_slicing_alg.cpp:138), 1 = binary, 2 = binary_compressed */
int ppp_save_pcd(const char *path, const float *xyz, size_t n, size_t stride_floats, const float viewpoint[7], int binary);
/* a pcl::PointXYZRGB cloud (FIELDS x y z rgb, colour packed 0x00RRGGBB): what show() would hand to the viewer
 * (path_slicing_alg.cpp:69-80); xyz = n x 3 packed floats, rgb = n x 3 bytes; binary: 0 = ascii, 1 = binary */
int ppp_save_pcd_rgb(const char *path, const float *xyz, const unsigned char *rgb, size_t n, const float viewpoint[7], int binary);
void ppp_free(void *p);
/* SectPath::read_config / path_generater::read_config (path_slicing_alg.cpp:32-67,
 * path_dynamic_alg.cpp:34-75): same key set and parsing rules; absent keys keep config.txt's values */
typedef struct ppp_config {
    ppp_params params;
    char path_file[512];      /* pathFile */
    double depth, adjust_threshold, toolthickness;
    int smooth_cloud, remove_outlier, alignment, dynamic_adjustment; /* parsed, outside the hot path */
} ppp_config;
void ppp_default_config(ppp_config *c);
int ppp_read_config(const char *path, ppp_config *c);
/* pathFile writer (path_translation_alg.cpp:216-228): "x y z r p y " per line, ostream defaults */
int ppp_write_path_file(const char *path, const float *wp6, size_t W);

/* ---- a planner queue: workpieces in, lists out, a few handles behind it taking turns ----
 * A pass is three dependent launches that leave most of the chip idle most of the time; passes enqueued on different handles (HIP
 * streams) overlap.  The queue keeps `lanes` handles (0 = 2: best for a stream of new clouds, 50 us per 1 M-point cloud against 88 on one
 * lane and 62 on three; replays of resident clouds do best on three handles, four and more collide on the runtime's hardware queues)
 * with the given parameters and gives workpiece k to lane k mod lanes: ppp_queue_submit waits for that lane's earlier pass,
 * takes the cloud -- already in device memory, untouched until ppp_queue_wait of this ticket has returned -- without waiting for
 * its bounds where the lane's plan allows (ppp_set_cloud_device_async) and enqueues its pass; ppp_queue_wait returns the finished
 * list (device pointer into the lane, W rows of 6 floats; valid until `lanes` more workpieces have been submitted) or the error the
 * workpiece ended with.  Everything a caller could do with the handles itself (ppp_queue_lane gives them out, e.g. for
 * ppp_get_waypoints of the lane that holds a ticket); not thread-safe: one submitting thread. */
typedef struct ppp_queue_s *ppp_queue;
int ppp_queue_create(int device, int lanes, const ppp_params *params, ppp_queue *q);
void ppp_queue_destroy(ppp_queue q);
int ppp_queue_submit(ppp_queue q, const float *xyz_dev, size_t n, size_t stride_bytes, const float *viewpoint, long long *ticket);
int ppp_queue_wait(ppp_queue q, long long ticket, size_t *W, const float **list_dev);
int ppp_queue_lanes(ppp_queue q);
ppp_handle ppp_queue_lane(ppp_queue q, int i);
const char *ppp_queue_last_error(ppp_queue q);

/* ---- which launch sequence plans a cloud ---- */
/* Plan reuse (default on).  The first cloud of a handle sizes the window path's LDS capacities from a census of its own windows
 * (one more launch behind the conversion pass of ppp_set_cloud*).  A later cloud with the same point count and parameters, whose
 * slice walk has the same length and pad, inherits those capacities (+4 %) and skips that launch -- a planner that is fed one scan
 * after the other (the constructors of the reference's planner classes: src/Path_Alg/path_slicing_alg.cpp:3-30) saves it on every
 * workpiece.  The pass checks every window against its capacity on the device; one that does not fit makes the engine plan this
 * cloud again from its own census and repeat the pass (same results, one wasted pass).
 * Such a later cloud need not be waited for either: while the handle holds that window plan, ppp_set_cloud_device_async and
 * ppp_set_cloud_pcd (the calls whose points no caller takes back at once) only enqueue the conversion pass and return, and ppp_run_async / ppp_gen_path_async / ppp_get_path_async put the new cloud's pass behind it on
 * the earlier plan; bounds, walk and capacities are checked on the device against the record the conversion pass leaves, and any
 * other call (ppp_sync included) first reads that record and plans the cloud as a waiting ppp_set_cloud* would have -- a pass the
 * plan did not fit is repeated by the engine.  An error of the new cloud itself (no finite point, ...) then surfaces at that
 * call instead of at the call that set the cloud.  0 turns both off (PPP_NO_DEFERRED_PLAN=1 in the environment: the waiting only). */
int ppp_set_plan_reuse(ppp_handle h, int on);
/* How many handles the caller runs side by side on this device (default 1).  From two on the plan trades a little of a pass's own
 * latency for room on the CUs: the per-slice workgroups of small windows stay at 512 threads, so that a neighbouring pass's binning
 * and finish workgroups fit beside them (1 M points / 256 slices on three handles: 0.038 -> 0.029 ms per workpiece; one pass alone
 * 0.065 -> 0.068 ms).  Same lists either way.  ppp_queue_create sets it on its lanes. */
int ppp_set_side_by_side(ppp_handle h, int handles);
/* The engine has two launch sequences for the same hot path, with the same results up to the last bits of the normals'
 * float sums (both within the tolerances of tests/): the WINDOW path (three launches: every point binned once into the
 * window of its slice, one fused per-slice kernel, the finish; kd pairing, no dynamic adjustment / alignment, tool steps
 * wide enough that the slices' +-pad windows do not overlap, windows that fit a workgroup's LDS) and the SLAB-INDEX path
 * (six launches, any parameters; also what the single-call mirrors and the dynamic adjustment search).  The plan picks the
 * window path whenever it applies; a pass whose windows overflow or whose searches reach beyond them is repeated on the
 * slab-index path by itself.  ppp_set_fast_path(h, 0) keeps a handle on the slab-index path (tests, comparisons);
 * ppp_get_fast_path tells which one the current plan uses. */
int ppp_set_fast_path(ppp_handle h, int on);
int ppp_get_fast_path(ppp_handle h, int *active);

/* ---- measurement ---- */
/* When enabled every kernel launch is bracketed by hipEvents on the handle's stream. */
int ppp_enable_timing(ppp_handle h, int on);
/* names: cap entries of 48 chars; ms: time of each kernel since the last call, summed over its
 * launches; launches (may be NULL): how many launches that sum covers */
int ppp_get_kernel_times(ppp_handle h, char *names, float *ms, int *launches, size_t cap, size_t *n);

#ifdef __cplusplus
}
#endif
#endif
