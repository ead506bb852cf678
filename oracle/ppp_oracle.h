/*
 * ppp_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a CPU restatement of the reference's
 * point-cloud -> tool-path hot path (tsai0507/PolishPathPlanning).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it, and only as the checker.  The product (polishpathplanning_amd/) never
 * links, imports or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or sample
 * data, and cannot be built here (PCL / GSL / Eigen are absent, see
 * DESIGN.md).  The oracle is pinned by independent cross-checks only
 * (tests/test_oracle_*.py).
 */
#ifndef PPP_ORACLE_H
#define PPP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pairing flavour of insert_point */
enum { PPO_PAIR_KD = 0,     /* src/Path_Alg/path_slicing_alg.cpp:164-237 */
       PPO_PAIR_BRUTE = 1   /* src/Path_Generation.cpp:107-206          */ };

/* slice-position walk */
enum { PPO_WALK_SECTPATH = 0,   /* path_slicing_alg.cpp:308-330 float centre-out         */
       PPO_WALK_CENTER_INT = 1, /* path_dynamic_alg.cpp:308-372 int centre-out (connect) */
       PPO_WALK_SDIR_INT = 2,   /* dynamic_alg_sdir.cpp:349-374 int single direction     */
       PPO_WALK_V1_CONTACT = 3, /* Path_Generation.cpp:711-725 float min+R               */
       PPO_WALK_V1_SLICING = 4  /* Path_Generation.cpp:295-304 float min+step/2          */ };

typedef struct ppo_params {
    double tool_radius;      /* Tool_Radius (mm)                               */
    double path_resolution;  /* PathResolution                                 */
    double rpy_resolution;   /* RPYresolution                                  */
    float  ee_length;        /* "End effector length" (m)                      */
    int    change_range;     /* ChangeRange: cloud x1000, waypoints /1000      */
    int    pairing;          /* PPO_PAIR_*                                     */
    int    walk;             /* PPO_WALK_*                                     */
    double trim;             /* 10 (path_translation_alg.cpp:158) or 5 (contour_alg.cpp:496) */
    int    drop_ends;        /* 1: erase first+last spline (path_translation_alg.cpp:149-150) */
    int    smooth;           /* 1: postion_smooth() is applied (:212)          */
    float  handeye[6];       /* x y z rx ry rz (Path_Generate_Algorithm.h:43-48) */
    float  viewpoint[3];     /* PCD VIEWPOINT translation (sensor_origin_)     */
    float  normal_radius;    /* 2.5 (path_slicing_alg.cpp:147)                 */
    int    reference_complexity; /* 1: per-slice O(N) PassThrough scans and whole-cloud
                                    normal estimation exactly where the reference runs them */
    int    smooth_max_sweeps;    /* cap for the smoothing loop (see ppp_oracle.cpp)   */
    /* dynamic adjustment (path_dynamic_alg.cpp:77-306), SURVEY.md 8f rank 1 */
    int    dynamic_adjustment;   /* Dynamic_adjustment (config.txt:13)                */
    double depth;                /* depth             (config.txt:5)                  */
    double adjust_threshold;     /* Adjust_Threshold  (config.txt:3)                  */
    double toolthickness;        /* toolthickness     (config.txt:4)                  */
    int    curvature_k;          /* 50 (path_dynamic_alg.cpp:87); 10 in Path_Generation.cpp:372 */
    int    threads;              /* 1 = as the reference (single-threaded hot path); > 1: OpenMP over the slices and over
                                    the points of the normal estimation, for bench.py's all-cores context figure only */
} ppo_params;

typedef struct ppo_handle ppo_handle;

void ppo_default_params(ppo_params *p);

/* xyz: n points, `stride` floats apart (3 = packed, 8 = pcl::PointXYZRGB). */
ppo_handle *ppo_create(const float *xyz, size_t n, size_t stride, const ppo_params *p);
void ppo_destroy(ppo_handle *h);

size_t ppo_num_points(const ppo_handle *h);
void ppo_get_points(const ppo_handle *h, float *xyz_packed); /* scaled cloud */
void ppo_minmax(const ppo_handle *h, float mn[3], float mx[3]);

/* a3: plane positions in Path_set order (ascending x). returns S (or needed size if > cap) */
int ppo_slice_positions(const ppo_handle *h, float *px, int cap);
/* a4: PassThrough band [position-2, position+2]; returns count (always), fills up to cap */
int ppo_ranged_x_index(const ppo_handle *h, int position, int *out, int cap);
/* a5/a6: returns m >= 0 nodes (ascending y) or < 0 on a reference-crash condition */
int ppo_insert_point(ppo_handle *h, const int *indices, int n, float plane_x,
                     double *y, double *x, double *z, int cap);

/* GenPath: all slices -> splines. returns S >= 0, or -(1+s) when slice s has < 3 nodes
   or an empty side (the reference aborts there). */
int ppo_gen_path(ppo_handle *h);
int ppo_num_slices(const ppo_handle *h);
int ppo_get_nodes(const ppo_handle *h, int s, double *y, double *x, double *z, int cap);
/* the boundary spline (compute_boundary, path_dynamic_alg.cpp:183-235) slice s was adjusted against; returns its knot count, 0 = none */
int ppo_get_boundary(const ppo_handle *h, int s, double *y, double *x, double *z, int cap);
int ppo_get_slice_indices(const ppo_handle *h, int s, int *out, int cap);
/* Spline::point for slice s; returns 0, or -1 if any y is outside [miny, bigy] (GSL_EDOM) */
int ppo_eval_spline(const ppo_handle *h, int s, const double *y, int k, double *xyz);

/* getPath: returns W >= 0 */
int ppo_get_path(ppo_handle *h);
int ppo_num_waypoints(const ppo_handle *h);
void ppo_get_waypoints(const ppo_handle *h, float *out6);
int ppo_get_tail_index(const ppo_handle *h, int *tail, int cap);
/* intermediate stages of getPath (for stage-by-stage parity) */
void ppo_get_waypoints_xyz(const ppo_handle *h, float *xyz);      /* a9: sampled, mm      */
void ppo_get_waypoint_nn(const ppo_handle *h, int *nn);           /* a11: 1-NN ids        */
void ppo_get_waypoint_normals(const ppo_handle *h, float *n4);    /* a10: nx ny nz curv   */
void ppo_get_waypoints_presmooth(const ppo_handle *h, float *o6); /* a12: after HandEye   */
void ppo_get_waypoints_smoothed(const ppo_handle *h, float *o6);  /* a13: after smoothing */
int  ppo_smooth_sweeps(const ppo_handle *h);
int  ppo_rpy_oob(const ppo_handle *h);                            /* B.6 hazard hit       */

/* whole-cloud normal estimation (a10), nx ny nz curvature per point */
void ppo_estimate_normals(ppo_handle *h, float *n4);
void ppo_normal_at(ppo_handle *h, int idx, float n4[4]);
/* SectPath::remove_outlier (path_slicing_alg.cpp:101-108; pcl::StatisticalOutlierRemoval): replaces the cloud, returns
   the new size (or -1); threshold / distances (one float per ORIGINAL point) are optional outputs for the tests */
int ppo_remove_outlier(ppo_handle *h, int mean_k, double std_mul, double *threshold, float *distances);
/* path_generater::voxel_down (Path_Generation.cpp:53-59; pcl::VoxelGrid): replaces the cloud, returns the new size */
int ppo_voxel_down(ppo_handle *h, float lx, float ly, float lz, int *overflow);
/* SectPath::trans2center (path_slicing_alg.cpp:82-99): PCA alignment in place; the handle's get_path then applies
   invTransAlign as path_translation_alg.cpp:146-172 does.  T16 row-major TransAlign; centroid / cov9 as accumulated. */
int ppo_trans2center(ppo_handle *h, float T16[16], float centroid[3], float cov9[9]);
/* Eigen::EigenSolver<Matrix3f>: eigenvalues as they come off the Schur form, eigenvectors in the columns of evecs9 (row-major) */
int ppo_eigensolver3f(const float A9[9], float evals[3], float evecs9[9]);
/* SectPath::smooth (path_slicing_alg.cpp:111-139; pcl::MovingLeastSquares, order 3, radius 15): replaces the cloud */
int ppo_smooth_mls(ppo_handle *h, double radius, int order);
/* dynamic adjustment building blocks (for the cross-check tests) */
/* kdtree.nearestKSearch(q, k): ascending distance; returns the count */
int ppo_knn(ppo_handle *h, const float q[3], int k, int *out);
/* pcl::PrincipalCurvaturesEstimation::computePointPrincipalCurvatures on the k nearest points of q:
   out = pcx pcy pcz pc1 pc2 */
void ppo_principal_curvature(ppo_handle *h, const float q[3], float out[5]);
/* Area2Cloud(point, flag, key) (path_dynamic_alg.cpp:110-180): key 0 = left (min x), 1 = right (max x) */
void ppo_area2cloud(ppo_handle *h, const double p[3], int key, float out[3]);
/* full-cloud kd-tree queries (used by the cross-check tests) */
int ppo_nearest(ppo_handle *h, const float q[3], float *d2);
int ppo_radius_search(ppo_handle *h, const float q[3], float r, int *out, int cap);

/* ---- stateless restatements of the third-party numerics ---- */
/* GSL gsl_interp_steffen: knots xs (strictly increasing), values ys; evaluates at xq[k] */
int ppo_steffen(int n, const double *xs, const double *ys, const double *xq, int k, double *out);
/* pcl::eigen33 smallest-eigenvalue form on a float 3x3 symmetric matrix (row major) */
void ppo_eigen33(const float cov[9], float *eigenvalue, float eigenvector[3]);
/* Eigen Matrix3f::eulerAngles(2,1,0) on a row-major 3x3 */
void ppo_euler_zyx(const float m[9], float e[3]);
/* path_translation_alg.cpp:3-35 on one waypoint (in place) */
void ppo_handeye(const float he[6], float wp[6]);
/* approach/orientation/normal frame + euler of a11 for one normal -> (roll,pitch,yaw) */
void ppo_pose_from_normal(const float n[3], float rpy[3]);
/* path_translation_alg.cpp:114-141 on a list of n waypoints (in place); returns sweeps */
int ppo_position_smooth(float *wp6, int n, int max_sweeps);
/* path_translation_alg.cpp:37-86; returns 1 if the B.6 out-of-range read was hit */
int ppo_reduce_rpy(float *wp6, int n, const int *tail, int ntail, double rpy_res);
/* path_translation_alg.cpp:89-112 */
void ppo_trans_flange(float *wp6, int n, float ee_len);

#ifdef __cplusplus
}
#endif
#endif
