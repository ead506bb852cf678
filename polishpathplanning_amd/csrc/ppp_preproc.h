/*
 * ppp_preproc.h -- cloud preprocessing of the planner constructors, on the resident cloud.
 *
 * remove_outlier: SectPath::remove_outlier (path_slicing_alg.cpp:101-108) = pcl::StatisticalOutlierRemoval with
 * setMeanK(50), setStddevMulThresh(1.0) (PCL 1.12 filters/impl/statistical_outlier_removal.hpp, applyFilterIndices):
 *   per finite point the mean of the distances to its mean_k nearest neighbours (the point itself excluded; float sqrt,
 *   double sum in ascending distance order, cast to float), mean and (n-1) variance of those floats over the cloud in
 *   double, threshold = mean + mul * stddev, a point is dropped when its mean distance is above the threshold;
 *   non-finite points carry 0 and stay.  The cloud keeps its order.
 * One wave per point for the k-NN (wave_knn of ppp_dynamic.h on the x-slab index).
 */
#pragma once
#include "ppp_dynamic.h"

struct SorStats { double sum, sq_sum, threshold; int valid, n_kept; };

__global__ void __launch_bounds__(64 * DYN_WAVES) k_sor_dist(DevMeta *m, const float4 *__restrict__ sorted4, const int *__restrict__ slab_start,
                                                             const float *__restrict__ slab_xmin, const float *__restrict__ slab_xmax,
                                                             int mean_k, float r0, float *dist)
{
    __shared__ DynWaveLds s_w[DYN_WAVES];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * DYN_WAVES + wv;
    if (m->err || p >= m->n_sorted) return;
    StampCtx sc; sc.begin(15, false);
    SlabView V{sorted4, slab_start, slab_xmin, slab_xmax, m, nullptr, 0, 0};
    const float4 q = sorted4[p];
    const DynGrid G = dyn_grid(m);
    float no_nn[3];
    const int kk = wave_knn(V, G, s_w[wv], q.x, q.y, q.z, mean_k + 1, r0, nullptr, no_nn, sc);
    if (kk < mean_k + 1) { if (lane == 0) set_err(m, DERR_QUERY, -1); return; } /* PCL would read past its neighbour vectors */
    float sd = 0.f;
    if (lane < kk) {
        const float4 c = V.at(s_w[wv].sel[lane]);
        sd = sqrtf(dist2_flann(q.x, q.y, q.z, c.x, c.y, c.z));
    }
    /* for (k = 1; k < mean_k + 1; ++k) dist_sum += sqrt(nn_dists[k]): the additions in the reference's order */
    double s = 0.0;
    for (int r = 1; r <= mean_k; ++r) s += (double)__shfl(sd, r, 64);
    if (lane == 0) dist[idx_of(q)] = (float)(s / mean_k);
}

/* sum and sum of squares of the per-point means: fixed-order two-stage reduction (deterministic run to run) */
__global__ void __launch_bounds__(256) k_sor_partial(const float *__restrict__ dist, int n, double *part)
{
    __shared__ double s_a[4], s_b[4];
    double a = 0, b = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float d = dist[i];
        a += (double)d;
        b += (double)(d * d); /* float product, as `sq_sum += distance * distance` */
    }
    a = wave_sum(a); b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]);
        part[2 * blockIdx.x + 1] = (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]);
    }
}

__global__ void __launch_bounds__(256) k_sor_threshold(const DevMeta *m, const double *__restrict__ part, int nparts, double std_mul, SorStats *st)
{
    __shared__ double s_a[4], s_b[4];
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) { a += part[2 * i]; b += part[2 * i + 1]; }
    a = wave_sum(a); b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double sum = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]), sq_sum = (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]);
        const double valid = (double)m->n_sorted;
        const double mean = sum / valid;
        const double variance = (sq_sum - sum * sum / valid) / (valid - 1);
        st->sum = sum; st->sq_sum = sq_sum; st->valid = m->n_sorted;
        st->threshold = mean + std_mul * sqrt(variance);
    }
}

/* ordered compaction of the kept points: per-block counts, scan of the counts, scatter */
#define SOR_CHUNK 1024
__global__ void __launch_bounds__(256) k_sor_count(const float *__restrict__ dist, int n, const SorStats *st, int *block_cnt)
{
    __shared__ int s_c[4];
    const double thr = st->threshold;
    int c = 0;
    for (int i = blockIdx.x * SOR_CHUNK + threadIdx.x; i < min(n, (blockIdx.x + 1) * SOR_CHUNK); i += blockDim.x) c += !((double)dist[i] > thr);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

__global__ void __launch_bounds__(1024) k_sor_scan(int *block_cnt, int nblocks, SorStats *st)
{
    __shared__ int s_scr[17];
    __shared__ int s_run;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const int c = i < nblocks ? block_cnt[i] : 0;
        int tot;
        const int pre = block_exscan(c, s_scr, &tot);
        const int run = s_run;
        if (i < nblocks) block_cnt[i] = run + pre;
        __syncthreads();
        if (threadIdx.x == 0) s_run = run + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) st->n_kept = s_run;
}

__global__ void __launch_bounds__(256) k_sor_compact(const float *__restrict__ dist, int n, const SorStats *st, const int *__restrict__ block_off,
                                                     const float *__restrict__ X, const float *__restrict__ Y, const float *__restrict__ Z,
                                                     float *X2, float *Y2, float *Z2)
{
    __shared__ int s_scr[17];
    __shared__ int s_run;
    const double thr = st->threshold;
    if (threadIdx.x == 0) s_run = block_off[blockIdx.x];
    __syncthreads();
    const int i0 = blockIdx.x * SOR_CHUNK, i1 = min(n, i0 + SOR_CHUNK);
    for (int base = i0; base < i1; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const int keep = (i < i1) && !((double)dist[i] > thr);
        int tot;
        const int pre = block_exscan(keep, s_scr, &tot);
        const int run = s_run;
        if (keep) { X2[run + pre] = X[i]; Y2[run + pre] = Y[i]; Z2[run + pre] = Z[i]; }
        __syncthreads();
        if (threadIdx.x == 0) s_run = run + tot;
        __syncthreads();
    }
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* voxel_down: path_generater::voxel_down (Path_Generation.cpp:53-59) = pcl::VoxelGrid<PointXYZRGB> with setLeafSize, */
/* filter(*cloud) and the class defaults (downsample_all_data_, min_points_per_voxel_ = 0, no filter field) -- PCL    */
/* filters/impl/voxel_grid.hpp applyFilter: voxel id of every finite point from the float expressions below, points   */
/* ordered by id, one output point per occupied voxel = float sum of its points / float count (CentroidPoint's        */
/* AccumulatorXYZ), output in ascending id.  PCL orders with an unstable sort, so the summation order inside a voxel   */
/* is an artefact of its sort; here it is ascending point index (stable radix sort), as in the oracle.                */
/* Launches: key -> radix sort of (id, index) [ppp_sort.hip] -> head count / scan -> gather + sequential run sums.    */
/* ------------------------------------------------------------------------------------------------------------------ */
struct VoxGrid { float inv[3]; float min_b[3]; int mul[3]; unsigned none; };
struct VoxStats { int n_out; };

__global__ void __launch_bounds__(256) k_vox_key(const float *__restrict__ X, const float *__restrict__ Y, const float *__restrict__ Z, int n,
                                                 VoxGrid g, unsigned *key, int *idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = X[i], y = Y[i], z = Z[i];
    unsigned k = g.none;
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
        /* ijk0 = static_cast<int>(std::floor(x * inverse_leaf_size_[0]) - static_cast<float>(min_b_[0])) */
        const int i0 = (int)(floorf(x * g.inv[0]) - g.min_b[0]);
        const int i1 = (int)(floorf(y * g.inv[1]) - g.min_b[1]);
        const int i2 = (int)(floorf(z * g.inv[2]) - g.min_b[2]);
        k = (unsigned)(i0 * g.mul[0] + i1 * g.mul[1] + i2 * g.mul[2]);
    }
    key[i] = k;
    idx[i] = i;
}

#define VOX_CHUNK 1024
__global__ void __launch_bounds__(256) k_vox_count(const unsigned *__restrict__ key, int n, unsigned none, int *block_cnt)
{
    __shared__ int s_c[4];
    int c = 0;
    for (int i = blockIdx.x * VOX_CHUNK + threadIdx.x; i < min(n, (blockIdx.x + 1) * VOX_CHUNK); i += blockDim.x) {
        const unsigned k = key[i];
        c += (k != none) && (i == 0 || key[i - 1] != k);
    }
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

__global__ void __launch_bounds__(1024) k_vox_scan(int *block_cnt, int nblocks, VoxStats *st)
{
    __shared__ int s_scr[17];
    __shared__ int s_run;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const int c = i < nblocks ? block_cnt[i] : 0;
        int tot;
        const int pre = block_exscan(c, s_scr, &tot);
        const int run = s_run;
        if (i < nblocks) block_cnt[i] = run + pre;
        __syncthreads();
        if (threadIdx.x == 0) s_run = run + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) st->n_out = s_run;
}

/* the points in sorted order, so that a run is contiguous in memory for the sequential sums below */
__global__ void __launch_bounds__(256) k_vox_gather(const float *__restrict__ X, const float *__restrict__ Y, const float *__restrict__ Z,
                                                    const int *__restrict__ idx, int n, float4 *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = idx[i];
    out[i] = make_float4(X[j], Y[j], Z[j], 0.f);
}

/* one thread per occupied voxel: centroid.add(point) in order, then xyz / n */
__global__ void __launch_bounds__(256) k_vox_reduce(const unsigned *__restrict__ key, const float4 *__restrict__ pts, int n, unsigned none,
                                                    const int *__restrict__ block_off, float *X2, float *Y2, float *Z2)
{
    __shared__ int s_scr[17];
    __shared__ int s_run;
    if (threadIdx.x == 0) s_run = block_off[blockIdx.x];
    __syncthreads();
    const int i0 = blockIdx.x * VOX_CHUNK, i1 = min(n, i0 + VOX_CHUNK);
    for (int base = i0; base < i1; base += blockDim.x) {
        const int i = base + threadIdx.x;
        unsigned k = none;
        int head = 0;
        if (i < i1) { k = key[i]; head = (k != none) && (i == 0 || key[i - 1] != k); }
        int tot;
        const int pre = block_exscan(head, s_scr, &tot);
        const int run = s_run;
        if (head) {
            float sx = 0.f, sy = 0.f, sz = 0.f;
            int j = i;
            do { const float4 p = pts[j]; sx += p.x; sy += p.y; sz += p.z; ++j; } while (j < n && key[j] == k);
            const float c = (float)(j - i);
            X2[run + pre] = sx / c; Y2[run + pre] = sy / c; Z2[run + pre] = sz / c;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_run = run + tot;
        __syncthreads();
    }
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* smooth: SectPath::smooth (path_slicing_alg.cpp:111-139; v1 Path_Generation.cpp:340-360) = pcl::MovingLeastSquares  */
/* with setPolynomialOrder(3), setSearchRadius(15), SIMPLE projection, no upsampling; the projected points replace    */
/* the cloud.  PCL surface/impl/mls.hpp: per point the neighbours within the radius (fewer than 3: the point is not   */
/* in the output), their mean and covariance in double (shifted by the first neighbour = the point itself), the plane */
/* normal by pcl::eigen33<double>, the point projected on the plane (`mean`), weights exp(-d^2 / r^2) about `mean`,    */
/* a polynomial of the given order in the plane's (u, v) frame fitted by the weighted normal equations with Eigen's   */
/* LLT, result = mean + c[0] * normal.                                                                                */
/* One thread per point, walked in slab order (neighbouring lanes scan the same windows); two passes over the          */
/* neighbourhood, no neighbour list.  f64 throughout: the products of ~300 neighbours per point are summed in slab     */
/* order here and in distance order in PCL, and Eigen's blocked products group them differently again, so the double   */
/* results agree to rounding (1e-13 relative) and the float outputs are identical but for a 1-ulp case per ~1e5.       */
/* ------------------------------------------------------------------------------------------------------------------ */
__device__ inline void pcl_roots2_d(double b, double c, double roots[3])
{
    roots[0] = 0.0;
    double d = b * b - 4.0 * c;
    if (d < 0.0) d = 0.0;
    const double sd = sqrt(d);
    roots[2] = 0.5 * (b + sd);
    roots[1] = 0.5 * (b - sd);
}
__device__ inline void pcl_roots_d(const double m[3][3], double roots[3])
{
    const double c0 = m[0][0] * m[1][1] * m[2][2] + 2.0 * m[0][1] * m[0][2] * m[1][2] -
                      m[0][0] * m[1][2] * m[1][2] - m[1][1] * m[0][2] * m[0][2] - m[2][2] * m[0][1] * m[0][1];
    const double c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] - m[0][2] * m[0][2] +
                      m[1][1] * m[2][2] - m[1][2] * m[1][2];
    const double c2 = m[0][0] + m[1][1] + m[2][2];
    if (fabs(c0) < 2.220446049250313e-16) { pcl_roots2_d(c2, c1, roots); return; }
    const double s_inv3 = 1.0 / 3.0, s_sqrt3 = sqrt(3.0);
    const double c2_over_3 = c2 * s_inv3;
    double a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0) a_over_3 = 0.0;
    const double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
    double q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0) q = 0.0;
    const double rho = sqrt(-a_over_3);
    const double theta = atan2(sqrt(-q), half_b) * s_inv3;
    const double cos_theta = cos(theta), sin_theta = sin(theta);
    roots[0] = c2_over_3 + 2.0 * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    double t;
    if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
    if (roots[1] >= roots[2]) {
        t = roots[1]; roots[1] = roots[2]; roots[2] = t;
        if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
    }
    if (roots[0] <= 0) pcl_roots2_d(c2, c1, roots);
}
__device__ inline void pcl_eigen33_smallest_d(const double cov[9], double ev[3])
{
    double scale = 0.0;
    for (int i = 0; i < 9; ++i) scale = fmax(scale, fabs(cov[i]));
    if (scale <= 2.2250738585072014e-308) scale = 1.0;
    double m[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m[i][j] = cov[3 * i + j] / scale;
    double roots[3];
    pcl_roots_d(m, roots);
    m[0][0] -= roots[0]; m[1][1] -= roots[0]; m[2][2] -= roots[0];
    double cp[3][3];
    auto cross = [](const double a[3], const double b[3], double o[3]) {
        o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
    };
    cross(m[0], m[1], cp[0]);
    cross(m[0], m[2], cp[1]);
    cross(m[1], m[2], cp[2]);
    double len[3];
    for (int i = 0; i < 3; ++i) len[i] = sqrt(cp[i][0] * cp[i][0] + cp[i][1] * cp[i][1] + cp[i][2] * cp[i][2]);
    int idx = 0;
    if (len[1] > len[idx]) idx = 1;
    if (len[2] > len[idx]) idx = 2;
    for (int d = 0; d < 3; ++d) ev[d] = cp[idx][d] / len[idx];
}

/* out4[cloud index] = (x, y, z, 1.0) for the points MLS keeps; the buffer is zero-filled before the launch */
template <int ORDER>
__global__ void __launch_bounds__(256) k_mls(DevMeta *m, const float4 *__restrict__ sorted4, const int *__restrict__ slab_start,
                                             const float *__restrict__ slab_xmin, const float *__restrict__ slab_xmax,
                                             float radius, double sqr_gauss, float4 *out4)
{
    constexpr int NC = (ORDER + 1) * (ORDER + 2) / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (m->err || i >= m->n_sorted) return;
    const int B = m->B;
    const float4 p = sorted4[i];
    const float r2 = radius * radius;
    /* visit(c) for every indexed point within the radius (FLANN: dist <= radius^2 in float) */
    auto for_each_neighbour = [&](auto visit) {
        auto scan_slab = [&](int b) {
            const int s0 = slab_start[b], s1 = slab_start[b + 1];
            if (s0 >= s1) return;
            int lo = s0, hi = s1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (sorted4[mid].y < p.y) lo = mid + 1; else hi = mid; }
            for (int j = lo; j < s1; ++j) {
                const float4 c = sorted4[j];
                const float dy = p.y - c.y;
                if (dy * dy > r2) break;
                if (dist2_flann(p.x, p.y, p.z, c.x, c.y, c.z) <= r2) visit(c);
            }
            for (int j = lo - 1; j >= s0; --j) {
                const float4 c = sorted4[j];
                const float dy = p.y - c.y;
                if (dy * dy > r2) break;
                if (dist2_flann(p.x, p.y, p.z, c.x, c.y, c.z) <= r2) visit(c);
            }
        };
        const int b = slab_of(m, p.x);
        scan_slab(b);
        for (int bb = b + 1; bb < B; ++bb) {
            if (slab_start[bb] == slab_start[bb + 1]) continue;
            const float dx = slab_xmin[bb] - p.x;
            if (dx > 0.f && dx * dx > r2) break;
            scan_slab(bb);
        }
        for (int bb = b - 1; bb >= 0; --bb) {
            if (slab_start[bb] == slab_start[bb + 1]) continue;
            const float dx = p.x - slab_xmax[bb];
            if (dx > 0.f && dx * dx > r2) break;
            scan_slab(bb);
        }
    };
    /* computeMeanAndCovarianceMatrix<double>, shifted by the nearest neighbour (the point itself, distance 0) */
    const double K[3] = {(double)p.x, (double)p.y, (double)p.z};
    double accu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int nn = 0;
    for_each_neighbour([&](const float4 &c) {
        const double x = (double)c.x - K[0], y = (double)c.y - K[1], z = (double)c.z - K[2];
        accu[0] += x * x; accu[1] += x * y; accu[2] += x * z;
        accu[3] += y * y; accu[4] += y * z; accu[5] += z * z;
        accu[6] += x; accu[7] += y; accu[8] += z;
        ++nn;
    });
    if (nn < 3) return;
    for (int k = 0; k < 9; ++k) accu[k] /= (double)nn;
    const double centroid[3] = {accu[6] + K[0], accu[7] + K[1], accu[8] + K[2]};
    double cov[9];
    cov[0] = accu[0] - accu[6] * accu[6];
    cov[1] = accu[1] - accu[6] * accu[7];
    cov[2] = accu[2] - accu[6] * accu[8];
    cov[4] = accu[3] - accu[7] * accu[7];
    cov[5] = accu[4] - accu[7] * accu[8];
    cov[8] = accu[5] - accu[8] * accu[8];
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    double n[3];
    pcl_eigen33_smallest_d(cov, n);
    float4 o = make_float4(p.x, p.y, p.z, 1.0f);
    if (isfinite(n[0]) && isfinite(n[1]) && isfinite(n[2])) {
        const double d4 = -1 * (n[0] * centroid[0] + n[1] * centroid[1] + n[2] * centroid[2]);
        const double distance = (K[0] * n[0] + K[1] * n[1] + K[2] * n[2]) + d4;
        const double mean[3] = {K[0] - distance * n[0], K[1] - distance * n[1], K[2] - distance * n[2]};
        double res[3] = {mean[0], mean[1], mean[2]};
        if (ORDER > 1 && nn >= NC) {
            /* v_axis = plane_normal.unitOrthogonal(); u_axis = plane_normal.cross(v_axis) */
            double va[3], ua[3];
            const double prec = 1e-12;
            if (!(fabs(n[0]) <= fabs(n[2]) * prec) || !(fabs(n[1]) <= fabs(n[2]) * prec)) {
                const double invnm = 1.0 / sqrt(n[0] * n[0] + n[1] * n[1]);
                va[0] = -n[1] * invnm; va[1] = n[0] * invnm; va[2] = 0;
            } else {
                const double invnm = 1.0 / sqrt(n[1] * n[1] + n[2] * n[2]);
                va[0] = 0; va[1] = -n[2] * invnm; va[2] = n[1] * invnm;
            }
            ua[0] = n[1] * va[2] - n[2] * va[1];
            ua[1] = n[2] * va[0] - n[0] * va[2];
            ua[2] = n[0] * va[1] - n[1] * va[0];
            /* lower triangle of P w P^T (row-major packed) and P w f */
            double A[NC * (NC + 1) / 2], cv[NC];
#pragma unroll
            for (int k = 0; k < NC * (NC + 1) / 2; ++k) A[k] = 0.0;
#pragma unroll
            for (int k = 0; k < NC; ++k) cv[k] = 0.0;
            for_each_neighbour([&](const float4 &c) {
/* the one place of the engine where products may fuse into the additions: 65 f64 accumulations per neighbour, whose
   grouping already differs from Eigen's blocked products -- the result is compared to the last float bit, not the last double bit */
#pragma clang fp contract(fast)
                const double dm[3] = {(double)c.x - mean[0], (double)c.y - mean[1], (double)c.z - mean[2]};
                const double w = exp(-(dm[0] * dm[0] + dm[1] * dm[1] + dm[2] * dm[2]) / sqr_gauss);
                const double u_coord = dm[0] * ua[0] + dm[1] * ua[1] + dm[2] * ua[2];
                const double v_coord = dm[0] * va[0] + dm[1] * va[1] + dm[2] * va[2];
                const double f = dm[0] * n[0] + dm[1] * n[1] + dm[2] * n[2];
                double T[NC];
                {
                    int j = 0;
                    double u_pow = 1;
#pragma unroll
                    for (int ui = 0; ui <= ORDER; ++ui) {
                        double v_pow = 1;
#pragma unroll
                        for (int vi = 0; vi <= ORDER - ui; ++vi) { T[j++] = u_pow * v_pow; v_pow *= v_coord; }
                        u_pow *= u_coord;
                    }
                }
#pragma unroll
                for (int a = 0; a < NC; ++a) {
                    const double pw = T[a] * w;
#pragma unroll
                    for (int b = 0; b <= a; ++b) A[a * (a + 1) / 2 + b] += pw * T[b];
                    cv[a] += pw * f;
                }
            });
            /* Eigen LLT, unblocked, lower; a non-positive pivot stops the factorisation and the solve runs on what is there */
#define MLS_A(r, c) A[(r) * ((r) + 1) / 2 + (c)]
            bool live = true;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if (live) {
                    double x = MLS_A(k, k);
#pragma unroll
                    for (int j = 0; j < k; ++j) x -= MLS_A(k, j) * MLS_A(k, j);
                    if (x <= 0.0) live = false;
                    else {
                        x = sqrt(x);
                        MLS_A(k, k) = x;
#pragma unroll
                        for (int r = k + 1; r < NC; ++r) {
                            double t = MLS_A(r, k);
#pragma unroll
                            for (int j = 0; j < k; ++j) t -= MLS_A(r, j) * MLS_A(k, j);
                            MLS_A(r, k) = t / x;
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < NC; ++r) {
                double t = cv[r];
#pragma unroll
                for (int j = 0; j < r; ++j) t -= MLS_A(r, j) * cv[j];
                cv[r] = t / MLS_A(r, r);
            }
#pragma unroll
            for (int r = NC - 1; r >= 0; --r) {
                double t = cv[r];
#pragma unroll
                for (int j = r + 1; j < NC; ++j) t -= MLS_A(j, r) * cv[j];
                cv[r] = t / MLS_A(r, r);
            }
#undef MLS_A
            if (isfinite(cv[0])) { res[0] = mean[0] + cv[0] * n[0]; res[1] = mean[1] + cv[0] * n[1]; res[2] = mean[2] + cv[0] * n[2]; }
        }
        o.x = (float)res[0]; o.y = (float)res[1]; o.z = (float)res[2];
    }
    out4[idx_of(p)] = o;
}

/* ordered compaction of the flagged float4 records (w != 0) into X2 / Y2 / Z2 */
__global__ void __launch_bounds__(256) k_flag_count(const float4 *__restrict__ rec, int n, int *block_cnt)
{
    __shared__ int s_c[4];
    int c = 0;
    for (int i = blockIdx.x * VOX_CHUNK + threadIdx.x; i < min(n, (blockIdx.x + 1) * VOX_CHUNK); i += blockDim.x) c += rec[i].w != 0.f;
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}
__global__ void __launch_bounds__(256) k_flag_compact(const float4 *__restrict__ rec, int n, const int *__restrict__ block_off,
                                                      float *X2, float *Y2, float *Z2)
{
    __shared__ int s_scr[17];
    __shared__ int s_run;
    if (threadIdx.x == 0) s_run = block_off[blockIdx.x];
    __syncthreads();
    const int i0 = blockIdx.x * VOX_CHUNK, i1 = min(n, i0 + VOX_CHUNK);
    for (int base = i0; base < i1; base += blockDim.x) {
        const int i = base + threadIdx.x;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < i1) r = rec[i];
        const int keep = r.w != 0.f;
        int tot;
        const int pre = block_exscan(keep, s_scr, &tot);
        const int run = s_run;
        if (keep) { X2[run + pre] = r.x; Y2[run + pre] = r.y; Z2[run + pre] = r.z; }
        __syncthreads();
        if (threadIdx.x == 0) s_run = run + tot;
        __syncthreads();
    }
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* trans2center: SectPath::trans2center (path_slicing_alg.cpp:82-99; v1 Path_Generation.cpp:60-92).                     */
/* pcl::compute3DCentroid and pcl::computeCovarianceMatrix accumulate in FLOAT, point after point: at a million points  */
/* the rounding of those running sums moves the centroid by hundredths of a millimetre and tilts the axes by 1e-5, and  */
/* everything downstream (band membership, pairing) is decided on the aligned coordinates -- so the sums are reproduced */
/* bit for bit.  A running float sum is sequential, but while it stays inside one binade [2^e, 2^(e+1)) every addition  */
/* is  S <- S + round(x / ulp)  on the integer mantissa S, ties resolved by the parity of the result.  So: one workgroup  */
/* per sum, 8192 values per step; each value becomes a pair (increment if S is even, increment if S is odd), the pairs  */
/* are composed by scans, every prefix is checked to stay inside the binade, and the first value that leaves it is       */
/* added with a real float addition before the scan resumes behind it (k_seq_sum).  Non-finite points contribute +0      */
/* (PCL skips them).                                                                                                    */
/* ------------------------------------------------------------------------------------------------------------------ */
#define SEQ_E 8
#define SEQ_CHUNK (64 * SEQ_E)

__global__ void __launch_bounds__(256) k_seq_prep_centroid(const float *__restrict__ X, const float *__restrict__ Y, const float *__restrict__ Z,
                                                           int n, size_t stride, float *V)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = X[i], y = Y[i], z = Z[i];
    const bool fin = isfinite(x) && isfinite(y) && isfinite(z);
    V[i] = fin ? x : 0.f; V[stride + i] = fin ? y : 0.f; V[2 * stride + i] = fin ? z : 0.f;
}

/* the six products of computeCovarianceMatrix, in its order: yy yz zz, then pt *= pt.x: xx yx zx */
__global__ void __launch_bounds__(256) k_seq_prep_cov(const float *__restrict__ X, const float *__restrict__ Y, const float *__restrict__ Z,
                                                      int n, float cx, float cy, float cz, size_t stride, float *V)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x0 = X[i], y0 = Y[i], z0 = Z[i];
    const bool fin = isfinite(x0) && isfinite(y0) && isfinite(z0);
    const float x = x0 - cx, y = y0 - cy, z = z0 - cz;
    V[i] = fin ? y * y : 0.f;
    V[stride + i] = fin ? y * z : 0.f;
    V[2 * stride + i] = fin ? z * z : 0.f;
    V[3 * stride + i] = fin ? x * x : 0.f;
    V[4 * stride + i] = fin ? y * x : 0.f;
    V[5 * stride + i] = fin ? z * x : 0.f;
}

/* out[b] = the float obtained by adding vals[b * stride + 0 .. n-1] to 0.f one after the other.
   One workgroup of 16 waves per sum, a tile of 16 x 512 values per step: every wave composes its 512 values (wave scan),
   the 16 chunk totals are composed in order, every lane replays its 8 values from the now known mantissa and checks the
   binade; the first value that leaves it (if any) is added for real and the tile resumes behind it.  The next tile's
   loads are in flight meanwhile.  Mantissa arithmetic in int32: before the first value that leaves the binade every
   prefix is below 2^24 in magnitude, so the wrapped sums are exact there (and parities survive wrapping); whatever
   lies behind that value is not used.  Sums that change binade all the time fall back to plain additions by one
   lane for a stretch (see below): 2 ns per value, against 0.4 ns in the scan. */
#define SEQ_WAVES 16
#define SEQ_TILE (SEQ_WAVES * SEQ_CHUNK)
#define SEQ_SERIAL_BELOW 2048
#define SEQ_SERIAL_RUN 4096
__global__ void __launch_bounds__(64 * SEQ_WAVES) k_seq_sum(const float *__restrict__ vals, size_t stride, int n, float *out)
{
    __shared__ __attribute__((aligned(16))) float s_x[SEQ_TILE];
    __shared__ int s_te[SEQ_WAVES], s_to[SEQ_WAVES], s_bad[SEQ_WAVES], s_sb[SEQ_WAVES], s_send;
    __shared__ float s_s;
    __shared__ int s_done;
    const float *v = vals + (size_t)blockIdx.x * stride;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int slot = w * SEQ_CHUNK + lane * SEQ_E; /* this lane's 8 values inside the tile */
    auto load = [&](int base, float (&x)[SEQ_E]) {
        const int i0 = base + slot;
        if (i0 + SEQ_E <= n) {
            const float4 a = *(const float4 *)(v + i0), b = *(const float4 *)(v + i0 + 4);
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < SEQ_E; ++k) x[k] = (i0 + k < n) ? v[i0 + k] : 0.f; /* past the end: +0 */
        }
    };
    if (threadIdx.x == 0) s_s = 0.f;
    const int LO = 1 << 23, HI = 1 << 24;
    float xr[SEQ_E], x[SEQ_E];
    load(0, xr);
    for (int base = 0; base < n; base += SEQ_TILE) {
#pragma unroll
        for (int k = 0; k < SEQ_E; ++k) { x[k] = xr[k]; s_x[slot + k] = x[k]; }
        if (threadIdx.x == 0) s_done = 0;
        load(base + SEQ_TILE, xr); /* in flight while this tile is worked through */
        __syncthreads();
        for (;;) {
            const int done = s_done;
            if (done >= SEQ_TILE) break;
            const float s = s_s;
            const int E = (int)((__float_as_uint(s) >> 23) & 0xffu);
            if (E < 24 || E == 255) { /* zero, tiny, inf, nan: no binade to work in -- one real addition */
                __syncthreads();
                if (threadIdx.x == 0) { s_s = s + s_x[done]; s_done = done + 1; }
                __syncthreads();
                continue;
            }
            const float inv_u = __uint_as_float((unsigned)(127 + 150 - E) << 23), u = __uint_as_float((unsigned)(127 + E - 150) << 23);
            const int S0 = (int)(s * inv_u); /* exact: the signed 24-bit mantissa */
            int de[SEQ_E], dd[SEQ_E]; /* increment when S is even / odd */
            unsigned bigmask = 0;
            int ae = 0, ao = 0; /* the lane's values composed: total increment entering with S even / odd */
#pragma unroll
            for (int k = 0; k < SEQ_E; ++k) {
                const bool active = slot + k >= done;
                const float q = active ? x[k] * inv_u : 0.f; /* exact scaling by a power of two */
                const bool big = !(fabsf(q) < 67108864.f); /* 2^26 ulps (or not a number): leaves the binade for sure */
                const float fl = floorf(q);
                const int d_e = big ? 0 : (int)rintf(q); /* ties to even: the increment that keeps an even S even */
                const int d_o = (!big && (q - fl) == 0.5f) ? (2 * (int)fl + 1 - d_e) : d_e;
                de[k] = d_e; dd[k] = d_o;
                bigmask |= (big ? 1u : 0u) << k;
                ae += (ae & 1) ? d_o : d_e;
                ao += ((1 + ao) & 1) ? d_o : d_e;
            }
            int ie = ae, io = ao; /* inclusive wave scan of the compositions */
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int le = __shfl_up(ie, o, 64), lo = __shfl_up(io, o, 64);
                if (lane >= o) {
                    const int ne = le + ((le & 1) ? io : ie), no = lo + (((1 + lo) & 1) ? io : ie);
                    ie = ne; io = no;
                }
            }
            int xe = __shfl_up(ie, 1, 64), xo = __shfl_up(io, 1, 64);
            if (lane == 0) { xe = 0; xo = 0; }
            if (lane == 63) { s_te[w] = ie; s_to[w] = io; }
            __syncthreads();
            int pe = 0, po = 0; /* the chunks before this wave's, composed */
            for (int c = 0; c < w; ++c) {
                const int ce = s_te[c], co = s_to[c];
                const int ne = pe + ((pe & 1) ? co : ce), no = po + (((1 + po) & 1) ? co : ce);
                pe = ne; po = no;
            }
            const int Sc = S0 + ((S0 & 1) ? po : pe);
            int S = Sc + ((Sc & 1) ? xo : xe);
            int bad = -1, S_before = S;
#pragma unroll
            for (int k = 0; k < SEQ_E; ++k) {
                if (slot + k >= done && bad < 0) {
                    const int Sn = S + ((S & 1) ? dd[k] : de[k]);
                    const int a = Sn < 0 ? -Sn : Sn;
                    /* 2^23 itself may be a sum rounded up from the finer grid below: leave it to the real addition too */
                    if (((bigmask >> k) & 1u) || a <= LO || a >= HI) { bad = k; S_before = S; }
                    else S = Sn;
                }
            }
            const unsigned long long m = __ballot(bad >= 0);
            if (m == 0) {
                if (lane == 63) { s_bad[w] = 0x7fffffff; if (w == SEQ_WAVES - 1) s_send = S; }
            } else {
                const int f = __ffsll((long long)m) - 1;
                if (lane == f) { s_bad[w] = slot + bad; s_sb[w] = S_before; }
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int c = 0;
                while (c < SEQ_WAVES && s_bad[c] == 0x7fffffff) ++c; /* chunks before the first violation are exact */
                if (c == SEQ_WAVES) { s_s = (float)s_send * u; s_done = SEQ_TILE; }
                else {
                    const int b = s_bad[c];
                    float sv = (float)s_sb[c] * u + s_x[b]; /* the real addition across the binade boundary */
                    int d = b + 1;
                    /* a sum that hovers around zero (a centred coordinate, an off-diagonal product) changes binade every
                       few hundred values: a parallel step costs about as much as 1500 plain additions by one lane, so
                       after a short run the next stretch is simply added one value after the other */
                    if (b - done < SEQ_SERIAL_BELOW) {
                        const int end = min(SEQ_TILE, d + SEQ_SERIAL_RUN);
                        /* 128-bit LDS reads issued a batch ahead of the dependent additions that consume them */
                        while (d < end && (d & 3)) { sv = sv + s_x[d]; ++d; }
                        const float4 *x4 = (const float4 *)s_x;
                        float4 A[8], Bq[8];
                        if (d + 64 <= end) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) A[k] = x4[(d >> 2) + k];
                        }
                        while (d + 64 <= end) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) Bq[k] = x4[(d >> 2) + 8 + k];
#pragma unroll
                            for (int k = 0; k < 8; ++k) { sv = sv + A[k].x; sv = sv + A[k].y; sv = sv + A[k].z; sv = sv + A[k].w; }
                            if (d + 128 <= end) {
#pragma unroll
                                for (int k = 0; k < 8; ++k) A[k] = x4[(d >> 2) + 16 + k];
                            }
#pragma unroll
                            for (int k = 0; k < 8; ++k) { sv = sv + Bq[k].x; sv = sv + Bq[k].y; sv = sv + Bq[k].z; sv = sv + Bq[k].w; }
                            d += 64;
                        }
                        for (; d < end; ++d) sv = sv + s_x[d];
                    }
                    s_s = sv; s_done = d;
                }
            }
            __syncthreads();
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = s_s;
}

/* pcl::transformPointCloud with a float Matrix4f, SSE2 build (detail::Transformer<float>::se3): per row
   m0 * x + (m1 * y + (m2 * z + m3)); non-finite points pass unchanged.  src == dst allowed. */
struct Mat34 { float m[3][4]; };
__global__ void __launch_bounds__(256) k_transform_se3(const float *X, const float *Y, const float *Z, int n, Mat34 T, float *X2, float *Y2, float *Z2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = X[i], y = Y[i], z = Z[i];
    float o[3] = {x, y, z};
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = T.m[r][0] * x + (T.m[r][1] * y + (T.m[r][2] * z + T.m[r][3]));
    }
    X2[i] = o[0]; Y2[i] = o[1]; Z2[i] = o[2];
}

/* ------------------------------------------------------------------ */
/* Slice-range handles (SURVEY.md 8e case ii): the points of the cloud   */
/* whose x lies in [lo, hi] -- the interval a handle indexes --, in the  */
/* cloud's own order (ties on the cloud index break as in the whole      */
/* cloud), with their cloud indices.  Built once per plan; the hot path  */
/* then streams the part only.                                           */
/* ------------------------------------------------------------------ */
__global__ void __launch_bounds__(256) k_part_count(const float *__restrict__ X, int n, float lo, float hi, int *block_cnt)
{
    __shared__ int s_c[4];
    int c = 0;
    for (int i = blockIdx.x * VOX_CHUNK + threadIdx.x; i < min(n, (blockIdx.x + 1) * VOX_CHUNK); i += blockDim.x) {
        const float x = X[i];
        c += (x >= lo && x <= hi); /* NaN (a dropped point) fails both */
    }
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}
__global__ void __launch_bounds__(256) k_part_compact(const float *__restrict__ X, const float *__restrict__ Y, const float *__restrict__ Z,
                                                      int n, float lo, float hi, const int *__restrict__ block_off, float *X2, float *Y2,
                                                      float *Z2, int *idx2)
{
    __shared__ int s_scr[17];
    __shared__ int s_run;
    if (threadIdx.x == 0) s_run = block_off[blockIdx.x];
    __syncthreads();
    const int i0 = blockIdx.x * VOX_CHUNK, i1 = min(n, i0 + VOX_CHUNK);
    for (int base = i0; base < i1; base += blockDim.x) {
        const int i = base + threadIdx.x;
        float x = NAN;
        if (i < i1) x = X[i];
        const int keep = (x >= lo && x <= hi);
        int tot;
        const int pre = block_exscan(keep, s_scr, &tot);
        const int run = s_run;
        if (keep) { X2[run + pre] = x; Y2[run + pre] = Y[i]; Z2[run + pre] = Z[i]; idx2[run + pre] = i; }
        __syncthreads();
        if (threadIdx.x == 0) s_run = run + tot;
        __syncthreads();
    }
}
