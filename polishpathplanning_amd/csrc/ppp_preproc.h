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
    const int kk = wave_knn(V, s_w[wv], q.x, q.y, q.z, mean_k + 1, r0, sc);
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
