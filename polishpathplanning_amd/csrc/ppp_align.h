/*
 * ppp_align.h -- host side of trans2center: the 3 x 3 float eigen-decomposition the
 * reference takes from Eigen::EigenSolver<Matrix3f> (path_slicing_alg.cpp:92-94), TransAlign and its inverse
 * (path_translation_alg.cpp:146).  Nine floats come down from the device, sixteen go back: this is plan-time host
 * arithmetic like the slice walk, not a data path.
 *
 * EigenSolver is restated from Eigen 3.3 / 3.4 (RealSchur: scaling, Hessenberg reduction, Francis double-shift steps;
 * EigenSolver: back substitution on T, back transformation, normalised columns).  Eigenvalue order and eigenvector
 * signs are whatever the iteration leaves -- the reference uses them unsorted.  Matrices row-major m[r][c].
 */
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>

namespace ppp_align {

struct EigenSolver3f {
    float T[3][3], U[3][3], ev[3];
    bool complex_pair = false, converged = true;

    static void householder(const float *v, int n, float *ess, float &tau, float &beta)
    {   /* MatrixBase::makeHouseholder */
        float tail = 0.f;
        for (int i = 1; i < n; ++i) tail = (i == 1) ? v[1] * v[1] : tail + v[i] * v[i];
        const float c0 = v[0];
        if (tail <= std::numeric_limits<float>::min()) {
            tau = 0.f; beta = c0;
            for (int i = 0; i + 1 < n; ++i) ess[i] = 0.f;
            return;
        }
        beta = std::sqrt(c0 * c0 + tail);
        if (c0 >= 0.f) beta = -beta;
        for (int i = 0; i + 1 < n; ++i) ess[i] = v[i + 1] / (c0 - beta);
        tau = (beta - c0) / beta;
    }
    static void reflect_rows(float M[3][3], int r0, int nr, int c0, int nc, const float *ess, float tau)
    {   /* applyHouseholderOnTheLeft on a block */
        if (nr == 1) { for (int j = 0; j < nc; ++j) M[r0][c0 + j] *= 1.f - tau; return; }
        if (tau == 0.f) return;
        for (int j = c0; j < c0 + nc; ++j) {
            float t = ess[0] * M[r0 + 1][j];
            for (int i = 2; i < nr; ++i) t += ess[i - 1] * M[r0 + i][j];
            t += M[r0][j];
            M[r0][j] -= tau * t;
            for (int i = 1; i < nr; ++i) M[r0 + i][j] -= t * (tau * ess[i - 1]);
        }
    }
    static void reflect_cols(float M[3][3], int r0, int nr, int c0, int nc, const float *ess, float tau)
    {   /* applyHouseholderOnTheRight on a block */
        if (nc == 1) { for (int i = 0; i < nr; ++i) M[r0 + i][c0] *= 1.f - tau; return; }
        if (tau == 0.f) return;
        for (int i = r0; i < r0 + nr; ++i) {
            float t = M[i][c0 + 1] * ess[0];
            for (int j = 2; j < nc; ++j) t += M[i][c0 + j] * ess[j - 1];
            t += M[i][c0];
            M[i][c0] -= tau * t;
            for (int j = 1; j < nc; ++j) M[i][c0 + j] -= ess[j - 1] * (tau * t);
        }
    }
    static void givens(float p, float q, float &c, float &s)
    {   /* JacobiRotation::makeGivens, real */
        if (q == 0.f) { c = p < 0.f ? -1.f : 1.f; s = 0.f; return; }
        if (p == 0.f) { c = 0.f; s = q < 0.f ? 1.f : -1.f; return; }
        if (std::fabs(p) > std::fabs(q)) {
            const float t = q / p;
            float u = std::sqrt(1.f + t * t);
            if (p < 0.f) u = -u;
            c = 1.f / u; s = -t * c;
        } else {
            const float t = p / q;
            float u = std::sqrt(1.f + t * t);
            if (q < 0.f) u = -u;
            s = -1.f / u; c = -t * s;
        }
    }

    void compute(const float A[3][3])
    {
        const float eps = std::numeric_limits<float>::epsilon(), tiny = std::numeric_limits<float>::min();
        const int N = 3;
        float scale = 0.f;
        for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) { scale = std::max(scale, std::fabs(A[i][j])); T[i][j] = 0.f; U[i][j] = (i == j) ? 1.f : 0.f; }
        if (!(scale < tiny)) {
            float H[3][3];
            for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) H[i][j] = A[i][j] / scale;
            /* Hessenberg form: one reflector annihilates H20 (the length-1 second one is the identity) */
            const float col[2] = {H[1][0], H[2][0]};
            float e0, h0, b0;
            householder(col, 2, &e0, h0, b0);
            H[1][0] = b0; H[2][0] = e0;
            reflect_rows(H, 1, 2, 1, 2, &e0, h0);
            reflect_cols(H, 0, 3, 1, 2, &e0, h0);
            reflect_rows(U, 1, 2, 1, 2, &e0, h0); /* Q = the reflector on the identity's lower corner */
            for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) T[i][j] = H[i][j];
            T[2][0] = 0.f;
            schur(eps, tiny);
            for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) T[i][j] *= scale;
        }
        vectors(eps);
    }

private:
    void schur(float eps, float tiny)
    {   /* RealSchur::computeFromHessenberg */
        const int N = 3, max_iters = 40 * N;
        float norm = 0.f;
        for (int j = 0; j < N; ++j) {
            float c = std::fabs(T[0][j]);
            for (int i = 1; i < std::min(N, j + 2); ++i) c += std::fabs(T[i][j]);
            norm += c;
        }
        if (norm == 0.f) return;
        const float as_zero = std::max(norm * (eps * eps), tiny);
        float exshift = 0.f;
        int iu = N - 1, iter = 0, total = 0;
        while (iu >= 0) {
            int il = iu;
            for (; il > 0; --il) {
                const float s = std::max((std::fabs(T[il - 1][il - 1]) + std::fabs(T[il][il])) * eps, as_zero);
                if (std::fabs(T[il][il - 1]) <= s) break;
            }
            if (il == iu) {
                T[iu][iu] = T[iu][iu] + exshift;
                if (iu > 0) T[iu][iu - 1] = 0.f;
                --iu; iter = 0;
            } else if (il == iu - 1) {
                const float p = 0.5f * (T[iu - 1][iu - 1] - T[iu][iu]);
                const float q = p * p + T[iu][iu - 1] * T[iu - 1][iu];
                T[iu][iu] += exshift;
                T[iu - 1][iu - 1] += exshift;
                if (q >= 0.f) {
                    const float z = std::sqrt(std::fabs(q));
                    float c, s;
                    givens(p >= 0.f ? p + z : p - z, T[iu][iu - 1], c, s);
                    for (int j = iu - 1; j < N; ++j) { const float x = T[iu - 1][j], y = T[iu][j]; T[iu - 1][j] = c * x - s * y; T[iu][j] = s * x + c * y; }
                    for (int i = 0; i <= iu; ++i) { const float x = T[i][iu - 1], y = T[i][iu]; T[i][iu - 1] = c * x - s * y; T[i][iu] = s * x + c * y; }
                    T[iu][iu - 1] = 0.f;
                    for (int i = 0; i < N; ++i) { const float x = U[i][iu - 1], y = U[i][iu]; U[i][iu - 1] = c * x - s * y; U[i][iu] = s * x + c * y; }
                } else complex_pair = true;
                if (iu > 1) T[iu - 1][iu - 2] = 0.f;
                iu -= 2; iter = 0;
            } else {
                float sh[3] = {T[iu][iu], T[iu - 1][iu - 1], T[iu][iu - 1] * T[iu - 1][iu]};
                if (iter == 10) { /* Wilkinson's ad hoc shift */
                    exshift += sh[0];
                    for (int i = 0; i <= iu; ++i) T[i][i] -= sh[0];
                    const float s = std::fabs(T[iu][iu - 1]) + std::fabs(T[iu - 1][iu - 2]);
                    sh[0] = 0.75f * s; sh[1] = 0.75f * s; sh[2] = -0.4375f * s * s;
                }
                if (iter == 30) { /* MATLAB's ad hoc shift */
                    float s = (sh[1] - sh[0]) / 2.0f;
                    s = s * s + sh[2];
                    if (s > 0.f) {
                        s = std::sqrt(s);
                        if (sh[1] < sh[0]) s = -s;
                        s = s + (sh[1] - sh[0]) / 2.0f;
                        s = sh[0] - sh[2] / s;
                        exshift += s;
                        for (int i = 0; i <= iu; ++i) T[i][i] -= s;
                        sh[0] = sh[1] = sh[2] = 0.964f;
                    }
                }
                ++iter; ++total;
                if (total > max_iters) { converged = false; return; }
                int im = iu - 2;
                float first[3] = {0.f, 0.f, 0.f};
                for (; im >= il; --im) {
                    const float d = T[im][im], r = sh[0] - d, s = sh[1] - d;
                    first[0] = (r * s - sh[2]) / T[im + 1][im] + T[im][im + 1];
                    first[1] = T[im + 1][im + 1] - d - r - s;
                    first[2] = T[im + 2][im + 1];
                    if (im == il) break;
                    const float lhs = T[im][im - 1] * (std::fabs(first[1]) + std::fabs(first[2]));
                    const float rhs = first[0] * (std::fabs(T[im - 1][im - 1]) + std::fabs(d) + std::fabs(T[im + 1][im + 1]));
                    if (std::fabs(lhs) < eps * rhs) break;
                }
                for (int k = im; k <= iu - 2; ++k) {
                    float v[3] = {first[0], first[1], first[2]};
                    if (k != im) { v[0] = T[k][k - 1]; v[1] = T[k + 1][k - 1]; v[2] = T[k + 2][k - 1]; }
                    float ess[2], tau, beta;
                    householder(v, 3, ess, tau, beta);
                    if (beta != 0.f) {
                        if (k == im && k > il) T[k][k - 1] = -T[k][k - 1];
                        else if (k != im) T[k][k - 1] = beta;
                        reflect_rows(T, k, 3, k, N - k, ess, tau);
                        reflect_cols(T, 0, std::min(iu, k + 3) + 1, k, 3, ess, tau);
                        reflect_cols(U, 0, N, k, 3, ess, tau);
                    }
                }
                const float v[2] = {T[iu - 1][iu - 2], T[iu][iu - 2]};
                float ess, tau, beta;
                householder(v, 2, &ess, tau, beta);
                if (beta != 0.f) {
                    T[iu - 1][iu - 2] = beta;
                    reflect_rows(T, iu - 1, 2, iu - 1, N - iu + 1, &ess, tau);
                    reflect_cols(T, 0, iu + 1, iu - 1, 2, &ess, tau);
                    reflect_cols(U, 0, N, iu - 1, 2, &ess, tau);
                }
                for (int i = im + 2; i <= iu; ++i) { T[i][i - 2] = 0.f; if (i > im + 2) T[i][i - 3] = 0.f; }
            }
        }
    }
    void vectors(float eps)
    {   /* EigenSolver::compute (eigenvalues), doComputeEigenvectors, eigenvectors() */
        const int N = 3;
        for (int i = 0; i < N; ++i) ev[i] = T[i][i];
        for (int i = 0; i + 1 < N; ++i) if (T[i + 1][i] != 0.f) complex_pair = true;
        if (complex_pair || !converged) return;
        float norm = 0.f;
        for (int j = 0; j < N; ++j) {
            const int a = std::max(j - 1, 0);
            float r = std::fabs(T[j][a]);
            for (int k = a + 1; k < N; ++k) r += std::fabs(T[j][k]);
            norm += r;
        }
        if (norm != 0.f) {
            for (int n = N - 1; n >= 0; --n) {
                const float p = ev[n];
                int l = n;
                T[n][n] = 1.f;
                for (int i = n - 1; i >= 0; --i) {
                    const float w = T[i][i] - p;
                    float r = T[i][l] * T[l][n];
                    for (int k = l + 1; k <= n; ++k) r += T[i][k] * T[k][n];
                    l = i;
                    T[i][n] = (w != 0.f) ? -r / w : -r / (eps * norm);
                    const float t = std::fabs(T[i][n]);
                    if ((eps * t) * t > 1.f) for (int k = i; k < N; ++k) T[k][n] /= t;
                }
            }
            for (int j = N - 1; j >= 0; --j) {
                float col[3];
                for (int i = 0; i < N; ++i) {
                    float a = U[i][0] * T[0][j];
                    for (int k = 1; k <= j; ++k) a += U[i][k] * T[k][j];
                    col[i] = a;
                }
                for (int i = 0; i < N; ++i) U[i][j] = col[i];
            }
        }
        for (int j = 0; j < N; ++j) {
            const float z = (U[0][j] * U[0][j] + U[1][j] * U[1][j]) + U[2][j] * U[2][j];
            if (z > 0.f) { const float s = std::sqrt(z); for (int i = 0; i < N; ++i) U[i][j] /= s; }
        }
    }
};

/* TransAlign = [V^T | -V^T c] (path_slicing_alg.cpp:96-97) */
inline void trans_align(const EigenSolver3f &es, const float c[3], float TA[4][4])
{
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) TA[i][j] = es.U[j][i];
        TA[i][3] = ((-es.U[0][i]) * c[0] + (-es.U[1][i]) * c[1]) + (-es.U[2][i]) * c[2];
    }
    TA[3][0] = TA[3][1] = TA[3][2] = 0.f; TA[3][3] = 1.f;
}

/* Matrix4f::inverse(): the cofactor form of Eigen's generic 4 x 4 path (the SSE build runs Intel's routine, whose
   entries differ in the last bits -- 1e-7 relative on the waypoints) */
inline void inverse4(const float m[4][4], float r[4][4])
{
    auto det3 = [&](int i1, int i2, int i3, int j1, int j2, int j3) { return m[i1][j1] * (m[i2][j2] * m[i3][j3] - m[i2][j3] * m[i3][j2]); };
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const int i1 = (i + 1) % 4, i2 = (i + 2) % 4, i3 = (i + 3) % 4, j1 = (j + 1) % 4, j2 = (j + 2) % 4, j3 = (j + 3) % 4;
            const float cof = det3(i1, i2, i3, j1, j2, j3) + det3(i2, i3, i1, j1, j2, j3) + det3(i3, i1, i2, j1, j2, j3);
            r[j][i] = ((i + j) & 1) ? -cof : cof;
        }
    const float det = ((m[0][0] * r[0][0] + m[1][0] * r[0][1]) + m[2][0] * r[0][2]) + m[3][0] * r[0][3];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r[i][j] /= det;
}

} // namespace ppp_align
