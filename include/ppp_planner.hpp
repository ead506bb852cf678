/*
 * ppp_planner.hpp -- shared plumbing of the drop-in planner classes (Path_Generate.h,
 * Path_Generate_Algorithm.h, robot_path.h): owns one engine handle, loads the PCD, forwards
 * every method to the C ABI of include/ppp_hip.h.  Header-only on purpose: the reference
 * defines two different classes named path_generater (one per executable), so nothing here
 * may live in a shared translation unit.
 *
 * Types in the public signatures: with Eigen available (-DPPP_WITH_EIGEN or <Eigen/Dense> on
 * the include path) the real Eigen::Vector3f / Vector3d are used; otherwise the three-float
 * stand-ins below, which cover what the reference's callers do with them (construct from
 * three numbers, index with []).  PCL is not needed: the cloud lives in HBM.
 */
#ifndef PPP_PLANNER_HPP
#define PPP_PLANNER_HPP

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "ppp_hip.h"

#if defined(PPP_WITH_EIGEN) || (defined(__has_include) && __has_include(<Eigen/Dense>))
#include <Eigen/Dense>
#else
namespace Eigen {
template <typename T>
struct PppVec3 {
    T v[3];
    PppVec3() : v{0, 0, 0} {}
    PppVec3(T a, T b, T c) : v{a, b, c} {}
    T &operator[](int i) { return v[i]; }
    const T &operator[](int i) const { return v[i]; }
    T &operator()(int i) { return v[i]; }
    const T &operator()(int i) const { return v[i]; }
};
typedef PppVec3<float> Vector3f;
typedef PppVec3<double> Vector3d;
} // namespace Eigen
#endif

typedef std::map<double, std::vector<double>> MAP; /* Path_Generate_Algorithm.h:53 */

namespace ppp {

/* Engine handles handed back by the planners of this process, for the next planner to take.  The reference builds a planner
   per workpiece from the file name (path_slicing_alg.cpp:3-30, Path_Generate.cpp:8-33); a handle is a HIP stream plus the
   engine's device buffers and its plan, so the second workpiece of a process pays neither ppp_create nor -- for a cloud of the
   same size and parameters -- the window census (ppp_set_plan_reuse).  A taken handle is given its cloud and parameters
   anew by open(); nothing of the earlier workpiece is readable through it.  PPP_NO_HANDLE_POOL=1 in the environment (or
   HandlePool::keep(0)) switches it off.  Handles still pooled when the process ends go with the HIP runtime. */
class HandlePool {
public:
    static ppp_handle take(int device)
    {
        std::lock_guard<std::mutex> g(mu());
        std::vector<std::pair<int, ppp_handle>> &f = free_list();
        for (size_t i = f.size(); i-- > 0;)
            if (f[i].first == device) {
                ppp_handle h = f[i].second;
                f.erase(f.begin() + (long)i);
                ++reused();
                return h;
            }
        return nullptr;
    }
    static void give(int device, ppp_handle h)
    {
        {
            std::lock_guard<std::mutex> g(mu());
            if (free_list().size() < limit()) {
                free_list().push_back(std::make_pair(device, h));
                return;
            }
        }
        ppp_destroy(h);
    }
    /* how many handles may wait in the pool (default 4; 0 = every planner creates and destroys its own) */
    static void keep(size_t n)
    {
        std::vector<ppp_handle> out;
        {
            std::lock_guard<std::mutex> g(mu());
            limit() = n;
            while (free_list().size() > n) { out.push_back(free_list().back().second); free_list().pop_back(); }
        }
        for (ppp_handle h : out) ppp_destroy(h);
    }
    static size_t taken_from_pool()
    {
        std::lock_guard<std::mutex> g(mu());
        return reused();
    }

private:
    /* (never destroyed: a planner with static storage may hand its handle back after main() has returned) */
    static std::mutex &mu() { static std::mutex *m = new std::mutex; return *m; }
    static std::vector<std::pair<int, ppp_handle>> &free_list() { static auto *f = new std::vector<std::pair<int, ppp_handle>>; return *f; }
    static size_t &reused() { static size_t r = 0; return r; }
    static size_t &limit()
    {
        static size_t n = [] { const char *e = std::getenv("PPP_NO_HANDLE_POOL"); return (e && *e && *e != '0') ? 0 : 4; }();
        return n;
    }
};

/* One engine handle + the state every planner class of the reference keeps. */
class Planner {
public:
    Planner() { ppp_default_config(&cfg_); }
    ~Planner()
    {
        if (!h_) return;
        /* every method of the classes returns with its results on the host, so the stream is idle here */
        HandlePool::give(dev_, h_);
    }
    Planner(const Planner &) = delete;
    Planner &operator=(const Planner &) = delete;

    bool ok() const { return h_ != nullptr && loaded_; }
    ppp_handle handle() const { return h_; }
    ppp_config &config() { return cfg_; }

    /* constructor body of the reference classes: load, recolour (no-op here), scale, keep */
    bool open(const std::string &cloud_name)
    {
        if (!h_) {
            dev_ = device_from_env();
            h_ = HandlePool::take(dev_);
            if (h_) {
                /* switches a caller may have thrown through handle() on the earlier workpiece: back to the defaults */
                ppp_set_fast_path(h_, 1);
                ppp_enable_timing(h_, 0);
                ppp_set_plan_reuse(h_, 1);
            } else if (ppp_create(dev_, &h_) != PPP_OK) {
                std::fprintf(stderr, "ppp: no MI355X device available (the engine has no CPU fallback)\n");
                h_ = nullptr;
                return false;
            }
        }
        if (!apply_params()) return false;
        /* pcl::io::loadPCDFile + the x1000 loop: the file's records go straight to HBM (ppp_set_cloud_pcd) */
        size_t n = 0;
        ppp_pcd_layout lay;
        if (ppp_pcd_probe(cloud_name.c_str(), &lay) != PPP_OK) {
            std::fprintf(stderr, "Cloudn't read file!\n"); /* path_slicing_alg.cpp:11 */
            loaded_ = false;
            return false;
        }
        int rc = ppp_set_cloud_pcd(h_, cloud_name.c_str(), &n, nullptr);
        if (rc == PPP_ERR_IO || rc == PPP_ERR_UNSUPPORTED) {
            std::fprintf(stderr, "Cloudn't read file!\n");
            loaded_ = false;
            return false;
        }
        if (rc != PPP_OK) return report(rc);
        loaded_ = true;
        return true;
    }
    bool remove_outlier(int mean_k, double stddev_mul)
    {
        if (!ok()) return false;
        size_t n = 0;
        int rc = ppp_remove_outlier(h_, mean_k, stddev_mul, &n, nullptr);
        return rc == PPP_OK ? true : report(rc);
    }
    bool voxel_down(float lx, float ly, float lz)
    {
        if (!ok()) return false;
        int overflow = 0;
        int rc = ppp_voxel_down(h_, lx, ly, lz, nullptr, &overflow);
        if (rc == PPP_OK && overflow) std::fprintf(stderr, "[pcl::VoxelGrid::applyFilter] Leaf size is too small for the input dataset. Integer indices would overflow.\n");
        return rc == PPP_OK ? true : report(rc);
    }
    /* SectPath::smooth: MLS on the resident cloud, then the "smooth_<name>" ascii PCD the reference leaves behind
       (path_slicing_alg.cpp:126-138; coordinates back in metres when ChangeRange).  A name with a directory part makes
       an unwritable "smooth_dir/..." path there (pcl::io throws); here the file is skipped with a note. */
    bool smooth_mls(double radius, int order, const std::string &cloud_name, bool scale_back)
    {
        if (!ok()) return false;
        size_t n = 0;
        int rc = ppp_smooth_mls(h_, radius, order, &n);
        if (rc != PPP_OK) return report(rc);
        std::vector<float> xyz(3 * (n ? n : 1));
        rc = ppp_get_cloud(h_, xyz.data(), n, &n);
        if (rc != PPP_OK) return report(rc);
        if (scale_back) for (size_t i = 0; i < 3 * n; ++i) xyz[i] /= 1000;
        const float vp[7] = {0, 0, 0, 1, 0, 0, 0};
        const std::string side = "smooth_" + cloud_name;
        if (ppp_save_pcd(side.c_str(), xyz.data(), n, 3, vp, 0) != PPP_OK) std::fprintf(stderr, "ppp: could not write %s\n", side.c_str());
        return true;
    }
    bool trans2center()
    {
        if (!ok()) return false;
        int rc = ppp_trans2center(h_, nullptr, nullptr, nullptr);
        return rc == PPP_OK ? true : report(rc);
    }
    bool apply_params()
    {
        int rc = ppp_set_params(h_, &cfg_.params);
        return rc == PPP_OK ? true : report(rc);
    }
    bool report(int rc) const
    {
        std::fprintf(stderr, "ppp error %d: %s\n", rc, h_ ? ppp_last_error(h_) : "no handle");
        return false;
    }
    std::vector<int> rangedX_index(int position)
    {
        std::vector<int> out(4096);
        size_t n = 0;
        int rc = ppp_ranged_x_index(h_, position, out.data(), out.size(), &n);
        if (rc == PPP_OK && n > out.size()) { /* the call reports the full count and copies what fits: ask again with room */
            out.resize(n);
            rc = ppp_ranged_x_index(h_, position, out.data(), out.size(), &n);
        }
        if (rc != PPP_OK) { report(rc); n = 0; }
        out.resize(n);
        return out;
    }
    /* estimate_normal() (path_slicing_alg.cpp:141-150, Path_Generation.cpp:323-333): pcl::NormalEstimation, radius 2.5,
       over the whole cloud; the field stays readable as cloud_normals() -- n x (nx ny nz curvature), cloud index order */
    bool estimate_normal()
    {
        if (!ok()) return false;
        size_t n = 0;
        ppp_num_points(h_, &n);
        normals_.assign(4 * n, 0.f);
        int rc = ppp_estimate_normals(h_, normals_.data());
        if (rc != PPP_OK) { normals_.clear(); return report(rc); }
        return true;
    }
    const std::vector<float> &cloud_normals() const { return normals_; }
    size_t num_points() const
    {
        size_t n = 0;
        if (h_) ppp_num_points(h_, &n);
        return n;
    }
    MAP insert_point(const std::vector<int> &indices, float plane_x)
    {
        MAP Node;
        std::vector<double> y(indices.size() + 1), x(indices.size() + 1), z(indices.size() + 1);
        size_t m = 0;
        int rc = ppp_insert_point(h_, indices.data(), indices.size(), plane_x, y.data(), x.data(), z.data(), indices.size(), &m);
        if (rc != PPP_OK) { report(rc); return Node; }
        for (size_t i = 0; i < m; ++i) Node[y[i]] = {x[i], z[i]};
        return Node;
    }
    bool gen_path()
    {
        int rc = ppp_gen_path_async(h_);
        if (rc == PPP_OK) rc = ppp_sync(h_);
        return rc == PPP_OK ? true : report(rc);
    }
    /* getPath(): returns the list and writes pathFile exactly like path_translation_alg.cpp:216-228 */
    bool get_path(std::vector<float> &wp6)
    {
        int rc = ppp_get_path_async(h_);
        if (rc == PPP_OK) rc = ppp_sync(h_);
        if (rc != PPP_OK) return report(rc);
        size_t W = 0;
        ppp_num_waypoints(h_, &W);
        wp6.resize(6 * W);
        rc = ppp_get_waypoints(h_, wp6.data(), W, &W);
        if (rc != PPP_OK) return report(rc);
        std::cout << "!!!!! GOT PATH !!!!!" << std::endl;
        if (ppp_write_path_file(cfg_.path_file, wp6.data(), W) == PPP_OK) std::cout << "File saved: " << cfg_.path_file << std::endl;
        return true;
    }
    int num_slices()
    {
        int S = 0;
        ppp_num_slices(h_, &S);
        return S;
    }
    /* show() (path_slicing_alg.cpp:69-80, Path_Generation.cpp:37-51) opens a PCLVisualizer on `other_cloud + cloud`: the
       spline knots insert_point added (coloured node_rgb: red, white in contour_alg.cpp:228-230) followed by the cloud itself,
       white, with the cloud point nearest to every millimetre of every path recoloured path_rgb (drawpath,
       path_slicing_alg.cpp:269-288).  No viewer here: with PPP_SHOW_PCD=<file> in the environment that very cloud is written
       as a PointXYZRGB PCD (binary) for any viewer; without it a notice is printed.  with_boundaries: the planner with the
       dynamic adjustment also paints, before it adjusts a slice, the boundary curve it adjusts it against
       (drawpath(*boundary, 0,255,0), path_dynamic_alg.cpp:320-322); slices are painted in the order thread_worker takes them
       (the start slice, then step by step outwards; the reference's two threads interleave as they please -- here left before
       right), so a later curve recolours an earlier one where they share a cloud point, as there. */
    void show_dump(const unsigned char node_rgb[3], const unsigned char path_rgb[3], bool with_boundaries = false)
    {
        const char *out = std::getenv("PPP_SHOW_PCD");
        if (!out || !out[0] || !ok()) {
            std::printf("show(): viewer not built; the cloud stays resident on the GPU (PPP_SHOW_PCD=<file> writes what the viewer would show)\n");
            return;
        }
        size_t n = 0;
        if (ppp_get_cloud(h_, nullptr, 0, &n) != PPP_OK) { report(PPP_ERR_ARG); return; }
        std::vector<float> cloud(3 * (n ? n : 1));
        if (n && ppp_get_cloud(h_, cloud.data(), n, &n) != PPP_OK) { report(PPP_ERR_HIP); return; }
        std::vector<unsigned char> crgb(3 * (n ? n : 1), 255);
        std::vector<float> nodes;
        int S = 0;
        if (ppp_num_slices(h_, &S) != PPP_OK) S = 0; /* show() before GenPath: the bare cloud */
        /* drawpath: dy = miny; while (dy < maxy) { nearest cloud point of path.point(dy) takes the colour; dy += 1; } */
        auto samples = [](double miny, double maxy) {
            std::vector<double> q;
            for (double dy = miny; dy < maxy; dy += 1) q.push_back(dy);
            return q;
        };
        auto paint = [&](const std::vector<double> &xyz, const unsigned char rgb[3]) {
            std::vector<float> qf(xyz.begin(), xyz.end());
            std::vector<int> nn(qf.size() / 3, -1);
            if (nn.empty() || ppp_nearest(h_, qf.data(), nn.size(), nn.data()) != PPP_OK) return;
            for (int id : nn) if (id >= 0 && (size_t)id < n) for (int c = 0; c < 3; ++c) crgb[3 * (size_t)id + c] = rgb[c];
        };
        /* painting order: by the slice's step in its chain when the boundaries are painted too, else as the slices stand */
        std::vector<std::pair<int, int>> order;
        for (int s = 0; s < S; ++s) {
            int step = s;
            if (with_boundaries && ppp_get_boundary(h_, s, nullptr, nullptr, nullptr, 0, nullptr, &step) != PPP_OK) step = s;
            order.push_back(std::make_pair(step, s));
        }
        if (with_boundaries) std::sort(order.begin(), order.end());
        std::vector<std::vector<float>> slice_nodes(S > 0 ? S : 0);
        const unsigned char green[3] = {0, 255, 0};
        size_t painted_boundaries = 0;
        for (const auto &os : order) {
            const int s = os.second;
            if (with_boundaries) {
                size_t mb = 0;
                if (ppp_get_boundary(h_, s, nullptr, nullptr, nullptr, 0, &mb, nullptr) == PPP_OK && mb >= 3) {
                    std::vector<double> by(mb), bx(mb), bz(mb);
                    ppp_spline sp = nullptr;
                    if (ppp_get_boundary(h_, s, by.data(), bx.data(), bz.data(), mb, &mb, nullptr) == PPP_OK &&
                        ppp_spline_create(device_from_env(), mb, by.data(), bx.data(), bz.data(), &sp) == PPP_OK) {
                        const std::vector<double> q = samples(by.front(), by.back());
                        std::vector<double> xyz(3 * q.size());
                        if (!q.empty() && ppp_spline_eval(sp, q.data(), q.size(), xyz.data()) == PPP_OK) { paint(xyz, green); ++painted_boundaries; }
                        ppp_spline_destroy(sp);
                    }
                }
            }
            size_t m = 0;
            if (ppp_get_nodes(h_, s, nullptr, nullptr, nullptr, 0, &m) != PPP_OK || m < 3) continue;
            std::vector<double> y(m), x(m), z(m);
            ppp_get_nodes(h_, s, y.data(), x.data(), z.data(), m, &m);
            for (size_t i = 0; i < m; ++i) { slice_nodes[s].push_back((float)x[i]); slice_nodes[s].push_back((float)y[i]); slice_nodes[s].push_back((float)z[i]); }
            const std::vector<double> q = samples(y.front(), y.back());
            if (q.empty()) continue;
            std::vector<double> xyz(3 * q.size());
            if (ppp_eval_spline(h_, s, q.data(), q.size(), xyz.data()) != PPP_OK) continue;
            paint(xyz, path_rgb);
        }
        for (int s = 0; s < S; ++s) nodes.insert(nodes.end(), slice_nodes[s].begin(), slice_nodes[s].end()); /* other_cloud: in slice order */
        if (with_boundaries) std::printf("show(): %zu boundary curves painted\n", painted_boundaries);
        const size_t nn_ = nodes.size() / 3;
        std::vector<float> all(nodes);
        all.insert(all.end(), cloud.begin(), cloud.begin() + 3 * n);
        std::vector<unsigned char> rgb(3 * nn_);
        for (size_t i = 0; i < nn_; ++i) for (int c = 0; c < 3; ++c) rgb[3 * i + c] = node_rgb[c];
        rgb.insert(rgb.end(), crgb.begin(), crgb.begin() + 3 * n);
        const float vp[7] = {0, 0, 0, 1, 0, 0, 0};
        if (ppp_save_pcd_rgb(out, all.data(), rgb.data(), nn_ + n, vp, 1) == PPP_OK)
            std::printf("show(): %zu inserted nodes + %zu cloud points written to %s\n", nn_, n, out);
        else std::fprintf(stderr, "ppp: could not write %s\n", out);
    }
    void show_notice()
    {   /* the classes without a colour scheme of their own */
        const unsigned char red[3] = {255, 0, 0};
        show_dump(red, red);
    }
    static int device_from_env()
    {
        const char *e = std::getenv("PPP_DEVICE");
        return e ? std::atoi(e) : 0;
    }

private:
    ppp_handle h_ = nullptr;
    int dev_ = 0;
    ppp_config cfg_;
    bool loaded_ = false;
    std::vector<float> normals_;
};

} // namespace ppp
#endif
