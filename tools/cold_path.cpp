// The cold path as a C caller sees it (tools/cold_path.py measures it through ctypes): clouds already in device memory,
// ppp_set_cloud_device_async (PPP_COLD_WAITING_CALL=1: ppp_set_cloud_device) + ppp_run_async + ppp_sync per never-seen cloud.   usage: cold_path a.pcd b.pcd ...  (same size, >= 3 files)
// build: hipcc -O2 -std=c++17 -I include -o /tmp/cold_path tools/cold_path.cpp -L polishpathplanning_amd -lppp_hip -Wl,-rpath,$PWD/polishpathplanning_amd
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ppp_hip.h"

int main(int argc, char **argv)
{
    if (argc < 4) { std::printf("usage: cold_path a.pcd b.pcd c.pcd ...\n"); return 2; }
    ppp_handle h = nullptr;
    if (ppp_create(0, &h) != PPP_OK) { std::printf("no device\n"); return 1; }
    ppp_params p;
    ppp_default_params(&p);
    p.tool_radius = 6; p.walk = PPP_WALK_CENTER_INT;
    if (ppp_set_params(h, &p) != PPP_OK) return 1;
    std::vector<float *> dev;
    std::vector<size_t> ns;
    /* PPP_COLD_COPY_BEFORE=1: every cloud is copied to the device again right before it is timed (as tools/cold_path.py and bench.py do:
       the first launch behind a copy and a device-wide wait starts later than one behind another pass) */
    const bool copy_before = std::getenv("PPP_COLD_COPY_BEFORE") != nullptr;
    std::vector<float *> host;
    for (int i = 1; i < argc; ++i) {
        float *xyz = nullptr, vp[7];
        size_t n = 0;
        if (ppp_load_pcd(argv[i], &xyz, &n, vp) != PPP_OK) { std::printf("cannot read %s\n", argv[i]); return 1; }
        float *d = nullptr;
        if (hipMalloc((void **)&d, n * 12) != hipSuccess || hipMemcpy(d, xyz, n * 12, hipMemcpyHostToDevice) != hipSuccess) return 1;
        host.push_back(xyz);
        dev.push_back(d); ns.push_back(n);
    }
    (void)hipDeviceSynchronize();
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    std::vector<double> tot, a_, b_, c_;
    for (size_t i = 0; i < dev.size(); ++i) {
        if (copy_before) { if (hipMemcpy(dev[i], host[i], ns[i] * 12, hipMemcpyHostToDevice) != hipSuccess) return 1; (void)hipDeviceSynchronize(); }
        const auto t0 = std::chrono::steady_clock::now();
        int rc = std::getenv("PPP_COLD_WAITING_CALL") ? ppp_set_cloud_device(h, dev[i], ns[i], 12, nullptr) : ppp_set_cloud_device_async(h, dev[i], ns[i], 12, nullptr);
        const auto t1 = std::chrono::steady_clock::now();
        if (rc == PPP_OK) rc = ppp_run_async(h);
        const auto t2 = std::chrono::steady_clock::now();
        if (rc == PPP_OK) rc = ppp_sync(h);
        const auto t3 = std::chrono::steady_clock::now();
        size_t W = 0;
        if (rc == PPP_OK) rc = ppp_num_waypoints(h, &W);
        if (rc != PPP_OK) { std::printf("error %d: %s\n", rc, ppp_last_error(h)); return 1; }
        std::printf("cloud %zu: set_cloud_device %.1f  run_async %.1f  sync %.1f  total %.1f us   W %zu\n", i, us(t0, t1), us(t1, t2), us(t2, t3), us(t0, t3), W);
        if (i >= 2) { tot.push_back(us(t0, t3)); a_.push_back(us(t0, t1)); b_.push_back(us(t1, t2)); c_.push_back(us(t2, t3)); }
    }
    if (std::getenv("PPP_COLD_STREAM")) {
        /* a stream of never-seen clouds through the planner queue (ppp_queue_*: handles taking turns; a lane waits for its own pass of
           `lanes` clouds ago, takes the next cloud without waiting for its bounds and enqueues its pass while the other lanes' passes
           run).  Per cloud = the whole loop / clouds; every list's row count is read back on the way. */
        for (int lanes = 1; lanes <= 3; ++lanes) {
            ppp_queue q = nullptr;
            if (ppp_queue_create(0, lanes, &p, &q) != PPP_OK) return 1;
            const int rounds = 60;
            std::vector<long long> tk((size_t)rounds, -1);
            size_t Wsum = 0;
            auto run = [&](int count) -> int {
                for (int k = 0; k < count; ++k) {
                    if (k >= lanes) { size_t W = 0; if (ppp_queue_wait(q, tk[(size_t)(k - lanes)], &W, nullptr) != PPP_OK) return 1; Wsum += W; }
                    const size_t i = (size_t)k % dev.size();
                    if (ppp_queue_submit(q, dev[i], ns[i], 12, nullptr, &tk[(size_t)k]) != PPP_OK) return 1;
                }
                for (int k = std::max(0, count - lanes); k < count; ++k) { size_t W = 0; if (ppp_queue_wait(q, tk[(size_t)k], &W, nullptr) != PPP_OK) return 1; Wsum += W; }
                return 0;
            };
            if (run(3 * lanes)) { std::printf("stream: %s\n", ppp_queue_last_error(q)); return 1; } /* every lane warm: a plan of this size each */
            Wsum = 0;
            const auto s0 = std::chrono::steady_clock::now();
            if (run(rounds)) { std::printf("stream: %s\n", ppp_queue_last_error(q)); return 1; }
            const double total = us(s0, std::chrono::steady_clock::now());
            std::printf("stream of never-seen clouds through the planner queue, %d lane(s): %d clouds in %.1f us = %.1f us per cloud (%zu waypoints)\n", lanes, rounds, total,
                        total / rounds, Wsum);
            ppp_queue_destroy(q);
        }
    }
    std::sort(tot.begin(), tot.end());
    std::printf("C caller, clouds 2..: min set_cloud_device %.1f, run_async %.1f, sync %.1f; total min %.1f, median %.1f us\n", *std::min_element(a_.begin(), a_.end()),
                *std::min_element(b_.begin(), b_.end()), *std::min_element(c_.begin(), c_.end()), tot.front(), tot[tot.size() / 2]);
    ppp_destroy(h);
    return 0;
}
