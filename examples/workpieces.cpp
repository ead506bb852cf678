// A run of workpieces in one process: ./workpieces a.pcd b.pcd c.pcd ...
// Every cloud gets a planner of its own, constructed from the file name and driven exactly like src/connect.cpp:21-27 of the
// reference does for its single cloud (constructor, GenPath, getPath); the list of workpiece i is left in "<pathFile>.<i>".
// The planners after the first take the engine handle the one before them gave back (ppp::HandlePool, ppp_planner.hpp): no
// second ppp_create, and no window census for a cloud of the size and parameters of the last one.  The last line reports how
// many planners were served from the pool and the wall time of each workpiece.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>
#include "Path_Generate_Algorithm.h"

int main(int argc, char **argv)
{
    std::vector<std::string> pcds;
    for (int i = 1; i < argc; ++i) {
        size_t n = strlen(argv[i]);
        if (n > 4 && strcmp(argv[i] + n - 4, ".pcd") == 0) pcds.push_back(argv[i]);
    }
    if (pcds.empty()) {
        std::cout << "./workpieces cad_name.pcd [more.pcd ...]" << std::endl;
        return (-1);
    }
    const char *cfg = std::getenv("PPP_CONFIG");
    std::string configFile = cfg ? cfg : "../config.txt";
    std::string pathFile;
    {
        ppp_config c;
        ppp_default_config(&c);
        if (ppp_read_config(configFile.c_str(), &c) == PPP_OK) pathFile = c.path_file;
    }
    std::vector<double> ms, ms_open, ms_gen, ms_get;
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    for (size_t i = 0; i < pcds.size(); ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        size_t W = 0;
        {
            path_generater path_planner = {configFile, pcds[i]}; /* file -> HBM, plan */
            ms_open.push_back(since(t0));
            const auto t1 = std::chrono::steady_clock::now();
            path_planner.GenPath();
            ms_gen.push_back(since(t1));
            const auto t2 = std::chrono::steady_clock::now();
            path_planner.getPath();                              /* the pass's second half, list to the host, pathFile */
            ms_get.push_back(since(t2));
            W = path_planner.waypoints().size();
        }
        ms.push_back(since(t0));
        if (!W) {
            std::fprintf(stderr, "workpieces: no path for %s\n", pcds[i].c_str());
            return 1;
        }
        const std::string kept = pathFile + "." + std::to_string(i);
        if (!pathFile.empty() && std::rename(pathFile.c_str(), kept.c_str()) != 0) {
            std::fprintf(stderr, "workpieces: could not keep %s\n", kept.c_str());
            return 1;
        }
    }
    std::printf("workpieces: %zu planned, %zu planners served from the handle pool; ms per workpiece:", pcds.size(), ppp::HandlePool::taken_from_pool());
    for (double m : ms) std::printf(" %.2f", m);
    std::printf("\nworkpieces: of which constructor / GenPath / getPath:");
    for (size_t i = 0; i < ms.size(); ++i) std::printf(" %.2f/%.2f/%.2f", ms_open[i], ms_gen[i], ms_get[i]);
    std::printf("\n");
    return 0;
}
