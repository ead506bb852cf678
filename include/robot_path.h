/*
 * robot_path.h -- drop-in for the reference's include/robot_path.h (class RobotPath, :58-98).
 * The reference header does not compile (its constructor is mis-named SectPath, :61-65) and no
 * .cpp implements it; its member list is the July snapshot src/path_connect_ex0720.cpp:
 * single-direction float walk from min.x + Radius, +-5 mm trim, no first/last drop, no position
 * smoothing, hand-eye constants of robot_path.h:36-41.  That behaviour is what this class runs.
 */
#ifndef ROBOT_PATH_H
#define ROBOT_PATH_H

#include <string>
#include <vector>
#include "Spline.h"

class RobotPath {
public:
    RobotPath() {}
    RobotPath(std::string configName, std::string CloudFileName, double Radius) : cloud_name(CloudFileName)
    {
        ppp_read_config(configName.c_str(), &planner.config());
        ppp_params &p = planner.config().params;
        p.tool_radius = Radius; p.pairing = PPP_PAIR_KD; p.walk = PPP_WALK_V1_CONTACT;
        p.trim = 5; p.drop_ends = 0; p.smooth = 0; /* path_connect_ex0720.cpp:440-448 */
        const float he[6] = {0.792078f, -0.042662f, 0.6656017f, -3.1531625f, -0.048573f, 1.609157f};
        for (int i = 0; i < 6; ++i) p.handeye[i] = he[i];
        planner.open(cloud_name);
    }
    void show() { planner.show_notice(); }
    void estimate_normal() { planner.estimate_normal(); } /* path_connect_ex0720.cpp: pcl::NormalEstimation, whole cloud */
    const std::vector<float> &cloud_normals() const { return planner.cloud_normals(); }
    void GenPath()
    {
        if (!planner.gen_path()) return;
        Path_set.clear();
        for (int s = 0; s < planner.num_slices(); ++s) Path_set.emplace_back(planner.handle(), s);
    }
    void getPath()
    {
        std::vector<float> wp;
        if (!planner.get_path(wp)) return;
        WayPointsList.assign(wp.size() / 6, std::vector<float>(6));
        for (size_t w = 0; w < WayPointsList.size(); ++w)
            for (int d = 0; d < 6; ++d) WayPointsList[w][d] = wp[6 * w + d];
    }
    const std::vector<std::vector<float>> &waypoints() const { return WayPointsList; }

private:
    ppp::Planner planner;
    std::vector<Spline> Path_set;
    std::string cloud_name;
    std::vector<std::vector<float>> WayPointsList;
};

#endif
