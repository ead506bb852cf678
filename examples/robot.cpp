// A caller of the RobotPath drop-in (include/robot_path.h): ./robot cloud.pcd [radius].  The reference declares the class
// (robot_path.h:58-98) but nothing constructs it -- the header does not compile upstream -- so this is the shape of
// src/connect.cpp with the three-argument constructor.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include "ppp_planner.hpp"
#include "robot_path.h"

int main(int argc, char **argv)
{
    std::string pcd;
    double radius = 6;
    for (int i = 1; i < argc; ++i) {
        size_t n = strlen(argv[i]);
        if (n > 4 && strcmp(argv[i] + n - 4, ".pcd") == 0) pcd = argv[i];
        else radius = atof(argv[i]);
    }
    if (pcd.empty()) {
        std::cout << "./robot cad_name.pcd [radius]" << std::endl;
        return (-1);
    }
    const char *cfg = std::getenv("PPP_CONFIG");
    std::string configFile = cfg ? cfg : "../config.txt";
    RobotPath path_planner(configFile, pcd, radius);
    path_planner.GenPath();
    path_planner.getPath();
    path_planner.show();
    std::cout << "waypoints: " << path_planner.waypoints().size() << std::endl;
    return 0;
}
