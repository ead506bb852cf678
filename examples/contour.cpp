// Mirror of the reference's src/contour.cpp (:7-29): ./contour cloud.pcd constructs the contour planner and calls show().
// With PPP_CONTOUR_PLAN=1 it also plans and writes pathFile (GenPath + getPath), which the reference leaves to its viewer session.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include "contour_alg.h"

int main(int argc, char **argv)
{
    std::string pcd;
    for (int i = 1; i < argc; ++i) {
        size_t n = strlen(argv[i]);
        if (n > 4 && strcmp(argv[i] + n - 4, ".pcd") == 0) pcd = argv[i];
    }
    if (pcd.empty()) {
        std::cout << "./slicing_method cad_name.pcd" << std::endl;
        return (-1);
    }
    const char *cfg = std::getenv("PPP_CONFIG");
    std::string configFile = cfg ? cfg : "../config.txt";
    SectPath path_planner = {configFile, pcd};
    const char *plan = std::getenv("PPP_CONTOUR_PLAN");
    if (plan && plan[0] == '1') {
        path_planner.GenPath();
        path_planner.getPath();
    }
    path_planner.show();
    return 0;
}
