// Mirror of the reference's src/connect.cpp / src/connect1.cpp (:7-30): ./connect cloud.pcd
// reads ../config.txt (or $PPP_CONFIG), plans, writes pathFile.  Build: see examples/Makefile.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include "Path_Generate_Algorithm.h"

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
    path_generater path_planner = {configFile, pcd};
    path_planner.GenPath();
    path_planner.getPath();
    path_planner.show();
    return 0;
}
