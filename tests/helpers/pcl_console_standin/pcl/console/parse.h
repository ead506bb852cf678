/*
 * TEST INFRASTRUCTURE -- stand-in for PCL's <pcl/console/parse.h>, used only by
 * tests/test_host_logic.py::test_reference_drivers_compile_unchanged to compile the reference's
 * src/{main,connect,connect1,contour}.cpp against the drop-in headers in an image without PCL.
 * The drivers use exactly one function of it (src/connect.cpp:17): the positions of the
 * arguments that end in the given extension (case-insensitive, longer than 4 characters),
 * as pcl::console::parse_file_extension_argument of PCL 1.10-1.12 returns them.
 * A maintainer's build uses the real PCL console headers (libpcl_common) instead.
 */
#pragma once
#include <algorithm>
#include <cctype>
#include <string>
#include <vector>

namespace pcl {
namespace console {
inline std::vector<int> parse_file_extension_argument(int argc, const char *const *argv, const std::string &extension)
{
    std::vector<int> indices;
    for (int i = 1; i < argc; ++i) {
        std::string fname(argv[i]), ext(extension);
        if (fname.size() <= 4) continue;
        std::transform(fname.begin(), fname.end(), fname.begin(), [](unsigned char c) { return (char)std::tolower(c); });
        std::transform(ext.begin(), ext.end(), ext.begin(), [](unsigned char c) { return (char)std::tolower(c); });
        const std::string::size_type it = fname.rfind(ext);
        if (it != std::string::npos && ext.size() == fname.size() - it) indices.push_back(i);
    }
    return indices;
}
} // namespace console
} // namespace pcl
