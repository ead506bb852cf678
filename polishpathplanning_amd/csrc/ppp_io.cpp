/*
 * ppp_io.cpp -- host-side file formats of the reference: PCD v0.7 in, config.txt in,
 * pathFile out.  No device code; part of libppp_hip.so so the drop-in classes (ppp_host.cpp)
 * and the bindings share one implementation.
 */
#include "../../include/ppp_hip.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cerrno>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace {
struct Field { std::string name; int size = 4; char type = 'F'; int count = 1; int offset = 0; };

double read_scalar(const unsigned char *p, int size, char type)
{
    if (type == 'F') {
        if (size == 4) { float v; memcpy(&v, p, 4); return v; }
        if (size == 8) { double v; memcpy(&v, p, 8); return v; }
    } else if (type == 'I') {
        if (size == 1) { int8_t v; memcpy(&v, p, 1); return v; }
        if (size == 2) { int16_t v; memcpy(&v, p, 2); return v; }
        if (size == 4) { int32_t v; memcpy(&v, p, 4); return v; }
        if (size == 8) { int64_t v; memcpy(&v, p, 8); return (double)v; }
    } else if (type == 'U') {
        if (size == 1) { uint8_t v; memcpy(&v, p, 1); return v; }
        if (size == 2) { uint16_t v; memcpy(&v, p, 2); return v; }
        if (size == 4) { uint32_t v; memcpy(&v, p, 4); return v; }
        if (size == 8) { uint64_t v; memcpy(&v, p, 8); return (double)v; }
    }
    return NAN;
}

/* LZF (Marc Lehmann's liblzf, the codec of PCD "binary_compressed"): control byte < 32 = literal run of
   ctrl + 1 bytes; otherwise a back reference of length (ctrl >> 5) + 2 (7 = one more length byte follows)
   at distance ((ctrl & 31) << 8 | next byte) + 1.  Returns the decoded size, 0 on malformed input. */
size_t lzf_decompress(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len)
{
    const unsigned char *ip = in, *in_end = in + in_len;
    unsigned char *op = out, *out_end = out + out_len;
    while (ip < in_end) {
        unsigned ctrl = *ip++;
        if (ctrl < 32) {
            ctrl++;
            if (op + ctrl > out_end || ip + ctrl > in_end) return 0;
            memcpy(op, ip, ctrl);
            op += ctrl; ip += ctrl;
        } else {
            size_t len = ctrl >> 5;
            if (ip >= in_end) return 0;
            if (len == 7) { len += *ip++; if (ip >= in_end) return 0; }
            const size_t dist = ((size_t)(ctrl & 0x1f) << 8) + *ip++ + 1;
            len += 2;
            if (dist > (size_t)(op - out) || op + len > out_end) return 0;
            const unsigned char *ref = op - dist;
            for (size_t k = 0; k < len; ++k) op[k] = ref[k]; /* may overlap: byte by byte */
            op += len;
        }
    }
    return (size_t)(op - out);
}

/* greedy LZF encoder (3-byte hash, distances <= 8192, matches <= 264); any liblzf decoder reads it */
std::vector<unsigned char> lzf_compress(const unsigned char *in, size_t n)
{
    std::vector<unsigned char> out;
    out.reserve(n + n / 16 + 64);
    std::vector<long long> table(1 << 14, -1);
    size_t lit_start = 0, i = 0;
    auto flush_literals = [&](size_t end) {
        while (lit_start < end) {
            size_t run = std::min<size_t>(32, end - lit_start);
            out.push_back((unsigned char)(run - 1));
            out.insert(out.end(), in + lit_start, in + lit_start + run);
            lit_start += run;
        }
    };
    while (i + 2 < n) {
        const unsigned h = ((in[i] << 16 | in[i + 1] << 8 | in[i + 2]) * 2654435761u) >> 18;
        const long long cand = table[h];
        table[h] = (long long)i;
        if (cand >= 0 && i - (size_t)cand <= 8192 && in[cand] == in[i] && in[cand + 1] == in[i + 1] && in[cand + 2] == in[i + 2]) {
            size_t len = 3;
            while (i + len < n && len < 264 && in[cand + len] == in[i + len]) ++len;
            flush_literals(i);
            const size_t dist = i - (size_t)cand - 1, l = len - 2;
            if (l < 7) out.push_back((unsigned char)((l << 5) | (dist >> 8)));
            else { out.push_back((unsigned char)((7u << 5) | (dist >> 8))); out.push_back((unsigned char)(l - 7)); }
            out.push_back((unsigned char)(dist & 0xff));
            i += len;
            lit_start = i;
        } else ++i;
    }
    flush_literals(n);
    return out;
}
} // namespace

extern "C" {

void ppp_free(void *p) { free(p); }

static int load_pcd_impl(const char *path, float **xyz, size_t *n, float viewpoint[7])
{
    if (!path || !xyz || !n) return PPP_ERR_ARG;
    *xyz = nullptr; *n = 0;
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return PPP_ERR_IO;
    std::vector<Field> fields;
    size_t points = 0, width = 0, height = 1;
    bool have_points = false;
    float vp[7] = {0, 0, 0, 1, 0, 0, 0};
    std::string data_kind, line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ls(line);
        std::string key;
        ls >> key;
        if (key == "VERSION") continue;
        if (key == "FIELDS" || key == "COLUMNS") { std::string nm; while (ls >> nm) { Field fd; fd.name = nm; fields.push_back(fd); } }
        else if (key == "SIZE") { for (auto &fd : fields) ls >> fd.size; }
        else if (key == "TYPE") { for (auto &fd : fields) ls >> fd.type; }
        else if (key == "COUNT") { for (auto &fd : fields) ls >> fd.count; }
        else if (key == "WIDTH") ls >> width;
        else if (key == "HEIGHT") ls >> height;
        else if (key == "VIEWPOINT") { for (int i = 0; i < 7; ++i) ls >> vp[i]; }
        else if (key == "POINTS") { ls >> points; have_points = true; }
        else if (key == "DATA") { ls >> data_kind; break; }
    }
    if (!have_points) points = width * height;
    if (fields.empty() || data_kind.empty()) return PPP_ERR_IO;
    /* bytes left after the header: sizes the header (or a compressed block) claims beyond that are refused before
       anything is allocated for them */
    const std::streamoff data_pos = f.tellg();
    f.seekg(0, std::ios::end);
    const size_t remaining = (size_t)std::max<std::streamoff>(0, f.tellg() - data_pos);
    f.seekg(data_pos);
    int off = 0, ix = -1, iy = -1, iz = -1;
    for (size_t i = 0; i < fields.size(); ++i) {
        /* PCL's field sizes are 1, 2, 4 or 8 bytes; a negative or absurd SIZE / COUNT would turn the record offsets below
           into reads outside the record */
        const int sz = fields[i].size, cnt = fields[i].count;
        if (!(sz == 1 || sz == 2 || sz == 4 || sz == 8) || cnt < 0 || cnt > (1 << 20)) return PPP_ERR_IO;
        const long long step = (long long)sz * std::max(1, cnt);
        if ((long long)off + step > (1ll << 30)) return PPP_ERR_IO;
        fields[i].offset = off;
        off += (int)step;
        if (fields[i].name == "x") ix = (int)i;
        if (fields[i].name == "y") iy = (int)i;
        if (fields[i].name == "z") iz = (int)i;
    }
    if (ix < 0 || iy < 0 || iz < 0) return PPP_ERR_IO;
    for (int q : {ix, iy, iz}) { /* x, y, z are read as F4 / F8 (or an integer type of their size) from inside the record */
        if (fields[q].offset < 0 || fields[q].offset + fields[q].size > off) return PPP_ERR_IO;
    }
    if (data_kind == "binary" && (points > remaining || (size_t)off * points > remaining)) return PPP_ERR_IO;
    if (data_kind == "ascii" && points > remaining) return PPP_ERR_IO; /* a point takes at least one byte */
    uint32_t csize = 0, usize = 0;
    if (data_kind == "binary_compressed") {
        /* pcl::PCDReader: uint32 compressed size, uint32 uncompressed size, LZF stream.  Both sizes are checked against the
           file and the header before anything is allocated: an LZF stream expands by less than 90x (a 3-byte token copies
           at most 264 bytes) */
        unsigned char hdr[8];
        f.read((char *)hdr, 8);
        if (f.gcount() != 8) return PPP_ERR_IO;
        memcpy(&csize, hdr, 4); memcpy(&usize, hdr + 4, 4);
        if ((size_t)csize + 8 > remaining || (size_t)usize > (size_t)csize * 90 + 64) return PPP_ERR_IO;
        if (off <= 0 || points > (size_t)usize || (size_t)usize != (size_t)off * points) return PPP_ERR_IO;
    }
    if (off <= 0 || points > ((size_t)1 << 40)) return PPP_ERR_IO;
    float *out = (float *)malloc(sizeof(float) * 3 * std::max<size_t>(points, 1));
    if (!out) return PPP_ERR_IO;
    if (data_kind == "ascii") {
        size_t got = 0;
        while (got < points && std::getline(f, line)) {
            if (line.empty()) continue;
            std::istringstream ls(line);
            std::string tok;
            float v[3] = {NAN, NAN, NAN};
            for (size_t i = 0; i < fields.size(); ++i) {
                for (int c = 0; c < std::max(1, fields[i].count); ++c) {
                    if (!(ls >> tok)) break;
                    if (c == 0 && ((int)i == ix || (int)i == iy || (int)i == iz)) {
                        float val = (float)strtod(tok.c_str(), nullptr); /* accepts nan / inf */
                        v[(int)i == ix ? 0 : ((int)i == iy ? 1 : 2)] = val;
                    }
                }
            }
            memcpy(out + 3 * got, v, 12);
            ++got;
        }
        if (got != points) { free(out); return PPP_ERR_IO; }
    } else if (data_kind == "binary") {
        std::vector<unsigned char> rec((size_t)off * points);
        f.read((char *)rec.data(), (std::streamsize)rec.size());
        if ((size_t)f.gcount() != rec.size()) { free(out); return PPP_ERR_IO; }
        for (size_t i = 0; i < points; ++i) {
            const unsigned char *p = rec.data() + i * off;
            out[3 * i + 0] = (float)read_scalar(p + fields[ix].offset, fields[ix].size, fields[ix].type);
            out[3 * i + 1] = (float)read_scalar(p + fields[iy].offset, fields[iy].size, fields[iy].type);
            out[3 * i + 2] = (float)read_scalar(p + fields[iz].offset, fields[iz].size, fields[iz].type);
        }
    } else if (data_kind == "binary_compressed") {
        /* pcl::PCDReader: uint32 compressed size, uint32 uncompressed size, LZF stream; the decoded block is
           field-major (all x, then all y, ...) */
        std::vector<unsigned char> comp(csize), raw(usize);
        f.read((char *)comp.data(), (std::streamsize)csize);
        if ((size_t)f.gcount() != (size_t)csize) { free(out); return PPP_ERR_IO; }
        if (usize && lzf_decompress(comp.data(), csize, raw.data(), usize) != usize) { free(out); return PPP_ERR_IO; }
        const int idx3[3] = {ix, iy, iz};
        for (int d = 0; d < 3; ++d) {
            const Field &fd = fields[idx3[d]];
            const size_t per = (size_t)fd.size * std::max(1, fd.count);
            const unsigned char *base = raw.data() + (size_t)fd.offset * points; /* blocks follow the record order */
            for (size_t i = 0; i < points; ++i) out[3 * i + d] = (float)read_scalar(base + i * per, fd.size, fd.type);
        }
    } else {
        free(out);
        return PPP_ERR_UNSUPPORTED;
    }
    *xyz = out; *n = points;
    if (viewpoint) memcpy(viewpoint, vp, sizeof(vp));
    return PPP_OK;
}

static int save_pcd_impl(const char *path, const float *xyz, size_t n, size_t stride_floats, const float viewpoint[7], int binary)
{
    if (!path || (!xyz && n) || stride_floats < 3) return PPP_ERR_ARG;
    FILE *f = fopen(path, "wb");
    if (!f) return PPP_ERR_IO;
    const float dvp[7] = {0, 0, 0, 1, 0, 0, 0};
    const float *vp = viewpoint ? viewpoint : dvp;
    fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n");
    fprintf(f, "WIDTH %zu\nHEIGHT 1\nVIEWPOINT %g %g %g %g %g %g %g\nPOINTS %zu\nDATA %s\n", n, vp[0], vp[1], vp[2], vp[3], vp[4], vp[5],
            vp[6], n, binary == 2 ? "binary_compressed" : (binary ? "binary" : "ascii"));
    if (binary == 2) { /* pcl::PCDWriter::writeBinaryCompressed layout */
        std::vector<float> soa(3 * n);
        for (size_t i = 0; i < n; ++i) for (int d = 0; d < 3; ++d) soa[(size_t)d * n + i] = xyz[i * stride_floats + d];
        std::vector<unsigned char> comp = lzf_compress((const unsigned char *)soa.data(), 12 * n);
        if (comp.size() > 0xffffffffull || 12 * n > 0xffffffffull) { fclose(f); return PPP_ERR_CAPACITY; }
        const uint32_t csize = (uint32_t)comp.size(), usize = (uint32_t)(12 * n);
        fwrite(&csize, 4, 1, f); fwrite(&usize, 4, 1, f);
        if (csize) fwrite(comp.data(), 1, csize, f);
        fclose(f);
        return PPP_OK;
    }
    for (size_t i = 0; i < n; ++i) {
        const float *p = xyz + i * stride_floats;
        if (binary) fwrite(p, 4, 3, f);
        else fprintf(f, "%.9g %.9g %.9g\n", p[0], p[1], p[2]);
    }
    fclose(f);
    return PPP_OK;
}

/* pcl::PointXYZRGB cloud as PCL writes it: FIELDS x y z rgb, the colour packed 0x00RRGGBB in one 4-byte field */
static int save_pcd_rgb_impl(const char *path, const float *xyz, const unsigned char *rgb, size_t n, const float viewpoint[7], int binary)
{
    if (!path || ((!xyz || !rgb) && n) || binary < 0 || binary > 1) return PPP_ERR_ARG;
    FILE *f = fopen(path, "wb");
    if (!f) return PPP_ERR_IO;
    const float dvp[7] = {0, 0, 0, 1, 0, 0, 0};
    const float *vp = viewpoint ? viewpoint : dvp;
    fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\n");
    fprintf(f, "WIDTH %zu\nHEIGHT 1\nVIEWPOINT %g %g %g %g %g %g %g\nPOINTS %zu\nDATA %s\n", n, vp[0], vp[1], vp[2], vp[3], vp[4], vp[5],
            vp[6], n, binary ? "binary" : "ascii");
    for (size_t i = 0; i < n; ++i) {
        const float *p = xyz + 3 * i;
        const uint32_t c = ((uint32_t)rgb[3 * i] << 16) | ((uint32_t)rgb[3 * i + 1] << 8) | (uint32_t)rgb[3 * i + 2];
        if (binary) { fwrite(p, 4, 3, f); fwrite(&c, 4, 1, f); }
        else fprintf(f, "%.9g %.9g %.9g %u\n", p[0], p[1], p[2], c);
    }
    fclose(f);
    return PPP_OK;
}

void ppp_default_config(ppp_config *c)
{   /* config.txt:1-13 */
    memset(c, 0, sizeof(*c));
    ppp_default_params(&c->params);
    snprintf(c->path_file, sizeof(c->path_file), "WayPoints_test2.txt");
    c->depth = 0.01; c->adjust_threshold = 1; c->toolthickness = 10;
    c->smooth_cloud = 0; c->remove_outlier = 0; c->alignment = 0; c->dynamic_adjustment = 1;
}

/* std::stod's acceptance without its exceptions: leading whitespace, the longest numeric prefix, trailing text ignored;
   no conversion or a value out of range is an error (std::stod throws there, out of the reference's constructor) */
static bool parse_double(const std::string &v, double *out)
{
    errno = 0;
    char *end = nullptr;
    const double d = strtod(v.c_str(), &end);
    if (end == v.c_str() || errno == ERANGE) return false;
    *out = d;
    return true;
}

static int read_config_impl(const char *path, ppp_config *c)
{
    if (!path || !c) return PPP_ERR_ARG;
    std::ifstream cFile(path);
    if (!cFile.is_open()) {
        std::cerr << "Couldn't open config file for reading.\n"; /* path_slicing_alg.cpp:36 */
        return PPP_ERR_IO;
    }
    std::string line;
    while (std::getline(cFile, line)) {
        line.erase(std::remove_if(line.begin(), line.end(), [](unsigned char ch) { return std::isspace(ch); }), line.end());
        auto pos = line.find("=");
        if (line.empty() || line[0] == '#' || pos == std::string::npos) continue;
        std::string name = line.substr(0, pos), value = line.substr(pos + 1);
        double d = 0;
        bool ok = true;
        if (name == "pathFile") snprintf(c->path_file, sizeof(c->path_file), "%s", value.c_str());
        else if (name == "Tool_Radius") { if ((ok = parse_double(value, &d))) c->params.tool_radius = d; }
        else if (name == "depth") { if ((ok = parse_double(value, &d))) c->depth = d; }
        else if (name == "Adjust_Threshold") { if ((ok = parse_double(value, &d))) c->adjust_threshold = d; }
        else if (name == "toolthickness") { if ((ok = parse_double(value, &d))) c->toolthickness = d; }
        else if (name == "PathResolution") { if ((ok = parse_double(value, &d))) c->params.path_resolution = d; }
        else if (name == "RPYresolution") { if ((ok = parse_double(value, &d))) c->params.rpy_resolution = d; }
        else if (name == "Endeffectorlength") { if ((ok = parse_double(value, &d))) c->params.ee_length = (float)d; }
        else if (name == "Alignment") c->alignment = value == "true";
        else if (name == "Smooth") c->smooth_cloud = value == "true";
        else if (name == "ChangeRange") c->params.change_range = value == "true";
        else if (name == "RemoveOutlier") c->remove_outlier = value == "true";
        else if (name == "Dynamic_adjustment") c->dynamic_adjustment = value == "true";
        if (!ok) return PPP_ERR_ARG;
    }
    /* the adjustment parameters travel inside ppp_params */
    c->params.depth = c->depth; c->params.adjust_threshold = c->adjust_threshold; c->params.toolthickness = c->toolthickness;
    return PPP_OK;
}

static int write_path_file_impl(const char *path, const float *wp6, size_t W)
{
    if (!path || (!wp6 && W)) return PPP_ERR_ARG;
    std::ofstream outputFile(path);
    if (!outputFile.is_open()) {
        std::cerr << "Unable to open file: " << path << std::endl;
        return PPP_ERR_IO;
    }
    for (size_t w = 0; w < W; ++w) {
        for (int i = 0; i < 6; i++) outputFile << wp6[6 * w + i] << " ";
        outputFile << std::endl;
    }
    outputFile.close();
    return PPP_OK;
}


/* The boundary never lets a C++ exception (std::bad_alloc on a header that promises 10^12 points, ...) escape. */
int ppp_load_pcd(const char *path, float **xyz, size_t *n, float viewpoint[7])
{
    try { return load_pcd_impl(path, xyz, n, viewpoint); } catch (...) { if (xyz) *xyz = nullptr; if (n) *n = 0; return PPP_ERR_IO; }
}
int ppp_save_pcd(const char *path, const float *xyz, size_t n, size_t stride_floats, const float viewpoint[7], int binary)
{
    try { return save_pcd_impl(path, xyz, n, stride_floats, viewpoint, binary); } catch (...) { return PPP_ERR_IO; }
}
int ppp_save_pcd_rgb(const char *path, const float *xyz, const unsigned char *rgb, size_t n, const float viewpoint[7], int binary)
{
    try { return save_pcd_rgb_impl(path, xyz, rgb, n, viewpoint, binary); } catch (...) { return PPP_ERR_IO; }
}
int ppp_read_config(const char *path, ppp_config *c)
{
    try { return read_config_impl(path, c); } catch (...) { return PPP_ERR_IO; }
}
int ppp_write_path_file(const char *path, const float *wp6, size_t W)
{
    try { return write_path_file_impl(path, wp6, W); } catch (...) { return PPP_ERR_IO; }
}
} /* extern "C" */
