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
#include <charconv>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
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
            /* a run is 1..32 bytes: away from the buffers' ends two fixed 16-byte moves beat a variable-length copy */
            if ((size_t)(out_end - op) >= 32 && (size_t)(in_end - ip) >= 32) { memcpy(op, ip, 16); memcpy(op + 16, ip + 16, 16); }
            else memcpy(op, ip, ctrl);
            op += ctrl; ip += ctrl;
        } else {
            size_t len = ctrl >> 5;
            if (ip >= in_end) return 0;
            if (len == 7) { len += *ip++; if (ip >= in_end) return 0; }
            const size_t dist = ((size_t)(ctrl & 0x1f) << 8) + *ip++ + 1;
            len += 2;
            if (dist > (size_t)(op - out) || op + len > out_end) return 0;
            const unsigned char *ref = op - dist;
            if (dist >= len) memcpy(op, ref, len);
            else for (size_t k = 0; k < len; ++k) op[k] = ref[k]; /* overlapping: byte by byte */
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
/* A decimal token to float, correctly rounded (= strtof, which is what `istringstream >> float` calls).  Fast path: up to 19
   significant digits w and a decimal exponent with |e| <= 22 make w and 10^|e| exact doubles, so d = w * 10^e (or w / 10^-e) is
   the correctly rounded double; its cast to float is the correctly rounded float unless d sits within an ulp of the midpoint of
   two floats (the 29 bits a float drops read 0x0fffffff .. 0x10000001) -- then, and for anything that is not a plain decimal
   in the comfortable range, strtof decides.  (std::from_chars of this compiler's library takes 48 ns a number: it wraps strtod.) */
static float parse_float_token(const char *t0, const char *lim, const char **tok_end)
{
    static const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char *q = t0;
    bool neg = false;
    if (q < lim && (*q == '-' || *q == '+')) { neg = *q == '-'; ++q; }
    uint64_t w = 0;
    int digits = 0, e10 = 0;
    bool any = false, plain = true;
    while (q < lim && (unsigned)(*q - '0') <= 9u) { any = true; if (w || *q != '0') { if (digits < 19) { w = w * 10 + (uint64_t)(*q - '0'); ++digits; } else plain = false; } ++q; }
    if (q < lim && *q == '.') {
        ++q;
        while (q < lim && (unsigned)(*q - '0') <= 9u) { any = true; if (w || *q != '0') { if (digits < 19) { w = w * 10 + (uint64_t)(*q - '0'); ++digits; --e10; } else plain = false; } else --e10; ++q; }
    }
    if (any && q < lim && (*q == 'e' || *q == 'E')) {
        const char *r = q + 1;
        bool eneg = false;
        if (r < lim && (*r == '-' || *r == '+')) { eneg = *r == '-'; ++r; }
        int ev = 0, ed = 0;
        while (r < lim && (unsigned)(*r - '0') <= 9u) { if (ed < 6) ev = ev * 10 + (*r - '0'); ++ed; ++r; }
        if (ed == 0 || ed >= 6) plain = false; else { e10 += eneg ? -ev : ev; q = r; }
    }
    const bool at_end = q >= lim || (unsigned char)*q <= ' '; /* the token is over where the number is */
    if (at_end) *tok_end = q;
    else { while (q < lim && (unsigned char)*q > ' ') ++q; *tok_end = q; plain = false; }
    if (any && plain && w < ((uint64_t)1 << 53) && e10 >= -22 && e10 <= 22) {
        if (w == 0) return neg ? -0.f : 0.f;
        const double d = e10 >= 0 ? (double)w * P10[e10] : (double)w / P10[-e10];
        if (d > 1e-30 && d < 1e30) {
            uint64_t bits;
            memcpy(&bits, &d, 8);
            const uint32_t low = (uint32_t)(bits & 0x1fffffff);
            if (low < 0x0fffffffu || low > 0x10000001u) return neg ? -(float)d : (float)d;
        }
    }
    char tmp[64];
    const size_t tl = std::min<size_t>((size_t)(*tok_end - t0), sizeof(tmp) - 1);
    memcpy(tmp, t0, tl);
    tmp[tl] = 0;
    return strtof(tmp, nullptr); /* 0 for no number at all, nan / inf in any spelling, hex */
}
/* fn(0) .. fn(parts - 1), the first on this thread, the others on threads of their own; a thread that cannot be started has its
   part run here instead (nothing is left joinable behind an exception) */
template <typename F>
static void run_parts(size_t parts, F &&fn)
{
    std::vector<std::thread> th;
    th.reserve(parts);
    std::vector<size_t> here(1, 0);
    for (size_t t = 1; t < parts; ++t) {
        try { th.emplace_back(fn, t); } catch (...) { here.push_back(t); }
    }
    for (size_t t : here) fn(t);
    for (auto &x : th) x.join();
}
} // namespace

extern "C" {

void ppp_free(void *p) { free(p); }

/* the header of a PCD v0.7 file, checked: where the payload starts, how many bytes follow it, where x, y, z sit in a record */
struct PcdHeader {
    std::vector<Field> fields;
    size_t points = 0, remaining = 0;
    float vp[7] = {0, 0, 0, 1, 0, 0, 0};
    std::string data_kind;
    std::streamoff data_pos = 0;
    int off = 0, ix = -1, iy = -1, iz = -1;
};

static int parse_pcd_header(std::ifstream &f, PcdHeader &H)
{
    std::vector<Field> &fields = H.fields;
    size_t width = 0, height = 1;
    bool have_points = false;
    std::string line;
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
        else if (key == "VIEWPOINT") { for (int i = 0; i < 7; ++i) ls >> H.vp[i]; }
        else if (key == "POINTS") { ls >> H.points; have_points = true; }
        else if (key == "DATA") { ls >> H.data_kind; break; }
    }
    if (!have_points) H.points = width * height;
    if (fields.empty() || H.data_kind.empty()) return PPP_ERR_IO;
    /* bytes left after the header: sizes the header (or a compressed block) claims beyond that are refused before
       anything is allocated for them */
    H.data_pos = f.tellg();
    f.seekg(0, std::ios::end);
    H.remaining = (size_t)std::max<std::streamoff>(0, f.tellg() - H.data_pos);
    f.seekg(H.data_pos);
    int off = 0;
    for (size_t i = 0; i < fields.size(); ++i) {
        /* PCL's field sizes are 1, 2, 4 or 8 bytes; a negative or absurd SIZE / COUNT would turn the record offsets below
           into reads outside the record */
        const int sz = fields[i].size, cnt = fields[i].count;
        if (!(sz == 1 || sz == 2 || sz == 4 || sz == 8) || cnt < 0 || cnt > (1 << 20)) return PPP_ERR_IO;
        const long long step = (long long)sz * std::max(1, cnt);
        if ((long long)off + step > (1ll << 30)) return PPP_ERR_IO;
        fields[i].offset = off;
        off += (int)step;
        if (fields[i].name == "x") H.ix = (int)i;
        if (fields[i].name == "y") H.iy = (int)i;
        if (fields[i].name == "z") H.iz = (int)i;
    }
    H.off = off;
    if (H.ix < 0 || H.iy < 0 || H.iz < 0) return PPP_ERR_IO;
    for (int q : {H.ix, H.iy, H.iz}) { /* x, y, z are read as F4 / F8 (or an integer type of their size) from inside the record */
        if (fields[q].offset < 0 || fields[q].offset + fields[q].size > off) return PPP_ERR_IO;
    }
    if (H.data_kind == "binary" && (H.points > H.remaining || (size_t)off * H.points > H.remaining)) return PPP_ERR_IO;
    if (H.data_kind == "ascii" && H.points > H.remaining) return PPP_ERR_IO; /* a point takes at least one byte */
    if (off <= 0 || H.points > ((size_t)1 << 40)) return PPP_ERR_IO;
    return PPP_OK;
}

static int probe_pcd_impl(const char *path, ppp_pcd_layout *L)
{
    if (!path || !L) return PPP_ERR_ARG;
    memset(L, 0, sizeof(*L));
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return PPP_ERR_IO;
    PcdHeader H;
    const int rc = parse_pcd_header(f, H);
    if (rc != PPP_OK) return rc;
    if (H.data_kind == "ascii") L->data_kind = 0;
    else if (H.data_kind == "binary") L->data_kind = 1;
    else if (H.data_kind == "binary_compressed") L->data_kind = 2;
    else return PPP_ERR_UNSUPPORTED;
    const Field &fx = H.fields[H.ix], &fy = H.fields[H.iy], &fz = H.fields[H.iz];
    L->points = H.points;
    L->record_bytes = (size_t)H.off;
    L->x_offset = fx.offset; L->y_offset = fy.offset; L->z_offset = fz.offset;
    L->xyz_float32 = fx.type == 'F' && fx.size == 4 && fy.type == 'F' && fy.size == 4 && fz.type == 'F' && fz.size == 4;
    L->data_offset = (long long)H.data_pos;
    memcpy(L->viewpoint, H.vp, sizeof(H.vp));
    return PPP_OK;
}

static int load_pcd_impl(const char *path, float **xyz, size_t *n, float viewpoint[7])
{
    if (!path || !xyz || !n) return PPP_ERR_ARG;
    *xyz = nullptr; *n = 0;
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return PPP_ERR_IO;
    PcdHeader H;
    { const int rc = parse_pcd_header(f, H); if (rc != PPP_OK) return rc; }
    const std::vector<Field> &fields = H.fields;
    const size_t points = H.points, remaining = H.remaining;
    const float *vp = H.vp;
    const std::string &data_kind = H.data_kind;
    const int off = H.off, ix = H.ix, iy = H.iy, iz = H.iz;
    std::string line;
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
    float *out = (float *)malloc(sizeof(float) * 3 * std::max<size_t>(points, 1));
    if (!out) return PPP_ERR_IO;
    if (data_kind == "ascii") {
        /* One read of the rest of the file, then a walk over its lines: token number j of a line belongs to the field that
           covers column j (COUNT columns per field); the first column of x, y and z is parsed, everything else skipped.
           A number becomes the correctly rounded float (parse_float_token), what `istringstream >> float` -- PCL's
           copyStringValue -- gives through strtof; (float)strtod would round twice.  (The first version parsed
           every line through an istringstream: 0.55 s for a 1 M-point file; this: see DESIGN.md 4e.) */
        char *text = (char *)malloc(remaining + 1);
        if (!text) { free(out); return PPP_ERR_IO; }
        f.read(text, (std::streamsize)remaining);
        const size_t len = (size_t)f.gcount();
        text[len] = 0;
        int col_x = -1, col_y = -1, col_z = -1, cols = 0;
        for (size_t i = 0; i < fields.size(); ++i) {
            if ((int)i == ix) col_x = cols;
            if ((int)i == iy) col_y = cols;
            if ((int)i == iz) col_z = cols;
            cols += std::max(1, fields[i].count);
        }
        const int last_col = std::max(col_x, std::max(col_y, col_z));
        /* rows of [p, end): parsed into dst (at most max_rows of them) or, without dst, counted -- a line with no token is no row */
        auto walk = [&](const char *p, const char *end, float *dst, size_t max_rows) -> size_t {
            size_t got = 0;
            while (got < max_rows && p < end) {
                const char *eol = (const char *)memchr(p, '\n', (size_t)(end - p));
                if (!eol) eol = end;
                const char *q = p;
                p = eol < end ? eol + 1 : end;
                if (!dst) {
                    while (q < eol && (unsigned char)*q <= ' ') ++q;
                    if (q < eol) ++got;
                    continue;
                }
                float v[3] = {NAN, NAN, NAN};
                int col = 0;
                bool any = false;
                while (col <= last_col) { /* (columns are separated by blanks; any control character counts as one) */
                    while (q < eol && (unsigned char)*q <= ' ') ++q;
                    if (q >= eol) break;
                    any = true;
                    if (col == col_x || col == col_y || col == col_z) {
                        const char *te;
                        v[col == col_x ? 0 : (col == col_y ? 1 : 2)] = parse_float_token(q, eol, &te);
                        q = te;
                    } else while (q < eol && (unsigned char)*q > ' ') ++q;
                    ++col;
                }
                if (!any) continue; /* an empty line */
                memcpy(dst + 3 * got, v, 12);
                ++got;
            }
            return got;
        };
        size_t got = 0;
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t parts = len >= ((size_t)4 << 20) ? std::min<size_t>(hw ? hw : 1, 8) : 1;
        if (parts <= 1) got = walk(text, text + len, out, points);
        else { /* big files: the text cut at line ends into one piece per thread; rows counted first, so every piece knows where its rows go */
            std::vector<const char *> cut(parts + 1);
            cut[0] = text; cut[parts] = text + len;
            for (size_t t = 1; t < parts; ++t) {
                const char *c = std::max<const char *>(cut[t - 1], text + len * t / parts);
                const char *nl = (const char *)memchr(c, '\n', (size_t)(text + len - c));
                cut[t] = nl ? nl + 1 : text + len;
            }
            std::vector<size_t> rows(parts, 0), first(parts + 1, 0);
            run_parts(parts, [&](size_t t) { rows[t] = walk(cut[t], cut[t + 1], nullptr, (size_t)-1); });
            for (size_t t = 0; t < parts; ++t) first[t + 1] = first[t] + rows[t];
            run_parts(parts, [&](size_t t) {
                if (first[t] < points) walk(cut[t], cut[t + 1], out + 3 * first[t], std::min(rows[t], points - first[t]));
            });
            got = std::min(first[parts], points); /* (rows beyond POINTS are ignored, as in the single walk) */
        }
        free(text);
        if (got != points) { free(out); return PPP_ERR_IO; }
    } else if (data_kind == "binary") {
        const bool f4 = fields[ix].type == 'F' && fields[ix].size == 4 && fields[iy].type == 'F' && fields[iy].size == 4 &&
                        fields[iz].type == 'F' && fields[iz].size == 4;
        const int ox = fields[ix].offset, oy = fields[iy].offset, oz = fields[iz].offset;
        if (f4 && off == 12 && ox == 0 && oy == 4 && oz == 8) { /* records ARE the output: read in place */
            f.read((char *)out, (std::streamsize)(12 * points));
            if ((size_t)f.gcount() != 12 * points) { free(out); return PPP_ERR_IO; }
        } else {
            unsigned char *rec = (unsigned char *)malloc(std::max<size_t>((size_t)off * points, 1));
            if (!rec) { free(out); return PPP_ERR_IO; }
            f.read((char *)rec, (std::streamsize)((size_t)off * points));
            if ((size_t)f.gcount() != (size_t)off * points) { free(rec); free(out); return PPP_ERR_IO; }
            if (f4) { /* the usual "x y z rgb" records: three 4-byte moves per point */
                for (size_t i = 0; i < points; ++i) {
                    const unsigned char *p = rec + i * off;
                    memcpy(out + 3 * i + 0, p + ox, 4);
                    memcpy(out + 3 * i + 1, p + oy, 4);
                    memcpy(out + 3 * i + 2, p + oz, 4);
                }
            } else {
                for (size_t i = 0; i < points; ++i) {
                    const unsigned char *p = rec + i * off;
                    out[3 * i + 0] = (float)read_scalar(p + ox, fields[ix].size, fields[ix].type);
                    out[3 * i + 1] = (float)read_scalar(p + oy, fields[iy].size, fields[iy].type);
                    out[3 * i + 2] = (float)read_scalar(p + oz, fields[iz].size, fields[iz].type);
                }
            }
            free(rec);
        }
    } else if (data_kind == "binary_compressed") {
        /* pcl::PCDReader: uint32 compressed size, uint32 uncompressed size, LZF stream; the decoded block is
           field-major (all x, then all y, ...) */
        std::unique_ptr<unsigned char[]> comp(new unsigned char[(size_t)csize + 1]), raw(new unsigned char[(size_t)usize + 1]); /* (no zero fill) */
        f.read((char *)comp.get(), (std::streamsize)csize);
        if ((size_t)f.gcount() != (size_t)csize) { free(out); return PPP_ERR_IO; }
        if (usize && lzf_decompress(comp.get(), csize, raw.get(), usize) != usize) { free(out); return PPP_ERR_IO; }
        const int idx3[3] = {ix, iy, iz};
        for (int d = 0; d < 3; ++d) {
            const Field &fd = fields[idx3[d]];
            const size_t per = (size_t)fd.size * std::max(1, fd.count);
            const unsigned char *base = raw.get() + (size_t)fd.offset * points; /* blocks follow the record order */
            if (fd.type == 'F' && fd.size == 4) for (size_t i = 0; i < points; ++i) memcpy(out + 3 * i + d, base + i * per, 4);
            else for (size_t i = 0; i < points; ++i) out[3 * i + d] = (float)read_scalar(base + i * per, fd.size, fd.type);
        }
    } else {
        free(out);
        return PPP_ERR_UNSUPPORTED;
    }
    *xyz = out; *n = points;
    if (viewpoint) memcpy(viewpoint, vp, 7 * sizeof(float));
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

/* printf("%g") of a float (promoted, precision 6) -- what `ofstream << float` prints -- without the general-purpose machinery:
   six significant digits of |v| are round(|v| x 10^k) for the k that brings it into [1e5, 1e6).  10^|k| is exact in double up
   to 10^22 and the one multiplication or division is correctly rounded, so the scaled value is within 2^-53 of the truth (1e-10
   absolute; for 0 <= k <= 12 it IS the truth: a 24-bit significand times 5^k fits a double).  Its rounding to nearest, ties to
   even, is then the correctly rounded decimal printf prints -- unless an inexact scaled value lies within 1e-7 of a half:
   then, and for magnitudes beyond those powers of ten, std::to_chars (specified as printf's %.6g) decides.  Returns the end. */
static char *format_g6(char *o, float vf)
{
    static const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    uint32_t bits;
    memcpy(&bits, &vf, 4);
    if (vf != vf) { /* vsnprintf shows a NaN's sign bit */
        if (bits >> 31) *o++ = '-';
        memcpy(o, "nan", 3);
        return o + 3;
    }
    if (bits >> 31) *o++ = '-';
    const double a = std::fabs((double)vf);
    if (a == 0.0) { *o++ = '0'; return o; }
    if (a >= 1e-16 && a < 1e21) {
        /* floor(log10 a) guessed from the binary exponent (log10(2) = 1233 / 4096 to five digits: at most one off), settled on
           the scaled value below */
        const int e2 = (int)((bits >> 23) & 0xff) - 127;
        int X = (e2 * 1233) >> 12;
        for (int attempt = 0; attempt < 3; ++attempt) {
            const int k = 5 - X;
            if (k < -22 || k > 22) break;
            const double scaled = k >= 0 ? a * P10[k] : a / P10[-k];
            if (scaled < 1e5) { --X; continue; }
            if (scaled >= 1e6) { ++X; continue; }
            long n = (long)scaled;             /* scaled >= 1e5: truncation is the floor */
            const double fr = scaled - (double)n;
            const bool exact = k >= 0 && k <= 12; /* 24 bits x 5^k fit a double's 53: the product has no rounding at all */
            if (!exact && std::fabs(fr - 0.5) < 1e-7) break; /* too close to call here */
            if (fr > 0.5 || (fr == 0.5 && (n & 1))) ++n; /* to nearest, ties to even */
            if (n >= 1000000) { n = 100000; ++X; }
            char d[6]; /* the six digits, two at a time */
            {
                static const char PAIRS[201] = "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
                                               "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
                const unsigned u = (unsigned)n, hi = u / 10000, lo = u % 10000;
                memcpy(d, PAIRS + 2 * hi, 2); memcpy(d + 2, PAIRS + 2 * (lo / 100), 2); memcpy(d + 4, PAIRS + 2 * (lo % 100), 2);
            }
            int last = 5;
            while (last > 0 && d[last] == '0') --last; /* %g drops trailing zeros */
            if (X < -4 || X >= 6) {
                *o++ = d[0];
                if (last > 0) { *o++ = '.'; for (int i = 1; i <= last; ++i) *o++ = d[i]; }
                *o++ = 'e';
                int e = X;
                if (e < 0) { *o++ = '-'; e = -e; } else *o++ = '+';
                *o++ = (char)('0' + e / 10); *o++ = (char)('0' + e % 10);
            } else if (X >= 0) {
                for (int i = 0; i <= X; ++i) *o++ = d[i];
                if (last > X) { *o++ = '.'; for (int i = X + 1; i <= last; ++i) *o++ = d[i]; }
            } else {
                *o++ = '0'; *o++ = '.';
                for (int i = -1; i > X; --i) *o++ = '0';
                for (int i = 0; i <= last; ++i) *o++ = d[i];
            }
            return o;
        }
    }
    const std::to_chars_result r = std::to_chars(o, o + 15, a, std::chars_format::general, 6);
    return r.ptr;
}

/* pathFile as path_translation_alg.cpp:216-228 writes it: six `ofstream << float << " "` per waypoint, then std::endl.  The
   bytes are the reference's; the way there is not: std::endl's flush per line (one write() per waypoint: 0.1 s for the 26 k
   waypoints of a 1 M-point workpiece, against 0.07 ms of planning) becomes one write of the whole text. */
static int write_path_file_impl(const char *path, const float *wp6, size_t W)
{
    if (!path || (!wp6 && W)) return PPP_ERR_ARG;
    FILE *f = fopen(path, "wb");
    if (!f) {
        std::cerr << "Unable to open file: " << path << std::endl;
        return PPP_ERR_IO;
    }
    const size_t per = 6 * 16 + 1; /* "-1.23457e-38 " is 13 characters */
    /* long lists are formatted by several threads, each its own run of waypoints into its own text; written in order */
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t parts = W >= 16384 ? std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hw ? hw : 1, 8), W / 8192)) : 1;
    std::vector<std::unique_ptr<char[]>> text(parts);
    std::vector<size_t> len(parts, 0);
    try { /* (every allocation here, on this thread: nothing in a worker may throw) */
        for (size_t pi = 0; pi < parts; ++pi) text[pi].reset(new char[std::max<size_t>(W * (pi + 1) / parts - W * pi / parts, 1) * per]);
    } catch (...) { fclose(f); return PPP_ERR_IO; }
    auto run = [&](size_t pi) {
        const size_t w0 = W * pi / parts, w1 = W * (pi + 1) / parts;
        char *o = text[pi].get();
        for (size_t w = w0; w < w1; ++w) {
            for (int i = 0; i < 6; i++) {
                o = format_g6(o, wp6[6 * w + i]);
                *o++ = ' ';
            }
            *o++ = '\n';
        }
        len[pi] = (size_t)(o - text[pi].get());
    };
    run_parts(parts, run);
    bool ok = true;
    for (size_t pi = 0; pi < parts && ok; ++pi) ok = fwrite(text[pi].get(), 1, len[pi], f) == len[pi];
    return (fclose(f) == 0 && ok) ? PPP_OK : PPP_ERR_IO;
}


/* The boundary never lets a C++ exception (std::bad_alloc on a header that promises 10^12 points, ...) escape. */
int ppp_load_pcd(const char *path, float **xyz, size_t *n, float viewpoint[7])
{
    try { return load_pcd_impl(path, xyz, n, viewpoint); } catch (...) { if (xyz) *xyz = nullptr; if (n) *n = 0; return PPP_ERR_IO; }
}
int ppp_pcd_probe(const char *path, ppp_pcd_layout *layout)
{
    try { return probe_pcd_impl(path, layout); } catch (...) { return PPP_ERR_IO; }
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
