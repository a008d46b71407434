// pcreg_amd/csrc/io_formats.cpp -- the two on-disk formats the drivers read and write:
//   .pcd  point clouds      pcread / pcwrite, completeExperimentFast.m:12-13,30,403
//   .mat  descriptor caches load, completeExperimentFast.m:21-24,312-313  (MAT-file Level 5, v6/v7)
// Host code only (no HIP): a reader so that real data can be ingested, and the PCD writer for the
// final aligned surface.  MAT v7.3 files are HDF5 and not handled here.
#include "common.hpp"
#include <zlib.h>
#include <cerrno>
#include <cmath>
#include <sstream>
#include <vector>

namespace pcreg {
namespace {

// ------------------------------------------------------------------------------------ PCD
struct PcdField { std::string name; int size = 4; char type = 'F'; int count = 1; int offset = 0; };
struct PcdHeader {
    std::vector<PcdField> fields;
    long points = 0; int width = 0, height = 1; int point_size = 0;
    enum { ASCII, BINARY, COMPRESSED } data = ASCII;
    long data_offset = 0;
};

int pcd_parse_header(FILE* f, PcdHeader& h, const char* path) {
    char line[4096];
    std::vector<std::string> names; std::vector<int> sizes, counts; std::vector<char> types;
    bool have_points = false, have_data = false;
    while (fgets(line, sizeof line, f)) {
        std::istringstream is(line);
        std::string key; is >> key;
        if (key.empty() || key[0] == '#') continue;
        if (key == "FIELDS" || key == "COLUMNS") { std::string s; while (is >> s) names.push_back(s); }
        else if (key == "SIZE") { int v; while (is >> v) sizes.push_back(v); }
        else if (key == "TYPE") { std::string s; while (is >> s) types.push_back(s[0]); }
        else if (key == "COUNT") { int v; while (is >> v) counts.push_back(v); }
        else if (key == "WIDTH") is >> h.width;
        else if (key == "HEIGHT") is >> h.height;
        else if (key == "POINTS") { is >> h.points; have_points = true; }
        else if (key == "DATA") {
            std::string s; is >> s;
            if (s == "ascii") h.data = PcdHeader::ASCII;
            else if (s == "binary") h.data = PcdHeader::BINARY;
            else if (s == "binary_compressed") h.data = PcdHeader::COMPRESSED;
            else { set_error("%s: unknown PCD DATA kind '%s'", path, s.c_str()); return PCREG_E_ARG; }
            have_data = true;
            break;
        }
    }
    if (!have_data || names.empty()) { set_error("%s: not a PCD file (no FIELDS/DATA header)", path); return PCREG_E_ARG; }
    if (!have_points) h.points = (long)h.width * h.height;
    if (sizes.size() != names.size() || types.size() != names.size()) { set_error("%s: PCD header: FIELDS/SIZE/TYPE disagree", path); return PCREG_E_ARG; }
    if (counts.empty()) counts.assign(names.size(), 1);
    int off = 0;
    for (size_t i = 0; i < names.size(); ++i) {
        PcdField fd; fd.name = names[i]; fd.size = sizes[i]; fd.type = types[i]; fd.count = counts[i]; fd.offset = off;
        off += fd.size * fd.count;
        h.fields.push_back(fd);
    }
    h.point_size = off;
    h.data_offset = ftell(f);
    return PCREG_OK;
}

double pcd_value(const unsigned char* p, const PcdField& f) {
    switch (f.type) {
        case 'F': if (f.size == 4) { float v; memcpy(&v, p, 4); return v; } else { double v; memcpy(&v, p, 8); return v; }
        case 'U': if (f.size == 1) return *p; if (f.size == 2) { uint16_t v; memcpy(&v, p, 2); return v; } { uint32_t v; memcpy(&v, p, 4); return v; }
        default:  if (f.size == 1) return *(const int8_t*)p; if (f.size == 2) { int16_t v; memcpy(&v, p, 2); return v; } { int32_t v; memcpy(&v, p, 4); return v; }
    }
}

// LZF decompression (the format PCL's binary_compressed uses): control byte < 32: literal run of ctrl+1
// bytes; otherwise a back reference of length (ctrl >> 5) + 2 (7 -> +next byte) at distance
// ((ctrl & 31) << 8 | next) + 1.
bool lzf_decompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len) {
    size_t ip = 0, op = 0;
    while (ip < in_len) {
        unsigned ctrl = in[ip++];
        if (ctrl < 32) {
            size_t run = ctrl + 1;
            if (ip + run > in_len || op + run > out_len) return false;
            memcpy(out + op, in + ip, run); ip += run; op += run;
        } else {
            size_t len = ctrl >> 5;
            if (len == 7) { if (ip >= in_len) return false; len += in[ip++]; }
            if (ip >= in_len) return false;
            size_t dist = ((size_t)(ctrl & 31) << 8 | in[ip++]) + 1;
            len += 2;
            if (dist > op || op + len > out_len) return false;
            for (size_t k = 0; k < len; ++k, ++op) out[op] = out[op - dist];
        }
    }
    return op == out_len;
}

int pcd_load(const char* path, PcdHeader& h, std::vector<unsigned char>& aos) {
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("%s: %s", path, strerror(errno)); return PCREG_E_ARG; }
    int rc = pcd_parse_header(f, h, path);
    if (rc) { fclose(f); return rc; }
    const size_t n = (size_t)h.points, ps = (size_t)h.point_size;
    aos.assign(n * ps, 0);
    if (h.data == PcdHeader::BINARY) {
        if (fread(aos.data(), 1, n * ps, f) != n * ps) { fclose(f); set_error("%s: truncated binary PCD", path); return PCREG_E_ARG; }
    } else if (h.data == PcdHeader::COMPRESSED) {
        uint32_t sz[2];
        if (fread(sz, 4, 2, f) != 2 || sz[1] != n * ps) { fclose(f); set_error("%s: bad binary_compressed sizes", path); return PCREG_E_ARG; }
        std::vector<unsigned char> in(sz[0]), soa(sz[1]);
        if (fread(in.data(), 1, sz[0], f) != sz[0] || !lzf_decompress(in.data(), sz[0], soa.data(), sz[1])) {
            fclose(f); set_error("%s: corrupt binary_compressed PCD", path); return PCREG_E_ARG;
        }
        size_t base = 0;                       // decompressed layout: field by field
        for (const PcdField& fd : h.fields) {
            const size_t w = (size_t)fd.size * fd.count;
            for (size_t i = 0; i < n; ++i) memcpy(&aos[i * ps + fd.offset], &soa[base + i * w], w);
            base += w * n;
        }
    } else {
        char tok[128];
        for (size_t i = 0; i < n; ++i)
            for (const PcdField& fd : h.fields)
                for (int c = 0; c < fd.count; ++c) {
                    if (fscanf(f, "%127s", tok) != 1) { fclose(f); set_error("%s: truncated ascii PCD (point %zu)", path, i); return PCREG_E_ARG; }
                    unsigned char* p = &aos[i * ps + fd.offset + (size_t)c * fd.size];
                    if (fd.type == 'F') { if (fd.size == 4) { float v = strtof(tok, nullptr); memcpy(p, &v, 4); } else { double v = strtod(tok, nullptr); memcpy(p, &v, 8); } }
                    else if (fd.type == 'U') { unsigned long v = strtoul(tok, nullptr, 10); memcpy(p, &v, fd.size); }   // little-endian host
                    else { long v = strtol(tok, nullptr, 10); memcpy(p, &v, fd.size); }
                }
    }
    fclose(f);
    return PCREG_OK;
}

// ------------------------------------------------------------------------------------ MAT v5
enum { miINT8 = 1, miUINT8, miINT16, miUINT16, miINT32, miUINT32, miSINGLE, miDOUBLE = 9, miINT64 = 12, miUINT64, miMATRIX, miCOMPRESSED };
struct MatVar { std::string name; int cls = 0; std::vector<int> dims; int data_type = 0; const unsigned char* data = nullptr; size_t data_bytes = 0; };

bool mat_read_tag(const unsigned char* p, size_t avail, uint32_t& type, uint32_t& bytes, size_t& hdr) {
    if (avail < 8) return false;
    uint32_t w0; memcpy(&w0, p, 4);
    if (w0 >> 16) { type = w0 & 0xFFFF; bytes = w0 >> 16; hdr = 4; return bytes <= 4; }     // small element
    type = w0; memcpy(&bytes, p + 4, 4); hdr = 8;
    return true;
}
size_t pad8(size_t n) { return (n + 7) & ~(size_t)7; }

// parses one miMATRIX body (numeric, real); returns false for classes this reader does not handle
bool mat_parse_matrix(const unsigned char* p, size_t n, MatVar& v) {
    size_t pos = 0; uint32_t t, b; size_t h;
    if (!mat_read_tag(p, n, t, b, h) || t != miUINT32 || b != 8) return false;              // array flags
    uint32_t flags; memcpy(&flags, p + h, 4);
    v.cls = flags & 0xFF;
    const bool is_complex = flags & 0x0800;
    pos = h + 8;
    if (!mat_read_tag(p + pos, n - pos, t, b, h) || t != miINT32) return false;               // dimensions
    v.dims.resize(b / 4);
    memcpy(v.dims.data(), p + pos + h, b);
    pos += h == 4 ? 8 : h + pad8(b);
    if (!mat_read_tag(p + pos, n - pos, t, b, h) || t != miINT8) return false;                // name
    v.name.assign((const char*)p + pos + h, b);
    pos += h == 4 ? 8 : h + pad8(b);
    if (v.cls < 6 || v.cls > 15 || is_complex) return true;                                  // cell/struct/char/sparse/complex: listed, not readable
    if (!mat_read_tag(p + pos, n - pos, t, b, h)) return false;                               // real part
    v.data_type = (int)t; v.data = p + pos + h; v.data_bytes = b;
    return pos + h + b <= n;
}

int mat_load(const char* path, std::vector<std::vector<unsigned char>>& storage, std::vector<MatVar>& vars) {
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("%s: %s", path, strerror(errno)); return PCREG_E_ARG; }
    std::vector<unsigned char> file;
    unsigned char buf[1 << 16]; size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + got);
    fclose(f);
    if (file.size() < 128 || memcmp(file.data(), "MATLAB 5.0 MAT-file", 19) != 0) {
        set_error("%s: not a Level-5 MAT-file (v7.3 files are HDF5 and are not supported)", path); return PCREG_E_ARG;
    }
    if (file[126] != 'I' || file[127] != 'M') { set_error("%s: big-endian MAT-files are not supported", path); return PCREG_E_ARG; }
    storage.push_back(std::move(file));
    const std::vector<unsigned char>& fl = storage.front();
    size_t pos = 128;
    while (pos + 8 <= fl.size()) {
        uint32_t t, b; size_t h;
        if (!mat_read_tag(fl.data() + pos, fl.size() - pos, t, b, h)) break;
        const unsigned char* body = fl.data() + pos + h;
        if (pos + h + b > fl.size()) { set_error("%s: truncated MAT-file", path); return PCREG_E_ARG; }
        if (t == miCOMPRESSED) {
            // inflate with a growing buffer (the element holds exactly one miMATRIX)
            std::vector<unsigned char> out(std::max<size_t>(4 * (size_t)b, 1024));
            z_stream zs; memset(&zs, 0, sizeof zs);
            if (inflateInit(&zs) != Z_OK) { set_error("zlib init failed"); return PCREG_E_ARG; }
            zs.next_in = const_cast<unsigned char*>(body); zs.avail_in = b;
            size_t produced = 0; int zr;
            do {
                if (produced == out.size()) out.resize(out.size() * 2);
                zs.next_out = out.data() + produced; zs.avail_out = (uInt)std::min<size_t>(out.size() - produced, 1u << 30);
                zr = inflate(&zs, Z_NO_FLUSH);
                produced = zs.total_out;
            } while (zr == Z_OK);
            inflateEnd(&zs);
            if (zr != Z_STREAM_END) { set_error("%s: corrupt compressed element", path); return PCREG_E_ARG; }
            out.resize(produced);
            storage.push_back(std::move(out));
            const std::vector<unsigned char>& o = storage.back();
            uint32_t t2, b2; size_t h2;
            if (mat_read_tag(o.data(), o.size(), t2, b2, h2) && t2 == miMATRIX && h2 + b2 <= o.size()) {
                MatVar v; if (mat_parse_matrix(o.data() + h2, b2, v)) vars.push_back(v);
            }
        } else if (t == miMATRIX) {
            MatVar v; if (mat_parse_matrix(body, b, v)) vars.push_back(v);
        }
        pos += t == miCOMPRESSED ? h + b : (h == 4 ? 8 : h + pad8(b));      // compressed elements are not padded
    }
    return PCREG_OK;
}

double mat_elem(const unsigned char* p, int type, size_t i) {
    switch (type) {
        case miDOUBLE: { double v; memcpy(&v, p + 8 * i, 8); return v; }
        case miSINGLE: { float v; memcpy(&v, p + 4 * i, 4); return v; }
        case miINT8: return ((const int8_t*)p)[i];   case miUINT8: return p[i];
        case miINT16: { int16_t v; memcpy(&v, p + 2 * i, 2); return v; }  case miUINT16: { uint16_t v; memcpy(&v, p + 2 * i, 2); return v; }
        case miINT32: { int32_t v; memcpy(&v, p + 4 * i, 4); return v; }  case miUINT32: { uint32_t v; memcpy(&v, p + 4 * i, 4); return v; }
        case miINT64: { int64_t v; memcpy(&v, p + 8 * i, 8); return (double)v; } case miUINT64: { uint64_t v; memcpy(&v, p + 8 * i, 8); return (double)v; }
        default: return NAN;
    }
}
size_t mat_type_size(int type) {
    switch (type) { case miDOUBLE: case miINT64: case miUINT64: return 8; case miSINGLE: case miINT32: case miUINT32: return 4;
                    case miINT16: case miUINT16: return 2; default: return 1; }
}

}  // namespace
}  // namespace pcreg

using namespace pcreg;

extern "C" {

int pcreg_pcd_info(const char* path, int* n_points, int* has_rgb) {
    PCREG_ARG(path && n_points);
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("%s: %s", path, strerror(errno)); return PCREG_E_ARG; }
    PcdHeader h; int rc = pcd_parse_header(f, h, path);
    fclose(f);
    if (rc) return rc;
    *n_points = (int)h.points;
    if (has_rgb) { *has_rgb = 0; for (const PcdField& fd : h.fields) if (fd.name == "rgb" || fd.name == "rgba") *has_rgb = 1; }
    return PCREG_OK;
}

int pcreg_pcd_read(const char* path, float* xyz, int ld, uint32_t* rgb, int n) {
    PCREG_ARG(path && xyz && n >= 0 && ld >= n);
    PcdHeader h; std::vector<unsigned char> aos;
    int rc = pcd_load(path, h, aos);
    if (rc) return rc;
    if ((long)n != h.points) { set_error("%s holds %ld points, the buffer %d", path, h.points, n); return PCREG_E_ARG; }
    const PcdField* fx = nullptr; const PcdField* fy = nullptr; const PcdField* fz = nullptr; const PcdField* fc = nullptr;
    for (const PcdField& fd : h.fields) {
        if (fd.name == "x") fx = &fd; else if (fd.name == "y") fy = &fd; else if (fd.name == "z") fz = &fd;
        else if (fd.name == "rgb" || fd.name == "rgba") fc = &fd;
    }
    if (!fx || !fy || !fz) { set_error("%s: no x/y/z fields", path); return PCREG_E_ARG; }
    for (int i = 0; i < n; ++i) {
        const unsigned char* p = &aos[(size_t)i * h.point_size];
        xyz[i] = (float)pcd_value(p + fx->offset, *fx); xyz[i + (size_t)ld] = (float)pcd_value(p + fy->offset, *fy);
        xyz[i + 2 * (size_t)ld] = (float)pcd_value(p + fz->offset, *fz);
        if (rgb) { uint32_t c = 0; if (fc) memcpy(&c, p + fc->offset, 4); rgb[i] = c; }   // packed 0x00RRGGBB, whatever TYPE says
    }
    return PCREG_OK;
}

int pcreg_pcd_write(const char* path, const float* xyz, int n, int ld, const uint32_t* rgb, int binary) {
    PCREG_ARG(path && xyz && n >= 0 && ld >= n);
    FILE* f = fopen(path, "wb");
    if (!f) { set_error("%s: %s", path, strerror(errno)); return PCREG_E_ARG; }
    fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n");
    if (rgb) fprintf(f, "FIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\n");
    else     fprintf(f, "FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n");
    fprintf(f, "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n", n, n, binary ? "binary" : "ascii");
    for (int i = 0; i < n; ++i) {
        const float p[3] = {xyz[i], xyz[i + (size_t)ld], xyz[i + 2 * (size_t)ld]};
        if (binary) { fwrite(p, 4, 3, f); if (rgb) fwrite(&rgb[i], 4, 1, f); }
        else if (rgb) fprintf(f, "%.9g %.9g %.9g %u\n", p[0], p[1], p[2], rgb[i]);
        else fprintf(f, "%.9g %.9g %.9g\n", p[0], p[1], p[2]);
    }
    const bool ok = fclose(f) == 0;
    if (!ok) { set_error("%s: write failed", path); return PCREG_E_ARG; }
    return PCREG_OK;
}

// Variable `name` (NULL or "" = the first numeric array) of a Level-5 MAT-file: rows x cols (further
// dimensions folded into cols).  With out == NULL only the shape is returned.
int pcreg_mat_read_double(const char* path, const char* name, double* out, int* rows, int* cols) {
    PCREG_ARG(path && rows && cols);
    std::vector<std::vector<unsigned char>> storage; storage.reserve(64);
    std::vector<MatVar> vars;
    int rc = mat_load(path, storage, vars);
    if (rc) return rc;
    const MatVar* v = nullptr;
    for (const MatVar& c : vars) {
        if (name && *name) { if (c.name == name) { v = &c; break; } }
        else if (c.data) { v = &c; break; }
    }
    if (!v) { set_error("%s: no variable '%s'", path, name && *name ? name : "<first numeric>"); return PCREG_E_ARG; }
    if (!v->data) { set_error("%s: variable '%s' is not a real numeric array", path, v->name.c_str()); return PCREG_E_ARG; }
    size_t r = v->dims.empty() ? 0 : (size_t)v->dims[0], c = 1;
    for (size_t k = 1; k < v->dims.size(); ++k) c *= (size_t)v->dims[k];
    if (v->dims.size() < 2) c = v->dims.empty() ? 0 : 1;
    if (r * c * mat_type_size(v->data_type) > v->data_bytes) { set_error("%s: variable '%s' is truncated", path, v->name.c_str()); return PCREG_E_ARG; }
    *rows = (int)r; *cols = (int)c;
    if (out) for (size_t i = 0; i < r * c; ++i) out[i] = mat_elem(v->data, v->data_type, i);
    return PCREG_OK;
}

}  // extern "C"
