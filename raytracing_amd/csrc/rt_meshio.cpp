// Mesh file reader of librt_amd.so (CPU only): Wavefront OBJ and PLY (ascii, binary_little_endian)
// into the three arrays one entry of the reference's Meshes initializer list holds
// (src/scene.cpp:8-30: vertex positions, vertex normals, indexed triangles; one index per vertex,
// shared by position and normal).  The reference has no loader (its meshes are the literals of
// src/main.cpp:25-121); this exists because every workload beyond the built-in scene needs geometry
// from somewhere.
//
//   * polygons are split into a fan around their first vertex;
//   * OBJ position/normal index pairs are merged into single vertices (one vertex per distinct pair);
//   * a file without normals gets area-weighted vertex normals (sum of the face cross products of the
//     faces around a vertex, normalised), so flat regions shade flat and curved ones smooth.
#include "rt_host.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

struct rt_mesh_file {
    std::vector<float> positions, normals;
    std::vector<uint32_t> tris;
};

namespace {

thread_local std::string g_meshError;

int meshFail(const std::string& msg) {
    g_meshError = msg;
    return -1;
}

bool readWholeFile(const char* path, std::vector<char>& out) {
    std::FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    bool const ok = !std::ferror(f);
    std::fclose(f);
    return ok;
}

void computeNormals(rt_mesh_file& m) {
    size_t const nv = m.positions.size() / 3;
    std::vector<double> acc(3 * nv, 0.0);
    for (size_t t = 0; t + 2 < m.tris.size(); t += 3) {
        uint32_t const i = m.tris[t], j = m.tris[t + 1], k = m.tris[t + 2];
        const float *a = &m.positions[3 * i], *b = &m.positions[3 * j], *c = &m.positions[3 * k];
        double const e1[3] = {double(b[0]) - a[0], double(b[1]) - a[1], double(b[2]) - a[2]};
        double const e2[3] = {double(c[0]) - a[0], double(c[1]) - a[1], double(c[2]) - a[2]};
        double const n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        for (uint32_t v : {i, j, k})
            for (int x = 0; x < 3; ++x) acc[3 * v + x] += n[x];
    }
    m.normals.assign(3 * nv, 0.0f);
    for (size_t v = 0; v < nv; ++v) {
        double const len = std::sqrt(acc[3 * v] * acc[3 * v] + acc[3 * v + 1] * acc[3 * v + 1] + acc[3 * v + 2] * acc[3 * v + 2]);
        if (len > 0.0)
            for (int x = 0; x < 3; ++x) m.normals[3 * v + x] = static_cast<float>(acc[3 * v + x] / len);
        else m.normals[3 * v + 1] = 1.0f;        // isolated or degenerate vertex: any unit vector
    }
}

// ---- OBJ ----------------------------------------------------------------------------------------

int loadObj(const std::vector<char>& data, rt_mesh_file& m) {
    std::vector<float> pos, nrm;
    std::map<std::pair<long, long>, uint32_t> merged;         // (position index, normal index) -> vertex
    bool anyNormalIndex = false, anyMissingNormalIndex = false;
    size_t lineNo = 0;
    const char* p = data.data();
    const char* const end = p + data.size();
    std::vector<uint32_t> face;
    while (p < end) {
        const char* eol = static_cast<const char*>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        if (!eol) eol = end;
        std::string line(p, eol);
        p = eol < end ? eol + 1 : end;
        ++lineNo;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const char* s = line.c_str();
        while (*s == ' ' || *s == '\t') ++s;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            float v[3];
            if (std::sscanf(s + 1, "%f %f %f", v, v + 1, v + 2) != 3) return meshFail("OBJ line " + std::to_string(lineNo) + ": bad vertex");
            pos.insert(pos.end(), v, v + 3);
        }
        else if (s[0] == 'v' && s[1] == 'n' && (s[2] == ' ' || s[2] == '\t')) {
            float v[3];
            if (std::sscanf(s + 2, "%f %f %f", v, v + 1, v + 2) != 3) return meshFail("OBJ line " + std::to_string(lineNo) + ": bad normal");
            nrm.insert(nrm.end(), v, v + 3);
        }
        else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            face.clear();
            const char* q = s + 1;
            for (;;) {
                while (*q == ' ' || *q == '\t') ++q;
                if (!*q) break;
                char* e = nullptr;
                long vi = std::strtol(q, &e, 10), ni = 0;
                if (e == q) return meshFail("OBJ line " + std::to_string(lineNo) + ": bad face entry");
                q = e;
                bool hasNormal = false;
                if (*q == '/') {
                    ++q;
                    if (*q != '/') (void)std::strtol(q, &e, 10), q = e;      // texture index, unused
                    if (*q == '/') {
                        ++q;
                        ni = std::strtol(q, &e, 10);
                        hasNormal = e != q;
                        q = e;
                    }
                }
                long const np = static_cast<long>(pos.size() / 3), nn = static_cast<long>(nrm.size() / 3);
                if (vi < 0) vi = np + vi + 1;                   // negative indices count from the end
                if (hasNormal && ni < 0) ni = nn + ni + 1;
                if (vi < 1 || vi > np) return meshFail("OBJ line " + std::to_string(lineNo) + ": vertex index out of range");
                if (hasNormal && (ni < 1 || ni > nn)) return meshFail("OBJ line " + std::to_string(lineNo) + ": normal index out of range");
                if (hasNormal) anyNormalIndex = true; else anyMissingNormalIndex = true;
                auto const key = std::make_pair(vi - 1, hasNormal ? ni - 1 : -1L);
                auto it = merged.find(key);
                if (it == merged.end()) {
                    uint32_t const id = static_cast<uint32_t>(m.positions.size() / 3);
                    m.positions.insert(m.positions.end(), &pos[3 * key.first], &pos[3 * key.first] + 3);
                    if (hasNormal) m.normals.insert(m.normals.end(), &nrm[3 * key.second], &nrm[3 * key.second] + 3);
                    else m.normals.insert(m.normals.end(), {0.0f, 0.0f, 0.0f});
                    it = merged.emplace(key, id).first;
                }
                face.push_back(it->second);
            }
            if (face.size() < 3) return meshFail("OBJ line " + std::to_string(lineNo) + ": face with fewer than 3 vertices");
            for (size_t k = 1; k + 1 < face.size(); ++k) {
                m.tris.push_back(face[0]); m.tris.push_back(face[k]); m.tris.push_back(face[k + 1]);
            }
        }
    }
    if (anyNormalIndex && anyMissingNormalIndex) return meshFail("OBJ: some face entries have normals and some do not");
    if (!anyNormalIndex) computeNormals(m);
    return 0;
}

// ---- PLY ----------------------------------------------------------------------------------------

struct PlyProperty {
    std::string name;
    int type = 0, countType = 0, itemType = 0;      // scalar: type; list: countType + itemType
    bool isList = false;
};
struct PlyElement {
    std::string name;
    size_t count = 0;
    std::vector<PlyProperty> props;
};

// type codes: 1 int8, 2 uint8, 3 int16, 4 uint16, 5 int32, 6 uint32, 7 float32, 8 float64
int plyType(const std::string& t) {
    static const std::pair<const char*, int> names[] = {
        {"char", 1}, {"int8", 1}, {"uchar", 2}, {"uint8", 2}, {"short", 3}, {"int16", 3}, {"ushort", 4}, {"uint16", 4},
        {"int", 5}, {"int32", 5}, {"uint", 6}, {"uint32", 6}, {"float", 7}, {"float32", 7}, {"double", 8}, {"float64", 8}};
    for (auto const& n : names) if (t == n.first) return n.second;
    return 0;
}
size_t plySize(int type) { static const size_t s[] = {0, 1, 1, 2, 2, 4, 4, 4, 8}; return s[type]; }

double plyBinary(const char*& p, const char* end, int type, bool& ok) {
    size_t const n = plySize(type);
    if (static_cast<size_t>(end - p) < n) { ok = false; return 0.0; }
    double v = 0.0;
    switch (type) {
        case 1: { int8_t x; std::memcpy(&x, p, 1); v = x; break; }
        case 2: { uint8_t x; std::memcpy(&x, p, 1); v = x; break; }
        case 3: { int16_t x; std::memcpy(&x, p, 2); v = x; break; }
        case 4: { uint16_t x; std::memcpy(&x, p, 2); v = x; break; }
        case 5: { int32_t x; std::memcpy(&x, p, 4); v = x; break; }
        case 6: { uint32_t x; std::memcpy(&x, p, 4); v = x; break; }
        case 7: { float x; std::memcpy(&x, p, 4); v = x; break; }
        case 8: { double x; std::memcpy(&x, p, 8); v = x; break; }
        default: ok = false;
    }
    p += n;
    return v;
}

double plyAscii(const char*& p, const char* end, bool& ok) {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) ++p;
    if (p >= end) { ok = false; return 0.0; }
    char tmp[64];
    size_t n = 0;
    while (p < end && n + 1 < sizeof tmp && !(*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) tmp[n++] = *p++;
    tmp[n] = 0;
    char* e = nullptr;
    double const v = std::strtod(tmp, &e);
    if (e == tmp) ok = false;
    return v;
}

int loadPly(const std::vector<char>& data, rt_mesh_file& m) {
    const char* p = data.data();
    const char* const end = p + data.size();
    std::vector<PlyElement> elements;
    int format = 0;                      // 1 ascii, 2 binary little endian
    bool headerDone = false;
    while (p < end && !headerDone) {
        const char* eol = static_cast<const char*>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        if (!eol) return meshFail("PLY: header not terminated");
        std::string line(p, eol);
        p = eol + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        char a[64] = {0}, b[64] = {0}, c[64] = {0}, d[64] = {0}, e[64] = {0};
        int const n = std::sscanf(line.c_str(), "%63s %63s %63s %63s %63s", a, b, c, d, e);
        std::string const key = a;
        if (key == "format") {
            if (std::string(b) == "ascii") format = 1;
            else if (std::string(b) == "binary_little_endian") format = 2;
            else return meshFail("PLY: unsupported format " + std::string(b));
        }
        else if (key == "element" && n >= 3) {
            PlyElement el;
            el.name = b;
            el.count = static_cast<size_t>(std::strtoull(c, nullptr, 10));
            elements.push_back(el);
        }
        else if (key == "property" && !elements.empty()) {
            PlyProperty pr;
            if (std::string(b) == "list" && n >= 5) {
                pr.isList = true; pr.countType = plyType(c); pr.itemType = plyType(d); pr.name = e;
                if (!pr.countType || !pr.itemType) return meshFail("PLY: unknown list type");
            }
            else if (n >= 3) {
                pr.type = plyType(b); pr.name = c;
                if (!pr.type) return meshFail("PLY: unknown property type " + std::string(b));
            }
            else return meshFail("PLY: bad property line");
            elements.back().props.push_back(pr);
        }
        else if (key == "end_header") headerDone = true;
    }
    if (!headerDone || !format) return meshFail("PLY: incomplete header");
    bool haveNormals = false;
    for (auto const& el : elements) {
        int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, ilist = -1;
        for (size_t k = 0; k < el.props.size(); ++k) {
            auto const& nm = el.props[k].name;
            if (nm == "x") ix = int(k); else if (nm == "y") iy = int(k); else if (nm == "z") iz = int(k);
            else if (nm == "nx") inx = int(k); else if (nm == "ny") iny = int(k); else if (nm == "nz") inz = int(k);
            else if (el.props[k].isList && (nm == "vertex_indices" || nm == "vertex_index")) ilist = int(k);
        }
        bool const isVertex = el.name == "vertex", isFace = el.name == "face";
        if (isVertex && (ix < 0 || iy < 0 || iz < 0)) return meshFail("PLY: vertex element without x, y, z");
        if (isFace && ilist < 0) return meshFail("PLY: face element without vertex_indices");
        if (isVertex) haveNormals = inx >= 0 && iny >= 0 && inz >= 0;
        // An element with rows but no properties would consume no data: its row loop could not end.
        if (el.count > 0 && el.props.empty()) return meshFail("PLY: element " + el.name + " has rows but no properties");
        std::vector<double> scalars(el.props.size());
        std::vector<uint32_t> face;
        for (size_t row = 0; row < el.count; ++row) {
            bool ok = true;
            for (size_t k = 0; k < el.props.size() && ok; ++k) {
                auto const& pr = el.props[k];
                if (!pr.isList) { scalars[k] = format == 1 ? plyAscii(p, end, ok) : plyBinary(p, end, pr.type, ok); continue; }
                double const cnt = format == 1 ? plyAscii(p, end, ok) : plyBinary(p, end, pr.countType, ok);
                if (!ok || !(cnt >= 0 && cnt <= 1e6)) { ok = false; break; }      // negated range test: NaN fails it
                bool const wanted = isFace && int(k) == ilist;
                if (wanted) face.clear();
                for (size_t i = 0; i < static_cast<size_t>(cnt) && ok; ++i) {
                    double const v = format == 1 ? plyAscii(p, end, ok) : plyBinary(p, end, pr.itemType, ok);
                    if (wanted) {
                        if (!(v >= 0 && v <= 4294967294.0)) { ok = false; break; }
                        face.push_back(static_cast<uint32_t>(v));
                    }
                }
            }
            if (!ok) return meshFail("PLY: truncated or malformed data in element " + el.name);
            if (isVertex) {
                m.positions.insert(m.positions.end(), {float(scalars[ix]), float(scalars[iy]), float(scalars[iz])});
                if (haveNormals) m.normals.insert(m.normals.end(), {float(scalars[inx]), float(scalars[iny]), float(scalars[inz])});
            }
            else if (isFace) {
                if (face.size() < 3) return meshFail("PLY: face with fewer than 3 vertices");
                for (size_t k = 1; k + 1 < face.size(); ++k) {
                    m.tris.push_back(face[0]); m.tris.push_back(face[k]); m.tris.push_back(face[k + 1]);
                }
            }
        }
    }
    size_t const nv = m.positions.size() / 3;
    for (uint32_t i : m.tris)
        if (i >= nv) return meshFail("PLY: vertex index out of range");
    if (!haveNormals) computeNormals(m);
    return 0;
}

}  // namespace

extern "C" {

const char* rt_mesh_last_error(void) { return g_meshError.c_str(); }

int rt_mesh_load(const char* path, rt_mesh_file** out) {
    if (!path || !out) return meshFail("rt_mesh_load: null argument");
    std::vector<char> data;
    bool readOk = false;
    try { readOk = readWholeFile(path, data); }
    catch (std::bad_alloc const&) { return meshFail("rt_mesh_load: out of host memory"); }
    if (!readOk) return meshFail(std::string("rt_mesh_load: cannot read ") + path);
    // No exception may leave through the C ABI: a mesh too large for host memory is an error return.
    try {
        std::unique_ptr<rt_mesh_file> m(new rt_mesh_file);
        bool const ply = data.size() >= 3 && std::memcmp(data.data(), "ply", 3) == 0;
        int const rc = ply ? loadPly(data, *m) : loadObj(data, *m);
        if (rc != 0) return rc;
        if (m->positions.size() / 3 > 0xFFFFFFFEull || m->tris.size() / 3 > 0xFFFFFFFEull)
            return meshFail("rt_mesh_load: mesh too large for 32-bit indices");
        *out = m.release();
        return 0;
    }
    catch (std::bad_alloc const&) { return meshFail("rt_mesh_load: out of host memory"); }
    catch (...) { return meshFail("rt_mesh_load: unexpected exception"); }
}

void rt_mesh_counts(const rt_mesh_file* m, uint32_t* vertex_count, uint32_t* tri_count) {
    if (vertex_count) *vertex_count = m ? static_cast<uint32_t>(m->positions.size() / 3) : 0;
    if (tri_count) *tri_count = m ? static_cast<uint32_t>(m->tris.size() / 3) : 0;
}

void rt_mesh_view(const rt_mesh_file* m, const float** positions, const float** normals, const uint32_t** tris) {
    if (positions) *positions = m ? m->positions.data() : nullptr;
    if (normals) *normals = m ? m->normals.data() : nullptr;
    if (tris) *tris = m ? m->tris.data() : nullptr;
}

void rt_mesh_destroy(rt_mesh_file* m) { delete m; }

}  // extern "C"
