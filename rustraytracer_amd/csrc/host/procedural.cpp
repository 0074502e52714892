// procedural.cpp -- "P-N" stand-in mesh and the tobj-style OBJ reader.
//
// data/dragon/dragon.obj and data/statue.obj are listed in the reference's
// .MISSING_LARGE_BLOBS, so BASELINE's mesh configs are driven by
// a deterministic procedural surface of the same triangle count unless the user
// supplies the OBJ (parse_obj below follows src/parser.rs:8-87 / tobj 2.0.2
// `load_obj(path, true)`: first model, fan triangulation, single index, f32 data).
#include <algorithm>
#include <array>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <unordered_map>

#include "rr_host.hpp"

namespace rr {

namespace {

inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// lattice hash -> [0,1)
inline double lattice(int64_t x, int64_t y, int64_t z, uint64_t seed) {
    uint64_t h = mix64(seed + 0x9E3779B97F4A7C15ull * (uint64_t)x);
    h = mix64(h + 0xD1B54A32D192ED03ull * (uint64_t)y);
    h = mix64(h + 0x8CB92BA72F3D8DD7ull * (uint64_t)z);
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}
inline double smooth(double t) { return t * t * (3.0 - 2.0 * t); }
double value_noise(double x, double y, double z, uint64_t seed) {
    double fx = std::floor(x), fy = std::floor(y), fz = std::floor(z);
    int64_t ix = (int64_t)fx, iy = (int64_t)fy, iz = (int64_t)fz;
    double tx = smooth(x - fx), ty = smooth(y - fy), tz = smooth(z - fz);
    double c[2][2][2];
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++)
            for (int d = 0; d < 2; d++) c[a][b][d] = lattice(ix + a, iy + b, iz + d, seed);
    double x00 = c[0][0][0] + (c[1][0][0] - c[0][0][0]) * tx, x10 = c[0][1][0] + (c[1][1][0] - c[0][1][0]) * tx;
    double x01 = c[0][0][1] + (c[1][0][1] - c[0][0][1]) * tx, x11 = c[0][1][1] + (c[1][1][1] - c[0][1][1]) * tx;
    double y0 = x00 + (x10 - x00) * ty, y1 = x01 + (x11 - x01) * ty;
    return y0 + (y1 - y0) * tz;
}

struct Key {
    std::array<uint32_t, 6> k;
    bool operator==(const Key& o) const { return k == o.k; }
};
struct KeyHash {
    size_t operator()(const Key& a) const {
        uint64_t h = 0x1234;
        for (uint32_t v : a.k) h = mix64(h ^ v);
        return (size_t)h;
    }
};

}  // namespace

Mesh procedural_mesh(uint64_t n_faces, const Mat4& trans) {
    Mesh out;
    if (n_faces == 0) return out;
    uint32_t f = 1;
    while ((uint64_t)20 * f * f < n_faces) f++;
    // icosahedron
    const double t = (1.0 + std::sqrt(5.0)) / 2.0;
    const double iv[12][3] = {{-1, t, 0}, {1, t, 0}, {-1, -t, 0}, {1, -t, 0}, {0, -1, t}, {0, 1, t},
                              {0, -1, -t}, {0, 1, -t}, {t, 0, -1}, {t, 0, 1}, {-t, 0, -1}, {-t, 0, 1}};
    const int faces[20][3] = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4},
                              {11, 10, 2}, {10, 7, 6}, {7, 1, 8}, {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8},
                              {3, 8, 9}, {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};
    std::unordered_map<Key, uint32_t, KeyHash> ids;
    std::vector<double> dirs;  // unit directions
    auto vertex = [&](int a, int b, int c, uint32_t wa, uint32_t wb, uint32_t wc) -> uint32_t {
        std::array<std::pair<uint32_t, uint32_t>, 3> e = {{{(uint32_t)a, wa}, {(uint32_t)b, wb}, {(uint32_t)c, wc}}};
        std::sort(e.begin(), e.end());
        Key key{};
        int n = 0;
        for (auto& p : e)
            if (p.second > 0) {
                key.k[2 * n] = p.first + 1;
                key.k[2 * n + 1] = p.second;
                n++;
            }
        auto it = ids.find(key);
        if (it != ids.end()) return it->second;
        double x = 0, y = 0, z = 0;
        for (auto& p : e)
            if (p.second > 0) {
                x += iv[p.first][0] * (double)p.second;
                y += iv[p.first][1] * (double)p.second;
                z += iv[p.first][2] * (double)p.second;
            }
        double l = std::sqrt(x * x + y * y + z * z);
        uint32_t id = (uint32_t)(dirs.size() / 3);
        dirs.push_back(x / l);
        dirs.push_back(y / l);
        dirs.push_back(z / l);
        ids.emplace(key, id);
        return id;
    };
    std::vector<uint32_t> ind;
    ind.reserve((size_t)60 * f * f);
    for (int fi = 0; fi < 20; fi++) {
        int a = faces[fi][0], b = faces[fi][1], c = faces[fi][2];
        for (uint32_t i = 0; i < f; i++)
            for (uint32_t j = 0; j + i < f; j++) {
                // grid point (i,j): weights (f-i-j, i, j)
                uint32_t v00 = vertex(a, b, c, f - i - j, i, j);
                uint32_t v10 = vertex(a, b, c, f - i - j - 1, i + 1, j);
                uint32_t v01 = vertex(a, b, c, f - i - j - 1, i, j + 1);
                ind.push_back(v00); ind.push_back(v10); ind.push_back(v01);
                if (i + j + 1 < f) {
                    uint32_t v11 = vertex(a, b, c, f - i - j - 2, i + 1, j + 1);
                    ind.push_back(v10); ind.push_back(v11); ind.push_back(v01);
                }
            }
    }
    ind.resize((size_t)n_faces * 3);  // trim to exactly n_faces
    size_t nv = dirs.size() / 3;
    // displacement: 5 octaves of value noise, seed 1234
    std::vector<double> pos(nv * 3);
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (size_t v = 0; v < nv; v++) {
        double dx = dirs[3 * v], dy = dirs[3 * v + 1], dz = dirs[3 * v + 2];
        double r = 1.0, amp = 0.35, freq = 1.7;
        for (int o = 0; o < 5; o++) {
            r += amp * (value_noise(dx * freq + 11.5, dy * freq + 7.25, dz * freq + 3.75, 1234 + o) - 0.5);
            amp *= 0.5;
            freq *= 2.1;
        }
        pos[3 * v] = dx * r;
        pos[3 * v + 1] = dy * r;
        pos[3 * v + 2] = dz * r;
        for (int a = 0; a < 3; a++) {
            mn[a] = std::min(mn[a], pos[3 * v + a]);
            mx[a] = std::max(mx[a], pos[3 * v + a]);
        }
    }
    const double ext[3] = {1.0, 0.7, 0.45};
    std::vector<double> pf(nv * 3);
    for (size_t v = 0; v < nv; v++)
        for (int a = 0; a < 3; a++) {
            double q = ((pos[3 * v + a] - mn[a]) / (mx[a] - mn[a]) - 0.5) * ext[a];
            pf[3 * v + a] = (double)(float)q;  // f32 like tobj, widened (parser.rs:25-27)
        }
    // area-weighted vertex normals (f32 like an OBJ's vn records)
    std::vector<double> nrm(nv * 3, 0.0);
    for (size_t k = 0; k + 2 < ind.size(); k += 3) {
        uint32_t i0 = ind[k], i1 = ind[k + 1], i2 = ind[k + 2];
        double e1[3], e2[3];
        for (int a = 0; a < 3; a++) {
            e1[a] = pf[3 * i1 + a] - pf[3 * i0 + a];
            e2[a] = pf[3 * i2 + a] - pf[3 * i0 + a];
        }
        double c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        for (int a = 0; a < 3; a++) {
            nrm[3 * i0 + a] += c[a];
            nrm[3 * i1 + a] += c[a];
            nrm[3 * i2 + a] += c[a];
        }
    }
    out.p.resize(nv * 3);
    out.n.resize(nv * 3);
    for (size_t v = 0; v < nv; v++) {
        double l = std::sqrt(nrm[3 * v] * nrm[3 * v] + nrm[3 * v + 1] * nrm[3 * v + 1] + nrm[3 * v + 2] * nrm[3 * v + 2]);
        Vec3 n{0, 1, 0};
        if (l > 0.0) n = Vec3{(double)(float)(nrm[3 * v] / l), (double)(float)(nrm[3 * v + 1] / l), (double)(float)(nrm[3 * v + 2] / l)};
        Vec3 p = trans.transform_point(Vec3{pf[3 * v], pf[3 * v + 1], pf[3 * v + 2]});  // parser.rs:29
        Vec3 tn = trans.transform_vector(n);                                           // parser.rs:45
        out.p[3 * v] = p.x; out.p[3 * v + 1] = p.y; out.p[3 * v + 2] = p.z;
        out.n[3 * v] = tn.x; out.n[3 * v + 1] = tn.y; out.n[3 * v + 2] = tn.z;
    }
    out.ind = std::move(ind);
    return out;
}

// tobj 2.0.2 load_obj(path, triangulate = true) semantics for what parser.rs reads:
// positions / normals / texcoords of the FIRST model re-indexed to a single index
// per unique (v, vt, vn) triple; polygons fan-triangulated; values parsed as f32.
bool parse_obj(const std::string& path, const Mat4& trans, Mesh& out, std::string& err) {
    std::ifstream in(path);
    if (!in) {
        err = "Failed to parse obj " + path;  // parser.rs:84
        return false;
    }
    std::vector<float> pos, nor, tex;
    struct Idx {
        int v, vt, vn;
        bool operator<(const Idx& o) const {
            if (v != o.v) return v < o.v;
            if (vt != o.vt) return vt < o.vt;
            return vn < o.vn;
        }
    };
    std::map<Idx, uint32_t> remap;
    std::vector<float> opos, onor, otex;
    std::vector<uint32_t> oind;
    bool model_open = false, model_closed = false;
    std::string line;
    auto fix = [](int i, size_t n) { return i > 0 ? i - 1 : (int)((long long)n + i < -1 ? -1 : (long long)n + i); };
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ss(line);
        std::string tag;
        if (!(ss >> tag)) continue;
        if (tag == "v") {
            float x, y, z;
            ss >> x >> y >> z;
            pos.push_back(x); pos.push_back(y); pos.push_back(z);
        } else if (tag == "vn") {
            float x, y, z;
            ss >> x >> y >> z;
            nor.push_back(x); nor.push_back(y); nor.push_back(z);
        } else if (tag == "vt") {
            float u = 0, v = 0;
            ss >> u >> v;
            tex.push_back(u); tex.push_back(v);
        } else if (tag == "o" || tag == "g") {
            if (model_open) model_closed = true;  // only models[0] is read (parser.rs:21)
        } else if (tag == "f") {
            if (model_closed) continue;
            model_open = true;
            std::vector<uint32_t> poly;
            std::string tok;
            while (ss >> tok) {
                // v, v/vt, v//vn or v/vt/vn; every index a decimal integer that fits an int (sscanf("%d") wraps a value
                // that does not -- "4294967297" read as 1 -- where tobj fails the parse, so the fields are read with strtoll)
                Idx id{0, -1, -1};
                long long f3[3] = {0, 0, 0};
                bool have[3] = {false, false, false};
                int nf = 0;
                bool bad = false;
                const char* s = tok.c_str();
                for (;;) {
                    if (nf >= 3) { bad = true; break; }
                    if (*s != '/' && *s != '\0') {
                        char* end = nullptr;
                        errno = 0;
                        const long long v = std::strtoll(s, &end, 10);
                        if (end == s || errno == ERANGE || v > 2147483647ll || v < -2147483647ll - 1) { bad = true; break; }
                        f3[nf] = v;
                        have[nf] = true;
                        s = end;
                    }
                    nf++;
                    if (*s == '/') { s++; continue; }
                    if (*s != '\0') bad = true;
                    break;
                }
                if (bad || !have[0] || (nf == 2 && !have[1]) || (nf == 3 && !have[2])) {
                    err = "Failed to parse obj " + path;
                    return false;
                }
                id.v = fix((int)f3[0], pos.size() / 3);
                if (have[1]) id.vt = fix((int)f3[1], tex.size() / 2);
                if (have[2]) id.vn = fix((int)f3[2], nor.size() / 3);
                if (id.v < 0 || (size_t)id.v >= pos.size() / 3) {
                    err = "Failed to parse obj " + path;
                    return false;
                }
                auto it = remap.find(id);
                uint32_t k;
                if (it == remap.end()) {
                    k = (uint32_t)(opos.size() / 3);
                    remap.emplace(id, k);
                    for (int q = 0; q < 3; q++) opos.push_back(pos[3 * id.v + q]);
                    if (id.vn >= 0 && (size_t)id.vn < nor.size() / 3)
                        for (int q = 0; q < 3; q++) onor.push_back(nor[3 * id.vn + q]);
                    if (id.vt >= 0 && (size_t)id.vt < tex.size() / 2)
                        for (int q = 0; q < 2; q++) otex.push_back(tex[2 * id.vt + q]);
                } else {
                    k = it->second;
                }
                poly.push_back(k);
            }
            for (size_t q = 1; q + 1 < poly.size(); q++) {
                oind.push_back(poly[0]); oind.push_back(poly[q]); oind.push_back(poly[q + 1]);
            }
        }
    }
    size_t nv = opos.size() / 3;
    if (nv == 0 || oind.empty()) {
        err = "Failed to parse obj " + path;
        return false;
    }
    out = Mesh();
    out.p.resize(nv * 3);
    for (size_t v = 0; v < nv; v++) {
        Vec3 p = trans.transform_point(Vec3{(double)opos[3 * v], (double)opos[3 * v + 1], (double)opos[3 * v + 2]});
        out.p[3 * v] = p.x; out.p[3 * v + 1] = p.y; out.p[3 * v + 2] = p.z;
    }
    if (onor.size() == nv * 3) {
        out.n.resize(nv * 3);
        for (size_t v = 0; v < nv; v++) {
            Vec3 n = trans.transform_vector(Vec3{(double)onor[3 * v], (double)onor[3 * v + 1], (double)onor[3 * v + 2]});
            out.n[3 * v] = n.x; out.n[3 * v + 1] = n.y; out.n[3 * v + 2] = n.z;
        }
    }
    if (otex.size() == nv * 2) {
        out.uv.resize(nv * 2);
        for (size_t q = 0; q < nv * 2; q++) out.uv[q] = (double)otex[q];
    }
    out.ind = std::move(oind);
    return true;
}

}  // namespace rr
