// voxelizer.cpp -- see voxelizer.h.  Build with -ffp-contract=off: the mask has to match the
// reference's fp32 arithmetic bit for bit (mesh_voxelizer.cpp:89-121, 146-216).
#include "voxelizer.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>

namespace lbmvox {

namespace {

inline Vec3 sub(const Vec3& a, const Vec3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
// same component formulas as glm::cross / glm::dot, which the reference calls
inline Vec3 cross(const Vec3& a, const Vec3& b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline float dot(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

constexpr float kEps = 0.0000001f;  // mesh_voxelizer.cpp:12

// +X ray against one triangle; true with the hit distance when it counts as an intersection.
inline bool hit_plus_x(const Vec3& origin, const Triangle& t, float& dist) {
    const Vec3 dir{1.0f, 0.0f, 0.0f};
    const Vec3 e1 = sub(t.v1, t.v0);
    const Vec3 e2 = sub(t.v2, t.v0);
    const Vec3 h = cross(dir, e2);
    const float a = dot(e1, h);
    if (a > -kEps && a < kEps) return false;
    const float f = 1.0f / a;
    const Vec3 s = sub(origin, t.v0);
    const float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    const Vec3 q = cross(s, e1);
    const float v = f * dot(dir, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    dist = f * dot(e2, q);
    return dist > kEps;
}

struct Extent {
    float ymin, ymax, zmin, zmax;
};

}  // namespace

void Mesh::reset_bounds() {
    const float hi = std::numeric_limits<float>::max();
    const float lo = std::numeric_limits<float>::lowest();
    bmin_ = {hi, hi, hi};
    bmax_ = {lo, lo, lo};
}

void Mesh::grow(const Vec3& v) {
    bmin_ = {std::min(bmin_.x, v.x), std::min(bmin_.y, v.y), std::min(bmin_.z, v.z)};
    bmax_ = {std::max(bmax_.x, v.x), std::max(bmax_.y, v.y), std::max(bmax_.z, v.z)};
}

void Mesh::set_triangles(const float* xyz9, size_t ntris) {
    tris_.clear();
    reset_bounds();
    for (size_t i = 0; i < ntris; i++) {
        const float* p = xyz9 + 9 * i;
        Triangle t{{p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]}};
        tris_.push_back(t);
        grow(t.v0); grow(t.v1); grow(t.v2);
    }
}

namespace {

// Faces with more than four vertices.  The reference's OBJ loader cuts ears (tiny_obj_loader.h:1751-1973), and
// which triangles come out decides the mask wherever a face is concave, so the same search is run here: project
// on the two axes of the first real corner's dominant normal component, walk a cursor round the remaining
// vertices, take a triple as an ear when it turns the way the loader's area term says and holds no other
// remaining vertex, drop its middle vertex, stop after a whole round without an ear.
template <class Emit>
void clip_ears(const std::vector<Vec3>& verts, const std::vector<long>& face, Emit&& emit) {
    const size_t n = face.size();
    auto coord = [](const Vec3& p, int axis) { return axis == 0 ? p.x : (axis == 1 ? p.y : p.z); };
    int ax0 = 1, ax1 = 2;
    for (size_t k = 0; k < n; k++) {
        const Vec3 &a = verts[face[k]], &b = verts[face[(k + 1) % n]], &c = verts[face[(k + 2) % n]];
        const Vec3 e0 = sub(b, a), e1 = sub(c, b);
        const float nx = std::fabs(e0.y * e1.z - e0.z * e1.y);
        const float ny = std::fabs(e0.z * e1.x - e0.x * e1.z);
        const float nz = std::fabs(e0.x * e1.y - e0.y * e1.x);
        const float eps = std::numeric_limits<float>::epsilon();
        if (nx > eps || ny > eps || nz > eps) {
            const bool x_dominates = nx > ny && nx > nz;
            if (!x_dominates) {
                ax0 = 0;
                if (nz > nx && nz > ny) ax1 = 1;
            }
            break;
        }
    }
    // is (tx, ty) inside the triangle (px, py)?  Crossing-number test over the three edges.
    auto inside = [](const float* px, const float* py, float tx, float ty) {
        bool in = false;
        for (int i = 0, j = 2; i < 3; j = i++)
            if ((py[i] > ty) != (py[j] > ty) && tx < (px[j] - px[i]) * (ty - py[i]) / (py[j] - py[i]) + px[i]) in = !in;
        return in;
    };
    std::vector<long> left(face);
    size_t cursor = 0, budget = n, seen = n;
    while (left.size() > 3 && budget > 0) {
        const size_t m = left.size();
        if (cursor >= m) cursor -= m;
        if (seen != m) { seen = m; budget = m; } else { budget--; }
        long id[3];
        float px[3], py[3];
        for (int k = 0; k < 3; k++) {
            id[k] = left[(cursor + k) % m];
            px[k] = coord(verts[id[k]], ax0);
            py[k] = coord(verts[id[k]], ax1);
        }
        const float turn = (px[1] - px[0]) * (py[2] - py[1]) - (py[1] - py[0]) * (px[2] - px[1]);
        const float area = (px[0] * py[1] - py[0] * px[1]) * 0.5f;
        bool ear = !(turn * area < 0.0f);
        for (size_t o = 3; ear && o < m; o++) {
            const Vec3& q = verts[left[(cursor + o) % m]];
            if (inside(px, py, coord(q, ax0), coord(q, ax1))) ear = false;
        }
        if (!ear) { cursor++; continue; }
        emit(verts[id[0]], verts[id[1]], verts[id[2]]);
        left.erase(left.begin() + static_cast<long>((cursor + 1) % m));
    }
    if (left.size() == 3) emit(verts[left[0]], verts[left[1]], verts[left[2]]);
}

}  // namespace

bool Mesh::load_obj(const std::string& path) {
    std::ifstream in(path);
    if (!in) return false;
    tris_.clear();
    reset_bounds();
    std::vector<Vec3> verts;
    std::string line;
    auto emit = [&](const Vec3& a, const Vec3& b, const Vec3& c) {
        tris_.push_back({a, b, c});
        grow(a); grow(b); grow(c);
    };
    while (std::getline(in, line)) {
        const char* p = line.c_str();
        while (*p == ' ' || *p == '\t') p++;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char* e = const_cast<char*>(p + 1);
            const double x = std::strtod(e, &e);
            const double y = std::strtod(e, &e);
            const double z = std::strtod(e, &e);
            verts.push_back({static_cast<float>(x), static_cast<float>(y), static_cast<float>(z)});
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            std::vector<long> ids;
            const char* q = p + 1;
            for (;;) {
                while (*q == ' ' || *q == '\t') q++;
                if (*q == '\0' || *q == '\r' || *q == '\n' || *q == '#') break;
                char* e = nullptr;
                const long vi = std::strtol(q, &e, 10);
                if (e == q) break;
                while (*e && *e != ' ' && *e != '\t' && *e != '\r' && *e != '\n') e++;  // "/vt/vn"
                q = e;
                ids.push_back(vi > 0 ? vi - 1 : static_cast<long>(verts.size()) + vi);
            }
            if (ids.size() < 3) continue;
            bool ok = true;
            for (long id : ids) ok = ok && id >= 0 && static_cast<size_t>(id) < verts.size();
            if (!ok) continue;
            if (ids.size() == 3) {
                emit(verts[ids[0]], verts[ids[1]], verts[ids[2]]);
            } else if (ids.size() == 4) {
                const Vec3 &a = verts[ids[0]], &b = verts[ids[1]], &c = verts[ids[2]], &d = verts[ids[3]];
                const Vec3 ac = sub(c, a), bd = sub(d, b);
                const float len_ac = ac.x * ac.x + ac.y * ac.y + ac.z * ac.z;
                const float len_bd = bd.x * bd.x + bd.y * bd.y + bd.z * bd.z;
                if (len_ac < len_bd) { emit(a, b, c); emit(a, c, d); }
                else                 { emit(a, b, d); emit(b, c, d); }
            } else {
                clip_ears(verts, ids, emit);
            }
        }
    }
    return true;
}

void Mesh::rotate_y(float angle_degrees) {
    const float rad = angle_degrees * 0.01745329251994329576923690768489f;  // glm::radians
    const float c = std::cos(rad);
    const float s = std::sin(rad);
    reset_bounds();
    for (Triangle& t : tris_) {
        for (Vec3* v : {&t.v0, &t.v1, &t.v2}) {
            const float x = v->x * c + v->z * s;
            const float z = -v->x * s + v->z * c;
            v->x = x;
            v->z = z;
        }
        grow(t.v0); grow(t.v1); grow(t.v2);
    }
}

void Mesh::voxelize(uint8_t* buffer, int width, int height, int depth, float scale) const {
    const Vec3 grid_c{width / 2.0f, height / 2.0f, depth / 2.0f};
    Vec3 mesh_c;
    mesh_c.x = (bmin_.x + bmax_.x) / 2;
    mesh_c.y = bmin_.y + grid_c.y / scale;  // lowest mesh point sits on grid y = 0
    mesh_c.z = (bmin_.z + bmax_.z) / 2;

    const size_t nt = tris_.size();
    std::vector<Extent> ext(nt);
    for (size_t i = 0; i < nt; i++) {
        const Triangle& t = tris_[i];
        ext[i] = {std::min({t.v0.y, t.v1.y, t.v2.y}), std::max({t.v0.y, t.v1.y, t.v2.y}),
                  std::min({t.v0.z, t.v1.z, t.v2.z}), std::max({t.v0.z, t.v1.z, t.v2.z})};
    }

#pragma omp parallel
    {
        std::vector<uint32_t> cand;
        std::vector<float> hits;
#pragma omp for schedule(dynamic, 1)
        for (int z = 0; z < depth; z++) {
            const float oz = (static_cast<float>(z) - grid_c.z) / scale + mesh_c.z;
            // triangles whose z extent admits this row of rays (same test the reference applies
            // per ray, hoisted: it does not depend on y)
            cand.clear();
            for (size_t i = 0; i < nt; i++)
                if (!(oz < ext[i].zmin || oz > ext[i].zmax)) cand.push_back(static_cast<uint32_t>(i));
            if (cand.empty()) continue;
            for (int y = 0; y < height; y++) {
                Vec3 o;
                o.x = (0.0f - grid_c.x) / scale + mesh_c.x;
                o.y = (static_cast<float>(y) - grid_c.y) / scale + mesh_c.y;
                o.z = oz;
                hits.clear();
                for (uint32_t i : cand) {
                    if (o.y < ext[i].ymin || o.y > ext[i].ymax) continue;
                    float d;
                    if (hit_plus_x(o, tris_[i], d)) hits.push_back(d);
                }
                if (hits.size() < 2) continue;
                std::sort(hits.begin(), hits.end());
                uint8_t* row = buffer + (static_cast<size_t>(z) * height + y) * width;
                for (size_t i = 0; i + 1 < hits.size(); i += 2) {
                    const int x0 = std::max(0, static_cast<int>(std::ceil(hits[i] * scale)));
                    const int x1 = std::min(width, static_cast<int>(std::floor(hits[i + 1] * scale)));
                    if (x1 > x0) std::memset(row + x0, 255, static_cast<size_t>(x1 - x0));
                }
            }
        }
    }
}

bool build_wind_tunnel_mask(const std::string& obj_path, int width, int height, int depth,
                            std::vector<uint8_t>& mask, float* scale_out) {
    Mesh mesh;
    const bool loaded = mesh.load_obj(obj_path);
    if (!loaded) mesh.set_triangles(nullptr, 0);
    mask.assign(static_cast<size_t>(width) * height * depth, 0);
    mesh.rotate_y(-90.0f);
    const Vec3 lo = mesh.min_bound(), hi = mesh.max_bound();
    const float sx = (static_cast<float>(width) * 0.8f) / (hi.x - lo.x);
    const float sy = (static_cast<float>(height) * 0.8f) / (hi.y - lo.y);
    const float sz = (static_cast<float>(depth) * 0.8f) / (hi.z - lo.z);
    const float scale = std::min({sx, sy, sz});
    if (scale_out) *scale_out = scale;
    mesh.voxelize(mask.data(), width, height, depth, scale);
    for (int z = 0; z < depth; z++)
        std::memset(mask.data() + static_cast<size_t>(z) * width * height, 255, static_cast<size_t>(width));
    return loaded;
}

}  // namespace lbmvox
