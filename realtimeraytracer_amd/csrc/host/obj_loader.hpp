// obj_loader.hpp — Wavefront OBJ/MTL reader exposing the tinyobjloader data model the reference's
// ingest code consumes (reference src/core/file.cppm:44-269 calls tinyobj::LoadObj from the vendored
// external/tinyobjloader 2.0.0-rc.13).  Written from the documented behaviour of that loader, not
// from its source; pinned against the real loader (compiled from the reference tree into
// oracle/_ref) by tests/test_obj_ingest.py and the fixtures under tests/golden/.
//
// Behaviour reproduced (the parts the reference depends on):
//   * attrib_t {vertices xyz, normals xyz, texcoords uv}; 0-based index_t {vertex,normal,texcoord},
//     -1 when absent; negative (relative) OBJ indices resolved against the counts at that line;
//   * a shape per `o` / `g` statement, emitted only if it has faces; `usemtl` changes the
//     per-face material id without splitting the shape;
//   * triangulate=true: triangles pass through; quads are split along the shorter diagonal
//     (strictly shorter 0-2, else 1-3); polygons with more corners are fan-triangulated from corner 0
//     (the real loader ear-clips those — documented divergence for non-convex n-gons, n > 4);
//   * MTL: newmtl, Ka, Kd, Ks, map_Kd, map_Ks, map_Pm, map_d; unrecognised keys go to
//     unknown_parameter (the reference reads unknown_parameter["metallic"], file.cppm:229-231).
#pragma once
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace rtr::obj {

struct index_t { int vertex_index = -1, normal_index = -1, texcoord_index = -1; };
struct mesh_t {
    std::vector<index_t> indices;
    std::vector<unsigned int> num_face_vertices;
    std::vector<int> material_ids;
};
struct shape_t { std::string name; mesh_t mesh; };
struct attrib_t { std::vector<float> vertices, normals, texcoords; };
struct material_t {
    std::string name;
    float ambient[3] = {0, 0, 0}, diffuse[3] = {0, 0, 0}, specular[3] = {0, 0, 0};
    std::string diffuse_texname, specular_texname, metallic_texname, alpha_texname;
    std::map<std::string, std::string> unknown_parameter;
};

namespace detail {

inline std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

// strtof-compatible scalar parse; tinyobjloader's own parser agrees with strtof on ordinary
// decimal literals (the only kind our scenes and the Cornell model contain)
inline float parse_float(const char*& p) {
    while (*p == ' ' || *p == '\t') ++p;
    char* end = nullptr;
    float v = std::strtof(p, &end);
    if (end == p) { while (*p && *p != ' ' && *p != '\t') ++p; return 0.0f; }
    p = end;
    return v;
}

inline bool fix_index(int idx, int n, int* out) {
    if (idx > 0) { *out = idx - 1; return true; }
    if (idx == 0) return false;                 // 0 is not allowed by the OBJ spec
    *out = n + idx; return *out >= 0;
}

// i, i/j, i//k, i/j/k
inline bool parse_triple(const char*& p, int nv, int nvn, int nvt, index_t* out) {
    index_t r;
    char* end = nullptr;
    long v = std::strtol(p, &end, 10);
    if (end == p) return false;
    if (!fix_index((int)v, nv, &r.vertex_index)) return false;
    p = end;
    if (*p != '/') { *out = r; return true; }
    ++p;
    if (*p == '/') {                            // i//k
        ++p;
        long k = std::strtol(p, &end, 10);
        if (end == p) return false;
        if (!fix_index((int)k, nvn, &r.normal_index)) return false;
        p = end; *out = r; return true;
    }
    long j = std::strtol(p, &end, 10);
    if (end == p) return false;
    if (!fix_index((int)j, nvt, &r.texcoord_index)) return false;
    p = end;
    if (*p != '/') { *out = r; return true; }
    ++p;
    long k = std::strtol(p, &end, 10);
    if (end == p) return false;
    if (!fix_index((int)k, nvn, &r.normal_index)) return false;
    p = end; *out = r; return true;
}

struct Face { std::vector<index_t> corners; };

inline void flush_faces(shape_t* shape, std::vector<Face>& faces, int material, const std::string& name, bool triangulate,
                        const std::vector<float>& v, std::string* warn) {
    if (faces.empty()) return;
    shape->name = name;
    for (const Face& f : faces) {
        size_t n = f.corners.size();
        if (n < 3) { if (warn) *warn += "Degenerated face found\n."; continue; }
        auto push_tri = [&](const index_t& a, const index_t& b, const index_t& c) {
            shape->mesh.indices.push_back(a); shape->mesh.indices.push_back(b); shape->mesh.indices.push_back(c);
            shape->mesh.num_face_vertices.push_back(3);
            shape->mesh.material_ids.push_back(material);
        };
        if (triangulate && n != 3) {
            if (n == 4) {
                const index_t &i0 = f.corners[0], &i1 = f.corners[1], &i2 = f.corners[2], &i3 = f.corners[3];
                size_t a = (size_t)i0.vertex_index, b = (size_t)i1.vertex_index, c = (size_t)i2.vertex_index, d = (size_t)i3.vertex_index;
                if (3 * a + 2 >= v.size() || 3 * b + 2 >= v.size() || 3 * c + 2 >= v.size() || 3 * d + 2 >= v.size()) {
                    if (warn) *warn += "Face with invalid vertex index found.\n";
                    continue;
                }
                float e02[3], e13[3];
                for (int k = 0; k < 3; ++k) { e02[k] = v[3 * c + k] - v[3 * a + k]; e13[k] = v[3 * d + k] - v[3 * b + k]; }
                float sqr02 = e02[0] * e02[0] + e02[1] * e02[1] + e02[2] * e02[2];
                float sqr13 = e13[0] * e13[0] + e13[1] * e13[1] + e13[2] * e13[2];
                if (sqr02 < sqr13) { push_tri(i0, i1, i2); push_tri(i0, i2, i3); }
                else { push_tri(i0, i1, i3); push_tri(i1, i2, i3); }
            } else {
                for (size_t k = 1; k + 1 < n; ++k) push_tri(f.corners[0], f.corners[k], f.corners[k + 1]);
            }
        } else {
            for (const index_t& c : f.corners) shape->mesh.indices.push_back(c);
            shape->mesh.num_face_vertices.push_back((unsigned int)n);
            shape->mesh.material_ids.push_back(material);
        }
    }
    faces.clear();
}

inline bool load_mtl(const std::string& path, std::vector<material_t>* materials, std::map<std::string, int>* material_map,
                     std::string* warn) {
    std::ifstream in(path);
    if (!in) { if (warn) *warn += "Material file [ " + path + " ] not found.\n"; return false; }
    material_t cur; bool has = false;
    std::string line;
    auto commit = [&]() {
        if (!has) return;
        (*material_map)[cur.name] = (int)materials->size();
        materials->push_back(cur);
    };
    while (std::getline(in, line)) {
        std::string t = trim(line);
        if (t.empty() || t[0] == '#') continue;
        size_t sp = t.find_first_of(" \t");
        std::string key = t.substr(0, sp);
        std::string rest = sp == std::string::npos ? std::string() : trim(t.substr(sp + 1));
        const char* p = rest.c_str();
        if (key == "newmtl") { commit(); cur = material_t(); cur.name = rest; has = true; }
        else if (key == "Ka") { for (int k = 0; k < 3; ++k) cur.ambient[k] = parse_float(p); }
        else if (key == "Kd") { for (int k = 0; k < 3; ++k) cur.diffuse[k] = parse_float(p); }
        else if (key == "Ks") { for (int k = 0; k < 3; ++k) cur.specular[k] = parse_float(p); }
        else if (key == "map_Kd") cur.diffuse_texname = rest;
        else if (key == "map_Ks") cur.specular_texname = rest;
        else if (key == "map_Pm") cur.metallic_texname = rest;
        else if (key == "map_d") cur.alpha_texname = rest;
        else if (key == "Ke" || key == "Ns" || key == "Ni" || key == "d" || key == "Tr" || key == "illum" || key == "Tf" ||
                 key == "map_Ka" || key == "map_Ns" || key == "map_bump" || key == "map_Bump" || key == "bump" ||
                 key == "map_Ke" || key == "Pr" || key == "Pm" || key == "Ps" || key == "Pc" || key == "Pcr" ||
                 key == "aniso" || key == "anisor" || key == "map_Pr" || key == "map_Ps" || key == "norm" || key == "disp" ||
                 key == "refl" || key == "map_Ni") { /* recognised by tinyobjloader, unused by the reference */ }
        else cur.unknown_parameter[key] = rest;
    }
    commit();
    return true;
}

}  // namespace detail

// Same signature shape as tinyobj::LoadObj(attrib, shapes, materials, warn, err, filename, mtl_basedir, triangulate).
inline bool LoadObj(attrib_t* attrib, std::vector<shape_t>* shapes, std::vector<material_t>* materials, std::string* warn,
                    std::string* err, const char* filename, const char* mtl_basedir = nullptr, bool triangulate = true) {
    attrib->vertices.clear(); attrib->normals.clear(); attrib->texcoords.clear();
    shapes->clear(); materials->clear();
    std::ifstream in(filename);
    if (!in) { if (err) *err += std::string("Cannot open file [") + filename + "]\n"; return false; }
    std::string baseDir;
    if (mtl_basedir) {
        baseDir = mtl_basedir;
        if (!baseDir.empty() && baseDir.back() != '/' && baseDir.back() != '\\') baseDir += '/';
    } else {
        std::string f(filename);
        size_t slash = f.find_last_of("/\\");
        if (slash != std::string::npos) baseDir = f.substr(0, slash + 1);
    }
    std::vector<float>& v = attrib->vertices; std::vector<float>& vn = attrib->normals; std::vector<float>& vt = attrib->texcoords;
    std::map<std::string, int> material_map;
    std::vector<detail::Face> faces;
    shape_t shape; std::string name; int material = -1;
    std::string line; size_t line_num = 0;
    while (std::getline(in, line)) {
        ++line_num;
        std::string t = detail::trim(line);
        if (t.empty() || t[0] == '#') continue;
        const char* p = t.c_str();
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2; for (int k = 0; k < 3; ++k) v.push_back(detail::parse_float(p));
        } else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) {
            p += 3; for (int k = 0; k < 3; ++k) vn.push_back(detail::parse_float(p));
        } else if (p[0] == 'v' && p[1] == 't' && (p[2] == ' ' || p[2] == '\t')) {
            p += 3; for (int k = 0; k < 2; ++k) vt.push_back(detail::parse_float(p));
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            detail::Face f;
            for (;;) {
                while (*p == ' ' || *p == '\t') ++p;
                if (!*p || *p == '#') break;
                index_t idx;
                if (!detail::parse_triple(p, (int)(v.size() / 3), (int)(vn.size() / 3), (int)(vt.size() / 2), &idx)) {
                    if (err) { std::ostringstream ss; ss << "Failed to parse `f' line (e.g. a zero value for vertex index or invalid relative vertex index). Line " << line_num << ").\n"; *err += ss.str(); }
                    return false;
                }
                f.corners.push_back(idx);
            }
            faces.push_back(std::move(f));
        } else if (t.compare(0, 6, "usemtl") == 0 && t.size() > 6 && (t[6] == ' ' || t[6] == '\t')) {
            std::string mname = detail::trim(t.substr(7));
            int newId = -1;
            auto it = material_map.find(mname);
            if (it != material_map.end()) newId = it->second;
            else if (warn) *warn += "material [ '" + mname + "' ] not found in .mtl\n";
            if (newId != material) {
                detail::flush_faces(&shape, faces, material, name, triangulate, v, warn);
                material = newId;
            }
        } else if (t.compare(0, 6, "mtllib") == 0 && t.size() > 6 && (t[6] == ' ' || t[6] == '\t')) {
            std::istringstream ss(t.substr(7));
            std::string fn; bool found = false;
            while (ss >> fn) {
                std::string w;
                if (detail::load_mtl(baseDir + fn, materials, &material_map, &w)) { found = true; break; }
                if (warn) *warn += w;
            }
            if (!found && warn) *warn += "Failed to load material file(s). Use default material.\n";
        } else if ((p[0] == 'g' || p[0] == 'o') && (p[1] == ' ' || p[1] == '\t' || p[1] == '\0')) {
            detail::flush_faces(&shape, faces, material, name, triangulate, v, warn);
            if (!shape.mesh.indices.empty()) shapes->push_back(shape);
            shape = shape_t();
            faces.clear();
            std::string rest = detail::trim(t.substr(1));
            if (p[0] == 'g') {
                // multiple group names are joined with single spaces
                std::istringstream ss(rest); std::string w, joined;
                while (ss >> w) { if (!joined.empty()) joined += ' '; joined += w; }
                name = joined;
            } else {
                name = rest;
            }
        }
        // 'l', 'p', 's', 't' and anything else: ignored (the reference reads only triangle meshes)
    }
    detail::flush_faces(&shape, faces, material, name, triangulate, v, warn);
    if (!shape.mesh.indices.empty()) shapes->push_back(shape);
    return true;
}

}  // namespace rtr::obj
