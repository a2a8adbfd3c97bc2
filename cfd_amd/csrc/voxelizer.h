// voxelizer.h -- host-side obstacle voxeliser of the product path.
//
// Replaces the reference's MeshVoxelizer (Engine/src/Utils/mesh_voxelizer.{h,cpp}) and the
// placement logic of lbm_solver::load_mesh (Engine/src/Physics/lbm_solver.cu:521-556).  The
// reference runs this step on the host too; the result here goes straight into the bit-packed
// device mask.  The mask must match the reference bit for bit, so the ray/triangle arithmetic is
// the same fp32 sequence; what differs is the search structure (triangles are bucketed per z row
// once instead of being culled again for every ray).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace lbmvox {

struct Vec3 {
    float x, y, z;
};
struct Triangle {
    Vec3 v0, v1, v2;
};

class Mesh {
   public:
    // mesh_voxelizer.cpp:20-86 -- OBJ `v`/`f` records; quads split by the shorter diagonal with
    // ties to [0,1,3],[1,2,3] as the vendored tinyobjloader does (tiny_obj_loader.h:1520-1616).
    bool load_obj(const std::string& path);
    void set_triangles(const float* xyz9, size_t ntris);
    // mesh_voxelizer.cpp:218-246
    void rotate_y(float angle_degrees);
    // mesh_voxelizer.cpp:146-216 -- sets 255 inside, leaves other bytes untouched.
    void voxelize(uint8_t* buffer, int width, int height, int depth, float scale) const;

    const std::vector<Triangle>& triangles() const { return tris_; }
    Vec3 min_bound() const { return bmin_; }
    Vec3 max_bound() const { return bmax_; }

   private:
    void reset_bounds();
    void grow(const Vec3& v);
    std::vector<Triangle> tris_;
    Vec3 bmin_{}, bmax_{};
};

// lbm_solver.cu:521-556: rotate -90 deg about y, scale = min over axes of 0.8*extent/mesh size,
// voxelise into a zeroed mask, force y = 0 solid.  Returns false when the file could not be read
// (the mask then holds only the floor, as in the reference).
bool build_wind_tunnel_mask(const std::string& obj_path, int width, int height, int depth,
                            std::vector<uint8_t>& mask, float* scale_out);

}  // namespace lbmvox
