// tg_scene.h — interface between the host API (tg_api.hip) and the scene-camera translation unit (tg_scene.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace tg {

// The envs' fixed world camera (get_visual_obs, base_tactile_env.py:212-245): one indexed triangle set shared by all envs, every
// triangle rigid in one of n_frames frames (0 world, 1 + i moving link i, n_frames - 1 the task's stimulus / free body); per env the
// eye <- frame transforms [n_frames][12] (R row-major, t) rounded once to float.  Projection constants are derived on the host in
// double and rounded once, as for the tactile camera: window x = hw + kx x/w, y = hh - ky y/w (w = -z_eye), aspect = W / H.
// vstart / vcount: the chunk's own copy of its (<= 64) distinct vertices, contiguous in `verts`, so that a wavefront transforms each vertex
// of the chunk once (a lane per vertex) and the chunk's triangles pick their corners by a one-byte local index.
struct SceneChunk { float cx, cy, cz, r; int start, frame, vstart; uint16_t count, vcount; };

struct SceneParams {
    int W, H, n_tris, n_frames;
    float kx, ky, hw, hh, near_, far_, inv_near, inv_far;
    float light_eye[3];            // unit vector towards the light, eye space
    uint8_t background[3];
    const float* verts;            // device [n_cverts][3]: chunk-ordered (a vertex shared by two chunks is stored once in each)
    const int32_t* tris;           // device [n_tris][3], indices into verts
    const uint32_t* tri_local;     // device [n_tris]: the same three corners relative to the chunk's vstart, i0 | i1 << 8 | i2 << 16
    const uint32_t* tri_attr;      // device [n_tris]: frame << 24 | r << 16 | g << 8 | b
    const SceneChunk* chunks;      // device [n_chunks]: runs of <= 64 spatially sorted triangles of one frame with a bounding sphere
    int n_chunks;
    const unsigned long long* static_keys;   // device [tiles][tile pixels]: z keys of the world frame's triangles (launch_scene_static), nullptr: draw them per env
    // surface envs: the task's body is a per-env heightfield (createCollisionShape(GEOM_HEIGHTFIELD), base_surface_env.py:402-432), drawn in frame
    // n_frames - 1 with one colour; vertices and triangle order as in tg_raster.h (Stimulus): vertex (i, j) = ((i - (rows-1)/2) s,
    // (j - (cols-1)/2) s, h[j rows + i] - zoff), cell (i, j) -> (i,j),(i,j+1),(i+1,j) and (i+1,j),(i,j+1),(i+1,j+1)
    const double* hf_heights;      // device [n_envs][rows * cols], nullptr: no heightfield
    const float* hf_zoff;          // device [n_envs]
    const uint8_t* hf_sel;         // nullptr, or device [n_envs]: hf_heights / hf_zoff are [3][hf_n][...] and this names each env's live third (State::hsel)
    int hf_n;                      // number of envs (the stride of a third)
    int hf_rows, hf_cols;
    float hf_scale;
    uint32_t hf_rgb;               // r << 16 | g << 8 | b
    // Translucent spheres blended over the finished image (goal indicators, trajectory markers; PARITY_ASSUMPTIONS A33b): device
    // [n_envs][n_spheres][8] floats = centre in eye space (3), radius, r, g, b (0..255), alpha (0: slot unused); written per env by k_scene_xf.
    // Per pixel centre and sphere, in list order: the ray's near intersection at eye depth w counts when near <= w <= far and 1 / w is
    // strictly above the opaque key's 1 / w; shaded like a triangle (0.6 + 0.35 max(0, n . l)) and blended
    // out = (uint8)(alpha src + (1 - alpha) dst + 0.5); no depth write, no test between translucent fragments (oracle: mb_blend_spheres).
    const float* spheres;
    int n_spheres;
};

SceneParams make_scene_params(int W, int H, double fov_deg, double near_, double far_);

// Sorts the triangles by (frame, Morton code of the centroid) - the image does not depend on their order - and cuts them into chunks of
// <= 64 triangles with <= 64 distinct vertices; cverts receives the chunk-ordered vertex copies, tris is re-pointed at them.
void build_scene_chunks(const float* verts, int32_t* tris /*[n][3], reordered + re-indexed*/, uint32_t* attr /*[n], reordered*/, int n_tris,
                        std::vector<SceneChunk>& chunks, std::vector<float>& cverts, std::vector<uint32_t>& tri_local);

// Draws out[env] (uint8 [H][W][3]) for every env (mask == nullptr) or the envs whose mask byte is non-zero; with save_prev the
// previous image of a drawn env is first copied to save_prev[env] (the terminal observation of an auto-reset).
void launch_scene(const SceneParams& P, const float* xf, int n_envs, const uint8_t* mask, uint8_t* out, uint8_t* save_prev, hipStream_t stream);
// One-off: draws the triangles of frame 0 (world: the same for every env, the camera is fixed) and stores their z keys, W * H of them, tile by
// tile; xf_env0 = any env's [n_frames][12] transforms (only frame 0 is read).
void launch_scene_static(const SceneParams& P, const float* xf_env0, unsigned long long* static_keys, hipStream_t stream);
int scene_prepare(const SceneParams& P);   // one-time kernel attributes (large dynamic LDS); call outside stream capture; non-zero: the scene does not fit

void scene_debug_stats();          // -DTG_SCENE_STATS builds: per-workgroup work counters to stderr (development)

}  // namespace tg
