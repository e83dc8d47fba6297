// tg_raster.h — interface between the host API (tg_api.hip) and the raster translation unit (tg_raster.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tg {

// Projection constants of the in-sensor camera (tactile_sensor.py:127-148), derived on the host in double precision
// and rounded once to float so that the oracle and the device use identical values:
//   kx = (1/tan(fov/2)) * W/2, ky likewise with H; window x = hw + kx * x/w, y = hh - ky * y/w;
//   depth = C0 + C1/w with C0 = far/(far-near), C1 = -near*far/(far-near)   (OpenGL depth-buffer value in [0,1]).
struct RasterParams {
    int W, H;
    float kx, ky, hw, hh, C0, C1, near_;
    float zcull;   // max of the undeformed depth image: triangles entirely behind it can never win the z-test
    int turn_off_border;
    int term_layer;   // fused auto-reset (launch_render's term_xform): the grid z of the terminal layer: 1 = dispatched last (TG_TERM_LAYER=0: first - measured, no gain)
    // k_render_blocks (small shared meshes, 128-multiple images), both made by make_block_tables or both null (then k_render_small draws):
    const float* blockmax;    // [regions][64]: largest undeformed depth of each kBlockW x (256 / kBlockW) block of each 128 x 128 region
    unsigned long long* drawn;   // [n_envs][regions]: bit b = block b of the env's image (`out`) differs from tmpl (written by k_render_blocks; null: every launch rewrites every block)
    const uint8_t* tmpl;      // [H][W]: t_s_camera's image of an untouched sensor (zero inside, the pasted ring outside)
    unsigned long long* kt;   // profiling mode: per-wavefront {start, end} wall-clock slots (tg_kt.hpp); null otherwise
#ifdef TG_TL_STAMPS
    unsigned long long* tl;
#endif
};
constexpr int kBlockW = 16;

// What the tactile camera looks at: a triangle mesh shared by all envs (edge, cube, ...) placed by a per-env rigid
// transform, or a per-env heightfield (surface_follow: createCollisionShape(GEOM_HEIGHTFIELD), base_surface_env.py:402-432)
// whose vertices are synthesised from the height samples:  vertex (i, j) = ((i - (rows-1)/2) s, (j - (cols-1)/2) s,
// h[j*rows + i] - zoff), cell (i, j) split into (i,j),(i,j+1),(i+1,j) and (i+1,j),(i,j+1),(i+1,j+1)
// [btHeightfieldTerrainShape::getVertex / processAllTriangles, PARITY_ASSUMPTIONS A15-A16].
struct Stimulus {
    int kind;                 // 0 mesh, 1 heightfield
    const float* verts;       // mesh
    const int32_t* tris;
    const float* soup;        // mesh, pre-expanded on the host: [n_tris][3][3] vertex coordinates (what the kernel reads)
    int n_tris;
    const double* heights;    // heightfield: [n_envs][rows*cols]; with hsel: [3][n_envs][rows*cols]
    const float* zoff;        // [n_envs]; with hsel: [3][n_envs]
    const uint8_t* hsel;      // nullptr, or [n_envs]: the live third of heights / zoff per env (env states, round 6: the surface an episode ran on stays
                              // in its slot for the terminal image while the next episode's is already in the next one - State::hsel)
    int rows, cols;
    float scale;
    int skip_quad_reject;     // 1: stimuli whose few triangles fill the camera's view (the pole's plate): the per-quad reject never fires
    int closed_outward;       // mesh: every surface is closed and consistently wound with outward normals (verified on the host):
                              // back faces are dropped at set-up for envs whose stimulus lies wholly beyond the near plane
    int fills_view;           // mesh: the stimulus covers the whole image in every frame (object_balance's plate on the sensor): launch_render keeps it off the block kernel
    int win_side;             // heightfield: largest side (in vertices) the frustum window can have (set by launch_render): sizes the LDS staging
};

RasterParams make_raster_params(int W, int H, double fov_deg, double near_, double far_, int turn_off_border, const float* nodef_dep_host);

// Device tables of k_render_blocks into P (one allocation, returned in *d_mem for the owner to hipFree; null and 0 when the image size has
// no block kernel); -1 when the allocation or the upload fails.
int make_block_tables(RasterParams& P, const float* nodef_dep_host, const float* nodef_gray_host, const uint8_t* border_host, int n_envs, void** d_mem);

// uint8(nodef_gray) per pixel (the border paste value), converted once on the host
void make_gray_u8(const float* nodef_gray_host, int npix, uint8_t* out_host);

// term_xform / term_mask / term_out (all or none): for the envs whose term_mask byte is non-zero the kernel first draws the image of
// term_xform (same SoA layout as xform) into term_out, then the regular one: the fused auto-reset of tg_step.
void launch_render(const RasterParams& P, const Stimulus& stim, const float* xform, int xform_soa, int n_envs,
                   const uint8_t* mask, const float* nodef_dep, const uint8_t* gray_u8, const uint8_t* border, uint8_t* out,
                   uint8_t* save_prev, const float* term_xform, const uint8_t* term_mask, uint8_t* term_out, hipStream_t stream);

void raster_debug_stats();   // development (-DTG_BLK_STAMPS): prints k_render_blocks' phase stamps; otherwise nothing

}  // namespace tg
