// tg_api_ops.hip - the context-free function-level entry points of the C ABI (include/tactile_gym_hip.h): one PyBullet call of the reference each
// (calculateInverseDynamics, calculateMassMatrix, calculateJacobian, stepSimulation ticks, calculateInverseKinematics, getCameraImage +
// t_s_camera, the heightfield generator) over a batch of states, for tests and for callers that want the pieces.  Split out of tg_api.hip in round 6.
#include "tg_ctx.hpp"

using namespace tg;

extern "C" {

// ---------------------------------------------------------------------------------------------------- function-level entry points
#define TG_FN_DISPATCH(robot, dtype, KERNEL, n, ...)                                                                              \
    do {                                                                                                                          \
        DevBuf rb;                                                                                                                \
        dim3 grid(((n) + 63) / 64), block(64);                                                                                    \
        if ((dtype) == TG_PHYSICS_F64) {                                                                                          \
            if (int rc = upload_robot<double>(robot, rb)) return rc;                                                              \
            if ((robot)->topology == 0) hipLaunchKernelGGL((KERNEL<double, 0>), grid, block, 0, 0, (const DevRobot<double>*)rb.p, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<double, 1>), grid, block, 0, 0, (const DevRobot<double>*)rb.p, __VA_ARGS__);          \
        } else {                                                                                                                  \
            if (int rc = upload_robot<float>(robot, rb)) return rc;                                                               \
            if ((robot)->topology == 0) hipLaunchKernelGGL((KERNEL<float, 0>), grid, block, 0, 0, (const DevRobot<float>*)rb.p, __VA_ARGS__);   \
            else hipLaunchKernelGGL((KERNEL<float, 1>), grid, block, 0, 0, (const DevRobot<float>*)rb.p, __VA_ARGS__);            \
        }                                                                                                                         \
        TG_HIP(hipDeviceSynchronize());                                                                                           \
    } while (0)

int tg_inverse_dynamics(const tg_robot* robot, int32_t dtype, int32_t n, const double* q, const double* qd, const double* qdd, double* tau) {
    if (int rc = check_robot(robot)) return rc;
    if (int rc = need_device()) return rc;
    const size_t bytes = (size_t)n * robot->ndof * 8;
    DevBuf a, b, c, d;
    if (a.alloc(bytes) || b.alloc(bytes) || c.alloc(bytes) || d.alloc(bytes)) return fail(-2, "hipMalloc failed");
    TG_HIP(hipMemcpy(a.p, q, bytes, hipMemcpyHostToDevice)); TG_HIP(hipMemcpy(b.p, qd, bytes, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(c.p, qdd, bytes, hipMemcpyHostToDevice));
    TG_FN_DISPATCH(robot, dtype, k_inverse_dynamics, n, n, (const double*)a.p, (const double*)b.p, (const double*)c.p, (double*)d.p);
    TG_HIP(hipMemcpy(tau, d.p, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int tg_mass_matrix(const tg_robot* robot, int32_t dtype, int32_t n, const double* q, double* M) {
    if (int rc = check_robot(robot)) return rc;
    if (int rc = need_device()) return rc;
    const int nd = robot->ndof;
    DevBuf a, b;
    if (a.alloc((size_t)n * nd * 8) || b.alloc((size_t)n * nd * nd * 8)) return fail(-2, "hipMalloc failed");
    TG_HIP(hipMemcpy(a.p, q, (size_t)n * nd * 8, hipMemcpyHostToDevice));
    TG_FN_DISPATCH(robot, dtype, k_mass_matrix, n, n, (const double*)a.p, (double*)b.p);
    TG_HIP(hipMemcpy(M, b.p, (size_t)n * nd * nd * 8, hipMemcpyDeviceToHost));
    return 0;
}

int tg_jacobian_tcp(const tg_robot* robot, int32_t dtype, int32_t n, const double* q, double* J, double* pos, double* rot) {
    if (int rc = check_robot(robot)) return rc;
    if (int rc = need_device()) return rc;
    const int nd = robot->ndof;
    DevBuf a, b, c, d;
    if (a.alloc((size_t)n * nd * 8) || b.alloc((size_t)n * 6 * nd * 8) || c.alloc((size_t)n * 24) || d.alloc((size_t)n * 72)) return fail(-2, "hipMalloc failed");
    TG_HIP(hipMemcpy(a.p, q, (size_t)n * nd * 8, hipMemcpyHostToDevice));
    TG_FN_DISPATCH(robot, dtype, k_jacobian, n, n, (const double*)a.p, (double*)b.p, (double*)c.p, (double*)d.p);
    if (J) TG_HIP(hipMemcpy(J, b.p, (size_t)n * 6 * nd * 8, hipMemcpyDeviceToHost));
    if (pos) TG_HIP(hipMemcpy(pos, c.p, (size_t)n * 24, hipMemcpyDeviceToHost));
    if (rot) TG_HIP(hipMemcpy(rot, d.p, (size_t)n * 72, hipMemcpyDeviceToHost));
    return 0;
}

int tg_sim_ticks(const tg_robot* robot, int32_t dtype, int32_t n, int32_t n_ticks, int32_t iters, double dt, int32_t motor_mode,
                 const double* q_des, const double* qd_des, double max_force, double* q, double* qd) {
    if (int rc = check_robot(robot)) return rc;
    if (int rc = need_device()) return rc;
    const size_t bytes = (size_t)n * robot->ndof * 8;
    DevBuf a, b, c, d;
    if (a.alloc(bytes) || b.alloc(bytes) || c.alloc(bytes) || d.alloc(bytes)) return fail(-2, "hipMalloc failed");
    TG_HIP(hipMemcpy(a.p, q, bytes, hipMemcpyHostToDevice)); TG_HIP(hipMemcpy(b.p, qd, bytes, hipMemcpyHostToDevice));
    if (q_des) TG_HIP(hipMemcpy(c.p, q_des, bytes, hipMemcpyHostToDevice));
    if (qd_des) TG_HIP(hipMemcpy(d.p, qd_des, bytes, hipMemcpyHostToDevice));
    TG_FN_DISPATCH(robot, dtype, k_sim_ticks, n, n, n_ticks, iters, dt, motor_mode, q_des ? (const double*)c.p : (const double*)nullptr,
                   qd_des ? (const double*)d.p : (const double*)nullptr, max_force, (double*)a.p, (double*)b.p);
    TG_HIP(hipMemcpy(q, a.p, bytes, hipMemcpyDeviceToHost)); TG_HIP(hipMemcpy(qd, b.p, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int tg_inverse_kinematics(const tg_robot* robot, int32_t dtype, int32_t n, const double* q0, const double* target_pos, const double* target_rot,
                          int32_t max_iters, double threshold, double* q_out, int32_t* iters) {
    if (int rc = check_robot(robot)) return rc;
    if (int rc = need_device()) return rc;
    const int nd = robot->ndof;
    DevBuf a, b, c, d, e;
    if (a.alloc((size_t)n * nd * 8) || b.alloc((size_t)n * 24) || c.alloc((size_t)n * 72) || d.alloc((size_t)n * nd * 8) || e.alloc((size_t)n * 4))
        return fail(-2, "hipMalloc failed");
    TG_HIP(hipMemcpy(a.p, q0, (size_t)n * nd * 8, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(b.p, target_pos, (size_t)n * 24, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(c.p, target_rot, (size_t)n * 72, hipMemcpyHostToDevice));
    TG_FN_DISPATCH(robot, dtype, k_ik, n, n, (const double*)a.p, (const double*)b.p, (const double*)c.p, max_iters, threshold, (double*)d.p,
                   (int32_t*)e.p);
    TG_HIP(hipMemcpy(q_out, d.p, (size_t)n * nd * 8, hipMemcpyDeviceToHost));
    if (iters) TG_HIP(hipMemcpy(iters, e.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

int tg_render_tactile(const tg_sensor* sen, const tg_mesh* mesh, int32_t n, const float* xf, uint8_t* out) {
    if (!sen || !mesh || !xf || !out) return fail(-1, "NULL argument");
    if (int rc = need_device()) return rc;
    const int H = sen->image_h, W = sen->image_w;
    if (!((H % 128 == 0 && W % 128 == 0) || (H == 64 && W == 64))) return fail(-1, "image size must be 64x64 or a multiple of 128");
    const size_t npix = (size_t)H * W;
    DevBuf nd, ng, bm, vv, tt, xx, oo;
    if (nd.alloc(npix * 4) || ng.alloc(npix * 4) || bm.alloc(npix) || vv.alloc((size_t)mesh->n_verts * 12) || tt.alloc((size_t)mesh->n_tris * 12) ||
        xx.alloc((size_t)n * 48) || oo.alloc(npix * n))
        return fail(-2, "hipMalloc failed");
    TG_HIP(hipMemcpy(nd.p, sen->nodef_dep, npix * 4, hipMemcpyHostToDevice));
    { std::vector<uint8_t> g8(npix); make_gray_u8(sen->nodef_gray, (int)npix, g8.data()); TG_HIP(hipMemcpy(ng.p, g8.data(), npix, hipMemcpyHostToDevice)); }
    TG_HIP(hipMemcpy(bm.p, sen->border_mask, npix, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(vv.p, mesh->verts, (size_t)mesh->n_verts * 12, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(tt.p, mesh->tris, (size_t)mesh->n_tris * 12, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(xx.p, xf, (size_t)n * 48, hipMemcpyHostToDevice));
    TG_HIP(hipMemset(oo.p, 0, npix * n));
    RasterParams P = make_raster_params(W, H, sen->fov_deg, sen->near_plane, sen->far_plane, sen->turn_off_border, sen->nodef_dep);
    DevBuf bt;
    if (make_block_tables(P, sen->nodef_dep, sen->nodef_gray, sen->border_mask, n, &bt.p)) return fail(-2, "hipMalloc failed");
    Stimulus S{};
    S.closed_outward = (mesh_closed_outward(mesh) && getenv("TG_NO_BACKFACE_CULL") == nullptr) ? 1 : 0;   // env var: A/B measurements only
    DevBuf sp;
    {
        std::vector<float> soup((size_t)mesh->n_tris * 9);
        for (int t = 0; t < mesh->n_tris; ++t)
            for (int k = 0; k < 3; ++k)
                for (int a = 0; a < 3; ++a) soup[(size_t)t * 9 + 3 * k + a] = mesh->verts[3 * (size_t)mesh->tris[3 * t + k] + a];
        if (sp.alloc(soup.size() * 4 + 4)) return fail(-2, "hipMalloc failed");
        TG_HIP(hipMemcpy(sp.p, soup.data(), soup.size() * 4, hipMemcpyHostToDevice));
    }
    S.kind = 0; S.verts = (const float*)vv.p; S.tris = (const int32_t*)tt.p; S.soup = (const float*)sp.p; S.n_tris = mesh->n_tris;
    launch_render(P, S, (const float*)xx.p, 0, n, nullptr, (const float*)nd.p, (const uint8_t*)ng.p, (const uint8_t*)bm.p, (uint8_t*)oo.p, nullptr, nullptr, nullptr, nullptr, 0);
    TG_HIP(hipDeviceSynchronize());
    TG_HIP(hipMemcpy(out, oo.p, npix * n, hipMemcpyDeviceToHost));
    return 0;
}

int tg_render_tactile_heightfield(const tg_sensor* sen, int32_t rows, int32_t cols, double grid_scale, int32_t n, const double* heights,
                                  const float* zoff, const float* xf, uint8_t* out) {
    if (!sen || !heights || !zoff || !xf || !out) return fail(-1, "NULL argument");
    if (int rc = need_device()) return rc;
    const int H = sen->image_h, W = sen->image_w;
    if (!((H % 128 == 0 && W % 128 == 0) || (H == 64 && W == 64))) return fail(-1, "image size must be 64x64 or a multiple of 128");
    const size_t npix = (size_t)H * W, cells = (size_t)rows * cols;
    DevBuf nd, ng, bm, hh, zz, xx, oo;
    if (nd.alloc(npix * 4) || ng.alloc(npix * 4) || bm.alloc(npix) || hh.alloc(cells * n * 8) || zz.alloc((size_t)n * 4) || xx.alloc((size_t)n * 48) ||
        oo.alloc(npix * n))
        return fail(-2, "hipMalloc failed");
    TG_HIP(hipMemcpy(nd.p, sen->nodef_dep, npix * 4, hipMemcpyHostToDevice));
    { std::vector<uint8_t> g8(npix); make_gray_u8(sen->nodef_gray, (int)npix, g8.data()); TG_HIP(hipMemcpy(ng.p, g8.data(), npix, hipMemcpyHostToDevice)); }
    TG_HIP(hipMemcpy(bm.p, sen->border_mask, npix, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(hh.p, heights, cells * n * 8, hipMemcpyHostToDevice)); TG_HIP(hipMemcpy(zz.p, zoff, (size_t)n * 4, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(xx.p, xf, (size_t)n * 48, hipMemcpyHostToDevice));
    TG_HIP(hipMemset(oo.p, 0, npix * n));
    RasterParams P = make_raster_params(W, H, sen->fov_deg, sen->near_plane, sen->far_plane, sen->turn_off_border, sen->nodef_dep);
    Stimulus S{};
    S.kind = 1; S.heights = (const double*)hh.p; S.zoff = (const float*)zz.p; S.rows = rows; S.cols = cols; S.scale = (float)grid_scale;
    S.n_tris = (rows - 1) * (cols - 1) * 2;
    launch_render(P, S, (const float*)xx.p, 0, n, nullptr, (const float*)nd.p, (const uint8_t*)ng.p, (const uint8_t*)bm.p, (uint8_t*)oo.p, nullptr, nullptr, nullptr, nullptr, 0);
    TG_HIP(hipDeviceSynchronize());
    TG_HIP(hipMemcpy(out, oo.p, npix * n, hipMemcpyDeviceToHost));
    return 0;
}

int tg_gen_heightfield(int32_t n, const int64_t* seeds, int32_t rows, int32_t cols, double interp, double range, double* heights, float* zoff) {
    if (!seeds || !heights) return fail(-1, "NULL argument");
    if (int rc = need_device()) return rc;
    const size_t cells = (size_t)rows * cols;
    DevBuf sd, hh, zz;
    if (sd.alloc((size_t)n * 8) || hh.alloc(cells * n * 8) || zz.alloc((size_t)n * 4)) return fail(-2, "hipMalloc failed");
    TG_HIP(hipMemcpy(sd.p, seeds, (size_t)n * 8, hipMemcpyHostToDevice));
    launch_gen_surface(n, nullptr, (const int64_t*)sd.p, rows, cols, interp, range, 1, 0, (double*)hh.p, (float*)zz.p, 0);
    TG_HIP(hipDeviceSynchronize());
    TG_HIP(hipMemcpy(heights, hh.p, cells * n * 8, hipMemcpyDeviceToHost));
    if (zoff) TG_HIP(hipMemcpy(zoff, zz.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
