/*
 * lbm_oracle.c -- CPU ORACLE for the D3Q19 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file restates, in plain C, the algorithm of the reference solver
 * (1neskk/CFD, Engine/src/Physics/lbm_solver.cu and Engine/src/Utils/mesh_voxelizer.cpp).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 * the product path (cfd_amd/csrc) never links, imports or calls anything in oracle/.
 *
 * Pinning: the reference is CUDA-only (needs nvcc + cuda_runtime.h + glm, none of which exist
 * in the build image) and ships no golden vectors (Engine/tests/test_lbm_solver.cu is stale and
 * numeric-free), so it cannot be built here without writing header stand-ins, which this project
 * does not do.  The oracle is pinned instead against the known answers that the survey phase
 * recorded from the reference kernel text executed on the host (SURVEY.md Appendix A.1/A.2,
 * BASELINE.md section 4): mask hashes/counts for assets/cube.obj and field statistics after 10,
 * 20 and 100 steps.  tests/test_oracle_pins.py checks every one of them.
 *
 * All field arithmetic is fp32 in the reference's operation order; build with
 * -ffp-contract=off and without -ffast-math so every operation rounds once (IEEE).
 *
 * Reference citations are "file:line" into /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define Q 19

/* Lattice numbering contract: values of lbm_solver.cu:9-35. */
static const float W19[Q] = {
    1.0f / 3.0f,
    1.0f / 18.0f, 1.0f / 18.0f, 1.0f / 18.0f, 1.0f / 18.0f, 1.0f / 18.0f, 1.0f / 18.0f,
    1.0f / 36.0f, 1.0f / 36.0f, 1.0f / 36.0f, 1.0f / 36.0f,
    1.0f / 36.0f, 1.0f / 36.0f, 1.0f / 36.0f, 1.0f / 36.0f,
    1.0f / 36.0f, 1.0f / 36.0f, 1.0f / 36.0f, 1.0f / 36.0f};
static const int CX[Q] = {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 1, -1, 1, -1, 0, 0, 0, 0};
static const int CY[Q] = {0, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0, 1, -1, 1, -1};
static const int CZ[Q] = {0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, -1, -1, 1, 1, -1, -1, 1};
static const int INV[Q] = {0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15, 18, 17};
/* reflect_y[15..18] is a point inversion, not a mirror (quirk Q1, lbm_solver.cu:31-32). */
static const int RFY[Q] = {0, 1, 2, 4, 3, 5, 6, 9, 10, 7, 8, 11, 12, 13, 14, 16, 15, 18, 17};
static const int RFZ[Q] = {0, 1, 2, 3, 4, 6, 5, 7, 8, 9, 10, 13, 14, 11, 12, 17, 18, 15, 16};

/* Source codes of the boundary classification. */
enum { SRC_PULL = 0, SRC_INLET = 1, SRC_OUTLET = 2, SRC_FLOOR_BB = 3, SRC_CEIL_RFY = 4,
       SRC_SIDE_RFZ = 5, SRC_SOLID_BB = 6,
       /* cavity preset (own extension, not in the reference) */
       SRC_WALL_BB = 7, SRC_LID_BB = 8 };

enum { PRESET_WIND_TUNNEL = 0, PRESET_CAVITY = 1 };

void orc_tables(float* w, int* cx, int* cy, int* cz, int* inv, int* rfy, int* rfz) {
    for (int k = 0; k < Q; k++) {
        w[k] = W19[k]; cx[k] = CX[k]; cy[k] = CY[k]; cz[k] = CZ[k];
        inv[k] = INV[k]; rfy[k] = RFY[k]; rfz[k] = RFZ[k];
    }
}

/* ---- init: lbm_solver.cu:37-64 -------------------------------------------------------- */
void orc_init(float* f, float* rho, float* u4, int W, int H, int D) {
    const size_t n = (size_t)W * H * D;
    const float r = 1.0f, ux = 0.0f, uy = 0.0f, uz = 0.0f;
    const float usq = ux * ux + uy * uy + uz * uz;
    for (size_t i = 0; i < n; i++) {
        for (int k = 0; k < Q; k++) {
            float cu = CX[k] * ux + CY[k] * uy + CZ[k] * uz;
            f[(size_t)k * n + i] = W19[k] * r * (1.0f + 3.0f * cu + 4.5f * cu * cu - 1.5f * usq);
        }
        rho[i] = r;
        u4[4 * i + 0] = ux; u4[4 * i + 1] = uy; u4[4 * i + 2] = uz; u4[4 * i + 3] = 0.0f;
    }
}

/* ---- collision of one cell: lbm_solver.cu:326-401 (moments, stress, relaxation, rebuild) ---- */
static inline void collide_cell(const float* f, float rho, float ux, float uy, float uz,
                                float omega_shear, float omega_bulk, float* out) {
    float mxx = 0.0f, myy = 0.0f, mzz = 0.0f, mxy = 0.0f, myz = 0.0f, mzx = 0.0f;
    for (int k = 0; k < Q; k++) {
        float fk = f[k];
        mxx += fk * CX[k] * CX[k];
        myy += fk * CY[k] * CY[k];
        mzz += fk * CZ[k] * CZ[k];
        mxy += fk * CX[k] * CY[k];
        myz += fk * CY[k] * CZ[k];
        mzx += fk * CZ[k] * CX[k];
    }
    float Pxx = mxx - rho * ux * ux - rho * (1.0f / 3.0f);
    float Pyy = myy - rho * uy * uy - rho * (1.0f / 3.0f);
    float Pzz = mzz - rho * uz * uz - rho * (1.0f / 3.0f);
    float Pxy = mxy - rho * ux * uy;
    float Pyz = myz - rho * uy * uz;
    float Pzx = mzx - rho * uz * ux;

    float trace = Pxx + Pyy + Pzz;
    float third = trace * (1.0f / 3.0f);
    float corr = third * (omega_shear - omega_bulk);

    float Nxx = Pxx * (1.0f - omega_shear) + corr;
    float Nyy = Pyy * (1.0f - omega_shear) + corr;
    float Nzz = Pzz * (1.0f - omega_shear) + corr;
    float Nxy = Pxy * (1.0f - omega_shear);
    float Nyz = Pyz * (1.0f - omega_shear);
    float Nzx = Pzx * (1.0f - omega_shear);

    float usq = ux * ux + uy * uy + uz * uz;
    for (int k = 0; k < Q; k++) {
        float cu = CX[k] * ux + CY[k] * uy + CZ[k] * uz;
        float feq = W19[k] * rho * (1.0f + 3.0f * cu + 4.5f * cu * cu - 1.5f * usq);
        float Qxx = (float)(CX[k] * CX[k]) - 1.0f / 3.0f;
        float Qyy = (float)(CY[k] * CY[k]) - 1.0f / 3.0f;
        float Qzz = (float)(CZ[k] * CZ[k]) - 1.0f / 3.0f;
        float Qxy = (float)(CX[k] * CY[k]);
        float Qyz = (float)(CY[k] * CZ[k]);
        float Qzx = (float)(CZ[k] * CX[k]);
        float fneq = 4.5f * W19[k] *
                     (Qxx * Nxx + Qyy * Nyy + Qzz * Nzz +
                      2.0f * (Qxy * Nxy + Qyz * Nyz + Qzx * Nzx));
        out[k] = feq + fneq;
    }
}

/* One cell, literal form: moments (lbm_solver.cu:320-334) then collision.  For host-logic tests. */
void orc_collide_cell(const float* f, float omega_shear, float omega_bulk, float* out, float* rho_u4) {
    float rho = 0.0f, ux = 0.0f, uy = 0.0f, uz = 0.0f;
    for (int k = 0; k < Q; k++) {
        rho += f[k];
        ux += f[k] * CX[k];
        uy += f[k] * CY[k];
        uz += f[k] * CZ[k];
    }
    if (rho > 1e-9f) {
        float inv_rho = 1.0f / rho;
        ux *= inv_rho; uy *= inv_rho; uz *= inv_rho;
    } else {
        rho = 1.0f;
    }
    rho_u4[0] = rho; rho_u4[1] = ux; rho_u4[2] = uy; rho_u4[3] = uz;
    collide_cell(f, rho, ux, uy, uz, omega_shear, omega_bulk, out);
}

/* Lid term of the cavity preset (own extension): moving-wall half-way bounce-back with rho_w=1.
 * The product kernel uses the same single-rounded expression. */
static inline float lid_term(int k, float U) { return (6.0f * W19[k]) * ((float)CX[k] * U); }

/* ---- one reference step on a whole (undecomposed) domain --------------------------------
 * Literal restatement of k_stream_collide_cumulant, lbm_solver.cu:242-402, including the raw
 * linear index of the outlet branch (quirk Q2, :292-296).  Layout is the reference's:
 * f[k*n + z*W*H + y*W + x], solid u8[n], rho[n], u4[n][4].
 * preset 1 (cavity) swaps the boundary chain for a closed box with a lid at y=H-1 moving in +x.
 */
void orc_step(const float* f_in, float* f_out, float* rho_out, float* u_out,
              const unsigned char* solid, int W, int H, int D,
              float inlet_velocity, float omega_shear, float omega_bulk, int preset) {
    const long long n = (long long)W * H * D;
#pragma omp parallel for schedule(static)
    for (int z = 0; z < D; z++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                long long idx = (long long)z * W * H + (long long)y * W + x;
                if (solid[idx] > 0) { /* :258-267 */
                    rho_out[idx] = 0.0f;
                    u_out[4 * idx + 0] = 0.0f; u_out[4 * idx + 1] = 0.0f;
                    u_out[4 * idx + 2] = 0.0f; u_out[4 * idx + 3] = 0.0f;
                    for (int k = 0; k < Q; k++) f_out[k * n + idx] = f_in[k * n + idx];
                    continue;
                }
                float f[Q];
                float rho = 0.0f, ux = 0.0f, uy = 0.0f, uz = 0.0f;
                for (int k = 0; k < Q; k++) {
                    int nx = x - CX[k], ny = y - CY[k], nz = z - CZ[k];
                    if (preset == PRESET_CAVITY) {
                        if (ny >= H) {
                            f[k] = f_in[INV[k] * n + idx] + lid_term(k, inlet_velocity);
                        } else if (nx < 0 || nx >= W || ny < 0 || nz < 0 || nz >= D) {
                            f[k] = f_in[INV[k] * n + idx];
                        } else {
                            long long ni = (long long)nz * W * H + (long long)ny * W + nx;
                            f[k] = (solid[ni] > 0) ? f_in[INV[k] * n + idx] : f_in[k * n + ni];
                        }
                    } else if (nx < 0) { /* inlet :283-291 */
                        float r = 1.0f;
                        float uix = inlet_velocity, uiy = 0.0f, uiz = 0.0f;
                        float usq = uix * uix + uiy * uiy + uiz * uiz;
                        float cu = CX[k] * uix + CY[k] * uiy + CZ[k] * uiz;
                        f[k] = W19[k] * r * (1.0f + 3.0f * cu + 4.5f * cu * cu - 1.5f * usq);
                    } else if (nx >= W) { /* outlet :292-296, ny/nz unchecked (Q2) */
                        long long ni = (long long)nz * W * H + (long long)ny * W + (W - 1);
                        f[k] = f_in[k * n + ni];
                    } else if (ny < 0) { /* floor :297-300 */
                        f[k] = f_in[INV[k] * n + idx];
                    } else if (ny >= H) { /* ceiling :301-304 */
                        f[k] = f_in[RFY[k] * n + idx];
                    } else if (nz < 0 || nz >= D) { /* side walls :305-308 */
                        f[k] = f_in[RFZ[k] * n + idx];
                    } else { /* :309-318 */
                        long long ni = (long long)nz * W * H + (long long)ny * W + nx;
                        if (solid[ni] > 0) f[k] = f_in[INV[k] * n + idx];
                        else f[k] = f_in[k * n + ni];
                    }
                    rho += f[k];
                    ux += f[k] * CX[k];
                    uy += f[k] * CY[k];
                    uz += f[k] * CZ[k];
                }
                if (rho > 1e-9f) { /* :327-334 */
                    float inv_rho = 1.0f / rho;
                    ux *= inv_rho; uy *= inv_rho; uz *= inv_rho;
                } else {
                    rho = 1.0f;
                }
                rho_out[idx] = rho;
                u_out[4 * idx + 0] = ux; u_out[4 * idx + 1] = uy;
                u_out[4 * idx + 2] = uz; u_out[4 * idx + 3] = 0.0f;
                float o[Q];
                collide_cell(f, rho, ux, uy, uz, omega_shear, omega_bulk, o);
                for (int k = 0; k < Q; k++) f_out[k * n + idx] = o[k];
            }
}

/* ---- one step of a z-slab (test infrastructure for the multi-GPU protocol) -----------------
 * The slab owns global layers [z0, z0+D) of a domain Dg deep and stores D+2 layers: local layer
 * zl = 0 and zl = D+1 are ghost layers filled by the halo exchange (cfd_amd/slab.py).  Same chain
 * as orc_step; wall tests use global z; the outlet's unchecked index (Q2) is followed in logical
 * (plane, y, z) terms and a read that leaves the slab in z lands in the ghost layer, where the
 * ring exchange deposits the values it needs.  Arrays: f[k][(D+2)][H][W], solid[(D+2)][H][W],
 * rho[D][H][W], u[D][H][W][4].  Updates owned layers zl in [zl_begin, zl_end). */
void orc_step_slab(const float* f_in, float* f_out, float* rho_out, float* u_out,
                   const unsigned char* solid, int W, int H, int D, int Dg, int z0,
                   float inlet_velocity, float omega_shear, float omega_bulk, int preset,
                   int zl_begin, int zl_end) {
    const long long layer = (long long)W * H;
    const long long n = layer * (D + 2);
#define SOFF(x, y, zl) ((long long)(zl) * layer + (long long)(y) * W + (x))
    for (int zl = zl_begin; zl < zl_end; zl++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const long long idx = SOFF(x, y, zl);
                const long long ridx = idx - layer;
                const int zg = z0 + zl - 1;
                if (solid[idx] > 0) {
                    rho_out[ridx] = 0.0f;
                    u_out[4 * ridx + 0] = u_out[4 * ridx + 1] = u_out[4 * ridx + 2] = u_out[4 * ridx + 3] = 0.0f;
                    for (int k = 0; k < Q; k++) f_out[k * n + idx] = f_in[k * n + idx];
                    continue;
                }
                float f[Q];
                float rho = 0.0f, ux = 0.0f, uy = 0.0f, uz = 0.0f;
                for (int k = 0; k < Q; k++) {
                    int nx = x - CX[k], ny = y - CY[k], nzg = zg - CZ[k], nzl = zl - CZ[k];
                    if (preset == PRESET_CAVITY) {
                        if (ny >= H) f[k] = f_in[INV[k] * n + idx] + lid_term(k, inlet_velocity);
                        else if (nx < 0 || nx >= W || ny < 0 || nzg < 0 || nzg >= Dg) f[k] = f_in[INV[k] * n + idx];
                        else f[k] = (solid[SOFF(nx, ny, nzl)] > 0) ? f_in[INV[k] * n + idx] : f_in[k * n + SOFF(nx, ny, nzl)];
                    } else if (nx < 0) {
                        float usq = inlet_velocity * inlet_velocity + 0.0f * 0.0f + 0.0f * 0.0f;
                        float cu = CX[k] * inlet_velocity + CY[k] * 0.0f + CZ[k] * 0.0f;
                        f[k] = W19[k] * 1.0f * (1.0f + 3.0f * cu + 4.5f * cu * cu - 1.5f * usq);
                    } else if (nx >= W) {
                        int pl = k, qy = ny, qzg = nzg;
                        if (qy < 0) { qy += H; qzg -= 1; } else if (qy >= H) { qy -= H; qzg += 1; }
                        if (qzg < 0) pl -= 1; else if (qzg >= Dg) pl += 1;   /* layer stays in the ghost */
                        f[k] = f_in[pl * n + SOFF(W - 1, qy, qzg - z0 + 1)];
                    } else if (ny < 0) f[k] = f_in[INV[k] * n + idx];
                    else if (ny >= H) f[k] = f_in[RFY[k] * n + idx];
                    else if (nzg < 0 || nzg >= Dg) f[k] = f_in[RFZ[k] * n + idx];
                    else f[k] = (solid[SOFF(nx, ny, nzl)] > 0) ? f_in[INV[k] * n + idx] : f_in[k * n + SOFF(nx, ny, nzl)];
                    rho += f[k];
                    ux += f[k] * CX[k];
                    uy += f[k] * CY[k];
                    uz += f[k] * CZ[k];
                }
                if (rho > 1e-9f) { float ir = 1.0f / rho; ux *= ir; uy *= ir; uz *= ir; } else rho = 1.0f;
                rho_out[ridx] = rho;
                u_out[4 * ridx + 0] = ux; u_out[4 * ridx + 1] = uy; u_out[4 * ridx + 2] = uz; u_out[4 * ridx + 3] = 0.0f;
                float o[Q];
                collide_cell(f, rho, ux, uy, uz, omega_shear, omega_bulk, o);
                for (int k = 0; k < Q; k++) f_out[k * n + idx] = o[k];
            }
#undef SOFF
}

/* Convenience: N steps with A/B swap (lbm_solver.cu:477-494).  On return f_a holds the
 * current populations (an odd number of swaps is undone by a copy). */
void orc_run(float* f_a, float* f_b, float* rho, float* u4, const unsigned char* solid,
             int W, int H, int D, float U, float ws, float wb, int preset, int nsteps) {
    float *cur = f_a, *nxt = f_b;
    for (int s = 0; s < nsteps; s++) {
        orc_step(cur, nxt, rho, u4, solid, W, H, D, U, ws, wb, preset);
        float* t = cur; cur = nxt; nxt = t;
    }
    if (cur != f_a) {
        size_t bytes = (size_t)Q * W * H * D * sizeof(float);
        /* swap contents so that f_a = current, f_b = previous */
        float* tmp = (float*)malloc(bytes);
        memcpy(tmp, f_a, bytes); memcpy(f_a, f_b, bytes); memcpy(f_b, tmp, bytes);
        free(tmp);
    }
}

/* ---- boundary classification: which branch feeds f[k] ------------ */
void orc_classify(const unsigned char* solid, int W, int H, int D, int preset,
                  unsigned char* cls /* [n][19] */, unsigned char* is_solid /* [n] */) {
    for (int z = 0; z < D; z++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                size_t idx = (size_t)z * W * H + (size_t)y * W + x;
                is_solid[idx] = solid[idx] > 0;
                for (int k = 0; k < Q; k++) {
                    int nx = x - CX[k], ny = y - CY[k], nz = z - CZ[k];
                    unsigned char c;
                    if (solid[idx] > 0) c = SRC_PULL; /* solid cells never stream */
                    else if (preset == PRESET_CAVITY) {
                        if (ny >= H) c = SRC_LID_BB;
                        else if (nx < 0 || nx >= W || ny < 0 || nz < 0 || nz >= D) c = SRC_WALL_BB;
                        else c = solid[(size_t)nz * W * H + (size_t)ny * W + nx] > 0 ? SRC_SOLID_BB : SRC_PULL;
                    }
                    else if (nx < 0) c = SRC_INLET;
                    else if (nx >= W) c = SRC_OUTLET;
                    else if (ny < 0) c = SRC_FLOOR_BB;
                    else if (ny >= H) c = SRC_CEIL_RFY;
                    else if (nz < 0 || nz >= D) c = SRC_SIDE_RFZ;
                    else c = solid[(size_t)nz * W * H + (size_t)ny * W + nx] > 0 ? SRC_SOLID_BB : SRC_PULL;
                    cls[idx * Q + k] = c;
                }
            }
}

/* ---- curl magnitude: lbm_solver.cu:612-656 ---------------------------------------------- */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void orc_curl(const float* u4, float* curl, int W, int H, int D) {
#define UAT(ix, iy, iz, c) \
    u4[4 * ((size_t)clampi(iz, 0, D - 1) * W * H + (size_t)clampi(iy, 0, H - 1) * W + clampi(ix, 0, W - 1)) + (c)]
#pragma omp parallel for schedule(static)
    for (int z = 0; z < D; z++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                float dv_dx = (UAT(x + 1, y, z, 1) - UAT(x - 1, y, z, 1)) * 0.5f;
                float dw_dx = (UAT(x + 1, y, z, 2) - UAT(x - 1, y, z, 2)) * 0.5f;
                float du_dy = (UAT(x, y + 1, z, 0) - UAT(x, y - 1, z, 0)) * 0.5f;
                float dw_dy = (UAT(x, y + 1, z, 2) - UAT(x, y - 1, z, 2)) * 0.5f;
                float du_dz = (UAT(x, y, z + 1, 0) - UAT(x, y, z - 1, 0)) * 0.5f;
                float dv_dz = (UAT(x, y, z + 1, 1) - UAT(x, y, z - 1, 1)) * 0.5f;
                float cxv = dw_dy - dv_dz;
                float cyv = du_dz - dw_dx;
                float czv = dv_dx - du_dy;
                curl[(size_t)z * W * H + (size_t)y * W + x] = sqrtf(cxv * cxv + cyv * cyv + czv * czv);
            }
#undef UAT
}

/* ---- solid rect: lbm_solver.cu:498-513 ---------------------------------------------------- */
void orc_add_solid_rect(unsigned char* solid, int W, int H, int D,
                        int rx, int ry, int rz, int rw, int rh, int rd) {
    for (int z = 0; z < D; z++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++)
                if (x >= rx && x < rx + rw && y >= ry && y < ry + rh && z >= rz && z < rz + rd)
                    solid[(size_t)z * W * H + (size_t)y * W + x] = 255;
}

/* =========================================================================================
 * Mesh path: OBJ -> triangles -> rotate_y(-90) -> auto-scale -> +X ray-cast voxelisation.
 * Follows mesh_voxelizer.cpp:20-246 and lbm_solver.cu:521-556.  The OBJ reader restates the
 * behaviour of the vendored tinyobjloader that matters here: `v` and `f` records, 1-based and
 * negative indices, `a/b/c` index triplets, triangles kept, quads split by the shorter
 * diagonal with ties going to [0,1,3],[1,2,3] (tiny_obj_loader.h:1520-1616).  Faces with more
 * than four vertices are ear-clipped the way the loader's built-in triangulation does it
 * (tiny_obj_loader.h:1751-1973; the reference does not define TINYOBJLOADER_USE_MAPBOX_EARCUT).
 * ========================================================================================= */
typedef struct { float x, y, z; } v3;
typedef struct { v3 a, b, c; } tri_t;

typedef struct {
    tri_t* tris; size_t ntris, cap;
    v3 bmin, bmax;
} orc_mesh;

static v3 v3sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
/* glm::cross: (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y) */
static v3 v3cross(v3 a, v3 b) {
    v3 r = {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
    return r;
}
static float v3dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 v3min(v3 a, v3 b) { v3 r = {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; return r; }
static v3 v3max(v3 a, v3 b) { v3 r = {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; return r; }

orc_mesh* orc_mesh_new(void) {
    orc_mesh* m = (orc_mesh*)calloc(1, sizeof(orc_mesh));
    m->bmin.x = m->bmin.y = m->bmin.z = FLT_MAX;
    m->bmax.x = m->bmax.y = m->bmax.z = -FLT_MAX;
    return m;
}
void orc_mesh_free(orc_mesh* m) { if (m) { free(m->tris); free(m); } }
size_t orc_mesh_ntris(const orc_mesh* m) { return m->ntris; }
void orc_mesh_bounds(const orc_mesh* m, float* out6) {
    out6[0] = m->bmin.x; out6[1] = m->bmin.y; out6[2] = m->bmin.z;
    out6[3] = m->bmax.x; out6[4] = m->bmax.y; out6[5] = m->bmax.z;
}
void orc_mesh_get_tris(const orc_mesh* m, float* out /* ntris*9 */) {
    memcpy(out, m->tris, m->ntris * sizeof(tri_t));
}

static void mesh_push(orc_mesh* m, v3 a, v3 b, v3 c) {
    if (m->ntris == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 64;
        m->tris = (tri_t*)realloc(m->tris, m->cap * sizeof(tri_t));
    }
    tri_t t = {a, b, c};
    m->tris[m->ntris++] = t;
    /* bounds: mesh_voxelizer.cpp:69-70 */
    m->bmin = v3min(m->bmin, a); m->bmax = v3max(m->bmax, a);
    m->bmin = v3min(m->bmin, b); m->bmax = v3max(m->bmax, b);
    m->bmin = v3min(m->bmin, c); m->bmax = v3max(m->bmax, c);
}

/* point in polygon, tiny_obj_loader.h:1449-1461 (W. R. Franklin's pnpoly) */
static int orc_pnpoly(int nvert, const float* vertx, const float* verty, float testx, float testy) {
    int i, j, c = 0;
    for (i = 0, j = nvert - 1; i < nvert; j = i++) {
        if (((verty[i] > testy) != (verty[j] > testy)) &&
            (testx < (vertx[j] - vertx[i]) * (testy - verty[i]) / (verty[j] - verty[i]) + vertx[i]))
            c = !c;
    }
    return c;
}
static float v3axis(v3 p, int axis) { return axis == 0 ? p.x : (axis == 1 ? p.y : p.z); }

/* Faces with more than four vertices: tiny_obj_loader.h:1751-1973.  The polygon is projected on the two axes
 * picked from its first non-degenerate corner (:1756-1808); then ears are cut off: starting at `guess`, a
 * vertex triple is an ear when its turn has the sign of the "area" term (:1859-1877, the loader's own
 * formula) and no other remaining vertex lies inside it (:1879-1911); the ear is emitted, its middle vertex
 * removed (:1913-1943), and the search gives up after one fruitless round (:1816-1840).  The last three
 * vertices make the final triangle (:1948-1973). */
static void orc_ear_clip(orc_mesh* m, const v3* verts, const long* face, int n) {
    int axes[2] = {1, 2};
    for (int k = 0; k < n; k++) {
        v3 p0 = verts[face[k % n]], p1 = verts[face[(k + 1) % n]], p2 = verts[face[(k + 2) % n]];
        float e0x = p1.x - p0.x, e0y = p1.y - p0.y, e0z = p1.z - p0.z;
        float e1x = p2.x - p1.x, e1y = p2.y - p1.y, e1z = p2.z - p1.z;
        float cx = fabsf(e0y * e1z - e0z * e1y);
        float cy = fabsf(e0z * e1x - e0x * e1z);
        float cz = fabsf(e0x * e1y - e0y * e1x);
        const float eps = FLT_EPSILON;
        if (cx > eps || cy > eps || cz > eps) {
            if (!(cx > cy && cx > cz)) {
                axes[0] = 0;
                if (cz > cx && cz > cy) axes[1] = 1;
            }
            break;
        }
    }
    long rem[64];
    int nrem = n;
    for (int i = 0; i < n; i++) rem[i] = face[i];
    size_t guess = 0;
    size_t remaining_iterations = (size_t)n, previous_remaining = (size_t)n;
    while (nrem > 3 && remaining_iterations > 0) {
        size_t np = (size_t)nrem;
        if (guess >= np) guess -= np;
        if (previous_remaining != np) { previous_remaining = np; remaining_iterations = np; }
        else remaining_iterations--;
        long ind[3];
        float vx[3], vy[3];
        for (int k = 0; k < 3; k++) {
            ind[k] = rem[(guess + (size_t)k) % np];
            vx[k] = v3axis(verts[ind[k]], axes[0]);
            vy[k] = v3axis(verts[ind[k]], axes[1]);
        }
        float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0];
        float e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
        float cross = e0x * e1y - e0y * e1x;
        float area = (vx[0] * vy[1] - vy[0] * vx[1]) * 0.5f;
        if (cross * area < 0.0f) { guess += 1; continue; }
        int overlap = 0;
        for (size_t other = 3; other < np; other++) {
            size_t idx = (guess + other) % np;
            float tx = v3axis(verts[rem[idx]], axes[0]), ty = v3axis(verts[rem[idx]], axes[1]);
            if (orc_pnpoly(3, vx, vy, tx, ty)) { overlap = 1; break; }
        }
        if (overlap) { guess += 1; continue; }
        mesh_push(m, verts[ind[0]], verts[ind[1]], verts[ind[2]]);
        size_t removed = (guess + 1) % np;
        while (removed + 1 < np) { rem[removed] = rem[removed + 1]; removed++; }
        nrem--;
    }
    if (nrem == 3) mesh_push(m, verts[rem[0]], verts[rem[1]], verts[rem[2]]);
}

/* returns 0 on success (mesh_voxelizer.cpp:20-86) */
int orc_mesh_load_obj(orc_mesh* m, const char* path) {
    FILE* fp = fopen(path, "r");
    if (!fp) return -1;
    size_t nv = 0, vcap = 1024;
    v3* verts = (v3*)malloc(vcap * sizeof(v3));
    char line[4096];
    m->ntris = 0;
    m->bmin.x = m->bmin.y = m->bmin.z = FLT_MAX;
    m->bmax.x = m->bmax.y = m->bmax.z = -FLT_MAX;
    while (fgets(line, sizeof line, fp)) {
        char* p = line;
        while (*p == ' ' || *p == '\t') p++;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            double x = 0, y = 0, z = 0;
            char* q = p + 1;
            x = strtod(q, &q); y = strtod(q, &q); z = strtod(q, &q);
            if (nv == vcap) { vcap *= 2; verts = (v3*)realloc(verts, vcap * sizeof(v3)); }
            verts[nv].x = (float)x; verts[nv].y = (float)y; verts[nv].z = (float)z;
            nv++;
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            long ids[64]; int cnt = 0;
            char* q = p + 1;
            for (;;) {
                while (*q == ' ' || *q == '\t') q++;
                if (*q == '\0' || *q == '\n' || *q == '\r' || *q == '#') break;
                char* e;
                long vi = strtol(q, &e, 10);
                if (e == q) break;
                while (*e && *e != ' ' && *e != '\t' && *e != '\n' && *e != '\r') e++; /* skip /vt/vn */
                q = e;
                long zero = vi > 0 ? vi - 1 : (long)nv + vi; /* negative = relative */
                if (cnt < 64) ids[cnt++] = zero;
            }
            if (cnt < 3) continue;
            int ok = 1;
            for (int i = 0; i < cnt; i++) if (ids[i] < 0 || (size_t)ids[i] >= nv) ok = 0;
            if (!ok) continue;
            if (cnt == 3) {
                mesh_push(m, verts[ids[0]], verts[ids[1]], verts[ids[2]]);
            } else if (cnt == 4) {
                v3 p0 = verts[ids[0]], p1 = verts[ids[1]], p2 = verts[ids[2]], p3 = verts[ids[3]];
                float e02x = p2.x - p0.x, e02y = p2.y - p0.y, e02z = p2.z - p0.z;
                float e13x = p3.x - p1.x, e13y = p3.y - p1.y, e13z = p3.z - p1.z;
                float s02 = e02x * e02x + e02y * e02y + e02z * e02z;
                float s13 = e13x * e13x + e13y * e13y + e13z * e13z;
                if (s02 < s13) { mesh_push(m, p0, p1, p2); mesh_push(m, p0, p2, p3); }
                else           { mesh_push(m, p0, p1, p3); mesh_push(m, p1, p2, p3); }
            } else {
                orc_ear_clip(m, verts, ids, cnt);
            }
        }
    }
    free(verts);
    fclose(fp);
    return 0;
}

/* mesh_voxelizer.cpp:218-246; glm::radians(d) = d * 0.01745329251994329576923690768489f */
void orc_mesh_rotate_y(orc_mesh* m, float angle_degrees) {
    float rad = angle_degrees * 0.01745329251994329576923690768489f;
    float c = cosf(rad), s = sinf(rad);
    m->bmin.x = m->bmin.y = m->bmin.z = FLT_MAX;
    m->bmax.x = m->bmax.y = m->bmax.z = -FLT_MAX;
    for (size_t i = 0; i < m->ntris; i++) {
        v3* vs[3] = {&m->tris[i].a, &m->tris[i].b, &m->tris[i].c};
        for (int j = 0; j < 3; j++) {
            float x = vs[j]->x * c + vs[j]->z * s;
            float z = -vs[j]->x * s + vs[j]->z * c;
            vs[j]->x = x; vs[j]->z = z;
        }
        for (int j = 0; j < 3; j++) { m->bmin = v3min(m->bmin, *vs[j]); }
        for (int j = 0; j < 3; j++) { m->bmax = v3max(m->bmax, *vs[j]); }
    }
}

/* Moller-Trumbore, mesh_voxelizer.cpp:89-121 */
static int ray_tri(v3 o, v3 d, const tri_t* t, float* tout) {
    const float EPS = 0.0000001f;
    v3 e1 = v3sub(t->b, t->a), e2 = v3sub(t->c, t->a);
    v3 h = v3cross(d, e2);
    float a = v3dot(e1, h);
    if (a > -EPS && a < EPS) return 0;
    float f = 1.0f / a;
    v3 s = v3sub(o, t->a);
    float u = f * v3dot(s, h);
    if (u < 0.0f || u > 1.0f) return 0;
    v3 q = v3cross(s, e1);
    float v = f * v3dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return 0;
    float tt = f * v3dot(e2, q);
    *tout = tt;
    return tt > EPS;
}

static int cmp_float(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

static float min3(float a, float b, float c) { float m = a < b ? a : b; return m < c ? m : c; }
static float max3(float a, float b, float c) { float m = a > b ? a : b; return m > c ? m : c; }

/* mesh_voxelizer.cpp:146-216 */
void orc_mesh_voxelize(const orc_mesh* m, unsigned char* buf, int W, int H, int D, float scale) {
    v3 gc = {W / 2.0f, H / 2.0f, D / 2.0f};
    v3 mc;
    mc.x = (m->bmin.x + m->bmax.x) / 2;
    mc.y = m->bmin.y + gc.y / scale;
    mc.z = (m->bmin.z + m->bmax.z) / 2;
    size_t nt = m->ntris;
    float* tb = (float*)malloc((nt ? nt : 1) * 4 * sizeof(float));
    for (size_t i = 0; i < nt; i++) {
        const tri_t* t = &m->tris[i];
        tb[4 * i + 0] = min3(t->a.y, t->b.y, t->c.y);
        tb[4 * i + 1] = max3(t->a.y, t->b.y, t->c.y);
        tb[4 * i + 2] = min3(t->a.z, t->b.z, t->c.z);
        tb[4 * i + 3] = max3(t->a.z, t->b.z, t->c.z);
    }
#pragma omp parallel for collapse(2) schedule(dynamic, 8)
    for (int z = 0; z < D; z++)
        for (int y = 0; y < H; y++) {
            v3 gs = {0.0f, (float)y, (float)z};
            v3 o;
            o.x = (gs.x - gc.x) / scale + mc.x;
            o.y = (gs.y - gc.y) / scale + mc.y;
            o.z = (gs.z - gc.z) / scale + mc.z;
            v3 dir = {1.0f, 0.0f, 0.0f};
            size_t cap = 16, cnt = 0;
            float* hits = (float*)malloc(cap * sizeof(float));
            for (size_t i = 0; i < nt; i++) {
                if (o.y < tb[4 * i + 0] || o.y > tb[4 * i + 1] ||
                    o.z < tb[4 * i + 2] || o.z > tb[4 * i + 3]) continue;
                float t;
                if (ray_tri(o, dir, &m->tris[i], &t)) {
                    if (cnt == cap) { cap *= 2; hits = (float*)realloc(hits, cap * sizeof(float)); }
                    hits[cnt++] = t;
                }
            }
            qsort(hits, cnt, sizeof(float), cmp_float);
            for (size_t i = 0; i + 1 < cnt; i += 2) {
                int xs = (int)ceilf(hits[i] * scale);
                int xe = (int)floorf(hits[i + 1] * scale);
                if (xs < 0) xs = 0;
                if (xe > W) xe = W;
                for (int x = xs; x < xe; x++)
                    buf[(size_t)z * W * H + (size_t)y * W + x] = 255;
            }
            free(hits);
        }
    free(tb);
}

/* lbm_solver.cu:521-556: load, rotate -90 about y, uniform 0.8 fill scale, voxelise into a
 * zeroed mask, then force the y=0 plane solid.  `floor_plane`=0 leaves the floor out (used to
 * check the survey's "before floor" hashes).  Returns the scale via *scale_out. */
int orc_load_mesh(const char* path, unsigned char* solid, int W, int H, int D,
                  int floor_plane, float* scale_out) {
    orc_mesh* m = orc_mesh_new();
    int rc = orc_mesh_load_obj(m, path); /* failure only logs in the reference (:522-524) */
    memset(solid, 0, (size_t)W * H * D);
    orc_mesh_rotate_y(m, -90.0f);
    float sx = m->bmax.x - m->bmin.x, sy = m->bmax.y - m->bmin.y, sz = m->bmax.z - m->bmin.z;
    float scx = ((float)W * 0.8f) / sx;
    float scy = ((float)H * 0.8f) / sy;
    float scz = ((float)D * 0.8f) / sz;
    float scale = scx < scy ? scx : scy;
    if (scz < scale) scale = scz;
    if (scale_out) *scale_out = scale;
    orc_mesh_voxelize(m, solid, W, H, D, scale);
    if (floor_plane)
        for (int z = 0; z < D; z++)
            for (int x = 0; x < W; x++) solid[(size_t)z * W * H + x] = 255;
    orc_mesh_free(m);
    return rc;
}

/* FNV-1a-64 over a byte buffer (the survey's mask fingerprint). */
unsigned long long orc_fnv1a64(const unsigned char* p, size_t nbytes) {
    unsigned long long h = 1469598103934665603ULL;
    for (size_t i = 0; i < nbytes; i++) { h ^= p[i]; h *= 1099511628211ULL; }
    return h;
}
