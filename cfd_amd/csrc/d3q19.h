// d3q19.h -- lattice numbering contract and the per-cell arithmetic of the hot path.
//
// The tables carry the VALUES of the reference's __constant__ arrays
// (Engine/src/Physics/lbm_solver.cu:9-35); every consumer of the populations (read-back,
// halo exchange, boundary classification) depends on this numbering.
//
// collide_cell() is the reference's moment / stress / relaxation / rebuild sequence
// (lbm_solver.cu:339-401) with every multiplication by a lattice constant (0, +1, -1) folded
// away by hand.  Each remaining operation is the same fp32 operation, on the same operands, in
// the same order as the reference evaluates it, so with contraction off the result is
// bit-identical to the literal loop form:
//   * x*1 = x, x*(-1) = -x and a + (-b) = a - b are exact in IEEE arithmetic;
//   * a term f*0 is +-0 and adding it to a running sum never changes the sum's value;
//   * directions k and inv(k) have cu_k = -cu_inv(k) exactly, share (4.5*cu)*cu and share the
//     whole non-equilibrium part (Q_ab is even in c).
// tests/test_abi_and_host.py::test_folded_collision_is_bit_exact checks this bit for bit against the
// literal oracle.
#pragma once

#if defined(__HIPCC__)
#define LBM_HD __host__ __device__ __forceinline__
#else
#define LBM_HD inline
#endif

namespace d3q19 {

constexpr int Q = 19;

// k :  0  1  2  3  4  5  6  7  8  9 10 11 12 13 14 15 16 17 18
constexpr int CX[Q] = {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 1, -1, 1, -1, 0, 0, 0, 0};
constexpr int CY[Q] = {0, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0, 1, -1, 1, -1};
constexpr int CZ[Q] = {0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, -1, -1, 1, 1, -1, -1, 1};
constexpr int INV[Q] = {0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15, 18, 17};
// reflect_y keeps the reference's point inversion of the yz diagonals (lbm_solver.cu:31-32).
constexpr int RFY[Q] = {0, 1, 2, 4, 3, 5, 6, 9, 10, 7, 8, 11, 12, 13, 14, 16, 15, 18, 17};
constexpr int RFZ[Q] = {0, 1, 2, 3, 4, 6, 5, 7, 8, 9, 10, 13, 14, 11, 12, 17, 18, 15, 16};

// Populations that cross a z face (used by the slab halo exchange).
constexpr int UP5[5] = {5, 11, 14, 15, 18};    // cz = +1
constexpr int DOWN5[5] = {6, 12, 13, 16, 17};  // cz = -1

constexpr float W0 = 1.0f / 3.0f;
constexpr float W1 = 1.0f / 18.0f;
constexpr float W2 = 1.0f / 36.0f;
LBM_HD constexpr float weight(int k) { return k == 0 ? W0 : (k < 7 ? W1 : W2); }

// Boundary-classification codes (include/lbm_abi.h).
enum Src : unsigned char {
    SRC_PULL = 0, SRC_INLET = 1, SRC_OUTLET = 2, SRC_FLOOR_BB = 3, SRC_CEIL_RFY = 4,
    SRC_SIDE_RFZ = 5, SRC_SOLID_BB = 6, SRC_WALL_BB = 7, SRC_LID_BB = 8
};

// Bracket of the inlet equilibrium (lbm_solver.cu:285-291 with r = 1, u = (U,0,0)): every
// inlet-fed direction has cu = U, so f_k = w_k * inlet_bracket(U).
LBM_HD float inlet_bracket(float U) {
    const float usq = U * U;
    return 1.0f + 3.0f * U + 4.5f * U * U - 1.5f * usq;
}

// Moving-lid term of the cavity preset (not in the reference): (6 w_k) * (cx_k * U).
LBM_HD float lid_term(int k, float U) { return (6.0f * weight(k)) * ((float)CX[k] * U); }

struct Moments {
    float rho, ux, uy, uz;
};

// lbm_solver.cu:320-334: zeroth and first moments in k order, then the 1/rho normalisation.
LBM_HD Moments moments_of(const float* f) {
    Moments m;
    float rho = f[0];
    rho += f[1]; rho += f[2]; rho += f[3]; rho += f[4]; rho += f[5]; rho += f[6];
    rho += f[7]; rho += f[8]; rho += f[9]; rho += f[10]; rho += f[11]; rho += f[12];
    rho += f[13]; rho += f[14]; rho += f[15]; rho += f[16]; rho += f[17]; rho += f[18];
    float ux = f[1];
    ux -= f[2]; ux += f[7]; ux -= f[8]; ux += f[9]; ux -= f[10];
    ux += f[11]; ux -= f[12]; ux += f[13]; ux -= f[14];
    float uy = f[3];
    uy -= f[4]; uy += f[7]; uy -= f[8]; uy -= f[9]; uy += f[10];
    uy += f[15]; uy -= f[16]; uy += f[17]; uy -= f[18];
    float uz = f[5];
    uz -= f[6]; uz += f[11]; uz -= f[12]; uz -= f[13]; uz += f[14];
    uz += f[15]; uz -= f[16]; uz -= f[17]; uz += f[18];
    if (rho > 1e-9f) {
        const float inv_rho = 1.0f / rho;
        ux *= inv_rho; uy *= inv_rho; uz *= inv_rho;
    } else {
        rho = 1.0f;  // reference "safety" branch: momentum left un-normalised
    }
    m.rho = rho; m.ux = ux; m.uy = uy; m.uz = uz;
    return m;
}

// lbm_solver.cu:339-401.  f: streamed populations, m: their moments; out: post-collision.
LBM_HD void collide_cell(const float* f, const Moments& m, float omega_shear, float omega_bulk,
                         float* out) {
    const float rho = m.rho, ux = m.ux, uy = m.uy, uz = m.uz;
    // raw second moments, k order
    float mxx = f[1]; mxx += f[2]; mxx += f[7]; mxx += f[8]; mxx += f[9]; mxx += f[10];
    mxx += f[11]; mxx += f[12]; mxx += f[13]; mxx += f[14];
    float myy = f[3]; myy += f[4]; myy += f[7]; myy += f[8]; myy += f[9]; myy += f[10];
    myy += f[15]; myy += f[16]; myy += f[17]; myy += f[18];
    float mzz = f[5]; mzz += f[6]; mzz += f[11]; mzz += f[12]; mzz += f[13]; mzz += f[14];
    mzz += f[15]; mzz += f[16]; mzz += f[17]; mzz += f[18];
    float mxy = f[7]; mxy += f[8]; mxy -= f[9]; mxy -= f[10];
    float myz = f[15]; myz += f[16]; myz -= f[17]; myz -= f[18];
    float mzx = f[11]; mzx += f[12]; mzx -= f[13]; mzx -= f[14];

    const float third = 1.0f / 3.0f;
    const float r3 = rho * third;
    const float rux = rho * ux, ruy = rho * uy, ruz = rho * uz;
    const float Pxx = mxx - rux * ux - r3;
    const float Pyy = myy - ruy * uy - r3;
    const float Pzz = mzz - ruz * uz - r3;
    const float Pxy = mxy - rux * uy;
    const float Pyz = myz - ruy * uz;
    const float Pzx = mzx - ruz * ux;

    const float trace = Pxx + Pyy + Pzz;
    const float corr = (trace * third) * (omega_shear - omega_bulk);
    const float oms = 1.0f - omega_shear;
    const float Nxx = Pxx * oms + corr;
    const float Nyy = Pyy * oms + corr;
    const float Nzz = Pzz * oms + corr;
    const float Nxy = Pxy * oms;
    const float Nyz = Pyz * oms;
    const float Nzx = Pzx * oms;

    // Q_aa = c_a^2 - 1/3 takes two values
    const float QA = 1.0f - third;
    const float QB = 0.0f - third;
    const float Ax = QA * Nxx, Ay = QA * Nyy, Az = QA * Nzz;
    const float Bx = QB * Nxx, By = QB * Nyy, Bz = QB * Nzz;
    const float bxby = Bx + By, axby = Ax + By, bxay = Bx + Ay;
    const float d000 = bxby + Bz;        // k = 0
    const float d100 = axby + Bz;        // k = 1,2
    const float d010 = bxay + Bz;        // k = 3,4
    const float d001 = bxby + Az;        // k = 5,6
    const float d110 = (Ax + Ay) + Bz;   // k = 7..10
    const float d101 = axby + Az;        // k = 11..14
    const float d011 = bxay + Az;        // k = 15..18
    const float txy = 2.0f * Nxy, tyz = 2.0f * Nyz, tzx = 2.0f * Nzx;

    const float c0 = 4.5f * W0, c1 = 4.5f * W1, c2 = 4.5f * W2;
    const float w0r = W0 * rho, w1r = W1 * rho, w2r = W2 * rho;
    const float usq = ux * ux + uy * uy + uz * uz;
    const float c15 = 1.5f * usq;

    out[0] = w0r * (1.0f - c15) + c0 * d000;

#if defined(__HIP_DEVICE_COMPILE__) && !defined(LBM_NO_PACKED_PAIRS)
    // Directions k and inv(k) run through the same five operations on (+a, -a): written on two-float vectors
    // so that they issue as packed instructions (v_pk_add_f32 / v_pk_mul_f32, two results per issue slot).
    // Each component is the same IEEE operation sequence as the scalar form below: 1 + a, + b, - c15, * w rho,
    // + fneq, and 1 - a = 1 + (-a) exactly.
    typedef float lbm_f2 __attribute__((ext_vector_type(2)));
#define LBM_PAIR(KP, KM, CU, WR, CW, NEQ)                         \
    {                                                             \
        const float cu = (CU);                                    \
        const float a = 3.0f * cu;                                \
        const float b = 4.5f * cu * cu;                           \
        const float fneq = (CW) * (NEQ);                          \
        lbm_f2 t = {a, -a};                                       \
        t = 1.0f + t;                                             \
        t = t + b;                                                \
        t = t - c15;                                              \
        t = (WR) * t;                                             \
        t = t + fneq;                                             \
        out[KP] = t.x;                                            \
        out[KM] = t.y;                                            \
    }
#else
#define LBM_PAIR(KP, KM, CU, WR, CW, NEQ)                         \
    {                                                             \
        const float cu = (CU);                                    \
        const float a = 3.0f * cu;                                \
        const float b = 4.5f * cu * cu;                           \
        const float fneq = (CW) * (NEQ);                          \
        out[KP] = (WR) * (1.0f + a + b - c15) + fneq;             \
        out[KM] = (WR) * (1.0f - a + b - c15) + fneq;             \
    }
#endif
    LBM_PAIR(1, 2, ux, w1r, c1, d100)
    LBM_PAIR(3, 4, uy, w1r, c1, d010)
    LBM_PAIR(5, 6, uz, w1r, c1, d001)
    LBM_PAIR(7, 8, ux + uy, w2r, c2, d110 + txy)
    LBM_PAIR(9, 10, ux - uy, w2r, c2, d110 - txy)
    LBM_PAIR(11, 12, ux + uz, w2r, c2, d101 + tzx)
    LBM_PAIR(13, 14, ux - uz, w2r, c2, d101 - tzx)
    LBM_PAIR(15, 16, uy + uz, w2r, c2, d011 + tyz)
    LBM_PAIR(17, 18, uy - uz, w2r, c2, d011 - tyz)
#undef LBM_PAIR
}

#if defined(__HIPCC__)
// moments_of() + collide_cell() in one piece, written on two-float vectors wherever two quantities run through
// the same operation sequence: (ux, uy), (mxx, myy), (uz, mzz), (mzx, myz), the direction pairs (k, inv k) and
// the pairs of such pairs that share |c|.  Every component is the same chain of IEEE operations on the same
// operands in the same order as in the scalar functions above (a - b is a + (-b), 1 - a is 1 + (-a)), so the
// results are bit-identical; gfx950 issues a two-float operation (v_pk_add_f32 / v_pk_mul_f32) in the slot of
// one scalar operation, and reads either half of any register pair for either component, so broadcasts and
// sign flips cost nothing.  Used by k_fused2, which is bound by instruction issue; tests/test_fused_gpu.py
// compares it with the oracle and with the one-step kernels (which keep the scalar functions).
typedef float lbm_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void collide_packed(const float* f, float omega_shear, float omega_bulk, float* out) {
    float rho = f[0];
    rho += f[1]; rho += f[2]; rho += f[3]; rho += f[4]; rho += f[5]; rho += f[6];
    rho += f[7]; rho += f[8]; rho += f[9]; rho += f[10]; rho += f[11]; rho += f[12];
    rho += f[13]; rho += f[14]; rho += f[15]; rho += f[16]; rho += f[17]; rho += f[18];
    const lbm_v2 p13 = {f[1], f[3]}, p24 = {f[2], f[4]};
    const lbm_v2 pa = {f[11], f[15]}, pb = {f[12], f[16]}, pc = {f[13], f[17]}, pd = {f[14], f[18]};
    const lbm_v2 s7 = {f[7], f[7]}, s8 = {f[8], f[8]}, s9 = {f[9], f[9]}, s10 = {f[10], f[10]};
    lbm_v2 U = p13;                       // (ux, uy): lbm_solver.cu:320-334, k order per component
    U -= p24; U += s7; U -= s8; U += lbm_v2{f[9], -f[9]}; U += lbm_v2{-f[10], f[10]};
    U += pa; U -= pb; U += pc; U -= pd;
    lbm_v2 M = p13;                       // (mxx, myy)
    M += p24; M += s7; M += s8; M += s9; M += s10; M += pa; M += pb; M += pc; M += pd;
    lbm_v2 Z = {f[5], f[5]};              // (uz, mzz)
    Z += lbm_v2{-f[6], f[6]}; Z += lbm_v2{f[11], f[11]}; Z += lbm_v2{-f[12], f[12]};
    Z += lbm_v2{-f[13], f[13]}; Z += lbm_v2{f[14], f[14]}; Z += lbm_v2{f[15], f[15]};
    Z += lbm_v2{-f[16], f[16]}; Z += lbm_v2{-f[17], f[17]}; Z += lbm_v2{f[18], f[18]};
    lbm_v2 D = pa;                        // (mzx, myz)
    D += pb; D -= pc; D -= pd;
    float mxy = f[7]; mxy += f[8]; mxy -= f[9]; mxy -= f[10];
    float uz = Z.x;
    const float mzz = Z.y;
    if (rho > 1e-9f) {
        const float inv_rho = 1.0f / rho;
        U *= inv_rho; uz *= inv_rho;
    } else {
        rho = 1.0f;
    }

    const float third = 1.0f / 3.0f;
    const float r3 = rho * third;
    const lbm_v2 RU = rho * U;
    const float ruz = rho * uz;
    lbm_v2 P = M - RU * U;                // (Pxx, Pyy)
    P = P - r3;
    const float Pzz = mzz - ruz * uz - r3;
    const float Pxy = mxy - RU.x * U.y;
    const lbm_v2 Pq = D - lbm_v2{ruz * U.x, RU.y * uz};   // (Pzx, Pyz)

    const float trace = P.x + P.y + Pzz;
    const float corr = (trace * third) * (omega_shear - omega_bulk);
    const float oms = 1.0f - omega_shear;
    const lbm_v2 N = P * oms + corr;      // (Nxx, Nyy)
    const float Nzz = Pzz * oms + corr;
    const float txy = 2.0f * (Pxy * oms);
    const lbm_v2 T = 2.0f * (Pq * oms);   // (tzx, tyz)

    const float QA = 1.0f - third;
    const float QB = 0.0f - third;
    const lbm_v2 A = QA * N, B = QB * N;
    const float Az = QA * Nzz, Bz = QB * Nzz;
    const float bxby = B.x + B.y;
    const lbm_v2 ab = {A.x + B.y, B.x + A.y};             // (axby, bxay)
    const float d000 = bxby + Bz;
    const lbm_v2 d1 = ab + Bz;            // (d100, d010)
    const float d001 = bxby + Az;
    const float d110 = (A.x + A.y) + Bz;
    const lbm_v2 d2 = ab + Az;            // (d101, d011)

    const float c0 = 4.5f * W0, c1 = 4.5f * W1, c2 = 4.5f * W2;
    const float w0r = W0 * rho, w1r = W1 * rho, w2r = W2 * rho;
    const lbm_v2 UU = U * U;
    const float usq = UU.x + UU.y + uz * uz;
    const float c15 = 1.5f * usq;

    out[0] = w0r * (1.0f - c15) + c0 * d000;

// (3 cu, -3 cu) as cu * (3, -3): the negated product is the product negated, exactly
    const lbm_v2 pm3 = {3.0f, -3.0f};
#define LBM_PAIR2(KP, KM, CU, B_, WR, NEQ)   \
    {                                        \
        lbm_v2 t = lbm_v2{(CU), (CU)} * pm3; \
        t = 1.0f + t;                        \
        t = t + (B_);                        \
        t = t - c15;                         \
        t = (WR) * t;                        \
        t = t + (NEQ);                       \
        out[KP] = t.x;                       \
        out[KM] = t.y;                       \
    }
    {   // k = 1..4: cu = (ux, uy)
        const lbm_v2 b = 4.5f * U * U, q = c1 * d1;
        LBM_PAIR2(1, 2, U.x, b.x, w1r, q.x)
        LBM_PAIR2(3, 4, U.y, b.y, w1r, q.y)
    }
    {   // k = 5, 6
        const float b = 4.5f * uz * uz, q = c1 * d001;
        LBM_PAIR2(5, 6, uz, b, w1r, q)
    }
    const lbm_v2 zpm = lbm_v2{uz, uz} * lbm_v2{1.0f, -1.0f};   // (uz, -uz)
    {   // k = 7..10: cu = ux +- uy
        const lbm_v2 cu = lbm_v2{U.x, U.x} + lbm_v2{U.y, U.y} * lbm_v2{1.0f, -1.0f};
        const lbm_v2 b = 4.5f * cu * cu, q = c2 * (lbm_v2{d110, d110} + lbm_v2{txy, txy} * lbm_v2{1.0f, -1.0f});
        LBM_PAIR2(7, 8, cu.x, b.x, w2r, q.x)
        LBM_PAIR2(9, 10, cu.y, b.y, w2r, q.y)
    }
    const lbm_v2 Tx = lbm_v2{T.x, T.x} * lbm_v2{1.0f, -1.0f}, Ty = lbm_v2{T.y, T.y} * lbm_v2{1.0f, -1.0f};
    {   // k = 11..14: cu = ux +- uz
        const lbm_v2 cu = lbm_v2{U.x, U.x} + zpm;
        const lbm_v2 b = 4.5f * cu * cu, q = c2 * (lbm_v2{d2.x, d2.x} + Tx);
        LBM_PAIR2(11, 12, cu.x, b.x, w2r, q.x)
        LBM_PAIR2(13, 14, cu.y, b.y, w2r, q.y)
    }
    {   // k = 15..18: cu = uy +- uz
        const lbm_v2 cu = lbm_v2{U.y, U.y} + zpm;
        const lbm_v2 b = 4.5f * cu * cu, q = c2 * (lbm_v2{d2.y, d2.y} + Ty);
        LBM_PAIR2(15, 16, cu.x, b.x, w2r, q.x)
        LBM_PAIR2(17, 18, cu.y, b.y, w2r, q.y)
    }
#undef LBM_PAIR2
}
#endif

}  // namespace d3q19
