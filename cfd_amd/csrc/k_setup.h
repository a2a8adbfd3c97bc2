// k_setup.h -- set-up and read-back kernels: equilibrium fill, packed mask, solid rectangles.
// Part of the single translation unit lbm_hip.hip (included inside its anonymous namespace, after
// <hip/hip_runtime.h>, include/lbm_abi.h and d3q19.h); not a stand-alone header.
// ------------------------------------------------------------------------------------------
// set-up and read-back kernels
// ------------------------------------------------------------------------------------------
// k_init (lbm_solver.cu:37-64): equilibrium at rho = 1, u = 0 is f_k = w_k exactly.  Fills the
// whole plane (ghost layers and row padding too) of both buffers.
// Every float of the allocation [first, last) relative to plane 0 is filled (ghost layers, row
// padding, skew and the AA front guard too), with the weight of the population whose run it
// belongs to.
__global__ void k_fill_planes(float* __restrict__ fa, float* __restrict__ fb, Geo g, int rows_layout,
                              long long first, long long last) {
    const long long i = first + gidx();
    if (i >= last) return;
    long long k;
    if (rows_layout) {
        long long r = i % g.row;
        if (r < 0) r += g.row;
        k = r / g.Wp;
    } else {
        k = i < 0 ? 0 : i / g.plane;
    }
    const float w = weight((int)(k < Q ? k : Q - 1));
    fa[i] = w;
    fb[i] = w;
}

__global__ void k_fill_fields(float* __restrict__ rho, float4* __restrict__ u, long long n) {
    const long long i = gidx();
    if (i >= n) return;
    rho[i] = 1.0f;
    u[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// dense u8 mask (reference wire format) -> bit-packed words of the owned layers
__global__ void k_pack_mask(const unsigned char* __restrict__ dense, unsigned long long* __restrict__ mask, Geo g) {
    const long long wi = gidx();  // word index over owned rows
    const long long nwords = (long long)g.D * g.H * g.Wm;
    if (wi >= nwords) return;
    const long long row = wi / g.Wm;
    const int xw = (int)(wi - row * g.Wm);
    const unsigned char* src = dense + row * g.W + (long long)xw * 64;
    const int cnt = min(64, g.W - xw * 64);
    unsigned long long bits = 0;
    for (int b = 0; b < cnt; b++) bits |= (unsigned long long)(src[b] > 0) << b;
    mask[(long long)g.H * g.Wm + wi] = bits;  // skip the bottom ghost layer
}

__global__ void k_unpack_mask(const unsigned long long* __restrict__ mask, unsigned char* __restrict__ dense, Geo g) {
    const long long i = gidx();
    const long long n = (long long)g.W * g.H * g.D;
    if (i >= n) return;
    const long long row = i / g.W;
    const int x = (int)(i - row * g.W);
    const unsigned long long w = mask[(long long)g.H * g.Wm + row * g.Wm + (x >> 6)];
    dense[i] = ((w >> (x & 63)) & 1ull) ? 255 : 0;
}

// k_add_solid_rect (lbm_solver.cu:498-513) on the packed mask: one thread per word, no atomics.
// Coordinates are global; this context owns global layers [z0, z0+D).
__global__ void k_add_rect(unsigned long long* __restrict__ mask, Geo g, int rx, int ry, int rz, int rw, int rh, int rd) {
    const long long wi = gidx();
    const long long nwords = (long long)(g.D + 2) * g.H * g.Wm;  // ghost layers included
    if (wi >= nwords) return;
    const long long row = wi / g.Wm;
    const int xw = (int)(wi - row * g.Wm);
    const int zl = (int)(row / g.H);
    const int y = (int)(row - (long long)zl * g.H);
    const int zg = zl - 1 + g.z0;
    if (zg < 0 || zg >= g.Dg) return;
    // 64-bit compares: rx + rw may exceed int range for hostile arguments
    if (y < ry || (long long)y >= (long long)ry + rh || zg < rz || (long long)zg >= (long long)rz + rd) return;
    long long lo = max((long long)rx, (long long)xw * 64);
    long long hi = min(min((long long)rx + rw, (long long)g.W), (long long)xw * 64 + 64);
    if (hi <= lo) return;
    const int b0 = (int)(lo - (long long)xw * 64), cnt = (int)(hi - lo);
    const unsigned long long bits = (cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull)) << b0;
    mask[wi] |= bits;
}
