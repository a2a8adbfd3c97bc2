// k_halo.h -- z-slab halo messages (A/B and AA) and mask ghost layers.
// Part of the single translation unit lbm_hip.hip (included inside its anonymous namespace, after
// <hip/hip_runtime.h>, include/lbm_abi.h and d3q19.h); not a stand-alone header.
// ---- z-slab halo messages -------------------------------------------------------------------
// message = 5 population layers (layer floats each) followed by `tail` floats of outlet-edge values:
//   top message   (side 1, from layer zl = D): planes 5,11,14,15,18; tail = f10(W-1,H-1), f9(W-1,H-1), f13(W-1,y) for all y
//   bottom message(side 0, from layer zl = 1): planes 6,12,13,16,17; tail = f8(W-1,0),  f9(W-1,0)
// The tails are what the unchecked outlet index of the reference (lbm_solver.cu:292-296) reads
// across a z face; they land in ghost-layer planes that the regular exchange does not use.
__device__ __forceinline__ int halo_plane(int side, int j) { return side ? UP5[j] : DOWN5[j]; }

__global__ void k_halo_pack(const float* __restrict__ f, float* __restrict__ msg, Geo g, int side) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    const int zl = side ? g.D : 1;
    if (j < 5) {
        if (i < g.mlayer) msg[(long long)j * g.mlayer + i] = f[(long long)halo_plane(side, j) * g.plane + cell_off(g, (int)(i % g.Wp), (int)(i / g.Wp), zl)];
        return;
    }
    float* tail = msg + 5 * g.mlayer;
    if (i >= g.H + 2) return;
    float v = 0.0f;
    if (side) {
        if (i == 0) v = f[10 * g.plane + cell_off(g, g.W - 1, g.H - 1, zl)];
        else if (i == 1) v = f[9 * g.plane + cell_off(g, g.W - 1, g.H - 1, zl)];
        else v = f[13 * g.plane + cell_off(g, g.W - 1, (int)i - 2, zl)];
    } else {
        if (i == 0) v = f[8 * g.plane + cell_off(g, g.W - 1, 0, zl)];
        else if (i == 1) v = f[9 * g.plane + cell_off(g, g.W - 1, 0, zl)];
    }
    tail[i] = v;
}

// side 0: message from the lower neighbour's TOP face -> bottom ghost layer (zl = 0)
// side 1: message from the upper neighbour's BOTTOM face -> top ghost layer (zl = D+1)
__global__ void k_halo_unpack(float* __restrict__ f, const float* __restrict__ msg, Geo g, int side) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    const int zl = side ? g.D + 1 : 0;
    if (j < 5) {
        // the message that arrives on our bottom face is the sender's top message (planes UP5)
        if (i < g.mlayer) f[(long long)halo_plane(1 - side, j) * g.plane + cell_off(g, (int)(i % g.Wp), (int)(i / g.Wp), zl)] = msg[(long long)j * g.mlayer + i];
        return;
    }
    const float* tail = msg + 5 * g.mlayer;
    if (side == 0) {
        if (i >= g.H + 2) return;
        if (i == 0) f[10 * g.plane + cell_off(g, g.W - 1, g.H - 1, zl)] = tail[0];
        else if (i == 1) f[9 * g.plane + cell_off(g, g.W - 1, g.H - 1, zl)] = tail[1];
        else f[13 * g.plane + cell_off(g, g.W - 1, (int)i - 2, zl)] = tail[i];
    } else {
        if (i == 0) f[8 * g.plane + cell_off(g, g.W - 1, 0, zl)] = tail[0];
        else if (i == 1) f[9 * g.plane + cell_off(g, g.W - 1, 0, zl)] = tail[1];
    }
}

__global__ void k_halo_unpack2(float* __restrict__ f, const float* __restrict__ from_below,
                               const float* __restrict__ from_above, Geo g, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(3);   // between the two exchanges of a pair: runs beside k_fused2, which the next boundary step waits behind
    const int side = blockIdx.y / 6;
    const float* msg = side ? from_above : from_below;
    if (!msg) return;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y % 6;
    const int zl = side ? g.D + 1 : 0;
    if (j < 5) {
        if (i < g.mlayer) f[(long long)halo_plane(1 - side, j) * g.plane + cell_off(g, (int)(i % g.Wp), (int)(i / g.Wp), zl)] = msg[(long long)j * g.mlayer + i];
        return;
    }
    const float* tail = msg + 5 * g.mlayer;
    if (side == 0) {
        if (i >= g.H + 2) return;
        if (i == 0) f[10 * g.plane + cell_off(g, g.W - 1, g.H - 1, zl)] = tail[0];
        else if (i == 1) f[9 * g.plane + cell_off(g, g.W - 1, g.H - 1, zl)] = tail[1];
        else f[13 * g.plane + cell_off(g, g.W - 1, (int)i - 2, zl)] = tail[i];
    } else {
        if (i == 0) f[8 * g.plane + cell_off(g, g.W - 1, 0, zl)] = tail[0];
        else if (i == 1) f[9 * g.plane + cell_off(g, g.W - 1, 0, zl)] = tail[1];
    }
}

// ---- AA z-slab halo ---------------------------------------------------------------------------
// Which slots carry the five populations that cross a face depends on the layout of the moment:
//   swapped  (after an EVEN step): F_k of a boundary-layer cell sits in the OWNED layer at plane
//            inv(k); it goes to the same plane of the neighbour's GHOST layer, where the
//            neighbour's odd step pulls it from;
//   streamed (after an ODD step, or after init / import): the cell has pushed F_k into OUR GHOST
//            layer at plane k; it goes to the neighbour's OWNED boundary layer, same plane, where
//            the neighbour's even step reads its own slots.
// Only cells x < W, 0 <= y < H travel: the x/y ghost shell of a layer belongs to the wall logic of
// the slab that wrote it.  Message = 5 layers of H*Wp floats + the 7 outlet-column planes of the
// boundary layer (7*H floats, what aa_outlet_read needs across the face, ring link included).
__global__ void k_aa_halo_pack(const float* __restrict__ M, const float* __restrict__ ob, float* __restrict__ msg,
                               Geo g, int side, int streamed) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (j < 5) {
        if (i >= g.mlayer) return;
        const int x = (int)(i % g.Wp), y = (int)(i / g.Wp);
        if (x >= g.W) return;
        const int zl = streamed ? (side ? g.D + 1 : 0) : (side ? g.D : 1);
        const int pl = streamed ? (side ? UP5[j] : DOWN5[j]) : (side ? DOWN5[j] : UP5[j]);
        msg[(long long)j * g.mlayer + i] = M[(long long)pl * g.plane + cell_off(g, x, y, zl)];
        return;
    }
    if (i >= (long long)OB_N * g.H) return;
    msg[5 * g.mlayer + i] = ob[ob_at(g, (int)(i / g.H), side ? g.D - 1 : 0, (int)(i % g.H))];
}
// side: the face of THIS slab the message arrived on (0: from the lower neighbour's top face)
__global__ void k_aa_halo_unpack(float* __restrict__ M, float* __restrict__ ob, const float* __restrict__ msg,
                                 Geo g, int side, int streamed) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (j < 5) {
        if (i >= g.mlayer) return;
        // the ring link of the wind-tunnel preset carries only the outlet column: beyond a domain end
        // there is no cell, and the ghost slots there belong to this slab's own wall logic
        if (side ? (g.z0 + g.D == g.Dg) : (g.z0 == 0)) return;
        const int x = (int)(i % g.Wp), y = (int)(i / g.Wp);
        if (x >= g.W) return;
        const int zl = streamed ? (side ? g.D : 1) : (side ? g.D + 1 : 0);
        const int pl = streamed ? (side ? DOWN5[j] : UP5[j]) : (side ? UP5[j] : DOWN5[j]);
        M[(long long)pl * g.plane + cell_off(g, x, y, zl)] = msg[(long long)j * g.mlayer + i];
        return;
    }
    if (i >= (long long)OB_N * g.H) return;
    ob[ob_at(g, (int)(i / g.H), side ? g.D : -1, (int)(i % g.H))] = msg[5 * g.mlayer + i];
}

// one layer of the packed mask <-> dense u8 (ghost exchange of the static mask)
__global__ void k_mask_layer_get(const unsigned long long* __restrict__ mask, unsigned char* __restrict__ dense, Geo g, int zl) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)g.W * g.H) return;
    const int y = (int)(i / g.W), x = (int)(i - (long long)y * g.W);
    dense[i] = solid_at(g, mask, x, y, zl) ? 255 : 0;
}
__global__ void k_mask_layer_set(unsigned long long* __restrict__ mask, const unsigned char* __restrict__ dense, Geo g, int zl) {
    const long long wi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= (long long)g.H * g.Wm) return;
    const int y = (int)(wi / g.Wm), xw = (int)(wi - (long long)y * g.Wm);
    const unsigned char* src = dense + (long long)y * g.W + (long long)xw * 64;
    const int cnt = min(64, g.W - xw * 64);
    unsigned long long bits = 0;
    for (int b = 0; b < cnt; b++) bits |= (unsigned long long)(src[b] > 0) << b;
    mask[(long long)zl * g.H * g.Wm + wi] = bits;
}

// k_compute_curl (lbm_solver.cu:612-656): |curl u| by clamped central differences.
