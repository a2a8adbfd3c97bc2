"""Oracle-backed slab engine for CPU tests of the multi-GPU protocol (cfd_amd/slab.py).

Implements the engine interface that slab.SlabSolver / LocalSlabCluster drive (step split, halo
pack/unpack with the documented message layout, static mask ghost layers) on top of
oracle.orc_step_slab, with torch CPU tensors as messages so gloo can carry them.  TEST ONLY."""
import numpy as np
import torch

from oracle import oracle as O

UP5 = [5, 11, 14, 15, 18]
DOWN5 = [6, 12, 13, 16, 17]


class OracleSlabEngine:
    def __init__(self, W, H, depth_global, z0, D, preset, two_streams=False, pairs=False):
        """two_streams: answer SlabSolver's two-stream schedule (stream calls are no-ops on the CPU).  pairs: offer
        the lbm_pair_* protocol (include/lbm_abi.h): a pair of steps as two split steps whose interior launches are
        deferred the way the HIP engine defers them -- the same two messages per face in the same order."""
        self.W, self.H, self.D, self.Dg, self.z0, self.preset = W, H, D, depth_global, z0, preset
        self.has_side = two_streams
        self.pairs = pairs
        self.pair_phase = 0
        self.log = []
        self.layer = W * H
        self.nl = self.layer * (D + 2)
        w = O.tables()["w"]
        self.f = [np.repeat(w, self.nl).astype(np.float32).reshape(19, D + 2, H, W).copy() for _ in range(2)]
        self.cur = 0
        self.solid = np.zeros((D + 2, H, W), np.uint8)
        self.rho = np.ones(self.layer * D, np.float32)
        self.u = np.zeros(4 * self.layer * D, np.float32)
        self.params = (np.float32(0.08), np.float32(1.98), np.float32(1.90))
        self.phase = 0

    def set_solid(self, m):
        self.solid[1:-1] = np.asarray(m, np.uint8).reshape(self.D, self.H, self.W)

    def _run(self, zb, ze):
        if ze <= zb:
            return
        fin, fout = self.f[self.cur], self.f[1 - self.cur]
        O.lib().orc_step_slab(O._fp(fin), O._fp(fout), O._fp(self.rho), O._fp(self.u), O._up(self.solid),
                              self.W, self.H, self.D, self.Dg, self.z0, *self.params, self.preset, zb, ze)

    def step_boundary(self, msg_bottom=None, msg_top=None):
        self._run(1, 2)
        if self.D > 1:
            self._run(self.D, self.D + 1)
        self.phase = 1
        if msg_bottom is not None:
            self.halo_pack(0, msg_bottom)
        if msg_top is not None:
            self.halo_pack(1, msg_top)

    def halo_unpack_both(self, from_below, from_above):
        if from_below is not None:
            self.halo_unpack(0, from_below)
        if from_above is not None:
            self.halo_unpack(1, from_above)

    def step_interior(self):
        self._run(2, self.D)
        self.phase = 2

    def step_finish(self):
        self.cur = 1 - self.cur
        self.phase = 0
        self.log.append("single")

    step_phase_interior = step_interior

    # ---- the two-stream schedule and the pair protocol, as cfd_amd/slab.py SlabSolver drives them ----
    def side_ctx(self):
        import contextlib
        return contextlib.nullcontext()

    def side_wait_main(self):
        pass

    def main_wait_side(self):
        pass

    def pair_available(self):
        return bool(self.pairs) and self.phase == 0 and self.pair_phase == 0

    def pair_boundary_first(self, msg_bottom=None, msg_top=None):
        assert self.pair_phase == 0 and self.phase == 0
        self.step_boundary(msg_bottom, msg_top)        # boundary layers t -> t+1, messages
        self.pair_phase = 1

    def pair_interior(self):
        assert self.pair_phase == 1
        self.step_interior()                           # (the HIP engine advances the interior by two steps here)
        self.pair_phase = 2

    def pair_boundary_second(self, msg_bottom=None, msg_top=None):
        assert self.pair_phase == 2                    # the first step's messages have been unpacked
        self.cur = 1 - self.cur
        self.phase = 0
        self.step_boundary(msg_bottom, msg_top)        # boundary layers t+1 -> t+2, messages
        self.pair_phase = 3

    def pair_finish(self):
        assert self.pair_phase == 3
        self.step_interior()
        self.cur = 1 - self.cur
        self.phase = 0
        self.pair_phase = 0
        self.log.append("pair")

    def _state(self):
        return self.f[1 - self.cur] if self.phase else self.f[self.cur]

    def message_floats(self):
        return 5 * self.layer + self.H + 2

    def new_message(self):
        return torch.zeros(self.message_floats(), dtype=torch.float32)

    def new_mask_layer(self):
        return torch.zeros(self.layer, dtype=torch.uint8)

    def halo_pack(self, side, msg):
        f = self._state()
        zl = self.D if side else 1
        m = msg.numpy()
        for j, k in enumerate(UP5 if side else DOWN5):
            m[j * self.layer:(j + 1) * self.layer] = f[k, zl].reshape(-1)
        t = m[5 * self.layer:]
        t[:] = 0
        if side:
            t[0] = f[10, zl, self.H - 1, self.W - 1]
            t[1] = f[9, zl, self.H - 1, self.W - 1]
            t[2:] = f[13, zl, :, self.W - 1]
        else:
            t[0] = f[8, zl, 0, self.W - 1]
            t[1] = f[9, zl, 0, self.W - 1]

    def halo_unpack(self, side, msg):
        f = self._state()
        zl = self.D + 1 if side else 0
        m = msg.numpy()
        for j, k in enumerate(DOWN5 if side else UP5):
            f[k, zl] = m[j * self.layer:(j + 1) * self.layer].reshape(self.H, self.W)
        t = m[5 * self.layer:]
        if side == 0:
            f[10, zl, self.H - 1, self.W - 1] = t[0]
            f[9, zl, self.H - 1, self.W - 1] = t[1]
            f[13, zl, :, self.W - 1] = t[2:]
        else:
            f[8, zl, 0, self.W - 1] = t[0]
            f[9, zl, 0, self.W - 1] = t[1]

    def get_mask_layer(self, side):
        return self.solid[self.D if side else 1].reshape(-1).copy()

    def set_mask_ghost(self, side, layer):
        self.solid[self.D + 1 if side else 0] = np.asarray(layer, np.uint8).reshape(self.H, self.W)

    def populations(self):
        return self.f[self.cur][:, 1:-1].reshape(19, -1).copy()
