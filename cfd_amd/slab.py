"""z-slab decomposition of the D3Q19 path across the GPUs of one node.

The reference is single-GPU (no NCCL/MPI call sites), so this layer has no counterpart there; it
keeps the reference's arithmetic per cell, which makes an N-slab run bit-identical to the 1-GPU
run.  One process per GPU; the halo messages travel by RCCL send/recv through torch.distributed
(backend "nccl" is RCCL on ROCm), overlapped with the interior update:

    boundary layers -> pack 2 messages -> isend/irecv (RCCL stream)  ||  interior layers
    -> wait -> unpack into ghost layers -> A/B swap

The engine (HIP context, or an oracle-backed stand-in in the CPU tests) and the transport (RCCL
through torch.distributed, gloo in CPU tests, or a local copy between slabs that live in one
process) are both pluggable; the schedule and the neighbour topology live here.
"""
import os

import numpy as np
import torch

from . import lbm

SIDE_BOTTOM, SIDE_TOP = 0, 1


def slab_range(depth_global, world, rank):
    """Even split of [0, depth_global) into `world` contiguous slabs; the first `rem` get one more."""
    base, rem = divmod(depth_global, world)
    z0 = rank * base + min(rank, rem)
    return z0, base + (1 if rank < rem else 0)


def balanced_ranges(depth_global, ranges, times, min_layers=8):
    """Re-cut the depth into len(ranges) contiguous slabs of equal COST.  `ranges` = [(z0, layers)] of a run whose
    slabs took `times` per step each (any unit, measured with no waiting on neighbours in it): the cost of a layer
    is taken as its slab's time / layers, and the new cuts put an equal share of the summed cost into every slab
    (at least `min_layers` each).  A wind tunnel around an obstacle is not uniform in z: of the eight 64-layer
    slabs of BASELINE.json configs[3] the two that hold the car's flanks step 20 % slower than the four through
    its middle (more solid cells: less to update), and the slowest slab sets the step of all.
    Returns [(z0, layers)]; any partition gives the same fields bit for bit (tests/test_slab_gpu.py)."""
    P = len(ranges)
    if P != len(times) or P < 1 or sum(d for _, d in ranges) != depth_global or min_layers * P > depth_global:
        raise ValueError("balanced_ranges: ranges / times do not describe a partition of the depth")
    cost = np.zeros(depth_global)
    for (z0, d), t in zip(ranges, times):
        cost[z0:z0 + d] = float(t) / d
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    cuts = [0]
    for k in range(1, P):
        z = int(np.searchsorted(cum, cum[-1] * k / P))
        if z > 0 and abs(cum[z - 1] - cum[-1] * k / P) < abs(cum[z] - cum[-1] * k / P):
            z -= 1
        z = max(z, cuts[-1] + min_layers)
        z = min(z, depth_global - (P - k) * min_layers)
        cuts.append(z)
    cuts.append(depth_global)
    return [(cuts[k], cuts[k + 1] - cuts[k]) for k in range(P)]


def neighbours(preset, world, rank):
    """(down, up) ranks or None.  The wind-tunnel preset exchanges around a ring because the
    reference's unchecked outlet index wraps from the last z layer to the first (quirk Q2,
    lbm_solver.cu:292-296); the cavity preset is an open chain."""
    if world == 1:
        return (0, 0) if preset == lbm.PRESET_WIND_TUNNEL else (None, None)
    if preset == lbm.PRESET_WIND_TUNNEL:
        return (rank - 1) % world, (rank + 1) % world
    return (rank - 1 if rank > 0 else None), (rank + 1 if rank < world - 1 else None)


class HipSlabEngine:
    """One HIP context owning global layers [z0, z0+D)."""

    def __init__(self, W, H, depth_global, z0, D, preset, device=-1, two_streams=False, scheme=lbm.SCHEME_AB):
        self.device = torch.device("cuda", torch.cuda.current_device() if device < 0 else device)
        # The context runs on torch streams of its own (non-null handles: handing over the legacy
        # default stream, handle 0, would make the library create a private stream that
        # torch.distributed knows nothing about).  `stream` carries the interior update and the
        # unpack; `bstream` carries the boundary layers, and the driver makes it torch's current
        # stream around the exchange so RCCL orders itself behind them -- the boundary layers and the
        # messages then run BESIDE the interior kernel, not before it.  Both streams have the SAME
        # priority: a high-priority side stream deadlocked in the event waits between the two streams
        # on this ROCm, so that mode no longer exists.
        self.stream = torch.cuda.Stream(device=self.device)
        two_streams = os.environ.get("LBM_SLAB_STREAMS", "2" if two_streams else "1") == "2"
        self.bstream = None
        if two_streams:
            self.bstream = torch.cuda.Stream(device=self.device)
        self.solver = lbm.Solver(W, H, D, preset=preset, device=self.device.index, z0=z0,
                                 depth_global=depth_global, stream=self.stream.cuda_stream, scheme=scheme,
                                 boundary_stream=self.bstream.cuda_stream if two_streams else None)
        self._lib = self.solver._lib
        self._ctx = self.solver._ctx
        self.W, self.H, self.D, self.z0 = W, H, D, z0

    def stream_ctx(self):
        return torch.cuda.stream(self.stream)

    # the two-stream schedule as SlabSolver sees it (CPU engines in the tests answer the same calls with no-ops)
    @property
    def has_side(self):
        return self.bstream is not None

    def side_ctx(self):
        """launches and transport calls inside run on / are ordered behind the boundary stream"""
        return torch.cuda.stream(self.bstream)

    def side_wait_main(self):
        self.bstream.wait_stream(self.stream)

    def main_wait_side(self):
        self.stream.wait_stream(self.bstream)

    def set_streams(self, main, side=None):
        """Rebind the context to other torch streams (see SlabSolver.autotune_streams)."""
        torch.cuda.synchronize(self.device)
        self.solver._chk(self._lib.lbm_set_streams(self._ctx, main.cuda_stream,
                                                    None if side is None else side.cuda_stream))
        self.stream, self.bstream = main, side

    def message_floats(self):
        import ctypes as C
        n = C.c_size_t(0)
        self.solver._chk(self._lib.lbm_halo_message_floats(self._ctx, C.byref(n)))
        return n.value

    def new_message(self):
        with self.stream_ctx():
            return torch.zeros(self.message_floats(), dtype=torch.float32, device=self.device)

    def new_mask_layer(self):
        with self.stream_ctx():
            return torch.zeros(self.W * self.H, dtype=torch.uint8, device=self.device)

    def halo_pack(self, side, msg):
        self._check_message(msg)
        self.solver._chk(self._lib.lbm_halo_pack(self._ctx, side, msg.data_ptr()))

    def halo_unpack(self, side, msg):
        self._check_message(msg)
        self.solver._chk(self._lib.lbm_halo_unpack(self._ctx, side, msg.data_ptr()))

    def get_mask_layer(self, side):
        import ctypes as C
        a = np.empty(self.W * self.H, np.uint8)
        self.solver._chk(self._lib.lbm_halo_get_mask_layer(
            self._ctx, side, a.ctypes.data_as(C.POINTER(C.c_uint8)), a.size))
        return a

    def set_mask_ghost(self, side, layer):
        import ctypes as C
        a = np.ascontiguousarray(layer, np.uint8)
        self.solver._chk(self._lib.lbm_halo_set_mask_ghost(
            self._ctx, side, a.ctypes.data_as(C.POINTER(C.c_uint8)), a.size))

    def _check_message(self, msg):
        """The library takes raw device pointers and cannot see a buffer's size: a message that is too
        short, on another device or not fp32 would be written out of bounds by the boundary kernel."""
        if msg is None:
            return
        if msg.dtype != torch.float32 or msg.device != self.device or not msg.is_contiguous() \
                or msg.numel() < self.message_floats():
            raise ValueError(f"halo message must be a contiguous float32 tensor of >= {self.message_floats()} "
                             f"elements on {self.device} (got {msg.dtype}, {msg.numel()} on {msg.device})")

    def step_boundary(self, msg_bottom=None, msg_top=None):
        """boundary layers; the outgoing messages are written by the same launch"""
        self._check_message(msg_bottom)
        self._check_message(msg_top)
        self.solver._chk(self._lib.lbm_step_boundary_layers(
            self._ctx, None if msg_bottom is None else msg_bottom.data_ptr(),
            None if msg_top is None else msg_top.data_ptr()))

    def halo_unpack_both(self, from_below, from_above):
        self._check_message(from_below)
        self._check_message(from_above)
        self.solver._chk(self._lib.lbm_halo_unpack_both(
            self._ctx, None if from_below is None else from_below.data_ptr(),
            None if from_above is None else from_above.data_ptr()))

    def step_interior(self):
        self.solver._chk(self._lib.lbm_step_interior_layers(self._ctx))

    # a PAIR of steps (include/lbm_abi.h lbm_pair_*): fused interior, boundary layers and rim stepped twice
    def pair_available(self):
        import ctypes as C
        a = C.c_int(0)
        self.solver._chk(self._lib.lbm_pair_available(self._ctx, C.byref(a)))
        return bool(a.value)

    def pair_boundary_first(self, msg_bottom=None, msg_top=None):
        self._check_message(msg_bottom)
        self._check_message(msg_top)
        self.solver._chk(self._lib.lbm_pair_boundary_first(
            self._ctx, None if msg_bottom is None else msg_bottom.data_ptr(),
            None if msg_top is None else msg_top.data_ptr()))

    def pair_interior(self):
        self.solver._chk(self._lib.lbm_pair_interior(self._ctx))

    def pair_boundary_second(self, msg_bottom=None, msg_top=None):
        self._check_message(msg_bottom)
        self._check_message(msg_top)
        self.solver._chk(self._lib.lbm_pair_boundary_second(
            self._ctx, None if msg_bottom is None else msg_bottom.data_ptr(),
            None if msg_top is None else msg_top.data_ptr()))

    def pair_finish(self):
        self.solver._chk(self._lib.lbm_pair_finish(self._ctx))

    step_phase_interior = step_interior

    def step_finish(self):
        self.solver._chk(self._lib.lbm_step_finish(self._ctx))


class LoopbackP2P:
    """Transport of a one-rank ring (every message goes to the sender itself) as device copies on
    torch's current stream: what RCCL's send/recv to self does, without a communicator.  Lets one GPU
    run SlabSolver's real schedule -- the two-stream ordering included -- in tests."""

    @staticmethod
    def exchange(items):
        for send, dst, recv, src in items:
            if dst is not None and src is not None:
                recv.copy_(send, non_blocking=True)
        return []

    @staticmethod
    def wait(handle):
        pass


class NoExchange:
    """A transport that moves nothing: the same schedule and launches with stale ghost layers.  For timing what
    the exchange itself adds to a step (bench.py: exposed communication)."""

    @staticmethod
    def exchange(items):
        return []

    @staticmethod
    def wait(handle):
        pass


class TorchP2P:
    """Halo transport over torch.distributed point-to-point ops (RCCL on GPUs, gloo on CPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group

    def exchange(self, items):
        """items: list of (send_tensor, dst_rank, recv_tensor, src_rank); entries may have None
        ranks.  Sends are issued in list order and so are receives, so two messages between the
        same pair of ranks (2-rank ring) match up in order.  Returns a handle for wait()."""
        dist = self.dist
        ops = []
        for send, dst, _, _ in items:
            if dst is not None:
                ops.append(dist.P2POp(dist.isend, send, dst, self.group))
        for _, _, recv, src in items:
            if src is not None:
                ops.append(dist.P2POp(dist.irecv, recv, src, self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    @staticmethod
    def wait(handle):
        for w in handle:
            w.wait()


class HostStagedP2P(TorchP2P):
    """TorchP2P with every message staged through host memory: for process groups whose backend cannot take device
    tensors (gloo).  Lets several ranks that SHARE one GPU run the N > 1 code path of bench.py and the tests -- RCCL
    refuses two ranks on one device.  A rehearsal transport: it synchronises the stream it is called on."""

    def exchange(self, items):
        torch.cuda.current_stream().synchronize()      # the pack kernels that filled the messages
        host, back = [], []
        for send, dst, recv, src in items:
            hs = send.cpu() if dst is not None else None
            hr = torch.empty(recv.shape, dtype=recv.dtype) if src is not None else None
            host.append((hs, dst, hr, src))
            if hr is not None:
                back.append((hr, recv))
        return super().exchange(host), back

    @staticmethod
    def wait(handle):
        works, back = handle
        for w in works:
            w.wait()
        for hr, recv in back:
            recv.copy_(hr)                             # on the current stream, ahead of the unpack kernels


class _DevicePointer:
    """a raw device allocation as torch sees it (torch.as_tensor reads __cuda_array_interface__)"""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class IpcCopyP2P:
    """Halo transport without kernels of its own: the sender copies its message straight into the receiver's buffer
    (mapped through an IPC memory handle; a peer copy over xGMI between GPUs), and the two ranks of a link keep in step
    through stream-ordered flags in fine-grained device memory that the neighbours map as well (lbm_ipc_* /
    lbm_flag_post / lbm_flag_await, include/lbm_abi.h).  Why: RCCL's send / recv kernel needs CUs, waits for them beside
    k_fused2 and costs a slab step 10-12 % (profiles/r03_tuning.md section 8); a copy and two one-lane kernels do not.

    bind(links, nfloats, device) is collective and all-or-nothing: it allocates this rank's receive buffers (one per item
    with a source) and flags, trades the IPC handles, maps the neighbours' and returns the receive buffers as torch
    tensors for SlabSolver to use -- or None on EVERY rank if any step failed on any rank; the transport then passes
    everything to `fallback` (TorchP2P over RCCL, HostStagedP2P under gloo), as it always does for exchanges into other
    tensors (the mask layers).  Every rank owns the flags it WAITS on: arrived[i] (item i of a neighbour has landed in my
    recv_i) and free[i] (the receiver of my send_i has consumed it).  Per exchange number q, on torch's current stream,
    in this order:
        post   free[i] of src's block = q - 1   everything before this call on the stream has read my receive buffers
        await  my free[i] >= q - 1              the receiver is done with what I sent last time
        copy   send_i -> the receiver's recv_i;  post arrived[i] of dst's block = q
        await  my arrived[i] >= q               my neighbours' messages have landed
    Every rank posts before it waits, so no cycle of waits exists.  An await gives up after `timeout_ms` and raises the
    rank's timed-out word instead of hanging the GPU: check timed_out() after a run.  A one-rank ring (dst == me)
    copies inside the process with the same flags.  Item i of the sender lands in item i of the receiver, as with the
    ordered send / recv pairs of TorchP2P."""

    LINE = 16                                          # 32-bit words per flag: a 64-byte line each

    def __init__(self, fallback, dist, rank, world, timeout_ms=10000):
        self.fallback, self.dist, self.rank, self.world = fallback, dist, rank, world
        self.timeout_ms = timeout_ms
        self._lib = lbm.load_library()
        self.seq = 0
        self.links = None                              # [(dst, src)] per item once bound
        self.recv, self._recv_ptr = [], []             # my receive buffers: torch views, device addresses
        self.remote = []                               # where my item i goes: device address at dst (or my own)
        self._opened = []
        self.block = None                              # my flags (device address)
        self.peer_block = {}                           # rank -> device address of its flags here
        self.error = None

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self._lib.lbm_last_error(None).decode()}")

    def _flag(self, rank, item, kind):
        """device address of a flag in `rank`'s block: kind 0 arrived, 1 free"""
        base = self.block if rank == self.rank else self.peer_block[rank]
        return base + 4 * self.LINE * (item * 2 + kind)

    def _gather(self, mine):
        if self.world == 1:
            return [mine]
        everyone = [None] * self.world
        self.dist.all_gather_object(everyone, mine)
        return everyone

    def bind(self, links, nfloats, device):
        import ctypes as C
        lib = self._lib
        n = len(links)
        self.nbytes = 4 * nfloats
        mine = None
        try:                                           # (a rank that fails still takes part in every collective below)
            with torch.cuda.device(device):
                blk, handle = C.c_void_p(), C.create_string_buffer(64)
                self._chk(lib.lbm_ipc_alloc(4 * self.LINE * (2 * n + 1), 1, C.byref(blk), handle), "lbm_ipc_alloc (flags)")
                self.block = blk.value
                self._timed_out = self.block + 4 * self.LINE * 2 * n
                mine = {"flags": handle.raw, "recv": [None] * n}
                self._recv_ptr = [None] * n
                for i, (_, src) in enumerate(links):
                    if src is not None:
                        p, h = C.c_void_p(), C.create_string_buffer(64)
                        self._chk(lib.lbm_ipc_alloc(self.nbytes, 0, C.byref(p), h), "lbm_ipc_alloc (message)")
                        self._recv_ptr[i] = p.value
                        mine["recv"][i] = h.raw
        except Exception as e:                         # noqa: BLE001
            self.error, mine = e, None
        everyone = self._gather(mine)
        ok = all(x is not None for x in everyone)
        if ok:
            try:
                with torch.cuda.device(device):
                    for r in sorted({r for pair in links for r in pair if r is not None and r != self.rank}):
                        p = C.c_void_p()
                        self._chk(lib.lbm_ipc_open(everyone[r]["flags"], C.byref(p)), "lbm_ipc_open (flags)")
                        self.peer_block[r] = p.value
                        self._opened.append(p.value)
                    self.remote = [None] * n
                    for i, (dst, _) in enumerate(links):
                        if dst == self.rank:
                            self.remote[i] = self._recv_ptr[i]     # a ring of one: the message comes back to its sender
                        elif dst is not None:
                            p = C.c_void_p()
                            self._chk(lib.lbm_ipc_open(everyone[dst]["recv"][i], C.byref(p)), "lbm_ipc_open (message)")
                            self.remote[i] = p.value
                            self._opened.append(p.value)
            except Exception as e:                     # noqa: BLE001
                self.error, ok = e, False
        ok = all(self._gather(bool(ok)))               # (also: every handle has been opened while its owner was alive)
        if not ok:
            self._release()
            return None
        self.links = list(links)
        self.recv = [None if p is None else torch.as_tensor(_DevicePointer(p, nfloats), device=device) for p in self._recv_ptr]
        self.seq = 0
        return self.recv

    def _bound(self, items):
        return self.links is not None and len(items) == len(self.links) and all(
            (d, c) == link and (c is None or r.data_ptr() == p) and (d is None or s.numel() * s.element_size() == self.nbytes)
            for (s, d, r, c), link, p in zip(items, self.links, self._recv_ptr))

    def exchange(self, items):
        if not self._bound(items):
            return ("fallback", self.fallback.exchange(items))
        lib, stream = self._lib, torch.cuda.current_stream().cuda_stream
        self.seq = (self.seq + 1) & 0xffffffff
        q, before = self.seq, (self.seq - 1) & 0xffffffff
        # two launches and the copies between them: [post free -> await free] copies [post arrived -> await arrived]
        srcs = [i for i, (_, src) in enumerate(self.links) if src is not None]
        dsts = [i for i, (dst, _) in enumerate(self.links) if dst is not None]
        self._sync(stream, [(self._flag(self.links[i][1], i, 1), before) for i in srcs],
                   [(self._flag(self.rank, i, 1), before) for i in dsts])
        for i in dsts:
            self._chk(lib.lbm_ipc_copy(stream, self.remote[i], items[i][0].data_ptr(), self.nbytes), "lbm_ipc_copy")
        self._sync(stream, [(self._flag(self.links[i][0], i, 0), q) for i in dsts],
                   [(self._flag(self.rank, i, 0), q) for i in srcs])
        return ("ipc", None)

    def _sync(self, stream, posts, waits):
        import ctypes as C
        pf = (C.c_void_p * max(1, len(posts)))(*[p for p, _ in posts])
        pv = (C.c_uint32 * max(1, len(posts)))(*[v for _, v in posts])
        wf = (C.c_void_p * max(1, len(waits)))(*[p for p, _ in waits])
        wv = (C.c_uint32 * max(1, len(waits)))(*[v for _, v in waits])
        self._chk(self._lib.lbm_flag_sync(stream, pf, pv, len(posts), wf, wv, len(waits), self.timeout_ms, self._timed_out), "lbm_flag_sync")

    def wait(self, handle):
        kind, h = handle
        if kind == "fallback":
            self.fallback.wait(h)

    def verify(self, rounds=3):
        """collective: known patterns through every link, `rounds` times; True on every rank only if every message
        arrived intact on every rank and no wait ran into its time limit"""
        if self.links is None:
            return False
        ok = True
        try:
            dev = next(t for t in self.recv if t is not None).device
            send = [torch.empty(self.nbytes // 4, dtype=torch.float32, device=dev) for _ in self.links]
            for k in range(rounds):
                for i, (dst, _) in enumerate(self.links):
                    if dst is not None:
                        send[i].fill_(float(1 + self.rank * 64 + i * 8 + k))
                self.wait(self.exchange([(send[i], d, self.recv[i] if c is not None else send[i], c) for i, (d, c) in enumerate(self.links)]))
                for i, (_, src) in enumerate(self.links):
                    if src is not None:
                        ok = ok and bool((self.recv[i] == float(1 + src * 64 + i * 8 + k)).all())
            torch.cuda.synchronize()
            ok = ok and not self.timed_out()
        except Exception as e:                         # noqa: BLE001
            self.error, ok = e, False
        return all(self._gather(bool(ok)))

    def timed_out(self):
        """an await of THIS rank gave up (blocking read of the word the device sets)"""
        import ctypes as C
        if self.block is None:
            return False
        v = C.c_uint32(0)
        self._chk(self._lib.lbm_flag_peek(self._timed_out, C.byref(v)), "lbm_flag_peek")
        return v.value != 0

    def _release(self, collective=True):
        """collective when world > 1: nobody unmaps or frees memory a neighbour may still write into
        (collective=False: an error path, where the other ranks may never get here)"""
        torch.cuda.synchronize()
        if self.world > 1 and collective:
            self.dist.barrier()
        self.links, self.recv, self.remote = None, [], []
        for p in self._opened:
            self._lib.lbm_ipc_close(p)
        self._opened, self.peer_block = [], {}
        if self.world > 1 and collective:
            self.dist.barrier()
        for p in self._recv_ptr:
            if p is not None:
                self._lib.lbm_ipc_free(p)
        self._recv_ptr = []
        if self.block is not None:
            self._lib.lbm_ipc_free(self.block)
        self.block = None

    def close(self, collective=True):
        """collective unless told otherwise; the tensors bind() returned are gone afterwards"""
        if self.block is not None or self._opened or self._recv_ptr:
            self._release(collective)


class SlabSolver:
    """One rank's share of a W x H x depth_global domain.  Mirrors lbm_solver's surface for the
    calls that make sense per slab (init, step, set_solid with the slab's own layers, read-back)."""

    def __init__(self, engine, preset, world, rank, transport):
        self.e = engine
        self.preset, self.world, self.rank = preset, world, rank
        self.t = transport
        self.down, self.up = neighbours(preset, world, rank)
        self.send = [engine.new_message(), engine.new_message()]
        self.recv = [engine.new_message(), engine.new_message()]
        self.use_pairs = os.environ.get("LBM_SLAB_PAIRS", "1") != "0"
        if hasattr(transport, "bind"):                 # (collective: every rank builds its SlabSolver at the same point)
            # a transport that writes into its neighbours' memory brings the receive buffers itself (same order as _items)
            got = transport.bind([(self.up, self.down), (self.down, self.up)], engine.message_floats(), engine.device)
            if got is not None:
                for side, t in ((SIDE_BOTTOM, got[0]), (SIDE_TOP, got[1])):
                    if t is not None:
                        self.recv[side] = t

    # message leaving through our TOP face goes to `up` and lands on its BOTTOM ghost, and vice versa
    def _items(self, send, recv):
        return [(send[SIDE_TOP], self.up, recv[SIDE_BOTTOM], self.down),
                (send[SIDE_BOTTOM], self.down, recv[SIDE_TOP], self.up)]

    def _ctx(self):
        import contextlib
        return self.e.stream_ctx() if hasattr(self.e, "stream_ctx") else contextlib.nullcontext()

    def exchange_mask(self):
        """Static ghost layers of the solid mask; call after the mask changed on any rank."""
        with self._ctx():
            self._exchange_mask()

    def exchange_populations(self):
        """Fill the ghost layers from the current state (after init / set_populations)."""
        with self._ctx():
            self._exchange_populations()

    def step(self, nsteps=1):
        with self._ctx():
            self._step(nsteps)

    def autotune_streams(self, pool_size=4, trial_steps=6):
        """HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES = 4 here).  Which
        queue a stream lands on follows creation order, and streams that share a queue serialise:
        measured on one MI355X with an RCCL ring to itself, the same 256^3 step takes 445 us when
        the interior stream, the boundary stream and RCCL's own stream sit on three queues and
        495 - 630 us when two of them collide.  So: time a few steps on every ordered pair of
        `pool_size` fresh streams and keep the fastest pair.  Every rank runs the same number of
        trial steps (the trials are ordinary steps of the simulation), so ranks stay in lockstep.
        Returns the list of (i, j, seconds per step)."""
        import time
        e = self.e
        if not hasattr(e, "set_streams") or getattr(e, "bstream", None) is None:
            return []
        pool = [torch.cuda.Stream(device=e.device) for _ in range(pool_size)]
        trials = []
        for i in range(pool_size):
            for j in range(pool_size):
                if i == j:
                    continue
                e.set_streams(pool[i], pool[j])
                self.step(2)
                torch.cuda.synchronize(e.device)
                t0 = time.perf_counter()
                self.step(trial_steps)
                torch.cuda.synchronize(e.device)
                trials.append((i, j, (time.perf_counter() - t0) / trial_steps))
        i, j, _ = min(trials, key=lambda t: t[2])
        e.set_streams(pool[i], pool[j])
        return trials

    def _exchange_mask(self):
        e = self.e
        send = [None, None]
        recv = [e.new_mask_layer(), e.new_mask_layer()]
        for side in (SIDE_BOTTOM, SIDE_TOP):
            send[side] = torch.from_numpy(e.get_mask_layer(side)).to(recv[0].device)
        # the wrap link of the ring carries no mask: the end walls are tested before any solid
        down = self.down if self.rank > 0 else None
        up = self.up if self.rank < self.world - 1 else None
        items = [(send[SIDE_TOP], up, recv[SIDE_BOTTOM], down),
                 (send[SIDE_BOTTOM], down, recv[SIDE_TOP], up)]
        self.t.wait(self.t.exchange(items))
        if down is not None:
            e.set_mask_ghost(SIDE_BOTTOM, recv[SIDE_BOTTOM].cpu().numpy())
        if up is not None:
            e.set_mask_ghost(SIDE_TOP, recv[SIDE_TOP].cpu().numpy())

    def _exchange_populations(self):
        e = self.e
        e.halo_pack(SIDE_BOTTOM, self.send[SIDE_BOTTOM])
        e.halo_pack(SIDE_TOP, self.send[SIDE_TOP])
        self.t.wait(self.t.exchange(self._items(self.send, self.recv)))
        if self.down is not None:
            e.halo_unpack(SIDE_BOTTOM, self.recv[SIDE_BOTTOM])
        if self.up is not None:
            e.halo_unpack(SIDE_TOP, self.recv[SIDE_TOP])

    def _step(self, nsteps=1):
        e = self.e
        two = bool(getattr(e, "has_side", False))
        msg_b = self.send[SIDE_BOTTOM] if self.down is not None else None
        msg_t = self.send[SIDE_TOP] if self.up is not None else None
        rcv_b = self.recv[SIDE_BOTTOM] if self.down is not None else None
        rcv_t = self.recv[SIDE_TOP] if self.up is not None else None
        # Two steps per exchange pattern where the engine offers it (HIP A/B slabs on two streams): the fused
        # kernel advances the interior by two steps while the boundary layers take two ordinary steps, each
        # followed by its exchange.  The last step of a batch stays a single split step (rho/u on demand).
        # A rank decides for itself: a pair sends the same two messages per face, in the same order, as two single
        # split steps, so neighbours that pair and neighbours that do not step together.
        pairs = 0
        if nsteps >= 3 and two and self.use_pairs and hasattr(e, "pair_available") and e.pair_available():
            pairs = (nsteps - 1) // 2
        for _ in range(pairs):
            e.side_wait_main()                         # previous step's unpack / swap is done
            with e.side_ctx():
                e.pair_boundary_first(msg_b, msg_t)    # boundary layers t -> t+1 + messages, on the side stream
            e.pair_interior()                          # rim t -> t+1, fused interior t -> t+2, rim t+1 -> t+2
            with e.side_ctx():
                self.t.wait(self.t.exchange(self._items(self.send, self.recv)))
                e.halo_unpack_both(rcv_b, rcv_t)       # neighbours' t+1 layers (the library routes them)
                e.pair_boundary_second(msg_b, msg_t)   # boundary layers t+1 -> t+2 + messages
                self.t.wait(self.t.exchange(self._items(self.send, self.recv)))
                e.halo_unpack_both(rcv_b, rcv_t)       # neighbours' t+2 layers into the new state's ghost layers
            e.main_wait_side()
            e.pair_finish()
        for _ in range(nsteps - 2 * pairs):
            if not two:
                # one stream: boundary layers, then the exchange travels while the interior updates
                e.step_boundary(msg_b, msg_t)
                h = self.t.exchange(self._items(self.send, self.recv))
                e.step_interior()
                self.t.wait(h)
            else:
                # two streams: boundary layers + exchange on the side stream, interior on the main
                # stream at the same time.  The interior launch is issued BEFORE the RCCL calls:
                # enqueueing a send/recv group costs the host ~100 us, and the GPU would sit idle
                # behind the short boundary kernel for that long if the interior came after it.
                e.side_wait_main()                     # previous step's unpack / swap is done
                with e.side_ctx():
                    e.step_boundary(msg_b, msg_t)      # library launches on the side stream
                e.step_phase_interior()
                with e.side_ctx():
                    h = self.t.exchange(self._items(self.send, self.recv))   # ordered behind the boundary layers only
                    self.t.wait(h)
                e.main_wait_side()
            e.halo_unpack_both(rcv_b, rcv_t)
            e.step_finish()


class LocalSlabCluster:
    """All slabs of a domain inside ONE process (one device, or CPU engines in tests): same
    schedule and message routing as SlabSolver/TorchP2P, with tensor copies as the transport.
    Used to check on a single GPU that an N-slab run equals the undecomposed run bit for bit."""

    def __init__(self, engines, preset):
        self.engines = engines
        self.preset = preset
        self.world = len(engines)
        self.send = [[e.new_message(), e.new_message()] for e in engines]
        self.recv = [[e.new_message(), e.new_message()] for e in engines]

    def _route(self):
        for r in range(self.world):
            down, up = neighbours(self.preset, self.world, r)
            if up is not None:
                self.recv[up][SIDE_BOTTOM].copy_(self.send[r][SIDE_TOP])
            if down is not None:
                self.recv[down][SIDE_TOP].copy_(self.send[r][SIDE_BOTTOM])

    def _unpack_all(self):
        for r, e in enumerate(self.engines):
            down, up = neighbours(self.preset, self.world, r)
            if down is not None:
                e.halo_unpack(SIDE_BOTTOM, self.recv[r][SIDE_BOTTOM])
            if up is not None:
                e.halo_unpack(SIDE_TOP, self.recv[r][SIDE_TOP])

    def exchange_mask(self):
        layers = [[e.get_mask_layer(SIDE_BOTTOM), e.get_mask_layer(SIDE_TOP)] for e in self.engines]
        for r, e in enumerate(self.engines):
            if r > 0:
                e.set_mask_ghost(SIDE_BOTTOM, layers[r - 1][SIDE_TOP])
            if r < self.world - 1:
                e.set_mask_ghost(SIDE_TOP, layers[r + 1][SIDE_BOTTOM])

    def exchange_populations(self):
        for r, e in enumerate(self.engines):
            e.halo_pack(SIDE_BOTTOM, self.send[r][SIDE_BOTTOM])
            e.halo_pack(SIDE_TOP, self.send[r][SIDE_TOP])
        self._sync()
        self._route()
        self._sync()
        self._unpack_all()

    def _sync(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def _exchange_into_ghosts(self):
        self._sync()
        self._route()
        self._sync()
        for r, e in enumerate(self.engines):
            down, up = neighbours(self.preset, self.world, r)
            e.halo_unpack_both(self.recv[r][SIDE_BOTTOM] if down is not None else None,
                               self.recv[r][SIDE_TOP] if up is not None else None)

    def step_pairs(self, npairs):
        """npairs x two steps: through the lbm_pair_* calls on the engines that offer them, as two single split
        steps on the others.  Both send the same two messages per face in the same order (the boundary layers at
        t+1, then at t+2), so neighbouring slabs may mix the two freely -- as ranks do under SlabSolver."""
        paired = [hasattr(e, "pair_available") and e.pair_available() for e in self.engines]
        for _ in range(npairs):
            for r, e in enumerate(self.engines):
                msgs = (self.send[r][SIDE_BOTTOM], self.send[r][SIDE_TOP])
                e.pair_boundary_first(*msgs) if paired[r] else e.step_boundary(*msgs)
            for r, e in enumerate(self.engines):
                e.pair_interior() if paired[r] else e.step_interior()
            self._exchange_into_ghosts()
            for r, e in enumerate(self.engines):
                if not paired[r]:
                    e.step_finish()
            for r, e in enumerate(self.engines):
                msgs = (self.send[r][SIDE_BOTTOM], self.send[r][SIDE_TOP])
                e.pair_boundary_second(*msgs) if paired[r] else e.step_boundary(*msgs)
            for r, e in enumerate(self.engines):
                if not paired[r]:
                    e.step_interior()
            self._exchange_into_ghosts()
            for r, e in enumerate(self.engines):
                e.pair_finish() if paired[r] else e.step_finish()
        return paired

    def step(self, nsteps=1):
        for _ in range(nsteps):
            for r, e in enumerate(self.engines):
                e.step_boundary(self.send[r][SIDE_BOTTOM], self.send[r][SIDE_TOP])
            for e in self.engines:
                e.step_interior()
            self._sync()
            self._route()
            self._sync()
            self._unpack_all()
            for e in self.engines:
                e.step_finish()
