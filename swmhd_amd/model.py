"""Host-side mirror of the ShallowWaterModel configuration the reference builds, driving the HIP engine.

Reference call sites being mirrored:
    jacobian_formulation/SWMHD_example.jl:21-42     ShallowWaterModel(grid, timestepper=:RungeKutta3, WENO5 ..., g=9.81,
        coriolis=FPlane(f=1), tracers=(:A), forcing=(u=Forcing(lorentz_force_func_x,...), v=...),
        formulation=VectorInvariantFormulation()); set!(model, u=, v=, h=, A=); Simulation(model, dt=0.01)
    divergence_formulation/divergence_sw_mhd.jl:19-39  same with ConservativeFormulation, forcing on uh, vh
One `time_step(dt)` == Oceananigans' RK3 `time_step!`: 3 x {calculate_tendencies!, rk3_substep!, store_tendencies!
(pointer swap), update_state! (halo fill)}.  All arithmetic happens in libswmhd.so; there is no CPU path.
"""
import torch

from . import _lib
from .distributed import SlabDecomposition, agree_rc, exchange_y_halos
from .fields import Field, _SFX, _stream_ptr
from .grid import Center, Face

VectorInvariantFormulation, ConservativeFormulation = "VectorInvariant", "Conservative"
import os as _os
_FROM_STATE = _os.environ.get("SWMHD_FROM_STATE", "1") != "0"      # (A/B knob for the Python-driven stages; the C step drivers always use it in fast builds)
RK3_GAMMA = (8.0 / 15.0, 5.0 / 12.0, 3.0 / 4.0)
RK3_ZETA = (0.0, -17.0 / 60.0, -5.0 / 12.0)


def loopback_rings(nranks, timeout_s=60.0):
    """`nranks` swmhd_ring handles of the in-process loopback transport (swmhd_ring_create_loopback): rank k's exchange copies the
    edge rows of ranks k-1 and k+1 (mod nranks) on the same GPU with RCCL's rendezvous semantics.  Pass handle k as
    `ShallowWaterModel(..., decomp=SlabDecomposition(Ny, nranks, k), ring=handle)` and drive each model from its own thread."""
    import ctypes
    arr = (ctypes.c_void_p * nranks)()
    _lib.check(_lib.lib().swmhd_ring_create_loopback(arr, nranks, float(timeout_s)), "swmhd_ring_create_loopback")
    return [ctypes.c_void_p(arr[k]) for k in range(nranks)]


class ShallowWaterModel:
    def __init__(self, grid, gravitational_acceleration=9.81, coriolis_f=1.0, formulation=VectorInvariantFormulation,
                 lorentz_forcing=True, dtype=torch.float64, device="cuda", strict=False, decomp=None, group=None,
                 overlap=True, fused=True, kernel="auto", fuse_halo=True, native_ring=True, boundary_conditions=None, ring=None):
        self.grid, self.g, self.f = grid, float(gravitational_acceleration), float(coriolis_f)
        self.formulation = formulation
        self.form_code = _lib.VECTOR_INVARIANT if formulation == VectorInvariantFormulation else _lib.CONSERVATIVE
        if not lorentz_forcing:
            self.lorentz_code = _lib.LORENTZ_NONE
        else:  # the forcing that goes with each formulation in the reference
            self.lorentz_code = _lib.LORENTZ_JACOBIAN if self.form_code == _lib.VECTOR_INVARIANT else _lib.LORENTZ_DIVERGENCE
        self.strict = strict
        self._flags = (_lib.STRICT if strict else _lib.FAST) | _lib.KERNEL_FLAGS[kernel]
        # topology = (Periodic | Bounded, Periodic | Bounded, Flat): Bounded directions get wall reconstructions in the kernels
        # (SWMHD_BOUNDED_X / _Y) and the boundary-condition halo fill (swmhd_fill_halo) instead of the periodic copy
        tx, ty = grid.topo_codes()
        self._bounded = (tx == _lib.BOUNDED, ty == _lib.BOUNDED)
        self._flags |= (_lib.BOUNDED_X if self._bounded[0] else 0) | (_lib.BOUNDED_Y if self._bounded[1] else 0)
        self.decomp = decomp or SlabDecomposition(grid.Ny_global, 1, 0)
        if any(self._bounded) and self.decomp.ring:
            raise _lib.SwmhdError("y-slab decomposition (ring halo exchange) supports (Periodic, Periodic) grids only (SWMHD_ENOTSUP)")
        # boundary_conditions = {"A": FieldBoundaryConditions(north = GradientBoundaryCondition(-0.05), ...)}  (SWMHD_example.jl:18-22)
        self.boundary_conditions = dict(boundary_conditions or {})
        for name, bc in self.boundary_conditions.items():
            sides = [(bc.west, 0), (bc.east, 0), (bc.south, 1), (bc.north, 1)]
            if any(b is not None and not self._bounded[d] for b, d in sides):
                raise _lib.SwmhdError(f"boundary condition on a Periodic side of {name} (Oceananigans rejects it as well)")
        self.group, self.overlap = group, overlap
        n1, n2 = ("u", "v") if self.form_code == _lib.VECTOR_INVARIANT else ("uh", "vh")
        self.names = (n1, n2, "h", "A")
        locs = ((Face, Center), (Center, Face), (Center, Center), (Center, Center))
        mk = lambda loc: Field(grid, loc, dtype, device)
        self._state = {n: mk(l) for n, l in zip(self.names, locs)}
        self.fused = fused                    # one kernel per RK3 stage (tendencies + substep), state ping-ponged
        # Periodic "gather on read" (SWMHD_WRAP_X / _Y): the tendency kernels take the periodic image instead of the halo cell, so
        # no halo-fill launch runs between RK3 stages -- x and y on one GPU, x on a slab (its y halos come from the ring).  The halos
        # of the state are then filled lazily, when something other than a tendency kernel is about to read them (_ensure_halos).
        self._rwrap = 0
        if fuse_halo and grid.Nx >= grid.Hx and grid.Ny >= grid.Hy:
            self._rwrap = (0 if self._bounded[0] else _lib.WRAP_X) | (0 if (self.decomp.ring or self._bounded[1]) else _lib.WRAP_Y)
        self._halo_stale = False
        self._exchange_in_flight = False      # torch p2p overlap path: a y exchange is queued on the comm stream
        self._alt = {n: mk(l) for n, l in zip(self.names, locs)} if fused else None
        self.Gn = [mk(l) for l in locs]     # Gⁿ
        self.Gm = [mk(l) for l in locs]     # G⁻
        self.sfx = _SFX[dtype]
        self.clock_time, self.iteration = 0.0, 0
        self._comm_stream = torch.cuda.Stream() if (self.decomp.ring and torch.cuda.is_available()) else None
        self._L = _lib.lib()
        self.tendency_events = None   # bench.py: list collecting (start, end) HIP events around every tendency launch
        if any(not f.data.is_cuda for f in self._state.values()):
            raise _lib.SwmhdError("ShallowWaterModel runs on the GPU only (no CPU fallback)")
        # y-slab ring: the native RCCL ring (swmhd_ring_*) when the process group is RCCL; torch.distributed p2p otherwise
        # (gloo rehearsals).  The ring's step driver needs the fused stage kernel and a slab taller than its two strips.
        # `ring`: a swmhd_ring handle created by the caller -- one of swmhd_ring_create_loopback's (loopback_rings below): several
        # slabs of one domain in ONE process on one GPU, each driven from its own host thread and stream.  The model owns it.
        self._ring = None
        if ring is not None:
            if not (self.decomp.ring and fused and grid.Ny > 2 * grid.Hy):
                raise _lib.SwmhdError("ring= needs a ring decomposition, the fused stage kernel and a slab taller than its two strips")
            self._ring, self._comm_stream = ring, None
        elif self.decomp.ring and native_ring and fused and grid.Ny > 2 * grid.Hy:
            self._ring = self._create_ring()

    def _create_ring(self):
        import ctypes, os
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_backend(self.group) == "nccl"):
            return None
        dev = self._raw_fields[0].data.device
        rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")     # the copy torch has loaded
        rccl = rccl.encode() if os.path.exists(rccl) else None
        # The ranks AGREE that every one of them can load RCCL (swmhd_ring_available: dlopen + symbol lookup, creates nothing) before any
        # of them enters the collective ncclCommInitRank: a rank that cannot must not leave its peers blocked inside communicator creation
        # -- then all of them take the torch.distributed p2p path instead.
        rc = self._L.swmhd_ring_available(rccl)
        ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:
            if rc != 0:
                import sys
                print(f"swmhd_amd: native ring unavailable ({self._L.swmhd_strerror(rc).decode()}); using torch.distributed p2p",
                      file=sys.stderr)
            return None
        ident = torch.zeros(_lib.RING_ID_BYTES, dtype=torch.uint8)
        if self.decomp.rank == 0:
            buf = (ctypes.c_ubyte * _lib.RING_ID_BYTES)()
            _lib.check(self._L.swmhd_ring_unique_id(rccl, buf), "swmhd_ring_unique_id")
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        ident = ident.to(dev)
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast(ident, src=src, group=self.group)
        raw = bytes(ident.cpu().tolist())
        ring = ctypes.c_void_p()
        with torch.cuda.device(dev):
            rc = self._L.swmhd_ring_create(ctypes.byref(ring), rccl, self.decomp.world_size, self.decomp.rank,
                                           (ctypes.c_ubyte * _lib.RING_ID_BYTES).from_buffer_copy(raw))
        import os
        if os.environ.get("SWMHD_TEST_RING_CREATE_FAIL_RANK") == str(self.decomp.rank):     # tests: a rank whose communicator failed
            if rc == 0:
                self._L.swmhd_ring_destroy(ring)
            rc = 4
        # every rank learns whether ALL communicators exist: a rank that failed raises, and so do its peers (they destroy theirs
        # first) -- otherwise they would block in the first exchange until the launcher's deadline
        worst = agree_rc(rc, self.group, dev)
        if worst != 0:
            if rc == 0:
                self._L.swmhd_ring_destroy(ring)
                raise _lib.SwmhdError(f"swmhd_ring_create failed on another rank (rc {worst}); this rank destroyed its communicator")
            _lib.check(rc, "swmhd_ring_create")
        self._comm_stream = None      # the ring owns the comm stream of the native path
        return ring

    def _ring_check(self, rc, what):
        if rc == 4:   # SWMHD_ECOMM
            raise _lib.SwmhdError(f"{what}: {self._L.swmhd_ring_last_error(self._ring).decode()}")
        _lib.check(rc, what)

    def ring_time_launches(self, n):
        """bench.py: have the native ring driver record HIP events around its next n interior launches."""
        self._ring_check(self._L.swmhd_ring_time_launches(self._ring, n), "swmhd_ring_time_launches")

    def ring_launch_times(self, capacity=4096):
        import ctypes
        ms, rows = (ctypes.c_float * capacity)(), (ctypes.c_int * capacity)()
        n = self._L.swmhd_ring_launch_times(self._ring, ms, rows, capacity)
        return [(ms[k], rows[k]) for k in range(max(n, 0))]

    def _join(self):
        """Order the current stream behind whatever halo exchange is still in flight."""
        if self._ring is not None:
            self._ring_check(self._L.swmhd_ring_join(self._ring, _stream_ptr()), "swmhd_ring_join")
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        self._exchange_in_flight = False

    def close(self):
        """Release the ring (RCCL communicator, comm stream).  Collective over the ranks like its creation; call it before
        torch.distributed.destroy_process_group()."""
        ring, self._ring = getattr(self, "_ring", None), None
        if ring is not None:
            self._L.swmhd_ring_destroy(ring)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- set!(model, u=..., v=..., h=..., A=...) ---------------------------------------------------------------
    def set(self, **kw):
        self._join()
        for k, v in kw.items():
            self._state[k].set(v)
        self.update_state()
        return self

    @property
    def solution(self):
        """The prognostic fields by name (u|uh, v|vh, h, A), halos current."""
        self._ensure_halos()
        return self._state

    @property
    def fields(self):
        self._ensure_halos()
        return [self._state[n] for n in self.names]

    @property
    def _raw_fields(self):
        """The prognostic fields without touching their halos (which may be stale between stages)."""
        return [self._state[n] for n in self.names]

    def _ensure_halos(self):
        if self._halo_stale:
            self.update_state()

    # --- update_state!: fill halos (periodic x locally; y locally or by ring exchange) --------------------------
    def _fill_bc(self, stream=None):
        """fill_halo_regions! with boundary conditions (at least one Bounded direction): swmhd_fill_halo."""
        import ctypes
        g = self.grid
        q = self._raw_fields
        ct = ctypes.c_double if self.sfx == "f64" else ctypes.c_float
        grads = []
        for n in self.names:
            bc = self.boundary_conditions.get(n)
            grads += bc.gradients() if bc is not None else [float("nan")] * 4
        f = getattr(self._L, f"swmhd_fill_halo_{self.sfx}")
        tx, ty = g.topo_codes()
        rc = f(_lib.ptr_array([x.ptr for x in q]), 4, g.Nx, g.Ny, g.Hx, g.Hy, q[0].stride_y, tx, ty, 0b0001, 0b0010,
               (ct * 16)(*grads), g.dx, g.dy, _stream_ptr(stream))
        _lib.check(rc, "swmhd_fill_halo")

    def _fill_x(self, stream=None):
        if any(self._bounded):
            return self._fill_bc(stream)
        g = self.grid
        q = self._raw_fields
        ptrs = _lib.ptr_array([f.ptr for f in q])
        which = _lib.HALO_X | (0 if self.decomp.ring else _lib.HALO_Y)
        f = getattr(self._L, f"swmhd_fill_halo_periodic_multi_{self.sfx}")
        _lib.check(f(ptrs, 4, g.Nx, g.Ny, g.Hx, g.Hy, q[0].stride_y, which, _stream_ptr(stream)), "fill_halo_multi")

    def update_state(self):
        self._join()
        self._halo_stale = False
        self._fill_x()
        q = self._raw_fields
        if self._ring is not None:
            g = self.grid
            f = getattr(self._L, f"swmhd_ring_exchange_y_{self.sfx}")
            rc = f(self._ring, _lib.ptr_array([x.ptr for x in q]), 4, g.Nx, g.Ny, g.Hx, g.Hy, q[0].stride_y, _stream_ptr())
            self._ring_check(rc, "swmhd_ring_exchange_y")
        elif self.decomp.ring:
            exchange_y_halos([f.data for f in q], self.grid.Ny, self.grid.Hy, self.decomp, self.group)

    # --- calculate_tendencies! ------------------------------------------------------------------------------
    def calculate_tendencies(self, rows=None, stream=None):
        if self.tendency_events is not None and rows is None:
            e0, e1 = _lib.TimingEvent(), _lib.TimingEvent()
            e0.record()
            self._calculate_tendencies(rows, stream)
            e1.record()
            self.tendency_events.append((e0, e1, self.grid.Ny))
        else:
            self._calculate_tendencies(rows, stream)

    def _calculate_tendencies(self, rows=None, stream=None):
        g = self.grid
        q = self._raw_fields
        j0, j1 = (0, g.Ny) if rows is None else rows
        f = getattr(self._L, f"swmhd_tendencies_{self.sfx}")
        rc = f(q[0].ptr, q[1].ptr, q[2].ptr, q[3].ptr, self.Gn[0].ptr, self.Gn[1].ptr, self.Gn[2].ptr, self.Gn[3].ptr,
               g.Nx, g.Ny, g.Hx, g.Hy, q[0].stride_y, g.dx, g.dy, self.g, self.f, self.form_code, self.lorentz_code,
               j0, j1, self._flags | self._rwrap, _stream_ptr(stream))
        _lib.check(rc, "swmhd_tendencies")

    def _substep(self, dt, stage):
        g = self.grid
        U = _lib.ptr_array([f.ptr for f in self._raw_fields])
        Gn = _lib.ptr_array([f.ptr for f in self.Gn])
        Gm = _lib.ptr_array([f.ptr for f in self.Gm]) if stage > 0 else None
        f = getattr(self._L, f"swmhd_rk3_substep_{self.sfx}")
        rc = f(U, Gn, Gm, g.Nx, g.Ny, g.Hx, g.Hy, self._raw_fields[0].stride_y, dt, RK3_GAMMA[stage], RK3_ZETA[stage], 0, g.Ny,
               _lib.STRICT if self.strict else _lib.FAST, _stream_ptr())
        _lib.check(rc, "swmhd_rk3_substep")

    def _stage_fused(self, dt, stage, rows=None, extra_flags=0):
        """calculate_tendencies! + rk3_substep! in one launch: reads the current state, writes the new state into the
        alternate buffers (swmhd_tendencies_rk3_*)."""
        g = self.grid
        j0, j1 = (0, g.Ny) if rows is None else rows
        q = _lib.ptr_array([f.ptr for f in self._raw_fields])
        qn = _lib.ptr_array([self._alt[n].ptr for n in self.names])
        Gn = _lib.ptr_array([f.ptr for f in self.Gn])
        Gm = _lib.ptr_array([f.ptr for f in self.Gm]) if stage > 0 else None
        # Fast periodic builds: the second stage takes G- = (U1 - U0) / (dt gamma1) from the two states -- U0 is still in the buffer this
        # stage writes U2 to -- so the first stage stores no tendencies (swmhd.h SWMHD_GM_IS_PREV_STATE; 288 instead of 320 B/cell-step).
        from_state = not self.strict and not any(self._bounded) and _FROM_STATE
        zeta, store = RK3_ZETA[stage], (1 if stage < 2 else 0)
        if from_state and stage == 0:
            store = 0
        if from_state and stage == 1:
            Gm, zeta, extra_flags = qn, RK3_ZETA[1] / RK3_GAMMA[0], extra_flags | _lib.GM_IS_PREV_STATE
        f = getattr(self._L, f"swmhd_tendencies_rk3_{self.sfx}")
        timed = self.tendency_events is not None and 2 * (j1 - j0) > g.Ny    # whole grid, or the interior launch of a slab
        if timed:
            e0, e1 = _lib.TimingEvent(), _lib.TimingEvent()
            e0.record()
        rc = f(q, qn, Gn, Gm, g.Nx, g.Ny, g.Hx, g.Hy, self._raw_fields[0].stride_y, g.dx, g.dy, self.g, self.f, self.form_code,
               self.lorentz_code, dt, RK3_GAMMA[stage], zeta, store, j0, j1,
               self._flags | self._rwrap | extra_flags, _stream_ptr())
        if timed:
            e1.record()
            self.tendency_events.append((e0, e1, j1 - j0))
        _lib.check(rc, "swmhd_tendencies_rk3")

    # --- time_step!(model, dt): RungeKutta3 ------------------------------------------------------------------
    def _ring_steps(self, dt, n):
        """n RK3 steps of this slab through the native ring driver (swmhd_ring_step_rk3_*): one C call enqueues every launch."""
        import ctypes
        gr = self.grid
        swapped = ctypes.c_int(0)
        if self._halo_stale and not (self._rwrap & _lib.WRAP_X):
            self.update_state()
        q = _lib.ptr_array([f.ptr for f in self._raw_fields])
        qa = _lib.ptr_array([self._alt[nm].ptr for nm in self.names])
        Ga = _lib.ptr_array([f.ptr for f in self.Gn])
        Gb = _lib.ptr_array([f.ptr for f in self.Gm])
        f = getattr(self._L, f"swmhd_ring_step_rk3_{self.sfx}")
        rc = f(self._ring, q, qa, Ga, Gb, gr.Nx, gr.Ny, gr.Hx, gr.Hy, self._raw_fields[0].stride_y, gr.dx, gr.dy, self.g, self.f,
               self.form_code, self.lorentz_code, dt, n, self._flags | (self._rwrap & _lib.WRAP_X), ctypes.byref(swapped), _stream_ptr())
        self._ring_check(rc, "swmhd_ring_step_rk3")
        if swapped.value:
            self._state, self._alt = self._alt, self._state
            self.Gn, self.Gm = self.Gm, self.Gn
        if n > 0 and (self._rwrap & _lib.WRAP_X):
            self._halo_stale = True       # x halos were not filled; the y exchange of the final state is in flight (see _join)
        self.clock_time += n * dt
        self.iteration += n

    def time_step(self, dt):
        if self._ring is not None:
            return self._ring_steps(dt, 1)
        g, H = self.grid, 3          # strips and per-stage exchange: the stencil's reach, whatever the grid's halo depth
        multi = self.decomp.ring
        overlap = multi and self.overlap and self._comm_stream is not None and g.Ny > 2 * H
        for stage in range(3):
            # one RK3 stage over a row range: either the fused kernel or tendencies followed (later) by the substep
            run = (lambda rows=None, fl=0: self._stage_fused(dt, stage, rows, fl)) if self.fused else (lambda rows=None, fl=0: self.calculate_tendencies(rows=rows))
            if overlap and self._exchange_in_flight:
                # x halos are current; the y exchange of the previous stage is in flight on the comm stream.  Interior rows run
                # on the main stream; the two H-row boundary strips are queued on the COMM stream behind the exchange, so they
                # start the moment the halo rows land and overlap the tail of the interior kernel.
                run((H, g.Ny - H), _lib.LEAVE_ROOM)    # leave workgroup slots for the comm stream's kernels
                with torch.cuda.stream(self._comm_stream):
                    run((0, H))
                    run((g.Ny - H, g.Ny))
                torch.cuda.current_stream().wait_stream(self._comm_stream)
            else:
                run()
            if self.fused:
                self._state, self._alt = self._alt, self._state      # the new state becomes current
            else:
                self._substep(dt, stage)
            self.Gn, self.Gm = self.Gm, self.Gn          # store_tendencies!: G⁻ <- Gⁿ (pointer swap, 0 bytes)
            if any(self._bounded) or not (self._rwrap & _lib.WRAP_X):
                self._fill_x()                           # (otherwise the next stage reads the periodic images itself)
            else:
                self._halo_stale = True
            if multi:
                if overlap:
                    self._comm_stream.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(self._comm_stream):
                        exchange_y_halos([f.data for f in self._raw_fields], g.Ny, g.Hy, self.decomp, self.group, depth=H)
                    self._exchange_in_flight = True
                else:
                    exchange_y_halos([f.data for f in self._raw_fields], g.Ny, g.Hy, self.decomp, self.group, depth=H)
        self.clock_time += dt
        self.iteration += 1

    # --- HIP-graph replay of the step (single GPU): the reference's own grids are 64^2 .. 128^2 (SWMHD_example.jl:11), where a
    #     step is 6 launches of ~10 us kernels and the host would otherwise set the pace --------------------------------
    def capture_graph(self, dt):
        """Capture TWO RK3 steps (6 fused stages + 6 halo fills) into one HIP graph; two, because the ping-ponged state and the
        G-/Gn pointers return to their original roles after an even number of stages.  `time_steps` then replays it."""
        if self.decomp.ring or not self.fused:
            raise _lib.SwmhdError("capture_graph: single-GPU fused path only (halo exchange is not capturable)")
        self._ensure_halos()
        keep = [f.data.clone() for f in self._raw_fields] + [f.data.clone() for f in self.Gm]
        t0, i0 = self.clock_time, self.iteration
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up outside capture (lazy module loads etc.)
            self.time_step(dt); self.time_step(dt)
        torch.cuda.current_stream().wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self.time_step(dt); self.time_step(dt)
        self._graph_dt = dt
        # the graph has the device pointers of THIS role assignment baked in (state in `solution`, scratch in `_alt`, Gn/Gm as
        # they are now): it may only be replayed while the roles are the same, i.e. after an even number of eager steps
        self._graph_roles = self._roles()
        for f, k in zip(self._raw_fields + self.Gm, keep):   # capture does not execute; undo the two warm-up steps
            f.data.copy_(k)
        self._halo_stale = False
        self.clock_time, self.iteration = t0, i0
        return self

    def _roles(self):
        """Which buffer plays which role right now (every RK3 step swaps state<->scratch and Gn<->G- an odd number of times)."""
        return tuple(f.ptr for f in self._raw_fields) + tuple(f.ptr for f in self.Gn)

    def time_steps(self, n, dt):
        """n RK3 steps: graph replays (2 steps each) when a graph was captured for this dt; otherwise the native step driver
        (swmhd_step_rk3_*: one C call enqueues all 6n launches) on a single GPU, or Python-driven stages on several."""
        g = getattr(self, "_graph", None)
        if g is not None and self._graph_dt == dt and n >= 2 and self._roles() != self._graph_roles:
            # an odd number of steps has run since capture (a leftover step of an earlier call, or a plain time_step): the
            # graph would read the scratch buffers as the state.  One eager step restores the captured roles.
            self._driver_steps(dt, 1)
            n -= 1
        if g is not None and self._graph_dt == dt and self._roles() == self._graph_roles:
            for _ in range(n // 2):
                g.replay()
                self._halo_stale = self._halo_stale or bool(self._rwrap)
                self.clock_time += 2 * dt
                self.iteration += 2
            n = n % 2
        if n > 0:
            self._driver_steps(dt, n)

    def _driver_steps(self, dt, n):
        if self._ring is not None:
            return self._ring_steps(dt, n)
        if not self.decomp.ring and self.fused and self.tendency_events is None and not any(self._bounded):
            gr = self.grid
            import ctypes
            swapped = ctypes.c_int(0)
            q = _lib.ptr_array([f.ptr for f in self._raw_fields])
            qa = _lib.ptr_array([self._alt[nm].ptr for nm in self.names])
            Ga = _lib.ptr_array([f.ptr for f in self.Gn])
            Gb = _lib.ptr_array([f.ptr for f in self.Gm])
            f = getattr(self._L, f"swmhd_step_rk3_{self.sfx}")
            rc = f(q, qa, Ga, Gb, gr.Nx, gr.Ny, gr.Hx, gr.Hy, self._raw_fields[0].stride_y, gr.dx, gr.dy, self.g, self.f, self.form_code,
                   self.lorentz_code, dt, n, self._flags | self._rwrap, ctypes.byref(swapped), _stream_ptr())
            _lib.check(rc, "swmhd_step_rk3")
            if swapped.value:
                self._state, self._alt = self._alt, self._state
                self.Gn, self.Gm = self.Gm, self.Gn
            if n > 0 and self._rwrap:
                self._halo_stale = True
            self.clock_time += n * dt
            self.iteration += n
            return
        for _ in range(n):
            self.time_step(dt)

    # --- diagnostics (SWMHD_example.jl:47-77): energies and extrema in one device pass ------------------------
    def diagnostics(self, h_ref=1.0):
        """dict(kinetic_energy, magnetic_energy, potential_energy, total_energy, max_abs_u, max_abs_v, max_abs_A, min_h)
        over the whole (possibly decomposed) domain; energies as the reference's mean(...)*Lx*Ly."""
        g = self.grid
        self._join()
        if not hasattr(self, "_diag_ws"):
            self._diag_ws = torch.empty(_lib.DIAG_WORKSPACE, dtype=torch.float64, device=self.fields[0].data.device)
            self._diag_out = torch.empty(_lib.DIAG_NOUT, dtype=torch.float64, device=self.fields[0].data.device)
        q = self.fields
        f = getattr(self._L, f"swmhd_diagnostics_{self.sfx}")
        rc = f(q[0].ptr, q[1].ptr, q[2].ptr, q[3].ptr, g.Nx, g.Ny, g.Hx, g.Hy, q[0].stride_y, g.dx, g.dy, self.g, h_ref,
               self.form_code, 0, g.Ny, self._diag_ws.data_ptr(), self._diag_out.data_ptr(), _stream_ptr())
        _lib.check(rc, "swmhd_diagnostics")
        out = self._diag_out.clone()
        if self.decomp.world_size > 1:   # the only collective in the package, 7 scalars, off the data path
            import torch.distributed as dist
            sums, maxs, mins = out[:3].clone(), out[3:6].clone(), out[6:].clone()
            if sums.is_cuda and dist.get_backend(self.group) == "gloo":
                sums, maxs, mins = sums.cpu(), maxs.cpu(), mins.cpu()
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(maxs, op=dist.ReduceOp.MAX, group=self.group)
            dist.all_reduce(mins, op=dist.ReduceOp.MIN, group=self.group)
            out = torch.cat([sums.cpu(), maxs.cpu(), mins.cpu()])
        v = out.cpu().tolist()
        return dict(kinetic_energy=v[0], magnetic_energy=v[1], potential_energy=v[2], total_energy=v[0] + v[1] + v[2],
                    max_abs_u=v[3], max_abs_v=v[4], max_abs_A=v[5], min_h=v[6])

    # --- checkpoint: prognostic fields with halos + G⁻ + clock, one .npz per rank ---------------------------------
    def save_checkpoint(self, path):
        import numpy as np
        self.synchronize()
        np.savez(path, time=self.clock_time, iteration=self.iteration,
                 **{n: f.numpy() for n, f in zip(self.names, self.fields)},
                 **{"Gm_" + n: f.numpy() for n, f in zip(self.names, self.Gm)})

    def load_checkpoint(self, path):
        import numpy as np
        self._join()
        z = np.load(path, allow_pickle=False)
        self._halo_stale = False              # the checkpoint holds the parents, halos included
        for n, f in zip(self.names, self._raw_fields):
            f.data.copy_(torch.from_numpy(z[n]).to(f.data.dtype))
        for n, f in zip(self.names, self.Gm):
            f.data.copy_(torch.from_numpy(z["Gm_" + n]).to(f.data.dtype))
        self.clock_time, self.iteration = float(z["time"]), int(z["iteration"])
        return self

    def synchronize(self):
        """Wait for everything enqueued; afterwards the halos of the state are current (they are filled lazily, see __init__)."""
        self._ensure_halos()
        self._join()
        torch.cuda.synchronize()
