"""y-slab decomposition and ring halo exchange.

The reference is single-process; periodic boundaries are Oceananigans' in-memory halo copies
(topology = (Periodic, Periodic, Flat), jacobian_formulation/SWMHD_example.jl:16).  Here the periodic y-direction becomes a
ring of ranks: rank r owns global rows [r*Ny/P, (r+1)*Ny/P) and all x.  Because parents are x-fastest, a block of Hy
halo rows (full padded width, x halos included) is ONE contiguous run -> zero-copy send of the interior edge rows and
zero-copy receive into the halo rows.  No collective other than neighbour send/recv is on the data path.

One process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI).  The exchange itself is device-agnostic
torch code so that world_size-2 `gloo` tests can cover it on CPU (tests/test_distributed_cpu.py).
"""
import torch
import torch.distributed as dist


class SlabDecomposition:
    """Pure bookkeeping: which rows does this rank own, who are its ring neighbours."""

    def __init__(self, Ny_global, world_size=1, rank=0, force_ring=False):
        if Ny_global % world_size:
            raise ValueError(f"Ny_global={Ny_global} not divisible by world_size={world_size}")
        self.Ny_global, self.world_size, self.rank = Ny_global, world_size, rank
        self.Ny_local = Ny_global // world_size
        self.j_offset = rank * self.Ny_local
        self.south = (rank - 1) % world_size   # owns rows below mine  (smaller j)
        self.north = (rank + 1) % world_size   # owns rows above mine
        # ring: y halos come from the neighbour exchange instead of the local periodic copy.  force_ring keeps the exchange
        # path on with ONE rank (every send goes to self): the RCCL rehearsal a one-GPU box allows (tools/ring_selftest.py).
        self.ring = world_size > 1 or force_ring

    def ring_halo(self):
        """Halo for the slab's grid: 9 rows in y where the native ring driver can use them (swmhd_ring_step_rk3's deep-halo schedule:
        one neighbour exchange per RK3 step instead of one per stage; needs Hy >= 9 and a slab of >= 32 rows), else the stencil's 3."""
        return (3, 9) if (self.ring and self.Ny_local >= 32) else (3, 3)

    def local_grid(self, grid_cls, Nx, x, y, halo=(3, 3), topology=("Periodic", "Periodic", "Flat")):
        return grid_cls(size=(Nx, self.Ny_local), x=x, y=y, halo=halo, topology=topology,
                        j_offset=self.j_offset, Ny_global=self.Ny_global)


def exchange_y_halos(parents, Ny, Hy, decomp, group=None, depth=None):
    """Fill the south/north halo rows of every parent tensor in `parents` (shape (Ny+2Hy, W), contiguous) from the ring
    neighbours.  x halos must already be filled (corners travel with the rows).  world_size 1: local periodic copy.
    `depth` (default Hy) rows next to the interior are exchanged: a grid with the deep 9-row slab halo needs only the stencil's 3
    when the stages are driven one by one.
    Returns after the exchange has been *enqueued* for CUDA/NCCL tensors (stream-ordered) or completed for CPU/gloo."""
    d = Hy if depth is None else depth
    if not (0 < d <= Hy):
        raise ValueError(f"exchange depth {d} outside (0, Hy = {Hy}]")
    if not decomp.ring:
        for p in parents:
            p[Hy - d:Hy].copy_(p[Ny + Hy - d:Ny + Hy])
            p[Ny + Hy:Ny + Hy + d].copy_(p[Hy:Hy + d])
        return
    backend = dist.get_backend(group)
    cuda_over_gloo = parents[0].is_cuda and backend == "gloo"   # rehearsal mode: stage the rows through the host
    ops, stash = [], []
    for p in parents:
        send_s, send_n = p[Hy:Hy + d], p[Ny + Hy - d:Ny + Hy]   # my southern / northern interior edge rows
        recv_s, recv_n = p[Hy - d:Hy], p[Ny + Hy:Ny + Hy + d]   # my south / north halo rows (next to the interior)
        if cuda_over_gloo:
            bufs = [send_s.cpu(), send_n.cpu(), torch.empty(recv_s.shape, dtype=p.dtype), torch.empty(recv_n.shape, dtype=p.dtype)]
            stash.append((recv_s, recv_n, bufs))
            send_s, send_n, recv_s, recv_n = bufs
        ops += [dist.P2POp(dist.isend, send_n, decomp.north, group), dist.P2POp(dist.irecv, recv_s, decomp.south, group),
                dist.P2POp(dist.isend, send_s, decomp.south, group), dist.P2POp(dist.irecv, recv_n, decomp.north, group)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for recv_s, recv_n, bufs in stash:
        recv_s.copy_(bufs[2]); recv_n.copy_(bufs[3])


def agree_rc(rc, group=None, device=None):
    """The largest return code any rank of `group` holds (0 = every rank succeeded).  Used after a per-rank step that must succeed
    everywhere or nowhere -- creating the native ring's communicator -- so that a failure on one rank raises on ALL of them instead
    of leaving the others to block in the first exchange.  world size 1 / no process group: rc itself."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return int(rc)
    if device is None:
        device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([int(rc)], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())
