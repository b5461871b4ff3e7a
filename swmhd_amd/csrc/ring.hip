// y-slab ring for the multi-GPU engine: an RCCL communicator, a high-priority comm stream and the native
// step driver that overlaps the neighbour exchange with the interior rows.
//
// The reference is single-process: its periodic y-boundary is Oceananigans' in-memory halo copy (topology =
// (Periodic, Periodic, Flat), jacobian_formulation/SWMHD_example.jl:16; fill_halo_regions! inside update_state!).  With one
// process per GPU that copy becomes a ring of ncclSend/ncclRecv between y-neighbours.  Parents are x-fastest, so the Hy edge
// rows of a field (full padded width, corners included) are ONE contiguous run: every send reads the interior edge rows in
// place and every receive lands directly in the halo rows -- no pack/unpack kernels, one grouped RCCL launch per exchange.
//
// RCCL is bound at run time (dlopen of the path the host passes -- the copy PyTorch already loaded when the host is the
// Python harness) so that libswmhd.so itself carries no link-time dependency on it; single-GPU users never touch it.
#include "common.hpp"
#include "../../include/swmhd.h"
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <dlfcn.h>
#include <memory>
#include <mutex>
#include <new>
#include <rccl/rccl.h>
#include <string>
#include <vector>

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

bool load_rccl(const char *path, RcclApi &api, std::string &err) {
    const char *candidates[] = {path, "librccl.so.1", "librccl.so"};
    for (const char *c : candidates) {
        if (!c || !*c) continue;
        api.handle = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (api.handle) break;
        err = dlerror();
    }
    if (!api.handle) return false;
#define SW_SYM(field, name)                                                                \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name));           \
    if (!api.field) { err = std::string("missing symbol ") + name; return false; }
    SW_SYM(GetUniqueId, "ncclGetUniqueId")
    SW_SYM(CommInitRank, "ncclCommInitRank")
    SW_SYM(CommDestroy, "ncclCommDestroy")
    SW_SYM(GroupStart, "ncclGroupStart")
    SW_SYM(GroupEnd, "ncclGroupEnd")
    SW_SYM(Send, "ncclSend")
    SW_SYM(Recv, "ncclRecv")
    SW_SYM(GetErrorString, "ncclGetErrorString")
#undef SW_SYM
    return true;
}

// ---- loopback transport ---------------------------------------------------------------------------------------------------------
// K rings in ONE process on ONE GPU (RCCL refuses two ranks per device): the exchange of ring r copies its neighbours' edge rows
// into its own halo rows with hipMemcpyAsync on its comm stream, with the rendezvous semantics RCCL gives a grouped send/recv:
//   a receive completes only after the sender has reached the matching exchange   (wait for the peer's "rows ready" event)
//   a send completes only after the receiver has taken the rows                     (wait for the peer's "consumed" event)
// so that the real driver (ring_step) runs with north != south, every ring on its own pair of streams.  Events can only be waited
// for once they have been recorded, hence a HOST-side rendezvous as well: exchange number k of a ring blocks until its neighbours
// have enqueued theirs -- the rings must be driven from one host thread each, as RCCL ranks are driven from one process each.  A
// neighbour that does not arrive within the hub's timeout ends the exchange with SWMHD_ECOMM instead of blocking for ever.
struct LoopHub {
    static constexpr int MAXF = 8, SLOTS = 4;
    struct Post {
        unsigned long seq = 0;                 // number of exchanges this rank has posted (rows ready)
        unsigned long done = 0;                // number of exchanges whose receives this rank has enqueued (rows consumed)
        const char *send_s[SLOTS][MAXF] = {}, *send_n[SLOTS][MAXF] = {};
        size_t bytes[SLOTS] = {};
        int nf[SLOTS] = {};
        hipEvent_t ready[SLOTS] = {}, consumed[SLOTS] = {};
    };
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Post> post;
    double timeout_s = 60.0;
    explicit LoopHub(int n) : post(n) {}
    ~LoopHub() {
        for (auto &p : post)
            for (int k = 0; k < SLOTS; ++k) { if (p.ready[k]) (void)hipEventDestroy(p.ready[k]); if (p.consumed[k]) (void)hipEventDestroy(p.consumed[k]); }
    }
};

}  // namespace

struct swmhd_ring {
    RcclApi api;
    ncclComm_t comm = nullptr;
    std::shared_ptr<LoopHub> hub;    // loopback transport (swmhd_ring_create_loopback); comm stays NULL then
    int nranks = 1, rank = 0, south = 0, north = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_main = nullptr, ev_comm = nullptr;
    const void *pending = nullptr;   // first parent of the state whose y-exchange is in flight on comm_stream (NULL: none)
    std::string err;
    // optional timing of the interior launches (bench.py's roofline leg)
    std::vector<hipEvent_t> t0, t1;
    std::vector<int> trows;
    size_t tcap = 0;
};

namespace {

int fail(swmhd_ring *r, const char *what, ncclResult_t rc) {
    r->err = std::string(what) + ": " + (r->api.GetErrorString ? r->api.GetErrorString(rc) : "rccl error");
    return SWMHD_ECOMM;
}
int hipfail(swmhd_ring *r, const char *what, hipError_t e) {
    r->err = std::string(what) + ": " + hipGetErrorString(e);
    return -(int)e;
}

template <typename T> constexpr ncclDataType_t nccl_type();
template <> constexpr ncclDataType_t nccl_type<double>() { return ncclFloat64; }
template <> constexpr ncclDataType_t nccl_type<float>() { return ncclFloat32; }

// One grouped launch: for every field, northern edge rows -> north neighbour's south halo, southern edge rows -> south
// neighbour's north halo.  Issue order (send_n, recv_s, send_s, recv_n per field) is what makes the 1- and 2-rank rings,
// where both neighbours are the same peer, pair up correctly: RCCL matches sends and receives of a peer in issue order.
// The same exchange through the loopback hub (see LoopHub): post my edge rows, copy my neighbours' into my halos, wait until mine were taken.
int loopback_exchange(swmhd_ring *r, const char *const *send_s, const char *const *send_n, char *const *recv_s, char *const *recv_n,
                      int nf, size_t bytes, hipStream_t s) {
    LoopHub &h = *r->hub;
    if (nf > LoopHub::MAXF) return SWMHD_EINVAL;
    hipError_t e;
    unsigned long k;
    int slot;
    {   // (1) rows ready: everything enqueued on s so far precedes the neighbours' reads
        std::unique_lock<std::mutex> lk(h.mu);
        LoopHub::Post &me = h.post[r->rank];
        k = me.seq + 1; slot = (int)(k % LoopHub::SLOTS);
        if (!me.ready[slot] && (e = hipEventCreateWithFlags(&me.ready[slot], hipEventDisableTiming)) != hipSuccess) return hipfail(r, "loopback event", e);
        if (!me.consumed[slot] && (e = hipEventCreateWithFlags(&me.consumed[slot], hipEventDisableTiming)) != hipSuccess) return hipfail(r, "loopback event", e);
        if ((e = hipEventRecord(me.ready[slot], s)) != hipSuccess) return hipfail(r, "loopback record", e);
        for (int f = 0; f < nf; ++f) { me.send_s[slot][f] = send_s[f]; me.send_n[slot][f] = send_n[f]; }
        me.bytes[slot] = bytes; me.nf[slot] = nf;
        me.seq = k;
    }
    h.cv.notify_all();
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(h.timeout_s);
    auto wait_for = [&](auto pred, const char *what) {
        std::unique_lock<std::mutex> lk(h.mu);
        if (!h.cv.wait_until(lk, deadline, pred)) {
            r->err = std::string("loopback exchange ") + std::to_string(k) + " of rank " + std::to_string(r->rank) + ": " + what +
                     " (every ring of the hub must be driven from its own host thread, and all of them must make the same calls)";
            return false;
        }
        return true;
    };
    // (2) receives: my south halo <- the south neighbour's northern edge rows; my north halo <- the north neighbour's southern ones
    const int peers[2] = {r->south, r->north};
    for (int side = 0; side < 2; ++side) {
        const int p = peers[side];
        if (!wait_for([&] { return h.post[p].seq >= k; }, "a neighbour did not reach its matching exchange")) return SWMHD_ECOMM;
        hipEvent_t ready; const char *src[LoopHub::MAXF]; size_t pbytes; int pnf;
        {
            std::unique_lock<std::mutex> lk(h.mu);
            const LoopHub::Post &pp = h.post[p];
            ready = pp.ready[slot]; pbytes = pp.bytes[slot]; pnf = pp.nf[slot];
            for (int f = 0; f < nf && f < pnf; ++f) src[f] = side == 0 ? pp.send_n[slot][f] : pp.send_s[slot][f];
        }
        if (pbytes != bytes || pnf != nf) { r->err = "loopback exchange: neighbour posted a different message (fields / bytes)"; return SWMHD_ECOMM; }
        if ((e = hipStreamWaitEvent(s, ready, 0)) != hipSuccess) return hipfail(r, "loopback wait", e);
        for (int f = 0; f < nf; ++f)
            if ((e = hipMemcpyAsync(side == 0 ? recv_s[f] : recv_n[f], src[f], bytes, hipMemcpyDeviceToDevice, s)) != hipSuccess)
                return hipfail(r, "loopback copy", e);
    }
    {   // (3) rows consumed
        std::unique_lock<std::mutex> lk(h.mu);
        LoopHub::Post &me = h.post[r->rank];
        if ((e = hipEventRecord(me.consumed[slot], s)) != hipSuccess) return hipfail(r, "loopback record", e);
        me.done = k;
    }
    h.cv.notify_all();
    // (4) my sends complete when both neighbours have taken my rows: later work on s may overwrite them
    for (int side = 0; side < 2; ++side) {
        const int p = peers[side];
        if (!wait_for([&] { return h.post[p].done >= k; }, "a neighbour did not take the rows sent to it")) return SWMHD_ECOMM;
        hipEvent_t consumed;
        { std::unique_lock<std::mutex> lk(h.mu); consumed = h.post[p].consumed[slot]; }
        if ((e = hipStreamWaitEvent(s, consumed, 0)) != hipSuccess) return hipfail(r, "loopback wait", e);
    }
    return SWMHD_OK;
}

template <typename T>
int exchange(swmhd_ring *r, T *const *fields, int nf, int Nx, int Ny, int Hx, int Hy, int64_t sy, hipStream_t s) {
    if (!r || !fields || nf <= 0) return SWMHD_EINVAL;
    if (Nx <= 0 || Ny < Hy || Hy <= 0 || sy < (int64_t)Nx + 2 * Hx) return SWMHD_EINVAL;
    const size_t count = (size_t)Hy * (size_t)sy;   // Hy full rows (the pitch padding of the last row travels too: harmless)
    if (r->hub) {
        if (nf > LoopHub::MAXF) return SWMHD_EINVAL;
        const char *ss[LoopHub::MAXF], *sn[LoopHub::MAXF]; char *rs[LoopHub::MAXF], *rn[LoopHub::MAXF];
        for (int f = 0; f < nf; ++f) {
            T *p = fields[f];
            if (!p) return SWMHD_EINVAL;
            ss[f] = (const char *)(p + (size_t)Hy * sy); sn[f] = (const char *)(p + (size_t)Ny * sy);
            rs[f] = (char *)p; rn[f] = (char *)(p + (size_t)(Ny + Hy) * sy);
        }
        return loopback_exchange(r, ss, sn, rs, rn, nf, count * sizeof(T), s);
    }
    ncclResult_t rc = r->api.GroupStart();
    if (rc != ncclSuccess) return fail(r, "ncclGroupStart", rc);
    for (int f = 0; f < nf; ++f) {
        T *p = fields[f];
        if (!p) { r->api.GroupEnd(); return SWMHD_EINVAL; }
        T *send_s = p + (size_t)Hy * sy, *send_n = p + (size_t)Ny * sy;
        T *recv_s = p, *recv_n = p + (size_t)(Ny + Hy) * sy;
        if ((rc = r->api.Send(send_n, count, nccl_type<T>(), r->north, r->comm, s)) != ncclSuccess) break;
        if ((rc = r->api.Recv(recv_s, count, nccl_type<T>(), r->south, r->comm, s)) != ncclSuccess) break;
        if ((rc = r->api.Send(send_s, count, nccl_type<T>(), r->south, r->comm, s)) != ncclSuccess) break;
        if ((rc = r->api.Recv(recv_n, count, nccl_type<T>(), r->north, r->comm, s)) != ncclSuccess) break;
    }
    ncclResult_t rc2 = r->api.GroupEnd();
    if (rc != ncclSuccess) return fail(r, "ncclSend/ncclRecv", rc);
    if (rc2 != ncclSuccess) return fail(r, "ncclGroupEnd", rc2);
    return SWMHD_OK;
}

template <typename T> struct Api;
template <> struct Api<double> {
    static constexpr auto stage = swmhd_tendencies_rk3_f64;
    static constexpr auto halo = swmhd_fill_halo_periodic_multi_f64;
};
template <> struct Api<float> {
    static constexpr auto stage = swmhd_tendencies_rk3_f32;
    static constexpr auto halo = swmhd_fill_halo_periodic_multi_f32;
};

// nsteps RK3 steps of one y-slab.  Two schedules: the deep-halo one (Hy >= 9, x wrapped on read: ONE exchange per step, described
// where it is implemented below) and the per-stage one.  Per stage (X = current state, Y = the other buffer set):
//   main stream : rows [Hy, Ny-Hy) of X -> Y          (need no remote data; the exchange of X is still in flight)
//   comm stream : ... exchange of X ... ; rows [0,Hy) and [Ny-Hy,Ny) of X -> Y     (one launch, queued behind the exchange)
//   x wrapped on read : comm: exchange of Y straight after the strips ; comm waits for main's interior ; main waits for the strips
//   x halos in memory : main: wait(comm) ; x-halo fill of Y ; record ; comm waits ; comm: exchange of Y
// Either way the exchange of Y overlaps the next stage's interior rows.  A thin slab is bound by the chain exchange -> strips ->
// exchange on the comm stream (tools/ring_rehearsal.py), which is why that chain has no hop through the main stream.
// The first stage of the first call finds no exchange in flight and the caller's halos current: it runs all rows at once.
template <typename T>
int ring_step(swmhd_ring *r, T *const *q, T *const *q_alt, T *const *Ga, T *const *Gb, int Nx, int Ny, int Hx, int Hy, int64_t sy,
              T dx, T dy, T grav, T fcor, int formulation, int lorentz, T dt, int nsteps, int flags, int *state_in_alt,
              void *stream) {
    if (!r || !q || !q_alt || !Ga || !Gb || nsteps < 0) return SWMHD_EINVAL;
    if (flags & SWMHD_WRAP_Y) return SWMHD_EINVAL;   // y images belong to the neighbours (x may be wrapped on read: no x-halo kernel then)
    if (Ny < 2 * Hy + 1) return SWMHD_EINVAL;                          // a slab needs interior rows between its two strips
    const T gam[3] = {T(8.0 / 15.0), T(5.0 / 12.0), T(3.0 / 4.0)};
    const T zet[3] = {T(0), T(-17.0 / 60.0), T(-5.0 / 12.0)};
    hipStream_t s = (hipStream_t)stream, c = r->comm_stream;
    T *cur[4], *alt[4], *gn[4], *gm[4];
    for (int f = 0; f < 4; ++f) {
        if (!q[f] || !q_alt[f] || !Ga[f] || !Gb[f]) return SWMHD_EINVAL;
        cur[f] = q[f]; alt[f] = q_alt[f]; gn[f] = Ga[f]; gm[f] = Gb[f];
    }
    if (r->pending && r->pending != (const void *)cur[0]) {   // an exchange of some other state is in flight: drain it first
        hipError_t e = hipEventRecord(r->ev_comm, c);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, r->ev_comm, 0);
        if (e != hipSuccess) return hipfail(r, "join", e);
        r->pending = nullptr;
    }
    int swaps = 0;
    hipError_t e;
    const bool from_state = !(flags & SWMHD_STRICT);   // fast builds: no tendency store in the first stage (see step_common in swmhd_api.hip)
    // whatever the caller enqueued on its stream so far precedes everything this call puts on the comm stream
    if ((e = hipEventRecord(r->ev_main, s)) != hipSuccess) return hipfail(r, "record", e);
    if ((e = hipStreamWaitEvent(c, r->ev_main, 0)) != hipSuccess) return hipfail(r, "wait", e);
    // Error exit from the middle of a step: whatever was enqueued stays enqueued, so order the caller's stream behind the comm
    // stream and forget the in-flight exchange -- `pending` must never describe an exchange that was not (fully) issued -- and
    // tell the caller which buffer set holds the newest completed stage.
    auto bail = [&](int code) {
        if (hipEventRecord(r->ev_comm, c) == hipSuccess) (void)hipStreamWaitEvent(s, r->ev_comm, 0);
        r->pending = nullptr;
        if (state_in_alt) *state_in_alt = swaps & 1;
        return code;
    };
    if ((flags & SWMHD_WRAP_X) && Hy >= 9 && Ny >= 32) {
        // ---- deep-halo schedule: ONE exchange per step instead of one per stage ------------------------------------------------
        // With Hy >= 9 a slab evaluates the rows of its neighbours it needs for stages 2 and 3 itself (redundantly: 18 extra rows
        // per step) from the 9 halo rows exchanged once per step.  Rows per stage (north side mirrored):
        //     stage 1   interior [3, Ny-3)     boundary [-6, 3)      stage 2   interior [9, Ny-9)    boundary [-3, 9)
        //     stage 3   interior [12, Ny-12)   boundary [0, 12)
        // Interior launches (main stream) read only what earlier interior launches of the same step wrote -- plus, for stage 1,
        // the boundary rows of the previous step's last stage: ONE wait of the main stream per step.  Boundary launches (comm
        // stream, behind the exchange) read rows of the previous interior launch: the comm stream waits twice, off the critical
        // path.  Stage 2's interior starts at row 9, not 6, because it overwrites the buffer whose rows [0, 9) the exchange in
        // flight is still sending.  A thin slab (strong scaling) is then bound by its interior launches, not by the chain
        // exchange -> strips -> exchange of the per-stage schedule below (tools/ring_rehearsal.py).
        const int ilo[3] = {3, 9, 12}, blo[3] = {-6, -3, 0};
        // boundary zones of a wide slab take the row-marching kernel too (both zones in one launch, a few dozen workgroups beside the
        // interior launch); the LDS-tiled kernel re-loads a 10-row halo per 4-row tile and costs ten times as much per row
        const int bflags = flags | ((!(flags & (SWMHD_STRICT | SWMHD_TILE_KERNEL | SWMHD_MARCH_KERNEL)) && Nx >= 1024) ? SWMHD_MARCH_KERNEL : 0);
        for (int n = 0; n < nsteps; ++n) {
            for (int st = 0; st < 3; ++st) {
                const T *cq[4] = {cur[0], cur[1], cur[2], cur[3]};
                const T *cgm[4] = {gm[0], gm[1], gm[2], gm[3]};
                const bool fs = from_state && st == 1;        // G- of the second stage from the two states (swmhd.h SWMHD_GM_IS_PREV_STATE)
                if (fs) for (int f = 0; f < 4; ++f) cgm[f] = alt[f];
                const T *const *pgm = st == 0 ? nullptr : cgm;
                const int store = st == 1 ? 1 : (st == 0 && !from_state ? 1 : 0);
                const T zeta_st = fs ? zet[1] / gam[0] : zet[st];
                const int sflags = fs ? SWMHD_GM_IS_PREV_STATE : 0;
                const int jb = ilo[st], je = Ny - ilo[st];
                const bool timed = r->t0.size() < r->tcap;
                hipEvent_t a = nullptr, b = nullptr;
                if (timed) {
                    if ((e = hipEventCreateWithFlags(&a, hipEventDisableSystemFence)) != hipSuccess) return bail(hipfail(r, "hipEventCreate", e));
                    if ((e = hipEventCreateWithFlags(&b, hipEventDisableSystemFence)) != hipSuccess) { (void)hipEventDestroy(a); return bail(hipfail(r, "hipEventCreate", e)); }
                    (void)hipEventRecord(a, s);
                }
                int rc = Api<T>::stage(cq, alt, gn, pgm, Nx, Ny, Hx, Hy, sy, dx, dy, grav, fcor, formulation, lorentz, dt, gam[st], zeta_st,
                                       store, jb, je, flags | SWMHD_LEAVE_ROOM | sflags, (void *)s);
                if (timed) {
                    if (rc) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
                    else { (void)hipEventRecord(b, s); r->t0.push_back(a); r->t1.push_back(b); r->trows.push_back(je - jb); }
                }
                if (rc) return bail(rc);
                if (st < 2) {   // the next boundary launch reads rows of this interior launch
                    if ((e = hipEventRecord(r->ev_main, s)) != hipSuccess) return bail(hipfail(r, "record", e));
                }
                // boundary rows of this stage, both sides in one launch, behind the exchange (stage 1) / the previous boundary launch
                if ((rc = swmhd::tendencies_rk3_two_ranges<T>(cq, alt, gn, pgm, Nx, Ny, Hx, Hy, (long)sy, dx, dy, grav, fcor, formulation,
                                                              lorentz, dt, gam[st], zeta_st, store, blo[st], ilo[st], Ny - ilo[st],
                                                              Ny - blo[st], bflags | sflags, (void *)c)))
                    return bail(rc);
                if (st < 2) {
                    if ((e = hipStreamWaitEvent(c, r->ev_main, 0)) != hipSuccess) return bail(hipfail(r, "wait", e));
                }
                for (int f = 0; f < 4; ++f) { T *t = cur[f]; cur[f] = alt[f]; alt[f] = t; t = gn[f]; gn[f] = gm[f]; gm[f] = t; }
                ++swaps;
            }
            // end of the step: the main stream's next interior launch reads the last boundary rows; the exchange of the new state
            // (its 9 edge rows are exactly those boundary rows) follows them on the comm stream
            r->pending = nullptr;
            if ((e = hipEventRecord(r->ev_comm, c)) != hipSuccess) return bail(hipfail(r, "record", e));
            if ((e = hipStreamWaitEvent(s, r->ev_comm, 0)) != hipSuccess) return bail(hipfail(r, "wait", e));
            if (int rc = exchange<T>(r, cur, 4, Nx, Ny, Hx, Hy, sy, c)) return bail(rc);
            r->pending = cur[0];
        }
        if (state_in_alt) *state_in_alt = swaps & 1;
        return SWMHD_OK;
    }
    for (int n = 0; n < nsteps; ++n)
        for (int st = 0; st < 3; ++st) {
            const T *cq[4] = {cur[0], cur[1], cur[2], cur[3]};
            const T *cgm[4] = {gm[0], gm[1], gm[2], gm[3]};
            const bool fs = from_state && st == 1;
            if (fs) for (int f = 0; f < 4; ++f) cgm[f] = alt[f];
            const T *const *pgm = st == 0 ? nullptr : cgm;
            const int store = st == 1 ? 1 : (st == 0 && !from_state ? 1 : 0);
            const T zeta_st = fs ? zet[1] / gam[0] : zet[st];
            const int sflags = fs ? SWMHD_GM_IS_PREV_STATE : 0;
            auto run = [&](int j0, int j1, hipStream_t on, int extra = 0) {
                return Api<T>::stage(cq, alt, gn, pgm, Nx, Ny, Hx, Hy, sy, dx, dy, grav, fcor, formulation, lorentz, dt, gam[st],
                                     zeta_st, store, j0, j1, flags | extra | sflags, (void *)on);
            };
            int rc;
            const bool split = r->pending != nullptr;
            const int jb = split ? Hy : 0, je = split ? Ny - Hy : Ny;
            const bool timed = r->t0.size() < r->tcap;
            hipEvent_t a = nullptr, b = nullptr;
            if (timed) {
                // (timing-only events: without the system-scope fence a default event performs when it is recorded -- with the comm
                //  stream's kernels running beside the interior launch that fence cost 16 % of the step, 1.53 vs 1.32 ms)
                if ((e = hipEventCreateWithFlags(&a, hipEventDisableSystemFence)) != hipSuccess) return bail(hipfail(r, "hipEventCreate", e));
                if ((e = hipEventCreateWithFlags(&b, hipEventDisableSystemFence)) != hipSuccess) { (void)hipEventDestroy(a); return bail(hipfail(r, "hipEventCreate", e)); }
                (void)hipEventRecord(a, s);
            }
            // (interior rows: leave a few workgroup slots free, or the exchange and the strips could not start before it ends)
            rc = run(jb, je, s, split ? SWMHD_LEAVE_ROOM : 0);
            if (timed) {
                if (rc) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
                else { (void)hipEventRecord(b, s); r->t0.push_back(a); r->t1.push_back(b); r->trows.push_back(je - jb); }
            }
            if (rc) return bail(rc);
            if (split) {   // both strips in one launch: they sit on the exchange -> strips -> exchange chain that bounds a thin slab
                if ((rc = swmhd::tendencies_rk3_two_ranges<T>(cq, alt, gn, pgm, Nx, Ny, Hx, Hy, (long)sy, dx, dy, grav, fcor, formulation,
                                                              lorentz, dt, gam[st], zeta_st, store, 0, Hy, Ny - Hy, Ny, flags | sflags, (void *)c)))
                    return bail(rc);
                if ((e = hipEventRecord(r->ev_comm, c)) != hipSuccess) return bail(hipfail(r, "record", e));
                if ((e = hipStreamWaitEvent(s, r->ev_comm, 0)) != hipSuccess) return bail(hipfail(r, "wait", e));
            }
            for (int f = 0; f < 4; ++f) { T *t = cur[f]; cur[f] = alt[f]; alt[f] = t; t = gn[f]; gn[f] = gm[f]; gm[f] = t; }
            ++swaps;
            r->pending = nullptr;   // the exchange of the OLD state has been consumed; none of the new state is in flight yet
            if (split && (flags & SWMHD_WRAP_X)) {
                // The rows the exchange sends are exactly the strips' output and (x wrapped on read) no x-halo kernel touches them:
                // the exchange of the new state follows the strips on the comm stream directly, without a round trip through the
                // main stream; only the NEXT stage's strips wait for this stage's interior rows.
                if ((rc = exchange<T>(r, cur, 4, Nx, Ny, Hx, Hy, sy, c))) return bail(rc);
                r->pending = cur[0];
                if ((e = hipEventRecord(r->ev_main, s)) != hipSuccess) return bail(hipfail(r, "record", e));
                if ((e = hipStreamWaitEvent(c, r->ev_main, 0)) != hipSuccess) return bail(hipfail(r, "wait", e));
                continue;
            }
            if (!(flags & SWMHD_WRAP_X) && (rc = Api<T>::halo(cur, 4, Nx, Ny, Hx, Hy, sy, SWMHD_HALO_X, (void *)s))) return bail(rc);
            if ((e = hipEventRecord(r->ev_main, s)) != hipSuccess) return bail(hipfail(r, "record", e));
            if ((e = hipStreamWaitEvent(c, r->ev_main, 0)) != hipSuccess) return bail(hipfail(r, "wait", e));
            if ((rc = exchange<T>(r, cur, 4, Nx, Ny, Hx, Hy, sy, c))) return bail(rc);
            r->pending = cur[0];
        }
    if (state_in_alt) *state_in_alt = swaps & 1;
    return SWMHD_OK;
}

}  // namespace

extern "C" {

int swmhd_ring_available(const char *rccl_path) {
    RcclApi api;
    std::string err;
    return load_rccl(rccl_path, api, err) ? SWMHD_OK : SWMHD_ENOTSUP;
}

int swmhd_ring_unique_id(const char *rccl_path, void *id128) {
    if (!id128) return SWMHD_EINVAL;
    RcclApi api;
    std::string err;
    if (!load_rccl(rccl_path, api, err)) return SWMHD_ENOTSUP;
    ncclUniqueId id;
    if (api.GetUniqueId(&id) != ncclSuccess) return SWMHD_ECOMM;
    static_assert(sizeof(id) == SWMHD_RING_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof(id));
    return SWMHD_OK;   // (the dlopen handle is deliberately kept: the library stays loaded for swmhd_ring_create)
}

int swmhd_ring_create(swmhd_ring **out, const char *rccl_path, int nranks, int rank, const void *id128) {
    if (!out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return SWMHD_EINVAL;
    swmhd_ring *r = new (std::nothrow) swmhd_ring;
    if (!r) return SWMHD_EINVAL;
    if (!load_rccl(rccl_path, r->api, r->err)) { delete r; return SWMHD_ENOTSUP; }
    r->nranks = nranks; r->rank = rank;
    r->south = (rank + nranks - 1) % nranks;
    r->north = (rank + 1) % nranks;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    if (r->api.CommInitRank(&r->comm, nranks, id, rank) != ncclSuccess) { delete r; return SWMHD_ECOMM; }
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // hi = numerically lowest = highest priority
    hipError_t e = hipStreamCreateWithPriority(&r->comm_stream, hipStreamNonBlocking, hi);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_main, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_comm, hipEventDisableTiming);
    if (e != hipSuccess) { swmhd_ring_destroy(r); return -(int)e; }
    *out = r;
    return SWMHD_OK;
}

int swmhd_ring_create_loopback(swmhd_ring **out, int nranks, double timeout_s) {
    if (!out || nranks < 1 || nranks > 64) return SWMHD_EINVAL;
    std::shared_ptr<LoopHub> hub;
    try { hub = std::make_shared<LoopHub>(nranks); } catch (...) { return SWMHD_EINVAL; }
    if (timeout_s > 0) hub->timeout_s = timeout_s;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    for (int k = 0; k < nranks; ++k) out[k] = nullptr;
    for (int k = 0; k < nranks; ++k) {
        swmhd_ring *r = new (std::nothrow) swmhd_ring;
        hipError_t e = r ? hipSuccess : hipErrorOutOfMemory;
        if (r) {
            r->hub = hub; r->nranks = nranks; r->rank = k;
            r->south = (k + nranks - 1) % nranks; r->north = (k + 1) % nranks;
            e = hipStreamCreateWithPriority(&r->comm_stream, hipStreamNonBlocking, hi);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_main, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_comm, hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            if (r) swmhd_ring_destroy(r);
            for (int j = 0; j < k; ++j) { swmhd_ring_destroy(out[j]); out[j] = nullptr; }
            return -(int)e;
        }
        out[k] = r;
    }
    return SWMHD_OK;
}

int swmhd_ring_destroy(swmhd_ring *r) {
    if (!r) return SWMHD_OK;
    if (r->comm_stream) (void)hipStreamSynchronize(r->comm_stream);
    for (auto ev : r->t0) (void)hipEventDestroy(ev);
    for (auto ev : r->t1) (void)hipEventDestroy(ev);
    if (r->comm) r->api.CommDestroy(r->comm);
    if (r->ev_main) (void)hipEventDestroy(r->ev_main);
    if (r->ev_comm) (void)hipEventDestroy(r->ev_comm);
    if (r->comm_stream) (void)hipStreamDestroy(r->comm_stream);
    delete r;
    return SWMHD_OK;
}

const char *swmhd_ring_last_error(const swmhd_ring *r) { return r ? r->err.c_str() : "null ring"; }

void *swmhd_ring_comm_stream(const swmhd_ring *r) { return r ? (void *)r->comm_stream : nullptr; }

int swmhd_ring_join(swmhd_ring *r, void *stream) {
    if (!r) return SWMHD_EINVAL;
    if (!r->pending) return SWMHD_OK;
    hipError_t e = hipEventRecord(r->ev_comm, r->comm_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)stream, r->ev_comm, 0);
    if (e != hipSuccess) return hipfail(r, "join", e);
    r->pending = nullptr;
    return SWMHD_OK;
}

int swmhd_ring_time_launches(swmhd_ring *r, int max_launches) {
    if (!r || max_launches < 0) return SWMHD_EINVAL;
    for (auto ev : r->t0) (void)hipEventDestroy(ev);
    for (auto ev : r->t1) (void)hipEventDestroy(ev);
    r->t0.clear(); r->t1.clear(); r->trows.clear();
    r->tcap = (size_t)max_launches;
    return SWMHD_OK;
}

int swmhd_ring_launch_times(swmhd_ring *r, float *ms, int *rows, int capacity) {
    if (!r || !ms || !rows) return -1;
    int n = 0;
    for (size_t i = 0; i < r->t0.size() && n < capacity; ++i) {
        if (hipEventSynchronize(r->t1[i]) != hipSuccess) break;
        if (hipEventElapsedTime(&ms[n], r->t0[i], r->t1[i]) != hipSuccess) break;
        rows[n++] = r->trows[i];
    }
    return n;
}

#define SWMHD_DEF_RING(sfx, T)                                                                                                   \
    int swmhd_ring_exchange_y_##sfx(swmhd_ring *r, T *const *fields, int nfields, int Nx, int Ny, int Hx, int Hy, int64_t sy,    \
                                    void *stream) {                                                                              \
        return exchange<T>(r, fields, nfields, Nx, Ny, Hx, Hy, sy, (hipStream_t)stream);                                         \
    }                                                                                                                            \
    int swmhd_ring_step_rk3_##sfx(swmhd_ring *r, T *const *q, T *const *q_alt, T *const *Ga, T *const *Gb, int Nx, int Ny,       \
                                  int Hx, int Hy, int64_t sy, T dx, T dy, T g, T f, int formulation, int lorentz, T dt,          \
                                  int nsteps, int flags, int *state_in_alt, void *stream) {                                      \
        return ring_step<T>(r, q, q_alt, Ga, Gb, Nx, Ny, Hx, Hy, sy, dx, dy, g, f, formulation, lorentz, dt, nsteps, flags,      \
                            state_in_alt, stream);                                                                               \
    }

SWMHD_DEF_RING(f64, double)
SWMHD_DEF_RING(f32, float)

}  // extern "C"
