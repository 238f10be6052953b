// Several GPUs behind the `lbm` driver: the lattice cut into y-slabs, one per device, all driven by
// this one process through the row-range entry points of the C ABI.
//
// Takes the place of run_simulation_mpi (JulianKarrer/lbm src/main.cpp:92-401): the reference cuts
// the lattice over MPI ranks (main.cpp:96-144), packs edge populations, stages them through the host
// and overlaps the exchange with the inner update (main.cpp:217-398).  Here:
//   * slab d owns rows [y_below, y_below + n) of the global lattice (remainder rows to the lowest
//     slabs, main.cpp:133-141) in arrays [9][n + 2G][NX]: G ghost rows below and above hold copies of
//     the neighbouring slabs' edge rows, all nine populations;
//   * a block of k <= G time steps per slab is {interior rows [2G, n) by one multi-step launch on the
//     device's compute stream; the two edge regions [G, 2G) and [n, n + G) by two small launches on
//     the device's high-priority edge stream; then two strided device-to-device copies that put the
//     new edge rows into the neighbours' ghost rows (peer copies over xGMI when the neighbour lives
//     on another device)}.  The interior never reads a ghost row, so it overlaps the exchange;
//   * ping-pong events order block b's interior after block b-1's edges, and block b's edges after
//     block b-1's interior and after the neighbours' exchange of block b-1.
// Slabs are initialised from their global row index, so any number of slabs reproduces the
// single-device populations bit for bit.  More slabs than devices are dealt round robin (slab d on
// device d mod #devices): that is how the schedule is tested on one GPU.
#pragma once

struct Slab {
    int device = 0;
    UINT n = 0, y_below = 0;       // rows owned, global index of the first one
    void *f = nullptr, *buf = nullptr, *vel = nullptr;
    hipEvent_t interior_done[2] = {nullptr, nullptr}, edges_done[2] = {nullptr, nullptr}, xfer_done[2] = {nullptr, nullptr};
};

struct SlabRun {
    std::vector<Slab> slabs;
    std::vector<hipStream_t> compute, edge;   // per device
    UINT G = 1;                               // ghost rows per side
    UINT block = 0;                           // blocks run so far (parity of the ping-pong events)
    bool periodic = true;
};

static inline size_t slab_plane(const Options &o, const Slab &s, UINT G) { return (size_t)(s.n + 2 * G) * o.nx; }

// rows [y_lo, y_hi) of one slab, k steps, on `stream`
static void slab_sweep(const Run &r, const SlabRun &sr, const Slab &s, UINT y_lo, UINT y_hi, int k, hipStream_t stream)
{
    if (y_lo >= y_hi) return;
    const Options &o = r.o;
    const UINT rows = s.n + 2 * sr.G;
    int st;
    if (o.scene == LID_DRIVEN) {
        // the reference's lid scene always pushes (main.cpp:72-74)
        st = lbm_push_lid_fused(s.f, s.buf, kPrec, o.nx, rows, o.omega, o.rho, o.u, y_lo, y_hi,
                                (int64_t)s.y_below - (int64_t)sr.G, o.ny, k, stream);
    } else {
        st = o.push ? lbm_push_periodic_fused(s.f, s.buf, kPrec, o.nx, rows, o.omega, y_lo, y_hi, 0, k, stream)
                    : lbm_pull_periodic_fused(s.f, s.buf, kPrec, o.nx, rows, o.omega, y_lo, y_hi, 0, k, stream);
    }
    die_lbm(st, "slab sweep");
}

// rows [src_row, src_row + G) of all nine populations of `src` -> rows [dst_row, dst_row + G) of `dst`
static void slab_copy_rows(const Run &r, const SlabRun &sr, const Slab &src, const void *src_arr, UINT src_row,
                           const Slab &dst, void *dst_arr, UINT dst_row, hipStream_t stream)
{
    const Options &o = r.o;
    const size_t width = (size_t)sr.G * o.nx * sizeof(FLT);
    const char *sp = (const char *)src_arr + (size_t)src_row * o.nx * sizeof(FLT);
    char *dp = (char *)dst_arr + (size_t)dst_row * o.nx * sizeof(FLT);
    die_hip(hipMemcpy2DAsync(dp, slab_plane(o, dst, sr.G) * sizeof(FLT), sp, slab_plane(o, src, sr.G) * sizeof(FLT), width, 9,
                             hipMemcpyDeviceToDevice, stream),
            "halo copy");
}

// The edge rows of array `which` (0: f, 1: buf) of every slab go to the neighbours' ghost rows of the
// same array, on the edge streams; records xfer_done[parity].
static void slab_exchange(const Run &r, SlabRun &sr, int which, unsigned parity)
{
    const size_t N = sr.slabs.size();
    for (size_t d = 0; d < N; ++d) {
        Slab &s = sr.slabs[d];
        die_hip(hipSetDevice(s.device), "hipSetDevice");
        hipStream_t es = sr.edge[s.device];
        const void *mine = which ? s.buf : s.f;
        if (sr.periodic || d > 0) {            // my lowest owned rows -> the lower neighbour's upper ghost rows
            Slab &lo = sr.slabs[(d + N - 1) % N];
            slab_copy_rows(r, sr, s, mine, sr.G, lo, which ? lo.buf : lo.f, lo.n + sr.G, es);
        }
        if (sr.periodic || d + 1 < N) {        // my highest owned rows -> the upper neighbour's lower ghost rows
            Slab &up = sr.slabs[(d + 1) % N];
            slab_copy_rows(r, sr, s, mine, s.n, up, which ? up.buf : up.f, 0, es);
        }
        die_hip(hipEventRecord(s.xfer_done[parity], es), "event");
    }
}

static void slab_block(const Run &r, SlabRun &sr, int k)
{
    const size_t N = sr.slabs.size();
    const unsigned p = sr.block & 1u, q = p ^ 1u;
    const UINT G = sr.G;
    // interiors first: nothing of a block's edge work may sit in front of the long kernels in a queue
    for (Slab &s : sr.slabs) {
        die_hip(hipSetDevice(s.device), "hipSetDevice");
        hipStream_t cs = sr.compute[s.device];
        if (sr.block > 0) die_hip(hipStreamWaitEvent(cs, s.edges_done[q], 0), "wait");
        slab_sweep(r, sr, s, 2 * G, s.n, k, cs);
        die_hip(hipEventRecord(s.interior_done[p], cs), "event");
    }
    for (size_t d = 0; d < N; ++d) {
        Slab &s = sr.slabs[d];
        die_hip(hipSetDevice(s.device), "hipSetDevice");
        hipStream_t es = sr.edge[s.device];
        // initialisation / the previous block's interior, and the neighbours' previous exchange
        die_hip(hipStreamWaitEvent(es, s.interior_done[q], 0), "wait");
        die_hip(hipStreamWaitEvent(es, sr.slabs[(d + N - 1) % N].xfer_done[q], 0), "wait");
        die_hip(hipStreamWaitEvent(es, sr.slabs[(d + 1) % N].xfer_done[q], 0), "wait");
        slab_sweep(r, sr, s, G, 2 * G, k, es);
        slab_sweep(r, sr, s, s.n, s.n + G, k, es);
        die_hip(hipEventRecord(s.edges_done[p], es), "event");
    }
    slab_exchange(r, sr, 1, p);
    for (Slab &s : sr.slabs) std::swap(s.f, s.buf);
    ++sr.block;
}

static void slab_wait_all(SlabRun &sr)
{
    for (size_t dev = 0; dev < sr.compute.size(); ++dev) {
        if (!sr.compute[dev]) continue;
        die_hip(hipSetDevice((int)dev), "hipSetDevice");
        die_hip(hipStreamSynchronize(sr.compute[dev]), "sync");
        die_hip(hipStreamSynchronize(sr.edge[dev]), "sync");
    }
}

// output() of io.cpp:177-294 over the slabs: the sample is assembled on the host in global row order
static void slab_fetch_output(Run &r, SlabRun &sr)
{
    const Options &o = r.o;
    if (o.output == NONE) return;
    slab_wait_all(sr);
    const UINT G = sr.G;
    if (o.output == MAX_VEL) {
        double m = -1e30;       // io.cpp:184
        for (Slab &s : sr.slabs) {
            die_hip(hipSetDevice(s.device), "hipSetDevice");
            double v = 0;
            die_lbm(lbm_max_velocity(s.f, kPrec, o.nx, s.n + 2 * G, G, G + s.n, sr.compute[s.device], &v), "max velocity");
            if (v > m) m = v;   // the MPI_MAX reduction of io.cpp:213
        }
        r.max_vel = m;
        return;
    }
    const size_t plane = (size_t)o.nx * o.ny;
    r.host.resize(o.output == VEL_FIELD ? 2 * plane : plane);
    for (Slab &s : sr.slabs) {
        die_hip(hipSetDevice(s.device), "hipSetDevice");
        hipStream_t cs = sr.compute[s.device];
        const size_t part = (size_t)s.n * o.nx;
        if (o.output == VEL_MAGS) {
            die_lbm(lbm_velocity_magnitudes(s.f, s.vel, kPrec, o.nx, s.n + 2 * G, G, G + s.n, cs), "velocity magnitudes");
            die_hip(hipMemcpyAsync(r.host.data() + (size_t)s.y_below * o.nx, s.vel, part * sizeof(FLT), hipMemcpyDeviceToHost, cs), "copy");
        } else {
            die_lbm(lbm_compute_velocities(s.f, s.vel, kPrec, o.nx, s.n + 2 * G, G, G + s.n, cs), "velocities");
            die_hip(hipMemcpyAsync(r.host.data() + (size_t)s.y_below * o.nx, s.vel, part * sizeof(FLT), hipMemcpyDeviceToHost, cs), "copy");
            die_hip(hipMemcpyAsync(r.host.data() + plane + (size_t)s.y_below * o.nx, (const FLT *)s.vel + part, part * sizeof(FLT),
                                   hipMemcpyDeviceToHost, cs), "copy");
        }
    }
    slab_wait_all(sr);
}

// The time loop of main.cpp:217-398 over `nslabs` slabs.
static bool run_simulation_slabs(Run &r, unsigned nslabs)
{
    const Options &o = r.o;
    int ndev = 0;
    die_hip(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
    if (ndev < 1) { std::cerr << "no HIP device" << std::endl; return false; }
    if ((UINT)nslabs > o.ny) { std::cerr << "more slabs than rows" << std::endl; return false; }
    SlabRun sr;
    sr.periodic = o.scene == SHEAR_WAVE;
    sr.slabs.resize(nslabs);
    sr.compute.assign((size_t)ndev, nullptr);
    sr.edge.assign((size_t)ndev, nullptr);
    const UINT base = o.ny / nslabs, rem = o.ny % nslabs;
    // ghost depth = steps per block: what the library plans for a slab of the smallest height, as long
    // as a slab holds two edge regions of that depth
    int g = lbm_plan_steps_scene(kPrec, o.nx, base, (uint64_t)1 << 30, o.scene == LID_DRIVEN);
    if (g < 1) die_lbm(LBM_ERR_INVALID_ARGUMENT, "planning the slab blocks");
    while (g > 1 && base < 2 * (UINT)g) --g;
    sr.G = (UINT)g;
    if (base < 2 * sr.G) { std::cerr << "slabs of " << base << " rows are too thin: at least 2 rows per slab" << std::endl; return false; }
    const UINT G = sr.G;

    std::cerr << "running on" << o.nx << "x" << o.ny << std::endl;
    std::cerr << nslabs << " y-slab(s) of " << base << (rem ? "-" + std::to_string(base + 1) : std::string()) << " rows on "
              << (ndev < (int)nslabs ? ndev : (int)nslabs) << " device(s), " << G << " ghost row(s) per edge exchanged every "
              << G << " step(s)" << std::endl;
    if (o.tx == 64 || o.tx == 128 || o.tx == 256) lbm_set_launch_shape((unsigned)o.tx, 0);

    for (unsigned d = 0; d < nslabs; ++d) {
        Slab &s = sr.slabs[d];
        s.device = (int)(d % (unsigned)ndev);
        s.n = base + (d < rem ? 1 : 0);
        s.y_below = d * base + (d < rem ? d : rem);
        die_hip(hipSetDevice(s.device), "hipSetDevice");
        if (!sr.compute[s.device]) {
            int lo = 0, hi = 0;
            die_hip(hipDeviceGetStreamPriorityRange(&lo, &hi), "priority range");
            die_hip(hipStreamCreateWithFlags(&sr.compute[s.device], hipStreamNonBlocking), "stream");
            // one high-priority stream per device: the edge sweeps and the copies are short and everybody waits for them
            die_hip(hipStreamCreateWithPriority(&sr.edge[s.device], hipStreamNonBlocking, hi), "stream");
        }
        const size_t bytes = 9 * slab_plane(o, s, G) * sizeof(FLT);
        die_hip(hipMalloc(&s.f, bytes), "allocating f");
        die_hip(hipMalloc(&s.buf, bytes), "allocating buf");
        if (o.output == VEL_MAGS || o.output == VEL_FIELD)
            die_hip(hipMalloc(&s.vel, (size_t)2 * s.n * o.nx * sizeof(FLT)), "allocating vel");
        for (int e = 0; e < 2; ++e) {
            die_hip(hipEventCreateWithFlags(&s.interior_done[e], hipEventDisableTiming), "event");
            die_hip(hipEventCreateWithFlags(&s.edges_done[e], hipEventDisableTiming), "event");
            die_hip(hipEventCreateWithFlags(&s.xfer_done[e], hipEventDisableTiming), "event");
        }
    }
    // neighbours on different devices copy into each other's memory
    for (unsigned d = 0; d < nslabs; ++d) {
        const int a = sr.slabs[d].device, b = sr.slabs[(d + 1) % nslabs].device;
        if (a == b) continue;
        for (int dir = 0; dir < 2; ++dir) {
            const int from = dir ? b : a, to = dir ? a : b;
            int can = 0;
            die_hip(hipDeviceCanAccessPeer(&can, from, to), "peer query");
            if (!can) { std::cerr << "device " << from << " cannot access device " << to << std::endl; return false; }
            die_hip(hipSetDevice(from), "hipSetDevice");
            const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) die_hip(e, "enabling peer access");
            (void)hipGetLastError();
        }
    }
    // initial state from the global row index (init.cpp:22-26 with the right offset; main.cpp:141-144
    // gets it wrong and is not reproduced); ghost rows start defined
    for (Slab &s : sr.slabs) {
        die_hip(hipSetDevice(s.device), "hipSetDevice");
        hipStream_t cs = sr.compute[s.device];
        const size_t bytes = 9 * slab_plane(o, s, G) * sizeof(FLT);
        die_hip(hipMemsetAsync(s.f, 0, bytes, cs), "memset");
        die_hip(hipMemsetAsync(s.buf, 0, bytes, cs), "memset");
        if (o.scene == LID_DRIVEN) die_lbm(lbm_init_rest(s.f, kPrec, o.nx, s.n + 2 * G, o.rho, cs), "init_rest");
        else die_lbm(lbm_init_shearwave(s.f, kPrec, o.nx, s.n + 2 * G, o.rho, o.u, s.y_below, o.ny, (unsigned)G, cs), "init_shearwave");
        die_hip(hipStreamSynchronize(cs), "sync");
    }
    // ghost rows of f from the neighbours' initial edge rows; the first block's edges wait for it
    slab_exchange(r, sr, 0, 1);
    for (Slab &s : sr.slabs) {
        die_hip(hipSetDevice(s.device), "hipSetDevice");
        die_hip(hipEventRecord(s.interior_done[1], sr.compute[s.device]), "event");
    }

    const auto t0 = std::chrono::high_resolution_clock::now();
    bool pending = false;
    unsigned queued = 0;
    for (UINT t = 0; t < o.steps;) {
        if (o.out_every > 0 && t % o.out_every == 0) {
            if (pending && !emit_output(r)) return false;
            slab_fetch_output(r, sr);
            pending = true;
            queued = 0;
        }
        UINT until = o.steps;
        if (o.out_every > 0) {
            const UINT next_out = (t / o.out_every + 1) * o.out_every;
            if (next_out < until) until = next_out;
        }
        // the fewest blocks of at most G steps, as equal as possible
        const UINT left = until - t, blocks = (left + G - 1) / G;
        const int k = (int)((left + blocks - 1) / blocks);
        slab_block(r, sr, k);
        t += (UINT)k;
        if (pending && (t >= until || ++queued >= 64)) {
            if (!emit_output(r)) return false;
            pending = false;
        }
    }
    if (pending && !emit_output(r)) return false;
    slab_wait_all(sr);
    r.sweep_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::high_resolution_clock::now() - t0).count();
    for (Slab &s : sr.slabs) {
        die_hip(hipSetDevice(s.device), "hipSetDevice");
        (void)hipFree(s.f);
        (void)hipFree(s.buf);
        if (s.vel) (void)hipFree(s.vel);
    }
    return true;
}
