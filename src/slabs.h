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
//     the device's high-priority edge stream; then the exchange that puts the new edge rows into the
//     neighbours' ghost rows}.  The interior never reads a ghost row, so it overlaps the exchange;
//   * the exchange has two transports (LBM_SLAB_TRANSPORT=rccl|peer).  rccl (the default as soon as
//     the slabs live on more than one device): one RCCL communicator per device, made by this one
//     process (ncclCommInitAll); every edge is gathered into a contiguous [9][G][NX] staging buffer by
//     one strided local copy, all messages of a block go out as ncclSend / ncclRecv pairs inside one
//     ncclGroupStart / ncclGroupEnd on the devices' edge streams (the 8 MPI_Isend + 8 MPI_Irecv +
//     MPI_Waitall of main.cpp:223-230, :287-294, :326, device to device over xGMI instead of staged
//     through the host), and one strided local copy per edge scatters what arrived into the ghost
//     rows.  peer: two strided device-to-device copies per slab straight into the neighbours' arrays
//     (hipMemcpy2DAsync with peer access enabled; no staging, no collective library);
//   * ping-pong events order block b's interior after block b-1's edges, and block b's edges after
//     block b-1's interior and after the neighbours' exchange of block b-1.
// Slabs are initialised from their global row index, so any number of slabs reproduces the
// single-device populations bit for bit.  More slabs than devices are dealt round robin (slab d on
// device d mod #devices): that is how the schedule is tested on one GPU.
#pragma once

#include <rccl/rccl.h>
#include <fcntl.h>
#include <unistd.h>

struct Slab {
    int device = 0;
    UINT n = 0, y_below = 0;       // rows owned, global index of the first one
    void *f = nullptr, *buf = nullptr, *vel = nullptr;
    hipEvent_t interior_done[2] = {nullptr, nullptr}, edges_done[2] = {nullptr, nullptr}, xfer_done[2] = {nullptr, nullptr};
    // RCCL transport: contiguous [9][G][NX] staging of the lowest / highest owned rows (send) and of
    // what arrives for the lower / upper ghost rows (recv)
    void *send_lo = nullptr, *send_up = nullptr, *recv_lo = nullptr, *recv_up = nullptr;
    FLT *max_dev = nullptr;        // result word of the max-|u| reduction on the slab's device
};

struct SlabRun {
    std::vector<Slab> slabs;
    std::vector<hipStream_t> compute, edge;   // per device
    UINT G = 1;                               // ghost rows per side
    UINT block = 0;                           // blocks run so far (parity of the ping-pong events)
    bool periodic = true;
    bool rccl = false;                        // transport of the exchange
    std::vector<int> rank_of_device;          // RCCL rank of a device (-1: holds no slab)
    std::vector<ncclComm_t> comm;             // per rank
    FLT *max_host = nullptr;                  // pinned: one word per slab
};

static void die_nccl(ncclResult_t e, const char *what)
{
    if (e != ncclSuccess) {
        std::cerr << what << ": " << ncclGetErrorString(e) << std::endl;
        std::exit(1);
    }
}

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

// rows [row, row + G) of all nine populations of `arr` <-> a contiguous [9][G][NX] staging buffer
static void slab_stage_rows(const Run &r, const SlabRun &sr, const Slab &s, void *arr, UINT row, void *staging, bool gather,
                            hipStream_t stream)
{
    const Options &o = r.o;
    const size_t width = (size_t)sr.G * o.nx * sizeof(FLT);
    char *ap = (char *)arr + (size_t)row * o.nx * sizeof(FLT);
    const size_t apitch = slab_plane(o, s, sr.G) * sizeof(FLT);
    if (gather) die_hip(hipMemcpy2DAsync(staging, width, ap, apitch, width, 9, hipMemcpyDeviceToDevice, stream), "halo gather");
    else die_hip(hipMemcpy2DAsync(ap, apitch, staging, width, width, 9, hipMemcpyDeviceToDevice, stream), "halo scatter");
}

// The exchange over RCCL: gather, one group of send / receive pairs, scatter -- all on the devices'
// edge streams.  A message is posted on its sender's and on its receiver's communicator inside the
// same group, in one global order, so the sends and receives between any two ranks pair up in order
// (several slabs per device, a slab that is its own neighbour and a device that is its own peer
// included).
static void slab_exchange_rccl(const Run &r, SlabRun &sr, int which, unsigned parity)
{
    const Options &o = r.o;
    const size_t N = sr.slabs.size();
    const size_t count = (size_t)9 * sr.G * o.nx;
    const ncclDataType_t type = sizeof(FLT) == 4 ? ncclFloat : ncclDouble;
    for (size_t d = 0; d < N; ++d) {
        Slab &s = sr.slabs[d];
        die_hip(hipSetDevice(s.device), "hipSetDevice");
        hipStream_t es = sr.edge[s.device];
        void *mine = which ? s.buf : s.f;
        if (sr.periodic || d > 0) slab_stage_rows(r, sr, s, mine, sr.G, s.send_lo, true, es);
        if (sr.periodic || d + 1 < N) slab_stage_rows(r, sr, s, mine, s.n, s.send_up, true, es);
    }
    die_nccl(ncclGroupStart(), "ncclGroupStart");
    for (size_t d = 0; d < N; ++d) {
        Slab &s = sr.slabs[d];
        const int me = sr.rank_of_device[s.device];
        if (sr.periodic || d > 0) {            // my lowest owned rows -> the lower neighbour's upper ghost rows
            Slab &lo = sr.slabs[(d + N - 1) % N];
            const int peer = sr.rank_of_device[lo.device];
            die_nccl(ncclSend(s.send_lo, count, type, peer, sr.comm[me], sr.edge[s.device]), "ncclSend");
            die_nccl(ncclRecv(lo.recv_up, count, type, me, sr.comm[peer], sr.edge[lo.device]), "ncclRecv");
        }
        if (sr.periodic || d + 1 < N) {        // my highest owned rows -> the upper neighbour's lower ghost rows
            Slab &up = sr.slabs[(d + 1) % N];
            const int peer = sr.rank_of_device[up.device];
            die_nccl(ncclSend(s.send_up, count, type, peer, sr.comm[me], sr.edge[s.device]), "ncclSend");
            die_nccl(ncclRecv(up.recv_lo, count, type, me, sr.comm[peer], sr.edge[up.device]), "ncclRecv");
        }
    }
    die_nccl(ncclGroupEnd(), "ncclGroupEnd");
    for (size_t d = 0; d < N; ++d) {
        Slab &s = sr.slabs[d];
        die_hip(hipSetDevice(s.device), "hipSetDevice");
        hipStream_t es = sr.edge[s.device];
        void *mine = which ? s.buf : s.f;
        if (sr.periodic || d > 0) slab_stage_rows(r, sr, s, mine, 0, s.recv_lo, false, es);
        if (sr.periodic || d + 1 < N) slab_stage_rows(r, sr, s, mine, s.n + sr.G, s.recv_up, false, es);
        die_hip(hipEventRecord(s.xfer_done[parity], es), "event");
    }
}

// The edge rows of array `which` (0: f, 1: buf) of every slab go to the neighbours' ghost rows of the
// same array, on the edge streams; records xfer_done[parity].
static void slab_exchange(const Run &r, SlabRun &sr, int which, unsigned parity)
{
    if (sr.rccl) return slab_exchange_rccl(r, sr, which, parity);
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
        if (!sr.rccl) {
            // peer copies are issued by the neighbours' streams; over RCCL a slab's ghost rows are
            // filled on its own edge stream, in order
            die_hip(hipStreamWaitEvent(es, sr.slabs[(d + N - 1) % N].xfer_done[q], 0), "wait");
            die_hip(hipStreamWaitEvent(es, sr.slabs[(d + 1) % N].xfer_done[q], 0), "wait");
        }
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
        // every device reduces its slabs at the same time; one wait, then the MPI_MAX of io.cpp:213 on the host
        for (size_t d = 0; d < sr.slabs.size(); ++d) {
            Slab &s = sr.slabs[d];
            die_hip(hipSetDevice(s.device), "hipSetDevice");
            hipStream_t cs = sr.compute[s.device];
            die_lbm(lbm_max_velocity_dev(s.f, s.max_dev, kPrec, o.nx, s.n + 2 * G, G, G + s.n, cs), "max velocity");
            die_hip(hipMemcpyAsync(sr.max_host + d, s.max_dev, sizeof(FLT), hipMemcpyDeviceToHost, cs), "copy");
        }
        slab_wait_all(sr);
        double m = -1e30;       // io.cpp:184
        for (size_t d = 0; d < sr.slabs.size(); ++d)
            if ((double)sr.max_host[d] > m) m = (double)sr.max_host[d];
        r.max_vel = m;
        return;
    }
    const size_t plane = (size_t)o.nx * o.ny;
    r.host.resize(o.output == VEL_FIELD ? 2 * plane : plane);   // page-locked: the slabs' copies run side by side
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

// What the reference does ahead of its clock (MPI_Init, main.cpp:410, before t0 at :421): devices, streams,
// the transport of the exchange and -- for RCCL -- one communicator per device, all made by this process.
static bool slab_startup(const Options &o, SlabRun &sr, unsigned nslabs)
{
    int ndev = 0;
    die_hip(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
    if (ndev < 1) { std::cerr << "no HIP device" << std::endl; return false; }
    sr.compute.assign((size_t)ndev, nullptr);
    sr.edge.assign((size_t)ndev, nullptr);
    const int ndev_used = ndev < (int)nslabs ? ndev : (int)nslabs;
    // Transport of the exchange: RCCL as soon as two devices take part, plain copies on one device;
    // LBM_SLAB_TRANSPORT=rccl|peer names it outright (rccl on one device: the device is its own peer).
    const char *e = std::getenv("LBM_SLAB_TRANSPORT");
    if (e && std::strcmp(e, "rccl") == 0) sr.rccl = true;
    else if (e && std::strcmp(e, "peer") == 0) sr.rccl = false;
    else if (e && *e) { std::cerr << "LBM_SLAB_TRANSPORT must be rccl or peer" << std::endl; return false; }
    else sr.rccl = ndev_used > 1;
    for (int dev = 0; dev < ndev_used; ++dev) {
        die_hip(hipSetDevice(dev), "hipSetDevice");
        int lo = 0, hi = 0;
        die_hip(hipDeviceGetStreamPriorityRange(&lo, &hi), "priority range");
        die_hip(hipStreamCreateWithFlags(&sr.compute[dev], hipStreamNonBlocking), "stream");
        // one high-priority stream per device: the edge sweeps and the exchange are short and everybody waits for them
        die_hip(hipStreamCreateWithPriority(&sr.edge[dev], hipStreamNonBlocking, hi), "stream");
        warm_device(sr.compute[dev]);
    }
    if (sr.rccl) {
        std::vector<int> devs((size_t)ndev_used);
        sr.rank_of_device.assign((size_t)ndev, -1);
        for (int dev = 0; dev < ndev_used; ++dev) { devs[(size_t)dev] = dev; sr.rank_of_device[(size_t)dev] = dev; }
        sr.comm.resize((size_t)ndev_used);
        // RCCL greets its first communicator with a version banner on file descriptor 1, which is where the results go
        // (io.cpp:219), so the descriptor points at /dev/null while the communicators come up
        std::cout.flush();
        std::fflush(stdout);
        const int keep = dup(1), null_fd = open("/dev/null", O_WRONLY);
        if (keep >= 0 && null_fd >= 0) dup2(null_fd, 1);
        const ncclResult_t made = ncclCommInitAll(sr.comm.data(), ndev_used, devs.data());
        std::fflush(stdout);
        if (keep >= 0) { dup2(keep, 1); close(keep); }
        if (null_fd >= 0) close(null_fd);
        die_nccl(made, "ncclCommInitAll");
        // RCCL connects two ranks when they first talk, channel by channel as the message size asks for
        // them (hundreds of milliseconds): one message of the size of a block's edge around the ring of
        // devices, both ways and to itself, still ahead of the clock
        const size_t warm = (size_t)9 * lbm_fused_max_steps(kPrec, o.nx) * o.nx;   // elements of the deepest edge
        std::vector<FLT *> word((size_t)ndev_used, nullptr);
        for (int dev = 0; dev < ndev_used; ++dev) {
            die_hip(hipSetDevice(dev), "hipSetDevice");
            die_hip(hipMalloc((void **)&word[(size_t)dev], 6 * warm * sizeof(FLT)), "warm-up messages");
        }
        die_nccl(ncclGroupStart(), "ncclGroupStart");
        for (int dev = 0; dev < ndev_used; ++dev) {
            const int peers[3] = {(dev + 1) % ndev_used, (dev + ndev_used - 1) % ndev_used, dev};
            for (int j = 0; j < 3; ++j) {
                // message j of `dev` lands in slot 3 + j of its peer; a peer reached twice (two devices) or
                // thrice (one device) gets as many messages from this rank, in this order
                die_nccl(ncclSend(word[(size_t)dev] + warm * j, warm, sizeof(FLT) == 4 ? ncclFloat : ncclDouble, peers[j],
                                  sr.comm[(size_t)dev], sr.edge[(size_t)dev]), "ncclSend");
                die_nccl(ncclRecv(word[(size_t)peers[j]] + warm * (3 + j), warm, sizeof(FLT) == 4 ? ncclFloat : ncclDouble, dev,
                                  sr.comm[(size_t)peers[j]], sr.edge[(size_t)peers[j]]), "ncclRecv");
            }
        }
        die_nccl(ncclGroupEnd(), "ncclGroupEnd");
        for (int dev = 0; dev < ndev_used; ++dev) {
            die_hip(hipSetDevice(dev), "hipSetDevice");
            die_hip(hipStreamSynchronize(sr.edge[(size_t)dev]), "sync");
            (void)hipFree(word[(size_t)dev]);
        }
    }
    die_hip(hipSetDevice(0), "hipSetDevice");
    return true;
}

// The time loop of main.cpp:217-398 over `nslabs` slabs.
static bool run_simulation_slabs(Run &r, SlabRun &sr, unsigned nslabs)
{
    const Options &o = r.o;
    const int ndev = (int)sr.compute.size();
    if ((UINT)nslabs > o.ny) { std::cerr << "more slabs than rows" << std::endl; return false; }
    sr.periodic = o.scene == SHEAR_WAVE;
    sr.slabs.resize(nslabs);
    const UINT base = o.ny / nslabs, rem = o.ny % nslabs;
    // ghost depth = steps per block: what the library plans for a slab of the smallest height, as long
    // as a slab holds two edge regions of that depth
    int g = lbm_plan_steps_scene(kPrec, o.nx, base, (uint64_t)1 << 30, o.scene == LID_DRIVEN);
    if (g < 1) die_lbm(LBM_ERR_INVALID_ARGUMENT, "planning the slab blocks");
    while (g > 1 && base < 2 * (UINT)g) --g;
    sr.G = (UINT)g;
    if (base < 2 * sr.G) { std::cerr << "slabs of " << base << " rows are too thin: at least 2 rows per slab" << std::endl; return false; }
    const UINT G = sr.G;
    const int ndev_used = ndev < (int)nslabs ? ndev : (int)nslabs;

    std::cerr << "running on" << o.nx << "x" << o.ny << std::endl;
    std::cerr << nslabs << " y-slab(s) of " << base << (rem ? "-" + std::to_string(base + 1) : std::string()) << " rows on "
              << ndev_used << " device(s), " << G << " ghost row(s) per edge exchanged every "
              << G << " step(s) by " << (sr.rccl ? "RCCL send/recv" : "device-to-device copies") << std::endl;
    if (o.tx == 64 || o.tx == 128 || o.tx == 256) lbm_set_launch_shape((unsigned)o.tx, 0);

    for (unsigned d = 0; d < nslabs; ++d) {
        Slab &s = sr.slabs[d];
        s.device = (int)(d % (unsigned)ndev);
        s.n = base + (d < rem ? 1 : 0);
        s.y_below = d * base + (d < rem ? d : rem);
    }

    // LBM_TRACE_SETUP=1: where the set-up time goes, on stderr (ahead of the last line)
    const bool trace = std::getenv("LBM_TRACE_SETUP") != nullptr;
    const auto t_setup = std::chrono::high_resolution_clock::now();
    auto mark = [&](const char *what) {
        if (!trace) return;
        for (int dev = 0; dev < ndev_used; ++dev) { die_hip(hipSetDevice(dev), "hipSetDevice"); die_hip(hipDeviceSynchronize(), "sync"); }
        std::cerr << "  set-up: " << what << " at "
                  << std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - t_setup).count()
                  << " us" << std::endl;
    };
    // per slab: events.  Cheap, done in order.
    for (Slab &s : sr.slabs) {
        die_hip(hipSetDevice(s.device), "hipSetDevice");
        for (int e = 0; e < 2; ++e) {
            die_hip(hipEventCreateWithFlags(&s.interior_done[e], hipEventDisableTiming), "event");
            die_hip(hipEventCreateWithFlags(&s.edges_done[e], hipEventDisableTiming), "event");
            die_hip(hipEventCreateWithFlags(&s.xfer_done[e], hipEventDisableTiming), "event");
        }
    }
    die_hip(hipHostMalloc((void **)&sr.max_host, nslabs * sizeof(FLT), hipHostMallocDefault), "pinned result words");

    // Allocation and initial state, every device at the same time (one host thread per device: hipMalloc
    // of tens of GB and the initialisers' own stream synchronisation are host-side waits).  Slabs start
    // from their global row index (init.cpp:22-26 with the right offset; main.cpp:141-144 gets it
    // wrong and is not reproduced).  Only ghost rows are cleared: every owned row of f is written by
    // the initialiser and every owned row of buf by the first sweep; ghost rows that no neighbour
    // fills (below the bottom wall, above the lid) are read for nodes nobody stores and must only
    // be defined.
    auto setup_device = [&](int dev) {
        die_hip(hipSetDevice(dev), "hipSetDevice");
        hipStream_t cs = sr.compute[dev];
        for (Slab &s : sr.slabs) {
            if (s.device != dev) continue;
            const size_t plane_b = slab_plane(o, s, G) * sizeof(FLT), bytes = 9 * plane_b;
            const size_t edge_b = (size_t)G * o.nx * sizeof(FLT);
            die_hip(hipMalloc(&s.f, bytes), "allocating f");
            die_hip(hipMalloc(&s.buf, bytes), "allocating buf");
            die_hip(hipMalloc((void **)&s.max_dev, 256), "allocating the reduction's result word");
            if (o.output == VEL_MAGS || o.output == VEL_FIELD)
                die_hip(hipMalloc(&s.vel, (size_t)2 * s.n * o.nx * sizeof(FLT)), "allocating vel");
            if (sr.rccl) {
                for (void **p : {&s.send_lo, &s.send_up, &s.recv_lo, &s.recv_up})
                    die_hip(hipMalloc(p, 9 * edge_b), "allocating halo staging");
            }
            for (void *arr : {s.f, s.buf}) {
                die_hip(hipMemset2DAsync(arr, plane_b, 0, edge_b, 9, cs), "memset");
                die_hip(hipMemset2DAsync((char *)arr + (size_t)(s.n + G) * o.nx * sizeof(FLT), plane_b, 0, edge_b, 9, cs), "memset");
            }
            if (o.scene == LID_DRIVEN) {
                die_lbm(lbm_init_rest(s.f, kPrec, o.nx, s.n + 2 * G, o.rho, cs), "init_rest");
            } else {
                die_lbm(lbm_init_shearwave(s.f, kPrec, o.nx, s.n + 2 * G, o.rho, o.u, s.y_below, o.ny, (unsigned)G, cs), "init_shearwave");
            }
        }
        die_hip(hipStreamSynchronize(cs), "sync");
    };
    mark("events and pinned words");
    {
        std::vector<std::thread> workers;
        for (int dev = 1; dev < ndev_used; ++dev) workers.emplace_back(setup_device, dev);
        setup_device(0);
        for (std::thread &w : workers) w.join();
    }
    mark("allocation and initial state");
    if (!sr.rccl) {
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
    }
    // ghost rows of f from the neighbours' initial edge rows; the first block's edges wait for it
    slab_exchange(r, sr, 0, 1);
    mark("first exchange");
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
        for (void *p : {s.f, s.buf, s.vel, s.send_lo, s.send_up, s.recv_lo, s.recv_up, (void *)s.max_dev})
            if (p) (void)hipFree(p);
        for (int e = 0; e < 2; ++e)
            for (hipEvent_t ev : {s.interior_done[e], s.edges_done[e], s.xfer_done[e]})
                if (ev) (void)hipEventDestroy(ev);
    }
    sr.slabs.clear();
    (void)hipHostFree(sr.max_host);
    sr.max_host = nullptr;
    die_hip(hipSetDevice(0), "hipSetDevice");
    return true;
}

// The counterpart of slab_startup, behind the clock like MPI_Finalize (main.cpp:441): taking the
// communicators down costs RCCL some 0.4 s.
static void slab_shutdown(SlabRun &sr)
{
    for (ncclComm_t c : sr.comm) (void)ncclCommDestroy(c);
    sr.comm.clear();
    for (size_t dev = 0; dev < sr.compute.size(); ++dev) {
        if (!sr.compute[dev]) continue;
        if (hipSetDevice((int)dev) != hipSuccess) continue;
        (void)hipStreamDestroy(sr.compute[dev]);
        (void)hipStreamDestroy(sr.edge[dev]);
    }
    sr.compute.clear();
    sr.edge.clear();
}
