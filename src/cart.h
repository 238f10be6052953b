// The reference's 2-D Cartesian decomposition (JulianKarrer/lbm src/main.cpp:92-401) behind the `lbm` driver:
// the periodic lattice cut into PX x PY blocks, all driven by this one process, one multi-step launch per block
// and exchange.
//
// The reference gives every rank a halo ring of ONE node and exchanges eight messages per time step through
// host mirrors (main.cpp:166-201, :223-294, :326-374).  Here a block keeps a frame of G nodes (all nine
// populations) around its owned nodes, arrays [9][ny + 2G][nx + 2G]:
//   * a block of k <= G time steps is ONE launch of the multi-step periodic sweep over the block's array as if it
//     wrapped in x (rows [G, G + ny), ghost rows present).  The wrap joins the two frame edges, which is wrong, but
//     an error travels one column per step: after k <= G steps it has not left the frame, and every owned node holds
//     the bits of the single-device run;
//   * the frame is then refilled in two phases, device to device: columns over the owned rows (from the left and
//     right neighbours' outermost owned columns), then rows over the full width (from the lower and upper
//     neighbours -- the corners ride along with the frame columns those rows already hold): four messages per block
//     of up to eight steps instead of eight per step;
//   * fp32 rows get one more column where nx + 2G is odd (the multi-step kernels take two nodes per lane); it lies
//     beyond the frame and nothing owned ever depends on it.
// Blocks are dealt to the devices round robin (block b on device b mod #devices), which is how the schedule is tested
// on one GPU; neighbours on different devices copy into each other's memory (peer access).  Like the reference's MPI
// path (main.cpp:317-319) this decomposition runs the periodic scene only; inside one node the y-slabs of slabs.h are
// the faster choice (their exchange overlaps the interior sweep), this one is for lattices cut in both directions.
#pragma once

struct CartBlock {
    int device = 0;
    UINT cx = 0, cy = 0;           // position in the grid of blocks
    UINT nx = 0, ny = 0;           // nodes owned
    UINT x_below = 0, y_below = 0; // global index of the first owned column / row
    UINT W = 0, H = 0;             // array extents: nx + 2G (+1), ny + 2G
    void *f = nullptr, *buf = nullptr, *vel = nullptr;
    FLT *max_dev = nullptr;
    hipEvent_t swept[2] = {nullptr, nullptr}, cols[2] = {nullptr, nullptr}, rows[2] = {nullptr, nullptr};
};

struct CartRun {
    std::vector<CartBlock> blocks;
    std::vector<hipStream_t> stream;   // per block (block i on device i mod ndev)
    int ndev = 0;
    UINT px = 1, py = 1, G = 1;
    UINT count = 0;                    // blocks of steps run so far (parity of the events)
    FLT *max_host = nullptr;
};

static inline CartBlock &cart_at(CartRun &cr, UINT cx, UINT cy) { return cr.blocks[(size_t)(cy % cr.py) * cr.px + cx % cr.px]; }

// MPI_Dims_create(n, 2) as the reference calls it (main.cpp:103-104): the two factors of n closest to each other,
// larger first.
static void cart_dims(unsigned n, UINT &px, UINT &py)
{
    unsigned a = 1;
    for (unsigned d = 1; d * d <= n; ++d)
        if (n % d == 0) a = d;
    px = n / a;
    py = a;
}

// columns [src_col, src_col + G) of rows [G, G + ny) of all nine populations of `src` -> columns from dst_col of `dst`
static void cart_copy_columns(const CartRun &cr, const CartBlock &src, const void *src_arr, UINT src_col, const CartBlock &dst,
                              void *dst_arr, UINT dst_col, hipStream_t stream)
{
    const size_t sz = sizeof(FLT);
    for (int q = 0; q < 9; ++q) {
        const char *sp = (const char *)src_arr + (((size_t)q * src.H + cr.G) * src.W + src_col) * sz;
        char *dp = (char *)dst_arr + (((size_t)q * dst.H + cr.G) * dst.W + dst_col) * sz;
        die_hip(hipMemcpy2DAsync(dp, dst.W * sz, sp, src.W * sz, cr.G * sz, src.ny, hipMemcpyDeviceToDevice, stream), "frame columns");
    }
}

// rows [src_row, src_row + G) over the full width of all nine populations of `src` -> rows from dst_row of `dst`
static void cart_copy_rows(const CartRun &cr, const CartBlock &src, const void *src_arr, UINT src_row, const CartBlock &dst,
                           void *dst_arr, UINT dst_row, hipStream_t stream)
{
    const size_t sz = sizeof(FLT), width = (size_t)cr.G * src.W * sz;
    const char *sp = (const char *)src_arr + (size_t)src_row * src.W * sz;
    char *dp = (char *)dst_arr + (size_t)dst_row * dst.W * sz;
    die_hip(hipMemcpy2DAsync(dp, (size_t)dst.H * dst.W * sz, sp, (size_t)src.H * src.W * sz, width, 9, hipMemcpyDeviceToDevice, stream),
            "frame rows");
}

// Refill the frames of array `which` (0: f, 1: buf) of every block from its neighbours' owned nodes.
// after_sweeps: the arrays were just written by this block of steps' sweeps (swept[p] recorded).
static void cart_exchange(CartRun &cr, int which, unsigned p, bool after_sweeps)
{
    const UINT G = cr.G;
    auto arr = [&](CartBlock &b) { return which ? b.buf : b.f; };
    // columns: a sweep writes its whole rows, frame columns included, so a neighbour's columns may only land after it
    for (CartBlock &b : cr.blocks) {
        die_hip(hipSetDevice(b.device), "hipSetDevice");
        hipStream_t s = cr.stream[&b - cr.blocks.data()];
        CartBlock &left = cart_at(cr, b.cx + cr.px - 1, b.cy), &right = cart_at(cr, b.cx + 1, b.cy);
        if (after_sweeps) {
            die_hip(hipStreamWaitEvent(s, left.swept[p], 0), "wait");
            die_hip(hipStreamWaitEvent(s, right.swept[p], 0), "wait");
        }
        cart_copy_columns(cr, b, arr(b), G, left, arr(left), G + left.nx, s);          // my first owned columns -> its right frame
        cart_copy_columns(cr, b, arr(b), b.nx, right, arr(right), 0, s);               // my last owned columns -> its left frame
        die_hip(hipEventRecord(b.cols[p], s), "event");
    }
    // rows over the full width, once the frame columns they carry have arrived
    for (CartBlock &b : cr.blocks) {
        die_hip(hipSetDevice(b.device), "hipSetDevice");
        hipStream_t s = cr.stream[&b - cr.blocks.data()];
        CartBlock &left = cart_at(cr, b.cx + cr.px - 1, b.cy), &right = cart_at(cr, b.cx + 1, b.cy);
        CartBlock &lower = cart_at(cr, b.cx, b.cy + cr.py - 1), &upper = cart_at(cr, b.cx, b.cy + 1);
        die_hip(hipStreamWaitEvent(s, left.cols[p], 0), "wait");
        die_hip(hipStreamWaitEvent(s, right.cols[p], 0), "wait");
        // the neighbours' ghost rows must not be overwritten while their own column copies still read ... they do not:
        // column copies touch owned rows only; but their sweeps wrote nothing there either, so no further wait
        cart_copy_rows(cr, b, arr(b), G, lower, arr(lower), G + lower.ny, s);           // my lowest owned rows -> its upper ghost rows
        cart_copy_rows(cr, b, arr(b), b.ny, upper, arr(upper), 0, s);                   // my highest owned rows -> its lower ghost rows
        die_hip(hipEventRecord(b.rows[p], s), "event");
    }
}

static void cart_block(const Run &r, CartRun &cr, int k)
{
    const Options &o = r.o;
    const unsigned p = cr.count & 1u, q = p ^ 1u;
    const UINT G = cr.G;
    for (CartBlock &b : cr.blocks) {
        die_hip(hipSetDevice(b.device), "hipSetDevice");
        hipStream_t s = cr.stream[&b - cr.blocks.data()];
        // the frame of f: columns from left and right (waited for by this stream's own row phase), rows from below and above
        CartBlock &lower = cart_at(cr, b.cx, b.cy + cr.py - 1), &upper = cart_at(cr, b.cx, b.cy + 1);
        die_hip(hipStreamWaitEvent(s, lower.rows[q], 0), "wait");
        die_hip(hipStreamWaitEvent(s, upper.rows[q], 0), "wait");
        const int st = o.push ? lbm_push_periodic_fused(b.f, b.buf, kPrec, b.W, b.H, o.omega, G, G + b.ny, 0, k, s)
                              : lbm_pull_periodic_fused(b.f, b.buf, kPrec, b.W, b.H, o.omega, G, G + b.ny, 0, k, s);
        die_lbm(st, "block sweep");
        die_hip(hipEventRecord(b.swept[p], s), "event");
    }
    cart_exchange(cr, 1, p, true);
    for (CartBlock &b : cr.blocks) std::swap(b.f, b.buf);
    ++cr.count;
}

static void cart_wait_all(CartRun &cr)
{
    for (size_t i = 0; i < cr.blocks.size(); ++i) {
        die_hip(hipSetDevice(cr.blocks[i].device), "hipSetDevice");
        die_hip(hipStreamSynchronize(cr.stream[i]), "sync");
    }
}

// output() of io.cpp:177-294 over the blocks: the sample is assembled on the host in global order
static void cart_fetch_output(Run &r, CartRun &cr)
{
    const Options &o = r.o;
    if (o.output == NONE) return;
    const UINT G = cr.G;
    const size_t sz = sizeof(FLT);
    if (o.output == MAX_VEL) {
        for (size_t i = 0; i < cr.blocks.size(); ++i) {
            CartBlock &b = cr.blocks[i];
            die_hip(hipSetDevice(b.device), "hipSetDevice");
            hipStream_t s = cr.stream[&b - cr.blocks.data()];
            // the reduction takes whole rows: frame columns hold neighbours' nodes (harmless duplicates); the extra
            // column of an fp32 block holds nothing physical and gets a copy of an owned column first
            if (b.W > b.nx + 2 * G) {
                for (int q = 0; q < 9; ++q) {
                    char *row0 = (char *)b.f + ((size_t)q * b.H + G) * b.W * sz;
                    die_hip(hipMemcpy2DAsync(row0 + (size_t)(b.W - 1) * sz, b.W * sz, row0 + (size_t)G * sz, b.W * sz, sz, b.ny,
                                             hipMemcpyDeviceToDevice, s), "extra column");
                }
            }
            die_lbm(lbm_max_velocity_dev(b.f, b.max_dev, kPrec, b.W, b.H, G, G + b.ny, s), "max velocity");
            die_hip(hipMemcpyAsync(cr.max_host + i, b.max_dev, sz, hipMemcpyDeviceToHost, s), "copy");
        }
        cart_wait_all(cr);
        double m = -1e30;       // io.cpp:184; the MPI_MAX reduction of io.cpp:213 on the host
        for (size_t i = 0; i < cr.blocks.size(); ++i)
            if ((double)cr.max_host[i] > m) m = (double)cr.max_host[i];
        r.max_vel = m;
        return;
    }
    const size_t plane = (size_t)o.nx * o.ny;
    r.host.resize(o.output == VEL_FIELD ? 2 * plane : plane);
    for (CartBlock &b : cr.blocks) {
        die_hip(hipSetDevice(b.device), "hipSetDevice");
        hipStream_t s = cr.stream[&b - cr.blocks.data()];
        // planar [rows][W] fields of the block's rows; the owned columns go to their place in the global field
        if (o.output == VEL_MAGS) die_lbm(lbm_velocity_magnitudes(b.f, b.vel, kPrec, b.W, b.H, G, G + b.ny, s), "velocity magnitudes");
        else die_lbm(lbm_compute_velocities(b.f, b.vel, kPrec, b.W, b.H, G, G + b.ny, s), "velocities");
        const int fields = o.output == VEL_FIELD ? 2 : 1;
        for (int c = 0; c < fields; ++c) {
            const char *sp = (const char *)b.vel + ((size_t)c * b.ny * b.W + G) * sz;
            FLT *dp = r.host.data() + (size_t)c * plane + (size_t)b.y_below * o.nx + b.x_below;
            die_hip(hipMemcpy2DAsync(dp, o.nx * sz, sp, b.W * sz, b.nx * sz, b.ny, hipMemcpyDeviceToHost, s), "copy");
        }
    }
    cart_wait_all(cr);
}

// Devices and streams, ahead of the clock like slab_startup.  Every block has a stream of its own, also where
// several blocks share a device: the schedule's ordering then rests on its events alone, on one GPU as on many.
static bool cart_startup(CartRun &cr, unsigned nblocks)
{
    int ndev = 0;
    die_hip(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
    if (ndev < 1) { std::cerr << "no HIP device" << std::endl; return false; }
    cr.ndev = ndev;
    cr.stream.assign((size_t)nblocks, nullptr);
    const int used = ndev < (int)nblocks ? ndev : (int)nblocks;
    for (unsigned i = 0; i < nblocks; ++i) {
        die_hip(hipSetDevice((int)(i % (unsigned)ndev)), "hipSetDevice");
        die_hip(hipStreamCreateWithFlags(&cr.stream[i], hipStreamNonBlocking), "stream");
        if (i < (unsigned)ndev) warm_device(cr.stream[i]);
    }
    // any two devices that hold blocks may be neighbours
    for (int a = 0; a < used; ++a)
        for (int b = 0; b < used; ++b) {
            if (a == b) continue;
            int can = 0;
            die_hip(hipDeviceCanAccessPeer(&can, a, b), "peer query");
            if (!can) { std::cerr << "device " << a << " cannot access device " << b << std::endl; return false; }
            die_hip(hipSetDevice(a), "hipSetDevice");
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) die_hip(e, "enabling peer access");
            (void)hipGetLastError();
        }
    die_hip(hipSetDevice(0), "hipSetDevice");
    return true;
}

// The time loop of main.cpp:217-398 over a grid of px x py blocks.
static bool run_simulation_cart(Run &r, CartRun &cr, UINT px, UINT py)
{
    const Options &o = r.o;
    if (o.scene != SHEAR_WAVE) {
        std::cerr << "the 2-D decomposition runs the periodic scene only (as the reference's, main.cpp:317-319); use -gpus N without it"
                  << std::endl;
        return false;
    }
    const int ndev = cr.ndev;
    const unsigned nblocks = (unsigned)(px * py);
    cr.px = px;
    cr.py = py;
    const UINT bx = o.nx / px, rx = o.nx % px, by = o.ny / py, ry = o.ny % py;   // remainders to the lowest coordinates (main.cpp:133-141)
    if (bx < 1 || by < 1) { std::cerr << "more blocks than nodes along an axis" << std::endl; return false; }
    // frame depth = steps per launch: what the library plans for the smallest block (rows made even), capped by its extents
    UINT G = 1;
    for (int round = 0; round < 2; ++round) {
        UINT w = bx + 2 * G;
        if (kPrec == 4) w += w % 2;
        int g = lbm_plan_steps_scene(kPrec, w, by, (uint64_t)1 << 30, 0);
        if (g < 1) die_lbm(LBM_ERR_INVALID_ARGUMENT, "planning the blocks");
        G = (UINT)g;
        if (G > bx) G = bx;
        if (G > by) G = by;
    }
    cr.G = G;
    std::cerr << "running on" << o.nx << "x" << o.ny << std::endl;
    std::cerr << px << "x" << py << " block(s) of " << bx << (rx ? "-" + std::to_string(bx + 1) : std::string()) << " x " << by
              << (ry ? "-" + std::to_string(by + 1) : std::string()) << " nodes on " << (ndev < (int)nblocks ? ndev : (int)nblocks)
              << " device(s), frames of " << G << " node(s) refilled every " << G << " step(s) by device-to-device copies" << std::endl;

    cr.blocks.resize(nblocks);
    for (UINT cy = 0; cy < py; ++cy)
        for (UINT cx = 0; cx < px; ++cx) {
            CartBlock &b = cr.blocks[(size_t)cy * px + cx];
            b.device = (int)((cy * px + cx) % (UINT)ndev);
            b.cx = cx; b.cy = cy;
            b.nx = bx + (cx < rx ? 1 : 0);
            b.ny = by + (cy < ry ? 1 : 0);
            b.x_below = cx * bx + (cx < rx ? cx : rx);
            b.y_below = cy * by + (cy < ry ? cy : ry);
            b.W = b.nx + 2 * G;
            if (kPrec == 4) b.W += b.W % 2;
            b.H = b.ny + 2 * G;
        }
    die_hip(hipHostMalloc((void **)&cr.max_host, nblocks * sizeof(FLT), hipHostMallocDefault), "pinned result words");
    for (CartBlock &b : cr.blocks) {
        die_hip(hipSetDevice(b.device), "hipSetDevice");
        hipStream_t s = cr.stream[&b - cr.blocks.data()];
        const size_t bytes = (size_t)9 * b.W * b.H * sizeof(FLT);
        die_hip(hipMalloc(&b.f, bytes), "allocating f");
        die_hip(hipMalloc(&b.buf, bytes), "allocating buf");
        die_hip(hipMalloc((void **)&b.max_dev, 256), "allocating the reduction's result word");
        if (o.output == VEL_MAGS || o.output == VEL_FIELD)
            die_hip(hipMalloc(&b.vel, (size_t)2 * b.ny * b.W * sizeof(FLT)), "allocating vel");
        for (int e = 0; e < 2; ++e) {
            die_hip(hipEventCreateWithFlags(&b.swept[e], hipEventDisableTiming), "event");
            die_hip(hipEventCreateWithFlags(&b.cols[e], hipEventDisableTiming), "event");
            die_hip(hipEventCreateWithFlags(&b.rows[e], hipEventDisableTiming), "event");
        }
        // everything defined (the extra column and the ghost rows are read before anything fills them)
        die_hip(hipMemsetAsync(b.f, 0, bytes, s), "memset");
        die_hip(hipMemsetAsync(b.buf, 0, bytes, s), "memset");
        // the shear wave does not depend on x: every column of a row, frame included, gets the row's equilibrium
        // (init.cpp:22-26 with the global row index; main.cpp:141-144 gets the offset wrong and is not reproduced)
        die_lbm(lbm_init_shearwave(b.f, kPrec, b.W, b.H, o.rho, o.u, b.y_below, o.ny, (unsigned)G, s), "init_shearwave");
    }
    cart_wait_all(cr);
    cart_exchange(cr, 0, 1, false);     // ghost rows of the initial state; the first sweeps wait for rows[1]

    const auto t0 = std::chrono::high_resolution_clock::now();
    bool pending = false;
    unsigned queued = 0;
    for (UINT t = 0; t < o.steps;) {
        if (o.out_every > 0 && t % o.out_every == 0) {
            if (pending && !emit_output(r)) return false;
            cart_fetch_output(r, cr);
            pending = true;
            queued = 0;
        }
        UINT until = o.steps;
        if (o.out_every > 0) {
            const UINT next_out = (t / o.out_every + 1) * o.out_every;
            if (next_out < until) until = next_out;
        }
        // the fewest blocks of at most G steps, as equal as possible
        const UINT left = until - t, nb = (left + G - 1) / G;
        const int k = (int)((left + nb - 1) / nb);
        cart_block(r, cr, k);
        t += (UINT)k;
        if (pending && (t >= until || ++queued >= 64)) {
            if (!emit_output(r)) return false;
            pending = false;
        }
    }
    if (pending && !emit_output(r)) return false;
    cart_wait_all(cr);
    r.sweep_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::high_resolution_clock::now() - t0).count();
    for (CartBlock &b : cr.blocks) {
        die_hip(hipSetDevice(b.device), "hipSetDevice");
        for (void *p : {b.f, b.buf, b.vel, (void *)b.max_dev})
            if (p) (void)hipFree(p);
        for (int e = 0; e < 2; ++e)
            for (hipEvent_t ev : {b.swept[e], b.cols[e], b.rows[e]})
                if (ev) (void)hipEventDestroy(ev);
    }
    for (size_t i = 0; i < cr.stream.size(); ++i) {
        die_hip(hipSetDevice((int)(i % (size_t)ndev)), "hipSetDevice");
        (void)hipStreamDestroy(cr.stream[i]);
    }
    cr.stream.clear();
    cr.blocks.clear();
    (void)hipHostFree(cr.max_host);
    cr.max_host = nullptr;
    die_hip(hipSetDevice(0), "hipSetDevice");
    return true;
}
