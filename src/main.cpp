// `lbm` command-line driver on the HIP C ABI (include/lbm_hip.h).
//
// Keeps the command-line surface, the text output formats and the stderr contract of the
// reference driver (JulianKarrer/lbm src/main.cpp:31-89, :407-444; flags src/io.cpp:3-133; outputs
// src/io.cpp:177-294) so that the upstream scripts (benchmark.py, problem-size.py, shear-wave.py,
// shearwave-video.py, lid-driven-stationary.py) drive it unchanged:
//   * build/main, run with cwd=build; precision chosen at build time by src/macros.h;
//   * stdout (or -f file): -omv "<max>," per output; -ov NY lines + "#"; -ovf ux block, "#", uy
//     block, "#"; output happens BEFORE step t when t % of == 0;
//   * stderr: "running on<NX>x<NY>", ..., blank line, and as the LAST line
//     "<NX*NY*STEPS> lattice updates in |<ns>|ns => <MLUPS> MLUPS".
// All lattice arithmetic runs in the gfx950 kernels; the host only parses, launches and prints.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "fmt_g.h"
#include "lbm_hip.h"
#include "macros.h"

// defaults as the reference gives them (main.cpp:12-24)
#ifndef USE_SINGLE_PRECISION
#define USE_SINGLE_PRECISION true
#endif
#ifndef UNROLL_LOOPS
#define UNROLL_LOOPS true
#endif
#ifndef COALESCING
#define COALESCING true
#endif
#ifndef USE_MPI
#define USE_MPI false
#endif

#if USE_SINGLE_PRECISION
typedef float FLT;
#else
typedef double FLT;
#endif
static const int kPrec = (int)sizeof(FLT);

typedef unsigned long long UINT;  // reference global.h:32

enum OutputType { NONE, MAX_VEL, VEL_MAGS, VEL_FIELD };  // reference global.h:46-55
enum SceneType { SHEAR_WAVE, LID_DRIVEN };               // reference global.h:59-64

struct Options {
    UINT nx = 1024, ny = 1024, out_every = 0, steps = 50000, tx = 512, ty = 1;
    double omega = 0.5, rho = 1.0, u = 0.1;
    bool push = false;
    OutputType output = NONE;
    SceneType scene = SHEAR_WAVE;
    std::string file;
    bool use_file = false;
    unsigned gpus = 0;   // y-slabs (0 = not given): -gpus N, else LBM_GPUS, else every device when USE_MPI is set
};

// ------------------------------------------------------------------------------------------------
// Command line (reference io.cpp:3-133).  Note the reference registers -tx/-ty under the long
// names --x-grid-points/--y-grid-points a second time (io.cpp:52,58); its parser maps a name to the
// last argument registered under it, so the long names address the tile sizes and only -nx/-ny
// address the lattice.  That behaviour is kept.
static const char *kUsage =
    "Usage: lbm [--help] [--version] -nx VAR -ny VAR --output-frequency VAR --steps VAR --omega VAR "
    "--density VAR --initial-velocity VAR --x-grid-points VAR --y-grid-points VAR [--push] "
    "[[--output-max-vel]|[--output-velocity]|[--output-velocity-field]] [--file VAR] "
    "[[--shear-wave-decay]|[--lid-driven-cavity]]\n\n"
    "Optional arguments:\n"
    "  -h, --help                    shows help message and exits \n"
    "  -v, --version                 prints version information and exits \n"
    "  -nx                           Specify the number of grid points in the x-direction [default: 1024]\n"
    "  -ny                           Specify the number of grid points in the y-direction [default: 1024]\n"
    "  -of, --output-frequency       Specify how frequently (as an integer number of timesteps) simulation results should be output. 0 means no output. [default: 0]\n"
    "  -s, --steps                   Specify the total number of simulation time steps to run [default: 50000]\n"
    "  -w, --omega                   Specify the relaxation coefficient omega, which should obey be a float in (0;2) [default: 0.5]\n"
    "  -rho, --density               Specify the initial density at each fluid node [default: 1]\n"
    "  -u, --initial-velocity        Specify the initial velocity. For the shear wave decay, this is the velocity amplitude. For the lid-driven cavity, this is the horizontal velocity of the top lid wall. [default: 0.1]\n"
    "  -tx, --x-grid-points          Specify the kernel tiling size in x-direction for performance optimization [default: 512]\n"
    "  -ty, --y-grid-points          Specify the kernel tiling size in y-direction for performance optimization [default: 1]\n"
    "  -push, --push                 Specify whether to prefer a push-type streaming pattern over pull-type. Pulling is used by default. \n"
    "  -omv, --output-max-vel        If specified, the program outputs the maximum velocity magnitude as a comma-seperated list of floats with a trailing comma \n"
    "  -ov, --output-velocity        If specified, the program outputs the velocity magnitude at each node as #-seperated CSV tables of NY columns and NX rows \n"
    "  -ovf, --output-velocity-field If specified, the program outputs the components of the velocity field at each node as #-seperated CSV tables of NY columns and NX rows, first all x-components, then all y-components \n"
    "  -f, --file                    Specify a filename to write to instead of printing output to std::cout \n"
    "  -sw, --shear-wave-decay       Simulate a shear-wave decay with velocities in x-direction and periodic boundaries. This is the default. \n"
    "  -ld, --lid-driven-cavity      Simulate a lid driven cavity with bounce-back solid walls and a moving wall at the top, where the fluid is initially at rest. \n"
    "  -gpus, --gpus                 (not in the reference, which is started under mpirun) Cut the lattice into this many y-slabs, one per GPU of this node; the environment variable LBM_GPUS does the same \n";

static UINT parse_uint(const std::string &s)
{
    UINT x = 0;
    const char *first = s.data(), *last = s.data() + s.size();
    auto [ptr, ec] = std::from_chars(first, last, x, 10);
    if (ec == std::errc()) {
        if (ptr == last) return x;
        throw std::invalid_argument("pattern '" + s + "' does not match to the end");
    }
    if (ec == std::errc::invalid_argument) throw std::invalid_argument("pattern '" + s + "' not found");
    throw std::range_error("'" + s + "' not representable");
}

static double parse_double(const std::string &s)
{
    if (s.empty()) throw std::invalid_argument("pattern '" + s + "' not found");
    char *end = nullptr;
    const double v = std::strtod(s.c_str(), &end);
    if (end == s.c_str()) throw std::invalid_argument("pattern '" + s + "' not found");
    if (*end != '\0') throw std::invalid_argument("pattern '" + s + "' does not match to the end");
    return v;
}

static bool looks_like_option(const std::string &s)
{
    // a token starting with '-' that is not a (negative) number is an option name
    if (s.size() < 2 || s[0] != '-') return false;
    char *end = nullptr;
    std::strtod(s.c_str(), &end);
    return !(end != s.c_str() && *end == '\0');
}

static Options parse_args(int argc, char **argv)
{
    Options o;
    std::string first_output, first_scene;  // full names of the first flag used in an exclusive group
    auto exclusive = [](std::string &first, const char *full) {
        if (!first.empty() && first != full)
            throw std::runtime_error(std::string("Argument '") + full + "' not allowed with '" + first + "'");
        first = full;
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&]() -> std::string {
            if (i + 1 >= argc || looks_like_option(argv[i + 1]))
                throw std::runtime_error("Too few arguments for '" + a + "'.");
            return argv[++i];
        };
        if (a == "-h" || a == "--help") {
            std::cout << kUsage;
            std::exit(0);
        } else if (a == "-v" || a == "--version") {
            std::cout << "1.0" << std::endl;
            std::exit(0);
        } else if (a == "-nx") o.nx = parse_uint(value());
        else if (a == "-ny") o.ny = parse_uint(value());
        else if (a == "-of" || a == "--output-frequency") o.out_every = parse_uint(value());
        else if (a == "-s" || a == "--steps") o.steps = parse_uint(value());
        else if (a == "-w" || a == "--omega") o.omega = parse_double(value());
        else if (a == "-rho" || a == "--density") o.rho = parse_double(value());
        else if (a == "-u" || a == "--initial-velocity") o.u = parse_double(value());
        else if (a == "-tx" || a == "--x-grid-points") o.tx = parse_uint(value());
        else if (a == "-ty" || a == "--y-grid-points") o.ty = parse_uint(value());
        else if (a == "-push" || a == "--push") o.push = true;
        else if (a == "-omv" || a == "--output-max-vel") { exclusive(first_output, "-omv/--output-max-vel"); o.output = MAX_VEL; }
        else if (a == "-ov" || a == "--output-velocity") { exclusive(first_output, "-ov/--output-velocity"); o.output = VEL_MAGS; }
        else if (a == "-ovf" || a == "--output-velocity-field") { exclusive(first_output, "-ovf/--output-velocity-field"); o.output = VEL_FIELD; }
        else if (a == "-f" || a == "--file") { o.file = value(); o.use_file = true; }
        else if (a == "-sw" || a == "--shear-wave-decay") { exclusive(first_scene, "-sw/--shear-wave-decay"); o.scene = SHEAR_WAVE; }
        else if (a == "-ld" || a == "--lid-driven-cavity") { exclusive(first_scene, "-ld/--lid-driven-cavity"); o.scene = LID_DRIVEN; }
        else if (a == "-gpus" || a == "--gpus") o.gpus = (unsigned)parse_uint(value());   // not a flag of the reference (it runs under mpirun -n)
        else throw std::runtime_error("Unknown argument: " + a);
    }
    return o;
}

// ------------------------------------------------------------------------------------------------
static void die_hip(hipError_t e, const char *what)
{
    if (e != hipSuccess) {
        std::cerr << what << ": " << hipGetErrorString(e) << std::endl;
        std::exit(1);
    }
}
static void die_lbm(int status, const char *what)
{
    if (status != LBM_OK) {
        std::cerr << what << ": " << lbm_status_string(status) << ": " << lbm_last_error() << std::endl;
        std::exit(1);
    }
}

// C++ stream formatting of a float/double at default precision (what the reference's operator<<
// produces): %g with 6 significant digits.  fmt_g.h: identical text, 40 ns per value against 90 for
// std::to_chars and 320 for snprintf.
static inline int fmt_g(char *dst, size_t cap, FLT v) { return fmtg::format_g6<FLT>(dst, cap, v); }

// NY lines (y=0 first) of NX comma-separated values, then "#" (io.cpp:234-254, :268-286).
// check_nan: the -ov path stops at the first NaN, having already printed what precedes it.
static bool write_block(std::ostream &os, const FLT *a, UINT nx, UINT ny, bool check_nan)
{
    // where the scan order of io.cpp:236-243 meets the first NaN (ny*nx = none)
    UINT stop = nx * ny;
    if (check_nan)
        for (UINT i = 0; i < nx * ny; ++i)
            if (a[i] != a[i]) { stop = i; break; }
    const UINT full_rows = stop / nx;          // rows printed whole
    // Formatting is the expensive part of an output step (a million numbers per 1024^2 frame): rows
    // are formatted in chunks by a few host threads that take the next chunk as they finish one, and
    // this thread writes the chunks in order as they become ready -- formatting and writing overlap.
    unsigned nthreads = std::thread::hardware_concurrency();
    if (nthreads == 0) nthreads = 1;
    if (nthreads > 16) nthreads = 16;
    const UINT rows_per_chunk = std::max<UINT>(1, std::min<UINT>(64, (full_rows + 4 * nthreads - 1) / (4 * nthreads)));
    const UINT nchunks = (full_rows + rows_per_chunk - 1) / rows_per_chunk;
    std::vector<std::string> chunks(nchunks);
    std::vector<std::atomic<int>> ready(nchunks);
    for (auto &r : ready) r.store(0, std::memory_order_relaxed);
    std::atomic<UINT> next{0};
    auto format_chunk = [&](UINT c) {
        const UINT y0 = c * rows_per_chunk, y1 = std::min(full_rows, y0 + rows_per_chunk);
        std::string &out = chunks[c];
        out.reserve((size_t)(y1 - y0) * nx * 12);
        char tmp[64];
        for (UINT y = y0; y < y1; ++y) {
            for (UINT x = 0; x < nx; ++x) {
                out.append(tmp, (size_t)fmt_g(tmp, sizeof tmp, a[y * nx + x]));
                if (x < nx - 1) out.push_back(',');
            }
            out.push_back('\n');
        }
        ready[c].store(1, std::memory_order_release);
    };
    auto worker = [&]() {
        for (UINT c = next.fetch_add(1); c < nchunks; c = next.fetch_add(1)) format_chunk(c);
    };
    if (nthreads > 1 && nchunks > 1) {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t + 1 < nthreads && t < nchunks; ++t) pool.emplace_back(worker);
        for (UINT c = 0; c < nchunks; ++c) {
            while (!ready[c].load(std::memory_order_acquire)) std::this_thread::yield();
            os.write(chunks[c].data(), (std::streamsize)chunks[c].size());
            std::string().swap(chunks[c]);
        }
        for (auto &th : pool) th.join();
    } else {
        for (UINT c = 0; c < nchunks; ++c) {
            format_chunk(c);
            os.write(chunks[c].data(), (std::streamsize)chunks[c].size());
        }
    }
    if (stop < nx * ny) {
        // the row that holds the NaN: what precedes it is printed, then the run ends (io.cpp:239-243)
        std::string line;
        char tmp[64];
        const UINT y = full_rows, xn = stop % nx;
        for (UINT x = 0; x < xn; ++x) {
            line.append(tmp, (size_t)fmt_g(tmp, sizeof tmp, a[y * nx + x]));
            line.push_back(',');
        }
        os << line;
        std::cerr << "NaN encountered at x=" << xn << " y=" << y << std::endl;
        return false;
    }
    os << "#" << std::endl;
    return true;
}

// Page-locked host memory for the samples: a device-to-host copy into it runs on its own stream
// while the sweeps go on (a copy into pageable memory is staged through the host and holds its stream up).
struct PinnedSample {
    FLT *p = nullptr;
    size_t n = 0;
    void resize(size_t count)
    {
        if (count <= n) return;
        if (p) (void)hipHostFree(p);
        die_hip(hipHostMalloc((void **)&p, count * sizeof(FLT), hipHostMallocDefault), "page-locked sample buffer");
        n = count;
    }
    FLT *data() const { return p; }
    ~PinnedSample() { if (p) (void)hipHostFree(p); }
};

struct Run {
    Options o;
    std::ostream *out = &std::cout;
    std::ofstream file;
    void *f = nullptr, *buf = nullptr, *vel = nullptr;
    PinnedSample host;
    double max_vel = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;                       // samples travel to the host beside the sweeps
    hipEvent_t sample_ready = nullptr, sample_home = nullptr; // velocities computed / copied
    bool sample_in_flight = false;
    long long sweep_ns = 0;  // time spent in the steps alone (reported before the final line)
};

// output() of io.cpp:177-294, in two halves so that the text of one sample is formatted while the
// device already runs the next time steps: fetch_output brings the sample to the host (and waits for
// it), emit_output formats and prints it.
static void fetch_output(Run &r)
{
    const Options &o = r.o;
    switch (o.output) {
        case MAX_VEL:
            die_lbm(lbm_max_velocity(r.f, kPrec, o.nx, o.ny, 0, o.ny, r.stream, &r.max_vel), "max velocity");
            break;
        case VEL_MAGS:
        case VEL_FIELD: {
            // the velocities are formed on the sweep stream, in order with the sweeps; the copy to the
            // host follows on its own stream, and the sweep stream goes on with the next time steps
            // (emit_output waits for the copy: sample_home)
            const size_t count = (o.output == VEL_FIELD ? 2 : 1) * (size_t)o.nx * o.ny;
            if (o.output == VEL_MAGS)
                die_lbm(lbm_velocity_magnitudes(r.f, r.vel, kPrec, o.nx, o.ny, 0, o.ny, r.stream), "velocity magnitudes");
            else
                die_lbm(lbm_compute_velocities(r.f, r.vel, kPrec, o.nx, o.ny, 0, o.ny, r.stream), "velocities");
            r.host.resize(count);
            die_hip(hipEventRecord(r.sample_ready, r.stream), "event");
            die_hip(hipStreamWaitEvent(r.copy_stream, r.sample_ready, 0), "wait");
            die_hip(hipMemcpyAsync(r.host.data(), r.vel, count * sizeof(FLT), hipMemcpyDeviceToHost, r.copy_stream), "copy");
            die_hip(hipEventRecord(r.sample_home, r.copy_stream), "event");
            r.sample_in_flight = true;
        } break;
        default: break;
    }
}

static bool emit_output(Run &r)
{
    const Options &o = r.o;
    if (r.sample_in_flight) {
        die_hip(hipEventSynchronize(r.sample_home), "waiting for the sample");
        r.sample_in_flight = false;
    }
    switch (o.output) {
        case MAX_VEL: {
            char tmp[64];
            fmt_g(tmp, sizeof tmp, (FLT)r.max_vel);
            (*r.out) << tmp << "," << std::flush;
        } break;
        case VEL_MAGS:
            if (!write_block(*r.out, r.host.data(), o.nx, o.ny, true)) return false;
            break;
        case VEL_FIELD:
            write_block(*r.out, r.host.data(), o.nx, o.ny, false);
            write_block(*r.out, r.host.data() + (size_t)o.nx * o.ny, o.nx, o.ny, false);
            break;
        default: break;
    }
    return true;
}

// Start-up work of the current device that belongs ahead of the clock (the reference has Kokkos::initialize
// there, main.cpp:412 against :421): the sweeps' code objects, which HIP would otherwise load at the first
// launch, and the runtime's copy queues -- a process's first host-to-device copy of more than a few KB takes
// 4 ms on a device (the initialisers send a table of per-row values that way), the first copy back likewise
// (every sample of the outputs is one).
static void warm_device(hipStream_t stream)
{
    die_lbm(lbm_load_kernels(kPrec), "loading the kernels");
    const size_t bytes = 1 << 20;
    void *dev = nullptr;
    die_hip(hipMalloc(&dev, bytes), "start-up probe");
    std::vector<char> host(bytes, 0);
    die_hip(hipMemcpyAsync(dev, host.data(), bytes, hipMemcpyHostToDevice, stream), "start-up probe");
    die_hip(hipMemcpyAsync(host.data(), dev, bytes, hipMemcpyDeviceToHost, stream), "start-up probe");
    die_hip(hipStreamSynchronize(stream), "start-up probe");
    (void)hipFree(dev);
}

// run_simulation_single_node of main.cpp:31-89
static bool run_simulation(Run &r)
{
    const Options &o = r.o;
    const size_t bytes = (size_t)9 * o.nx * o.ny * sizeof(FLT);
    // LBM_TRACE_SETUP=1: where the set-up time goes, on stderr (ahead of the last line)
    const bool trace = std::getenv("LBM_TRACE_SETUP") != nullptr;
    const auto t_setup = std::chrono::high_resolution_clock::now();
    auto mark = [&](const char *what) {
        if (!trace) return;
        std::cerr << "  set-up: " << what << " at "
                  << std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - t_setup).count()
                  << " us" << std::endl;
    };
    die_hip(hipMalloc(&r.f, bytes), "allocating f");
    die_hip(hipMalloc(&r.buf, bytes), "allocating buf");
    if (o.output == VEL_MAGS || o.output == VEL_FIELD)
        die_hip(hipMalloc(&r.vel, (size_t)2 * o.nx * o.ny * sizeof(FLT)), "allocating vel");
    mark("arrays allocated");

    std::cerr << "running on" << o.nx << "x" << o.ny << std::endl;

    // -tx/-ty are the reference's MDRange tile; here -tx is the workgroup-width hint of the one-step and
    // of the multi-step sweeps (kept only when it names a supported width) and -ty is ignored
    if (o.tx == 64 || o.tx == 128 || o.tx == 256) lbm_set_launch_shape((unsigned)o.tx, 0);

    if (o.scene == LID_DRIVEN) die_lbm(lbm_init_rest(r.f, kPrec, o.nx, o.ny, o.rho, r.stream), "init_rest");
    else die_lbm(lbm_init_shearwave(r.f, kPrec, o.nx, o.ny, o.rho, o.u, 0, o.ny, 0, r.stream), "init_shearwave");
    mark("initial state written");

    const auto t0 = std::chrono::high_resolution_clock::now();
    // The reference's loop body is {output if due, one sweep, swap}.  Between two outputs the
    // iterations are handed to the library in runs of up to lbm_plan_steps_scene(): same populations,
    // fewer passes over memory.
    bool pending = false;      // a fetched sample waits to be printed
    unsigned queued = 0;       // launches handed to the device since the sample was fetched
    for (UINT t = 0; t < o.steps;) {
        if (o.out_every > 0 && t % o.out_every == 0) {
            if (pending && !emit_output(r)) return false;
            fetch_output(r);
            pending = true;
            queued = 0;
        }
        UINT until = o.steps;
        if (o.out_every > 0) {
            const UINT next_out = (t / o.out_every + 1) * o.out_every;
            if (next_out < until) until = next_out;
        }
        const int k = lbm_plan_steps_scene(kPrec, o.nx, o.ny, until - t, o.scene == LID_DRIVEN);
        if (k < 1) die_lbm(LBM_ERR_INVALID_ARGUMENT, "planning the next sweep");
        int s;
        if (o.scene == LID_DRIVEN) {
            // the reference's lid scene always pushes (main.cpp:72-74); -push is ignored there too
            s = lbm_push_lid_fused(r.f, r.buf, kPrec, o.nx, o.ny, o.omega, o.rho, o.u, 0, o.ny, 0, o.ny, k, r.stream);
        } else {
            s = o.push ? lbm_push_periodic_fused(r.f, r.buf, kPrec, o.nx, o.ny, o.omega, 0, o.ny, 1, k, r.stream)
                       : lbm_pull_periodic_fused(r.f, r.buf, kPrec, o.nx, o.ny, o.omega, 0, o.ny, 1, k, r.stream);
        }
        die_lbm(s, "sweep");
        std::swap(r.f, r.buf);
        t += (UINT)k;
        // print the sample once the device has its work up to the next sample (or a good stretch
        // of it) in its queue
        if (pending && (t >= until || ++queued >= 64)) {
            if (!emit_output(r)) return false;
            pending = false;
        }
    }
    if (pending && !emit_output(r)) return false;
    die_hip(hipStreamSynchronize(r.stream), "sync");
    r.sweep_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::high_resolution_clock::now() - t0).count();
    return true;
}

#include "slabs.h"
#include "cart.h"

int main(int argc, char *argv[])
{
    Run r;
#if !COALESCING
    // macros.h:7 / global.h:37-41: f and buf are [NY][NX][9] for the whole run
    die_lbm(lbm_set_layout(LBM_LAYOUT_AOS), "set_layout");
#endif
#if !UNROLL_LOOPS
    // macros.h:6 / pull-push.cpp:259-263: the periodic sweeps use the *_for spelling of the relaxation
    die_lbm(lbm_set_unroll_loops(0), "set_unroll_loops");
#endif
    try {
        r.o = parse_args(argc, argv);
        if (r.o.use_file) {
            r.file.open(r.o.file);
            if (!r.file.is_open()) {
                std::cerr << "Failed to open file: " << r.o.file << "\n";
                std::exit(1);
            }
            r.out = &r.file;
        }
    } catch (const std::exception &err) {
        std::cerr << err.what() << std::endl;
        std::cerr << kUsage;
        std::exit(1);
    }
    if (r.o.nx == 0 || r.o.ny == 0) {
        std::cerr << "lattice extents must be positive" << std::endl;
        return 1;
    }

    // runtime start-up happens before the clock starts, where the reference has Kokkos::initialize
    // (main.cpp:417-421)
    die_hip(hipSetDevice(0), "hipSetDevice");
    die_hip(hipFree(nullptr), "HIP runtime start-up");
    die_hip(hipStreamCreate(&r.stream), "stream");
    die_hip(hipStreamCreateWithFlags(&r.copy_stream, hipStreamNonBlocking), "stream");
    die_hip(hipEventCreateWithFlags(&r.sample_ready, hipEventDisableTiming), "event");
    die_hip(hipEventCreateWithFlags(&r.sample_home, hipEventDisableTiming), "event");
    {
        // the device code is loaded at the first launch: one one-element kernel of the library here, still
        // on the start-up side of the clock
        void *probe = nullptr;
        die_hip(hipMalloc(&probe, 256), "start-up probe");
        die_hip(hipMemset(probe, 0x3f, 256), "start-up probe");
        die_lbm(lbm_selftest_reciprocal(probe, (char *)probe + 128, kPrec, 1, r.stream), "start-up probe");
        die_hip(hipStreamSynchronize(r.stream), "start-up probe");
        (void)hipFree(probe);
    }

    // How many y-slabs: -gpus N, else LBM_GPUS=N; a build with USE_MPI set (the reference's multi-rank
    // build, macros.h:8, started there as `mpirun -n N ./main`) takes every visible device.  One slab
    // is the single-device path.
    unsigned nslabs = r.o.gpus;
    if (nslabs == 0) {
        const char *e = std::getenv("LBM_GPUS");
        if (e && *e) nslabs = (unsigned)std::strtoul(e, nullptr, 10);
    }
#if USE_MPI
    if (nslabs == 0) {
        int ndev = 1;
        die_hip(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
        nslabs = (unsigned)ndev;
    }
#endif
    if (nslabs > 1 && lbm_get_layout() != LBM_LAYOUT_SOA) {
        std::cerr << "the y-slab decomposition runs on [9][NY][NX] arrays only (COALESCING=true)" << std::endl;
        return 1;
    }

    // The reference cuts its lattice in both directions (MPI_Dims_create, main.cpp:103-104).  Here y-slabs are the
    // default (their exchange overlaps the sweep); LBM_DECOMPOSITION=2d cuts the N blocks of -gpus N into the grid
    // MPI_Dims_create would choose, LBM_GRID=PXxPY names the grid outright (src/cart.h).
    UINT grid_x = 0, grid_y = 0;
    if (const char *e = std::getenv("LBM_GRID")) {
        unsigned long long gx = 0, gy = 0;
        if (std::sscanf(e, "%llux%llu", &gx, &gy) != 2 || gx == 0 || gy == 0) {
            std::cerr << "LBM_GRID must look like 4x2 (blocks along x, blocks along y)" << std::endl;
            return 1;
        }
        grid_x = gx; grid_y = gy;
    } else if (const char *e = std::getenv("LBM_DECOMPOSITION")) {
        if (std::strcmp(e, "2d") == 0 || std::strcmp(e, "cart") == 0) {
            if (nslabs > 1) cart_dims(nslabs, grid_x, grid_y);
        } else if (std::strcmp(e, "slabs") != 0) {
            std::cerr << "LBM_DECOMPOSITION must be slabs or 2d" << std::endl;
            return 1;
        }
    }
    const bool cart = grid_x * grid_y > 1;
    if (cart && lbm_get_layout() != LBM_LAYOUT_SOA) {
        std::cerr << "the 2-D decomposition runs on [9][NY][NX] arrays only (COALESCING=true)" << std::endl;
        return 1;
    }
    SlabRun slab_run;
    CartRun cart_run;
    if (cart) {
        if (!cart_startup(cart_run, (unsigned)(grid_x * grid_y))) return 1;
    } else if (nslabs > 1 && !slab_startup(r.o, slab_run, nslabs)) {
        return 1;
    }

    // single device: code objects and copy queues ahead of the clock; the decompositions do it per device
    // in their start-up
    if (!cart && nslabs <= 1) warm_device(r.stream);

    const auto start = std::chrono::high_resolution_clock::now();
    const bool success = cart ? run_simulation_cart(r, cart_run, grid_x, grid_y)
                              : (nslabs > 1 ? run_simulation_slabs(r, slab_run, nslabs) : run_simulation(r));
    // the reference stops its clock without a fence (main.cpp:428); here every kernel has finished
    die_hip(hipDeviceSynchronize(), "sync");
    const auto end = std::chrono::high_resolution_clock::now();
    const UINT span = (UINT)std::chrono::duration_cast<std::chrono::nanoseconds>(end - start).count();

    const Options &o = r.o;
    if (success && r.sweep_ns > 0 && o.steps > 0) {
        const double mlups = (double)o.nx * (double)o.ny * (double)o.steps * 1000.0 / (double)r.sweep_ns;
        const double gbs = mlups * 1e-3 * 18.0 * sizeof(FLT);
        std::cerr << "time loop only (without allocation and initialisation): |" << r.sweep_ns << "|ns => "
                  << (UINT)mlups << " MLUPS, " << (UINT)gbs << " GB/s algorithmic = "
                  << (int)(gbs / 80.0 + 0.5) << " % of the 8 TB/s HBM peak" << std::endl;
    }
    const UINT mlups = span ? (o.nx * o.ny * o.steps * 1000) / span : 0;
    std::cerr << std::endl
              << o.nx * o.ny * o.steps << " lattice updates in |" << span << "|ns => " << mlups << " MLUPS"
              << std::endl;
    (*r.out) << std::flush;
    slab_shutdown(slab_run);
    (void)hipFree(r.f);
    (void)hipFree(r.buf);
    if (r.vel) (void)hipFree(r.vel);
    return success ? 0 : 1;
}
