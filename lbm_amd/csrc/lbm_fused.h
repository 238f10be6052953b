// Several time steps per pass over memory (temporal blocking) for the D2Q9 BGK sweeps on gfx950.
//
// A single collide-and-stream step moves 18 populations per node through HBM for ~110 flops: the
// one-step sweeps of lbm_device.h sit at the memory roofline with the vector ALUs 80 % idle.  These
// kernels advance K = 1..8 steps of the reference's time loop (main.cpp:64-83: sweep, swap, sweep,
// ...) while reading every population once and writing it once:
//
//   * one wavefront owns a strip of 64 x 8 bytes of a row and marches it through a segment of
//     rows; one row of 9 x 512 bytes is loaded per iteration (prefetched one iteration ahead).
//     On large lattices four wavefronts with adjacent strips form a workgroup (fused_group_waves):
//     they share nothing but the cache lines their windows have in common;
//     Eight bytes per lane, not sixteen: a lone wavefront per SIMD cannot hide its own issue and
//     memory stalls (measured 62 % VALU-active at K=4 with 16-byte lanes and 440 VGPRs); with half
//     the nodes per lane the K=4 pull kernel fits 256 VGPRs, two wavefronts share a SIMD and the
//     fp32 sweep gains 11 % (211 -> 234 GLUPS at 16384^2);
//   * the time levels between the loaded and the stored one never reach memory: per level a ring
//     of nine vectors holds the populations that the next rows still need ((2,5,6) of two rows,
//     (0,1,3) of one row), so iteration i relaxes row R at level 1, row R-1 at level 2, ... and
//     stores row R-K+1 at level K.  The last two rings sit in registers, earlier ones in LDS that
//     only this wavefront touches;
//   * the x neighbours that the streaming step needs come from the adjacent lanes through DPP
//     wavefront shifts -- no barriers, every global access a naturally aligned 8 bytes;
//   * the dependency cone is paid for with redundant work at the strip and segment edges: four
//     lanes (32 bytes) of x halo per side -- eight for fp64 beyond four steps, one node per step --
//     so that the 56 (48) stored lanes of a strip cover whole 64-byte lines (partially written
//     lines cost 20 % of the achieved HBM rate), and K or K-1 rows of y halo per segment end;
//   * arithmetic per node and time step is the one-step kernels' `collide`, in the same order, so
//     the result is bit-identical to K one-step sweeps.
// Tried and rejected: leaving the streaming-step loop early while a segment ramps up (step s has
// nothing real to do before iteration 2s) gains 2-4 % on 2048^2..4096^2 lattices but the extra
// control flow costs 1.5-4 % on large ones (16384^2 fp32: 294 -> 290, lid 292 -> 280 GLUPS);
// strips shared by 2 or 4 wavefronts (boundary scalars through LDS, one barrier
// per streaming step) halve or quarter the x halo and the over-fetched lines, but the barriers cost
// more than that saves (fp32 K=4: 245 -> 227 / 222 GLUPS, fp64 131 -> 115 / 112).
//
// Pull scheme (stored state: post-collision populations):  [gather from memory, relax] then K-1 x
// [gather from the rings, relax], store.   Push scheme (stored state: post-streaming populations):
// [relax the loaded row] then K-1 x [gather, relax], one more gather, store.
//
// Lid-driven cavity: the gathers apply the half-way bounce-back map of pull-push.cpp:496-528 read
// from the destination side (as lid_pull_node / lid_push_scatter do for one step): a population
// whose upstream node lies behind a wall is the destination node's own opposite population, the
// two that bounce off the lid carry -+ rho_eq*u_lid/6.  Only the two outermost strips hold wall
// nodes and only a handful of iterations per cavity touch the bottom or the lid row, so each
// iteration makes one wave-uniform test and runs either the periodic code or the code with the wall
// cases in it (fused_body::iteration): testing inside every streaming step cost the fp64 sweeps
// 12 % (16384^2, six steps: 132 -> 149 GLUPS with this and the DPP change below).
// Wavefronts whose stored nodes lie K nodes or more from every wall do not run that code at all but
// the periodic sweep's (k_fused): nothing a wall has touched within K steps reaches what they store.
//
// Array-of-structures lattices ([NY][NX][9]): same kernels, other load / store code (load_aos).
#pragma once
#include <type_traits>
#include "lbm_device.h"

#pragma clang fp contract(off)

// Streaming steps whose rings stay in registers (the rest live in LDS); tuning harnesses override
// (0 = the built-in choice, fused_reg_rings()).
#ifndef LBM_FUSED_REG_RINGS_PULL
#define LBM_FUSED_REG_RINGS_PULL 0
#endif
#ifndef LBM_FUSED_REG_RINGS_PUSH
#define LBM_FUSED_REG_RINGS_PUSH 0
#endif
// Rows requested ahead of their use (fp32 with two: no gain, 16384^2 six steps 318 vs 316 GLUPS on
// constant data, and the cavity kernels reach 255 VGPRs).
#ifndef LBM_FUSED_PREFETCH_F64
#define LBM_FUSED_PREFETCH_F64 2
#endif
#ifndef LBM_FUSED_PREFETCH_F32
#define LBM_FUSED_PREFETCH_F32 1
#endif
// Cache policy bits of the row loads: 0 = default.  2 (non-temporal) loses 15 % at five to seven
// steps per launch and 4 % at eight (16384^2 fp32 contracted: the lines that neighbouring strips
// share no longer meet in the cache); 1 (sc0) and 16 (sc1) change nothing.
#ifndef LBM_FUSED_LOAD_POLICY
#define LBM_FUSED_LOAD_POLICY 0
#endif
// Cache policy bits of the row stores: 2 = non-temporal (the write stream is never re-read within a
// sweep); plain stores (0) lose 11 % at five and six steps per launch, 6 % at seven, 3 % at eight.
#ifndef LBM_FUSED_STORE_POLICY
#define LBM_FUSED_STORE_POLICY 2
#endif
// Fewest wavefronts per SIMD the register allocation must allow (2: up to 256 VGPRs).
#ifndef LBM_FUSED_MIN_WAVES
#define LBM_FUSED_MIN_WAVES 2
#endif
// Rows of input staged in the wavefront's LDS ahead of their use by LDS-DMA loads (0 = rows are loaded
// into registers one or two iterations ahead; see fused_stage_rows()).
#ifndef LBM_FUSED_STAGE_ROWS
#define LBM_FUSED_STAGE_ROWS 0
#endif
// Wavefronts per workgroup (0 = the built-in choice, fused_group_waves()).
#ifndef LBM_FUSED_GROUP_WAVES
#define LBM_FUSED_GROUP_WAVES 0
#endif

namespace lbm {

constexpr int kFusedLaneBytes = 8;   // bytes of a row owned by one lane: 2 fp32 or 1 fp64 node
#ifndef LBM_FUSED_HALO_BYTES
#define LBM_FUSED_HALO_BYTES 32
#endif
constexpr int kFusedHaloBytes = LBM_FUSED_HALO_BYTES;  // granule of the x halo on either side of a strip: stored lanes then cover whole 64-byte lines
constexpr int kFusedMaxSteps = 8;    // 16384^2 fp32 pull, GLUPS at 5/6/7/8 steps per launch: strict arithmetic 308/325/334/335,
                                     // contracted 326/377/410/424; fp64 contracted 154/162/167/171 (five steps and fewer run
                                     // at the memory roof, each further step costs less than it saves until the rings of
                                     // nine or more steps would cut the residency)
// halo lanes per strip side for K steps: K nodes, rounded up to whole 32-byte granules -- or, NARROW,
// to whole lanes only.  Narrow halos leave 58 (fp64: 54) stored lanes at five steps where the
// granule leaves 56 (48); their strips end inside 64-byte lines, which costs a lattice that streams
// from HBM a fifth of its rate and a lattice that lives in the caches nothing: 1024-node rows are 9
// strips instead of 10, and a 1024^2 launch of five steps runs in 29 us instead of 31.
template <typename T, int V, int K, bool NARROW = false> constexpr int fused_halo_lanes()
{
#ifdef LBM_FUSED_TIMING_NO_HALO
    return 0;   // timing harnesses only: strips without an x halo (wrong results)
#endif
    constexpr int granule = NARROW ? V * (int)sizeof(T) : kFusedHaloBytes;
    return ((K * (int)sizeof(T) + granule - 1) / granule) * granule / (V * (int)sizeof(T));
}
// The one step count built with narrow halos: what small lattices take per launch (lbm_plan_steps).
constexpr int kFusedNarrowSteps = 5;

// Rings kept in registers.  With row addresses in scalar registers (row_load / row_store) the
// periodic fp32 kernels have room for four (16384^2, eight steps: 2 rings 388 GLUPS, 3 rings 410,
// 4 rings 412; push 344 / 388 / 411).  The fp64 push kernels hold three from four steps on (seven
// steps: 244 VGPRs; with one ring fewer in LDS a group of four wavefronts fits the compute unit
// twice, see fused_group_waves: 158 -> 173 GLUPS) and two at three steps, where a third spills; the
// fp64 pull kernels spill with three (-2 %) and badly with four (171 -> 81 GLUPS), the rolled
// arithmetic holds three (push, seven steps: 252 -> 272 GLUPS for the same reason as fp64 push; pull
// +1 %) and spills with four, the [NY][NX][9] kernels hold three in fp32 (two in the pull cavity, and two
// with the contracted arithmetic, whose schedule spills 12-24 bytes with three) and spill with three in fp64, the
// cavity kernels carry the wall cases in registers as well (fp32 push: three fit, pull: two).
// No instantiation of the library uses scratch memory.
template <typename T, int V, int K, bool PUSH, bool LID, bool AOS, int ARITH>
constexpr int fused_reg_rings()
{
    constexpr int forced = PUSH ? LBM_FUSED_REG_RINGS_PUSH : LBM_FUSED_REG_RINGS_PULL;
    if (forced > 0) return forced;
    if (AOS) return (sizeof(T) == 4 && (PUSH || !LID) && ARITH != kArithContracted) ? 3 : 2;
    if (ARITH == kArithRolled) return 3;
    if (sizeof(T) == 8) return PUSH && K >= 4 ? 3 : 2;
    if (LID) return PUSH ? 3 : 2;      // pull cavity kernels spill with three from six steps on
    return 4;
}

// Streaming steps done in registers or LDS (everything but the step that reads memory).
template <int K, bool PUSH, bool AOS> constexpr int fused_ring_steps() { return (PUSH || AOS) ? K : K - 1; }

// Cavity kernels on [9][NY][NX] arrays run the periodic sweep's code in wavefronts far from every
// wall (k_fused) -- from six steps on: launches of five steps and fewer are what small cavities
// take, which gain nothing from it (1024^2: +-1 %) and lose 4 % to the larger kernel (cfg: `-ld
// -nx 1024 -ny 1024`, 136 -> 129 GLUPS).
template <bool LID, bool AOS, int K> constexpr bool fused_cavity_bulk() { return LID && !AOS && K > kFusedNarrowSteps; }
// Register rings of that code inside a cavity kernel: the periodic sweep's own choice, but three
// where that is four in a pull kernel (next to the wall code's registers a fourth ring spills; as a
// kernel of its own the periodic sweep loses half a percent with three).
template <typename T, int V, int K, bool PUSH, bool AOS, int ARITH>
constexpr int fused_bulk_rings()
{
    constexpr int r = fused_reg_rings<T, V, K, PUSH, false, AOS, ARITH>();
    return (!PUSH && r > 3) ? 3 : r;
}

// Rows of input that a wavefront keeps in flight or landed in its own LDS (LDS-DMA: the loads have no
// register destination, so looking further ahead costs LDS, not registers).  Periodic sweeps over
// [9][NY][NX] arrays with eight bytes per lane only; the row length must then be a whole number of
// 16-byte pieces.  Measured at 16384^2 fp32, one to four rows ahead: launches of five and six steps
// take exactly as long as with the register prefetch (they run at the rate at which memory serves
// strips marched down by single wavefronts, with or without the arithmetic), seven and eight lose
// 2-25 % to the LDS the slots take from the rings.  Off in the library; kept for the tuning harness.
template <typename T, int V, int K, bool PUSH, bool LID, bool AOS, int ARITH>
constexpr int fused_stage_rows()
{
    if (AOS || LID || V * (int)sizeof(T) != kFusedLaneBytes) return 0;
    return LBM_FUSED_STAGE_ROWS;
}

// LDS of one wavefront: the rings that do not live in registers (4.5 KiB each) and the staged rows
// of input (4.5 KiB each); for a cavity kernel the larger of its two bodies' needs.
template <typename T, int V, int K, bool PUSH, bool LID, bool AOS, int ARITH>
constexpr int fused_lds_bytes_per_wave()
{
    constexpr int steps = fused_ring_steps<K, PUSH, AOS>();
    constexpr int own = steps - fused_reg_rings<T, V, K, PUSH, LID, AOS, ARITH>();
    constexpr int bulk = fused_cavity_bulk<LID, AOS, K>() ? steps - fused_bulk_rings<T, V, K, PUSH, AOS, ARITH>() : 0;
    constexpr int rings = own > bulk ? own : bulk;
    return ((rings > 0 ? rings : 0) + fused_stage_rows<T, V, K, PUSH, LID, AOS, ARITH>()) * 9 * kWave * V * (int)sizeof(T);
}

// Wavefronts per workgroup: a launch-time choice, 1 .. kFusedMaxGroupWaves (blockDim.x = 64 x that,
// dynamic LDS = fused_lds_bytes_per_wave() x that).  The wavefronts of a group take adjacent strips
// and share nothing but the cache: they start together on one compute unit, so the 128-byte lines
// that the 512-byte windows of neighbouring strips have in common are fetched once instead of once
// per strip (strips dealt one per workgroup land on different XCDs, each with an L2 of its own).
// fused_group_waves() is the choice for lattices of many rounds of wavefronts: four where two groups
// still fit the LDS of a compute unit (two wavefronts per SIMD), else one.  16384^2, groups of four
// against single wavefronts: fp64 seven steps (384-byte strips share two lines of five with either
// neighbour) 18 % fewer bytes fetched, 172 -> 180 GLUPS; fp32 eight steps 18 % fewer bytes fetched,
// 443 -> 447 GLUPS.  Small lattices keep one wavefront per workgroup, free to go to any compute unit.
constexpr int kFusedMaxGroupWaves = LBM_FUSED_GROUP_WAVES > 4 ? LBM_FUSED_GROUP_WAVES : 4;
template <typename T, int V, int K, bool PUSH, bool LID, bool AOS, int ARITH>
constexpr int fused_group_waves()
{
    if (LBM_FUSED_GROUP_WAVES > 0) return LBM_FUSED_GROUP_WAVES;
    if (AOS) return 1;
    return 8 * fused_lds_bytes_per_wave<T, V, K, PUSH, LID, AOS, ARITH>() <= 160 * 1024 ? kFusedMaxGroupWaves : 1;
}

// Most wavefronts a SIMD can hold of a kernel: what the LDS of the compute unit allows, spread over
// its four SIMDs and rounded up; 8 (the hardware's limit) for kernels without rings in LDS.
template <typename T, int V, int K, bool PUSH, bool LID, bool AOS, int ARITH>
constexpr int fused_max_waves_per_simd()
{
    constexpr int lds = fused_lds_bytes_per_wave<T, V, K, PUSH, LID, AOS, ARITH>();
    constexpr int per_simd = lds == 0 ? 8 : (160 * 1024 / (lds == 0 ? 1 : lds) + 3) / 4;
    return per_simd < 2 ? 2 : (per_simd > 8 ? 8 : per_simd);
}

// Fewest wavefronts per SIMD the register allocation must leave room for: LBM_FUSED_MIN_WAVES where
// the LDS lets that many be resident at all.
template <typename T, int V, int K, bool PUSH, bool LID, bool AOS, int ARITH>
constexpr int fused_min_waves_per_simd()
{
    constexpr int most = fused_max_waves_per_simd<T, V, K, PUSH, LID, AOS, ARITH>();
    return LBM_FUSED_MIN_WAVES < most ? LBM_FUSED_MIN_WAVES : most;
}

// The LDS of a workgroup, carved up by fused_body.
extern __shared__ __align__(16) unsigned char fused_lds_raw[];

struct FusedGeom {
    uint32_t nx, ny;        // row length, rows in the array
    uint32_t y_lo, y_hi;    // rows to advance
    int wrap_y;             // rows outside [0,ny) wrap around (whole periodic lattice on this device)
    uint32_t seg_rows;      // rows stored per wavefront
    uint32_t nstrips, nsegs;
    int xcd_map;            // != 0: deal whole segments to the XCDs (see fused_work)
    int bulk_code;          // cavity: != 0 lets wavefronts far from every wall run the periodic sweep's code (k_fused)
    int64_t y_goff;         // lid scenes: global row = row + y_goff
    uint64_t ny_glob;       //             cavity height
    // lid scenes on small grids: the strips at the side walls run the slower code with the wall
    // cases in every iteration, so they get shorter segments (0 = the same as everyone)
    uint32_t seg_rows_wall, nsegs_wall;
};

// Value held by the lane below / above in the wavefront (DPP wave_shr:1 / wave_shl:1).  Lane 0 /
// lane 63 have no such neighbour and receive zero (bound_ctrl); they are halo lanes whose result is
// never stored.  Receiving zero rather than keeping the own value leaves the destination register
// free to differ from the source, which spares a copy wherever the unshifted value stays live.
template <typename T>
__device__ __forceinline__ T from_lane_below(T v)
{
    constexpr int N = sizeof(T) / 4;
    union { T t; int w[N]; } a, b;
    a.t = v;
#pragma unroll
    for (int k = 0; k < N; ++k) b.w[k] = __builtin_amdgcn_update_dpp(0, a.w[k], 0x138, 0xf, 0xf, true);
    return b.t;
}
template <typename T>
__device__ __forceinline__ T from_lane_above(T v)
{
    constexpr int N = sizeof(T) / 4;
    union { T t; int w[N]; } a, b;
    a.t = v;
#pragma unroll
    for (int k = 0; k < N; ++k) b.w[k] = __builtin_amdgcn_update_dpp(0, a.w[k], 0x130, 0xf, 0xf, true);
    return b.t;
}

// d[i] = value at x-1 of a row held as one vector per lane / value at x+1
template <typename T, int V>
__device__ __forceinline__ void arrive_from_left(T (&d)[V], const T (&s)[V])
{
    d[0] = from_lane_below(s[V - 1]);
#pragma unroll
    for (int i = 1; i < V; ++i) d[i] = s[i - 1];
}
template <typename T, int V>
__device__ __forceinline__ void arrive_from_right(T (&d)[V], const T (&s)[V])
{
    d[V - 1] = from_lane_above(s[0]);
#pragma unroll
    for (int i = 0; i < V - 1; ++i) d[i] = s[i + 1];
}

// Row accesses of the [9][NY][NX] sweeps go through buffer descriptors: the 64-bit row base is
// wave-uniform and lives in scalar registers, a lane adds a 32-bit byte offset, and the range check
// of the descriptor drops the access of any lane whose offset lies at or beyond `bytes`.  The
// stores use that instead of a branch around them: lanes of the x halo pass kNoStore as their
// offset, iterations that store nothing pass bytes = 0.  (Stores under a divergent branch made the
// compiler wait for vmcnt(0) where the branch rejoins -- for the stores themselves, every
// iteration; and per-lane 64-bit addresses cost a vector add per access.)
constexpr uint32_t kNoStore = 0x80000000u;   // beyond any row: rows are shorter than 2 GiB here
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <typename T, int V>
__device__ __forceinline__ void row_load(T (&r)[V], const T *row, uint32_t bytes, uint32_t voff)
{
    const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(row), 0, (int)bytes, 0x00020000);
    if constexpr (V * sizeof(T) == 8) {
        union { u32x2 u; T t[V]; } c;
        c.u = __builtin_amdgcn_raw_buffer_load_b64(d, (int)voff, 0, LBM_FUSED_LOAD_POLICY);
#pragma unroll
        for (int i = 0; i < V; ++i) r[i] = c.t[i];
    } else {
        static_assert(V == 1 && sizeof(T) == 4, "8 or 4 bytes per lane");
        union { unsigned u; T t; } c;
        c.u = __builtin_amdgcn_raw_buffer_load_b32(d, (int)voff, 0, 0);
        r[0] = c.t;
    }
}
// non-temporal like store_a: the write stream is never re-read within a sweep
template <typename T, int V>
__device__ __forceinline__ void row_store(T *row, uint32_t bytes, uint32_t voff, const T (&r)[V])
{
    const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc(row, 0, (int)bytes, 0x00020000);
    if constexpr (V * sizeof(T) == 8) {
        union { u32x2 u; T t[V]; } c;
#pragma unroll
        for (int i = 0; i < V; ++i) c.t[i] = r[i];
        __builtin_amdgcn_raw_buffer_store_b64(c.u, d, (int)voff, 0, LBM_FUSED_STORE_POLICY);
    } else {
        union { unsigned u; T t; } c;
        c.t = r[0];
        __builtin_amdgcn_raw_buffer_store_b32(c.u, d, (int)voff, 0, 2);
    }
}


// LDS-DMA staging of one row of input (nine population rows of 512 bytes, the strip of one
// wavefront): loads that write the wavefront's LDS directly and have no register destination.
// An LDS-DMA load moves 16 bytes per lane to LDS address M0 + 16 x lane, so half a wavefront moves
// one 512-byte population row; the nine loads go out as five from lanes 0-31 and four from lanes
// 32-63, population row p landing at lds + 512 p.  Row bases are wave-uniform (scalar registers), a
// lane adds the byte offset of its 16 bytes.  The compiler neither sees these loads nor waits for
// them: stage_wait<N>() does (vmcnt counts loads and stores of a wavefront in issue order), and the
// reads of the LDS slot being refilled must have been issued before (they are waited for here).
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
__device__ __forceinline__ void stage_row_dma(uint32_t lds, uint32_t voff, const void *p0, const void *p1,
                                              const void *p2, const void *p3, const void *p4, const void *p5,
                                              const void *p6, const void *p7, const void *p8)
{
    uint32_t keep_m0;
    uint64_t keep_exec;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b32 %[km], m0\n\t"
        "s_mov_b64 %[ke], exec\n\t"
        "s_mov_b32 m0, %[lds]\n\t"
        "s_mov_b32 exec_lo, -1\n\t"
        "s_mov_b32 exec_hi, 0\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[vo], %[p0]\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[vo], %[p2]\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[vo], %[p4]\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[vo], %[p6]\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[vo], %[p8]\n\t"
        "s_mov_b32 exec_lo, 0\n\t"
        "s_mov_b32 exec_hi, -1\n\t"
        "s_sub_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[vo], %[p7]\n\t"
        "s_sub_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[vo], %[p5]\n\t"
        "s_sub_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[vo], %[p3]\n\t"
        "s_sub_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[vo], %[p1]\n\t"
        "s_mov_b64 exec, %[ke]\n\t"
        "s_mov_b32 m0, %[km]"
        : [km] "=&s"(keep_m0), [ke] "=&s"(keep_exec)
        : [lds] "s"(lds), [vo] "v"(voff), [p0] "s"(p0), [p1] "s"(p1), [p2] "s"(p2), [p3] "s"(p3), [p4] "s"(p4),
          [p5] "s"(p5), [p6] "s"(p6), [p7] "s"(p7), [p8] "s"(p8)
        : "memory", "scc");
}
// Wait until at most N vector memory operations of this wavefront are outstanding.
template <int N> __device__ __forceinline__ void stage_wait()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a six-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// What a lane knows about walls (lid scenes only; all false otherwise).
struct WallFlags {
    bool t, b;        // destination row is the lid row / the bottom row (wave-uniform)
    bool l, r;        // this lane's element 0 is the left wall node / element V-1 the right wall node
    bool w;           // some lane of this wavefront has l or r set (wave-uniform)
};

// Streaming step towards one destination row, sources held in registers:
//   A: populations (2,5,6) of the row below -- or, if fl.b, (4,7,8) of the destination row itself
//   B: populations (0,1,3) of the destination row
//   F: populations (4,7,8) of the row above (ignored if fl.t)
//   TOP: populations (2,5,6) of the destination row (read only if fl.t)
//   own7_l, own6_l: the destination row's populations 7 and 6 at the left wall node,
//   own8_r, own5_r: its populations 8 and 5 at the right wall node (read by wall lanes only)
// Result c[q] = population q arriving at the lane's V nodes.
// LID: a lid scene (its push form adds a zero lid term to two populations on every row);
// WALLS: this call may meet a wall row or a wall node (otherwise fl is not looked at).
template <typename T, int V, bool LID, bool PUSH, bool WALLS = LID>
__device__ __forceinline__ void gather_row(T (&c)[9][V], const T (&A)[3][V], const T (&B)[3][V],
                                           const T (&F)[3][V], const T (&TOP)[3][V], const T own7_l,
                                           const T own6_l, const T own8_r, const T own5_r,
                                           const WallFlags fl, const T df_lid)
{
#pragma unroll
    for (int i = 0; i < V; ++i) { c[0][i] = B[0][i]; c[2][i] = A[0][i]; }
    arrive_from_left<T, V>(c[1], B[1]);
    arrive_from_right<T, V>(c[3], B[2]);
    // the diagonal and vertical movers as they stream in from the rows below and above
    T s5[V], s6[V], s7[V], s8[V];
    arrive_from_left<T, V>(s5, A[1]);
    arrive_from_right<T, V>(s6, A[2]);
    arrive_from_right<T, V>(s7, F[1]);
    arrive_from_left<T, V>(s8, F[2]);
    if (LID && PUSH) {
        // the push form adds the (here zero) lid term to what it streams on (pull-push.cpp:522,528)
#pragma unroll
        for (int i = 0; i < V; ++i) { s6[i] = s6[i] + T(0.); s5[i] = s5[i] - T(0.); }
    }
    if (!WALLS || !(fl.t || fl.b)) {
#pragma unroll
        for (int i = 0; i < V; ++i) { c[5][i] = s5[i]; c[6][i] = s6[i]; c[4][i] = F[0][i]; c[7][i] = s7[i]; c[8][i] = s8[i]; }
    } else {
        // bottom wall: 2 <- 4 (already in A[0]), 5 <- 7, 6 <- 8 of the node itself (pull-push.cpp:520,
        // 523,526); lid: 4 <- 2, 7 <- 5 - df, 8 <- 6 + df of the node itself (:522,525,528).  Selects on
        // wave-uniform flags keep every element in its register.  The empty asm statement keeps the
        // compiler from folding this rare branch into selects that every row would execute.
        asm volatile("; wall row");
#pragma unroll
        for (int i = 0; i < V; ++i) {
            c[5][i] = fl.b ? A[1][i] : s5[i];
            c[6][i] = fl.b ? A[2][i] : s6[i];
            c[4][i] = fl.t ? TOP[0][i] : F[0][i];
            c[7][i] = fl.t ? TOP[1][i] - df_lid : s7[i];
            c[8][i] = fl.t ? TOP[2][i] + df_lid : s8[i];
        }
    }
    // Only the two outermost strips hold wall nodes: everyone else branches around the patches
    // (in fp64 they are a sixth of a node's instructions).
    if (WALLS && fl.w) {
        asm volatile("; wall strip");
        const T z = T(0.);
        if (fl.l) {   // left wall: 1 <- 3, 5 <- 7, 8 <- 6 (+ zero lid term) of the node itself
            c[1][0] = B[2][0];
            if (!fl.b) c[5][0] = own7_l;
            if (!fl.t) c[8][0] = own6_l + z;
        }
        if (fl.r) {   // right wall: 3 <- 1, 6 <- 8, 7 <- 5 (- zero lid term)
            c[3][V - 1] = B[1][V - 1];
            if (!fl.b) c[6][V - 1] = own8_r;
            if (!fl.t) c[7][V - 1] = own5_r - z;
        }
    }
}

// fp32 nodes are relaxed in explicit packed pairs (collide_nodes): 7 % faster than leaving the pairing
// to the compiler in the periodic sweeps (240 -> 256 GLUPS at 16384^2, four steps), 4-10 % in the
// lid-driven ones.
template <typename T, int V, int ARITH = kArithUnrolled>
__device__ __forceinline__ void relax_row(T (&o)[9][V], const T (&c)[9][V], const T om)
{
#ifdef LBM_FUSED_TIMING_NO_RELAX
    // timing harnesses only: the sweep's memory traffic without its arithmetic (wrong results)
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int e = 0; e < V; ++e) o[q][e] = c[q][e] * om;
#else
    collide_nodes<T, V, true, ARITH>(o, c, om);
#endif
}

// Array-of-structures lattices ([NY][NX][9], the reference's VIEW(view, x, y, q) = view(y, x, q) of
// global.h:40): a lane's V nodes are 9 V consecutive values.  They move as nine V-wide accesses and
// are sorted into populations in registers, so the lanes of one access sit 72 (36) bytes apart:
// every 64-byte line is touched by several accesses of the same wavefront and filled or drained
// in the cache, not in one instruction.
template <typename T, int V>
__device__ __forceinline__ void load_aos(T (&d)[9][V], const T *p)
{
    typedef typename VecTypes<T, V>::vec_t vec_t;
    typedef typename VecTypes<T, V>::vec_a vec_a;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const vec_t v = *reinterpret_cast<const vec_a *>(p + k * V);
#pragma unroll
        for (int j = 0; j < V; ++j) d[(k * V + j) % 9][(k * V + j) / 9] = v[j];
    }
}
// Plain stores: the nine accesses of a wavefront fill each line piecewise and must meet in the cache
// (non-temporal ones write the pieces through: 16384^2 fp32, six steps, 270 -> 116 GLUPS).
template <typename T, int V>
__device__ __forceinline__ void store_aos(T *p, const T (&d)[9][V])
{
    typedef typename VecTypes<T, V>::vec_t vec_t;
    typedef typename VecTypes<T, V>::vec_a vec_a;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        vec_t v;
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = d[(k * V + j) % 9][(k * V + j) / 9];
        *reinterpret_cast<vec_a *>(p + k * V) = v;
    }
}

// One row of input as it sits in registers between its load and its use.
template <typename T, int V>
struct RowIn {
    T v[9][V];      // pull: slots by destination population; push: the row's nine populations
    T w[2];         // lid pull, outermost strips: the row's own 6 at x = 0 and own 5 at x = nx-1
};

// What one wavefront of a launch works on.
struct FusedWork {
    uint32_t wave;          // index of the wavefront in its workgroup
    uint32_t strip, seg;    // strip of 64 lanes along x, segment of rows along y
    uint32_t seg_rows;      // rows per segment for this strip
    uint32_t lds_base;      // byte offset of this wavefront's part of the workgroup's LDS (set by the kernel)
    bool valid;             // false: the launch was rounded up, nothing to do
};

// A workgroup of several wavefronts stands for as many consecutive single-wavefront blocks; the
// wavefront's index is read into a scalar register, since row addresses must stay wave-uniform.
// Blocks are numbered strip fastest, and workgroups go to the 8 XCDs round-robin: the strips of
// one group share their common lines in one L2, the neighbours across a group boundary do not.
// xcd_map (single wavefronts only): XCD j takes segments j, j+8, ... and walks all strips of a
// segment in order, so every shared line meets its second reader in the same L2.  Measured at
// 16384^2 fp32: -3 % (and -10 % when each XCD takes a contiguous eighth of every segment's strips
// instead): address-order dispatch is worth more than the shared lines.
template <bool LID>
__device__ __forceinline__ FusedWork fused_work(const FusedGeom &g)
{
    FusedWork w;
    w.wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const uint32_t block = blockIdx.x * (blockDim.x / kWave) + w.wave;
    uint32_t nsegs = g.nsegs;
    w.seg_rows = g.seg_rows;
    if (LID && g.nsegs_wall != 0) {
        // first the wall strips with their own (shorter) segments -- the slowest wavefronts start
        // first --, then the strips without walls, segment by segment
        const uint32_t nwall = g.nstrips < 2 ? g.nstrips : 2u, nmid = g.nstrips - nwall;
        const uint32_t wall_blocks = nwall * g.nsegs_wall;
        if (block < wall_blocks) {
            w.strip = (block % nwall) ? g.nstrips - 1 : 0u;
            w.seg = block / nwall;
            w.seg_rows = g.seg_rows_wall;
            nsegs = g.nsegs_wall;
        } else {
            const uint32_t b = block - wall_blocks;
            w.strip = 1 + b % nmid;
            w.seg = b / nmid;
        }
    } else if (g.xcd_map) {
        const uint32_t xcd = block & 7u, slot = block >> 3;
        w.seg = (slot / g.nstrips) * 8 + xcd;
        w.strip = slot % g.nstrips;
    } else {
        w.strip = block % g.nstrips;
        w.seg = block / g.nstrips;
    }
    w.lds_base = 0;
    w.valid = w.seg < nsegs;
    return w;
}

// K time steps of one scene/scheme.  nstrips * nsegs wavefronts, in workgroups of blockDim.x / 64.
// AOS: the arrays are [NY][NX][9].  The pull scheme then reads each row's own nine (post-collision)
// populations like the push scheme does and streams every step in registers, the first one included.
// ARITH: spelling of the relaxation (lbm_device.h).  kArithRolled is the reference's UNROLL_LOOPS=false
// arithmetic (periodic scenes only, as there); kArithContracted the explicit fused multiply-adds.
// NARROW: halos of whole lanes (fused_halo_lanes).  RINGS: rings kept in registers, 0 = fused_reg_rings().
// `work` names the strip and segment of this wavefront and its part of the workgroup's LDS.
template <typename T, int V, int K, bool PUSH, bool LID, bool AOS = false, int ARITH = kArithUnrolled, bool NARROW = false,
          int RINGS = 0>
__device__ __forceinline__ void fused_body(const T *__restrict__ f, T *__restrict__ buf, const FusedGeom &g,
                                           const FusedWork &work, const T om, const T df_lid)
{
    static_assert(V * sizeof(T) == kFusedLaneBytes || (V == 1 && sizeof(T) == 4),
                  "a lane owns kFusedLaneBytes of a row (one fp32 node when rows have an odd length)");
    constexpr int H = fused_halo_lanes<T, V, K, NARROW>();
    constexpr int W = kWave - 2 * H;         // vectors stored per strip
#ifndef LBM_FUSED_TIMING_NO_HALO
    static_assert(H * V >= K, "x halo too narrow for K steps");
#endif
    constexpr bool OWN = PUSH || AOS;        // a row is loaded as its own nine populations
    constexpr int NG = fused_ring_steps<K, PUSH, AOS>();   // streaming steps done in registers or LDS
    constexpr int D = NG;                    // rows between a segment's first row and the first row relaxed
    constexpr int NR = NG > 0 ? NG : 1;

    const uint32_t strip = work.strip, seg = work.seg, seg_rows = work.seg_rows;
    const uint32_t lane = threadIdx.x % kWave;
    const uint32_t nxv = g.nx / V;
    int64_t vx = (int64_t)strip * W - H + (int64_t)lane;
    if (LID) {
        // no wrap: lanes beyond a side wall read the wall vector; what they compute is never used
        vx = vx < 0 ? 0 : (vx >= (int64_t)nxv ? (int64_t)nxv - 1 : vx);
    } else {
        vx %= (int64_t)nxv;
        if (vx < 0) vx += nxv;
    }
    const uint32_t x0 = (uint32_t)vx * V;
    const uint32_t wcount = min((uint32_t)W, nxv - strip * W);
    const bool store_lane = lane >= (uint32_t)H && lane < H + wcount;
    const uint32_t row_bytes = g.nx * (uint32_t)sizeof(T);
    const uint32_t xoff = x0 * (uint32_t)sizeof(T);                // a lane's byte offset in any row
    const uint32_t xoff_store = store_lane ? xoff : kNoStore;
    WallFlags fl0;
    fl0.t = false; fl0.b = false;
    fl0.l = LID && x0 == 0;
    fl0.r = LID && x0 + V == g.nx;
    // does any lane of this wavefront sit at a side wall (wave-uniform)?
    const bool wall_wave = LID && __ballot(fl0.l || fl0.r) != 0;
    fl0.w = wall_wave;

    const uint32_t ys = g.y_lo + seg * seg_rows;
    const uint32_t ye = min(ys + seg_rows, g.y_hi);
    const uint32_t iters = (ye - ys) + D + NG;
    const size_t plane = (size_t)g.ny * g.nx;

    // array row of lattice row R: wrapped for a periodic lattice, otherwise clamped into the array
    // (clamped rows are only ever read for nodes outside the dependency cone of the stored rows)
    auto row_index = [&](int64_t R) -> uint32_t {
        if (g.wrap_y) {
            R %= (int64_t)g.ny;
            if (R < 0) R += g.ny;
        } else {
            R = R < 0 ? 0 : (R >= (int64_t)g.ny ? (int64_t)g.ny - 1 : R);
        }
        return (uint32_t)R;
    };
    // Sliding window over the rows the next load needs, advanced incrementally (a 64-bit modulo per
    // row index would cost more scalar instructions than the relaxation costs vector ones): Rn is
    // the lattice row relaxed first in the iteration being loaded, y_d / y_m / y_u the array rows of
    // Rn-1, Rn, Rn+1.  Loads are requested in iteration order.
    int32_t Rn = (int32_t)ys - D;
    uint32_t y_d = row_index((int64_t)Rn - 1), y_m = row_index((int64_t)Rn), y_u = row_index((int64_t)Rn + 1);
    auto sat32 = [](int64_t v) -> int32_t { return v > INT32_MAX ? INT32_MAX : (v < INT32_MIN ? INT32_MIN : (int32_t)v); };
    auto advance_rows = [&]() {
        ++Rn;
        y_d = y_m;
        y_m = y_u;
        if (g.wrap_y) {
            y_u = (y_u + 1 == g.ny) ? 0u : y_u + 1;
        } else {
            const int32_t r = Rn + 1;
            y_u = r < 0 ? 0u : (r >= (int32_t)g.ny ? g.ny - 1 : (uint32_t)r);
        }
    };
    auto load_row = [&](RowIn<T, V> &d) {
        const uint32_t y = y_m, yd = y_d, yu = y_u;
        if (AOS) {
            load_aos<T, V>(d.v, f + ((size_t)y * g.nx + x0) * 9);
        } else if (PUSH) {
#pragma unroll
            for (int q = 0; q < 9; ++q) row_load<T, V>(d.v[q], rowp(f, plane, q, y, g.nx), row_bytes, xoff);
        } else {
            // pull-push.cpp:13-21.  Lid scenes load the wall rows like any other row (the row
            // beyond the wall is the wall row itself, clamped) and fetch what the walls reflect
            // when the row is used (iteration_of): tests here would cost every row scalar work.
            row_load<T, V>(d.v[7], rowp(f, plane, 7, yu, g.nx), row_bytes, xoff);
            row_load<T, V>(d.v[4], rowp(f, plane, 4, yu, g.nx), row_bytes, xoff);
            row_load<T, V>(d.v[8], rowp(f, plane, 8, yu, g.nx), row_bytes, xoff);
            row_load<T, V>(d.v[3], rowp(f, plane, 3, y, g.nx), row_bytes, xoff);
            row_load<T, V>(d.v[0], rowp(f, plane, 0, y, g.nx), row_bytes, xoff);
            row_load<T, V>(d.v[1], rowp(f, plane, 1, y, g.nx), row_bytes, xoff);
            row_load<T, V>(d.v[6], rowp(f, plane, 6, yd, g.nx), row_bytes, xoff);
            row_load<T, V>(d.v[2], rowp(f, plane, 2, yd, g.nx), row_bytes, xoff);
            row_load<T, V>(d.v[5], rowp(f, plane, 5, yd, g.nx), row_bytes, xoff);
            if (LID && wall_wave) {
                // per-lane addresses keep these on the vector memory path next to the row loads
                d.w[0] = rowp(f, plane, 6, y, g.nx)[x0];
                d.w[1] = rowp(f, plane, 5, y, g.nx)[x0 + V - 1];
            }
        }
        advance_rows();
    };

    // Per in-flight streaming step ("ring"): A = (2,5,6) two rows back, B = (0,1,3) one row back,
    // C = (2,5,6) one row back, w7/w8 = populations 7 / 8 one row back at the wall nodes.  The last
    // kRegRings rings live in registers; earlier ones live in this wavefront's own LDS (4.5 KiB per
    // ring, no barriers), where A and C are two sets that swap roles with the parity of the
    // iteration instead of being copied.  Measured at 16384^2 fp32: with two register rings K=4 runs
    // as fast as with all rings in registers and K=5/6 become possible (K=6: +12 %); with three
    // register rings the pull kernels of K>=5 spill.
    constexpr int kRegRings = RINGS > 0 ? RINGS : fused_reg_rings<T, V, K, PUSH, LID, AOS, ARITH>();
    constexpr int NLDS = NG > kRegRings ? NG - kRegRings : 0;
    constexpr int NREG = NG - NLDS > 0 ? NG - NLDS : 1;
    typedef typename VecTypes<T, V>::vec_t ring_vec_t;
    // this wavefront's part of the workgroup's LDS: [ring][set*3 + k | 6 + k][lane]
    ring_vec_t (*ring_lds)[9][kWave] = reinterpret_cast<ring_vec_t (*)[9][kWave]>(
        fused_lds_raw + work.lds_base);
    T ringA[NREG][3][V], ringB[NREG][3][V], ringC[NREG][3][V], ring7[NR], ring8[NR];
#pragma unroll
    for (int s = 0; s < NREG; ++s) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < V; ++e) { ringA[s][k][e] = T(1); ringB[s][k][e] = T(1); ringC[s][k][e] = T(1); }
    }
#pragma unroll
    for (int s = 0; s < NR; ++s) { ring7[s] = T(1); ring8[s] = T(1); }
    auto lds_get = [&](T (&d)[V], int ring, uint32_t slot) {
        const ring_vec_t v = ring_lds[ring][slot][lane];
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] = v[e];
    };
    auto lds_put = [&](int ring, uint32_t slot, const T (&d)[V]) {
        ring_vec_t v;
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = d[e];
        ring_lds[ring][slot][lane] = v;
    };

    // Lid scenes, pull, wavefronts of the two outermost strips: the side walls reflect the wall nodes'
    // own 7, 6 (left wall; element 0 of a left-wall lane is x = 0) and 8, 5 (right wall; element V-1
    // of a right-wall lane is x = nx-1) back into the row.  Own 6 and 5 are loaded with the row
    // (RowIn::w).  Own 7 and 8 of row R are what row R-1 pulls from above, so they are carried over
    // from the previous iteration's input; the first iteration loads them.
    T own7_left = T(0), own8_right = T(0);
    if (LID && !OWN && wall_wave) {
        const uint32_t y = row_index((int64_t)ys - D);
        own7_left = rowp(f, plane, 7, y, g.nx)[x0];
        own8_right = rowp(f, plane, 8, y, g.nx)[x0 + V - 1];
    }

    // Iteration i relaxes row Rc = ys - D + i first and then takes the rows Rc - 1, Rc - 2, ... one
    // streaming step further each.  to_bottom / to_lid: rows between Rc and the bottom / the lid row.
    // `walls` is a compile-time flag of the call: false = no wall row and no wall node takes part in
    // this iteration, and the code is that of the periodic sweeps.
    auto iteration_of = [&](const RowIn<T, V> &in, uint32_t i, int32_t to_bottom, int32_t to_lid, auto walls) {
        constexpr bool WALLS = decltype(walls)::value;
        auto wall_rows_below = [&](int k, bool &t, bool &b) {   // row Rc - k
            b = WALLS && to_bottom == k;
            t = WALLS && to_lid == k;
        };
        T o[9][V];
        if (PUSH) {
            relax_row<T, V, ARITH>(o, in.v, om);
        } else if (AOS) {
#pragma unroll
            for (int q = 0; q < 9; ++q)
#pragma unroll
                for (int e = 0; e < V; ++e) o[q][e] = in.v[q][e];
        } else {
            T A[3][V], B[3][V], F[3][V], c[9][V];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                A[0][e] = in.v[2][e]; A[1][e] = in.v[5][e]; A[2][e] = in.v[6][e];
                B[0][e] = in.v[0][e]; B[1][e] = in.v[1][e]; B[2][e] = in.v[3][e];
                F[0][e] = in.v[4][e]; F[1][e] = in.v[7][e]; F[2][e] = in.v[8][e];
            }
            WallFlags fl = fl0;
            wall_rows_below(0, fl.t, fl.b);
            // What the lid and the bottom wall reflect comes from row Rc itself and is loaded here,
            // when it is needed: only the iterations at those walls wait for it.  At the lid the
            // row's own (2,5,6) stand in for what would arrive from above, at the bottom wall its
            // own (4,7,8) for what would arrive from below.
            T w7 = T(0), w6 = T(0), w8 = T(0), w5 = T(0);
            if (WALLS) {
                const uint32_t y = row_index((int64_t)ys - D + (int64_t)i);
                if (fl.t) {
                    load_a<T, V>(F[0], rowp(f, plane, 2, y, g.nx) + x0);
                    load_a<T, V>(F[1], rowp(f, plane, 5, y, g.nx) + x0);
                    load_a<T, V>(F[2], rowp(f, plane, 6, y, g.nx) + x0);
                }
                if (fl.b) {
                    load_a<T, V>(A[0], rowp(f, plane, 4, y, g.nx) + x0);
                    load_a<T, V>(A[1], rowp(f, plane, 7, y, g.nx) + x0);
                    load_a<T, V>(A[2], rowp(f, plane, 8, y, g.nx) + x0);
                }
                w7 = own7_left; w6 = in.w[0]; w8 = own8_right; w5 = in.w[1];
                // what this row pulls from above is the next row's own
                own7_left = in.v[7][0];
                own8_right = in.v[8][V - 1];
            }
            gather_row<T, V, LID, false, WALLS>(c, A, B, F, F, w7, w6, w8, w5, fl, df_lid);
            relax_row<T, V, ARITH>(o, c, om);
        }
#pragma unroll
        for (int s = 0; s < NG; ++s) {
            // row Rc - s is held in o; the destination row is the one below it
            T F[3][V], c[9][V];
#pragma unroll
            for (int e = 0; e < V; ++e) { F[0][e] = o[4][e]; F[1][e] = o[7][e]; F[2][e] = o[8][e]; }
            WallFlags fl = fl0;
            wall_rows_below(s + 1, fl.t, fl.b);
            // what the next rows need of row Rc - s
            bool rt, rb;
            wall_rows_below(s, rt, rb);
            if (s < NLDS) {
                // keep the LDS reads of this step from being hoisted above the previous step's
                // arithmetic: hoisted, every LDS ring costs 18 more live registers and the kernel spills
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t par = i & 1u;           // set `par` holds row r-2, set `par^1` row r-1
                T A[3][V], B[3][V], C[3][V];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    lds_get(A[k], s, par * 3 + k);
                    lds_get(C[k], s, (par ^ 1u) * 3 + k);
                    lds_get(B[k], s, 6 + k);
                }
                gather_row<T, V, LID, PUSH, WALLS>(c, A, B, F, C, ring7[s], C[2][0], ring8[s], C[1][V - 1], fl, df_lid);
                if (WALLS && rb) {
                    // this is the bottom row: next iteration it bounces its own (4,7,8)
                    lds_put(s, (par ^ 1u) * 3 + 0, o[4]); lds_put(s, (par ^ 1u) * 3 + 1, o[7]); lds_put(s, (par ^ 1u) * 3 + 2, o[8]);
                }
                lds_put(s, par * 3 + 0, o[2]); lds_put(s, par * 3 + 1, o[5]); lds_put(s, par * 3 + 2, o[6]);
                lds_put(s, 6 + 0, o[0]); lds_put(s, 6 + 1, o[1]); lds_put(s, 6 + 2, o[3]);
            } else {
                constexpr int kNone = 0;
                const int q = s >= NLDS ? s - NLDS : kNone;   // register ring of this step
                gather_row<T, V, LID, PUSH, WALLS>(c, ringA[q], ringB[q], F, ringC[q], ring7[s], ringC[q][2][0],
                                            ring8[s], ringC[q][1][V - 1], fl, df_lid);
                if (WALLS && rb) {
                    // this is the bottom row: next iteration it bounces its own (4,7,8)
                    asm volatile("; bottom row");   // a branch, not selects on every row
#pragma unroll
                    for (int e = 0; e < V; ++e) { ringA[q][0][e] = o[4][e]; ringA[q][1][e] = o[7][e]; ringA[q][2][e] = o[8][e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < V; ++e) { ringA[q][0][e] = ringC[q][0][e]; ringA[q][1][e] = ringC[q][1][e]; ringA[q][2][e] = ringC[q][2][e]; }
                }
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    ringB[q][0][e] = o[0][e]; ringB[q][1][e] = o[1][e]; ringB[q][2][e] = o[3][e];
                    ringC[q][0][e] = o[2][e]; ringC[q][1][e] = o[5][e]; ringC[q][2][e] = o[6][e];
                }
            }
            if (WALLS) {   // read by wall lanes only, and a wavefront with wall lanes always comes here
                ring7[s] = o[7][0];
                ring8[s] = o[8][V - 1];
            }
            if (PUSH && s == NG - 1) {
#pragma unroll
                for (int q = 0; q < 9; ++q)
#pragma unroll
                    for (int e = 0; e < V; ++e) o[q][e] = c[q][e];
            } else {
                relax_row<T, V, ARITH>(o, c, om);
            }
        }
        if (AOS) {
            if (i >= (uint32_t)(D + NG) && i < iters && store_lane) {
                const uint32_t yo = ys + i - (D + NG);
                store_aos<T, V>(buf + ((size_t)yo * g.nx + x0) * 9, o);
            }
        } else {
            // no branch: the descriptor drops what must not be stored
            const bool stores = i >= (uint32_t)(D + NG) && i < iters;
            const uint32_t yo = stores ? ys + i - (D + NG) : ys;
            const uint32_t bytes = stores ? row_bytes : 0u;
#pragma unroll
            for (int q = 0; q < 9; ++q) row_store<T, V>(rowp(buf, plane, q, yo, g.nx), bytes, xoff_store, o[q]);
        }
    };
    // Lid scenes: one wave-uniform test per iteration picks the code.  Walls take part if the
    // wavefront holds wall nodes, or if one of the rows Rc .. Rc - NG that this iteration gathers
    // into or keeps for the next one is the bottom or the lid row.  Everything else -- all but two
    // strips and a few iterations of the first and last segments -- runs the periodic code: as
    // separate tests inside every streaming step the wall cases cost the fp64 lid sweeps 8-15 %.
    auto iteration = [&](const RowIn<T, V> &in, uint32_t i) {
        if (LID) {
            // rows between row Rc and the bottom row / the lid row, saturated to 32 bits
            const int64_t gy = (int64_t)ys - D + (int64_t)i + g.y_goff;
            const int32_t to_bottom = sat32(gy), to_lid = sat32(gy - ((int64_t)g.ny_glob - 1));
            if (wall_wave || (uint32_t)to_bottom <= (uint32_t)NG || (uint32_t)to_lid <= (uint32_t)NG)
                iteration_of(in, i, to_bottom, to_lid, std::true_type());
            else
                iteration_of(in, i, 0, 0, std::false_type());
        } else {
            iteration_of(in, i, 0, 0, std::false_type());
        }
    };

    // Rows are requested PF iterations ahead of their use; PF + 1 register buffers rotate through
    // a loop unrolled by PF + 1.  One iteration of fp64 work (~0.8 us) is shorter than the memory
    // latency under load, so fp64 looks two rows ahead (37 % of its wave time sat in s_waitcnt with
    // one); fp32 gains nothing from a second row.
    constexpr int SR = fused_stage_rows<T, V, K, PUSH, LID, AOS, ARITH>();
    if constexpr (SR > 0 && !NARROW) {
        // Rows travel through this wavefront's LDS, SR rows ahead: slot j holds row i0 + j until
        // iteration i0 + j has read it into registers, and is then refilled with row i0 + j + SR.
        // Lane l of an LDS-DMA load (l mod 32 in its half of the wavefront) fetches the 16 bytes of
        // the 8-byte lanes 2l and 2l+1 of the strip; first lane, halo and stored width of a strip are
        // even and so must be the row length in 8-byte lanes: the pair then never straddles the
        // periodic wrap.
        static_assert(!LID, "staged rows are built for the periodic sweeps");
        constexpr uint32_t kRowB = 9 * kWave * V * (uint32_t)sizeof(T);
        const uint32_t dl = lane % 32u;
        int64_t px = ((int64_t)strip * W - H + 2 * (int64_t)dl) % (int64_t)nxv;
        if (px < 0) px += nxv;
        const uint32_t doff = (uint32_t)px * (uint32_t)kFusedLaneBytes;
        const uint32_t stage0 = (uint32_t)(uintptr_t)(lds_byte_t *)fused_lds_raw + work.lds_base +
                                (uint32_t)NLDS * kRowB;
        // population held by slot position p: the order of the loads of the unstaged form
        constexpr int kPullPop[9] = {7, 4, 8, 3, 0, 1, 6, 2, 5};
        // request the next row into the slot at byte offset `slot` of the staging area
        auto stage_issue = [&](uint32_t slot) {
            const uint32_t y = y_m, yd = y_d, yu = y_u;
            const uint32_t dst = stage0 + slot;
            if (PUSH) {
                stage_row_dma(dst, doff, rowp(f, plane, 0, y, g.nx), rowp(f, plane, 1, y, g.nx), rowp(f, plane, 2, y, g.nx),
                              rowp(f, plane, 3, y, g.nx), rowp(f, plane, 4, y, g.nx), rowp(f, plane, 5, y, g.nx),
                              rowp(f, plane, 6, y, g.nx), rowp(f, plane, 7, y, g.nx), rowp(f, plane, 8, y, g.nx));
            } else {
                stage_row_dma(dst, doff, rowp(f, plane, 7, yu, g.nx), rowp(f, plane, 4, yu, g.nx), rowp(f, plane, 8, yu, g.nx),
                              rowp(f, plane, 3, y, g.nx), rowp(f, plane, 0, y, g.nx), rowp(f, plane, 1, y, g.nx),
                              rowp(f, plane, 6, yd, g.nx), rowp(f, plane, 2, yd, g.nx), rowp(f, plane, 5, yd, g.nx));
            }
            advance_rows();
        };
        auto stage_read = [&](RowIn<T, V> &cur, uint32_t slot) {
            const ring_vec_t(*src)[kWave] = reinterpret_cast<const ring_vec_t(*)[kWave]>(
                fused_lds_raw + work.lds_base + (uint32_t)NLDS * kRowB + slot);
#pragma unroll
            for (int p = 0; p < 9; ++p) {
                const ring_vec_t v = src[p][lane];
#pragma unroll
                for (int e = 0; e < V; ++e) cur.v[PUSH ? p : kPullPop[p]][e] = v[e];
            }
        };
        const T nothing[V] = {};
#pragma unroll
        for (int j = 0; j < SR; ++j) {
            stage_issue((uint32_t)j * kRowB);
            // as many (dropped) stores behind every request as an iteration issues: the count that
            // stage_wait takes is then the same from the first iteration on
#pragma unroll
            for (int q = 0; q < 9; ++q) row_store<T, V>(rowp(buf, plane, q, ys, g.nx), 0u, kNoStore, nothing);
        }
        // operations issued behind the request for a row when its turn comes: the stores of the
        // iteration that requested it, then SR - 1 iterations of nine loads and nine stores
        constexpr int kBehind = 9 + (SR - 1) * 18;
        {
            uint32_t slot = 0;
            for (uint32_t i = 0; i < iters; ++i) {
                stage_wait<kBehind>();
                RowIn<T, V> cur;
                stage_read(cur, slot);
                cur.w[0] = T(0);
                cur.w[1] = T(0);
                stage_issue(slot);
                iteration(cur, i);
                slot = slot + kRowB == SR * kRowB ? 0u : slot + kRowB;
            }
        }
        stage_wait<0>();   // nothing may still be on its way into LDS that the next workgroup owns
        return;
    }
    constexpr int PF = (sizeof(T) == 8 && V == 1) ? LBM_FUSED_PREFETCH_F64 : LBM_FUSED_PREFETCH_F32;
    RowIn<T, V> in[PF + 1];
#pragma unroll
    for (int j = 0; j <= PF; ++j) { in[j].w[0] = T(0); in[j].w[1] = T(0); }
#pragma unroll
    for (int j = 0; j < PF; ++j) load_row(in[j]);
    // Nothing in the loop body is conditional but the stores: the request for row i + PF always goes
    // out (the last PF requests of a segment fetch rows that nobody uses, clamped or wrapped into the
    // array like every row index), and the trip count is rounded up to whole groups of PF + 1
    // iterations (a surplus iteration computes a row outside the dependency cone and stores nothing).
    // Under a condition the loaded registers meet their old values in a phi, and the compiler moves
    // the first shuffles of the next iteration up into the block of the loads -- behind an
    // s_waitcnt vmcnt(0) that stalls the wavefront on the rows it has just asked for, every
    // iteration (or, for a conditional second half, on the stores it has just issued).
    const uint32_t iters_up = (iters + PF) / (PF + 1) * (PF + 1);
    for (uint32_t i0 = 0; i0 < iters_up; i0 += PF + 1) {
#pragma unroll
        for (int j = 0; j <= PF; ++j) {
            load_row(in[(j + PF) % (PF + 1)]);
            iteration(in[j], i0 + j);
        }
    }
}

// At least two wavefronts per SIMD (at most 256 VGPRs): a lone wavefront cannot overlap its own
// scalar bookkeeping, memory issue and hazard stalls with vector work.  Kernels with rings in LDS
// are also told the most a SIMD can hold (fused_max_waves_per_simd): their residency is set by the
// LDS (rolled push, seven steps: seven wavefronts per compute unit), which the compiler cannot see
// in a dynamic allocation, and an instantiation of 160 VGPRs or fewer would otherwise let the
// dispatcher stack three wavefronts on one SIMD and leave another with one (that kernel, 158 VGPRs:
// 236 -> 223 GLUPS).
template <typename T, int V, int K, bool PUSH, bool LID, bool AOS = false, int ARITH = kArithUnrolled, bool NARROW = false>
__global__ void __launch_bounds__(kWave * kFusedMaxGroupWaves)
    __attribute__((amdgpu_waves_per_eu(fused_min_waves_per_simd<T, V, K, PUSH, LID, AOS, ARITH>(), fused_max_waves_per_simd<T, V, K, PUSH, LID, AOS, ARITH>())))
k_fused(const T *__restrict__ f, T *__restrict__ buf, const FusedGeom g, const T om, const T df_lid)
{
    static_assert(ARITH != kArithRolled || (!LID && !AOS), "the rolled arithmetic exists for periodic [9][NY][NX] sweeps only");
    static_assert(!NARROW || (K == kFusedNarrowSteps && !AOS), "narrow halos are built for one step count");
    FusedWork work = fused_work<LID>(g);
    if (!work.valid) return;
    work.lds_base = work.wave * (uint32_t)fused_lds_bytes_per_wave<T, V, K, PUSH, LID, AOS, ARITH>();
    if constexpr (fused_cavity_bulk<LID, AOS, K>()) {
        // Cavity: a wavefront whose stored nodes lie K nodes or more from every wall computes nothing
        // that a wall has touched within K steps (walls act on the wall nodes only), and what its
        // halo lanes and rows compute nearer to a wall -- wrongly, without the wall cases -- lies
        // outside the dependency cone of what it stores.  It runs the periodic sweep's code (rows
        // clamped into the array: the launch has wrap_y = 0), which has no wall cases to carry in
        // registers and keeps a ring more there.  One wave-uniform test per wavefront.
        constexpr int W = kWave - 2 * fused_halo_lanes<T, V, K, NARROW>();
        const uint32_t nxv = g.nx / V;
        const uint32_t first = work.strip * W * V;
        const uint32_t end_v = (work.strip + 1) * W;
        const uint32_t last = (end_v < nxv ? end_v : nxv) * V - 1;
        const uint32_t ys = g.y_lo + work.seg * work.seg_rows;
        const uint32_t ye = min(ys + work.seg_rows, g.y_hi);
        const int64_t gy0 = (int64_t)ys + g.y_goff, gy1 = (int64_t)ye - 1 + g.y_goff;
        const bool bulk = g.bulk_code != 0 && first >= (uint32_t)K && (uint64_t)last + K <= (uint64_t)g.nx - 1 &&
                          gy0 >= K && gy1 + K <= (int64_t)g.ny_glob - 1;
        if (bulk) {
            fused_body<T, V, K, PUSH, false, AOS, ARITH, NARROW, fused_bulk_rings<T, V, K, PUSH, AOS, ARITH>()>(f, buf, g, work, om, df_lid);
            return;
        }
    }
    fused_body<T, V, K, PUSH, LID, AOS, ARITH, NARROW>(f, buf, g, work, om, df_lid);
}

}  // namespace lbm
