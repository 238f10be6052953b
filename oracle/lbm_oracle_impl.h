/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference D2Q9 BGK hot path.
 *
 * This file is included twice by lbm_oracle.c, once with FLT=float (suffix _f32) and once with
 * FLT=double (suffix _f64).  It restates, expression by expression, what the reference computes
 * under /root/reference/src (citations as file:line next to each function).  It is the checker the
 * HIP kernels are compared against; nothing in the product path may call it.
 *
 * Build rule: -ffp-contract=off, no fast-math, so every operation rounds once, exactly where the
 * reference's source text puts it.  (The reference itself is built -O3 -march=native, where GCC
 * may contract a*b+c; that moves results by a few ulp, far inside the 1e-12 / 1e-5 parity budget.)
 *
 * Storage layout: SoA [Q=9][NY][NX], x fastest -- reference global.h:37-41 (COALESCING=true),
 * global.h:112, main.cpp:37-38.  Element (q,y,x) lives at (q*NY + y)*NX + x.
 *
 * Lattice (pull-push.cpp:8-12): q -> (cx,cy): 0:(0,0) 1:(1,0) 2:(0,1) 3:(-1,0) 4:(0,-1)
 * 5:(1,1) 6:(-1,1) 7:(-1,-1) 8:(1,-1).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

#define AT(p, q, y, x) (p)[((size_t)(q) * ny + (size_t)(y)) * nx + (size_t)(x)]

/* BGK relaxation of one node: pull-push.cpp:25-72 (pull), :138-185 (push), :454-483 (lid).
 * The three sites hold the same expressions; only the statement order differs, which does not
 * change any value.  in[q]/out[q] are indexed by channel. */
static inline void FN(collide)(const FLT in[9], FLT out[9], const FLT OMEGA)
{
    const FLT f_0 = in[0], f_1 = in[1], f_2 = in[2], f_3 = in[3], f_4 = in[4];
    const FLT f_5 = in[5], f_6 = in[6], f_7 = in[7], f_8 = in[8];
    const FLT rho = f_0 + f_1 + f_2 + f_3 + f_4 + f_5 + f_6 + f_7 + f_8; /* :25 */
    const FLT one = (FLT)1.0;
    const FLT rho_inv = one / rho;                                         /* :27 */
    const FLT ux = (f_1 - f_3 + f_5 - f_6 - f_7 + f_8) * rho_inv;          /* :32 */
    const FLT uy = (f_2 - f_4 + f_5 + f_6 - f_7 - f_8) * rho_inv;          /* :35 */
    const FLT f45 = (FLT)4.5, f15 = (FLT)1.5;                              /* :39-43 */
    const FLT f49 = (FLT)(4. / 9.), f19 = (FLT)(1. / 9.), f136 = (FLT)(1. / 36.);
    const FLT w_1_36 = rho * f136;
    const FLT w_1_9 = rho * f19;
    const FLT w_4_9 = rho * f49;
    const FLT uxpuy = ux + uy;
    const FLT uxpuy_2 = f45 * uxpuy * uxpuy;
    const FLT uymux = uy - ux;
    const FLT uymux_2 = f45 * uymux * uymux;
    const FLT ux_2 = f45 * ux * ux;
    const FLT uy_2 = f45 * uy * uy;
    const FLT usq = f15 * (ux * ux + uy * uy);
    out[7] = f_7 + OMEGA * ((w_1_36 * (1 - 3 * uxpuy + uxpuy_2 - usq)) - f_7); /* :58 */
    out[4] = f_4 + OMEGA * ((w_1_9 * (1 - 3 * uy + uy_2 - usq)) - f_4);        /* :61 */
    out[8] = f_8 + OMEGA * ((w_1_36 * (1 - 3 * uymux + uymux_2 - usq)) - f_8); /* :64 */
    out[3] = f_3 + OMEGA * ((w_1_9 * (1 - 3 * ux + ux_2 - usq)) - f_3);        /* :66 */
    out[0] = f_0 + OMEGA * ((w_4_9 * (1 - usq)) - f_0);                        /* :68 */
    out[1] = f_1 + OMEGA * ((w_1_9 * (1 + 3 * ux + ux_2 - usq)) - f_1);        /* :69 */
    out[6] = f_6 + OMEGA * ((w_1_36 * (1 + 3 * uymux + uymux_2 - usq)) - f_6); /* :70 */
    out[2] = f_2 + OMEGA * ((w_1_9 * (1 + 3 * uy + uy_2 - usq)) - f_2);        /* :71 */
    out[5] = f_5 + OMEGA * ((w_1_36 * (1 + 3 * uxpuy + uxpuy_2 - usq)) - f_5); /* :72 */
}

/* The reference's other spelling of the relaxation, compiled when UNROLL_LOOPS is false:
 * pull_periodic_for (pull-push.cpp:90-119) and push_periodic_for (:204-231) hold the same loops. */
static inline void FN(collide_for)(const FLT f[9], FLT out[9], const FLT OMEGA)
{
    FLT rho = 0.;                                                          /* :91-94 */
    for (unsigned i = 0; i < 9; ++i) rho += f[i];
    const FLT f1 = (FLT)1.0;
    const FLT rho_inv = f1 / rho;                                          /* :96 */
    const FLT cx[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};                      /* :99-100 */
    const FLT cy[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
    FLT ux = 0., uy = 0.;
    for (unsigned i = 0; i < 9; ++i) {                                     /* :103-106 */
        ux += f[i] * cx[i];
        uy += f[i] * cy[i];
    }
    ux *= rho_inv;
    uy *= rho_inv;
    const FLT w[9] = {4. / 9., 1. / 9., 1. / 9., 1. / 9., 1. / 9., 1. / 36., 1. / 36., 1. / 36., 1. / 36.};
    for (unsigned i = 0; i < 9; ++i) {                                     /* :112-119 */
        const FLT cx_ux = cx[i] * ux;
        const FLT cy_uy = cy[i] * uy;
        const FLT c_i_u_r = cx_ux + cy_uy;
        const FLT u_2 = ux * ux + uy * uy;
        const FLT f_eq = w[i] * rho * (((FLT)1) + ((FLT)3) * c_i_u_r + ((FLT)4.5) * (c_i_u_r * c_i_u_r) - ((FLT)1.5) * u_2);
        out[i] = f[i] + OMEGA * (f_eq - f[i]);
    }
}

/* One contraction of FN(collide) that a compiler is free to apply to the reference's text: the
 * reference is built -O3 -march=native (CMakeLists.txt:18; nvcc and hipcc contract by default too),
 * where every a*b+c of pull-push.cpp:25-72 may become one fused multiply-add.  Which ones do is the
 * compiler's choice and differs from build to build; this function fixes one choice, spelled with
 * explicit fma calls so that it does not depend on the flags of THIS build:
 *     ux*ux + uy*uy          -> fma(ux, ux, uy*uy)
 *     1 -+ 3*a               -> fma(-+3, a, 1)
 *     ... + (f45*a)*a        -> fma(f45*a, a, ...)
 *     (w*bracket) - f        -> fma(w, bracket, -f)
 *     f + OMEGA*(...)        -> fma(OMEGA, ..., f)
 * Sums, the division and every other product round as in FN(collide).  It is the checker of the
 * HIP library's LBM_ARITH_CONTRACTED mode, and is itself held to FN(collide) within the parity
 * budget (1e-5 fp32 / 1e-12 fp64 on populations) and to the recorded decay series by the tests. */
static inline void FN(collide_fma)(const FLT in[9], FLT out[9], const FLT OMEGA)
{
    const FLT f_0 = in[0], f_1 = in[1], f_2 = in[2], f_3 = in[3], f_4 = in[4];
    const FLT f_5 = in[5], f_6 = in[6], f_7 = in[7], f_8 = in[8];
    const FLT rho = f_0 + f_1 + f_2 + f_3 + f_4 + f_5 + f_6 + f_7 + f_8;
    const FLT one = (FLT)1.0, three = (FLT)3.0;
    const FLT rho_inv = one / rho;
    const FLT ux = (f_1 - f_3 + f_5 - f_6 - f_7 + f_8) * rho_inv;
    const FLT uy = (f_2 - f_4 + f_5 + f_6 - f_7 - f_8) * rho_inv;
    const FLT f45 = (FLT)4.5, f15 = (FLT)1.5;
    const FLT f49 = (FLT)(4. / 9.), f19 = (FLT)(1. / 9.), f136 = (FLT)(1. / 36.);
    const FLT w_1_36 = rho * f136;
    const FLT w_1_9 = rho * f19;
    const FLT w_4_9 = rho * f49;
    const FLT uxpuy = ux + uy;
    const FLT uymux = uy - ux;
    const FLT p_s = f45 * uxpuy, p_d = f45 * uymux, p_x = f45 * ux, p_y = f45 * uy;
    const FLT usq = f15 * ORC_FMA(ux, ux, uy * uy);
#define ORC_RELAX(f, w, sign3, a, p) \
    ORC_FMA(OMEGA, ORC_FMA((w), ORC_FMA((p), (a), ORC_FMA((sign3), (a), one)) - usq, -(f)), (f))
    out[7] = ORC_RELAX(f_7, w_1_36, -three, uxpuy, p_s);
    out[4] = ORC_RELAX(f_4, w_1_9, -three, uy, p_y);
    out[8] = ORC_RELAX(f_8, w_1_36, -three, uymux, p_d);
    out[3] = ORC_RELAX(f_3, w_1_9, -three, ux, p_x);
    out[0] = ORC_FMA(OMEGA, ORC_FMA(w_4_9, one - usq, -f_0), f_0);
    out[1] = ORC_RELAX(f_1, w_1_9, three, ux, p_x);
    out[6] = ORC_RELAX(f_6, w_1_36, three, uymux, p_d);
    out[2] = ORC_RELAX(f_2, w_1_9, three, uy, p_y);
    out[5] = ORC_RELAX(f_5, w_1_36, three, uxpuy, p_s);
#undef ORC_RELAX
}

/* What the periodic sweeps call: pull-push.cpp:259-263, :281-285 and the six further
 * `#if UNROLL_LOOPS` sites of the inner / outer forms. */
static inline void FN(collide_sel)(const FLT in[9], FLT out[9], const FLT OMEGA)
{
    if (orc_arith == 1) FN(collide_for)(in, out, OMEGA);
    else if (orc_arith == 2) FN(collide_fma)(in, out, OMEGA);
    else FN(collide)(in, out, OMEGA);
}
/* The lid-driven sweep has one spelling in the reference (pull-push.cpp:454-483), and its
 * contraction. */
static inline void FN(collide_lid_sel)(const FLT in[9], FLT out[9], const FLT OMEGA)
{
    if (orc_arith == 2) FN(collide_fma)(in, out, OMEGA);
    else FN(collide)(in, out, OMEGA);
}

/* Equilibrium used by the initialisers: init.cpp:57-84.  The bracket is evaluated in double even
 * when FLT is float, because 9./2. and 3./2. are double literals (init.cpp:78-83); 1+3*cu and
 * cu*cu and u_2 are formed in FLT first. */
static inline FLT FN(f_eq)(FLT ux, FLT uy, FLT rho_i, unsigned dir)
{
    static const FLT w[9] = {(FLT)(4. / 9.),  (FLT)(1. / 9.),  (FLT)(1. / 9.),
                             (FLT)(1. / 9.),  (FLT)(1. / 9.),  (FLT)(1. / 36.),
                             (FLT)(1. / 36.), (FLT)(1. / 36.), (FLT)(1. / 36.)};
    static const FLT cx[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};
    static const FLT cy[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
    const FLT cx_ux = cx[dir] * ux;
    const FLT cy_uy = cy[dir] * uy;
    const FLT c_i_u_r = cx_ux + cy_uy;
    const FLT u_2 = ux * ux + uy * uy;
    const FLT a = 1 + 3 * c_i_u_r;   /* FLT */
    const FLT cc = c_i_u_r * c_i_u_r; /* FLT */
    const double bracket = ((double)a + 9. / 2. * (double)cc) - 3. / 2. * (double)u_2;
    const FLT wr = w[dir] * rho_i;
    return (FLT)((double)wr * bracket);
}

/* init_shearwave, single-device form: init.cpp:5-34 with USE_MPI=false (LX=LY=0, use_mpi()=0).
 * SIN is sinf for both precisions (global.h:29): the double product 2.0*M_PI*y_scaled is narrowed
 * to float, sinf returns float, and u_init*float is formed in FLT.  y_below/y_glob let a slab of a
 * larger lattice be initialised (init.h:11-12); rows [0,ny) of f get global rows y_below+y. */
void FN(orc_init_shearwave)(FLT *f, uint64_t nx, uint64_t ny, FLT rho_init, FLT u_init,
                            uint64_t y_below, uint64_t y_glob)
{
#pragma omp parallel for schedule(static)
    for (uint64_t y = 0; y < ny; ++y) {
        const FLT y_scaled = ((FLT)(y - 0ull + y_below)) / ((FLT)y_glob);
        const FLT ux = u_init * sinf((float)(2.0 * M_PI * y_scaled));
        FLT feq[9];
        for (unsigned dir = 0; dir < 9; ++dir) feq[dir] = FN(f_eq)(ux, (FLT)0., rho_init, dir);
        for (unsigned dir = 0; dir < 9; ++dir)
            for (uint64_t x = 0; x < nx; ++x) AT(f, dir, y, x) = feq[dir];
    }
}

/* init_rest: init.cpp:36-49. */
void FN(orc_init_rest)(FLT *f, uint64_t nx, uint64_t ny, FLT rho_init)
{
    FLT feq[9];
    for (unsigned dir = 0; dir < 9; ++dir) feq[dir] = FN(f_eq)((FLT)0., (FLT)0., rho_init, dir);
#pragma omp parallel for schedule(static)
    for (uint64_t y = 0; y < ny; ++y)
        for (unsigned dir = 0; dir < 9; ++dir)
            for (uint64_t x = 0; x < nx; ++x) AT(f, dir, y, x) = feq[dir];
}

/* One pull node: gather (pull-push.cpp:13-21), collide, store locally (:58-72). */
static inline void FN(pull_node)(const FLT *f, FLT *buf, uint64_t nx, uint64_t ny, uint64_t x,
                                 uint64_t y, uint64_t xl, uint64_t xr, uint64_t yd, uint64_t yu,
                                 FLT OMEGA)
{
    FLT in[9], out[9];
    in[7] = AT(f, 7, yu, xr);
    in[4] = AT(f, 4, yu, x);
    in[8] = AT(f, 8, yu, xl);
    in[3] = AT(f, 3, y, xr);
    in[0] = AT(f, 0, y, x);
    in[1] = AT(f, 1, y, xl);
    in[6] = AT(f, 6, yd, xr);
    in[2] = AT(f, 2, yd, x);
    in[5] = AT(f, 5, yd, xl);
    FN(collide_sel)(in, out, OMEGA);
    for (unsigned q = 0; q < 9; ++q) AT(buf, q, y, x) = out[q];
}

/* One push node: read locally (pull-push.cpp:126-134), collide, scatter (:175-185). */
static inline void FN(push_node)(const FLT *f, FLT *buf, uint64_t nx, uint64_t ny, uint64_t x,
                                 uint64_t y, uint64_t xl, uint64_t xr, uint64_t yd, uint64_t yu,
                                 FLT OMEGA)
{
    FLT in[9], out[9];
    for (unsigned q = 0; q < 9; ++q) in[q] = AT(f, q, y, x);
    FN(collide_sel)(in, out, OMEGA);
    AT(buf, 7, yd, xl) = out[7];
    AT(buf, 4, yd, x) = out[4];
    AT(buf, 8, yd, xr) = out[8];
    AT(buf, 3, y, xl) = out[3];
    AT(buf, 0, y, x) = out[0];
    AT(buf, 1, y, xr) = out[1];
    AT(buf, 6, yu, xl) = out[6];
    AT(buf, 2, yu, x) = out[2];
    AT(buf, 5, yu, xr) = out[5];
}

/* Row kernels: the interior of a row (x in [1,nx-1)) has xl=x-1, xr=x+1 with no wrap, so it is
 * written over plain row pointers and the compiler may vectorise it (vectorisation does not change
 * any rounding: no reassociation, no contraction in the strict build).  The two edge nodes go
 * through the single-node forms with wrapped indices. */
static inline void FN(pull_row_interior)(const FLT *restrict r0, const FLT *restrict r1,
                                         const FLT *restrict r2, const FLT *restrict r3,
                                         const FLT *restrict r4, const FLT *restrict r5,
                                         const FLT *restrict r6, const FLT *restrict r7,
                                         const FLT *restrict r8, FLT *restrict const w[9],
                                         uint64_t nx, FLT OMEGA)
{
    FLT *restrict w0 = w[0], *restrict w1 = w[1], *restrict w2 = w[2], *restrict w3 = w[3];
    FLT *restrict w4 = w[4], *restrict w5 = w[5], *restrict w6 = w[6], *restrict w7 = w[7];
    FLT *restrict w8 = w[8];
#pragma omp simd
    for (uint64_t x = 1; x < nx - 1; ++x) {
        FLT in[9], out[9];
        in[7] = r7[x + 1]; in[4] = r4[x]; in[8] = r8[x - 1];
        in[3] = r3[x + 1]; in[0] = r0[x]; in[1] = r1[x - 1];
        in[6] = r6[x + 1]; in[2] = r2[x]; in[5] = r5[x - 1];
        FN(collide_sel)(in, out, OMEGA);
        w0[x] = out[0]; w1[x] = out[1]; w2[x] = out[2]; w3[x] = out[3]; w4[x] = out[4];
        w5[x] = out[5]; w6[x] = out[6]; w7[x] = out[7]; w8[x] = out[8];
    }
}

/* Rows [y_lo,y_hi) of an [9][ny][nx] array, x periodic.  wrap_y!=0: y neighbours wrap by
 * comparison (pull-push.cpp:277-280) -- with y_lo=0,y_hi=ny this is pull_periodic_no_mpi
 * (pull-push.cpp:268-288).  wrap_y==0: rows y-1 / y+1 are read as they are (halo rows present),
 * the y part of pull_periodic_inner (pull-push.cpp:291-310). */
void FN(orc_pull_rows)(const FLT *f, FLT *buf, uint64_t nx, uint64_t ny, uint64_t y_lo,
                       uint64_t y_hi, int wrap_y, FLT OMEGA)
{
#pragma omp parallel for schedule(static)
    for (uint64_t y = y_lo; y < y_hi; ++y) {
        const uint64_t yu = wrap_y ? ((y + 1 == ny) ? 0 : y + 1) : y + 1;
        const uint64_t yd = wrap_y ? ((y == 0) ? ny - 1 : y - 1) : y - 1;
        if (nx >= 3) {
            FLT *w[9];
            for (unsigned q = 0; q < 9; ++q) w[q] = &AT(buf, q, y, 0);
            FN(pull_row_interior)(&AT(f, 0, y, 0), &AT(f, 1, y, 0), &AT(f, 2, yd, 0), &AT(f, 3, y, 0),
                                  &AT(f, 4, yu, 0), &AT(f, 5, yd, 0), &AT(f, 6, yd, 0),
                                  &AT(f, 7, yu, 0), &AT(f, 8, yu, 0), w, nx, OMEGA);
        }
        for (uint64_t x = 0; x < nx; x += (nx >= 3 && x == 0) ? nx - 1 : 1) {
            const uint64_t xr = (x + 1 == nx) ? 0 : x + 1;
            const uint64_t xl = (x == 0) ? nx - 1 : x - 1;
            FN(pull_node)(f, buf, nx, ny, x, y, xl, xr, yd, yu, OMEGA);
        }
    }
}

/* Push counterpart: push_periodic_no_mpi (pull-push.cpp:246-266) for wrap_y, the y part of
 * push_periodic_inner (:313-332) otherwise (rows y_lo-1 and y_hi of buf receive pushed values). */
void FN(orc_push_rows)(const FLT *f, FLT *buf, uint64_t nx, uint64_t ny, uint64_t y_lo,
                       uint64_t y_hi, int wrap_y, FLT OMEGA)
{
#pragma omp parallel for schedule(static)
    for (uint64_t y = y_lo; y < y_hi; ++y) {
        const uint64_t yu = wrap_y ? ((y + 1 == ny) ? 0 : y + 1) : y + 1;
        const uint64_t yd = wrap_y ? ((y == 0) ? ny - 1 : y - 1) : y - 1;
        if (nx >= 3) {
            const FLT *restrict r0 = &AT(f, 0, y, 0), *restrict r1 = &AT(f, 1, y, 0);
            const FLT *restrict r2 = &AT(f, 2, y, 0), *restrict r3 = &AT(f, 3, y, 0);
            const FLT *restrict r4 = &AT(f, 4, y, 0), *restrict r5 = &AT(f, 5, y, 0);
            const FLT *restrict r6 = &AT(f, 6, y, 0), *restrict r7 = &AT(f, 7, y, 0);
            const FLT *restrict r8 = &AT(f, 8, y, 0);
            FLT *restrict w7 = &AT(buf, 7, yd, 0), *restrict w4 = &AT(buf, 4, yd, 0);
            FLT *restrict w8 = &AT(buf, 8, yd, 0), *restrict w3 = &AT(buf, 3, y, 0);
            FLT *restrict w0 = &AT(buf, 0, y, 0), *restrict w1 = &AT(buf, 1, y, 0);
            FLT *restrict w6 = &AT(buf, 6, yu, 0), *restrict w2 = &AT(buf, 2, yu, 0);
            FLT *restrict w5 = &AT(buf, 5, yu, 0);
#pragma omp simd
            for (uint64_t x = 1; x < nx - 1; ++x) {
                FLT in[9], out[9];
                in[0] = r0[x]; in[1] = r1[x]; in[2] = r2[x]; in[3] = r3[x]; in[4] = r4[x];
                in[5] = r5[x]; in[6] = r6[x]; in[7] = r7[x]; in[8] = r8[x];
                FN(collide_sel)(in, out, OMEGA);
                w7[x - 1] = out[7]; w4[x] = out[4]; w8[x + 1] = out[8];
                w3[x - 1] = out[3]; w0[x] = out[0]; w1[x + 1] = out[1];
                w6[x - 1] = out[6]; w2[x] = out[2]; w5[x + 1] = out[5];
            }
        }
        for (uint64_t x = 0; x < nx; x += (nx >= 3 && x == 0) ? nx - 1 : 1) {
            const uint64_t xr = (x + 1 == nx) ? 0 : x + 1;
            const uint64_t xl = (x == 0) ? nx - 1 : x - 1;
            FN(push_node)(f, buf, nx, ny, x, y, xl, xr, yd, yu, OMEGA);
        }
    }
}

/* Rectangle [LX,HX) x [LY,HY) without any wrap: pull_periodic_inner / push_periodic_inner
 * (pull-push.cpp:291-332) as the reference's MPI path calls them (main.cpp:321). */
void FN(orc_pull_inner)(const FLT *f, FLT *buf, uint64_t nx, uint64_t ny, FLT OMEGA, uint64_t LX,
                        uint64_t LY, uint64_t HX, uint64_t HY)
{
#pragma omp parallel for schedule(static)
    for (uint64_t y = LY; y < HY; ++y)
        for (uint64_t x = LX; x < HX; ++x)
            FN(pull_node)(f, buf, nx, ny, x, y, x - 1, x + 1, y - 1, y + 1, OMEGA);
}

void FN(orc_push_inner)(const FLT *f, FLT *buf, uint64_t nx, uint64_t ny, FLT OMEGA, uint64_t LX,
                        uint64_t LY, uint64_t HX, uint64_t HY)
{
#pragma omp parallel for schedule(static)
    for (uint64_t y = LY; y < HY; ++y)
        for (uint64_t x = LX; x < HX; ++x)
            FN(push_node)(f, buf, nx, ny, x, y, x - 1, x + 1, y - 1, y + 1, OMEGA);
}

/* Ring x=1, x=nx-2, y=1, y=ny-2 of a halo'd array: pull_periodic_outer / push_periodic_outer
 * (pull-push.cpp:335-429). */
void FN(orc_pull_outer)(const FLT *f, FLT *buf, uint64_t nx, uint64_t ny, FLT OMEGA)
{
    for (uint64_t x = 1; x < nx - 1; ++x) {
        FN(pull_node)(f, buf, nx, ny, x, 1, x - 1, x + 1, 0, 2, OMEGA);
        FN(pull_node)(f, buf, nx, ny, x, ny - 2, x - 1, x + 1, ny - 3, ny - 1, OMEGA);
    }
    for (uint64_t y = 2; y < ny - 2; ++y) {
        FN(pull_node)(f, buf, nx, ny, 1, y, 0, 2, y - 1, y + 1, OMEGA);
        FN(pull_node)(f, buf, nx, ny, nx - 2, y, nx - 3, nx - 1, y - 1, y + 1, OMEGA);
    }
}

void FN(orc_push_outer)(const FLT *f, FLT *buf, uint64_t nx, uint64_t ny, FLT OMEGA)
{
    for (uint64_t x = 1; x < nx - 1; ++x) {
        FN(push_node)(f, buf, nx, ny, x, 1, x - 1, x + 1, 0, 2, OMEGA);
        FN(push_node)(f, buf, nx, ny, x, ny - 2, x - 1, x + 1, ny - 3, ny - 1, OMEGA);
    }
    for (uint64_t y = 2; y < ny - 2; ++y) {
        FN(push_node)(f, buf, nx, ny, 1, y, 0, 2, y - 1, y + 1, OMEGA);
        FN(push_node)(f, buf, nx, ny, nx - 2, y, nx - 3, nx - 1, y - 1, y + 1, OMEGA);
    }
}

/* push_lid_driven: pull-push.cpp:432-531.  All nodes are fluid, the four walls sit half a cell
 * outside; the lid is the wall above y=NY-1 and moves in +x with U_LID.  df uses the initial
 * density RHO_EQ, not the local one (:518).
 *
 * Slab form: rows [y_lo,y_hi) of an [9][ny][nx] array whose row r is global row y_off + r - ... :
 * global row = y_goff + r, global height ny_glob.  With y_goff=0, ny_glob=ny, y_lo=0, y_hi=ny this
 * is exactly the reference function. */
void FN(orc_push_lid_rows)(const FLT *f, FLT *buf, uint64_t nx, uint64_t ny, uint64_t y_lo,
                           uint64_t y_hi, int64_t y_goff, uint64_t ny_glob, FLT OMEGA,
                           FLT RHO_EQ, FLT U_LID)
{
#pragma omp parallel for schedule(static)
    for (uint64_t y = y_lo; y < y_hi; ++y) {
        const uint64_t yg = (uint64_t)((int64_t)y + y_goff);
        for (uint64_t x = 0; x < nx; ++x) {
            FLT in[9], o[9];
            for (unsigned q = 0; q < 9; ++q) in[q] = AT(f, q, y, x);
            FN(collide_lid_sel)(in, o, OMEGA);
            const int l = (x == 0), r = (x == nx - 1);          /* :496-499 */
            const int t = (yg == ny_glob - 1), b = (yg == 0);
            const int lb = l || b, lt = l || t, rb = r || b, rt = r || t; /* :500-503 */
            const uint64_t xl = (x > 0) ? x - 1 : x;             /* :505-508 */
            const uint64_t xr = (x < nx - 1) ? x + 1 : x;
            const uint64_t yu = t ? y : y + 1;
            const uint64_t yd = b ? y : y - 1;
            const FLT f6 = (FLT)6., f0 = (FLT)0.;
            const FLT df = t ? (RHO_EQ * U_LID / f6) : f0;       /* :516-518 */
            AT(buf, lb ? 5 : 7, lb ? y : yd, lb ? x : xl) = o[7];      /* :520 */
            AT(buf, l ? 1 : 3, y, xl) = o[3];                          /* :521 */
            AT(buf, lt ? 8 : 6, lt ? y : yu, lt ? x : xl) = o[6] + df; /* :522 */
            AT(buf, b ? 2 : 4, yd, x) = o[4];                          /* :523 */
            AT(buf, 0, y, x) = o[0];                                   /* :524 */
            AT(buf, t ? 4 : 2, yu, x) = o[2];                          /* :525 */
            AT(buf, rb ? 6 : 8, rb ? y : yd, rb ? x : xr) = o[8];      /* :526 */
            AT(buf, r ? 3 : 1, y, xr) = o[1];                          /* :527 */
            AT(buf, rt ? 7 : 5, rt ? y : yu, rt ? x : xr) = o[5] - df; /* :528 */
        }
    }
}

/* Lid-driven cavity in PULL form.  The reference has no such function (its lid scene always
 * pushes, main.cpp:72-74); this is the streaming map of push_lid_driven (pull-push.cpp:520-528)
 * read from the destination side: population q arriving at (x,y) comes from (x-cx,y-cy) when that
 * node exists, else it is the node's own opposite population bounced off the wall, with the lid
 * correction on the two channels that bounce off the moving wall (6->8 gains df, 5->7 loses df).
 * The stored state is post-collision (stream, then collide), as in pull_periodic. */
void FN(orc_pull_lid_rows)(const FLT *f, FLT *buf, uint64_t nx, uint64_t ny, uint64_t y_lo,
                           uint64_t y_hi, int64_t y_goff, uint64_t ny_glob, FLT OMEGA,
                           FLT RHO_EQ, FLT U_LID)
{
#pragma omp parallel for schedule(static)
    for (uint64_t y = y_lo; y < y_hi; ++y) {
        const uint64_t yg = (uint64_t)((int64_t)y + y_goff);
        for (uint64_t x = 0; x < nx; ++x) {
            const int l = (x == 0), r = (x == nx - 1);
            const int t = (yg == ny_glob - 1), b = (yg == 0);
            const FLT f6 = (FLT)6., f0 = (FLT)0.;
            const FLT df = t ? (RHO_EQ * U_LID / f6) : f0;
            FLT in[9], o[9];
            in[0] = AT(f, 0, y, x);
            in[1] = l ? AT(f, 3, y, x) : AT(f, 1, y, x - 1);
            in[3] = r ? AT(f, 1, y, x) : AT(f, 3, y, x + 1);
            in[2] = b ? AT(f, 4, y, x) : AT(f, 2, y - 1, x);
            in[4] = t ? AT(f, 2, y, x) : AT(f, 4, y + 1, x);
            in[5] = (l || b) ? AT(f, 7, y, x) : AT(f, 5, y - 1, x - 1);
            in[6] = (r || b) ? AT(f, 8, y, x) : AT(f, 6, y - 1, x + 1);
            in[7] = (r || t) ? (AT(f, 5, y, x) - df) : AT(f, 7, y + 1, x + 1);
            in[8] = (l || t) ? (AT(f, 6, y, x) + df) : AT(f, 8, y + 1, x - 1);
            FN(collide_lid_sel)(in, o, OMEGA);
            for (unsigned q = 0; q < 9; ++q) AT(buf, q, y, x) = o[q];
        }
    }
}

/* compute_velocities: io.cpp:136-174.  Sums first, one division, then the two products.
 * Output here is planar [2][ny][nx] (ux plane, uy plane) over rows [y_lo,y_hi) of the source
 * (the reference stores vel(x,y,c), an indexing choice that carries no arithmetic). */
void FN(orc_velocities)(const FLT *f, FLT *vel, uint64_t nx, uint64_t ny, uint64_t y_lo,
                        uint64_t y_hi)
{
    const uint64_t nrows = y_hi - y_lo;
#pragma omp parallel for schedule(static)
    for (uint64_t y = y_lo; y < y_hi; ++y)
        for (uint64_t x = 0; x < nx; ++x) {
            const FLT f_0 = AT(f, 0, y, x), f_1 = AT(f, 1, y, x), f_2 = AT(f, 2, y, x);
            const FLT f_3 = AT(f, 3, y, x), f_4 = AT(f, 4, y, x), f_5 = AT(f, 5, y, x);
            const FLT f_6 = AT(f, 6, y, x), f_7 = AT(f, 7, y, x), f_8 = AT(f, 8, y, x);
            const FLT ux = f_1 - f_3 + f_5 - f_6 - f_7 + f_8;
            const FLT uy = f_2 - f_4 + f_5 + f_6 - f_7 - f_8;
            const FLT one = (FLT)1.0;
            const FLT rho_inv = one / (f_0 + f_1 + f_2 + f_3 + f_4 + f_5 + f_6 + f_7 + f_8);
            vel[(0 * nrows + (y - y_lo)) * nx + x] = ux * rho_inv;
            vel[(1 * nrows + (y - y_lo)) * nx + x] = uy * rho_inv;
        }
}

/* Max velocity magnitude: io.cpp:184-208.  SQRT is sqrtf for both precisions (global.h:30): the
 * FLT sum of squares is narrowed to float, the float root widened back to FLT.  NaNs never win
 * the comparison (:205); the start value is -1e30 (:184). */
FLT FN(orc_max_velocity)(const FLT *f, uint64_t nx, uint64_t ny, uint64_t y_lo, uint64_t y_hi)
{
    FLT max_vel_mag = (FLT)-1e30;
#pragma omp parallel
    {
        FLT local_max = (FLT)-1e30;
#pragma omp for schedule(static) nowait
        for (uint64_t y = y_lo; y < y_hi; ++y)
            for (uint64_t x = 0; x < nx; ++x) {
                const FLT f_0 = AT(f, 0, y, x), f_1 = AT(f, 1, y, x), f_2 = AT(f, 2, y, x);
                const FLT f_3 = AT(f, 3, y, x), f_4 = AT(f, 4, y, x), f_5 = AT(f, 5, y, x);
                const FLT f_6 = AT(f, 6, y, x), f_7 = AT(f, 7, y, x), f_8 = AT(f, 8, y, x);
                const FLT one = (FLT)1.0;
                const FLT rho_inv =
                    one / (f_0 + f_1 + f_2 + f_3 + f_4 + f_5 + f_6 + f_7 + f_8);
                const FLT ux = (f_1 - f_3 + f_5 - f_6 - f_7 + f_8) * rho_inv;
                const FLT uy = (f_2 - f_4 + f_5 + f_6 - f_7 - f_8) * rho_inv;
                const FLT u = sqrtf((float)(ux * ux + uy * uy));
                if (u > local_max) local_max = u;
            }
#pragma omp critical
        {
            if (local_max > max_vel_mag) max_vel_mag = local_max;
        }
    }
    return max_vel_mag;
}

/* |u| field as the -ov output forms it: io.cpp:236-238 (sqrtf of the FLT sum of squares of the
 * compute_velocities results).  Planar [ny_rows][nx]. */
void FN(orc_velocity_magnitudes)(const FLT *f, FLT *mag, uint64_t nx, uint64_t ny, uint64_t y_lo,
                                 uint64_t y_hi)
{
#pragma omp parallel for schedule(static)
    for (uint64_t y = y_lo; y < y_hi; ++y)
        for (uint64_t x = 0; x < nx; ++x) {
            const FLT f_0 = AT(f, 0, y, x), f_1 = AT(f, 1, y, x), f_2 = AT(f, 2, y, x);
            const FLT f_3 = AT(f, 3, y, x), f_4 = AT(f, 4, y, x), f_5 = AT(f, 5, y, x);
            const FLT f_6 = AT(f, 6, y, x), f_7 = AT(f, 7, y, x), f_8 = AT(f, 8, y, x);
            const FLT sx = f_1 - f_3 + f_5 - f_6 - f_7 + f_8;
            const FLT sy = f_2 - f_4 + f_5 + f_6 - f_7 - f_8;
            const FLT one = (FLT)1.0;
            const FLT rho_inv = one / (f_0 + f_1 + f_2 + f_3 + f_4 + f_5 + f_6 + f_7 + f_8);
            const FLT ux = sx * rho_inv, uy = sy * rho_inv;
            mag[(y - y_lo) * nx + x] = sqrtf((float)(ux * ux + uy * uy));
        }
}

#undef AT
#undef FN
#undef CAT
#undef CAT_
