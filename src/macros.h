#ifndef MACROS_H
#define MACROS_H

// Build-time switches of the reference (its scripts rewrite this file before each build).  Each one
// can also be given on the compiler command line.
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

#endif
