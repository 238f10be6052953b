#ifndef MACROS_H
#define MACROS_H

#define USE_SINGLE_PRECISION true
#define UNROLL_LOOPS true
#define COALESCING true
#define USE_MPI false

#endif
