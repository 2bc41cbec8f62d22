// scatter_small.hip -- the scatter kernel with two 512-thread workgroups per CU (6144-slot tables).
// Same source as scatter.hip, other build-time constants; see the note at the top of scatter.hip.
// (The constants can be overridden on the command line for experiments; abi.hip's
// ATTPC_SC_SMALL_WGS must then match ATTPC_SC_WG_PER_CU.)
#define ATTPC_SC_VARIANT small
#ifndef ATTPC_SC_THREADS
#define ATTPC_SC_THREADS 512
#define ATTPC_SC_HASH_BITS 12
#define ATTPC_SC_STAGE 102  // two whole row passes (2 x 512 rows = 16 blocks of 64: two per wave)
#define ATTPC_SC_WG_PER_CU 2
#endif
#include "scatter.hip"
