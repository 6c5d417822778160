/* femo_hip_test.h -- entry points of libfemo_hip.so that exist for the test suite and the scaling model only.
 * They are exported by the same library but are NOT part of the product ABI (include/femo_hip.h): no
 * production caller needs them, and a reference-side binding never touches them.          */
#ifndef FEMO_HIP_TEST_H
#define FEMO_HIP_TEST_H

#include "femo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct femo_emu_group femo_emu_group;  /* in-process rank emulation */

/* ---- rank emulation on ONE GPU (tests) -----------------------------------------------------------
 * RCCL cannot place two ranks on one device.  With a femo_emu_group, `nranks` contexts on the same GPU,
 * each driven by its own host thread, behave as the ranks of a job: every collective of the library
 * (halo exchange, all-reduced scalars, all-reduced lattice accumulators) is staged through host memory
 * and a barrier instead of RCCL, so the multi-rank code paths run for real on a one-GPU box.  A rank
 * that never reaches a collective makes the others fail after 60 s instead of hanging.             */
int femo_emu_group_create(int nranks, femo_emu_group** out);
int femo_emu_group_destroy(femo_emu_group* group);
int femo_comm_emulate(femo_ctx* ctx, femo_emu_group* group, int rank);

/* ---- "model" communicator (bench.py's scaling_model) ------------------------------------------------
 * Makes ONE context run the N-rank code paths of the library alone on its GPU: ctx behaves as rank `rank` of
 * `nranks` (partitioned-mesh branches of the solvers, the pack kernel of the merged BPX-PCG loop, the split
 * interior / boundary SpMV with its halo pack), while every collective completes at once without moving a byte
 * -- an all-reduce leaves the rank's own contribution in place, a neighbour exchange fills the ghost entries with
 * zeros (one memset on the stream).  The collectives are still counted (femo_comm_stats).  What this measures:
 * everything a rank of an N-GPU job does per iteration except the time on the wire.  The numbers it computes
 * are those of the rank's block solved on its own (zero ghost values), not of the global problem.           */
int femo_comm_model(femo_ctx* ctx, int rank, int nranks);

#ifdef __cplusplus
}
#endif
#endif
