/* gsplat_dist.h — the ONE exchange step of the multi-GPU path as a C ABI (libgsplat_dist.so).
 *
 * SURVEY.md §8e / DESIGN.md §7: one camera per rank, replicated Gaussians, a sum all-reduce of the
 * flat gradient buffer [v_features_rest | v_features_dc | v_means | v_scales | v_quats | v_opacity]
 * after backward — or its factored form, gs_dist_allgather below.  OpenSplat is a C++ program without
 * any distributed code; a Model-level caller gets the exchange through these functions — RCCL over xGMI underneath,
 * enqueued on the caller's HIP stream (the one the operators of gsplat_hip.h were given), no torch
 * type in sight.  The Python harness (opensplat_amd/dist.py) uses torch.distributed instead, whose
 * "nccl" backend is the same RCCL.
 *
 * Bootstrap: rank 0 calls gs_dist_unique_id and ships the 128 bytes to the other ranks by whatever
 * channel the application has (a file, a socket, MPI, a torch store); every rank then calls
 * gs_dist_init with the same bytes.  One communicator per process and GPU.
 *
 * Return values: GS_OK or a negative GsStatus (gsplat_hip.h); GS_ERR_HIP also covers RCCL errors,
 * gs_dist_last_error() has the text.  Never throws, never exits.
 */
#ifndef GSPLAT_DIST_H
#define GSPLAT_DIST_H

#include "gsplat_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GS_DIST_ID_BYTES 128

typedef struct GsDistComm GsDistComm; /* opaque */

int gs_dist_unique_id(uint8_t id[GS_DIST_ID_BYTES]);

/* device: HIP device ordinal this rank renders on (made current for the call). */
int gs_dist_init(GsDistComm **comm, int world_size, int rank, const uint8_t id[GS_DIST_ID_BYTES],
                 int device);

/* In-place sum over all ranks of buf[0 .. count), fp32, enqueued on `stream`; returns at once.
 * world_size 1: a no-op — unless the environment variable GSPLAT_DIST_FORCE_COLLECTIVES was set to a
 * non-zero value when the communicator was created: RCCL's collective is then enqueued anyway (test
 * aid for one-GPU boxes; the same holds for gs_dist_allgather). */
int gs_dist_allreduce_sum(GsDistComm *comm, float *buf, size_t count, gs_stream_t stream);

/* The same as n_buckets collectives over consecutive slices (boundaries on multiples of 1024
 * elements): a consumer that synchronises per slice — the Adam step of those parameters — overlaps
 * with the transfer of the next one.  done_events: optional array of n_buckets hipEvent_t (as
 * void*), event k recorded on `stream` behind bucket k. */
int gs_dist_allreduce_sum_buckets(GsDistComm *comm, float *buf, size_t count, int n_buckets,
                                  void **done_events, gs_stream_t stream);

/* All-gather: every rank contributes send[0 .. count) and receives the world_size messages in rank
 * order in recv[0 .. world_size * count); send may be the rank's own slot of recv (in place).
 * The factored exchange moves the colour cotangents this way — 12 B per Gaussian and
 * camera instead of all-reducing the 12 K bytes of SH gradients they imply:
 *     gs_gaussian_backward(..., GS_FLAG_EMIT_VCOLOR)          // v_colour into the rank's message
 *     gs_dist_allreduce_sum(comm, geometry block, 11 N, s);   // means, scales, quats, opacity
 *     gs_dist_allgather(comm, message, gathered, count, s);   // [camera centre 4 | v_colour 3 N]
 *     gs_sh_backward_cameras(N, K, deg, world, means, gathered, count, gathered + 4, count, ...)
 * world_size 1: a device copy (nothing if send == recv). */
int gs_dist_allgather(GsDistComm *comm, const float *send, float *recv, size_t count,
                      gs_stream_t stream);

int gs_dist_world_size(const GsDistComm *comm);
int gs_dist_rank(const GsDistComm *comm);
int gs_dist_destroy(GsDistComm *comm);
const char *gs_dist_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_DIST_H */
