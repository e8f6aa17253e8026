/*
 * calibrating_amd_experimental.h -- entry points of libcalibrating_amd.so that have NO counterpart in the reference's
 * interface and are on no default path: the measurement hooks of the overlap / partitioning experiments (DESIGN.md
 * section 4, profiles/r05_cumask*.json, profiles/r06_pipeline.json, profiles/r06_corun.json -- measured at no gain).
 * Nothing in calibrating_amd/ (the product) calls them; tools/history/gpu_cumask*.py, tools/gpu_r6_*.py and one parity test
 * do.  They may change or go away between rounds.
 */
#ifndef CALIBRATING_AMD_EXPERIMENTAL_H
#define CALIBRATING_AMD_EXPERIMENTAL_H

#include "calibrating_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A HIP stream restricted to a subset of the compute units (hipExtStreamCreateWithCUMask): bit i of cu_mask[i / 32]
 * enables CU i in the driver's enumeration, which walks the XCDs first -- bit 0 = XCD 0's first CU, bit 1 = XCD 1's,
 * ... -- so the first N bits are N / 8 CUs of every XCD (measured: tools/history/gpu_cumask_probe.py).  An XCD whose share of
 * the mask is empty is left unrestricted, not disabled, and workgroups are dealt to the XCDs in equal shares: give
 * every XCD the same number of CUs, a multiple of its four shader engines (N a multiple of 32).  For spatial partitioning of concurrent work (one partition
 * for a VALU-bound kernel, the rest for an HBM-bound one).  The reference has no counterpart (single-threaded host
 * code); nothing on the default path creates such a stream.  *stream is a hipStream_t. */
int camd_stream_create_cu_mask(const uint32_t* cu_mask, int nwords, void** stream);
int camd_stream_destroy(void* stream);

/* CAMD_OPT_PHASES (camd_sgbm_set_option): which part of the work a camd_sgbm_compute call queues, a bit mask --
 *   1 = the matching-cost volume,  2 = the first aggregation pass,  4 = the last pass, winner-take-all and post filters
 * (bits 2 and 4 are separable on the band path only), 7 (default) = everything.  For callers that run the kernels of
 * consecutive batches side by side on two streams.
 * A handle owns ONE cost volume, ONE aggregated volume and one set of flags: a phase-1 call for the next batch on the
 * same handle races with the later phases of the previous batch.  Alternate between two handles and make every
 * phase-1 call wait (an event) for that handle's previous phase-4 call; ordering the calls of one batch (events
 * between the streams) is the caller's business too.  A call without bit 4 leaves `disp` untouched. */
enum { CAMD_OPT_PHASES = 6 };
/* CAMD_OPT_RESIDENT (camd_sgbm_set_option): value = 16 * a + b -- the cost kernel runs as `a` and the row-parallel last
 * aggregation pass (MODE_SGBM, band path) as `b` persistent workgroups per compute unit that take their work by ticket,
 * instead of one workgroup per work item (0 = the ordinary launch; a, b <= 4).  Neither launch then fills the chip, so
 * two of them on two streams are resident side by side for their whole duration -- ordinary launches on two streams
 * overlap only in their tails (profiles/r06_corun.json).  Results are bit-identical. */
enum { CAMD_OPT_RESIDENT = 7 };

#ifdef __cplusplus
}
#endif
#endif
