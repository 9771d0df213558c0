/* rtx_hip_debug.h — test and tuning hooks of librtx_hip.so.  NOT part of the drop-in ABI (include/rtx_hip.h):
 * nothing here has a counterpart in the reference, a caller of the path never needs it, and the rendered image
 * does not depend on any of it (tests/test_gpu_parity.py::test_work_unit_size_does_not_change_the_image).
 * The library reads no environment variables. */
#ifndef RTX_HIP_DEBUG_H
#define RTX_HIP_DEBUG_H

#include "rtx_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Force the work-unit schedule of the render kernel: units of `unit_a` samples per pixel, except for the last
 * `tail_span` samples of a launch, which are cut into `unit_b`-sample units.  0 / 0 / -1 restore the defaults. */
int rt_debug_set_work_units(rt_ctx* ctx, uint32_t unit_a, uint32_t unit_b, int32_t tail_span);

/* world_order != 0: the next rt_upload_scene keeps the scan lists in the caller's order instead of sorting them
 * along a Morton curve (A/B switch for the lazy root extraction; results are identical). */
int rt_debug_set_scan_order(rt_ctx* ctx, int world_order);

/* on = 0: the next rt_upload_scene keeps every static small sphere in the general scan list instead of moving those whose
 * centres share one coordinate into the planar list (cheaper tests); on != 0 restores the default.  Results are
 * identical. */
int rt_debug_set_planar_list(rt_ctx* ctx, int on);

/* on = 0: the next rt_upload_scene keeps the scan lists even for a scene of a handful of spheres, which by default sends
 * all of its spheres to the per-ray f64 list and runs the render kernel variant without a scan (all three scan-slot
 * counts of rt_debug_scan_layout are then 0); on != 0 restores the default.  Results are identical up to the fp32
 * rounding of the hit record's inputs (both decide hits in f64). */
int rt_debug_set_direct(rt_ctx* ctx, int on);

/* Effective only in experiment builds (-DRT_X_TAILORDER, rt_lib.hip: build_tail_order), no-ops otherwise.
 * on = 0: renders deal the small work units at the end of a launch in tile-index order, as the product build does,
 * instead of dealing the tiles that bred long paths (32 segments or more) earlier in the launch first, most of them
 * first; on != 0 restores the experiment's order.  Scheduling only: results are identical bit for bit. */
int rt_debug_set_tail_order(rt_ctx* ctx, int on);

/* After a render through an allocating entry point: out = {1 if the last launch built a tail order else 0, the number of
 * tiles on which a long path had been counted by then (they are dealt before the others, most counts first), 1 if the
 * order read back from the device is a permutation of the tiles with ascending class starts (checked on the host)}.
 * All zeros in the product build. */
int rt_debug_tail_order_stats(rt_ctx* ctx, uint32_t out[3]);

/* n = 1 or 2: the next rt_upload_scene gives every wave that many tile buffers in LDS (2: the in-flight paths of a
 * finished work unit keep adding into their tile's LDS sums; 1: they add straight into the global accumulators) when
 * that many fit; 0 restores the choice by LDS room and occupancy.  Results are identical. */
int rt_debug_set_tile_buffers(rt_ctx* ctx, int n);

/* How the last rt_upload_scene laid the world out for the scan, for tests and for bench.py's kernel name:
 * out = {big spheres, general static slots, planar slots, moving slots (each padded to whole chunks of 16),
 *        planar axis (0/1/2, -1 without a planar list), bit mask of the axes some moving sphere moves along (bit 0 =
 *        x; 2 = y only: the instantiation without the x and z interpolation; 0 without moving spheres),
 *        tile buffers per wave, noise variant (0/1)}. */
int rt_debug_scan_layout(rt_ctx* ctx, int32_t out[8]);

/* Process-wide switches read by rt_multi_create, so that EVERY line of the multi-device path can run on a box with one
 * GPU.  flags: bit 0 = a device ordinal may be listed more than once (each entry gets its own rt_ctx, stream and
 * workspace on that GPU); bit 1 = never use RCCL.  Either way the exchange step then copies each part's sums to the
 * root device (hipMemcpyPeerAsync) and adds them there instead of one ncclReduce -- the fallback rt_multi_create also
 * takes by itself when librccl is missing or ncclCommInitAll fails.  0 restores the default.  No scaling number may be
 * quoted from such a run. */
int rt_debug_multi(int flags);

#ifdef __cplusplus
}
#endif
#endif /* RTX_HIP_DEBUG_H */
