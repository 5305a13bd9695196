// sdfr_kernels.h -- host-callable launchers of the gfx950 kernels (sdfr_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sdfr_frame.h"

namespace sdfr {

// per-handle scratch of the wavefront schedule (allocated by the API, sized for `capacity` pixels)
struct WavefrontWorkspace
{
	size_t capacity;     // pixels
	float *ray_cur;      // [11][capacity] ray being traced for each pixel
	float *ray_queue;    // [8][capacity] pending rays, 48-byte records (GlobalRayStore)
	uint32_t *qdepth_lo; // [capacity] packed depths of slots 0-3
	uint32_t *qdepth_hi; // [capacity] packed depths of slots 4-7
	float *result;       // [8][capacity] march result: status/iter, t, d, normal xyz (+2 spare)
	float *accum;        // [4][capacity] rgb accumulator + hdr flag
	uint32_t *pstat;     // [3][capacity] per-pixel counters
	uint32_t *list_a;    // [capacity] pixels with a ray in flight (ping)
	uint32_t *list_b;    // [capacity] (pong)
	uint32_t *counters;  // [64] list sizes per round and misc
	RenderTotals *partials; // [capacity / 64 + 1] per-block counter sums of the pixel schedule
	uint32_t *tile_cursors; // [8 x 32] tile hand-out counters of the pixel kernel (TileQueue), zero between launches
};

hipError_t launch_pixel_schedule(int scene, const FrameU &U, const RowMap &rows, void *out, int format, uint32_t *pixel_stats,
	RenderTotals *totals, const WavefrontWorkspace &ws, hipStream_t stream, int launch_mode = 0);

hipError_t launch_wavefront_schedule(int scene, const FrameU &U, const RowMap &rows, void *out, int format, uint32_t *pixel_stats,
	RenderTotals *totals, const WavefrontWorkspace &ws, hipStream_t stream, hipEvent_t *march_events, hipEvent_t *shade_events,
	int *n_rounds_out);

hipError_t launch_assemble_strips(int width, int height, int world, const void *gathered, void *out_image, int format, int priv_count,
	int priv_period, hipStream_t stream);

// HDR::process: scene16/bloom1 RGBA16F, ldr8 RGBA8, all device pointers of width*height pixels
// mid_event (optional) is recorded between the two kernels
// flags: postprocess_flag_bytes(width, height) bytes of device scratch (one per 32 pixels of a row: lit or not)
size_t postprocess_flag_bytes(int width, int height);
hipError_t launch_postprocess(int width, int height, const void *scene16, void *bloom1, void *ldr8, unsigned char *flags, hipStream_t stream,
	hipEvent_t mid_event = nullptr);

hipError_t launch_selftest_math(int what, float c, unsigned long long *d_mismatches, hipStream_t stream);

// wavefront schedule, scene-independent start of a frame: primary rays, empty queues, round-0 list
hipError_t launch_wavefront_init(const FrameU &U, const RowMap &rm, uint32_t n_work, const WavefrontWorkspace &ws, uint32_t *pixel_stats, hipStream_t stream);

// folds the per-block partial sums of a pixel-schedule launch into `totals` and puts the tile cursors back to zero
hipError_t launch_reduce_totals(const RenderTotals *partials, uint32_t n_blocks, RenderTotals *totals, hipStream_t stream, uint32_t *tile_cursors);
int pixel_tile_cursor_words();
// how the pixel kernels are launched: persistent (resident waves pull tiles from the cursors) or one wave per
// tile.  launch_mode: 0 = the scene's own default (PersistentTiles), 1 = one wave per tile, 2 = persistent;
// the developer knobs SDFR_PIXEL_PERSISTENT=0|1 and SDFR_PIXEL_BLOCKS_PER_CU=n (cap of a persistent grid) override.
struct PixelLaunchMode { bool persistent; int blocks_per_cu; };
PixelLaunchMode pixel_launch_mode(int launch_mode, bool scene_default_persistent);

int pixel_block_threads(); // block size of the pixel kernels (partials are sized by it)
int device_cu_count(int device);
// work items (padded to whole tiles) of a launch: lists and per-pixel state are sized by this
uint32_t launch_work_items(int width, const RowMap &rm);

} // namespace sdfr
