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
// feedback_rows: tile rows of the launch if it was a persistent one whose rows should be re-ordered for the next frame (else 0)
hipError_t launch_reduce_totals(const RenderTotals *partials, uint32_t n_blocks, RenderTotals *totals, hipStream_t stream, uint32_t *tile_cursors,
	uint32_t feedback_rows, unsigned long long frame_pixels, uint32_t feedback_key);
// RowMap::feedback_key of a launch: scene (index, or a hash of a run-time scene's name), frame width and what the row map selects,
// hashed into the upper 22 bits; the low 10 bits ARE the number of units (tile rows or squares, <= SDFR_ROW_FEEDBACK_MAX = 512) the
// order was made for: two launches with equal keys have equally long orders whatever the hash does
uint32_t pixel_feedback_key(uint32_t scene_key, int width, const RowMap &rm, uint32_t feedback_rows);
int pixel_tile_cursor_words();
int scene_tile_w_log2(int scene); // the tile shape a built-in scene asks for (SceneTileShape); 3 = 8 x 8, also for run-time scenes
// how the pixel kernels are launched: persistent (resident waves pull tiles from the cursors) or one wave per
// tile.  launch_mode: 0 = the scene's own default (PersistentTiles), 1 = one wave per tile, 2 = persistent;
// the developer knobs SDFR_PIXEL_PERSISTENT=0|1 and SDFR_PIXEL_BLOCKS_PER_CU=n (cap of a persistent grid) override.
// retire_after (persistent launches; SDFR_PIXEL_RETIRE_AFTER=n overrides, 0 = never): see pixel_launch_blocks.
struct PixelLaunchMode { bool persistent; int blocks_per_cu; int retire_after; };
PixelLaunchMode pixel_launch_mode(int launch_mode, bool scene_default_persistent, int scene_retire_after = 8);
// Blocks of a pixel launch.  One wave per tile: as many as tiles.  Persistent: what stays resident -- and, when waves
// retire after `retire_after` tiles, the replacements as well: tiles / retire_after, plus half a chip of waves that
// end for want of tiles before they have had their share.  Why waves retire: a SIMD serves its oldest waves first, so
// of the waves that start together the ones in its upper slots crawl for the whole frame (tools/wave_trace.py: two
// tiles against sixty), and what they hold when the queue runs dry is finished by one or two waves per SIMD while the
// rest of the chip idles -- the last 7 % of a labyrinth frame, a fifth of a fractal frame.  A wave that leaves after 8
// tiles is replaced by a younger one, the crawlers become the oldest and catch up.  Measured (ms per frame, one frame
// in flight; never / 4 / 8 / 16): labyrinth 4K 1.375 / 1.370 / 1.359 / 1.366, cube_sea 1080p 0.883 / 0.844 / 0.840 /
// 0.877, fractal 4K 1.607 / 1.470 / 1.485 / 1.549 (one wave per tile: 1.397, -, 1.482).
uint32_t pixel_launch_blocks(const PixelLaunchMode &mode, uint32_t tiles, uint32_t resident_blocks);

int pixel_block_threads(); // block size of the pixel kernels (partials are sized by it)
int device_cu_count(int device);
// work items (padded to whole tiles) of a launch: lists and per-pixel state are sized by this
uint32_t launch_work_items(int width, const RowMap &rm);
uint32_t launch_capacity_items(int width, const RowMap &rm); // >= launch_work_items: what the workspace of a launch is sized for

} // namespace sdfr
