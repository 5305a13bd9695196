/* sdfr.h -- C ABI of the MI355X-native SDF raymarch renderer (libsdfr.so).
 *
 * This is the drop-in boundary for ONE hot path of Gotbread/sdf-playground: the SDF render
 * stage `class SDFRenderer` (Engine/SDFRenderer.h:17-44), i.e. the per-pixel raymarch loop
 * of Engine/shader/pshader_sdf.hlsl.  Each entry point names the reference interface it
 * replaces.  Plain C types only; no torch, no C++ types.
 *
 * Conventions
 *   - every call returns an sdfr_status (0 = ok, < 0 = error); sdfr_last_error() gives text.
 *     The reference returns bool and pops a MessageBox (Util.cpp:61-69); nothing here aborts.
 *   - a handle is bound to one GPU and one HIP stream; calls on one handle are not
 *     thread-safe (the reference is single-threaded, Application.cpp:65-95).
 *   - sdfr_render* enqueue work on the handle's stream and return; sdfr_sync waits.
 *   - images are row-major, row 0 = top row, 4 channels interleaved; alpha is the
 *     tone-mapping flag in {0, 1} (pshader_sdf.hlsl:636-638), not coverage.
 */
#ifndef SDFR_H
#define SDFR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdfr_renderer sdfr_renderer;

typedef enum sdfr_status
{
	SDFR_OK = 0,
	SDFR_ERR_INVALID_ARGUMENT = -1,
	SDFR_ERR_UNKNOWN_SCENE = -2,
	SDFR_ERR_UNKNOWN_VARIABLE = -3, /* value ignored, like ShaderVariableManager::setValue (ShaderUtil.cpp:234-240) */
	SDFR_ERR_NO_SCENE = -4,         /* render without a loaded scene: SDFRenderer::render returns false (SDFRenderer.cpp:70-73) */
	SDFR_ERR_HIP = -5,
	SDFR_ERR_NO_DEVICE = -6,
	SDFR_ERR_COMPILE = -7, /* a run-time scene does not compile: sdfr_last_error holds the compiler's messages */
	SDFR_ERR_COMM = -8,    /* RCCL could not be loaded, or one of its calls failed: sdfr_comm_last_error / sdfr_last_error */
	SDFR_ERR_INTERNAL = -9 /* a C++ exception (out of memory, ...) was caught at the C boundary: sdfr_last_error has its words; never an abort */
} sdfr_status;

/* ---- lifetime: SDFRenderer::init(Graphics&) (SDFRenderer.cpp:9-25) ----------------------- */
int sdfr_create(int device_ordinal, sdfr_renderer **out);
void sdfr_destroy(sdfr_renderer *r);
const char *sdfr_last_error(const sdfr_renderer *r);
/* work is enqueued on this hipStream_t (NULL = the default stream) */
int sdfr_set_stream(sdfr_renderer *r, void *hip_stream);

/* ---- scene selection: SceneManager list (SceneManager.cpp:142-158) + Application::loadScene
 *      (Application.cpp:318-322) + SDFRenderer::initShader (SDFRenderer.cpp:27-53).
 *      Names are the reference's file stems without "sdf_scene_". ------------------------------ */
int sdfr_scene_count(void);
const char *sdfr_scene_name(int index);
/* besides the listed names: "debug_materials", the library's own diagnostic scene (four objects wearing
 * the driver's MATERIAL_ITER / PLAIN / NORMAL1 / NORMAL2 views, pshader_sdf.hlsl:430-455, which no
 * reference scene emits; not part of the reference's scene list, hence not counted above), and "normal_test",
 * a scene with a map_normal callback (sdf_structs.hlsl:39-52: its own normal on one object, a wider
 * normal_sample_dist -- rounded corners -- on two others), which every scene of the reference leaves empty */
int sdfr_load_scene(sdfr_renderer *r, const char *name);
const char *sdfr_current_scene(const sdfr_renderer *r);
/* Compile a scene from source text at run time -- the reference's edit-and-reload workflow
 * (SceneManager.cpp:102-133 re-runs SDFRenderer::initShader -> D3DCompile when a scene file
 * changes).  `source` is HIP C++ defining `struct Scene` with the scene interface of
 * sdf_playground_amd/scenes/README.md (the reference's map / map_light / map_background split
 * into dist / material / light / background); VAR_<name>(min = .., max = .., ..) tags in the text
 * declare variables exactly as in the reference's .hlsl scenes (ShaderUtil.cpp:122-191).  On
 * SDFR_ERR_COMPILE the previously loaded scene stays active, as in the reference
 * (SceneManager.cpp:118-127).  Needs libhiprtc.so and the kernel headers (csrc/ beside the
 * library, or $SDFR_JIT_INCLUDE). */
int sdfr_load_scene_source(sdfr_renderer *r, const char *name, const char *source);
/* Same compilation without a device or a handle (build machines, editors): SDFR_OK, or
 * SDFR_ERR_COMPILE / SDFR_ERR_INVALID_ARGUMENT with the messages in `log`.  arch: "gfx950" (NULL = that). */
int sdfr_check_scene_source(const char *source, const char *arch, char *log, size_t log_bytes);

/* The same for a scene IN THE REFERENCE'S OWN DIALECT: the text of an .hlsl scene file as Application::loadScene substitutes
 * it for "sdf_scene.hlsl" (Application.cpp:229,320; pshader_sdf.hlsl:84) -- the four callbacks map / map_normal / map_light /
 * map_background with their HLSL signatures (sdf_structs.hlsl), the OBJECT / OBJECT_TRANSPARENT / MATERIAL macros
 * (pshader_sdf.hlsl:79-81), float2/3/4 with swizzles, the intrinsics, the shader libraries under their own names
 * (sdSphere ... turbulence), the frame globals (stime, eye, ...), VAR_ tags.  The 22 scene files of the reference load as
 * they are (their `#include "sdf_*.hlsl"` lines are dropped: the libraries are this library's).  The text is compiled as
 * the body of a C++ class after a short textual pass (csrc/sdfr_hlsl.h, sdfr_hlsl.cpp); plain IEEE arithmetic, like every
 * run-time scene.  Not supported: snoise(float2) / snoise(float4), HLSL objects that have no meaning here (textures,
 * samplers, semantics).  sdfr_translate_scene_hlsl returns the generated C++ (bytes needed incl. the terminator; `out` may
 * be NULL) -- for inspection and for compiling a scene with a host compiler. */
int sdfr_load_scene_hlsl(sdfr_renderer *r, const char *name, const char *hlsl_source);
int sdfr_check_scene_hlsl(const char *hlsl_source, const char *arch, char *log, size_t log_bytes);
int sdfr_translate_scene_hlsl(const char *hlsl_source, char *out, size_t out_bytes);

/* ---- parameter surface 1: shader variables = SDFRenderer::getVariableMap()
 *      (ShaderVariable.h:6-12; ShaderUtil.cpp:122-267).  Index order = std::map order
 *      (lexicographic by name), which is also the reference's constant-buffer order. --------- */
typedef struct sdfr_variable
{
	char name[48];
	float minval, maxval, start, step;
	float value;
} sdfr_variable;
int sdfr_var_count(const sdfr_renderer *r);
int sdfr_var_info(const sdfr_renderer *r, int index, sdfr_variable *out);
int sdfr_var_set(sdfr_renderer *r, const char *name, float value);
int sdfr_var_get(const sdfr_renderer *r, const char *name, float *out);
int sdfr_vars_reset(sdfr_renderer *r); /* VariableManager::resetVariables: value = start */

/* ---- parameter surface 2: camera + time = the camera constant buffer b0
 *      (SDFRenderer.h:29-34, SDFRenderer.cpp:85-95) and SDFRenderer::setParameters(stime). ---- */
int sdfr_set_camera(sdfr_renderer *r, const float eye[3], const float front[3], const float right[3], const float top[3]);
/* FPS-mode Camera of the reference (Camera.cpp:24-49,156-166): basis from eye -> lookat */
int sdfr_set_camera_lookat(sdfr_renderer *r, const float eye[3], const float lookat[3], float fovy, float aspect, float roll);
int sdfr_set_camera_direction(sdfr_renderer *r, const float eye[3], const float direction[3], float fovy, float aspect, float roll);
int sdfr_get_camera(const sdfr_renderer *r, float out_eye_front_right_top[12]);
int sdfr_set_time(sdfr_renderer *r, float stime);

/* ---- parameter surface 3: the driver's compile-time limits (pshader_sdf.hlsl:60-64,350)
 *      as run-time values.  Defaults = reference; anything else is a labelled extension. ------ */
typedef struct sdfr_limits
{
	int iter_count;       /* ITER_COUNT   100, >= 1 */
	int bounce_count;     /* BOUNCE_COUNT 16, 0..16 */
	int ray_count;        /* RAY_COUNT    8, 1..8 */
	int light_count;      /* LIGHT_COUNT  8, 0..8 */
	float range;          /* RANGE        100 */
	int max_cost_default; /* MaterialOutput.max_cost 7, 0..250 */
	/* EXTENSION, 0 = reference: n = 1..7 point lights orbiting at radius 5, height 3 overwrite slots
	 * 1..n of the scene's map_light table (the "8 lights" variant of the lense/gems benchmark
	 * configuration; the reference's scenes carry one light).  DESIGN.md section 2. */
	int extension_lights;
	/* EXTENSION, 0 = reference: reflection_color given to the labyrinth's marble (MATERIAL_MARBLE_DARK / _LIGHT).
	 * The reference's labyrinth has no reflective material; BASELINE configs[2] is worded "2 reflection bounces":
	 * with 0.25 and the default cost rule (a reflection costs 3 of max_cost 7) a ray is reflected at most twice. */
	float extension_marble_reflection;
	/* the driver's five epsilons, `static const float` in the reference (pshader_sdf.hlsl:31-35), as run-time values.
	 * Defaults = the reference's; anything else is a labelled EXTENSION.
	 *   dist_eps    1e-4  "how close to the object before terminating": the march's hit test (:211), the MATERIAL and
	 *                     OBJECT_TRANSPARENT macros (:80-81), sdSphereFast (sdf_primitives.hlsl:31), the directional
	 *                     light's normalisation (:535); 0 < dist_eps <= 1e-3 (the built-in scenes' culling bounds and
	 *                     escape rules are proved with 0.01 of slack)
	 *   grad_eps    1e-4  spacing of the forward-difference normal samples (:323) -- unless the scene's map_normal sets
	 *                     its own normal_sample_dist; > 0
	 *   reflect_eps 1e-3, refract_eps 1e-3  how far a reflected / refracted ray starts along its direction (:373,397,411); >= 0
	 *   shadow_eps  3e-4  shadow rays start max(shadow_eps, normal_sample_dist) off the surface (:520); >= 0 */
	float dist_eps, grad_eps, reflect_eps, refract_eps, shadow_eps;
} sdfr_limits;
int sdfr_get_limits(const sdfr_renderer *r, sdfr_limits *out);
int sdfr_set_limits(sdfr_renderer *r, const sdfr_limits *limits);

/* how the pipeline stages are scheduled on the GPU (results are identical).  Default: PIXEL,
 * the faster one on every scene measured on MI355X.  WAVEFRONT is kept as a second,
 * differently scheduled implementation for cross-checking (2.7x slower on the headline workload); nothing
 * selects it by itself. */
typedef enum sdfr_schedule
{
	SDFR_SCHEDULE_WAVEFRONT = 0, /* rays in HBM, persistent march waves refilled by ballot, separate shade kernel */
	SDFR_SCHEDULE_PIXEL = 1      /* one lane per pixel, start to finish; pending rays in HBM behind a register cache */
} sdfr_schedule;
int sdfr_set_schedule(sdfr_renderer *r, int schedule);
/* how the PIXEL schedule's kernel is launched (results are identical):
 *   PER_TILE     one single-wave workgroup per 8x8 tile;
 *   PERSISTENT   as many waves as the GPU keeps resident, each pulling tiles from a counter until none is left
 *                (the hardware deals workgroups to its 32 shader engines in strict rotation, so with one
 *                workgroup per tile an engine that draws long-running tiles holds up the others);
 *   AUTO         (default) the scene's own choice: persistent for the built-in scenes with expensive, uneven
 *                tiles, where it measured 2-6 % faster; per tile for the others and for run-time scenes. */
typedef enum sdfr_launch_mode
{
	SDFR_LAUNCH_AUTO = 0,
	SDFR_LAUNCH_PER_TILE = 1,
	SDFR_LAUNCH_PERSISTENT = 2
} sdfr_launch_mode;
int sdfr_set_launch_mode(sdfr_renderer *r, int mode);
/* Step shortcuts (on by default; SDFR_STEP_SHORTCUTS=0 in the environment makes off the default): a ray for which its
   scene can tell that nothing lies ahead any more (cube_sea: above the cubes and not descending) is booked as the miss it
   is going to be without marching its remaining steps.  No pixel and no ray or hit count changes; sdfr_stats.march_evals
   and the per-pixel step counters then fall short of the reference's step counts.  Off: every step is marched, the
   counters equal the reference's (what the parity tests compare).  The reference has no counterpart: it marches on
   (pshader_sdf.hlsl:179-220). */
int sdfr_set_step_shortcuts(sdfr_renderer *r, int enabled);
/* per-round HIP events around the march and shade kernels (sdfr_stats.ms_march / ms_shade); off by default */
int sdfr_set_profiling(sdfr_renderer *r, int enabled);

/* ---- output: the HDR render target SDFRenderer::render draws into
 *      (Application.cpp:274-284; R16G16B16A16_FLOAT, Postprocessing.cpp:23). ------------------ */
typedef enum sdfr_format
{
	SDFR_RGBA32F = 0, /* what the shader computes */
	SDFR_RGBA16F = 1, /* what the reference's render target stores */
	/* strips only (sdfr_render_strips / sdfr_assemble_strips): the RGBA32F frame in 13 bytes per pixel,
	 * lossless -- rgb as three floats, then one byte per pixel for alpha, which is the tone-map flag
	 * (pshader_sdf.hlsl:357-359, 0 or 1).  19 % fewer bytes through the inter-GPU gather; assembles
	 * into an RGBA32F image. */
	SDFR_STRIP_RGB32F_A8 = 2,
	/* strips only: the RGBA16F frame -- what the reference's render target stores (Postprocessing.cpp:23)
	 * -- in 7 bytes per pixel: rgb as three halves (round to nearest even of the fp32 result), then one
	 * byte per pixel for alpha.  Assembles into an RGBA16F image equal, bit for bit, to a direct
	 * SDFR_RGBA16F render.  The smallest wire format: 58 MB per 3840x2160 frame. */
	SDFR_STRIP_RGB16F_A8 = 3
} sdfr_format;

/* Render a width x height frame into `out` (device pointer if out_on_host == 0, else host).
 * pixel_stats (optional, same memory space as `out`): 3 uint32 per pixel = {rays, march
 * evaluations, hits}.  Replaces SDFRenderer::render (SDFRenderer.cpp:65-107). */
int sdfr_render(sdfr_renderer *r, int width, int height, void *out, int format, int out_on_host, uint32_t *pixel_stats);

/* A host that renders into host memory (out_on_host = 1) every frame: register the image buffer once.  It is
 * page-locked with the HIP runtime, so the copy of a frame runs at PCIe speed (3840x2160 RGBA32F: ~3 ms) instead
 * of through pageable staging (~12 ms).  The caller keeps the buffer alive and at this address until it registers
 * another one, passes NULL (unregister) or destroys the handle.  Other host destinations keep working, slower. */
int sdfr_register_host_target(sdfr_renderer *r, void *host_image, size_t bytes);

/* Multi-GPU: the frame is cut into strips of SDFR_STRIP_ROWS rows; strip s belongs to rank
 * s % world.  sdfr_render_strips renders this rank's strips into a compact buffer
 * (sdfr_strip_buffer_pixels pixels, strips in increasing order); sdfr_assemble_strips, on the
 * root, scatters a gathered [world][strip_buffer_pixels] array into the full image.  The
 * reference is single-GPU; this is the sharding of SURVEY.md 8(e). */
#define SDFR_STRIP_ROWS 8
int64_t sdfr_strip_buffer_pixels(int width, int height, int world);
int64_t sdfr_strip_buffer_bytes(int width, int height, int world, int format); /* bytes of one rank's compact buffer */
/* Unequal shares: the root's own pixels never cross a link, so when the links into the root bound the
 * frame rate the root should render more than 1 / world of it.  Of every `priv_period` consecutive
 * strips the first `priv_count` (0 <= priv_count < priv_period; 0 = off, the default) are PRIVATE
 * to the root, which renders them straight into the final image with sdfr_render_private_strips;
 * only the others are dealt round-robin, rendered into compact buffers and gathered.  Set the same
 * split on every rank's handle; it applies to sdfr_render_strips and sdfr_assemble_strips
 * (assembly leaves the private rows alone).  The _split variants size the compact buffers. */
int sdfr_set_strip_split(sdfr_renderer *r, int priv_count, int priv_period);
int64_t sdfr_strip_buffer_pixels_split(int width, int height, int world, int priv_count, int priv_period);
int64_t sdfr_strip_buffer_bytes_split(int width, int height, int world, int format, int priv_count, int priv_period);
int sdfr_render_private_strips(sdfr_renderer *r, int width, int height, void *out_image, int format);
int sdfr_render_strips(sdfr_renderer *r, int width, int height, int rank, int world, void *out_compact, int format);
int sdfr_assemble_strips(sdfr_renderer *r, int width, int height, int world, const void *gathered, void *out_image, int format);

/* ---- multi-GPU in the library itself: RCCL over xGMI (no reference counterpart --
 *      the reference drives one adapter, Graphics.cpp:34).  One communicator per process and GPU
 *      (sdfr_comm_create, from an id made on rank 0 and handed to the others by the caller's own
 *      means), or one per device of a single process (sdfr_comm_create_all).  librccl.so is opened
 *      on first use: the library has no load-time dependency on it.
 *
 *      sdfr_render_gather: every rank renders the shared strips of its rank (strip layout above, the
 *      handle's strip split included) in `wire_format` into a buffer the handle owns; the peers
 *      ncclSend theirs to rank 0, which ncclRecv's them (one group per frame, N - 1 messages on N - 1
 *      different xGMI links), scatters them into `root_image` and -- with a strip split -- renders its
 *      private strips straight into `root_image` meanwhile.  Transfer and assembly run on a stream of
 *      the handle's own; the call returns once everything is enqueued and the handle's stream is made
 *      to wait for it, so sdfr_sync (or any later work on that stream) sees the finished image.
 *        image_format SDFR_RGBA32F needs a 32-bit wire format (SDFR_RGBA32F, SDFR_STRIP_RGB32F_A8),
 *        image_format SDFR_RGBA16F a 16-bit one (SDFR_RGBA16F, SDFR_STRIP_RGB16F_A8);
 *        root_image is ignored on the other ranks (may be NULL).
 *      Every rank must call it with the same frame size, formats, strip split, and in the same order
 *      when several handles share one communicator (two frames in flight: two handles, two streams).
 *      sdfr_get_stats afterwards reports this rank's own share. -------------------------------------- */
typedef struct sdfr_comm sdfr_comm;
#define SDFR_COMM_ID_BYTES 128
int sdfr_comm_unique_id(void *id_out);                      /* ncclGetUniqueId; id_out: SDFR_COMM_ID_BYTES bytes */
int sdfr_comm_create(const void *id, int rank, int world, int device_ordinal, sdfr_comm **out); /* collective: ncclCommInitRank */
int sdfr_comm_create_all(const int *device_ordinals, int n, sdfr_comm **out_n);                 /* one process: ncclCommInitAll */
/* Teardown, bounded in time: drains the streams the communicator's transfers ran on (the comm streams of the handles
 * that used it), then ncclCommFinalize + ncclCommDestroy; if that has not finished within SDFR_COMM_CLOSE_TIMEOUT_S
 * (default 30 s) ncclCommAbort is tried and SDFR_ERR_COMM comes back with the call it was stuck in (sdfr_comm_last_error
 * (NULL)) instead of a hang.  Collective in effect: every rank closes.  `c` is gone afterwards either way.
 * sdfr_comm_destroy is the same without the status. */
int sdfr_comm_close(sdfr_comm *c);
void sdfr_comm_destroy(sdfr_comm *c);
/* Which librccl serves this library (path_out, may be NULL), its ncclGetVersion code, and how many DISTINCT librccl files
 * the process maps: more than one means some other component loaded a second copy by path -- a process must not run two
 * (DESIGN.md section 7).  The library itself opens a copy the process already maps (PyTorch's) before any other. */
int sdfr_comm_library_info(char *path_out, size_t path_bytes, int *nccl_version, int *copies_mapped);
int sdfr_comm_rank(const sdfr_comm *c);
int sdfr_comm_world(const sdfr_comm *c);
const char *sdfr_comm_last_error(const sdfr_comm *c);
/* `bytes` bytes travel rank -> (rank + 1) % world -> ... on `hip_stream` (world = 1: to itself) and are
 * compared at the destination: proves that the library, the communicator and the links work.  Blocking and
 * collective: every rank calls it (one process per rank, or one thread per communicator of sdfr_comm_create_all). */
int sdfr_comm_selftest(sdfr_comm *c, size_t bytes, void *hip_stream);
int sdfr_render_gather(sdfr_renderer *r, sdfr_comm *c, int width, int height, void *root_image, int image_format, int wire_format);
/* the same for the n handles / communicators of ONE process (sdfr_comm_create_all), rank i = index i */
int sdfr_render_gather_all(sdfr_renderer *const *r, sdfr_comm *const *c, int n, int width, int height, void *root_image, int image_format,
	int wire_format);

/* ---- the consumer of the render target: HDR::process
 *      (Postprocessing.cpp:130-174; bloom.hlsl; pshader_hdr.hlsl).  scene = the RGBA16F frame of
 *      sdfr_render; bloom_scratch = width*height*8 bytes; out = R8G8B8A8_UNORM (Graphics.cpp:65).
 *      Device pointers; enqueued on the handle's stream. ------------------------------------------ */
int sdfr_postprocess(sdfr_renderer *r, int width, int height, const void *scene_rgba16f, void *bloom_scratch_rgba16f, void *out_rgba8);

int sdfr_sync(sdfr_renderer *r);

/* ---- two frames in flight inside one handle (no counterpart: D3D11's immediate context pipelines the reference's draws by itself)
 *      The end of a frame runs on a nearly empty chip -- the last waves finishing their tiles -- and only the NEXT frame can fill it
 *      (DESIGN.md 4.1).  With n = 2 sdfr_render alternates between two internal streams, each with a workspace of its own (the
 *      memory of the ray queue twice), so that frame k + 1 starts while frame k drains; pixels and counters are those of n = 1.
 *        - sdfr_render returns at once, as always; the frames of one handle may finish out of order.
 *        - the stream given to sdfr_set_stream is not used while n = 2: order other work against a frame with
 *          sdfr_wait_frame(r, stream) -- `stream` waits (on the device, not the host) for the frame submitted last -- or sdfr_sync,
 *          which waits for both frames.  sdfr_get_stats / sdfr_get_timings report the frame submitted last (and wait for it).
 *        - two frames in flight write two buffers: a frame rendered into memory that overlaps the destination of the frame still in
 *          flight waits for that frame first (correct, but nothing overlaps) -- alternate between two images.
 *        - sdfr_render_strips / _gather / sdfr_postprocess run on the lane of the frame submitted last.
 *      n = 1 (the default) returns to one stream (the caller's) after waiting for both frames. ------------------------------------- */
int sdfr_set_frames_in_flight(sdfr_renderer *r, int n);
int sdfr_wait_frame(sdfr_renderer *r, void *hip_stream);

/* ---- observability: GPUProfiler::profile("setup"/"draw") (SDFRenderer.cpp:100,104) ---------- */
typedef struct sdfr_stats
{
	double ms_gpu;          /* HIP-event time of the last render on the handle's stream */
	double ms_march;        /* wavefront schedule: time inside the march kernels */
	double ms_shade;        /* wavefront schedule: time inside the shade kernels */
	uint64_t pixels;
	uint64_t rays;          /* bounce-loop iterations = primary + secondary rays */
	uint64_t march_evals;   /* scene-distance evaluations made while marching */
	uint64_t hits;
	uint32_t march_launches, shade_launches;
} sdfr_stats;
/* waits for the last render, then reports it */
int sdfr_get_stats(sdfr_renderer *r, sdfr_stats *out);

/* ---- named GPU timings: GPUProfiler::profile/getResults (GPUProfiler.h:12-35) with the names the
 *      reference's frame uses ("setup", "draw" SDFRenderer.cpp:100-104; "Bloom 1", "Bloom 2", "HDR"
 *      Postprocessing.cpp:147-171).  "setup" is host time (uniform latch); the vertical blur and
 *      the tone map are one kernel here and report as "Bloom 2 + HDR"; with sdfr_set_profiling the
 *      wavefront schedule adds "draw: march k" / "draw: shade k".  Covers the last sdfr_render* and
 *      the last sdfr_postprocess; waits for them.  Returns the number of entries available (may
 *      exceed `capacity`; only `capacity` are written), or a negative status. ------------------- */
typedef struct sdfr_timing
{
	char name[32];
	double ms;
} sdfr_timing;
int sdfr_get_timings(sdfr_renderer *r, sdfr_timing *out, int capacity);

/* ---- self-test of the kernels' fast exact arithmetic (no reference counterpart) ---------------
 * The kernels replace hipcc's generic correctly rounded fp32 sqrt, and divisions by scene
 * constants, with shorter sequences that give the SAME correctly rounded bits on their stated
 * domain (sdf_playground_amd/csrc/sdfr_math.h: sqrt1, div_c).  This runs the exhaustive
 * comparison on the GPU and returns the number of differing inputs (expected: 0).
 *   what = 0             sqrt1(a)   vs IEEE sqrt   for a = +0 and all a in [2^-96, FLT_MAX]
 *   what = 1, constant c a / c      vs IEEE divide for a = +-0 and all 2^-100 <= |a| <= 2^110
 *   what = 2, constant c negative control: a * (1/c) vs IEEE divide on the same inputs (> 0)
 *   what = 3, constant c as 1 but for a = +-0 and all 2^-60 <= |a| <= 2^40 (fast ground plane) */
int sdfr_selftest_math(sdfr_renderer *r, int what, float constant, uint64_t *mismatches);
/* Throws a C++ exception inside the library the way an allocation failure or a regex error would (what = 0: std::runtime_error,
 * 1: std::bad_alloc, 2: something that is not a std::exception) and returns what the guard at the C boundary makes of it:
 * SDFR_ERR_INTERNAL, with the exception's words in sdfr_last_error(r) when r is not NULL.  Needs no device; r may be NULL. */
int sdfr_selftest_exception(sdfr_renderer *r, int what);

#ifdef __cplusplus
}
#endif
#endif /* SDFR_H */
