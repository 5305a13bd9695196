#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5|5g]

Headline workload (BASELINE.json configs[2], concretised in SURVEY.md 8(d) cfg 3; --config 3, the
default): the `labyrinth` scene at 3840x2160, iter_count 256, reference cost rules, default variables,
on the fixed-seed 16-frame camera sweep  eye = (1.5 cos t, 5, 1.5 sin t), dir = (cos t, -0.35, sin t),
t = 2 pi (k + u_k) / 16, u_k = pcg_hash(0x5DF00003 + k) / 4294967295, stime = k / 60.
The other BASELINE configurations (SURVEY.md 8(d) cfg 2, 4, 5) run with --config; they are measured
for profiles/, they are not the headline.  One "step" = one full frame through the hot path (step s
renders sweep frame s % 16).  Metric: Mrays/s, ray := one iteration of the reference's bounce loop
(primary + secondary).

N > 1: one process per GPU.  Started by torch.distributed.run (the driver's way: RANK / LOCAL_RANK /
WORLD_SIZE in the environment), or -- when WORLD_SIZE is not set -- by this script itself, which
starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process BEFORE
anything touches a GPU and relays rank 0's line and the exit code.  Each frame is cut into 8-row
strips dealt round-robin over the ranks (strong scaling: the frame is fixed); the library renders
this rank's strips and gathers them on rank 0 with RCCL (sdfr_render_gather: ncclSend / ncclRecv over
xGMI, strips travel as SDFR_STRIP_RGB16F_A8, the reference's RGBA16F target in 7 bytes per pixel);
two frames are in flight on two handles.  value = rays of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the host driver of this pool only supports dmabuf IPC: without this RCCL's cross-process buffer
# sharing fails with hipIpcGetMemHandle (already exported by the image; kept for bare shells)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

SWEEP = 16
SPLIT_PERIOD = 16     # strips per period of the private/shared split (N > 1)
PEAK_FP32_VECTOR_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters (spec); 2 flop per lane per FMA at the packed rate
PEAK_VALU_LANE_OPS = 78.6e12      # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz: what the VALU can issue, one op per lane per cycle
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md (spec)
PROFILE_ROUND = "r04"   # the round whose committed counter profiles (profiles/pmc_<round>_cfg<c>.json) feed `roofline.executed`
CENSUS_FILE = os.path.join(ROOT, "profiles", "census_r03.json")


def pcg_hash(x):
    """The reference's PCG hash (noise.hlsl:6-11), uint32 wrap-around."""
    state = (x * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return ((word >> 22) ^ word) & 0xFFFFFFFF


def _theta(seed, k):
    return 2.0 * math.pi * (k + pcg_hash((seed + k) & 0xFFFFFFFF) / 4294967295.0) / SWEEP


# ---- the BASELINE.json configurations as concretised by SURVEY.md 8(d) --------------------------------
# camera(k) -> (kind, eye, target, stime); kind "dir": target is the viewing direction, "lookat": a point
def _cam2(k):
    t = _theta(0x5DF00002, k)
    return "dir", (3 * math.cos(t), 4.5, 3 * math.sin(t)), (math.cos(t + 0.6), -0.45, math.sin(t + 0.6)), k / 60.0


def _cam3(k):
    t = _theta(0x5DF00003, k)
    return "dir", (1.5 * math.cos(t), 5.0, 1.5 * math.sin(t)), (math.cos(t), -0.35, math.sin(t)), k / 60.0


def _cam4(k):
    t = _theta(0x5DF00004, k)
    return "lookat", (2.2 * math.cos(t), 1.6, 2.2 * math.sin(t)), (0.0, 1.0, 0.0), 0.0


def _cam5_lense(k):
    ph = -0.5 + (k + pcg_hash((0x5DF00005 + k) & 0xFFFFFFFF) / 4294967295.0) / SWEEP
    return "lookat", (7 * math.sin(ph), 0.5, 7 * math.cos(ph)), (0.0, 0.0, 0.0), k / 60.0


def _cam5_gems(k):
    t = _theta(0x5DF00005, k)
    return "lookat", (2.5 * math.cos(t), 2.0, 2.5 * math.sin(t)), (0.0, 1.0, 0.0), k / 60.0


def _cam1(k):
    return "lookat", (0.0, 2.0, -3.0), (0.0, 1.0, 0.0), 0.0   # the reference's start-up camera (Application.cpp:214-224)


CONFIGS = {
    # BASELINE.json configs[0]: "Single sphere SDF, 256x256, 64 max steps, 1 light, no secondary rays -- CPU scalar raymarch of scene()
    # (plumbing, no GPU)".  SURVEY.md 8(d) cfg 1: fast_sphere, iter_count 64, max_cost_default 2 (no child ray passes depth + 2 < 2).
    # `python bench.py --config 1` is the CPU line (the oracle, in full); it needs no GPU and renders on one only if there is one.
    "1": dict(key="fast_sphere_256_iter64", scene="fast_sphere", width=256, height=256, limits=dict(iter_count=64, max_cost_default=2),
              camera=_cam1, metric="Mrays/s (primary rays only), fast_sphere scene, 256x256, CPU scalar raymarch",
              workload="fast_sphere %dx%d, iter_count 64, max_cost_default 2 (no secondary rays; both labelled extensions of the reference's 100 / 7), "
                       "the reference's start-up camera, stime 0"),
    "2": dict(key="cube_sea_1080p_iter128", scene="cube_sea", width=1920, height=1080, limits=dict(iter_count=128, max_cost_default=6),
              camera=_cam2, metric="Mrays/s (primary+secondary), cube_sea scene, 1920x1080",
              workload="cube_sea %dx%d, iter_count 128, max_cost_default 6 (= exactly one reflection bounce, the configuration as worded; labelled "
                       "extension of the reference's 100 / 7), 16-frame fixed-seed camera sweep (seed 0x5DF00002), default variables"),
    "3": dict(key="labyrinth_4k_iter256", scene="labyrinth", width=3840, height=2160, limits=dict(iter_count=256),
              camera=_cam3, metric="Mrays/s (primary+secondary), labyrinth scene, 3840x2160",
              workload="labyrinth %dx%d, iter_count 256, reference cost rules (max_cost 7, hard shadows), 16-frame fixed-seed camera sweep "
                       "(seed 0x5DF00003), default variables"),
    "3r": dict(key="labyrinth_4k_iter256_reflective", scene="labyrinth", width=3840, height=2160,
               limits=dict(iter_count=256, extension_marble_reflection=0.25),
               camera=_cam3, metric="Mrays/s (primary+secondary), labyrinth scene with reflective marble (labelled extension), 3840x2160",
               workload="LABELLED EXTENSION, not the parity target: labyrinth %dx%d, iter_count 256, the marble of walls and vases given "
                        "reflection_color 0.25 (the reference's labyrinth has no reflective material; BASELINE configs[2] is worded '2 reflection "
                        "bounces': with the default cost rule a ray is reflected at most twice), hard shadows, 16-frame fixed-seed camera sweep "
                        "(seed 0x5DF00003), default variables"),
    "4": dict(key="fractal_4k_iter512", scene="fractal", width=3840, height=2160, limits=dict(iter_count=512),
              camera=_cam4, metric="Mrays/s (primary+secondary), fractal scene, 3840x2160",
              workload="fractal %dx%d, iter_count 512 (labelled extension of the reference's 100), reference cost rules, 16-frame fixed-seed camera "
                       "sweep (seed 0x5DF00004), stime 0"),
    "5": dict(key="lense_4k_depth4_8lights", scene="lense", width=3840, height=2160, limits=dict(iter_count=100, max_cost_default=9, extension_lights=7),
              camera=_cam5_lense, metric="Mrays/s (primary+secondary), lense scene, 3840x2160",
              workload="lense %dx%d, iter_count 100, max_cost_default 9 (recursion depth 4) and 7 orbiting point lights besides the scene's own "
                       "(labelled extensions), 16-frame fixed-seed camera sweep (seed 0x5DF00005), default variables"),
    "5g": dict(key="gems_4k_depth4_8lights", scene="gems", width=3840, height=2160, limits=dict(iter_count=100, max_cost_default=9, extension_lights=7),
               camera=_cam5_gems, metric="Mrays/s (primary+secondary), gems scene, 3840x2160",
               workload="gems %dx%d, iter_count 100, max_cost_default 9 (recursion depth 4) and 7 orbiting point lights besides the scene's own "
                        "(labelled extensions), 16-frame fixed-seed camera sweep (seed 0x5DF00005), default variables"),
}
HEADLINE = "3"
# names the headline keeps for tests and tools written against round 1
SCENE, WIDTH, HEIGHT, ITER_COUNT, SEED = "labyrinth", 3840, 2160, 256, 0x5DF00003


def sweep_camera(k, config=HEADLINE):
    """(eye, direction-or-lookat, stime) of sweep frame k."""
    _, eye, target, stime = CONFIGS[config]["camera"](k)
    return eye, target, stime


def make_camera(k, width=None, height=None, config=HEADLINE):
    import numpy as np
    import sdf_playground_amd as sp

    cfg = CONFIGS[config]
    width, height = width or cfg["width"], height or cfg["height"]
    kind, eye, target, stime = cfg["camera"](k)
    cam = sp.Camera()
    cam.SetEye(eye)
    (cam.SetLookat if kind == "lookat" else cam.SetDirection)(target)
    cam.SetFOVY(sp.to_radian(60.0))
    cam.SetAspect(float(np.float32(width) / np.float32(height)))
    return cam, stime


def oracle_frame(po, k, width, height, config=HEADLINE):
    """The oracle's frame description of sweep frame k (cpu_baseline, tools/make_census.py)."""
    import numpy as np
    import sdf_playground_amd as sp

    cfg = CONFIGS[config]
    kind, eye, target, stime = cfg["camera"](k)
    fovy, aspect = np.float32(sp.to_radian(60.0)), np.float32(width) / np.float32(height)
    basis = (po.camera_lookat if kind == "lookat" else po.camera_direction)(eye, target, fovy, aspect)
    f = po.default_frame(cfg["scene"], width, height, basis=basis, stime=stime)
    for name, v in cfg["limits"].items():
        setattr(f, name, v)
    return f


def load_census(config=HEADLINE):
    """flops per ray of a workload, from the operation-counting oracle build (committed)."""
    try:
        with open(CENSUS_FILE) as fh:
            return json.load(fh)[CONFIGS[config]["key"]]
    except Exception:
        return None


def load_pmc(config, width, height, schedule_name):
    """The executed-work view of the dominant kernel (rocprofv3 --pmc passes over this same command,
    tools/collect_profiles.sh).  Returned only when the profile is of THIS workload; it is a committed
    measurement of an earlier run, and says so (source, commit)."""
    path = os.path.join(ROOT, "profiles", "pmc_%s_cfg%s.json" % (PROFILE_ROUND, config))
    try:
        with open(path) as fh:
            pj = json.load(fh)
    except Exception:
        return None
    wl = pj.get("workload", {})
    if wl.get("config") != config or wl.get("width") != width or wl.get("height") != height or wl.get("schedule") != schedule_name:
        return None
    for name, k in pj.get("kernels", {}).items():
        if "k_pixel" in name:
            lane_ops = k["SQ_INSTS_VALU"] * 64.0 * k["valu_lane_utilization"]
            return {
                "source": os.path.relpath(path, ROOT), "commit": pj.get("commit"), "measured_in_this_run": False,
                "valu_wave_instructions_per_launch": k["SQ_INSTS_VALU"], "valu_lane_utilization": k["valu_lane_utilization"],
                "cycles_per_valu_instruction_per_simd": k["cycles_per_valu_inst_per_simd"], "waves_per_simd": k["avg_waves_per_simd"],
                "executed_lane_ops_per_launch": lane_ops, "hbm_bytes_per_launch": pj.get("hbm_bytes_per_launch"),
                "profiled_kernel_ms": k.get("avg_ms"),
            }
    return None


def host_cores():
    """CPUs this process may really use: the scheduler affinity and the cgroup CPU quota both
    cap it (a GPU box reports every core of the host but grants a share of them)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(math.ceil(float(quota) / float(period)))))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            n = min(n, max(1, int(math.ceil(q / per))))
    except Exception:
        pass
    return n


def cpu_baseline(config=HEADLINE, width=None, height=None, max_seconds=20.0):
    """The CPU oracle (a port: the reference HLSL cannot run here) timed on the host cores on a
    bounded sample of the same workload: the 16 sweep frames (fewer on slow hosts: it stops after
    max_seconds), every 2nd pixel in x and y; threads = the CPUs the cgroup really grants."""
    from oracle import pyoracle as po

    cfg = CONFIGS[config]
    width, height = width or cfg["width"], height or cfg["height"]
    cores = host_cores()
    step = 2
    rays = pixels = used = 0
    t_total = 0.0
    for k in range(SWEEP):
        f = oracle_frame(po, k, width, height, config)
        t0 = time.perf_counter()
        _, _, tot = po.render(cfg["scene"], f, step=(step, step), nthreads=cores)
        t_total += time.perf_counter() - t0
        pixels += int(tot[0])
        rays += int(tot[1])
        used += 1
        if t_total > max_seconds:
            break
    return {
        "value": rays / t_total / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": "oracle (scalar C++ restatement, g++ -O2 -ffp-contract=off), sweep frames 0..%d of the same %dx%d workload, every %dth pixel in x and y (%d pixels, %d rays, %.1f s)" % (used - 1, width, height, step, pixels, rays, t_total),
        "ms_per_frame_equiv": t_total / used * step * step * 1e3,
    }


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: two sweeps of warm-up (row feedback, clocks: with 4 warm-up steps the 32 timed ones measured 1.2 % slower), four sweeps timed
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--config", default=HEADLINE, choices=sorted(CONFIGS), help="BASELINE.json configuration (SURVEY.md 8d); 3 = the headline; 3r = --variant reflective")
    ap.add_argument("--variant", default="", choices=["", "reflective"],
                    help="reflective: config 3 with the labelled extension 'marble reflection_color 0.25' (BASELINE configs[2] as worded); never the headline")
    ap.add_argument("--schedule", default=os.environ.get("SDFR_SCHEDULE", "auto"), choices=["auto", "wavefront", "pixel"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact-steps", action="store_true",
                    help="march every step of every ray (sdfr_set_step_shortcuts off): the step counters then equal the reference's; the pixels are the same either way")
    ap.add_argument("--min-seconds", type=float, default=None,
                    help="N = 1: after the K timed steps keep looping the sweep for at least this long and report that steady-state figure beside "
                         "the K-step one (`steady`): 64 steps are 80 ms, shorter than a clock ramp (default: 2 s; 0 with --no-extra-passes)")
    ap.add_argument("--no-extra-passes", action="store_true",
                    help="N = 1: skip the informational passes after the timed region (every step marched; the reference's own iter_count 100)")
    ap.add_argument("--no-second-pass", action="store_true",
                    help="skip the informational two-frames-in-flight pass (N = 1): under a profiler its overlapping kernels would pollute the per-kernel statistics")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--force-distributed", action="store_true", help="take the strips + gather path even with one rank (testing)")
    ap.add_argument("--verify", action="store_true", help="after timing, check rank 0's assembled image of the last frame against a direct render")
    ap.add_argument("--private-strips", default="auto",
                    help="N > 1: of every 16 strips, how many rank 0 renders privately (its pixels do not travel); 'auto' = the fastest of a few "
                         "candidates timed during start-up")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "torch", "gloo"],
                    help="N > 1: rccl = the library's own gather (sdfr_render_gather); torch = torch.distributed.gather of the same strips "
                         "(also taken, on every rank, when the library's communicator cannot be made); gloo = the same strips staged "
                         "through host memory and gathered over gloo -- slow, for rehearsing N ranks where RCCL cannot run, e.g. several "
                         "ranks on ONE GPU (RCCL refuses two ranks on a device); never a result")
    ap.add_argument("--wire", default="f16", choices=["f16", "f32"],
                    help="N > 1: f16 = strips and image in the reference's RGBA16F target format (7 B/pixel on the links); f32 = lossless fp32 (13 B/pixel)")
    ap.add_argument("--frames-in-flight", type=int, default=3, help="N > 1: handles / streams the frames alternate between")
    ap.add_argument("--dry-launch", action="store_true", help="start the ranks, let each report its environment over gloo, touch no GPU (CPU test of the launcher)")
    return ap.parse_args(argv)


def _free_port():
    """A port for the launcher's rendezvous: one that binds now, from below the ephemeral range -- a port handed out by bind(0) can be taken
    as the source port of somebody's outgoing connection before torch.distributed.run listens on it (seen once on a GPU box: EADDRINUSE)."""
    import random

    for _ in range(64):
        p = random.randrange(20000, 32000)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", p))
            return p
        except OSError:
            continue
        finally:
            s.close()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(a, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes with
    torch.distributed.run and relay rank 0's JSON line and the exit code.  Runs before this process
    has made any GPU call, and starts children -- it never replaces itself."""
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    for attempt in range(3):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        sys.stderr.write(p.stderr)
        # the launcher could not listen on its port: no rank was started, nothing ran -- take another port (not a retry of a run that failed)
        if p.returncode != 0 and "EADDRINUSE" in p.stderr and not p.stdout.strip():
            continue
        break
    line = None
    for l in p.stdout.splitlines():
        try:
            d = json.loads(l)
            if isinstance(d, dict) and ("metric" in d or "dry_launch" in d):
                line = l
                continue
        except Exception:
            pass
        if l.strip():
            print(l, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if p.returncode != 0:
        return p.returncode
    return 0 if line is not None else 3


def dry_launch(a):
    """What a rank sees of its launch, gathered over gloo; no GPU call."""
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    sys.stdout.flush()
    saved_stdout = os.dup(1)   # gloo reports its connections on stdout; the contract is ONE line there
    os.dup2(2, 1)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "world_size": world, "pid": os.getpid(),
            "master_addr": os.environ.get("MASTER_ADDR"), "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    every = [None] * world
    dist.all_gather_object(every, mine)
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "gpus_arg": a.gpus, "ranks": every}), flush=True)
    os.dup2(2, 1)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse_args(argv)
    if a.variant == "reflective":
        if a.config != HEADLINE:
            print("bench.py: --variant reflective belongs to --config 3", file=sys.stderr)
            return 2
        a.config = "3r"
    if a.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return spawn_ranks(a, argv)      # BEFORE any GPU call (nothing below this line has run yet)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" in os.environ and world != a.gpus:
        print("bench.py: launched with WORLD_SIZE=%d but --gpus %d" % (world, a.gpus), file=sys.stderr)
        return 2
    if a.dry_launch:
        return dry_launch(a)
    if a.config == "1":
        return run_config1(a)
    return run(a, world)


def run_config1(a):
    """BASELINE.json configs[0]: the CPU scalar raymarch of fast_sphere at 256x256, 64 steps, no secondary rays -- the oracle,
    in full (every pixel, every step), threaded over the host cores the box grants; K steps = K renders of the frame.  No GPU
    is needed; if one is there the HIP path renders the same frame as well (reported beside, and compared bit for bit)."""
    import numpy as np
    from oracle import pyoracle as po

    cfg = CONFIGS["1"]
    W, H = a.width or cfg["width"], a.height or cfg["height"]
    f = oracle_frame(po, 0, W, H, "1")
    cores = host_cores()
    for _ in range(max(1, a.warmup)):
        ref, rst, tot = po.render(cfg["scene"], f, stats=True, nthreads=cores)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        po.render(cfg["scene"], f, nthreads=cores)
    elapsed = time.perf_counter() - t0
    t1 = time.perf_counter()
    po.render(cfg["scene"], f, nthreads=1)
    single = time.perf_counter() - t1
    rays = int(tot[1])
    out = {
        "metric": cfg["metric"], "value": rays * a.steps / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": 0, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["workload"] % (W, H), "baseline_config": "1", "parallelism": "cpu%d" % cores,
                   "rays_per_pixel": rays / float(W * H), "march_evals_per_ray": float(tot[2]) / max(1, rays)},
        "cpu_baseline": {"value": rays * a.steps / elapsed / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
                         "sample": "oracle (scalar C++ restatement, g++ -O2 -ffp-contract=off), the whole %dx%d frame, %d renders, %.2f s; one thread: %.3f Mrays/s"
                                   % (W, H, a.steps, elapsed, rays / single / 1e6)},
    }
    try:
        import torch

        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        import sdf_playground_amd as sp

        r = sp.SDFRenderer(0)
        r.initShader(cfg["scene"])
        r.setLimits(**cfg["limits"])
        cam, stime = make_camera(0, W, H, "1")
        r.setParameters(stime)
        img = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
        for _ in range(3):
            r.render(cam, W, H, out=img)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(a.steps):
            r.render(None, W, H, out=img)
        torch.cuda.synchronize()
        gms = (time.perf_counter() - t2) / a.steps * 1e3
        st = r.getStats()
        out["gpu"] = {"ms_per_step": gms, "value": float(st.rays) / (gms * 1e-3) / 1e6, "unit": "Mrays/s", "kernel_ms": st.ms_gpu,
                      "bit_identical_to_cpu": bool(np.array_equal(img.cpu().numpy().view(np.uint32), ref.view(np.uint32))) and int(st.rays) == rays,
                      "note": "the HIP path on the same frame, informational: a 65 536-pixel frame is 1 024 waves, a seventh of what the chip holds"}
        r.close()
    print(json.dumps(out), flush=True)
    return 0


def run(a, world):
    import numpy as np
    import torch

    import sdf_playground_amd as sp

    cfg = CONFIGS[a.config]
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # a launcher that gives every rank its own visible device (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank)
    # leaves exactly one ordinal; counting devices does not initialise the GPU on this image
    if local_rank >= max(1, torch.cuda.device_count()):
        local_rank = 0
    distributed = world > 1 or a.force_distributed
    saved_stdout = None
    dist = None
    if distributed:
        import faulthandler

        import torch.distributed as dist

        # a rank that hangs (a collective that never completes) must end the run, not sit in it: dump every thread's
        # stack and exit after 10 minutes without reaching the end
        faulthandler.dump_traceback_later(600, exit=True)

        # RCCL prints a version banner on stdout when the first communicator is made; the contract
        # is ONE line on stdout, so stdout points at stderr until the result line is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        if a.transport == "gloo":
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    host_staged = distributed and a.transport == "gloo"
    ctl = "cpu" if host_staged else "cuda"   # where the control tensors of the collectives live
    W, H = a.width or cfg["width"], a.height or cfg["height"]
    scene = cfg["scene"]
    schedule = {"auto": sp.SCHEDULE_PIXEL, "wavefront": sp.SCHEDULE_WAVEFRONT, "pixel": sp.SCHEDULE_PIXEL}[a.schedule]
    schedule_name = "pixel" if schedule == sp.SCHEDULE_PIXEL else "wavefront"

    def make_renderer(stream=None):
        h = sp.SDFRenderer(local_rank)
        h.initShader(scene)
        h.setLimits(**cfg["limits"])
        h.setSchedule(schedule)
        h.setStepShortcuts(not a.exact_steps)  # the library's default, said out loud: same pixels, rays and hits; rays known to be misses stop marching
        if distributed:
            # the persistent launch keeps every wave slot of the GPU until its frame ends; the transfer kernels of the
            # frame before (RCCL, another stream) would wait for slots instead of overlapping: one wave per tile here
            h.setLaunchMode(sp.LAUNCH_PER_TILE)
        if stream is not None:
            h.setStream(stream.cuda_stream)
        return h

    stream = torch.cuda.current_stream()
    r = make_renderer(stream)
    cameras = [make_camera(k, W, H, a.config) for k in range(SWEEP)]

    wire16 = a.wire == "f16"
    img_fmt = sp.RGBA16F if (distributed and wire16) else sp.RGBA32F
    img_dtype = torch.float16 if img_fmt == sp.RGBA16F else torch.float32
    wire_fmt = sp.STRIP_RGB16F_A8 if wire16 else sp.STRIP_RGB32F_A8
    depth = max(1, a.frames_in_flight) if distributed else 1
    images = [torch.empty((H, W, 4), dtype=img_dtype, device="cuda") for _ in range(depth)]
    image = images[0]

    transport = None
    comm = None
    split = (0, SPLIT_PERIOD)
    calibration = None
    if distributed:
        # frames alternate between `depth` renderer handles on as many streams, so that the tail of
        # frame k (a handful of waves still marching long rays: one wave alone needs ~0.2 ms, as long as
        # a rank's whole share of the frame at 8 GPUs) overlaps the head of frame k+1, and so do the
        # transfer and the assembly of frame k
        rs = [stream] + [torch.cuda.Stream() for _ in range(depth - 1)]
        rr = [r] + [make_renderer(rs[b]) for b in range(1, depth)]
        transport = a.transport
        if transport == "gloo":
            transport = "gloo (strips staged through host memory: a rehearsal of the N-rank flow, not a result)"
        if a.transport == "rccl":
            ok = torch.ones(1, dtype=torch.int32, device=ctl)
            try:
                ids = [sp.Comm.unique_id() if rank == 0 else None]
            except Exception as e:
                print("bench.py: rank %d: no communicator id (%r)" % (rank, e), file=sys.stderr)
                ids = [None]
                ok[0] = 0
            dist.broadcast_object_list(ids, src=0)
            if ids[0] is None:
                ok[0] = 0
            else:
                try:
                    comm = sp.Comm(ids[0], rank, world, local_rank)
                    comm.selftest(1 << 20, stream.cuda_stream)
                except Exception as e:
                    print("bench.py: rank %d: the library's communicator failed (%r); falling back to torch.distributed" % (rank, e), file=sys.stderr)
                    ok[0] = 0
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:     # every rank takes the same way
                transport = "torch (fallback: sdfr_comm_create or its self-test failed on some rank)"
                if comm is not None:
                    comm.close()
                    comm = None
        if comm is None:
            side = torch.cuda.Stream()
            r_asm = r_priv = None
            if rank == 0:
                r_asm = sp.SDFRenderer(local_rank)  # a handle bound to the side stream, for the assembly kernel
                r_asm.setStream(side.cuda_stream)
                r_priv = [make_renderer(rs[b]) for b in range(depth)]  # private strips: handles of their own, so that counters do not mix
            works = [None] * depth
            asm_done = [torch.cuda.Event() for _ in range(depth)]
            asm_used = [False] * depth
            local = gathered_flat = gather_lists = None

        def set_split(sp_split):
            nonlocal local, gathered_flat, gather_lists
            for h_ in rr + (([r_asm] + r_priv) if (comm is None and rank == 0) else []):
                h_.setStripSplit(*sp_split)
            if comm is None:
                torch.cuda.synchronize()
                nb = sp.strip_buffer_bytes(W, H, world, wire_fmt, sp_split)
                local = [torch.empty((nb,), dtype=torch.uint8, device="cuda") for _ in range(depth)]
                gathered_flat = [torch.empty((world, nb), dtype=torch.uint8, device="cuda") for _ in range(depth)] if rank == 0 else None
                gather_lists = [list(g.unbind(0)) for g in gathered_flat] if rank == 0 else None

    def step(s):
        """Enqueue frame s; returns the handles that render it (their counters add up to this rank's share)."""
        cam, stime = cameras[s % SWEEP]
        if not distributed:
            r.setParameters(stime)
            r.setCamera(cam)
            r.render(None, W, H, out=image)
            return [r]
        b = s % depth
        h = rr[b]
        h.setParameters(stime)
        h.setCamera(cam)
        if comm is not None:
            h.renderGather(comm, W, H, out=images[b] if rank == 0 else None, fmt=img_fmt, wire=wire_fmt)
            return [h]
        used = [h]
        if host_staged:
            # rehearsal transport: render, copy the strips to the host, gather over gloo, copy to the GPU, assemble -- one
            # frame at a time, every step synchronous
            with torch.cuda.stream(rs[b]):
                h.renderStrips(W, H, rank, world, local[b], fmt=wire_fmt)
            rs[b].synchronize()
            mine = local[b].cpu()
            if rank == 0:
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.gather(mine, gather_list=parts, dst=0)
                with torch.cuda.stream(rs[b]):
                    gathered_flat[b].copy_(torch.stack(parts))
                    if split[0] > 0:
                        hp = r_priv[b]
                        hp.setParameters(stime)
                        hp.setCamera(cam)
                        hp.renderPrivateStrips(W, H, images[b], fmt=img_fmt)
                        used.append(hp)
                    r_asm.setStream(rs[b].cuda_stream)
                    r_asm.assembleStrips(W, H, world, gathered_flat[b], images[b], fmt=wire_fmt)
            else:
                dist.gather(mine, dst=0)
            return used
        with torch.cuda.stream(rs[b]):
            if works[b] is not None:
                works[b].wait()                    # this frame's stream: local[b] is free once the gather `depth` frames ago is done
            if rank == 0 and asm_used[b]:
                rs[b].wait_event(asm_done[b])      # gathered_flat[b] is free once that frame's assembly is done
            h.renderStrips(W, H, rank, world, local[b], fmt=wire_fmt)
            if rank == 0:
                works[b] = dist.gather(local[b], gather_list=gather_lists[b], dst=0, async_op=True)
                if split[0] > 0:  # after the gather was issued: the private strips render while the peers' strips travel
                    hp = r_priv[b]
                    hp.setParameters(stime)
                    hp.setCamera(cam)
                    hp.renderPrivateStrips(W, H, images[b], fmt=img_fmt)
                    used.append(hp)
                with torch.cuda.stream(side):
                    works[b].wait()
                    r_asm.assembleStrips(W, H, world, gathered_flat[b], images[b], fmt=wire_fmt)
                    asm_done[b].record(side)
                    asm_used[b] = True
            else:
                works[b] = dist.gather(local[b], dst=0, async_op=True)
        return used

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    if distributed:
        # ---- how much of the frame should rank 0 keep for itself? ---------------------------------------
        # Its own pixels never cross a link, and for N > 1 the frame rate may be bounded by what the peers
        # push through their single link each.  Timed here, once: the pipeline itself with m = 0, 2, 4, ...
        # of every 16 strips private to rank 0; every rank adopts the fastest (max over ranks of the time
        # between two barriers).  A larger m must win by 3 % to be preferred.
        if a.private_strips != "auto":
            split = (max(0, min(SPLIT_PERIOD - 1, int(a.private_strips))), SPLIT_PERIOD)
            set_split(split)
            calibration = {"private_strips_of_16": split[0], "forced": True}
        else:
            trials = {}
            best = None
            for m in ((0,) if world == 1 and not a.force_distributed else (0, 2, 4, 6, 8)):
                split = (m, SPLIT_PERIOD)
                set_split(split)
                for s in range(2):
                    step(s)
                fence()
                t0c = time.perf_counter()
                for s in range(6):
                    step(s)
                fence()
                tm = torch.tensor([(time.perf_counter() - t0c) / 6 * 1e3], dtype=torch.float64, device=ctl)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                trials[m] = float(tm.item())
                if best is None or trials[m] < trials[best] * 0.97:
                    best = m
            split = (best, SPLIT_PERIOD)
            set_split(split)
            calibration = {"private_strips_of_16": best, "forced": False, "ms_per_frame_by_private_strips": trials}

    for s in range(a.warmup):
        step(s)
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for s in range(a.steps):
        es = rs[s % depth] if distributed else stream
        ev[s][0].record(es)
        step(s)
        ev[s][1].record(es)
    enqueue_ms = (time.perf_counter() - t0) / max(1, a.steps) * 1e3  # host time to enqueue a step
    fence()
    elapsed = time.perf_counter() - t0

    # rays are a property of the frames, independent of timing: counted outside the timed region, from
    # the kernels' own counters; N > 1: the duration of this rank's k_pixel launch alone comes from a
    # strips render without the gather
    rays_per_frame = []
    kernel_ms = []
    xfer_ms, xfer_bytes = [], 0
    scratch = None
    for k in range(min(SWEEP, a.steps)):
        hs = step(k)
        sts = [h_.getStats() for h_ in hs]
        rays_per_frame.append(sum(int(st.rays) for st in sts))
        if distributed and comm is not None:
            # the transfer of this frame on the handle's comm stream (sdfr_get_timings: "gather transfer <bytes> B")
            for name, ms in hs[0].getTimings().items():
                if name.startswith("gather transfer"):
                    xfer_ms.append(ms)
                    xfer_bytes = int(name.split()[2])
        if distributed and comm is not None:
            if scratch is None:
                scratch = torch.empty((sp.strip_buffer_bytes(W, H, world, wire_fmt, split),), dtype=torch.uint8, device="cuda")
            hs[0].renderStrips(W, H, rank, world, scratch, fmt=wire_fmt)
            kernel_ms.append(hs[0].getStats().ms_gpu)
        else:
            kernel_ms.append(sum(st.ms_gpu for st in sts))
    my_rays = sum(rays_per_frame[s % len(rays_per_frame)] for s in range(a.steps))
    step_ms = [e0.elapsed_time(e1) for e0, e1 in ev]

    t = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
    rays_t = torch.tensor([my_rays], dtype=torch.float64, device=ctl)
    per_rank = None
    if distributed:
        every = [torch.zeros(5, dtype=torch.float64, device=ctl) for _ in range(world)]
        mean_xfer = float(np.mean(xfer_ms)) if xfer_ms else 0.0
        dist.all_gather(every, torch.tensor([my_rays, elapsed, float(np.mean(kernel_ms)), mean_xfer, float(xfer_bytes)], dtype=torch.float64, device=ctl))
        per_rank = []
        for i, v in enumerate(every):
            e = {"rank": i, "rays": float(v[0].item()), "seconds": float(v[1].item()), "strips_kernel_ms": float(v[2].item())}
            if v[3].item() > 0:
                # a peer sends its strips over its one link to rank 0; rank 0 receives world - 1 messages at once, one per link
                links = (world - 1) if i == 0 else 1
                e.update({"transfer_ms": float(v[3].item()), "transfer_bytes": int(v[4].item()),
                          "GBps_per_link": float(v[4].item()) / links / (v[3].item() * 1e-3) / 1e9})
            per_rank.append(e)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays_t, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_rays = float(rays_t.item())

    verified = None
    if world > 1:
        a.verify = True   # an N-rank line certifies its own image: one more frame, gathered and compared with a direct render on rank 0
    if a.verify and rank == 0:
        last = a.steps - 1
        step(last)
        torch.cuda.synchronize()
        got = images[last % depth].clone()
        cam, stime = cameras[last % SWEEP]
        r.setParameters(stime)
        r.setCamera(cam)
        ref = torch.empty_like(got)
        r.render(None, W, H, out=ref, fmt=img_fmt)
        torch.cuda.synchronize()
        it = torch.int32 if img_fmt == sp.RGBA32F else torch.int16
        verified = bool(torch.equal(got.view(it), ref.view(it)))
    elif a.verify and distributed:
        step(a.steps - 1)  # every rank takes part in the extra gather
        torch.cuda.synchronize()

    scaling_base = None
    if distributed and world > 1:
        # what the driver's scaling ratio should be read against: ONE GPU rendering whole frames the way a rank renders its
        # share -- `depth` frames in flight on as many handles and streams, one wave per tile, no gather -- measured by rank 0
        # in this very job while the others wait (the N = 1 headline keeps one frame in flight: its kernel times mean what
        # they say, and it pays the tail of every frame)
        fence()
        if rank == 0:
            solo = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(depth)]

            def solo_step(s):
                h_ = rr[s % depth]
                h_.setParameters(cameras[s % SWEEP][1])
                h_.setCamera(cameras[s % SWEEP][0])
                h_.render(None, W, H, out=solo[s % depth])

            for s in range(depth + 1):
                solo_step(s)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for s in range(a.steps):
                solo_step(s)
            torch.cuda.synchronize()
            ms_b = (time.perf_counter() - tb) / a.steps * 1e3
            rays_b = 0
            for s in range(min(SWEEP, a.steps)):
                solo_step(s)
                rays_b += int(rr[s % depth].getStats().rays)
            rays_b = rays_b / min(SWEEP, a.steps)
            scaling_base = {"n_gpus": 1, "frames_in_flight": depth, "ms_per_step": ms_b, "value": rays_b / (ms_b * 1e-3) / 1e6, "unit": "Mrays/s",
                            "note": "rank 0 alone, whole frames, the same %d frames in flight and launch mode as the N-rank pipeline, no gather: "
                                    "value / scaling_base.value compares like with like" % depth}
            del solo
        fence()

    if rank == 0:
        census = load_census(a.config)
        out = {
            "metric": cfg["metric"],
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": cfg["workload"] % (W, H),
                "baseline_config": a.config,
                "schedule": schedule_name,
                "parallelism": "strips%d" % world if distributed else "single",
                "rays_per_pixel": total_rays / a.steps / (W * H),
                "step_shortcuts": not a.exact_steps,
                # persistent launches hand out the tile rows of a frame dearest first by what the PREVIOUS frame of the same scene, size and row
                # selection cost (sdfr_pixel_kernel.h, row feedback): the timed steps follow the warm-up steps of the same sweep
                "tile_row_order": "from the previous frame's row costs (warm-up frames included); pixels do not depend on it",
            },
        }
        if distributed:
            out["config"].update({
                "transport": ("sdfr_render_gather (RCCL ncclSend/ncclRecv inside libsdfr.so)" if comm is not None else
                              "torch.distributed.gather: " + str(transport)),
                "wire_format": "SDFR_STRIP_RGB16F_A8 (7 B/pixel)" if wire16 else "SDFR_STRIP_RGB32F_A8 (13 B/pixel)",
                "image": "RGBA16F on rank 0 (the reference's render-target format, Postprocessing.cpp:23)" if wire16 else "RGBA32F on rank 0",
                "frames_in_flight": depth,
                "strip_calibration": calibration,
                "per_rank": per_rank,
                # true whenever the line is NOT what `--gpus N` is meant to measure: the library's RCCL gather could not be used
                # (fallback), or a rehearsal / experimental transport was asked for
                "degraded": bool(comm is None),
            })
            if scaling_base is not None:
                out["scaling_base"] = scaling_base
                out["speedup_vs_scaling_base"] = out["value"] / scaling_base["value"]
        # roofline of the dominant kernel (k_pixel, the only kernel of the pixel schedule), this rank's
        # launch.  Kernel duration: HIP events on the launch stream -- N = 1: the pairs recorded around
        # every step of the timed region (a step launches k_pixel + the few-us counter fold and nothing
        # else); N > 1: a step also gathers, so the handle's own events around a strips render are used
        stats_pass_ms = float(np.mean(kernel_ms))
        mean_kernel_ms = stats_pass_ms if distributed else float(np.mean(step_ms))
        mean_rays = float(np.mean(rays_per_frame))
        if census:
            flops_per_launch = census["flops_per_ray"] * mean_rays
            achieved = flops_per_launch / (mean_kernel_ms * 1e-3) / 1e12
            pmc = load_pmc(a.config, W, H, schedule_name) if not distributed else None
            roof = {
                "bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_VECTOR_TFLOPS, "traffic": pmc["hbm_bytes_per_launch"] if pmc else None,
                "kernel_ms": mean_kernel_ms, "kernel_ms_counting_pass": stats_pass_ms, "flops_per_ray": census["flops_per_ray"],
                "numerator": "ALGORITHMIC: oracle operation census x rays of this launch; it counts every scene evaluation the reference makes, "
                             "also those the kernel's bounding-volume tests skip and the steps of rays it knows to be misses already (step shortcuts), so `frac` is an algorithmic-equivalent rate, not VALU busy time",
                "note": "FP32 vector (VALU) issue bounds this path, not HBM or MFMA (SURVEY.md 8d)",
            }
            if pmc:
                # executed work: lane-operations the VALU really issued (PMC) per second of THIS run's kernel time,
                # against one lane-op per lane per cycle.  The PMC counts are a committed profile of this same command.
                roof["executed"] = dict(pmc, lane_ops_per_s=pmc["executed_lane_ops_per_launch"] / (mean_kernel_ms * 1e-3),
                                        peak_lane_ops_per_s=PEAK_VALU_LANE_OPS,
                                        frac_executed=pmc["executed_lane_ops_per_launch"] / (mean_kernel_ms * 1e-3) / PEAK_VALU_LANE_OPS)
                roof["traffic_source"] = pmc["source"]
            out["roofline"] = roof
        bytes_per_launch = 16.0 * W * H / world
        out["roofline_hbm"] = {
            "bound": "hbm", "achieved": bytes_per_launch / (mean_kernel_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": bytes_per_launch / (mean_kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "note": "algorithmic bytes = one 16-byte RGBA32F store per pixel; reported because the north star asks for it",
        }
        out["step_ms_event_median"] = float(np.median(step_ms))
        out["host_enqueue_ms_per_step"] = enqueue_ms
        if verified is not None:
            out["verified"] = verified
        if not distributed and not a.no_second_pass:
            # Not the headline: the same sweep with TWO frames in flight inside the one handle (sdfr_set_frames_in_flight(2): two
            # internal streams and workspaces, frames rendered into two images in turn), measured after the timed region.  It shows
            # how much of `ms_per_step` is the tail of a frame (the last few waves march their long rays alone); `value` and
            # `roofline` above stay on one frame in flight, where a kernel's duration is its own.
            try:
                img2 = torch.empty_like(image)
                images2 = [image, img2]
                r.sync()
                r.setFramesInFlight(2)

                def step2(k):
                    r.setParameters(cameras[k % SWEEP][1])
                    r.setCamera(cameras[k % SWEEP][0])
                    r.render(None, W, H, out=images2[k & 1])

                for k in range(4):
                    step2(k)
                r.sync()
                t2 = time.perf_counter()
                for k in range(a.steps):
                    step2(k)
                r.sync()
                ms2 = (time.perf_counter() - t2) / max(1, a.steps) * 1e3
                out["two_frames_in_flight"] = {"ms_per_step": ms2, "value": total_rays / a.steps / (ms2 * 1e-3) / 1e6, "unit": "Mrays/s",
                                               "note": "informational: same frames, sdfr_set_frames_in_flight(2) on the one handle; not the headline"}
            except Exception as e:
                out["two_frames_in_flight"] = {"error": repr(e)}
            finally:
                r.setFramesInFlight(1)
        if not distributed and not a.no_extra_passes:
            # Informational passes over the same 16 sweep frames, after the timed region, one frame in flight:
            #   exact_steps        every step of every ray marched (sdfr_set_step_shortcuts off): the reference's own step count
            #   reference_limits   the reference's compile-time limits untouched (iter_count 100, max_cost 7, ...): "pure reference"
            def extra_pass(limits, shortcuts):
                h = make_renderer(stream)
                h.setLimits(**limits)
                h.setStepShortcuts(shortcuts)
                for k in range(3):
                    h.setParameters(cameras[k][1])
                    h.render(cameras[k][0], W, H, out=image)
                torch.cuda.synchronize()
                tp = time.perf_counter()
                for k in range(SWEEP):
                    h.setParameters(cameras[k][1])
                    h.render(cameras[k][0], W, H, out=image)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - tp) / SWEEP * 1e3
                rays = evals = 0
                for k in range(SWEEP):
                    h.setParameters(cameras[k][1])
                    h.render(cameras[k][0], W, H, out=image)
                    st = h.getStats()
                    rays += int(st.rays)
                    evals += int(st.march_evals)
                h.close()
                return {"ms_per_step": ms, "value": rays / SWEEP / (ms * 1e-3) / 1e6, "unit": "Mrays/s", "rays_per_pixel": rays / SWEEP / float(W * H),
                        "march_evals_per_ray": evals / max(1, rays)}

            try:
                if not a.exact_steps:
                    out["exact_steps"] = dict(extra_pass(cfg["limits"], False), note="every step marched (step shortcuts off): the reference's step count; same pixels, rays and hits")
                ref_limits = dict(iter_count=100, bounce_count=16, ray_count=8, light_count=8, range=100.0, max_cost_default=7, extension_lights=0,
                                  extension_marble_reflection=0.0)
                out["reference_limits"] = dict(extra_pass(ref_limits, not a.exact_steps),
                                               note="the reference's own limits (iter_count 100, max_cost 7, one light table): pure reference semantics, SURVEY.md 8(d)")
            except Exception as e:
                out["extra_passes_error"] = repr(e)
        min_seconds = a.min_seconds if a.min_seconds is not None else (0.0 if a.no_extra_passes else 2.0)
        if not distributed and min_seconds > 0:
            # a steady-state figure: the same steps, looped for at least --min-seconds (the K-step region is tens of milliseconds)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            n_steady = 0
            while True:
                for s in range(SWEEP):
                    step(n_steady + s)
                n_steady += SWEEP
                torch.cuda.synchronize()
                if time.perf_counter() - ts >= min_seconds:
                    break
            secs = time.perf_counter() - ts
            rays_steady = sum(rays_per_frame[s % len(rays_per_frame)] for s in range(n_steady))
            out["steady"] = {"seconds": secs, "steps": n_steady, "ms_per_step": secs / n_steady * 1e3, "value": rays_steady / secs / 1e6, "unit": "Mrays/s",
                             "note": "the sweep looped for >= --min-seconds, a synchronisation every 16 steps"}
        if not distributed and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(a.config, W, H)
            except Exception as e:  # the bench line must still be printed
                out["cpu_baseline"] = {"error": repr(e)}
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)

    if distributed:
        # The result is out.  A teardown that does not finish is a FAILURE and is reported as one: the watchdog says which
        # call the rank was in and ends the process with status 3 (a fresh exit, never a re-exec); the launcher relays it.
        # (Round 2 had a watchdog that exited 0 here, after a teardown hang in the test suite -- DESIGN.md section 7.
        # sdfr_comm_close is ordered and bounded by itself now; this is the outer net, and it fails loudly.)
        import threading

        stage = ["torch.cuda.synchronize"]

        def teardown_stuck():
            sys.stderr.write("bench.py rank %d: teardown did not finish within its limit; stuck in: %s\n" % (rank, stage[0]))
            sys.stderr.flush()
            faulthandler.dump_traceback(all_threads=True)
            os._exit(3)

        watchdog = threading.Timer(90.0 if rank == 0 else 100.0, teardown_stuck)
        watchdog.daemon = True
        watchdog.start()
        torch.cuda.synchronize()
        stage[0] = "dist.barrier (before teardown)"
        dist.barrier()
        faulthandler.cancel_dump_traceback_later()
        stage[0] = "sdfr_destroy (extra handles)"
        for h_ in rr[1:]:
            h_.close()
        if comm is not None:
            stage[0] = "sdfr_comm_close (streams drained, ncclCommFinalize, ncclCommDestroy)"
            comm.close()                # raises SdfrError if RCCL's teardown did not finish within its own limit
        stage[0] = "sdfr_destroy"
    r.close()
    if distributed:
        stage[0] = "dist.destroy_process_group"
        dist.destroy_process_group()
        watchdog.cancel()
    return 0


if __name__ == "__main__":
    sys.exit(main())
