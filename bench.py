#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], concretised in SURVEY.md 8(d) cfg 3): the `labyrinth`
scene at 3840x2160, iter_count 256, reference cost rules, default variables, on the fixed-seed
16-frame camera sweep  eye = (1.5 cos t, 5, 1.5 sin t), dir = (cos t, -0.35, sin t),
t = 2 pi (k + u_k) / 16, u_k = pcg_hash(0x5DF00003 + k) / 4294967295, stime = k / 60.
One "step" = one full frame through the hot path (step s renders sweep frame s % 16).
Metric: Mrays/s, ray := one iteration of the reference's bounce loop (primary + secondary).

N > 1 (launched by torch.distributed.run, one process per GPU): each frame is cut into 8-row
strips dealt round-robin over the ranks (strong scaling: the frame is fixed), every rank
renders its strips into a compact buffer, one RCCL gather per frame moves them to rank 0,
which scatters them into the image (double-buffered: gather and assembly of frame k overlap
the render of frame k+1).  value = rays of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the host driver of this pool only supports dmabuf IPC: without this RCCL's cross-process buffer
# sharing fails with hipIpcGetMemHandle (already exported by the image; kept for bare shells)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

import sdf_playground_amd as sp

SCENE = "labyrinth"
WIDTH, HEIGHT = 3840, 2160
ITER_COUNT = 256
SPLIT_PERIOD = 16     # strips per period of the private/shared split (N > 1)
SEED = 0x5DF00003
SWEEP = 16
PEAK_FP32_VECTOR_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters (spec)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md (spec)
CENSUS_FILE = os.path.join(ROOT, "profiles", "census_r01.json")
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_r01.json")


def pcg_hash(x):
    """The reference's PCG hash (noise.hlsl:6-11), uint32 wrap-around."""
    state = (x * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return ((word >> 22) ^ word) & 0xFFFFFFFF


def sweep_camera(k):
    """(eye, direction, stime) of sweep frame k."""
    u = pcg_hash((SEED + k) & 0xFFFFFFFF) / 4294967295.0
    t = 2.0 * math.pi * (k + u) / SWEEP
    return (1.5 * math.cos(t), 5.0, 1.5 * math.sin(t)), (math.cos(t), -0.35, math.sin(t)), k / 60.0


def make_camera(k, width=WIDTH, height=HEIGHT):
    eye, direction, stime = sweep_camera(k)
    cam = sp.Camera()
    cam.SetEye(eye)
    cam.SetDirection(direction)
    cam.SetFOVY(sp.to_radian(60.0))
    cam.SetAspect(float(np.float32(width) / np.float32(height)))
    return cam, stime


def load_census():
    """flops per ray of this workload, from the operation-counting oracle build (committed)."""
    try:
        with open(CENSUS_FILE) as fh:
            return json.load(fh)["labyrinth_4k_iter256"]
    except Exception:
        return None


def host_cores():
    """CPUs this process may really use: the scheduler affinity and the cgroup CPU quota both
    cap it (a GPU box reports every core of the host but grants a share of them)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),):
        try:
            quota, period = parse(open(path).read())
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


def cpu_baseline(max_seconds=20.0):
    """The CPU oracle (a port: the reference HLSL cannot run here) timed on the host cores on a
    bounded sample of the same workload: the 16 sweep frames (fewer on slow hosts: it stops after
    max_seconds), every 2nd pixel in x and y; threads = the CPUs the cgroup really grants."""
    from oracle import pyoracle as po

    cores = host_cores()
    step = 2
    frames = SWEEP
    rays = 0
    pixels = 0
    t_total = 0.0
    used = 0
    for k in range(frames):
        eye, direction, stime = sweep_camera(k)
        basis = po.camera_direction(eye, direction, np.float32(sp.to_radian(60.0)), np.float32(WIDTH) / np.float32(HEIGHT))
        f = po.default_frame(SCENE, WIDTH, HEIGHT, basis=basis, stime=stime)
        f.iter_count = ITER_COUNT
        t0 = time.perf_counter()
        _, _, tot = po.render(SCENE, f, step=(step, step), nthreads=cores)
        t_total += time.perf_counter() - t0
        pixels += int(tot[0])
        rays += int(tot[1])
        used += 1
        if t_total > max_seconds:
            break
    return {
        "value": rays / t_total / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": "oracle (scalar C++ restatement, g++ -O2 -ffp-contract=off), sweep frames 0..%d of the same 3840x2160 workload, every %dth pixel in x and y (%d pixels, %d rays, %.1f s)" % (used - 1, step, pixels, rays, t_total),
        "ms_per_frame_equiv": t_total / used * step * step * 1e3,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--schedule", default=os.environ.get("SDFR_SCHEDULE", "auto"), choices=["auto", "wavefront", "pixel"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--force-distributed", action="store_true", help="take the strips + gather path even with one rank (testing)")
    ap.add_argument("--verify", action="store_true", help="after timing, check rank 0's assembled image of the last frame against a direct render")
    ap.add_argument("--private-strips", default="auto",
                    help="N > 1: of every 16 strips, how many rank 0 renders privately (its pixels do not travel); 'auto' = chosen from the "
                         "render and gather times measured during start-up")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or a.force_distributed
    saved_stdout = None
    if distributed:
        import torch.distributed as dist

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
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    W, H = a.width, a.height

    r = sp.SDFRenderer(local_rank)
    r.initShader(SCENE)
    r.setLimits(iter_count=ITER_COUNT)
    schedule = {"auto": sp.SCHEDULE_PIXEL, "wavefront": sp.SCHEDULE_WAVEFRONT, "pixel": sp.SCHEDULE_PIXEL}[a.schedule]
    r.setSchedule(schedule)
    stream = torch.cuda.current_stream()
    r.setStream(stream.cuda_stream)

    if distributed:
        # Two frames in flight.  Frames alternate between two renderer handles on two streams, so
        # the tail of frame k (a handful of waves still marching 256-step rays: one wave alone needs
        # ~0.2 ms, as long as a rank's whole share of the frame at 8 GPUs) overlaps the head of frame
        # k+1; gather + assembly of frame k overlap the render of frame k+1 as well.
        rs = [stream, torch.cuda.Stream()]
        rr = [r, sp.SDFRenderer(local_rank)]
        rr[1].initShader(SCENE)
        rr[1].setLimits(iter_count=ITER_COUNT)
        rr[1].setSchedule(schedule)
        rr[1].setStream(rs[1].cuda_stream)
        side = torch.cuda.Stream()
        if rank == 0:
            r_asm = sp.SDFRenderer(local_rank)  # a handle bound to the side stream, for the assembly kernel
            r_asm.setStream(side.cuda_stream)
            # private strips (see below) are rendered by handles of their own, so that counters do not mix
            r_priv = [sp.SDFRenderer(local_rank), sp.SDFRenderer(local_rank)]
            for b in range(2):
                r_priv[b].initShader(SCENE)
                r_priv[b].setLimits(iter_count=ITER_COUNT)
                r_priv[b].setSchedule(schedule)
                r_priv[b].setStream(rs[b].cuda_stream)
    # the assembled frames, double-buffered like everything else that two frames in flight share
    images = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2 if distributed else 1)]
    image = images[0]

    cameras = [make_camera(k, W, H) for k in range(SWEEP)]

    split = (0, SPLIT_PERIOD)
    calibration = None
    if distributed:
        # ---- how much of the frame should rank 0 keep for itself? ---------------------------------------
        # Its own pixels never cross a link, and for N > 1 the frame rate is bounded by what the peers
        # push through their single link each.  Measured here, once: a full-frame render (t_full), and a
        # gather + assembly of equal shares with nothing to render (t_gather).  With p = m / 16 of the
        # strips private: rank 0 renders t_full * (p + (1 - p) / N) and assembles, the links carry
        # t_gather * (1 - p); the frame time is the larger one (they overlap: two frames in flight).
        def make_buffers(sp_split):
            nb = sp.strip_buffer_bytes(W, H, world, sp.STRIP_RGB32F_A8, sp_split)
            loc = [torch.empty((nb,), dtype=torch.uint8, device="cuda") for _ in range(2)]
            gat = [torch.empty((world, nb), dtype=torch.uint8, device="cuda") for _ in range(2)] if rank == 0 else None
            return loc, gat, ([list(g.unbind(0)) for g in gat] if rank == 0 else None)

        local, gathered_flat, gather_lists = make_buffers(split)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for k in range(2):
            r.setParameters(cameras[k][1]); r.setCamera(cameras[k][0]); r.render(None, W, H, out=image)
        e0.record(stream)
        for k in range(4):
            r.setParameters(cameras[k][1]); r.setCamera(cameras[k][0]); r.render(None, W, H, out=image)
        e1.record(stream)
        torch.cuda.synchronize()
        t_full = e0.elapsed_time(e1) / 4

        def gather_once(b):
            with torch.cuda.stream(side):  # the collective orders itself after, and the assembly behind, this stream
                if rank == 0:
                    dist.gather(local[b], gather_list=gather_lists[b], dst=0)
                    r_asm.assembleStrips(W, H, world, gathered_flat[b], images[b], fmt=sp.STRIP_RGB32F_A8)
                else:
                    dist.gather(local[b], dst=0)

        for b in range(2):
            local[b].zero_()
        torch.cuda.synchronize()
        for b in range(2):
            gather_once(b)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0g = time.perf_counter()
        for i in range(6):
            gather_once(i & 1)
        torch.cuda.synchronize()
        t_gather = (time.perf_counter() - t0g) / 6 * 1e3
        choice = torch.zeros(1, dtype=torch.int32, device="cuda")
        if rank == 0:
            if a.private_strips != "auto":
                m_best = max(0, min(SPLIT_PERIOD - 1, int(a.private_strips)))
            else:
                m_best, t_best = 0, None
                for m in range(0, SPLIT_PERIOD - 1):
                    p_ = m / SPLIT_PERIOD
                    t_m = max(t_full * (p_ + (1 - p_) / world), t_gather * (1 - p_))
                    if t_best is None or t_m < t_best * 0.97:   # prefer the smaller split unless it clearly pays
                        m_best, t_best = m, t_m
            choice[0] = m_best
        dist.broadcast(choice, src=0)
        split = (int(choice.item()), SPLIT_PERIOD)
        calibration = {"t_full_ms": t_full, "t_gather_ms": t_gather, "private_strips_of_16": split[0]}
        for h_ in rr + ([r_asm] + r_priv if rank == 0 else []):
            h_.setStripSplit(*split)
        local, gathered_flat, gather_lists = make_buffers(split)
        works = [None, None]
        asm_done = [torch.cuda.Event(), torch.cuda.Event()]
        asm_used = [False, False]

    def step(s):
        """Enqueue frame s; returns the handles that render it (their counters add up to this rank's share)."""
        cam, stime = cameras[s % SWEEP]
        if not distributed:
            r.setParameters(stime)
            r.setCamera(cam)
            r.render(None, W, H, out=image)
            return [r]
        b = s & 1
        h = rr[b]
        h.setParameters(stime)
        h.setCamera(cam)
        used = [h]
        with torch.cuda.stream(rs[b]):
            if works[b] is not None:
                works[b].wait()                    # this frame's stream: local[b] is free once gather s-2 is done
            if rank == 0 and asm_used[b]:
                rs[b].wait_event(asm_done[b])      # gathered_flat[b] is free once assembly s-2 is done
            h.renderStrips(W, H, rank, world, local[b], fmt=sp.STRIP_RGB32F_A8)
            if rank == 0:
                works[b] = dist.gather(local[b], gather_list=gather_lists[b], dst=0, async_op=True)
                if split[0] > 0:  # after the gather was issued: the private strips render while the peers' strips travel
                    hp = r_priv[b]
                    hp.setParameters(stime)
                    hp.setCamera(cam)
                    hp.renderPrivateStrips(W, H, images[b])
                    used.append(hp)
                with torch.cuda.stream(side):
                    works[b].wait()
                    r_asm.assembleStrips(W, H, world, gathered_flat[b], images[b], fmt=sp.STRIP_RGB32F_A8)
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

    for s in range(a.warmup):
        step(s)
    fence()
    # rays are a property of the frames, independent of timing: count them outside the timed region
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for s in range(a.steps):
        es = rs[s & 1] if distributed else stream
        ev[s][0].record(es)
        step(s)
        ev[s][1].record(es)
    enqueue_ms = (time.perf_counter() - t0) / max(1, a.steps) * 1e3  # host time to enqueue a step
    fence()
    elapsed = time.perf_counter() - t0

    # per-frame ray counts of this rank (exact, from the kernels' counters), outside the timing
    rays_per_frame = []
    kernel_ms = []
    for k in range(min(SWEEP, a.steps)):
        sts = [h_.getStats() for h_ in step(k)]
        rays_per_frame.append(sum(int(st.rays) for st in sts))
        kernel_ms.append(sum(st.ms_gpu for st in sts))
    my_rays = sum(rays_per_frame[s % len(rays_per_frame)] for s in range(a.steps))
    step_ms = [e0.elapsed_time(e1) for e0, e1 in ev]

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    rays_t = torch.tensor([my_rays], dtype=torch.float64, device="cuda")
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays_t, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_rays = float(rays_t.item())

    verified = None
    if a.verify and rank == 0:
        last = a.steps - 1
        step(last)
        torch.cuda.synchronize()
        got = images[last & 1 if distributed else 0].clone()
        cam, stime = make_camera(last % SWEEP, W, H)
        r.setParameters(stime)
        r.setCamera(cam)
        ref = torch.empty_like(got)
        r.render(None, W, H, out=ref)
        torch.cuda.synchronize()
        verified = bool(torch.equal(got.view(torch.int32), ref.view(torch.int32)))
    elif a.verify and distributed:
        step(a.steps - 1)  # every rank takes part in the extra gather
        torch.cuda.synchronize()

    if rank == 0:
        census = load_census()
        out = {
            "metric": "Mrays/s (primary+secondary), labyrinth scene, 3840x2160",
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
                "workload": "labyrinth %dx%d, iter_count %d, reference cost rules (max_cost 7, hard shadows), 16-frame fixed-seed camera sweep (seed 0x5DF00003), default variables" % (W, H, ITER_COUNT),
                "schedule": "pixel" if schedule == sp.SCHEDULE_PIXEL else "wavefront",
                "parallelism": "strips%d" % world if distributed else "single",
                "strip_calibration": calibration,
                "rays_per_pixel": total_rays / a.steps / (W * H),
            },
        }
        # roofline of the dominant kernel (the only kernel of the pixel schedule), N = 1 geometry
        # kernel duration: HIP events on the launch stream.  N = 1: the pairs recorded around every
        # step of the timed region (a step launches k_pixel + the 12-us counter fold and nothing
        # else); N > 1: a step also gathers, so the handle's own events of the counting pass are used
        stats_pass_ms = float(np.mean(kernel_ms))
        mean_kernel_ms = stats_pass_ms if distributed else float(np.mean(step_ms))
        mean_rays = float(np.mean(rays_per_frame))
        if census:
            flops_per_launch = census["flops_per_ray"] * mean_rays
            achieved = flops_per_launch / (mean_kernel_ms * 1e-3) / 1e12
            traffic = None
            pmc = None
            try:
                with open(PMC_FILE) as fh:
                    pj = json.load(fh)
                traffic = pj.get("hbm_bytes_per_launch")
                for name, k in pj.get("kernels", {}).items():
                    if "k_pixel" in name:
                        # executed work (rocprofv3 --pmc over this same command, profiles/pmc_r01.json): the
                        # census numerator above is algorithmic and includes evaluations the kernel culls
                        pmc = {"valu_wave_instructions_per_launch": k["SQ_INSTS_VALU"], "valu_lane_utilization": k["valu_lane_utilization"],
                               "cycles_per_valu_instruction_per_simd": k["cycles_per_valu_inst_per_simd"], "waves_per_simd": k["avg_waves_per_simd"]}
            except Exception:
                pass
            out["roofline"] = {
                "bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_VECTOR_TFLOPS, "traffic": traffic,
                "kernel_ms": mean_kernel_ms, "kernel_ms_counting_pass": stats_pass_ms, "flops_per_ray": census["flops_per_ray"], "pmc": pmc,
                "note": "FP32 vector (VALU) issue bounds this path, not HBM or MFMA (SURVEY.md 8d); flops = oracle operation census",
            }
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
        if not distributed:
            # Not the headline: the same sweep with TWO frames in flight (two handles on two streams),
            # measured after the timed region.  It shows how much of `ms_per_step` is the tail of a
            # frame (the last few waves march their 256-step rays alone); `value` and `roofline` above
            # stay on one frame in flight, where a kernel's duration is its own.
            try:
                s2 = torch.cuda.Stream()
                r2 = sp.SDFRenderer(local_rank)
                r2.initShader(SCENE)
                r2.setLimits(iter_count=ITER_COUNT)
                r2.setSchedule(schedule)
                r2.setStream(s2.cuda_stream)
                img2 = torch.empty_like(image)
                pair = [(r, image), (r2, img2)]

                def step2(k):
                    h_, im_ = pair[k & 1]
                    h_.setParameters(cameras[k % SWEEP][1])
                    h_.setCamera(cameras[k % SWEEP][0])
                    h_.render(None, W, H, out=im_)

                for k in range(4):
                    step2(k)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for k in range(a.steps):
                    step2(k)
                torch.cuda.synchronize()
                ms2 = (time.perf_counter() - t2) / max(1, a.steps) * 1e3
                out["two_frames_in_flight"] = {"ms_per_step": ms2, "value": total_rays / a.steps / (ms2 * 1e-3) / 1e6, "unit": "Mrays/s",
                                               "note": "informational: same frames, two streams; not the headline"}
                r2.close()
            except Exception as e:
                out["two_frames_in_flight"] = {"error": repr(e)}
        if not distributed and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the bench line must still be printed
                out["cpu_baseline"] = {"error": repr(e)}
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)

    r.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
