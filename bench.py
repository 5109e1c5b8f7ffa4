#!/usr/bin/env python3
"""Throughput of the render hot path on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one render of the random-spheres scene (488 spheres, seed 1) at 1920x1080, 100 samples per
pixel per GPU, depth 50: rank g traces samples [100g, 100(g+1)) of every pixel, the float sum images
are reduced to rank 0 over RCCL, rank 0 applies mean/gamma/quantise.  Prints ONE JSON line on rank 0.

    python bench.py --config c3 [--gpus N] [--spp S]

is the 8-GPU configuration of the same scene as a STRONG-scaling run: 3840x2160, S (default 1000) samples
per pixel in total, rank g traces sample_shard(S, g, N); same JSON line with "scaling": "strong".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

WIDTH, HEIGHT, SPP_PER_GPU, DEPTH, SCENE_SEED, SEED = 1920, 1080, 100, 50, 1, 0
C3_WIDTH, C3_HEIGHT, C3_SPP_TOTAL = 3840, 2160, 1000


def cpu_baseline(hittables, budget_s):
    """The CPU oracle in the reference's own configuration (pointer tree, rejection sampling, f64) on the
    host cores, over a bounded sample of the same workload: a band of rows at reduced spp."""
    import oracle_binding as ob

    threads = min(ob.lib().orc_max_threads(), int(os.environ.get("OMP_NUM_THREADS", "16")), os.cpu_count() or 1)
    scene = ob.Scene(hittables)
    scene.build_tree()
    cam = ob.camera_build([0, 0, 0], [13, 1.5, 3], 1.0, 0.1, WIDTH, HEIGHT)
    def run(stride, spp, th):
        _, _, st = scene.render(cam, spp=spp, depth=DEPTH, seed=SEED, traverse=ob.TRAVERSE_TREE, sampler=ob.SAMPLER_REFERENCE,
                                threads=th, rows=(stride // 2, HEIGHT), row_stride=stride)
        return st

    # calibrate on every 64th row, then size the sample (rows spread evenly over the frame, so sky and
    # ground are both in it) for the time budget
    cal = run(16, 1, threads)
    rate = cal["paths"] / max(cal["seconds"], 1e-6)
    want_rows = max(16.0, rate * budget_s / WIDTH)
    spp = int(max(1, min(64, want_rows // HEIGHT)))
    stride = int(max(1, HEIGHT // max(1.0, want_rows / spp)))
    st = run(stride, spp, threads)
    paths, segs, secs, n_rows = st["paths"], st["segments"], st["seconds"], st["paths"] // (WIDTH * spp)
    # one thread, as the reference runs (src/lib.rs:166-168), on a sample `threads` times smaller
    st1 = run(min(HEIGHT // 4, stride * threads), spp, 1)
    p1, t1 = st1["paths"], st1["seconds"]
    return {
        "value": round(paths / secs / 1e6, 4), "unit": "Mpaths/s", "cores": threads, "kind": "port",
        "sample": "%d rows of %dx%d spread over the frame at %d spp, depth %d: %d paths in %.1f s; reference-style tree + rejection sampling, f64"
                  % (n_rows, WIDTH, HEIGHT, spp, DEPTH, paths, secs),
        "single_thread_value": round(p1 / max(t1, 1e-9) / 1e6, 4),
        "segments_per_path": round(segs / max(paths, 1), 4),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--kernel-flags", type=int, default=0, help="RT_RENDER_KERNEL_* measurement aids (A/B runs)")
    ap.add_argument("--config", default="c2", choices=["c2", "c3"],
                    help="c2: 1920x1080, 100 spp per GPU (weak scaling, the headline); c3: 3840x2160, 1000 spp in total split over the GPUs (strong scaling)")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel: per GPU for c2, in total for c3 (0 = the configuration's own)")
    args = ap.parse_args()
    global WIDTH, HEIGHT
    strong = args.config == "c3"
    if strong:
        WIDTH, HEIGHT = C3_WIDTH, C3_HEIGHT

    import torch
    import torch.distributed as dist

    import raytracer_amd as rt
    from raytracer_amd import roofline

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the render path has no CPU fallback")
    # BENCH_REHEARSE=1: all ranks share the visible GPUs and reduce over gloo through host memory — a functional
    # rehearsal of the N > 1 path on a box with fewer GPUs than ranks (not a measurement)
    rehearse = os.environ.get("BENCH_REHEARSE") == "1"
    if rehearse:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1 and rehearse:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    hittables = rt.gen_scene(SCENE_SEED)
    scene = rt.Scene(hittables, device=local)
    cam = rt.default_camera(WIDTH, HEIGHT)
    n_pixels = WIDTH * HEIGHT
    d_sum = torch.zeros(HEIGHT, WIDTH, 3, dtype=torch.float32, device="cuda")
    d_rgb = torch.zeros(HEIGHT, WIDTH, 3, dtype=torch.uint8, device="cuda")
    total_spp = (args.spp or C3_SPP_TOTAL) if strong else (args.spp or SPP_PER_GPU) * world
    if total_spp < world:
        raise SystemExit("bench.py: %d samples for %d GPUs" % (total_spp, world))
    begin, end = rt.sample_shard(total_spp, rank, world)
    kernel_ms = []

    scene.reserve(cam, end - begin, sample_begin=begin)  # partial-sum buffer sized once: the timed calls only enqueue

    def step(timed):
        """raytracer_amd.shard.render_step with the GPU pieces plugged in (the CPU tests plug in the oracle)."""
        stream = torch.cuda.current_stream().cuda_stream
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def render_range(b, e):
            ev0.record()  # torch's current stream is the one the kernel is launched on
            scene.render_device(cam, d_sum.data_ptr(), b, e, DEPTH, SEED, stream=stream, flags=args.kernel_flags)
            ev1.record()
            return d_sum

        def reduce_to_root(t):
            if rehearse:
                host = t.cpu()
                dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                t.copy_(host)
            else:
                dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)

        def finalize(t, spp):
            scene.finalize_device(t.data_ptr(), n_pixels, spp, d_rgb.data_ptr(), stream=stream)
            return d_rgb

        rt.render_step(render_range, reduce_to_root, finalize, total_spp, rank, world)
        if timed:
            kernel_ms.append((ev0, ev1))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    render_ms = sum(a.elapsed_time(b) for a, b in kernel_ms) / max(len(kernel_ms), 1)

    if rank == 0:
        paths_per_step = n_pixels * total_spp
        value = paths_per_step * args.steps / elapsed / 1e6
        # segments per path, counted by the device that is being rated: one launch of the instrumented build of the
        # same kernel, outside the timed region (every shade visit of a live path closes one segment)
        from raytracer_amd import _abi
        import numpy as np
        scene.render_device(cam, d_sum.data_ptr(), begin, end, DEPTH, SEED, stream=torch.cuda.current_stream().cuda_stream,
                            flags=_abi.RT_RENDER_STATS)
        torch.cuda.synchronize()
        st = np.zeros(8, dtype=np.uint64)
        _abi.check(_abi.lib.rt_debug_render_stats(scene.handle, st.ctypes.data))
        seg = round((float(st[5]) - float(st[6])) / (n_pixels * (end - begin)), 4)
        base = cpu_baseline(hittables, args.cpu_seconds) if (args.cpu_seconds > 0 and world == 1) else None
        fpp = roofline.flops_per_path(len(hittables), seg)
        # HBM bytes per launch come from the committed counter passes over exactly this command line (one GPU, the
        # headline configuration, no measurement flags); any other run reports null
        profiled_run = world == 1 and not strong and not args.spp and args.kernel_flags == 0
        traffic = roofline.measured_traffic_bytes() if profiled_run else None
        achieved = (n_pixels * (end - begin)) * fpp / (render_ms * 1e-3) / 1e12 if render_ms > 0 else 0.0
        img_mean = float(d_rgb.float().mean().item())
        line = {
            "metric": "Mpaths/sec at %dx%d random-spheres" % (WIDTH, HEIGHT), "value": round(value, 3), "unit": "Mpaths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not rehearse else "synthetic; REHEARSAL on shared GPUs over gloo, not a measurement",
            "config": {"workload": "random-spheres 488 spheres (seed 1), %dx%d, %s, depth %d"
                                   % (WIDTH, HEIGHT, ("%d spp in total split over the GPUs" % total_spp) if strong
                                      else "%d spp per GPU (%d total)" % (total_spp // world, total_spp), DEPTH),
                       "paths_per_step": paths_per_step, "parallelism": "sample-split x%d + RCCL sum-reduce" % world,
                       "image_mean_u8": round(img_mean, 3)},
            "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": roofline.PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / roofline.PEAK_FP32_TFLOPS, 5),
                         "traffic": traffic["bytes_per_launch"] if traffic else None,
                         "traffic_source": traffic["source"] if traffic else "not collected for this command line (the committed counter passes cover `bench.py` on one GPU)",
                         "kernel": "render_kernel_ws<true, 256, 0, true, true, 2> (+ combine_kernel, 0.09 ms)", "kernel_ms": round(render_ms, 3),
                         "kernel_ms_source": "HIP events on the launching stream around rt_render_device, mean of the timed steps",
                         "algorithmic_flops_per_path": round(fpp, 1), "segments_per_path": seg,
                         "segments_per_path_source": "device: RT_RENDER_STATS launch of the same kernel outside the timed region"
                                                     + ("; CPU oracle counts %.4f" % base["segments_per_path"] if base else ""),
                         "note": "FP32 vector-ALU bound (BVH and spheres are cache resident, no dense contraction so no MFMA); "
                                 "algorithmic flops = reference arithmetic with a 2*ceil(log2 N)-box traversal model"},
        }
        if base:
            line["cpu_baseline"] = base
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
