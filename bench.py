#!/usr/bin/env python3
"""Headline benchmark: glass_monkey (tests/mesh_test.rs:96-134 of the reference) at
1920x1080, anti-aliasing on, on N MI355X — Mrays/s and ms/frame.

  python bench.py --gpus N --steps K --warmup W [--config NAME]

One step = one full frame through the HIP render path (pass 1 over all pixels, AA detect
+ resolve, clamp/gamma/encode) and, for N > 1, the gather of the RGB8 row blocks to
rank 0.  The scene is already resident in HBM when the timed region starts.  Rank 0
prints one JSON line.

With --gpus N > 1 and no WORLD_SIZE in the environment this process only starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child (before anything
here has touched the GPU), relays its output and exits with its status.

ray        = one closest-hit query that passed the depth/contribution gate, or one any-hit
             shadow query (counted on device by an instrumented, untimed frame)
roofline   = vector ALU: algorithmic flops (fixed cost per counted unit of work, see fill_counters
             in csrc/rt_api.hip; the counted units are in `work`) of the longer of the two trace
             passes / its HIP-event duration.  The flops are split by the arithmetic they ran in:
             f64 against the FP64 vector peak, and the box tests of the
             LDS-resident BVH walk, which run as packed f32, against the FP32 vector peak (`f32`).
             `peak` is the datasheet figure (78.6 TFLOP/s FP64 with FMA; the path is built with
             -ffp-contract=off like the reference's arithmetic, so what it can issue is
             `peak_no_fma`); `peak_measured` is what tools/peaks.hip reached on an MI355X
             (profiles/r03_peaks.json).  A pass is one launch sequence of the queue pipeline
             (analytic / BVH walk / settle / shade per ray generation); the events bracket the whole
             sequence on the launch stream.
             `traffic` = HBM bytes of that pass from rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this
             workload, read from the committed summary profiles/hbm_traffic.json - only while the
             kernel sources are the ones the summary was taken with (`kernels` hash), else null
cpu_baseline = the CPU restatement of the reference (oracle/, 8 threads like the
             reference's default pool, and all host cores) on a bounded sample of the same frame
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# name -> (scene, width, height, config overrides, CPU sample: every n-th row block); the first is BASELINE.json's metric configuration
CONFIGS = {
    "glass_monkey": ("glass_monkey", 1920, 1080, {}, 3),
    "light_room": ("light_room", 1920, 1080, {}, 2),
    "many_balls_forest": ("many_balls_forest", 3840, 2160, {}, 4),
    "subtraction_outside": ("subtraction_outside1", 3840, 2160, {}, 2),
    "subtraction_outside_ss": ("subtraction_outside1", 3840, 2160, {"aa_threshold": 0.0}, 3),   # supersampled: every interior hit pixel gets aa_rays more
}
TILE_ROWS = 8     # rows per block of the multi-GPU split (measured on 1080p: 8 balances the ranks best; 4 pays too much halo)
FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X datasheet, FP64 vector with FMA (half the 157.3 TF FP32 vector rate)
FP64_VECTOR_PEAK_NO_FMA_TFLOPS = 39.3   # one flop per lane and cycle: what un-fused arithmetic can issue
FP32_VECTOR_PEAK_TFLOPS = 157.3     # packed FP32 FMA
HBM_PEAK_GBS = 8000.0


def measured_peaks():
    """What tools/peaks.hip measured on an MI355X (committed under profiles/), or {}."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r03_peaks.json")))
    except (OSError, ValueError):
        return {}


def kernel_sources_hash():
    """Identifies the device code: sha256 over the kernel sources (csrc/*.h, *.hpp, *.hip and the variants)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "rust-raytracer_amd", "csrc")
    names = sorted(f for f in os.listdir(csrc) if f.endswith((".h", ".hpp", ".hip")))
    names += sorted(os.path.join("variants", f) for f in os.listdir(os.path.join(csrc, "variants")) if f.endswith(".hip"))
    for n in names:
        h.update(n.encode() + b"\0" + open(os.path.join(csrc, n), "rb").read())
    return h.hexdigest()[:16]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="glass_monkey", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def launch_ranks(args):
    """--gpus N without a launcher: start N ranks as a child job.  Nothing in this process has touched the GPU."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def host_cores():
    """Cores this process may use: its affinity mask, capped by the job's thread limit when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    limit = os.environ.get("OMP_NUM_THREADS", "")
    return max(1, min(n, int(limit)) if limit.isdigit() and int(limit) > 0 else n)


def cpu_baseline(oracle, scene, dim, cam, cfg, label, every):
    """Time the CPU oracle on the same frame, at the reference's default pool size (8 threads, src/render.rs:20) and,
    when the host gives this job more, on all of its cores: three runs per pool size, each over every
    (3 * `every`)-th row block starting at a different block, the median of the three reported."""
    cores = host_cores()
    rows = []
    for threads in sorted({8, cores}):
        runs = []
        for k in range(3):
            c = cfg.copy()
            c.num_threads = threads
            c.rank, c.nranks, c.tile_rows = 1 + k * every, 3 * every, TILE_ROWS
            t0 = time.time()
            out = oracle.render(scene, dim, cam, c, want_aux=True, faithful_cost=True)
            dt = time.time() - t0
            rays = out["counters"]["rays_total"]
            runs.append({"Mrays/s": rays / dt / 1e6, "rays": rays, "seconds": dt})
        med = sorted(runs, key=lambda r: r["Mrays/s"])[1]
        rows.append({"threads": threads, "Mrays/s": med["Mrays/s"], "rays": med["rays"], "seconds": med["seconds"],
                     "all": [round(r["Mrays/s"], 3) for r in runs]})
    first = rows[0]
    return {"value": first["Mrays/s"], "unit": "Mrays/s", "cores": first["threads"], "kind": "port",
            "host_cores": cores, "machine_cpus": os.cpu_count(), "runs": rows,
            "sample": f"{label}: median of three runs, each over every {3 * every}th {TILE_ROWS}-row block (+1-row AA halo) from a different "
                      f"first block ({first['rays']} rays in {first['seconds']:.1f} s in the median run); "
                      f"C++ restatement of the reference CPU path incl. its per-hit 4x4 inverse, {first['threads']} threads "
                      f"(reference default), not the Rust binary"}


def measured_traffic(config, n_gpus):
    """HBM bytes per frame from the committed rocprofv3 PMC summary of this command (1 GPU), or None."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if n_gpus != 1 or not os.path.exists(path):
        return None, None
    try:
        rec = json.load(open(path)).get(config)
    except (OSError, ValueError):
        return None, None
    if not rec:
        return None, None
    if rec.get("kernels") != kernel_sources_hash():
        return None, {"stale": True, "source": rec.get("source"), "kernels": rec.get("kernels")}   # the kernels have changed since the counters were read
    return rec.get("bytes_per_frame"), rec


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import torch
    import torch.distributed as dist
    scenes = importlib.import_module("harness.scenes")
    render = importlib.import_module("rust-raytracer_amd.render")
    rdist = importlib.import_module("rust-raytracer_amd.dist")
    abi = importlib.import_module("rust-raytracer_amd._abi")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the HIP render path has no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    scene_name, width, height, overrides, cpu_every = CONFIGS[args.config]
    scene = scenes.Scene(scene_name)
    dim, cam, cfg = scene.view(width=width, height=height, anti_alias=True, seed=1, **overrides)
    label = f"{scene_name} {width}x{height}, anti_alias on, aa_threshold {cfg.aa_threshold:g}, aa_rays {cfg.aa_rays}, recursion_depth {cfg.recursion_depth}, seed 1"
    ds = render.DeviceScene(scene, device=local_rank)
    stream = torch.cuda.current_stream(device)

    # --- instrumented, untimed frames: rays and algorithmic flops of this rank's share
    def counted(anti_alias):
        c = cfg.copy()
        c.anti_alias = 1 if anti_alias else 0
        c.rank, c.nranks, c.tile_rows = rank, world, TILE_ROWS
        cap = rdist.owned_rows_capacity(height, world, TILE_ROWS)
        buf = torch.zeros((cap, width, 3), dtype=torch.uint8, device=device)
        aux = abi.RtAux()
        aux.count_work = 1
        ds.render_device(dim, cam, c, buf.data_ptr(), stream.cuda_stream, aux)
        return aux.counters.as_dict()

    full = counted(True)
    p1 = counted(False)
    keys = ["rays_total", "rays_primary", "rays_secondary", "rays_shadow", "rays_aa", "aa_pixels", "instance_tests",
            "box_tests", "tri_tests", "shade_evals", "stack_overflows"]
    keys += ["box_tests_f32", "csg_events"]
    t = torch.tensor([float(full[k]) for k in keys] + [float(p1["flops"]), float(full["flops"] - p1["flops"]),
                                                        float(p1["flops_f32"]), float(full["flops_f32"] - p1["flops_f32"])],
                     dtype=torch.float64, device=device)
    local = t.clone()
    if world > 1:
        dist.all_reduce(t)
    work = {k: int(v) for k, v in zip(keys, t[:len(keys)].tolist())}
    rays_total = float(work["rays_total"])
    flops_pass1, flops_aa = float(local[-4].item()), float(local[-3].item())   # of this rank's share: the roofline line is rank 0's
    f32_pass1, f32_aa = float(local[-2].item()), float(local[-1].item())        # ... of which in f32
    if work["stack_overflows"] != 0:
        raise RuntimeError("secondary-ray stack overflowed: results would be incomplete")

    # --- timed region
    local_fn = rdist.hip_render_local(ds, dim, cam, cfg, check=False)

    def step():
        return rdist.render_frame(local_fn, width, height, TILE_ROWS, device)

    for _ in range(args.warmup):
        step()
    if ds.frame_dropped():
        step()                     # (the queues have been enlarged: one more untimed frame with the final sizes)
        if ds.frame_dropped():
            raise RuntimeError("warm-up frames keep overflowing the ray queues")
    stage_ms = {"pass1": 0.0, "aa": 0.0, "encode": 0.0}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # HIP events recorded by the library on the launch stream; reading them waits for this frame only
        tm = ds.last_timings()
        for k in stage_ms:
            stage_ms[k] += tm[k]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    lost = ds.frame_dropped()
    flags = torch.tensor([elapsed, float(lost)], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(flags, op=dist.ReduceOp.MAX)
    elapsed = float(flags[0].item())
    if flags[1].item() != 0:
        raise RuntimeError("a timed frame lost rays to a full queue: the measurement is invalid")

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        for k in stage_ms:
            stage_ms[k] /= args.steps
        # dominant launch sequence of rank 0's share
        launches = ("k_wf_analytic + k_wf_walk + k_wf_settle + k_wf_shade per generation" if full.get("schedule") == 1
                    else "k_trace, persistent lanes")
        if stage_ms["aa"] >= stage_ms["pass1"]:
            which, kernel, k_ms, k_flops, k_f32 = "aa", f"AA-resolve pass ({launches})", stage_ms["aa"], flops_aa, f32_aa
        else:
            which, kernel, k_ms, k_flops, k_f32 = "pass1", f"pass 1 ({launches})", stage_ms["pass1"], flops_pass1, f32_pass1
        k_f64 = k_flops - k_f32
        per_s = 1.0 / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        achieved, achieved_f32 = k_f64 * per_s, k_f32 * per_s
        peaks = measured_peaks()
        fb_bytes = width * height * (32 + 3)   # f64 colour+distance written once, RGB8 written once
        traffic, traffic_rec = measured_traffic(args.config, world)
        roofline = {"bound": "valu_fp64", "kernel": kernel, "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS,
                    "peak_no_fma": FP64_VECTOR_PEAK_NO_FMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / FP64_VECTOR_PEAK_TFLOPS, "frac_no_fma": achieved / FP64_VECTOR_PEAK_NO_FMA_TFLOPS,
                    "f32": {"achieved": achieved_f32, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved_f32 / FP32_VECTOR_PEAK_TFLOPS,
                            "what": "box tests of the LDS-resident BVH walk (packed f32 FMA, v_min3 / v_max3), 24 flop each"},
                    # f64 and f32 work share the same SIMDs and the same time: the two fractions add up
                    "frac_f64_plus_f32": achieved / FP64_VECTOR_PEAK_TFLOPS + achieved_f32 / FP32_VECTOR_PEAK_TFLOPS,
                    "traffic": None if not traffic_rec or traffic_rec.get("stale") else traffic_rec.get("bytes_" + which),
                    "traffic_per_frame": traffic,
                    "algorithmic_flops_per_launch": int(k_flops), "algorithmic_flops_f64": int(k_f64), "algorithmic_flops_f32": int(k_f32),
                    "avg_launch_ms": k_ms, "kernels": kernel_sources_hash(),
                    "hbm_secondary": {"algorithmic_bytes_per_frame": fb_bytes, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "achieved_algorithmic": fb_bytes / (ms_per_step * 1e-3) / 1e9}}
        if peaks:
            roofline["peak_measured"] = peaks.get("fp64_fma_tflops")
            roofline["peak_measured_no_fma"] = peaks.get("fp64_mul_add_tflops")
            roofline["frac_measured"] = achieved / peaks["fp64_fma_tflops"] if peaks.get("fp64_fma_tflops") else None
            roofline["f32"]["peak_measured"] = peaks.get("fp32_packed_fma_tflops")
            roofline["hbm_secondary"]["peak_measured"] = peaks.get("hbm_read_gbs")
            roofline["peaks_source"] = "profiles/r03_peaks.json (tools/peaks.hip on one MI355X)"
        if traffic_rec and traffic_rec.get("stale"):
            roofline["traffic_stale"] = "profiles/hbm_traffic.json was taken with other kernels (%s)" % traffic_rec.get("kernels")
        if traffic:
            roofline["hbm_secondary"]["achieved_measured"] = traffic / (ms_per_step * 1e-3) / 1e9
            roofline["hbm_secondary"]["traffic_source"] = traffic_rec.get("source")
        out = {
            "metric": f"Mrays/s, {scene_name} {width}x{height} AA on (ms/frame in ms_per_step)",
            "value": rays_total / (elapsed / args.steps) / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": label, "rays_per_frame": int(rays_total), "tile_rows": TILE_ROWS,
                       "parallelism": f"row-block interleave x{world}, RGB8 gather to rank 0"},
            "work": work,
            "stage_ms_rank0": stage_ms,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            from oracle import binding as oracle
            out["cpu_baseline"] = cpu_baseline(oracle, scene, dim, cam, cfg, label, cpu_every)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
