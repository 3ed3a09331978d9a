#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X.

A "step" is one full forward render of BASELINE config C2: Cornell box (mi.cornell_box(), util.py:565-699),
512x512 @ 256 spp, path integrator max_depth=6 / rr_depth=5, seed = step index: 67 108 864 camera paths through
PathIntegrator::sample + film splat + develop, scene already resident in HBM.  value = Msamples/s =
W*H*spp / wall-seconds, whole job.  With N > 1 ranks the sample plane is cut into 16-row tiles dealt round-robin
to the ranks (strong scaling, total work fixed); the only exchange is one all-reduce (RCCL) of the 4 MB RGBW
accumulator before develop.

With `--gpus N` and no launcher environment (no WORLD_SIZE) the script starts `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1` on itself as a child process (before anything touches the GPU) and exits with its code.

The same JSON line carries
  roofline      the resource that binds the fused path kernel: wave64 VALU instruction issue.  achieved = VALU instructions per
                launch (SQ_INSTS_VALU of the committed PMC pass of the leg, profiles/r03_pmc[_veach|_ptracer|_lens]_summary.json,
                scaled by lanes) / HIP-event time of the kernel measured live; peak = 1024 SIMDs x 2.4 GHz / 2 cycles per
                instruction (MI355X_MICROARCH.md); `peak_at_sustained_clock` prices the same issue rate at the clock the part
                holds under an all-VALU load (profiles/r02_valu_issue.json: 2.32 nominal cycles per instruction, i.e. ~2.07 GHz).
                Each PMC pass records the sha256 of csrc/ it was taken from: when it differs from this checkout's the block says
                `pmc_stale` and states no fraction.  `traffic` and `hbm` come from the same pass.  `contract` keeps the
                SURVEY.md 8(d) algorithmic bytes (88 N_gen + 48 N_isect + 196 N_shade + 104 N_shadow + 20 N_splat + 60 H W) of
                the unfused SoA formulation -- a bookkeeping figure, not a physical bound.  Every leg below carries the same block
                for its dominant kernel (the lens leg two: render and geometry adjoint).
  cpu_baseline  the CPU oracle (scalar C restatement, OpenMP over all host cores) on a bounded sample of the same
                workload (512x512 at a few spp), rank 0, N = 1 only
  prb           PRB backward (BASELINE config C3: 512x512 @ 64 spp, gradient w.r.t. a 256x256x3 albedo texture on the
                back wall + all constant albedos / the emitter) in Msamples/s -- measured after the timed steps.
  veach         BASELINE configs[3]: Veach-MIS scene 1024^2 @ 512 spp forward
  ptracer       particle tracer, Cornell box 512^2 @ 64 spp, box and Gaussian reconstruction filters (+ its adjoint)
  lens          BASELINE configs[4] as notebooks/doublet_full_optim.ipynb cell 7 runs it: DesignProblem(num_sources=3), three sensors
                in one scene, per iteration 3 x (render @ 256 spp + geometry adjoint @ 128 spp) + surface update + Adam
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package, load_oracle  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy peak is ~6.3 TB/s
N_SIMD = 256 * 4            # 256 CUs x 4 SIMDs
PEAK_CLOCK_GHZ = 2.4        # MI355X peak engine clock (MI355X_MICROARCH.md)
VALU_CYCLES_PEAK = 2.0      # cycles one SIMD needs per wave64 VALU instruction (guide); profiles/r02_valu_issue.json measures 2.2-2.5
VALU_PEAK_GINST = N_SIMD * PEAK_CLOCK_GHZ / VALU_CYCLES_PEAK      # 1228.8 G wave64-instructions / s
# committed PMC passes (tools/pmc.sh + tools/pmc_summary.py), one per leg; each records the sha256 of the kernel sources it was taken from
PMC_FILE = os.path.join("profiles", "r03_pmc_summary.json")                 # headline (C2) + prb (C3)
PMC_FILES = {"veach": os.path.join("profiles", "r03_pmc_veach_summary.json"), "ptracer": os.path.join("profiles", "r03_pmc_ptracer_summary.json"),
             "lens": os.path.join("profiles", "r03_pmc_lens_summary.json")}
TILE_ROWS = 16


def csrc_sha256():
    """sha256 over the kernel sources of THIS checkout (same recipe as tools/pmc_summary.py): a PMC pass taken from other sources is stale."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "mitsuba3-camera_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


_PMC_CACHE = {}


def pmc_load(pmc_file):
    if pmc_file not in _PMC_CACHE:
        try:
            with open(os.path.join(ROOT, pmc_file)) as f:
                _PMC_CACHE[pmc_file] = json.load(f)
        except Exception:
            _PMC_CACHE[pmc_file] = None
    return _PMC_CACHE[pmc_file]


def pmc_kernel(substr, pmc_file=None):
    """Counters of one kernel from a committed PMC pass (None when the file or the kernel is missing)."""
    d = pmc_load(pmc_file or PMC_FILE)
    if d is None:
        return None
    for name, rec in d["kernels"].items():
        if substr in name and "counters" in rec:
            return rec
    return None


def pmc_stale(pmc_file=None):
    """True when the committed pass was taken from other kernel sources than the ones this run executes (or carries no hash)."""
    d = pmc_load(pmc_file or PMC_FILE)
    return d is None or d.get("csrc_sha256") != csrc_sha256()


def microbench_cycles(waves=5):
    try:
        with open(os.path.join(ROOT, "profiles", "r02_valu_issue.json")) as f:
            for r in json.load(f)["results"]:
                if r["op"] == "v_fma_f32" and r["waves_per_simd"] == waves:
                    return r["cyc_per_inst_wall"]      # wall time x nominal clock: the same convention as cycles_per_inst_measured (DVFS included)
    except Exception:
        pass
    return None


def microbench(op, waves=8):
    """cycles (wall time x nominal clock) per VALU instruction of one stream of profiles/r02_valu_issue.json"""
    try:
        with open(os.path.join(ROOT, "profiles", "r02_valu_issue.json")) as f:
            for r in json.load(f)["results"]:
                if r["op"].startswith(op) and r["waves_per_simd"] == waves:
                    return r["cyc_per_inst_wall"]
    except Exception:
        pass
    return None


def issue_model(c):
    """Issue model from the microbenchmark streams (profiles/r02_valu_issue.json, 8 waves per SIMD, wall-clock cycles at the nominal clock):
      c_v  a wave64 VALU instruction                                  (v_fma stream)
      c_t  a transcendental inside a VALU stream                       (7 v_fma + 1 v_rcp stream: 8 * mix - 7 * c_v)
      c_s  a scalar instruction when the scalar unit is the limit      (1 s_add per v_fma: the pair costs c_s -- the CU's one scalar ALU
           serves four SIMDs; with 2 or 4 v_fma per s_add the scalar work hides behind the vector work except for a small interaction)
    SIMD cycles of a launch = max(V * (1 + a * min(S / V, 1)), S) with V = VALU * c_v + TRANS * (c_t - c_v), S = (SALU + SMEM + BRANCH) * c_s
    and a fitted on the 2 : 1 stream.  VMEM / LDS instructions (separate issue ports, few) and every latency are NOT priced: the model is
    what a perfectly latency-hidden schedule of this instruction mix would take."""
    fma, pair, mix = microbench("v_fma_f32", 8), microbench("v_fma_f32 + s_add_u32", 8), microbench("7 v_fma_f32 + 1 v_rcp_f32", 8)
    m21 = microbench("2 v_fma_f32 + 1 s_add_u32", 8)
    if not (fma and pair and mix and m21) or "SQ_INSTS_VALU" not in c:
        return None
    c_v, c_s, c_t = fma, pair, 8.0 * mix - 7.0 * fma
    a = (m21 / c_v - 1.0) / ((c_s / 2.0) / c_v)           # overhead per unit of (scalar time / vector time), from the 2 : 1 stream
    n_trans = c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
    n_scalar = sum(c.get(k, 0.0) for k in ("SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"))
    V = c["SQ_INSTS_VALU"] * c_v + n_trans * (c_t - c_v)
    S = n_scalar * c_s
    cycles = max(V * (1.0 + a * min(S / V, 1.0)), S) / N_SIMD
    return {"valu": int(c["SQ_INSTS_VALU"]), "transcendentals": int(n_trans), "scalar": int(n_scalar),
            "cycles_per_valu": round(c_v, 3), "cycles_per_transcendental": round(c_t, 2), "cycles_per_scalar_when_limiting": round(c_s, 3), "interaction": round(a, 3),
            "scalar_over_vector_time": round(S / V, 3), "modelled_ms_at_nominal_clock": round(cycles / (PEAK_CLOCK_GHZ * 1e6), 3),
            "note": "latency-free schedule of this instruction mix per tools/valu_issue_bench.hip (profiles/r02_valu_issue.json): vector issue + transcendentals, scalar instructions hidden behind them up to the scalar unit's own rate"}


def valu_roofline(kernel, pmc_name, kernel_ms, lanes_per_launch, contract_bytes, pmc_file=None, lanes_profiled=None):
    """roofline block of a fused path kernel: bound = wave64 VALU issue.  Instruction count and HBM traffic per launch are read from
    the committed PMC pass of the leg (scaled by lanes when this launch covers a different number of lanes than the profiled one), the
    time is the HIP-event time measured in this run.  A pass taken from other kernel sources than this checkout's is reported as
    `pmc_stale` and no fraction is stated.  `lanes_profiled`: lanes the profiled launch covered when its grid is not one thread per lane
    (the regenerating kernels run a fixed-size persistent grid)."""
    pmc_file = pmc_file or PMC_FILE
    rec = pmc_kernel(pmc_name, pmc_file)
    sec = kernel_ms * 1e-3
    sustained = microbench("v_fma_f32", 8)                                   # wall-clock cycles (at the nominal clock) per wave64 VALU instruction the part sustains
    out = {"bound": "valu_issue", "achieved": None, "peak": round(VALU_PEAK_GINST, 1), "unit": "Ginst/s (wave64 VALU)", "frac": None, "traffic": None,
           "kernel": kernel, "kernel_ms": round(kernel_ms, 4),
           "peak_at_sustained_clock": round(VALU_PEAK_GINST * VALU_CYCLES_PEAK / sustained, 1) if sustained else None}
    if rec is not None and pmc_stale(pmc_file):
        out["pmc_stale"] = True
        out["note"] = "%s was taken from other kernel sources than this checkout (sha256 of csrc/ differs): instruction counts not trusted, frac not stated; re-run tools/pmc.sh" % pmc_file
        rec = None
    if rec is not None and sec > 0:
        c = rec["counters"]; lanes_profiled = float(lanes_profiled or rec["grid"]); scale = lanes_per_launch / lanes_profiled
        insts = c["SQ_INSTS_VALU"] * scale
        out.update({"achieved": round(insts / sec / 1e9, 1), "frac": round(insts / sec / 1e9 / VALU_PEAK_GINST, 4),
                    "frac_of_peak_at_sustained_clock": round(insts / sec / 1e9 / (VALU_PEAK_GINST * VALU_CYCLES_PEAK / sustained), 4) if sustained else None,
                    "insts_per_launch": int(insts), "cycles_per_inst_measured": round(sec * PEAK_CLOCK_GHZ * 1e9 * N_SIMD / insts, 3),
                    "cycles_per_inst_peak": VALU_CYCLES_PEAK, "cycles_per_inst_microbench": microbench_cycles(),
                    "pmc": {"file": pmc_file, "kernel": pmc_name, "lanes_profiled": int(lanes_profiled), "grid_threads": int(rec["grid"]), "scaled_by_lanes": round(scale, 6),
                            "SQ_INSTS_VALU": c["SQ_INSTS_VALU"], "GRBM_GUI_ACTIVE": c.get("GRBM_GUI_ACTIVE"),
                            "cycles_per_inst_pmc": (round(c["GRBM_GUI_ACTIVE"] / 8 * N_SIMD / c["SQ_INSTS_VALU"], 3) if c.get("GRBM_GUI_ACTIVE") else None),
                            "waves": c.get("SQ_WAVES"), "insts_per_wave": {k[9:].lower(): round(c[k] / c["SQ_WAVES"], 1) for k in
                                ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH") if k in c and c.get("SQ_WAVES")}}})
        im = issue_model(c)
        if im is not None:
            im["modelled_ms_at_nominal_clock"] = round(im["modelled_ms_at_nominal_clock"] * scale, 3)
            for key in ("valu", "transcendentals", "scalar"):
                im[key] = int(im[key] * scale)
            im["frac_of_model"] = round(im["modelled_ms_at_nominal_clock"] / kernel_ms, 4)
            out["issue_model"] = im
        if "hbm_write_bytes" in rec and "hbm_read_bytes" in rec:
            traffic = (rec["hbm_read_bytes"][1] + rec["hbm_write_bytes"]) * scale        # read side: the x2-corrected upper bound
            out["traffic"] = int(traffic)
            out["hbm"] = {"traffic_measured": int(traffic), "read_bytes": [int(v * scale) for v in rec["hbm_read_bytes"]], "write_bytes": int(rec["hbm_write_bytes"] * scale),
                          "achieved_GBs": round(traffic / sec / 1e9, 1), "peak_GBs": HBM_PEAK_GBS, "frac_of_peak": round(traffic / sec / 1e9 / HBM_PEAK_GBS, 4)}
    elif "note" not in out:
        out["note"] = "no PMC record for this kernel in %s: instruction count unknown, frac not stated" % pmc_file
    out["contract"] = {"algorithmic_bytes_per_launch": int(contract_bytes), "GBs": round(contract_bytes / sec / 1e9, 1) if sec > 0 else None,
                       "note": "SURVEY.md 8(d) bytes of the unfused SoA wavefront formulation / kernel time; the fused kernel keeps that state in registers, so this is bookkeeping, not a bound"}
    return out


def algorithmic_bytes(c, n_pixels_developed):
    return (88 * c["n_gen"] + 48 * c["n_isect"] + 196 * c["n_shade"] + 104 * c["n_shadow"] + 20 * c["n_splat"]
            + 60 * n_pixels_developed)


def leg_failed(name, exc):
    """A secondary leg (prb, veach, ptracer, lens) raised: the traceback goes to stderr, the JSON line carries {"error": ...} in the leg's
    place, and the headline measurement -- taken before any of these legs -- is printed as usual."""
    import traceback
    print("bench.py: leg %r failed" % name, file=sys.stderr); traceback.print_exc(file=sys.stderr)
    return {"error": "%s: %s" % (type(exc).__name__, exc)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--prb-spp", type=int, default=64)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-prb", action="store_true", help="skip the PRB backward leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--prb-plain", action="store_true", help="A/B: PRB leg without the albedo texture (constant reflectances only)")
    ap.add_argument("--ptracer", action="store_true", help="(legs run by default now; kept for old command lines)")
    ap.add_argument("--veach", action="store_true", help="(legs run by default now; kept for old command lines)")
    ap.add_argument("--no-veach", action="store_true", help="skip the BASELINE configs[3] leg (Veach-MIS scene, 1024^2 @ 512 spp)")
    ap.add_argument("--no-ptracer", action="store_true", help="skip the particle tracer leg (Cornell box 512^2 @ 64 spp)")
    ap.add_argument("--ptracer-direct", action="store_true", help="A/B: also time the particle tracer's Gaussian splat with plain atomics (230 ms)")
    ap.add_argument("--no-lens", action="store_true", help="skip the BASELINE configs[4] leg (lens doublet, ptracer + geometry adjoint)")
    ap.add_argument("--lens-subdiv", type=int, default=7, help="icosphere level of the lens surface meshes (the notebook uses 7)")
    ap.add_argument("--only", default="", help="comma-separated legs to run besides the headline one: prb,veach,ptracer,lens,cpu (default: all)")
    ap.add_argument("--no-ab", action="store_true", help="skip the A/B launches inside the legs (profiling runs: one variant per kernel name)")
    ap.add_argument("--repack-always", "--wavefront", dest="wavefront", action="store_true", help="A/B: MI355_MODE_REPACK_ALWAYS (re-pack the workgroup's live paths after every bounce; still one fused launch)")
    ap.add_argument("--static", action="store_true", help="A/B: MI355_MODE_STATIC (never re-pack)")
    ap.add_argument("--regen", action="store_true", help="A/B: path regeneration inside a wave (MI355_FLAG_REGEN)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    args = ap.parse_args()
    if args.only:
        want = set(args.only.split(","))
        args.no_prb = args.no_prb or "prb" not in want; args.no_veach = args.no_veach or "veach" not in want
        args.no_ptracer = args.no_ptracer or "ptracer" not in want; args.no_lens = args.no_lens or "lens" not in want; args.no_cpu = args.no_cpu or "cpu" not in want

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, as a torch.distributed.run child.
        # Decided BEFORE this process makes any GPU call (torch is not even imported yet): a process that has touched the GPU must never be
        # replaced or forked.  The parent only waits; rank 0's JSON line goes to the inherited stdout; the child's exit code is ours.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus (%d) does not match WORLD_SIZE (%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    mi = load_package()
    lib = mi.abi.lib()
    d = mi.cornell_box()
    d["sensor"]["film"]["width"] = args.res; d["sensor"]["film"]["height"] = args.res
    scene = mi.load_dict(d)
    integ = mi.PathIntegrator({"max_depth": 6, "rr_depth": 5})
    integ.timing = True
    integ.regen = args.regen
    if args.static:
        integ.mode = mi.abi.MODE_STATIC
    if args.wavefront:
        integ.mode = mi.abi.MODE_REPACK_ALWAYS
    h = scene.handle(local_rank)
    dev = torch.device("cuda", local_rank)
    W = H = args.res
    npix = W * H
    n_tiles = (H + TILE_ROWS - 1) // TILE_ROWS
    my_tiles = [t for t in range(n_tiles) if t % world == rank]
    my_shard = mi.shard_tiles(W, H, rank, world, rows=TILE_ROWS)
    # N > 1: two film accumulators, so that the reduce of frame k (RCCL runs it on its own stream) overlaps the kernel of
    # frame k + 1; a frame is complete when its accumulator has been reduced to rank 0 and developed there.
    films = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in range(2 if world > 1 else 1)]
    pending = [None] * len(films)
    img = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    frame_no = [0]

    def finish(b):
        if pending[b] is not None:
            pending[b].wait(); pending[b] = None                          # the compute stream waits for the collective (no host sync)
            if rank == 0:
                mi.abi.check(lib.mi355_film_develop(h, 0, C.c_void_p(films[b].data_ptr()), C.c_void_p(img.data_ptr()), stream))

    def step(seed):
        b = frame_no[0] % len(films); frame_no[0] += 1
        finish(b)
        film = films[b]
        film.zero_()
        if world == 1:
            p = integ._params(scene, seed, args.spp)
            mi.abi.check(lib.mi355_render_accumulate(h, C.byref(p), C.c_void_p(film.data_ptr()), stream))
            mi.abi.check(lib.mi355_film_develop(h, 0, C.c_void_p(film.data_ptr()), C.c_void_p(img.data_ptr()), stream))
        else:
            p = integ._params(scene, seed, args.spp, shard=my_shard)      # all of this rank's tiles in ONE launch
            mi.abi.check(lib.mi355_render_accumulate(h, C.byref(p), C.c_void_p(film.data_ptr()), stream))
            pending[b] = dist.reduce(film, dst=0, op=dist.ReduceOp.SUM, async_op=True)   # the frame is assembled (and developed) on rank 0

    def barrier():
        for b in range(len(films)):                                       # every frame of the timed region is finished inside it
            finish(b)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(1000 + i)
    barrier()
    integ.counters(scene, reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    cnt = integ.counters(scene, reset=True)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(img).all()
    lanes_per_step = npix * args.spp
    value = lanes_per_step * args.steps / dt / 1e6

    MODE_NAME = "fused, re-pack after every bounce" if args.wavefront else ("static" if args.static else ("fused+regen" if args.regen else "fused"))
    KERNEL_NAME = {"static": "k_render<path,film>", "fused+regen": "k_render_regen"}.get(MODE_NAME, "k_render_compact")
    # ---- roofline of the dominant kernel (this rank's launches) ------------------------------------
    launches = max(cnt["kernel_launches"], 1)
    kernel_ms = cnt["kernel_ms"] / launches
    alg = algorithmic_bytes(cnt, npix * args.steps * (len(my_tiles) / n_tiles if world > 1 else 1.0)) / launches
    # next to the nominal 8 TB/s, the device-to-device copy rate measured on THIS box (read + write bytes / time)
    copy_gbs = None
    if rank == 0:
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev); dst = torch.empty_like(src)     # 1 GiB each
        src.fill_(1.0); dst.copy_(src); torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record(); torch.cuda.synchronize(dev)
        copy_gbs = 10 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
    lanes_launch = cnt["n_gen"] / launches
    roofline = valu_roofline(KERNEL_NAME, {"k_render_compact": "k_render_compact<0, 0>", "k_render_regen": "k_render_regen<0>"}.get(KERNEL_NAME, "k_render<0, 1, 0>"),
                             kernel_ms, lanes_launch, alg)
    roofline["launches"] = int(launches)
    roofline["events_per_lane"] = {k: round(cnt[k] / max(cnt["n_gen"], 1), 4) for k in ("n_isect", "n_shade", "n_shadow")}
    if roofline.get("hbm") is not None:
        roofline["hbm"]["peak_measured_copy_GBs"] = None if copy_gbs is None else round(copy_gbs, 1)

    # ---- per-frame latency with nothing overlapped (kernel, film reduce, develop: each followed by a device sync) -------------------
    def timed_phase(fn):
        torch.cuda.synchronize(dev); t = time.perf_counter(); fn(); torch.cuda.synchronize(dev); return (time.perf_counter() - t) * 1e3
    lat = {"kernel_ms": 0.0, "reduce_ms": 0.0, "develop_ms": 0.0}
    n_lat = 3
    for i in range(n_lat):
        film = films[0]; film.zero_()
        p = integ._params(scene, 500 + i, args.spp, shard=my_shard if world > 1 else None)
        lat["kernel_ms"] += timed_phase(lambda: mi.abi.check(lib.mi355_render_accumulate(h, C.byref(p), C.c_void_p(film.data_ptr()), stream)))
        if dist is not None:
            lat["reduce_ms"] += timed_phase(lambda: dist.reduce(film, dst=0, op=dist.ReduceOp.SUM))
        if rank == 0:
            lat["develop_ms"] += timed_phase(lambda: mi.abi.check(lib.mi355_film_develop(h, 0, C.c_void_p(film.data_ptr()), C.c_void_p(img.data_ptr()), stream)))
        if dist is not None:
            dist.barrier()
    lat = {k: round(v / n_lat, 3) for k, v in lat.items()}
    lat["frame_ms"] = round(lat["kernel_ms"] + lat["reduce_ms"] + lat["develop_ms"], 3)
    if dist is not None:                          # the slowest rank's figures
        tt = torch.tensor([lat["kernel_ms"], lat["reduce_ms"], lat["develop_ms"], lat["frame_ms"]], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        lat = dict(zip(("kernel_ms", "reduce_ms", "develop_ms", "frame_ms"), [round(float(v), 3) for v in tt.tolist()]))
    lat["note"] = "one frame with every phase followed by a device sync (max over ranks); `ms_per_step` above is the pipelined rate: the reduce of frame k overlaps the kernel of frame k + 1"
    integ.counters(scene, reset=True)

    # ---- PRB backward leg (config C3) ------------------------------------------------------------------
    prb = None
    if not args.no_prb:
        try:
            d3 = mi.cornell_box()
            d3["sensor"]["film"]["width"] = args.res; d3["sensor"]["film"]["height"] = args.res
            rng = np.random.default_rng(12345)
            d3["tex"] = {"type": "diffuse", "reflectance": {"type": "bitmap", "raw": True, "filter_type": "bilinear", "wrap_mode": "repeat",
                                                             "data": rng.uniform(0.2, 0.8, (256, 256, 3)).astype(np.float32)}}
            if not args.prb_plain:
                d3["back"]["bsdf"] = {"type": "ref", "id": "tex"}
            scene3 = mi.load_dict(d3)
            prb_i = mi.PRBIntegrator({"max_depth": 6, "rr_depth": 5}); prb_i.timing = True
            params = mi.traverse(scene3)
            gin = torch.ones((H, W, 3), dtype=torch.float32, device=dev)
            shard = (rank, world) if world > 1 else None
            n_prb = max(2, args.steps // 2)
            prb_i.render_backward(scene3, params, gin, seed=mi.sample_tea_32(0, 1)[0], spp=args.prb_spp, shard=shard)
            barrier(); prb_i.counters(scene3, reset=True)
            t0 = time.perf_counter()
            for i in range(n_prb):
                params.zero_grad()
                prb_i.render_backward(scene3, params, gin, seed=mi.sample_tea_32(i, 1)[0], spp=args.prb_spp, shard=shard)
            barrier()
            dtb = time.perf_counter() - t0
            c3 = prb_i.counters(scene3, reset=True)
            if dist is not None:
                tt = torch.tensor([dtb], dtype=torch.float64, device=dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dtb = float(tt.item())
            fwd_ms = None
            if world == 1:                     # Integrator::render_forward on the same scene (timing only): tangent 1 on every parameter
                for k_ in params._entries:
                    params.set_tangent(k_, 1.0)
                prb_i.render_forward(scene3, params, seed=0, spp=args.prb_spp); torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for i in range(n_prb):
                    prb_i.render_forward(scene3, params, seed=i, spp=args.prb_spp)
                torch.cuda.synchronize(dev); fwd_ms = round((time.perf_counter() - t0) / n_prb * 1e3, 3)
                prb_i.counters(scene3, reset=True)
            g = params.grad("tex.reflectance.data") if not args.prb_plain else params.grad_buffer(dev)
            prb = {"metric": "Msamples/s PRB backward (one-pass adjoint), Cornell %d^2 @%dspp, 256x256x3 albedo texture" % (args.res, args.prb_spp),
                   "value": round(npix * args.prb_spp * n_prb / dtb / 1e6, 2), "steps": n_prb, "ms_per_step": round(dtb / n_prb * 1e3, 3),
                   "kernel_ms": round(c3["kernel_ms"] / max(c3["kernel_launches"], 1), 4),
                   "roofline": valu_roofline("k_prepare_grad + k_prb_backward_1p (+ redo)", "k_prb_backward_1p<0, false>",
                                             c3["kernel_ms"] / max(c3["kernel_launches"], 1), c3["n_gen"] / max(c3["kernel_launches"], 1),
                                             # one walk, shade record 220 B, 24 B of dL per lane, 8 B per gradient float
                                             ((88 + 24) * c3["n_gen"] + 48 * c3["n_isect"] + 220 * c3["n_shade"] + 104 * c3["n_shadow"]) / max(c3["kernel_launches"], 1)
                                             + 8 * params.grad_buffer(dev).numel()),
                   "render_forward_ms": fwd_ms, "redo_groups": int(c3.get("n_redo_groups", 0)), "grad_abs_sum": float(g.abs().sum().item()), "finite": bool(torch.isfinite(g).all().item())}
            scene3.release()
        except Exception as e:                                      # a secondary leg never takes the headline line with it
            prb = leg_failed('prb', e)

    # ---- optional: config C4 (FULL kernels: glossy / specular BSDFs, spheres), tile-sharded like the headline leg ----
    c4 = None
    if not args.no_veach:
        try:
            sc4 = mi.load_dict(mi.veach_mis_scene(1024))
            i4 = mi.PathIntegrator({"max_depth": 6, "rr_depth": 5})
            tuned4 = i4.tune(sc4, 512)                                    # re-packing or path regeneration, whichever is faster on this scene
            i4.timing = True
            sh4 = mi.shard_tiles(1024, 1024, rank, world, rows=TILE_ROWS) if world > 1 else None
            film4 = torch.zeros((1024, 1024, 4), dtype=torch.float32, device=dev)
            h4 = sc4.handle(local_rank)
            def step4(seed):
                film4.zero_()
                p = i4._params(sc4, seed, 512, shard=sh4)                 # (carries MI355_FLAG_REGEN when tune() chose it)
                mi.abi.check(lib.mi355_render_accumulate(h4, C.byref(p), C.c_void_p(film4.data_ptr()), stream))
                if dist is not None:
                    dist.all_reduce(film4, op=dist.ReduceOp.SUM)
            step4(99); barrier(); i4.counters(sc4, reset=True)
            n4 = max(2, args.steps // 3)
            t0 = time.perf_counter()
            for i in range(n4):
                step4(i)
            barrier(); dt4 = time.perf_counter() - t0
            k4 = i4.counters(sc4, reset=True)
            if dist is not None:
                tt = torch.tensor([dt4], dtype=torch.float64, device=dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt4 = float(tt.item())
            c4 = {"metric": "Msamples/s forward, Veach-MIS scene 1024^2 @512spp (BASELINE configs[3])", "value": round(1024 * 1024 * 512 * n4 / dt4 / 1e6, 2),
                  "steps": n4, "ms_per_step": round(dt4 / n4 * 1e3, 2), "kernel_ms": round(k4["kernel_ms"] / max(k4["kernel_launches"], 1), 3),
                  "events_per_lane": {k: round(k4[k] / max(k4["n_gen"], 1), 4) for k in ("n_isect", "n_shade", "n_shadow")},
                  "variant": "regen" if i4.regen else "repack", "tuned_kernel_ms": {k: round(v, 2) for k, v in tuned4.items()},
                  "finite": bool(torch.isfinite(film4).all().item())}
            l4 = max(k4["kernel_launches"], 1)
            c4["roofline"] = valu_roofline("k_render_regen<2>" if i4.regen else "k_render_compact<0, 2>", "k_render_regen<2>" if i4.regen else "k_render_compact<0, 2>",
                                           k4["kernel_ms"] / l4, k4["n_gen"] / l4, algorithmic_bytes(k4, 1024 * 1024 * n4 / max(world, 1)) / l4, PMC_FILES["veach"],
                                           lanes_profiled=(1024 * 1024 * 512 if i4.regen else None))  # tools/pmc.sh profiles this same workload; the regen grid is persistent
            sc4.release()
        except Exception as e:                                      # a secondary leg never takes the headline line with it
            c4 = leg_failed('veach', e)

    # ---- optional: particle tracer (forward part) -----------------------------------------------------
    pt = None
    if not args.no_ptracer:
        try:
            pt = {}
            for rf in ("box", "gaussian") + (("gaussian-direct",) if args.ptracer_direct else ()):   # gaussian: deferred tile-sorted splat (default); -direct: plain atomics (A/B)
                dp = mi.cornell_box(); dp["sensor"]["film"]["width"] = args.res; dp["sensor"]["film"]["height"] = args.res
                dp["sensor"]["film"]["rfilter"] = {"type": rf.split("-")[0]}
                scp = mi.load_dict(dp)
                ip = mi.ParticleTracerIntegrator({"max_depth": 6, "rr_depth": 5}); ip.timing = True; ip.direct_splat = rf.endswith("-direct")
                shard = (rank, world) if world > 1 else None
                ip.render(scp, seed=0, spp=args.prb_spp, shard=shard); barrier(); ip.counters(scp, reset=True)
                n_pt = max(2, args.steps // 3)
                t0 = time.perf_counter()
                for i in range(n_pt):
                    imgp = ip.render(scp, seed=i, spp=args.prb_spp, shard=shard, evaluate=False)
                barrier(); dtp = time.perf_counter() - t0
                kp = ip.counters(scp, reset=True)
                pt[rf] = {"value": round(npix * args.prb_spp * n_pt / dtp / 1e6, 2), "unit": "Msamples/s (light paths)", "ms_per_step": round(dtp / n_pt * 1e3, 3),
                          "kernel_ms": round(kp["kernel_ms"] / max(kp["kernel_launches"], 1), 3),
                          "connections_per_lane": round(kp["n_shadow"] / max(kp["n_gen"], 1), 3), "finite": bool(torch.isfinite(imgp).all().item())}
                if not rf.endswith("-direct"):
                    lp = max(kp["kernel_launches"], 1)
                    # 28 B ray + 20 B hit per intersection, 20 B per connection record written, sorted and read (x3); kernel_ms covers
                    # k_ptracer AND the four deferred-splat passes (the library brackets the whole call), the VALU count is k_ptracer's
                    pt[rf]["roofline"] = valu_roofline("k_ptracer<0, false> (+ k_splat_*)", "k_ptracer<0, false>", kp["kernel_ms"] / lp, kp["n_gen"] / lp,
                                                       (48 * kp["n_isect"] + 60 * kp["n_shadow"]) / lp, PMC_FILES["ptracer"])
                if rf == "gaussian" and world == 1:      # the derivative passes of the light tracer on the same scene (timings only)
                    def timed(fn, n=3):
                        fn(); torch.cuda.synchronize(dev); t0 = time.perf_counter()
                        for _ in range(n):
                            fn()
                        torch.cuda.synchronize(dev); return round((time.perf_counter() - t0) / n * 1e3, 3)
                    gin = torch.ones((args.res, args.res, 3), dtype=torch.float32, device=dev)
                    pa = mi.traverse(scp); pr = mi.traverse(scp); pr.keep(r".*\.emitter\.radiance\.value")
                    cube = [i for i, sh in enumerate(scp.shapes) if sh.id == "small-box"][0]
                    vt = torch.zeros((scp.shapes[cube].vertex_positions.shape[0], 3), dtype=torch.float32, device=dev); vt[:, 1] = 1.0
                    pt[rf]["render_backward_all_params_ms"] = timed(lambda: ip.render_backward(scp, pa, gin, seed=0, spp=args.prb_spp))
                    pt[rf]["render_backward_radiance_ms"] = timed(lambda: ip.render_backward(scp, pr, gin, seed=0, spp=args.prb_spp))
                    pt[rf]["render_forward_design_ms"] = timed(lambda: ip.render_forward_design(scp, shape=cube, vertex_tangent=vt, seed=0, spp=args.prb_spp))
                scp.release()
        except Exception as e:                                      # a secondary leg never takes the headline line with it
            pt = leg_failed('ptracer', e)

    # ---- BASELINE configs[4]: the lens doublet of notebooks/doublet_full_optim.ipynb -- ptracer + reverse-mode geometry adjoint -----------
    lens = None
    if not args.no_lens:
        try:
            ld = mi.lens_design
            surfs = [{'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 140, 'z0': 48}, {'radial_extent': 8.0, 'K': 0.0, 'c': -1 / 24, 'z0': 49}, {'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 24, 'z0': 52}]
            mats = [ld.LensMaterial("sf11", 1.6447, 0.2 * 64.17), ld.LensMaterial("nbk7", 1.5168, 0.2 * 64.17)]
            t0 = time.perf_counter()
            system = ld.LensSystem([ld.EvenAsphericSurface(**sf) for sf in surfs], mats, subdiv_levels=args.lens_subdiv); system.disable_all_materials()
            # cell 5 of the notebook: the rear surface constrained to a 50 mm focal length focused on the film; cell 7: three field sources
            s1 = system.get_rear_surface_params(50.0)
            system.elements[0].surface.params['c'] = float(s1['c']); system.elements[0].surface.params['z0'] = float(s1['z0']); system.compute_paraxial_quantities()
            n_src = 3
            prob = ld.DesignProblem(system, resolution=(128, 128), spp=128, learning_rate=3e-5, iters=300, num_sources=n_src, max_field_angle=5.0, init_camera_z=0.1, fov=100.0).prepare()
            build_s = time.perf_counter() - t0
            scl, pl = prob.scene, prob.params
            lshard = (rank, world) if world > 1 else None
            prob.shard = lshard                                        # N > 1: sharded light-sample grid, film + gradient all-reduce inside the calls
            il = scl.integrator(); il.timing = True
            system.update(pl, prob.optimizer); pl.update()
            gin_l = torch.ones((128, 128, 3), dtype=torch.float32, device=dev)

            def timed_l(fn, n=3):
                fn(); torch.cuda.synchronize(dev); il.counters(scl, reset=True); t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize(dev); ms = (time.perf_counter() - t0) / n * 1e3
                c = il.counters(scl, reset=True)
                timed_l.last = c
                return round(ms, 3), round(c["kernel_ms"] / max(c["kernel_launches"], 1), 3)

            def upd():
                for e in system.elements:                   # SceneParameters.update of the six meshes: refit + normals on the device
                    for key in (e.lens_key, e.baffle_key):
                        pl["%s.vertex_positions" % key] = pl["%s.vertex_positions" % key].detach() * 1.0
                pl.update(); il.render(scl, seed=0, spp=1, shard=lshard)
            per_sensor = []; lens_roof = {}
            import itertools
            fresh = itertools.count(1000)                   # a new seed per call: the lane order (pt_sort_lanes) is computed inside every timed call, as for the
                                                            # first field of an iteration; the other two fields of an iteration reuse it ("lane_order_reused" below)
            for sidx in range(n_src):                       # every field's camera: the iteration renders and differentiates all three
                rs_ms, rs_k = timed_l(lambda: il.render(scl, sensor=sidx, seed=next(fresh), spp=256, shard=lshard))
                cr = timed_l.last
                bs_ms, bs_k = timed_l(lambda: il.render_backward(scl, pl, gin_l, sensor=sidx, seed=next(fresh), spp=128, geometry=True, colour=False, shard=lshard))   # what a design step asks for
                if sidx == 0:
                    # kernel_ms brackets the whole call: lane sort (k_pt_keys / _scan / _scatter) + the tracing kernel (+ the deferred splat); the VALU
                    # count and the HBM bytes are the tracing kernel's.  Algorithmic bytes: render = 28 B ray + 20 B hit per
                    # intersection + 3 x 20 B per connection record; adjoint = 144 B of path tape (write + read) + 36 B of scattered triangle adjoint per vertex
                    lr_ = max(cr["kernel_launches"], 1); lanes_r = cr["n_gen"] / lr_
                    lens_roof["render"] = valu_roofline("k_ptracer<1, false> (+ k_pt_*, k_splat_*)", "k_ptracer<1, false>", rs_k, lanes_r, (48 * cr["n_isect"] + 60 * cr["n_shadow"]) / lr_, PMC_FILES["lens"])
                    lanes_b = 128 * 128 * 128 / max(world, 1)
                    lens_roof["geometry_adjoint"] = valu_roofline("k_ptracer_geo_bwd<1, false> (+ k_pt_*, k_vertex_normals_bwd)", "k_ptracer_geo_bwd<1, false>", bs_k, lanes_b,
                                                                  180.0 * cr["n_shade"] / max(cr["n_gen"], 1) * lanes_b, PMC_FILES["lens"])
                per_sensor.append({"sensor": scl.sensors()[sidx].id, "render_256spp_ms": rs_ms, "render_kernels_ms": rs_k, "render_backward_geometry_ms": bs_ms, "render_backward_kernels_ms": bs_k})
            r_ms = round(sum(q["render_256spp_ms"] for q in per_sensor), 3); r_k = round(sum(q["render_kernels_ms"] for q in per_sensor), 3)
            b_ms = round(sum(q["render_backward_geometry_ms"] for q in per_sensor), 3); b_k = round(sum(q["render_backward_kernels_ms"] for q in per_sensor), 3)
            rr_ms, rr_k = timed_l(lambda: il.render(scl, sensor=1, seed=0, spp=256, shard=lshard))          # same seed as the previous call: the sorted lane list is reused
            br_ms, br_k = timed_l(lambda: il.render_backward(scl, pl, gin_l, sensor=1, seed=1, spp=128, geometry=True, colour=False, shard=lshard))
            # the three fields from ONE pass over the light paths (mi355_render_accumulate_sensors): what DesignProblem.step calls when unsharded
            rm_ms = rm_k = None
            if lshard is None and il.sensors_share_light_paths(scl, list(range(n_src))):
                rm_ms, rm_k = timed_l(lambda: il.render_sensors(scl, list(range(n_src)), seed=next(fresh), spp=256))
            bm_ms = bm_k = None
            if lshard is None and il.sensors_share_light_paths(scl, list(range(n_src))):       # ... and their geometry adjoints from one walk and one sweep (mi355_render_backward_sensors)
                bm_ms, bm_k = timed_l(lambda: (pl.zero_grad(), il.render_backward_sensors(scl, pl, [gin_l] * n_src, list(range(n_src)), seed=next(fresh), spp=128)))
            ru_ms = ru_k = bu_ms = bu_k = None
            if not args.no_ab:
                il.no_lane_sort = True                      # A/B: the natural order of the light samples (MI355_FLAG_NO_LANE_SORT), sensor 0
                ru_ms, ru_k = timed_l(lambda: il.render(scl, sensor=0, seed=0, spp=256, shard=lshard))
                bu_ms, bu_k = timed_l(lambda: il.render_backward(scl, pl, gin_l, sensor=0, seed=1, spp=128, geometry=True, colour=False, shard=lshard))
                il.no_lane_sort = False
            a_ms, a_k = timed_l(lambda: il.render_backward(scl, pl, gin_l, seed=1, spp=128, geometry=True, colour=True, shard=lshard))    # + reflectances / radiances (sensor 0)
            c_ms, c_k = timed_l(lambda: il.render_backward(scl, pl, gin_l, seed=1, spp=128, geometry=False, shard=lshard))
            u_ms, _ = timed_l(upd)
            m_tw = np.ascontiguousarray(scl.sensors()[1].to_world.affine12(), dtype=np.float32)          # what moving a camera costs: host-side only
            t0 = time.perf_counter()
            for _ in range(200):
                lib.mi355_scene_set_param(scl.handle(), mi.abi.PARAM_SENSOR_TO_WORLD, 1, m_tw.ctypes.data_as(mi.abi.f32p), 12)
            edit_us = (time.perf_counter() - t0) / 200 * 1e6
            losses = [prob.step(it)[0] for it in range(2)]; torch.cuda.synchronize(dev)            # (first iterations: torch allocator / autograd warm-up)
            t0 = time.perf_counter(); losses += [prob.step(it)[0] for it in range(2, 6)]; torch.cuda.synchronize(dev); it_ms = (time.perf_counter() - t0) / 4 * 1e3
            info = scl.info()
            if dist is not None:                            # (host timers: the slowest rank's)
                tt = torch.tensor([r_ms, b_ms, a_ms, c_ms, u_ms, it_ms], dtype=torch.float64, device=dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                r_ms, b_ms, a_ms, c_ms, u_ms, it_ms = [round(float(v), 3) for v in tt.tolist()]
            lens = {"metric": "lens doublet (BASELINE configs[4], notebooks/doublet_full_optim.ipynb cell 7): 3 field sources / 3 sensors, ptracer 128^2; per iteration 3 x (render @256 spp + render_backward @128 spp over all vertices and IORs)",
                    "sensors": per_sensor, "roofline": lens_roof, "sensor_edit_us": round(edit_us, 2),
                    "render_all_sensors_one_pass_ms": rm_ms, "render_all_sensors_one_pass_kernels_ms": rm_k,
                    "render_backward_all_sensors_one_pass_ms": bm_ms, "render_backward_all_sensors_one_pass_kernels_ms": bm_k,
                    "lane_order_reused_sensor1": {"render_256spp_ms": rr_ms, "render_kernels_ms": rr_k, "render_backward_geometry_ms": br_ms, "render_backward_kernels_ms": br_k},
                    "unsorted_lanes_sensor0": {"render_256spp_ms": ru_ms, "render_kernels_ms": ru_k, "render_backward_geometry_ms": bu_ms, "render_backward_kernels_ms": bu_k}, "geo_tracer_check": {"max_abs_err_y_mm": prob.geo_check[0], "max_abs_err_u": prob.geo_check[1], "reference_reports": [0.0003799945116043091, 8.578179404139519e-06]},
                    "n_gpus": world, "sharding": "light-sample grid in 16-row tiles per rank, film + gradient all-reduce (RCCL) inside render / render_backward" if world > 1 else "none",
                    "triangles": info["triangles"], "bvh_nodes": info["bvh_nodes"], "scene_build_s": round(build_s, 2),
                    "render_256spp_ms": r_ms, "render_kernels_ms": r_k, "render_backward_geometry_ms": b_ms, "render_backward_kernels_ms": b_k,
                    "render_backward_geometry_and_colour_ms": a_ms, "render_backward_colour_only_ms": c_ms, "update_refit_plus_1spp_ms": u_ms, "design_iteration_ms": round(it_ms, 2),
                    "light_paths_per_s_backward": round(n_src * 128 * 128 * 128 / (b_ms * 1e-3) / 1e6, 2), "losses": [round(v, 3) for v in losses]}
            scl.release()
        except Exception as e:                                      # a secondary leg never takes the headline line with it
            lens = leg_failed('lens', e)

    # ---- CPU baseline (oracle) on a bounded sample of the same workload --------------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        orc = load_oracle()
        o = orc.OracleScene(scene, mi.abi)
        threads = o.l.orc_num_threads()
        # ramp the calibration sample until it takes >= 2 s (so the per-thread film setup is amortised), then scale it
        # linearly to ~cpu_seconds (10-30 s) of CPU work
        o.render(o.params(integ, 0, 1))
        spp_cal = 8
        while True:
            t0 = time.perf_counter(); o.render(o.params(integ, 0, spp_cal)); tc = time.perf_counter() - t0
            if tc >= 2.0 or spp_cal >= 1024:
                break
            spp_cal *= 4
        spp_cpu = int(max(1, min(4096, round(spp_cal * args.cpu_seconds / tc))))
        t0 = time.perf_counter(); _, n = o.render(o.params(integ, 0, spp_cpu)); t2 = time.perf_counter() - t0
        cpu = {"value": round(n / t2 / 1e6, 4), "unit": "Msamples/s", "cores": int(threads), "kind": "port",
               "sample": "same scene/integrator, %dx%d @ %d spp (%d lanes, %.1f s), oracle/mi_oracle.c with OpenMP" % (W, H, spp_cpu, n, t2)}

    if rank == 0:
        out = {"metric": "Msamples/sec fwd + PRB grad, Cornell 512^2 @256spp", "value": round(value, 2), "unit": "Msamples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "Cornell box (mi.cornell_box), forward path trace %dx%d @ %d spp, max_depth=6, rr_depth=5 (BASELINE configs[1])" % (W, H, args.spp),
                          "lanes_per_step": lanes_per_step, "mode": MODE_NAME, "sharding": "16-row tiles round-robin (one launch per rank), film reduce to rank 0 overlapped with the next frame" if world > 1 else "none"},
               "roofline": roofline, "latency": lat, "cpu_baseline": cpu, "prb": prb}
        if c4 is not None:
            out["veach"] = c4
        if pt is not None:
            out["ptracer"] = pt
        if lens is not None:
            out["lens"] = lens
        print(json.dumps(out), flush=True)
    scene.release()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
