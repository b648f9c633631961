#!/usr/bin/env python3
"""Benchmark of the north-star hot path: one incremental training step of R50-C4 Faster R-CNN with Attentive RoI
Distillation + inclusive distillation (BASELINE.json configs[2]: task 15-5, --feat ard --dist_type id, batch 4 per GPU,
synthetic 600x1000 images, fp32), data-parallel over N MI355X with one RCCL all-reduce of the flat gradient per step.

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --gpus N ...            (WORLD_SIZE unset: starts its own N rank processes, one per GPU, before touching the GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus 8 --task 10-10 --batch-per-gpu 2      (BASELINE.json configs[3])

Rank 0 prints ONE JSON line (contract in the task description) plus the `roofline` and `cpu_baseline` objects.
"""
import argparse
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool: RCCL needs it before HIP initialises
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMG_ARD = 1304.0   # algorithmic conv/linear FLOPs of one ARD training image
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: v_mfma_f32_32x32x16_bf16, dense
PEAK_CLOCK_GHZ = 2.4             # same guide: the clock both peaks are quoted at
PROF_NAMES = ["conv_igemm_kernel<128,128>", "conv_igemm_kernel<128,64>", "conv_igemm_kernel<64,64>", "conv_igemm_kernel<128,64,small_c>",
              "conv_wgrad_kernel", "roi_align_fwd", "roi_align_bwd", "conv_igemm_bf16_kernel", "conv_wgrad_bf16_kernel",
              "conv_igemm_x6_kernel<128,128>", "conv_igemm_x6_kernel<128,64>", "conv_igemm_x6_kernel<64,64>",
              "conv_igemm_x6w_kernel<128,128>", "conv_igemm_x6w_kernel<128,64>", "conv_igemm_x6w_kernel<64,64>", "conv_tail64_x6w_kernel",
              "conv_igemm_x6w_kernel<128,128,NP=3>", "conv_igemm_x6w_kernel<128,64,NP=3>", "conv_igemm_x6w_kernel<64,64,NP=3>", "conv_wgrad_x6_kernel<3>"]
# (positions = abr::ProfId in csrc/common.h.  One row per TEMPLATE INSTANCE of the bf16x6 implicit GEMM, named as rocprofv3 names them
#  (`conv_igemm_x6w_kernel<128, 128, 1, 4, true>` ...: x6w = the weights-direct form, x6 = the form that splits the weight tile per workgroup),
#  so every row's fraction can be recomputed from profiles/ alone.  id 8 is the
#  weight-gradient kernel of the chosen arithmetic: conv_wgrad_x6_kernel under --math bf16x6, conv_wgrad_bf16_kernel under --math bf16.)


def cpu_baseline(model_target, images, targets, n_old):
    """Reference CPU path (forward + 4 losses; it has no CPU backward) on a bounded sample, on this box's host cores."""
    import torch
    from abr_iod_amd.utils.checkpoint import reference_state_dict
    from oracle.step_ref import cpu_forward_loss  # the oracle is the checker / baseline only

    cores = min(os.cpu_count() or 1, 32)  # oneDNN convs on this path stop scaling (and regress badly) beyond a few dozen threads
    torch.set_num_threads(cores)
    sd = reference_state_dict(model_target)
    n_img = min(2, images.shape[0])                     # BASELINE.json configs[0]: "2 synthetic 600x1000 images, CPU-only forward+loss"
    img = images[:n_img].cpu()
    gtb = [targets[i].bbox.cpu().numpy() for i in range(n_img)]
    gtl = [targets[i].get_field("labels").cpu().numpy() for i in range(n_img)]
    cpu_forward_loss(sd, img, gtb, gtl, n_old, timings={})  # warm-up (thread pools, oneDNN primitive caches)
    reps, acc = 6, {}     # ~10 s of host work: a bounded sample, the same batch each time
    for _ in range(reps):
        tm = {}
        cpu_forward_loss(sd, img, gtb, gtl, n_old, timings=tm)
        for k, v in tm.items():
            acc[k] = acc.get(k, 0.0) + v
    return {"value": round(reps * n_img / acc["total"], 4), "unit": "img/s", "cores": cores, "kind": "port",
            "sample": "{} passes over a batch of {} synthetic 600x1000 images (BASELINE.json configs[0]), target-model forward + 4 detector losses "
                      "only (the reference has no CPU backward: csrc/ROIAlign.h:44; no source model, no distillation), torch-CPU convs on {} threads + "
                      "oracle.c ROIAlign/NMS single-threaded as the reference's".format(reps, n_img, cores),
            "seconds_per_image": {k: round(v / (reps * n_img), 3) for k, v in acc.items()}}


def _pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (separate rocprofv3 --pmc passes of this same command,
    tools/pmc_traffic.sh: 2*FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md), launch-weighted over its template
    instances.  None when no summary names the kernel."""
    import glob
    prefix = {"conv_igemm_kernel<128,128>": "conv_igemm_kernel<128, 128,", "conv_wgrad_kernel": "conv_wgrad_kernel<",
              "conv_igemm_x6_kernel<128,128>": "conv_igemm_x6_kernel<128, 128,", "conv_igemm_x6_kernel<128,64>": "conv_igemm_x6_kernel<128, 64,",
              "conv_igemm_x6_kernel<64,64>": "conv_igemm_x6_kernel<64, 64,", "conv_wgrad_x6_kernel": "conv_wgrad_x6_kernel<6>",
              "conv_wgrad_x6_kernel<3>": "conv_wgrad_x6_kernel<3>",
              "conv_igemm_x6w_kernel<128,128>": "conv_igemm_x6w_kernel<128, 128,", "conv_igemm_x6w_kernel<128,64>": "conv_igemm_x6w_kernel<128, 64,",
              "conv_igemm_x6w_kernel<64,64>": "conv_igemm_x6w_kernel<64, 64,",
              "conv_igemm_x6w_kernel<128,128,NP=3>": "conv_igemm_x6w_kernel<128, 128,", "conv_igemm_x6w_kernel<128,64,NP=3>": "conv_igemm_x6w_kernel<128, 64,",
              "conv_igemm_x6w_kernel<64,64,NP=3>": "conv_igemm_x6w_kernel<64, 64,",
              "conv_igemm_kernel<64,64>": "conv_igemm_kernel<64, 64,", "conv_igemm_kernel<128,64>": "conv_igemm_kernel<128, 64, 4, 1, false"}.get(kernel)
    # template instances of the weights-direct kernel end in their product count: ", 6>" (bf16x6), ", 3>" (f16x3), ", 1>" (bf16)
    suffix = ", 3>" if "NP=3" in kernel else (", 6>" if "_x6w_" in kernel else "")
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        hits = [v for k, v in json.load(open(f))["kernels"].items() if prefix and k.startswith(prefix) and k.endswith(suffix)]
        n_l = sum(v["launches"] for v in hits)
        if n_l:
            return int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hits) / n_l), os.path.relpath(f, ROOT)
    return None, None


def _peak_of(name):
    if "NP=3" in name or name.endswith("<3>"):
        return round(PEAK_BF16_MFMA_TFLOPS / 3.0, 1)   # f16x3: three fp16 MFMA products (bf16 rate) per fp32 multiply-add
    if "_x6_" in name or "_x6w_" in name:
        return round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1)   # six bf16 MFMA products per fp32 multiply-add
    return PEAK_BF16_MFMA_TFLOPS if "bf16" in name else PEAK_FP32_MFMA_TFLOPS


class _Totals(dict):
    """abr_prof_totals' rows + `alg_bytes` (abr_prof_bytes of the same profiled region)"""
    alg_bytes = {}
    clocks = {}


def _tf(fl, ms):
    return round(fl / (ms * 1e-3) / 1e12, 2) if ms > 0 else 0.0


def _alg_bytes_per_launch(kernel, totals, math, alg):
    """algorithmic HBM bytes per launch of `kernel` over every launch of the profiled region (abr_prof_bytes / abr_prof_totals), or None"""
    name = kernel.replace("_x6_kernel", "_bf16_kernel") if (math == "bf16x6" and kernel == "conv_wgrad_x6_kernel") else kernel
    b = alg.get(name) or alg.get(kernel)
    n = (totals.get(kernel) or (0, 0))[0]
    return int(b / n) if b and n else None


def prof_rows(prof, totals, steps, math, event_overhead_ms=0.0):
    """One row per conv kernel (per TEMPLATE INSTANCE for the bf16x6 implicit GEMM): achieved = executed flops of the sampled launches /
    their summed durations, over ALL sampled launches (what rocprofv3 --kernel-trace --stats averages too)."""
    rows = []
    if math == "bf16x6":
        prof = [(nm.replace("_bf16_kernel", "_x6_kernel"),) + tuple(rest) for nm, *rest in prof]
        totals = {k.replace("_bf16_kernel", "_x6_kernel"): v for k, v in totals.items()}
    for name, n, ms, fl, n_o, ms_o, fl_o in prof:
        launches_all, flops_all = totals[name]
        if n + n_o == 0 or "roi_align" in name:
            continue
        # the forward / dgrad kernels stamp their own first-workgroup-in / last-workgroup-out times (wall_clock64, abr::prof_stamp_*): no event
        # pair, no dispatch bubble; the weight-gradient kernels are bracketed by HIP events (conv_wgrad.hip says why) -- each within 3 %
        # of the durations rocprofv3 --kernel-trace reports for the same run
        if "wgrad" in name:   # event-bracketed: the dispatch gap an event pair exposes (measured on an empty kernel) comes off every launch
            ms, ms_o = max(ms - n * event_overhead_ms, 0.5 * ms), max(ms_o - n_o * event_overhead_ms, 0.5 * ms_o)
        avg_ms = (ms + ms_o) / (n + n_o)
        rows.append({"kernel": name, "launches_per_step": round(launches_all / steps, 1), "sampled_launches": int(n + n_o),
                     "avg_launch_ms": round(avg_ms, 4), "ms_per_step": round(avg_ms * launches_all / steps, 3),
                     "gflop_per_launch": round(flops_all / max(launches_all, 1) / 1e9, 3),
                     "gflop_per_launch_sampled": round((fl + fl_o) / (n + n_o) / 1e9, 3),
                     "achieved": _tf(fl + fl_o, ms + ms_o), "peak": _peak_of(name), "frac": round(_tf(fl + fl_o, ms + ms_o) / _peak_of(name), 4),
                     "exclusive": {"launches": int(n), "avg_launch_ms": round(ms / max(n, 1), 4), "tflops": _tf(fl, ms)},
                     "overlapped": {"launches": int(n_o), "avg_launch_ms": round(ms_o / max(n_o, 1), 4), "tflops": _tf(fl_o, ms_o)}})
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows, totals


def _add_clocks(rows, clocks):
    """sustained shader clock under each self-stamping kernel (abr_prof_clocks), and the row's fraction of the peak AT that clock"""
    for x in rows:
        cyc, ms = clocks.get(x["kernel"]) or clocks.get(x["kernel"].replace("_x6_kernel", "_bf16_kernel")) or (0.0, 0.0)
        if cyc > 0 and ms > 0:
            x["sustained_clock_ghz"] = round(cyc / ms / 1e6, 3)
            x["frac_at_sustained_clock"] = round(x["frac"] * PEAK_CLOCK_GHZ / x["sustained_clock_ghz"], 4)


def roofline(prof, totals, a, elapsed, event_overhead_ms=0.0, serialised=None):
    """The `roofline` object.  `kernels_by_time` / `all_conv_kernels`: one row per conv kernel and template instance, timed INSIDE the
    step (other streams' kernels share the CUs: durations are stretched); `serialised`: the same rows from a short re-run of the step with
    every stream folded into one (kernel quality without time sharing); `whole_step` relates the step's executed and algorithmic conv
    flops to the wall clock.  The top row by time fills the contract's fields."""
    if not prof or not any(r[1] + r[4] > 0 for r in prof if "roi_align" not in r[0]):
        return {"bound": "mfma", "note": "no conv launch was sampled"}
    alg = getattr(totals, "alg_bytes", {})
    clocks = getattr(totals, "clocks", {})
    rows, totals = prof_rows(prof, totals, a.steps, a.math, event_overhead_ms)
    _add_clocks(rows, clocks)
    top = rows[0]
    traffic, traffic_src = _pmc_traffic(top["kernel"])
    step_s = elapsed / a.steps
    exec_flops_step = sum(v[1] for k, v in totals.items() if "roi_align" not in k) / a.steps
    alg_flops_step = GFLOP_PER_IMG_ARD * 1e9 * a.batch_per_gpu
    peak_step = (PEAK_FP32_MFMA_TFLOPS if a.math == "f32" else round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1) if a.math == "bf16x6"
                 else round(PEAK_BF16_MFMA_TFLOPS / 3.0, 1) if a.math == "f16x3" else None)
    r = {"bound": "mfma", "kernel": top["kernel"], "achieved": top["achieved"], "peak": top["peak"], "unit": "TFLOP/s", "frac": top["frac"],
         "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
         # `peak` is priced at the 2.4 GHz maximum clock; under matrix-core load the chip clocks to its power budget (MI355X_MICROARCH.md, "DVFS
         # give-back").  sustained_clock_ghz = workgroup 0's own s_memtime cycles / s_memrealtime time over the sampled launches
         "sustained_clock_ghz": top.get("sustained_clock_ghz"), "frac_at_sustained_clock": top.get("frac_at_sustained_clock"),
         # the algorithmic bytes of the same kernel's launches, counted live by the library (every operand and the output once, + a fused
         # residual / mask read): traffic / traffic_algorithmic = how much of the HBM-side traffic is re-reads
         "traffic_algorithmic": _alg_bytes_per_launch(top["kernel"], totals, a.math, alg),
         "traffic_over_algorithmic": (round(traffic / _alg_bytes_per_launch(top["kernel"], totals, a.math, alg), 3)
                                      if traffic and _alg_bytes_per_launch(top["kernel"], totals, a.math, alg) else None),
         "launches": "every launch position of the step sampled equally often over the timed region (1 launch in {} per step, rotating; "
                     "exclusive and stream-overlapped launches alike)".format(1 if a.time_all_kernels else max(d for d in range(1, 11) if a.steps % d == 0)),
         "flops_counted": "executed multiply-adds x2 of each launch (a Winograd F(4x4,3x3) conv executes 1/4 of its algorithmic MACs)",
         "avg_launch_ms": top["avg_launch_ms"], "avg_gflop_per_launch": top["gflop_per_launch_sampled"],
         "timing": "sampled launches: in-kernel wall_clock64 stamps (first workgroup in, last workgroup out) for the forward / dgrad kernels -- a "
                   "HIP-event pair would add its dispatch gap ({} us around an empty kernel on this box) to a ~150 us kernel; HIP events for the "
                   "weight-gradient kernels, whose traced duration includes the end-of-kernel write-back of their parked partial "
                   "tiles (that gap subtracted per launch)".format(round(event_overhead_ms * 1e3, 1)),
         "kernels_by_time": rows[:2],
         "all_conv_kernels": {x["kernel"]: dict({k: x[k] for k in ("launches_per_step", "avg_launch_ms", "ms_per_step", "gflop_per_launch", "achieved", "frac", "sustained_clock_ghz") if k in x},
                                                algorithmic_mb_per_launch=(round(_alg_bytes_per_launch(x["kernel"], totals, a.math, alg) / 1e6, 1)
                                                                           if _alg_bytes_per_launch(x["kernel"], totals, a.math, alg) else None))
                              for x in rows},
         "whole_step": {"executed_gflop": round(exec_flops_step / 1e9, 1), "executed_tflops": round(exec_flops_step / step_s / 1e12, 2),
                        "algorithmic_gflop": round(alg_flops_step / 1e9, 1), "algorithmic_tflops": round(alg_flops_step / step_s / 1e12, 2),
                        "note": "conv/linear flops of one rank's step / wall-clock step time (everything else in the step included)"}}
    if peak_step:
        r["whole_step"]["peak"] = peak_step
        r["whole_step"]["executed_frac"] = round(exec_flops_step / step_s / 1e12 / peak_step, 4)
        r["whole_step"]["algorithmic_frac"] = round(alg_flops_step / step_s / 1e12 / peak_step, 4)
    if not getattr(a, "standard_geometry", True):   # the 1304 GFLOP / image count is the 600x1000 step's
        for k in ("algorithmic_gflop", "algorithmic_tflops", "algorithmic_frac"):
            r["whole_step"].pop(k, None)
    elif a.math == "bf16x6":   # for orientation: the same flops against what the fp32 matrix pipe (round 1's arithmetic) could ever deliver
        r["whole_step"]["vs_fp32_mfma_peak"] = {"peak": PEAK_FP32_MFMA_TFLOPS, "executed_frac": round(exec_flops_step / step_s / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                                "algorithmic_frac": round(alg_flops_step / step_s / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
    r["note"] = ("per-launch durations inside the step are stretched by the kernels of the other HIP streams sharing the CUs (3 streams in the "
                 "forward pass, 2-3 in the backward pass): a row's ms_per_step can exceed the step, and `frac` of the overlapped step is a "
                 "time-sharing figure; `serialised` holds the same rows with every stream folded into one = the kernels' own rate")
    if serialised is not None:
        r["serialised"] = serialised
        own = serialised["kernels"].get(top["kernel"])
        if own:   # the same kernel's OWN rate (streams folded): the kernel-quality figure next to the in-step, time-shared one above
            r["own_rate"] = {"kernel": top["kernel"], "achieved": own["achieved"], "peak": own["peak"], "frac": own["frac"], "avg_launch_ms": own["avg_launch_ms"],
                             "source": "roofline.serialised (informational re-run behind the timed region)"}
    return r


# BASELINE.json configurations as the reference's launch scripts spell them (scripts/run_SI.sh:24-32, scripts/run_MI.sh:11-21):
#        task     dist_type feat  alpha beta gamma
TASKS = {"15-5": ("id", "ard", 0.5, 1.0, 1.0),      # configs[2]
         "10-10": ("id", "ard", 0.1, 0.5, 1.0),     # configs[3]
         "10-5": ("id", "ard", 1.0, 1.0, 1.0),      # configs[4] (step 1 of the multi-step schedule)
         "19-1": ("id", "ard", 1.0, 1.0, 5.0)}


def launch_ranks(n):
    """`python bench.py --gpus N` without an outer launcher: start N fresh rank processes (one per GPU; the reference does the same
    through torch.distributed.launch, scripts/run_SI.sh:6 + tools/train_incremental.py:406-409).  The parent has not imported
    torch or touched the GPU; it only waits and exits with the children's status -- no exec of a GPU-initialised process."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            time.sleep(0.2)
            bad = [p for p in procs if p.poll() not in (None, 0)]
            if bad:   # one rank died: the others would wait in a collective forever
                rc = bad[0].returncode
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                break
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc or max((abs(p.returncode or 0) for p in procs), default=0)


def pin_rank_to_cores(local_rank, local_world):
    """Give this rank process its own contiguous slice of the host cores it may run on (os.sched_setaffinity); returns the slice or None when the
    platform has no affinity call or the slice would be empty.  Must run before the GPU runtime starts its helper threads."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity") or os.environ.get("ABR_PIN_RANKS", "1") == "0":
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // local_world
        if per < 1:
            return None
        mine = cores[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        return [mine[0], mine[-1]]
    except OSError:
        return None


def fold_streams(on, optimizer):
    """Fold every HIP stream of the step into the current one (or restore the defaults): weight gradients, the source model, proposal
    selection, the cross-step prefetches and the weight preparation all run in issue order -- the step `roofline.serialised` times.
    Same work, same results (tests/test_gpu_streams_equivalence.py runs both forms)."""
    from abr_iod_amd import ops
    from abr_iod_amd.engine import trainer
    from abr_iod_amd.modeling.rpn import rpn
    if on:
        saved = (trainer.SOURCE_STREAM, trainer.SOURCE_HEAD_STREAM, trainer.PIPELINE_SOURCE, trainer.PIPELINE_TARGET_FROZEN, trainer.EARLY_SECOND_PASS,
                 ops.WGRAD_SIDE_STREAM, rpn.PROPOSALS_SIDE_STREAM, getattr(optimizer, "_prep_stream", None))
        trainer.SOURCE_STREAM = trainer.SOURCE_HEAD_STREAM = trainer.PIPELINE_SOURCE = trainer.PIPELINE_TARGET_FROZEN = trainer.EARLY_SECOND_PASS = False
        ops.WGRAD_SIDE_STREAM = False
        rpn.PROPOSALS_SIDE_STREAM = False
        if hasattr(optimizer, "_prep_stream"):
            optimizer._prep_stream = False
        return saved
    (trainer.SOURCE_STREAM, trainer.SOURCE_HEAD_STREAM, trainer.PIPELINE_SOURCE, trainer.PIPELINE_TARGET_FROZEN, trainer.EARLY_SECOND_PASS,
     ops.WGRAD_SIDE_STREAM, rpn.PROPOSALS_SIDE_STREAM, prep) = optimizer._folded_saved
    if prep is not None:
        optimizer._prep_stream = prep


def rccl_summary(optimizer):
    from abr_iod_amd import _lib
    from abr_iod_amd.solver.grad_reducer import rccl_debug_summary
    return {"rccl_version": int(_lib.lib().abr_comm_rccl_version()),
            "log": rccl_debug_summary(os.environ.get("NCCL_DEBUG_FILE", "")) if os.environ.get("NCCL_DEBUG_FILE") else []}


def read_prof(_lib):
    """(rows of abr_prof_end, totals of abr_prof_totals, event-pair overhead in ms) after a profiled region"""
    buf = (ctypes.c_double * (6 * len(PROF_NAMES)))()
    _lib.check(_lib.lib().abr_prof_end(ctypes.cast(buf, ctypes.c_void_p), len(PROF_NAMES)), "prof_end")
    # (name, exclusive launches / ms / flops, overlapped launches / ms / flops)  -- include/abr_iod_hip.h abr_prof_end
    prof = [(PROF_NAMES[i],) + tuple(buf[6 * i + j] for j in range(6)) for i in range(len(PROF_NAMES))]
    tot = (ctypes.c_double * (2 * len(PROF_NAMES)))()
    _lib.check(_lib.lib().abr_prof_totals(ctypes.cast(tot, ctypes.c_void_p), len(PROF_NAMES)), "prof_totals")
    totals = {PROF_NAMES[i]: (tot[2 * i], tot[2 * i + 1]) for i in range(len(PROF_NAMES))}   # (launches, flops) of ALL launches
    by = (ctypes.c_double * len(PROF_NAMES))()
    _lib.check(_lib.lib().abr_prof_bytes(ctypes.cast(by, ctypes.c_void_p), len(PROF_NAMES)), "prof_bytes")
    totals = _Totals(totals)
    totals.alg_bytes = {PROF_NAMES[i]: by[i] for i in range(len(PROF_NAMES))}   # algorithmic HBM bytes of ALL launches per kernel id (same region)
    ck = (ctypes.c_double * (2 * len(PROF_NAMES)))()
    _lib.check(_lib.lib().abr_prof_clocks(ctypes.cast(ck, ctypes.c_void_p), len(PROF_NAMES)), "prof_clocks")
    # (shader cycles, ms) of workgroup 0 of the sampled launches per kernel id: the clock the chip sustained under that kernel
    totals.clocks = {PROF_NAMES[i]: (ck[2 * i], ck[2 * i + 1]) for i in range(len(PROF_NAMES))}
    ov = ctypes.c_double(0.0)
    _lib.check(_lib.lib().abr_prof_event_overhead_ms(ctypes.cast(ctypes.byref(ov), ctypes.c_void_p), _lib.stream()), "prof_event_overhead_ms")
    return prof, totals, float(ov.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=4)
    ap.add_argument("--task", choices=sorted(TASKS), default="15-5",
                    help="15-5 = BASELINE configs[2] (the metric's configuration); 10-10 with --batch-per-gpu 2 = configs[3]; 10-5 = configs[4]")
    ap.add_argument("--image-size", default="600x1000",
                    help="HxW of the synthetic images.  600x1000 = the VOC-shaped geometry BASELINE.json's metric is quoted on (the default and the only "
                         "reported value); 800x1333 = the reference's own INPUT defaults (config/defaults.py:44-46: no voc YAML overrides them), an "
                         "informational line that shows the kernels are not tuned to one geometry (C4 50x84, 63 000 anchors)")
    ap.add_argument("--mosaic-squares", action="store_true",
                    help="BASELINE.json configs[4]'s shape variety: every second step runs a batch of SQUARE images (min(H,W) on a side), as the "
                         "mosaic canvases of the box-rehearsal data path are (voc_abr.py:712-714: mean(w,h)^2 -> ~600x600 after the resize); aspect "
                         "grouping keeps each batch homogeneous (data/build.py:93-100).  Informational (the metric's batches are all 600x1000)")
    ap.add_argument("--batch-pool", type=int, default=4,
                    help="number of DISTINCT synthetic batches the timed region rotates over (same geometry, same config; other images, other numbers "
                         "of ground-truth boxes -> other matches, other NMS survivor counts, other sampled RoIs).  1 = the single repeated batch of "
                         "rounds 1-5, which the default run still reports as `single_batch_informational`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-math", action="store_true", help="skip the short informational re-run on the fp32 MFMA kernels (v_mfma_f32_32x32x2_f32)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--math", choices=["f32", "bf16", "bf16-all", "bf16x6", "f16x3"], default="f16x3",
                    help="f16x3 (default since round 5): fp32 tensors, fp32 accumulation, fp32 error bound -- every contraction operand is written as two "
                         "fp16 terms scaled by the operand's amax and the three leading cross products run on the fp16 matrix cores (guarded; admitted by "
                         "tests/test_gpu_f16x3_admission.py at bf16x6's bounds).  bf16x6 (rounds 2-4's default): fp32 tensors, fp32 accumulation, fp32 error bound -- every contraction operand is split EXACTLY into "
                         "three bf16 terms and the six leading cross products run on the bf16 matrix cores (range-guarded; admitted by "
                         "tests/test_gpu_x6_admission.py).  f32 = the fp32 MFMA kernels (v_mfma_f32_32x32x2_f32).  bf16 = BASELINE.json "
                         "configs[4]'s bf16 MFMA backbone (cfg.DTYPE bfloat16: operands ROUNDED to bf16 in-kernel -- reduced precision, never "
                         "the headline); bf16-all = RPN head and layer4 as well.")
    ap.add_argument("--time-all-kernels", action="store_true", help="event-bracket every conv / ROIAlign launch, not only the dominant kernel")
    ap.add_argument("--fold-streams", action="store_true",
                    help="run the whole benchmark with every HIP stream of the step folded into one (for a serialised-stream rocprofv3 profile)")
    ap.add_argument("--no-serialised-leg", action="store_true", help="skip the short serialised-stream re-run behind the timed region (roofline.serialised)")
    ap.add_argument("--no-single-batch-leg", action="store_true", help="skip the short single-batch re-run behind the timed region (single_batch_informational)")
    ap.add_argument("--share-frozen-prefix", action="store_true",
                    help="(informational, never the headline) compute the frozen stem + layer1 ONCE per batch for the source and the target model when "
                         "their frozen weights compare equal (engine/trainer.py::SHARE_FROZEN_PREFIX); the reference computes them in both models")
    ap.add_argument("--allreduce-backend", choices=["torch", "abr"], default=os.environ.get("ABR_ALLREDUCE_BACKEND", "torch"),
                    help="who issues the gradient all-reduces: torch.distributed under backend nccl (= RCCL; default) or the library's own RCCL communicator "
                         "(abr_allreduce_flat, csrc/comm.hip)")
    ap.add_argument("--no-rccl-debug", dest="rccl_debug", action="store_false",
                    help="--gpus N > 1: do not switch on rank 0's NCCL_DEBUG=INFO log (INIT,TUNING) that config.gradient_exchange_rccl summarises")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launcher self-test: start the ranks, form the process group (gloo when there is no GPU), all-reduce a 1 per rank, print the count")
    ap.add_argument("--inject-failure", type=int, default=-1,
                    help="(tests of the launcher only) with --rendezvous-only: this rank raises right after the rendezvous while the others enter a "
                         "second collective -- launch_ranks must end them and return non-zero")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))
    pinned = None
    if a.gpus > 1:
        # BEFORE anything touches the GPU (or imports torch: its thread pools inherit the mask): each rank issues ~500 launches per step from one
        # Python thread plus autograd's; eight such pairs migrating over each other's cores is the one host effect a single-GPU box cannot show
        pinned = pin_rank_to_cores(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
        if int(os.environ.get("RANK", "0")) == 0 and a.rccl_debug:
            # what RCCL chose (channels, rings / trees, algorithm + protocol per message size) goes to a file that config.gradient_exchange_rccl quotes
            os.environ.setdefault("NCCL_DEBUG", "INFO")
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,TUNING")
            os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/abr_rccl_%p.log")

    import torch
    import torch.distributed as dist

    if a.rendezvous_only:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        assert world == a.gpus
        gpu = torch.cuda.device_count() >= world   # (device_count does not initialise the GPU)
        if gpu:
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        if world > 1:
            dist.init_process_group(backend="nccl" if gpu else "gloo", init_method="env://")
        one = torch.ones(1, device="cuda" if gpu else "cpu")
        if world > 1:
            dist.all_reduce(one)
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"rendezvous_only": True, "ranks": int(one.item()), "backend": ("rccl" if gpu else "gloo") if world > 1 else None}), flush=True)
        if a.inject_failure >= 0 and world > 1:
            if int(os.environ.get("RANK", "0")) == a.inject_failure:
                raise RuntimeError("injected failure on rank {} after the rendezvous".format(a.inject_failure))
            dist.all_reduce(one)   # the survivors wait here for a rank that is gone: only the launcher can end them
        if world > 1:
            dist.destroy_process_group()
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")  # "nccl" IS RCCL on ROCm
        dist.barrier()   # creates the communicator before any timed or hook-issued collective

    from abr_iod_amd import _lib
    from abr_iod_amd.engine import train_step
    from abr_iod_amd.engine import trainer as _trainer
    _trainer.SHARE_FROZEN_PREFIX[0] = bool(a.share_frozen_prefix)   # the headline ALWAYS computes both models' frozen prefix, whatever the environment says
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
    from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer

    B = a.batch_per_gpu
    if a.math == "bf16-all":
        os.environ["ABR_BF16_SCOPE"] = "all"
    os.environ["ABR_CONV_MATH"] = a.math if a.math in ("f32", "f16x3", "bf16x6") else "f16x3"   # (--math bf16: bf16 backbone, the default arithmetic everywhere else)
    dist_type, feat, alpha, beta, gamma = TASKS[a.task]
    n_old_cls, n_new_cls = {"15-5": (15, 5), "10-10": (10, 10), "10-5": (10, 5), "19-1": (19, 1)}[a.task]
    cfg_s, cfg_t = make_cfgs(a.task, dist_type=dist_type, feat=feat, alpha=alpha, beta=beta, gamma=gamma, ims_per_batch=B * world,
                             overrides=("DTYPE", "bfloat16") if a.math in ("bf16", "bf16-all") else ())
    model_source, model_target = build_models(cfg_s, cfg_t, seed=0)       # same seed on every rank = broadcast weights
    optimizer = make_optimizer(cfg_t, model_target)
    scheduler = make_lr_scheduler(cfg_t, optimizer)
    if world > 1:
        optimizer.reducer.backend = a.allreduce_backend
        optimizer.reducer.measure = True      # per-bucket wait of the main stream for the exchange (events; every rank alike)
    IH, IW = (int(v) for v in a.image_size.lower().split("x"))
    images, targets = synthetic_batch(B, IH, IW, seed=42 + rank,           # each rank its own shard of the global batch
                                      label_range=(n_old_cls + 1, n_old_cls + n_new_cls + 1))
    batches = [(images, targets)]
    # the headline rotates over a pool of distinct batches: 1-5, 1-3, 1-8 and 1-12 ground-truth boxes per image (VOC trainval averages ~2.4 objects
    # per image, up to ~40), each rank its own images
    POOL_MAX_BOXES = (5, 3, 8, 12, 2, 6, 10, 4)
    for j in range(1, max(1, a.batch_pool)):
        batches.append(synthetic_batch(B, IH, IW, seed=42 + rank + 1009 * j, label_range=(n_old_cls + 1, n_old_cls + n_new_cls + 1),
                                       max_boxes=POOL_MAX_BOXES[j % len(POOL_MAX_BOXES)]))
    if a.mosaic_squares:
        side = min(IH, IW)
        batches.append(synthetic_batch(B, side, side, seed=1042 + rank, label_range=(n_old_cls + 1, n_old_cls + n_new_cls + 1)))
    standard = (IH, IW) == (600, 1000) and not a.mosaic_squares
    a.standard_geometry = standard
    if not standard:
        a.no_cpu_baseline = a.no_alt_math = True   # those legs describe the metric's own geometry

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if a.fold_streams:
        optimizer._folded_saved = fold_streams(True, optimizer)
    def batch_of(i):
        return batches[i % len(batches)]

    for i in range(a.warmup):
        im, tg = batch_of(i)
        train_step(model_source, model_target, im, tg, optimizer, scheduler, cfg_t, next_images=batch_of(i + 1)[0])

    # EVERY rank samples kernel timings inside the timed region (or none does): the stamps / event pairs cost a few microseconds per sampled
    # launch, and the slowest rank sets the step -- rank 0 must not carry work the others do not.  Only rank 0 reports.
    time_kernels = not a.no_kernel_timing
    barrier()
    if time_kernels:
        # Every conv / ROIAlign launch is COUNTED (flops per launch: abr_prof_totals); launch i of a kernel in step s is bracketed with
        # a HIP event pair on its launch stream iff (i + s) % n == 0, n = the largest divisor of --steps that is <= 10: every launch
        # position of the step is sampled exactly steps/n times, so the sampled averages ARE the population averages rocprofv3 reports.
        # (ROIAlign launches are still bracketed by HIP events; the conv kernels stamp themselves.)  --time-all-kernels samples all.
        sample_n = 1 if a.time_all_kernels else max(d for d in range(1, 11) if a.steps % d == 0)
        _lib.check(_lib.lib().abr_prof_set_mask(0xFFFFFFFF, sample_n), "prof_set_mask")
        _lib.check(_lib.lib().abr_prof_begin(), "prof_begin")
    t0 = time.perf_counter()
    last = None
    for i in range(a.steps):
        if time_kernels:
            _lib.lib().abr_prof_step_begin()
        im, tg = batch_of(a.warmup + i)
        last = train_step(model_source, model_target, im, tg, optimizer, scheduler, cfg_t, next_images=batch_of(a.warmup + i + 1)[0])
    barrier()
    elapsed = time.perf_counter() - t0
    prof = serialised = None
    if time_kernels:
        prof, prof_totals, event_overhead_ms = read_prof(_lib)
        if rank != 0:
            prof = None
        if world == 1 and not a.no_serialised_leg and not a.fold_streams:
            # AFTER the timed region, never part of `value`: the same step with every stream folded into one, every conv launch timed.
            # Inside the real step up to three streams' kernels share the CUs, so a kernel's in-step duration says how the step's time is
            # SHARED, not how good the kernel is; these rows are the kernels' own rates (what a stand-alone microbenchmark measures).
            SER_STEPS = 5
            optimizer._folded_saved = fold_streams(True, optimizer)
            try:
                for _ in range(2):
                    train_step(model_source, model_target, images, targets, optimizer, scheduler, cfg_t)
                torch.cuda.synchronize()
                _lib.check(_lib.lib().abr_prof_set_mask(0xFFFFFFFF, 1), "prof_set_mask")
                _lib.check(_lib.lib().abr_prof_begin(), "prof_begin")
                ts = time.perf_counter()
                for _ in range(SER_STEPS):
                    _lib.lib().abr_prof_step_begin()
                    train_step(model_source, model_target, images, targets, optimizer, scheduler, cfg_t)
                torch.cuda.synchronize()
                es = time.perf_counter() - ts
                sprof, stot, sov = read_prof(_lib)
            finally:
                fold_streams(False, optimizer)
            srows, _ = prof_rows(sprof, stot, SER_STEPS, a.math, sov)
            _add_clocks(srows, getattr(stot, "clocks", {}))
            serialised = {"steps": SER_STEPS, "ms_per_step": round(1e3 * es / SER_STEPS, 3),
                          "conv_kernel_ms_per_step": round(sum(x["ms_per_step"] for x in srows), 3),
                          "kernels": {x["kernel"]: {k: x[k] for k in ("launches_per_step", "avg_launch_ms", "ms_per_step", "gflop_per_launch", "achieved", "peak", "frac",
                                                                         "sustained_clock_ghz", "frac_at_sustained_clock") if k in x}
                                      for x in srows},
                          "note": "every stream of the step folded into one, every conv launch timed (informational re-run behind the timed region)"}
    single = None
    if len(batches) > 1 and a.batch_pool > 1 and not a.mosaic_squares and world == 1 and not a.no_single_batch_leg:
        # AFTER the timed region, never part of `value`: rounds 1-5's workload -- ONE batch repeated (constant NMS survivor counts, steady allocator)
        SB = 10
        for _ in range(2):
            train_step(model_source, model_target, images, targets, optimizer, scheduler, cfg_t, next_images=images)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(SB):
            train_step(model_source, model_target, images, targets, optimizer, scheduler, cfg_t, next_images=images)
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        single = {"value": round(B * SB / e1, 3), "unit": "img/s", "ms_per_step": round(1e3 * e1 / SB, 3), "steps": SB,
                  "note": "informational: the same step on ONE repeated batch (the workload rounds 1-5 reported); not the reported value"}
    rccl_ranks = 1
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        one = torch.ones(1, dtype=torch.float32, device="cuda")
        dist.all_reduce(one)                      # every rank contributes 1 over RCCL: the sum is the number of ranks that really took part
        rccl_ranks = int(round(float(one.item())))
        assert rccl_ranks == dist.get_world_size() == world

    if rank == 0:
        total_imgs = B * world * a.steps
        value = total_imgs / elapsed
        loss_dict, total = last
        out = {
            "metric": "training images/sec (R50-C4 Faster R-CNN + ARD)", "value": round(value, 3), "unit": "img/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
            "dtype": "f32" if a.math == "f32" else "f32 (tensors, accumulation and error bound; contractions via a 2-term fp16 split of both operands scaled by "
                                                   "their amax, 3 cross products on the fp16 matrix cores: f16x3)" if a.math == "f16x3"
            else "f32 (tensors, accumulation and error bound; contractions via an exact 3-term bf16 split of both "
                                                   "operands, 6 cross products on the bf16 matrix cores, range-guarded)"
            if a.math == "bf16x6" else "bf16 MFMA operands / f32 accumulate / f32 tensors ({}); the fp32-accurate f16x3 contractions elsewhere".format(
                "backbone layer1-3" if a.math == "bf16" else "backbone, RPN head, layer4"),
            "config": {"workload": "BASELINE.json {}: task {} ABR step, --feat {} --dist_type {} (alpha {}, beta {}, gamma {}), "
                                   "R50-C4, {}, 512 RoIs/img + 64 distillation RoIs/img, source+target models, gradient all-reduce + SGD step".format(
                                       {"15-5": "configs[2]", "10-10": "configs[3]", "10-5": "configs[4]"}.get(a.task, "(extra task)"), a.task, feat,
                                       dist_type, alpha, beta, gamma, "{}x{}".format(IH, IW) + (" alternating with {0}x{0} (mosaic-shaped) batches".format(min(IH, IW))
                                                                                                 if a.mosaic_squares else "")),
                       "informational": ("the frozen stem + layer1 computed once for both models (--share-frozen-prefix): NOT the reference's work per step, "
                                         "not the headline" if a.share_frozen_prefix else None) if standard
                                        else "not the metric's geometry (BASELINE.json: 600x1000 batches): GFLOP / roofline figures per image do not apply",
                       "shared_frozen_prefix": bool(a.share_frozen_prefix),
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}", "math": a.math,
                       "batch_pool": {"distinct_batches": len(batches), "gt_boxes_per_image": [[len(t) for t in tg] for _, tg in batches],
                                      "note": "the timed region visits them in turn (step i runs batch i mod n, the next one prefetched)"},
                       "rccl_ranks": rccl_ranks, "collective": "RCCL all-reduce of the flat gradient, 3 buckets, 2 under backward" if world > 1 else None,
                       "gradient_exchange": optimizer.reducer.describe() if world > 1 else None,
                       "gradient_exchange_rccl": (rccl_summary(optimizer) if world > 1 else None),
                       "host_cores_of_rank0": pinned,
                       "gflop_per_img_algorithmic": GFLOP_PER_IMG_ARD if standard else None},
            "final_losses": {k: round(float(v.detach()), 5) for k, v in loss_dict.items()},
            "conv_math_at_end": getattr(model_target, "conv_math", None),   # "f32" here = the range guard took the run off the bf16x6 kernels
            "math": a.math,
        }
        if single is not None:
            out["single_batch_informational"] = single
        if a.math in ("bf16x6", "f16x3"):
            from abr_iod_amd import ops as _o
            fl = _o.x6_range_flags(reset=False)
            out["x6_range_guard"] = {"flags": fl, "tiny_operands_seen": bool(fl & _o.X6_FLAG_TINY), "non_finite_operands_seen": bool(fl & _o.X6_FLAG_NONFINITE)}
            if a.math == "f16x3":
                small, seen = _o.h3_range_stats(reset=False)
                out["x6_range_guard"].update({"stale_amax_word": bool(fl & _o.H3_FLAG_STALE), "operand_elements_inspected": seen,
                                              "amax_reductions_by_the_host_side": {"calls": _o.amax_reductions[0], "mb": round(_o.amax_reductions[1] / 1e6, 1),
                                                                                   "note": "whole run (warm-up, timed steps, informational legs): operands whose producer did not emit its amax word"},
                                              "elements_more_than_18_binades_below_amax": small, "fraction": (small / seen if seen else None),
                                              "fraction_at_the_trainers_last_poll": getattr(_trainer.trainer_state(model_target), "h3_small_fraction", None)})
        if prof:
            out["roofline"] = roofline(prof, prof_totals, a, elapsed, event_overhead_ms, serialised)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model_target, images, targets, len(cfg_t.MODEL.ROI_BOX_HEAD.NAME_OLD_CLASSES))
        if world == 1 and a.math in ("bf16x6", "f16x3") and not a.no_alt_math:
            # Informational only, AFTER the timed region above and never part of `value`: the same workload in the PREVIOUS headline arithmetic --
            # bf16x6 (rounds 2-4) under --math f16x3, the fp32 MFMA kernels (round 1; v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak) under --math bf16x6.
            alt = "bf16x6" if a.math == "f16x3" else "f32"
            os.environ["ABR_CONV_MATH"] = alt
            try:
                ms6, mt6 = build_models(cfg_s, cfg_t, seed=0)
            finally:
                os.environ["ABR_CONV_MATH"] = a.math
            opt6 = make_optimizer(cfg_t, mt6)
            sch6 = make_lr_scheduler(cfg_t, opt6)
            for _ in range(3):
                train_step(ms6, mt6, images, targets, opt6, sch6, cfg_t)
            torch.cuda.synchronize()
            t6 = time.perf_counter()
            for _ in range(10):
                l6 = train_step(ms6, mt6, images, targets, opt6, sch6, cfg_t)
            torch.cuda.synchronize()
            e6 = time.perf_counter() - t6
            out["alt_math_" + ("bf16x6" if alt == "bf16x6" else "f32_mfma")] = {"value": round(B * 10 / e6, 3), "unit": "img/s", "ms_per_step": round(1e3 * e6 / 10, 3), "steps": 10,
                                        "note": "informational: the same step with ABR_CONV_MATH={} (the previous headline arithmetic); not the reported value".format(alt),
                                        "final_total_loss": round(float(l6[1].detach()), 5)}
            del ms6, mt6, opt6, sch6, l6
            from abr_iod_amd import ops as _ops
            _ops.conv_cache_clear()   # the library's per-weight Winograd-domain copies of the models just dropped
        print(json.dumps(out), flush=True)
    if world > 1:
        optimizer.reducer.close()        # (the library's own communicator, when --allreduce-backend abr made one)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
