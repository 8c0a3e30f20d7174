#!/usr/bin/env python3
"""Headline benchmark: images/sec of one SCAE training step (forward +
SCAE.loss + backward [+ gradient all-reduce] + RMSprop step) on synthetic
MNIST-shaped batches, BASELINE.json's metric / configs[1]:
MNIST 40x40, 24 part / 24 object capsules, bs=128 per GPU, fp32.

  python bench.py --gpus N --steps K --warmup W
  N>1 works both ways: under `python -m torch.distributed.run --nproc-per-node N
  ... bench.py --gpus N` (RANK/WORLD_SIZE in the environment) and bare
  (`python bench.py --gpus N`): the bare parent starts N fresh rank processes
  itself -- before anything touches the GPU -- and relays rank 0's JSON line.

Prints ONE JSON line on rank 0 (contract in the task brief), extended with
  roofline     : the dominant hand-written kernel (K8 conv data gradient) timed
                 live with HIP events, K1 / other K8 kernels alongside
  cpu_baseline : the oracle (CPU restatement of the reference) timed on the
                 host cores of this box, same config, bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIGS = {
    # BASELINE.json configs[1] (the configuration the metric is quoted on)
    "mnist_24_24_bs128": dict(
        model=dict(image_shape=(1, 40, 40), n_classes=10, n_part_caps=24,
                   n_obj_caps=24,
                   scae_params=dict(reconstruct_alternatives=False)),
        batch=128),
    # the reference's hydra default (configs/model/mnist.yaml): 40 / 32
    "mnist_40_32_bs128": dict(
        model=dict(image_shape=(1, 40, 40), n_classes=10, n_part_caps=40,
                   n_obj_caps=32,
                   scae_params=dict(reconstruct_alternatives=False)),
        batch=128),
    # BASELINE.json configs[2] shape (the reference quotes it in bf16; fp32 here)
    "mnist_48_64_bs1024": dict(
        model=dict(image_shape=(1, 40, 40), n_classes=10, n_part_caps=48,
                   n_obj_caps=64,
                   scae_params=dict(reconstruct_alternatives=False)),
        batch=1024),
    # BASELINE.json configs[4]
    "cifar_32_32_bs256": dict(
        model=dict(image_shape=(3, 32, 32), n_classes=10, n_part_caps=32,
                   n_obj_caps=32,
                   scae_params=dict(reconstruct_alternatives=False)),
        batch=256),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_FP32_PEAK_TFLOPS = 157.3   # dense fp32 matrix peak (256 CUs x 256 flop/clk x 2.4 GHz)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 matrix peak (MI355X_MICROARCH.md; not the 2:1-sparse figure)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="mnist_24_24_bs128",
                    choices=sorted(CONFIGS))
    ap.add_argument("--replay", default="auto", choices=["auto", "graph", "launches"],
                    help="how the captured step is re-issued: a HIP-graph replay, or the "
                         "library's record of the captured kernel launches, one "
                         "hipLaunchKernel each (TrainStep(replay=...): the same launches, "
                         "bit-identical state).  auto (single rank): both are timed over 3 "
                         "blocks and the faster runs the timed region")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-optimizer", action="store_true")
    ap.add_argument("--bf16", "--bf16-attention", dest="bf16",
                    action="store_true",
                    help="configs[2]'s precision: bf16 operands / fp32 "
                         "accumulation on the GEMM-shaped kernels (K7, K8); "
                         "not the headline fp32 line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=30)
    ap.add_argument("--force-spawn", action="store_true",
                    help="take the rank-launcher + RCCL process-group path even "
                         "for --gpus 1 (a 1-rank nccl group; proves the N>1 "
                         "plumbing on a 1-GPU box)")
    ap.add_argument("--in-graph-probe-s", type=float, default=30.0,
                    help="N > 1, --comm-mode auto: time limit of the guarded probe (build, "
                         "capture, two replays) of the all-reduce captured inside the graph")
    ap.add_argument("--watchdog-s", type=float, default=1500.0,
                    help="a rank process ends itself (exit code 3) when the run takes longer "
                         "than this; 0: no watchdog")
    ap.add_argument("--blocks", type=int, default=21,
                    help="the K-step timed region is repeated this many times "
                         "back to back (each bracketed by barrier + "
                         "synchronize); ms_per_step is the median block")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the extra_workloads legs (other BASELINE "
                         "configs, bf16, materialising / alternatives steps)")
    ap.add_argument("--eager-render", action="store_true",
                    help="materialise transformed_templates / mixing_logits "
                         "in every step (SCAE.forward's full contract) "
                         "instead of rendering them lazily on access")
    ap.add_argument("--alternatives", action="store_true",
                    help="reconstruct_alternatives=True (the SCAE ctor "
                         "default): three more no-grad reconstructions per "
                         "forward, one of them over B x n_obj_caps images")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N>1: one all-reduce after the whole backward instead "
                         "of the bucketed one that overlaps the encoder backward "
                         "(= --comm-mode '1 bucket')")
    ap.add_argument("--comm-mode", default="auto",
                    choices=["auto", "2 buckets", "1 bucket", "in graph"],
                    help="N>1 / --force-spawn: how the gradient all-reduce is "
                         "scheduled (train_step.TrainStep).  auto: all three "
                         "are measured on the ranks at hand (comm.modes) and "
                         "the fastest one runs the timed region")
    return ap.parse_args()


def metric_name(workload):
    """BASELINE.json's metric, with the workload's own image / batch size."""
    m = CONFIGS[workload]
    C, H, W = m["model"]["image_shape"]
    data = "MNIST" if C == 1 else "CIFAR-10"
    return (f"images/sec SCAE fwd+bwd, {data} {H}x{W} bs={m['batch']}, "
            f"1/2/4/8 MI355X")


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes
    (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) from this
    process, which has not touched the GPU (no exec of a GPU-initialised
    process anywhere), relay rank 0's stdout, fail if any rank fails."""
    import socket
    import subprocess
    n = args.gpus
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
                       "HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
            env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for p in procs:
            if p.poll() not in (None, 0):
                failed = p.returncode
        time.sleep(0.05)
    if failed is not None:          # a rank died: the others would wait forever
        for p in procs:
            if p.poll() is None:
                p.kill()
    out = procs[0].stdout.read().decode()
    for p in procs:
        p.wait()
    sys.stdout.write(out)
    sys.stdout.flush()
    codes = [p.returncode for p in procs]
    if any(codes):
        raise SystemExit(f"rank exit codes {codes}")


def build_model(cfg, seed):
    from torch_scae_amd import factory
    np.random.seed(seed)
    torch.manual_seed(seed)
    return factory.make_scae(cfg["model"])


# ----------------------------------------------------------------------------
def k1_algorithmic_bytes(cfg):
    """Algorithmic HBM bytes per IMAGE of the three K1 kernels (DESIGN.md,
    section 'Kernels'; SURVEY.md 8d conventions: compact inputs read once,
    contract outputs written once, fp32)."""
    m = cfg["model"]
    C, H, W = m["image_shape"]
    M = m["n_part_caps"]
    K, hw, HW = M + 1, 11 * 11, H * W
    compact = (M * C * hw + M * hw + 6 * M + M) * 4       # templates, alpha, pose, presence
    return {
        "render_fwd_kernel": compact + (K * C * HW + K * HW) * 4,
        "logprob_fwd_kernel": compact + C * HW * 4 + (2 * C * HW + HW) * 4,
        "render_bwd_kernel": compact + (3 * C * HW + HW) * 4
        + (M * C * hw + M * hw + 7 * M + 4 * K) * 4,
    }


def time_k1_kernels(cfg, device, reps=200, pose_regime="unit"):
    """Average duration of each K1 kernel: `reps` back-to-back launches through
    the C ABI on torch's current HIP stream, bracketed by HIP events recorded
    on that same stream (torch.cuda.Event records on the current stream).
    All buffers are pre-allocated; nothing else runs in the timed region."""
    import ctypes
    from torch_scae_amd import _lib, ops
    m = cfg["model"]
    B = cfg["batch"]
    C, H, W = m["image_shape"]
    M = m["n_part_caps"]
    K = M + 1
    g = torch.Generator(device="cpu").manual_seed(0)
    f = lambda *s: torch.empty(*s, device=device)   # noqa: E731
    templates = torch.rand(B, M, C, 11, 11, generator=g).to(device)
    alpha = (torch.randn(M, 11, 11, generator=g) * 0.5).to(device)
    # two pose regimes: "unit" -- templates about the size of the image, any shear (cells of
    # ~3.6 pixels); "init" -- what a freshly initialised part encoder emits (scale ~0.48:
    # the template covers twice the image, cells of ~7.6 pixels, rotations of +-20 degrees)
    if pose_regime == "init":
        pose = torch.randn(B, M, 6, generator=g) * torch.tensor(
            [0.06, 0.2, 0.25, 0.16, 0.02, 0.26]) + torch.tensor(
            [0.48, -0.04, 0.0, 0.0, 0.48, 0.0])
    else:
        pose = torch.randn(B, M, 6, generator=g) * 0.3
        pose[:, :, 0] += 1.0
        pose[:, :, 4] += 1.0
    pose = pose.to(device)
    presence = torch.rand(B, M, generator=g).to(device)
    x = torch.rand(B, C, H, W, generator=g).to(device)
    bg_value = torch.zeros(1, device=device)
    bg_ml = torch.zeros(1, device=device)
    tensors = [templates, alpha, pose, presence, None, bg_value, bg_ml, None,
               None]
    desc, _ = ops._make_desc(tensors, (H, W))
    dref = ctypes.byref(desc)
    tt, ml = f(B, K, C, H, W), f(B, K, 1, H, W)
    lp, lse_post, lse_prior = f(B, C, H, W), f(B, C, H, W), f(B, 1, H, W)
    glp = torch.ones(B, C, H, W, device=device)
    g_t, g_a = f(B, M, C, 11, 11), f(B, M, 11, 11)
    g_pose, g_pres, g_scal = f(B, M, 6), f(B, M), f(B, K, 4)
    p, st = ops._p, ops._stream(x)
    lib = _lib.load()

    def render():
        return lib.scae_template_render_fwd_f32(dref, p(tt), p(ml), st)

    def logprob():
        return lib.scae_render_gmm_logprob_fwd_f32(dref, p(x), p(lp),
                                                   p(lse_post), p(lse_prior),
                                                   st)

    def bwd():
        return lib.scae_render_gmm_bwd_f32(
            dref, p(x), p(lse_post), p(lse_prior), p(glp), None, None, p(g_t),
            p(g_a), p(g_pose), p(g_pres), None, p(g_scal), st)

    out = {}
    for name, fn in (("render_fwd_kernel", render),
                     ("logprob_fwd_kernel", logprob),
                     ("render_bwd_kernel", bwd)):
        for _ in range(5):
            assert fn() == 0
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        best = float("inf")
        for _ in range(3):
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        out[name] = best * 1e-3 / reps          # seconds per launch
    return out


def conv_layers(cfg):
    """(IH, Cin, Cout, stride) of the implicit-GEMM layers of the CNN encoder
    (every layer but the first), from the model the workload builds."""
    from torch_scae_amd import factory
    m = factory.make_scae(dict(cfg["model"]))
    convs = [c for c in m.part_encoder.encoder.network
             if isinstance(c, torch.nn.Conv2d)]
    H = cfg["model"]["image_shape"][1]
    out = []
    for i, c in enumerate(convs):
        if i > 0:
            out.append((H, c.in_channels, c.out_channels, c.stride[0]))
        H = (H - 3) // c.stride[0] + 1
    return out


def time_k8_kernels(cfg, device, reps=50, bf16=False):
    """Average duration of every K8 implicit-GEMM launch of one step (forward,
    data gradient, weight gradient of each encoder layer), timed like the K1
    kernels: raw C-ABI launches on torch's current stream between HIP events."""
    import ctypes
    from torch_scae_amd import _lib, ops
    lib, B = _lib.load(), cfg["batch"]
    p, I = ops._p, ctypes.c_int
    out = {}
    for li, (IH, Ci, Co, s) in enumerate(conv_layers(cfg)):
        OH = (IH - 3) // s + 1
        f = lambda *sh: torch.randn(*sh, device=device)   # noqa: E731
        x, w, bias = f(B, IH, IH, Ci), f(Co, Ci, 3, 3), f(Co)
        wf, wd = f(3, Co, 9, Ci), f(Ci, 9, Co)   # (wf: + its fragment-major copy, 2.5 x)
        y, dy, dx, dw, db = f(B, OH, OH, Co), f(B, OH, OH, Co), f(B, IH, IH, Ci), \
            f(Co, Ci, 3, 3), f(Co)
        st = ops._stream(x)
        lib.scae_conv3x3_relayout_f32(p(w), p(wf), p(wd), Co, Ci, st)
        splits = lib.scae_conv3x3_wgrad_splits(B, OH, OH, Ci, Co)
        part = f(splits * (9 * Co * Ci + Co))
        geo = (I(B), I(IH), I(IH), I(Ci), I(Co), I(s))
        calls = {
            "conv_fwd_kernel": lambda: lib.scae_conv3x3_fwd_f32(
                p(x), p(wf), p(bias), p(y), None, None, *geo, st),
            "conv_dgrad_kernel": lambda: lib.scae_conv3x3_dgrad_f32(
                p(dy), p(wd), p(x), p(dx), *geo, st),
            "conv_wgrad_kernel": lambda: lib.scae_conv3x3_wgrad_f32(
                p(dy), p(x), p(part), p(dw), p(db), *geo, st),
            # what the step launches: both gradients of a layer together
            "conv_bwd_pair_kernel": lambda: lib.scae_conv3x3_bwd_pair_f32(
                p(dy), p(wd), p(x), p(dx), p(part), *geo, st),
        }
        if bf16 and lib.scae_conv3x3_bf16r_supported(B, IH, IH, Ci, Co, s):
            # configs[2]'s precision: the bf16-RESIDENT kernels (csrc/conv_bf16.hip) on bf16
            # tensors -- what the step launches; its "pair" is two launches
            h = lambda t: t.to(torch.bfloat16)     # noqa: E731
            xh, yh, dyh, dxh = h(x), h(y), h(dy), h(dx)
            wfh, wdh = h(wf[0]), h(wd)
            hs = lib.scae_conv3x3_wgrad_bf16r_splits(B, OH, OH, Ci, Co)
            parth = f(hs * (9 * Co * Ci + Co))

            def fwd_h():
                return lib.scae_conv3x3_fwd_bf16r(p(xh), p(wfh), p(bias), p(yh), None, None,
                                                  None, *geo, st)

            def dgrad_h():
                return lib.scae_conv3x3_dgrad_bf16r(p(dyh), p(wdh), p(xh), p(dxh), None, *geo, st)

            def wgrad_h():
                return lib.scae_conv3x3_wgrad_bf16r(p(dyh), p(xh), p(parth), *geo, st)
            calls = {"conv_fwd_kernel": fwd_h, "conv_dgrad_kernel": dgrad_h,
                     "conv_wgrad_kernel": wgrad_h,
                     "conv_bwd_pair_kernel": lambda: dgrad_h() or wgrad_h()}
        flops1 = 2.0 * B * OH * OH * Co * 9 * Ci       # MACs x 2 of one pass
        for name, fn in calls.items():
            flops = flops1 * (2 if name == "conv_bwd_pair_kernel" else 1)
            for _ in range(5):
                assert fn() == 0
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            best = float("inf")
            for _ in range(3):
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            out.setdefault(name, []).append(
                dict(layer=li + 2, seconds=best * 1e-3 / reps, flops=flops))
    return out


def roofline(cfg, device, bf16=False, light=False):
    """Roofline of the dominant hand-written kernel of the step.  By rocprofv3
    total time per step that is the implicit-GEMM backward kernel of the CNN
    encoder (K8, MFMA-bound; data and weight gradient of a layer in one
    launch): achieved = algorithmic FLOPs of its launches (2 x MACs of both
    gradient convolutions, zero-padding work not counted) / their measured
    duration.  The K1 likelihood kernels (VALU-issue bound, priced against HBM
    as the brief asks) and the other K8 passes are reported alongside."""
    k1 = time_k1_kernels(cfg, device, reps=40 if light else 200)
    alg = k1_algorithmic_bytes(cfg)
    B = cfg["batch"]
    k8 = time_k8_kernels(cfg, device, reps=15 if light else 50, bf16=bf16)
    # --bf16: the bf16-resident kernels on v_mfma_f32_32x32x16_bf16, priced against the dense
    # bf16 peak (shapes they do not cover stay on the fp32 kernels)
    peak = MFMA_BF16_PEAK_TFLOPS if bf16 else MFMA_FP32_PEAK_TFLOPS
    name = "conv_bwd_pair_kernel"
    launches = k8[name]
    secs = sum(l["seconds"] for l in launches)
    flops = sum(l["flops"] for l in launches)
    achieved = flops / secs / 1e12
    # HBM traffic per launch from the committed rocprofv3 --pmc passes
    # (counters cannot be read from inside this run)
    traffic = None
    pmc = next((q for q in (os.path.join(ROOT, "profiles", r, "k8_pmc.json")
                            for r in ("r06", "r05", "r04", "r03", "r02", "r01"))
                if os.path.exists(q)), "")
    if os.path.exists(pmc) and cfg is CONFIGS["mnist_24_24_bs128"] and not bf16:
        k = json.load(open(pmc))["kernels"].get(name)
        if k:
            traffic = k["hbm_bytes_per_launch_raw"]
    others = {
        k: {"us": round(v * 1e6, 2), "bound": "hbm",
            "GBps": round(alg[k] * B / v / 1e9, 1),
            "frac": round(alg[k] * B / v / 1e9 / HBM_PEAK_GBS, 4),
            "bytes_per_image": alg[k]} for k, v in k1.items()}
    for k, ls in k8.items():
        t, fl = sum(l["seconds"] for l in ls), sum(l["flops"] for l in ls)
        pk = peak
        others[k] = {"us_per_step": round(t * 1e6, 2), "launches": len(ls),
                     "bound": "mfma", "TFLOPs": round(fl / t / 1e12, 1),
                     "peak": pk, "frac": round(fl / t / 1e12 / pk, 4)}
    return {
        "kernel": name, "bound": "mfma", "achieved": round(achieved, 1),
        "peak": peak, "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": traffic,
        "us_per_launch": round(secs / len(launches) * 1e6, 2),
        "launches_per_step": len(launches),
        "algorithmic_flops_per_launch": flops / len(launches),
        "per_layer_us": [round(l["seconds"] * 1e6, 2) for l in launches],
        "other_kernels": others,
        "arithmetic": "fp32 operands and results; the weight-gradient tiles (and the forward "
                      "loops) multiply them as six exact bf16 partial products on the bf16 "
                      "matrix cores (csrc/bf16x6.h), the data-gradient tiles on the fp32 MFMA; "
                      "achieved / peak: algorithmic fp32 FLOP/s against the fp32 MFMA peak"
        if not bf16 else "bf16 operands, fp32 accumulation",
        "note": "achieved/us_per_launch/algorithmic_flops_per_launch are "
                "averages over the kernel's launches of one step (one per "
                "encoder layer); traffic = FETCH_SIZE+WRITE_SIZE per launch "
                "from profiles/r0x/k8_pmc.json (separate rocprofv3 --pmc "
                "passes); K1 byte figures: DESIGN.md section 4",
    }


def step_timeline(step):
    """Per-launch (name, lane, start us, duration us) of ONE real step, from a timing event
    in front of and behind every launch of the step's recorded launch list, on the launch's
    own stream (scae_launch_list_timeline): the kernels in their places in the step, caches
    and clocks as the step leaves them -- not a stand-alone loop of one kernel.  None when
    the step has no launch list or its C-ABI calls and kernel launches are not one to one."""
    import ctypes
    from torch_scae_amd import _lib
    kl = getattr(step, "_klist", None)
    if not kl or not step._launches:
        return None
    lib = _lib.load()
    n = lib.scae_launch_list_size(kl)
    names = [getattr(fn, "__name__", "?") for fn, _, _ in step._launches]
    if n != len(names):
        return None
    out = (ctypes.c_float * (2 * n))()
    P = ctypes.c_void_p
    side = step.plan.side_stream
    best = None
    for _ in range(7):
        rc = lib.scae_launch_list_timeline(
            kl, P(torch.cuda.current_stream().cuda_stream),
            None if side is None else P(side.cuda_stream), out, 2 * n)
        if rc != 0:
            return None
        t = list(out)
        if best is None or max(t[1::2]) < max(best[1::2]):
            best = t
    return [dict(name=names[i], lane=lib.scae_launch_list_lane(kl, i),
                 start_us=round(best[2 * i], 2),
                 us=round(best[2 * i + 1] - best[2 * i], 2)) for i in range(n)]


def host_cores():
    """Cores this process may actually use: affinity mask capped by the
    cgroup CPU quota (the GPU box shows 256 logical CPUs but grants 16)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def step_algorithmic(cfg):
    """(FLOPs, hot-path bytes) per image of one training step, SURVEY.md 8d's
    conventions: FLOPs = 3 x forward 2*MAC of the CNN encoder, set transformer,
    capsule MLPs, K1 and the pose products (its table); bytes = compact inputs read once +
    contract outputs written once for K1 / K2 / K3+K4, forward and backward."""
    m = cfg["model"]
    C, H, W = m["image_shape"]
    M, Oc = m["n_part_caps"], m["n_obj_caps"]
    K, hw = M + 1, 121
    e = 4
    A = 6 * M + 6 + 1 + 2 * M
    Din = 23 + C * hw
    k1 = (M * C * hw + 6 * M + M + C * H * W + K * C * H * W + K * H * W) * e \
        + (2 * (M * C * hw + 6 * M + M) + C * H * W) * e
    k2 = (M * Din + M + Oc * 256) * e + (2 * M * Din + M + 2 * Oc * 256) * e
    k3 = 3 * (Oc * A + 6 * Oc * M + 3 * Oc * M + Oc) * e
    # SURVEY.md 8d's table (fwd 2*MAC x 3), counted on the reference modules
    flops = {(24, 24, 1): 209e6, (40, 32, 1): 250e6, (48, 64, 1): 313e6,
             (32, 32, 3): 164e6}[(M, Oc, C)]
    return flops, k1 + k2 + k3


def cpu_baseline(cfg, steps):
    """The oracle (kind 'port': our CPU restatement of the reference, pinned
    to reference-captured vectors) timed on this box's host cores."""
    from oracle import scae_oracle as O
    torch.set_num_threads(host_cores())
    model = build_model(cfg, seed=0)
    sd = model.state_dict()
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ocfg = O.prepare_model_params(**cfg["model"])
    B = cfg["batch"]
    m = cfg["model"]
    g = torch.Generator().manual_seed(0)
    image = torch.rand(B, *m["image_shape"], generator=g)
    label = torch.randint(0, 10, (B,), generator=g)
    M, Oc = m["n_part_caps"], m["n_obj_caps"]

    def step():
        noise = (torch.rand(B, M), torch.rand(B, Oc, 1), torch.rand(B, Oc, M))
        O.train_step(P, ocfg, image, label, noise)

    for _ in range(3):
        step()
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    cores = torch.get_num_threads()
    # the same step followed by the reference's optimiser (stock
    # torch.optim.RMSprop with the reference's hyper-parameters,
    # base_experiment.py:44-77): what the GPU headline's "+ RMSprop step" is
    # like-for-like with
    opt = torch.optim.RMSprop(list(P.values()), lr=3e-5, momentum=0.9,
                              eps=1e-2 / float(B) ** 2)
    step()
    opt.step()
    with_opt = []
    for _ in range(max(3, steps // 3)):
        t0 = time.perf_counter()
        step()
        opt.step()
        with_opt.append(time.perf_counter() - t0)
    med_opt = float(np.median(with_opt))
    # the same step on ONE thread (SURVEY.md 8d asks for both), 3 steps
    torch.set_num_threads(1)
    step()
    one = []
    for _ in range(3):
        t0 = time.perf_counter()
        step()
        one.append(time.perf_counter() - t0)
    torch.set_num_threads(cores)
    med1 = float(np.median(one))
    return {"value": round(B / med, 1), "unit": "images/sec",
            "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{steps} steps of the same workload (B={B}) after 3 "
                      f"warm-up, median; forward + loss + backward WITHOUT the "
                      f"optimiser (compare with step_without_optimizer; the "
                      f"GPU headline includes its RMSprop launch)",
            "ms_per_step": round(med * 1e3, 2),
            "spread_ms": [round(min(times) * 1e3, 2), round(max(times) * 1e3, 2)],
            "with_optimizer": {
                "value": round(B / med_opt, 1), "unit": "images/sec",
                "ms_per_step": round(med_opt * 1e3, 2),
                "sample": f"{len(with_opt)} steps + torch.optim.RMSprop.step, "
                          f"median (like-for-like with the GPU headline)"},
            "one_thread": {"value": round(B / med1, 1), "unit": "images/sec",
                           "ms_per_step": round(med1 * 1e3, 2),
                           "sample": "3 steps after 1 warm-up, median"}}


# ----------------------------------------------------------------------------
def make_step(cfg, device, seed=0, alternatives=False, **kw):
    from torch_scae_amd.train_step import TrainStep
    if alternatives:
        cfg = dict(cfg, model=dict(cfg["model"], scae_params=dict(
            cfg["model"]["scae_params"], reconstruct_alternatives=True)))
    model = build_model(cfg, seed=seed).to(device).train()
    return TrainStep(model, cfg["batch"], cfg["model"]["image_shape"], **kw)


def synthetic_batches(cfg, device, seed, n_batches=8):
    g = torch.Generator(device="cpu").manual_seed(seed)
    B = cfg["batch"]
    images = torch.rand(n_batches, B, *cfg["model"]["image_shape"],
                        generator=g).to(device)
    labels = torch.randint(0, 10, (n_batches, B), generator=g).to(device)
    return images, labels


LIVE_SPAN = 20   # RMSprop steps on U[0,1) noise after which the state is put back to init


def timed_blocks(step, images, labels, steps, warmup, blocks, barrier,
                 reduce_max=None, before_block=None, after_block=None,
                 refresh=None):
    """`blocks` back-to-back timed regions of EXACTLY `steps` steps each,
    every one bracketed by barrier() (a dist.barrier when there are ranks +
    torch.cuda.synchronize) on both sides; per block the MAX over ranks.
    `warmup` untimed steps run once in front -- or, with ``before_block``
    (which puts the training state back where the measurement wants it, outside
    the timed region), in front of EVERY block, so that a timed step is never
    more than warmup + steps optimiser steps away from that state.  Blocks
    longer than LIVE_SPAN steps call ``refresh`` (the same restore: three
    device copies queued on the step's stream, no host synchronisation) every
    LIVE_SPAN steps INSIDE the timed region -- its cost is part of the
    reported time -- so that no K lets the state train away from init.
    -> list of block seconds."""
    n = images.shape[0]
    out = []
    for b in range(blocks):
        if before_block is not None:
            before_block(b)
        if before_block is not None or b == 0:
            for i in range(warmup):
                step(images[i % n], labels[i % n])
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            if refresh is not None and i and i % LIVE_SPAN == 0:
                refresh()
            step(images[i % n], labels[i % n])
        barrier()
        out.append(time.perf_counter() - t0)
        if after_block is not None:
            after_block(b)
    if reduce_max is not None:
        out = reduce_max(out)
    return out


def timing_summary(block_s, steps):
    ms = sorted(1e3 * t / steps for t in block_s)
    return {"blocks": len(ms), "steps_per_block": steps,
            "median_ms": round(float(np.median(ms)), 4),
            "min_ms": round(ms[0], 4), "max_ms": round(ms[-1], 4),
            "p10_ms": round(ms[len(ms) // 10], 4),
            "p90_ms": round(ms[-1 - len(ms) // 10], 4)}


def capsule_state(model, image):
    """Where training has taken the part capsules: presence and the scale of
    the pose's linear part for a batch (eval mode: no presence noise).  On
    U[0,1) noise images the capsules switch off (presence -> 1e-18, scale ->
    0.01: the data-dependent K1 backward then runs in its collapsed-pose
    regime); structured images keep them alive."""
    enc = model.part_encoder
    was = enc.training
    with torch.no_grad():
        parts = enc.eval()(image)
    enc.train(was)
    pres, pose = parts.presence.float(), parts.pose.float()
    lin = torch.stack([pose[..., 0], pose[..., 1], pose[..., 3],
                       pose[..., 4]], -1).abs().amax(-1)
    return {"presence_median": float(pres.median()),
            "presence_mean": float(pres.mean()),
            "presence_below_1e-16": round(float((pres < 1e-16).float().mean()), 4),
            "pose_scale_median": round(float(lin.median()), 4),
            "pose_scale_min": round(float(lin.min()), 4)}


def brief_state(st):
    return {"presence_median": float("%.3g" % st["presence_median"]),
            "presence_below_1e-16": st["presence_below_1e-16"],
            "pose_scale_median": st["pose_scale_median"]}


def structured_leg(device, n_steps=600):
    """cfg-2 trained on STRUCTURED synthetic images (data.stroke_batches: ten
    stroke glyphs under random affine warps) instead of U[0,1) noise: the same
    replayed step, timed over `n_steps` steps in blocks of 50, with the state
    of the part capsules before / after next to it (and, for contrast, after
    the same number of steps on noise)."""
    from torch_scae_amd.data import stroke_batches
    cfg = CONFIGS["mnist_24_24_bs128"]
    B = cfg["batch"]
    out = {"workload": "mnist_24_24_bs128 on structured synthetic images "
                       "(data.stroke_batches: 10 stroke glyphs, random affine "
                       f"warps), {n_steps} steps", "steps": n_steps}
    for kind in ("strokes", "noise"):
        step = make_step(cfg, device, seed=0)
        if kind == "strokes":
            images, labels = stroke_batches(32, B, cfg["model"]["image_shape"],
                                            seed=3000, device=device)
        else:
            images, labels = synthetic_batches(cfg, device, 3000, n_batches=32)
        n = images.shape[0]
        before = capsule_state(step.model, images[0])
        first = float(step(images[0], labels[0]))
        blocks = timed_blocks(step, images, labels, 50, 0, n_steps // 50,
                              torch.cuda.synchronize)
        t = timing_summary(blocks, 50)
        leg = {"ms_per_step": t["median_ms"],
               "images_per_sec": round(B / t["median_ms"] * 1e3, 1),
               "timing": t, "loss_first": round(first, 3),
               "loss_last": round(float(step.loss), 3),
               "capsules_before": before,
               "capsules_after": capsule_state(step.model, images[n - 1])}
        if kind == "strokes":
            out.update(leg)
        else:
            out["same_steps_on_uniform_noise"] = leg
        del step
        torch.cuda.empty_cache()
    return out


def extra_workloads(device, budget_s=25.0):
    """The other BASELINE.json configurations and step variants, one short
    measurement each on this GPU (same timing procedure, 5 blocks of 20 steps
    after 5 warm-ups): ms/step, images/s, the dominant hand-written kernel of
    that workload and its roofline fraction.  Bounded: a leg is skipped once
    the budget is spent."""
    legs = [
        ("mnist_24_24_bs128 eager render (SCAE.forward's full contract: "
         "transformed_templates + mixing_logits materialised every step)",
         "mnist_24_24_bs128", dict(lazy_render=False), {}),
        ("mnist_24_24_bs128 reconstruct_alternatives=True (SCAE ctor default; "
         "3 more no-grad reconstructions incl. B x n_obj_caps images, eager "
         "render)", "mnist_24_24_bs128", dict(lazy_render=False),
         dict(alternatives=True)),
        ("mnist_40_32_bs128 (the reference's hydra default)",
         "mnist_40_32_bs128", {}, {}),
        ("cifar_32_32_bs256 (BASELINE configs[4])", "cifar_32_32_bs256", {},
         {}),
        ("mnist_48_64_bs1024 fp32 (BASELINE configs[2]'s shape)",
         "mnist_48_64_bs1024", {}, {}),
        ("mnist_48_64_bs1024 bf16 (BASELINE configs[2])",
         "mnist_48_64_bs1024", dict(autocast_dtype=torch.bfloat16), {}),
    ]
    out, t_start = [], time.perf_counter()
    for label, wl, step_kw, model_kw in legs:
        if time.perf_counter() - t_start > budget_s:
            out.append({"workload": label, "skipped": "time budget"})
            continue
        cfg = CONFIGS[wl]
        bf16 = "autocast_dtype" in step_kw
        step = make_step(cfg, device, **model_kw, **step_kw)
        images, labels = synthetic_batches(cfg, device, 2000, n_batches=2)
        blocks = timed_blocks(step, images, labels, 20, 5, 5,
                              torch.cuda.synchronize)
        t = timing_summary(blocks, 20)
        leg = {"workload": label, "ms_per_step": t["median_ms"],
               "images_per_sec": round(cfg["batch"] / t["median_ms"] * 1e3, 1),
               "timing": t, "final_loss": round(float(step.loss), 3)}
        del step
        r = roofline(cfg, device, bf16=bf16, light=True)
        leg["dominant_kernel"] = {
            "kernel": r["kernel"], "bound": r["bound"],
            "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"],
            "frac": r["frac"], "us_per_step": round(
                r["us_per_launch"] * r["launches_per_step"], 1)}
        leg["k1"] = {k: {"us": v["us"], "frac": v["frac"]}
                     for k, v in r["other_kernels"].items() if "us" in v}
        out.append(leg)
        torch.cuda.empty_cache()
    return out


def guarded(fn, seconds):
    """fn() on a worker thread, given up after `seconds`: (result, error).  A call that
    does not return (a collective captured into a HIP graph that stalls on real ranks) leaves
    its thread behind -- a daemon, abandoned at exit -- and an error here, instead of a rank
    that waits until the watchdog ends it."""
    import threading
    box = {}

    def run():
        try:
            box["out"] = fn()
        except Exception as e:      # noqa: BLE001 (recorded, not raised)
            box["err"] = repr(e)[:200]
    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        return None, f"no result after {seconds:g} s (guarded probe given up)"
    return box.get("out"), box.get("err")


def probe_modes(make, which, images, labels, barrier, reduce_max, steps=40,
                warmup=10, blocks=3, agree=None, guard=(), guard_s=30.0):
    """ms per step of the collective modes `which` of TrainStep on the ranks at
    hand, each on a FRESH model (the step time depends on how far training has
    got, see capsule_state): the contract's timing procedure (barrier +
    synchronize around exactly `steps` steps, MAX over ranks), median of
    `blocks`.  Every rank gets the same numbers, hence the same choice.
    "off": the collective-free build of the same step (its ranks drift apart,
    so it runs last).

    A mode may fail on ONE rank only (memory, a graph capture with the
    collective inside): ``agree(ok) -> bool`` (an all-reduce over a gloo group,
    which does not depend on the state of the RCCL communicator) makes every
    rank see the failure BEFORE the mode's first collective-bearing timed
    region, so nobody is left waiting in a barrier; a mode whose build failed
    anywhere is recorded with its error and skipped everywhere.  A failure
    inside the timed region itself cannot be agreed on (the other ranks are
    already inside RCCL); it is recorded, and the remaining probes are skipped
    -- the communicator may be unusable after a half-captured collective."""
    out, broken = {}, False
    for mode in which:
        if broken:
            out[mode] = {"ms_per_step": None,
                         "error": "skipped: an earlier mode failed inside its "
                                  "timed region"}
            continue
        step, err = None, None

        def build(mode=mode, probe=mode in guard):
            torch.cuda.set_device(images.device)   # (the current device is per host thread)
            st = make(mode)
            st.prepare(images[0], labels[0])       # build + capture, no step
            torch.cuda.synchronize()
            if probe:                              # ... and two replays with the collective
                for i in range(2):
                    st(images[i % len(images)], labels[i % len(labels)])
                torch.cuda.synchronize()
            return st
        if mode in guard:
            # a mode that has never run on real ranks (the all-reduce captured INSIDE the
            # graph): built and stepped twice under a time limit, every rank at once
            step, err = guarded(build, guard_s)
        else:
            try:
                step = build()
            except Exception as e:      # a mode this stack cannot build
                err = repr(e)[:200]
        ok = agree(err is None) if agree is not None else err is None
        if not ok:
            out[mode] = {"ms_per_step": None,
                         "error": err or "failed to build on another rank"}
            del step
            torch.cuda.empty_cache()
            continue
        try:
            t = timed_blocks(step, images, labels, steps, warmup, blocks,
                             barrier, reduce_max)
            out[mode] = {"ms_per_step": round(float(np.median(t)) / steps * 1e3, 4),
                         "runs_as": step.collective_mode}
        except Exception as e:
            out[mode] = {"ms_per_step": None, "error": repr(e)[:200]}
            broken = True
        del step
        torch.cuda.empty_cache()
    return out


def comm_diagnostics(step, images, labels, steps, barrier, device, world):
    """N > 1 (or --force-spawn): the collective by itself, from HIP events on
    this rank's streams.  allreduce_us: the flat-gradient all-reduce(s) alone
    (mean of 20, back to back, nothing to overlap); step_ms: the timed step
    once more, per rank.  What the collective costs the step (graph split and
    stream hand-offs included) is in ``modes``: every mode and the
    collective-free build, each measured on a fresh model."""
    n = images.shape[0]

    def run(st, k):
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), \
            torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(k):
            st(images[i % n], labels[i % n])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k        # ms per step on this rank

    with_comm = run(step, steps)
    buckets = [0, 1] if step.split else [None]
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), \
        torch.cuda.Event(enable_timing=True)
    reps = 20
    for _ in range(3):
        for b in buckets:
            step._reduce(b)
    torch.cuda.synchronize()
    barrier()
    e0.record()
    for _ in range(reps):
        for b in buckets:
            step._reduce(b)
    e1.record()
    torch.cuda.synchronize()
    allreduce_us = e0.elapsed_time(e1) / reps * 1e3
    mine = torch.tensor([with_comm, allreduce_us], device=device,
                        dtype=torch.float64)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    every = torch.stack(every).cpu()
    nbytes = step.flat.numel * 4
    return {
        "mode": step.collective_mode,
        "gradient_bytes": nbytes,
        "bucket_bytes": [step.flat.n_front * 4,
                         (step.flat.numel - step.flat.n_front) * 4]
        if step.split else [nbytes],
        "allreduce_us": round(float(every[:, 1].max()), 1),
        "allreduce_busbw_GBps": round(
            2 * (world - 1) / max(1, world) * nbytes
            / (float(every[:, 1].max()) * 1e-6) / 1e9, 1),
        "step_ms": round(float(every[:, 0].max()), 4),
        "per_rank_ms": [round(float(v), 4) for v in every[:, 0]],
        "per_rank_allreduce_us": [round(float(v), 1) for v in every[:, 1]],
        "note": "HIP-event times per rank over %d steps; busbw = 2(N-1)/N x "
                "bytes / allreduce time (ring convention)" % steps,
    }


def main():
    args = parse()
    if "RANK" not in os.environ and (args.gpus > 1 or args.force_spawn):
        return launch_ranks(args)       # nothing has touched the GPU yet
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ONE JSON line on stdout: libraries that print banners there (RCCL's version block) go to
    # stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.watchdog_s > 0:
        # a rank that waits for a collective its peers never enter would wait forever: give up
        # loudly instead (the launcher then ends the other ranks)
        import threading

        def _give_up():
            print(f"[bench rank {rank}] watchdog: no result after {args.watchdog_s} s",
                  file=sys.stderr, flush=True)
            os._exit(3)
        t = threading.Timer(args.watchdog_s, _give_up)
        t.daemon = True
        t.start()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    collective = world > 1 or args.force_spawn
    rccl_ranks = None
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device, rank=rank,
                                world_size=world)
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)           # how many ranks RCCL really connects
        rccl_ranks = int(ones.item())
        assert rccl_ranks == dist.get_world_size() == world

    cfg = CONFIGS[args.workload]
    B = cfg["batch"]
    lazy = not args.eager_render
    images, labels = synthetic_batches(cfg, device, 1000 + rank)

    def make(mode):
        return make_step(cfg, device, alternatives=args.alternatives,
                         use_graph=not args.no_graph,
                         # (the capture keeps the library's launch list beside the graph;
                         # which of the two re-issues the step is chosen below)
                         replay="launches" if not collective and not args.no_graph
                         and args.replay != "graph" else "graph",
                         optimizer=not args.no_optimizer,
                         autocast_dtype=torch.bfloat16 if args.bf16 else None,
                         force_collective=args.force_spawn,
                         lazy_render=lazy, collective_mode=mode)

    def barrier():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(block_s):
        if not collective:
            return block_s
        t = torch.tensor(block_s, device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    # how the gradient all-reduce is scheduled: measured, not assumed
    mode, modes = None, None
    if collective:
        from torch_scae_amd.train_step import TrainStep
        # host-side agreement between the ranks goes over gloo: it must work
        # whatever state a failed probe has left the RCCL communicator in
        ctl = dist.new_group(backend="gloo")

        def agree(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=ctl)
            return bool(flag.item())

        mode = "1 bucket" if args.no_overlap else args.comm_mode
        # auto: with real ranks the two schedules that keep RCCL outside the HIP graph -- the
        # all-reduce captured INSIDE the graph has only ever run on a 1-rank group here, and a
        # capture that stalls on 8 ranks would cost the whole measurement; --comm-mode
        # "in graph" (or a 1-rank group) still measures it
        # (round 6) ... so with real ranks it is tried LAST, behind a guard: build + capture +
        # two replays on a worker thread with a time limit, and only if EVERY rank came
        # through (the gloo agreement) is it timed and allowed into the automatic choice.
        # A probe that stalls leaves the two measured schedules to choose from.
        auto = TrainStep.MODES if world == 1 else tuple(
            m for m in TrainStep.MODES if m != "in graph") + ("in graph",)
        which = auto if mode == "auto" else (mode,)
        guard = ("in graph",) if world > 1 and mode == "auto" else ()
        # ("off" -- the collective-free build, whose ranks drift apart -- before the guarded
        # probe: nothing is measured after a mode that may leave the communicator wedged)
        order = tuple(m for m in which if m not in guard) + ("off",) + guard
        modes = probe_modes(make, order, images, labels, barrier,
                            reduce_max, agree=agree, guard=guard,
                            guard_s=args.in_graph_probe_s)
        # per-rank diagnostics on stderr (rank 0's stdout carries the one JSON
        # line): what every rank measured, also when a mode raised
        print(f"[bench rank {rank}/{world}] rccl_ranks={rccl_ranks} "
              f"comm.modes={json.dumps(modes)}", file=sys.stderr, flush=True)
        usable = [m for m in which if modes[m]["ms_per_step"]]
        if not usable:
            raise SystemExit(f"rank {rank}: no collective mode ran: {modes}")
        if mode == "auto" or mode not in usable:
            mode = min(usable, key=lambda m: modes[m]["ms_per_step"])
    step = make(mode)

    # The state the headline is quoted on (parameters at the
    # build's init, noise on): the step is captured first (the capture's
    # warm-ups touch neither parameters nor optimiser state), its training
    # state snapshotted, and put back -- outside the timed region -- before
    # every timed block.  U[0,1) images carry nothing to model and the part
    # capsules switch off within a few hundred RMSprop steps; a block that
    # started wherever the previous one ended would time the data-dependent K1
    # backward at its zero-gradient exit (round 4's headline did).
    step.prepare(images[0], labels[0])
    snap = step.snapshot()
    state0 = capsule_state(step.model, images[0])
    per_block = []

    def before_block(b):
        step.restore(snap)

    def after_block(b):
        per_block.append(brief_state(capsule_state(step.model, images[0])))

    # graph replay or the launch list: the same launches either way; which is faster
    # depends on the box (the graph costs ~9 us on the device per replay, the list ~220 -
    # 360 us of host time per step beside a ~550 us step), so it is measured
    replay_probe = None
    has_list = bool(getattr(step, "_klist", None))
    if not has_list:
        step.replay = "graph"
    elif args.replay == "auto":
        replay_probe = {}
        for how in ("graph", "launches"):
            step.replay = how
            t = timed_blocks(step, images, labels, args.steps, args.warmup, 3, barrier,
                             reduce_max, before_block=before_block,
                             refresh=(lambda: step.restore(snap))
                             if args.steps > LIVE_SPAN else None)
            replay_probe[how] = round(float(np.median(t)) / args.steps * 1e3, 4)
        step.replay = min(replay_probe, key=replay_probe.get)
    else:
        step.replay = args.replay
    # the timed region of the contract -- W warm-ups, then EXACTLY K steps
    # between barrier + synchronize, MAX over ranks -- repeated `blocks` times
    # back to back, each from the restored state; the reported step time is
    # the MEDIAN block
    blocks = timed_blocks(step, images, labels, args.steps, args.warmup,
                          max(1, args.blocks), barrier, reduce_max,
                          before_block=before_block, after_block=after_block,
                          refresh=(lambda: step.restore(snap))
                          if args.steps > LIVE_SPAN else None)
    timing = timing_summary(blocks, args.steps)
    timing["state"] = (
        f"parameters + optimiser state restored to the build's init before "
        f"every block (outside the timed region): every timed step is within "
        f"{args.warmup} + {min(args.steps, LIVE_SPAN)} RMSprop steps of init"
        + (f"; blocks of {args.steps} steps also restore it every {LIVE_SPAN} "
           f"steps inside the timed region (3 device copies of the flat "
           f"buffers, counted in the time)" if args.steps > LIVE_SPAN else ""))
    timing["per_block"] = [
        dict(ms=round(1e3 * t / args.steps, 4), capsules_after=st)
        for t, st in zip(blocks, per_block)]
    # the same captured step re-issued from the library's record of its kernel launches
    # (TrainStep(replay="launches"), scae_launch_list_run: a hipLaunchKernel per launch, no
    # per-replay graph cost on the device)
    timing["replay"] = step.replay if step.use_graph else "eager"
    if replay_probe is not None:
        timing["replay_probe_ms"] = replay_probe
    if not collective and step.use_graph and has_list:
        chosen = step.replay
        other = "graph" if chosen == "launches" else "launches"
        step.replay = other
        t3 = timing_summary(timed_blocks(
            step, images, labels, args.steps, args.warmup, max(1, min(7, args.blocks)),
            barrier, reduce_max, before_block=before_block,
            refresh=(lambda: step.restore(snap))
            if args.steps > LIVE_SPAN else None), args.steps)
        step.replay = chosen
        timing["replay_as_launch_list_ms" if other == "launches"
               else "replay_as_graph_ms"] = t3["median_ms"]
    in_step = None
    if not collective and has_list and not args.no_roofline:
        step.restore(snap)
        for i in range(args.warmup):
            step(images[i % len(images)], labels[i % len(labels)])
        torch.cuda.synchronize()
        in_step = step_timeline(step)
    final_loss = float(step.loss)
    final_state = capsule_state(step.model, images[0])
    # "optimizer step reported separately" (the reference's step
    # is base_experiment.py:109-126 + its optimiser): the same step captured
    # without the RMSprop launch, same procedure (its state never moves)
    no_opt = None
    if not args.no_optimizer:
        st2 = make_step(cfg, device, alternatives=args.alternatives,
                        use_graph=not args.no_graph, optimizer=False,
                        autocast_dtype=torch.bfloat16 if args.bf16 else None,
                        force_collective=args.force_spawn, lazy_render=lazy,
                        collective_mode=mode)
        t2 = timing_summary(timed_blocks(
            st2, images, labels, args.steps, args.warmup,
            max(1, min(7, args.blocks)), barrier, reduce_max), args.steps)
        no_opt = {"ms_per_step": t2["median_ms"],
                  "images_per_sec": round(B * world / t2["median_ms"] * 1e3, 1),
                  "timing": t2,
                  "step": "forward + SCAE.loss + backward"
                          + (" + RCCL all-reduce" if collective else "")
                          + ", no optimiser launch (parameters stay at init)"}
        del st2
        torch.cuda.empty_cache()
    comm = None
    if collective:
        comm = comm_diagnostics(step, images, labels, max(10, args.steps),
                                barrier, device, world)
        comm["modes"] = modes
        comm["chosen"] = mode
        # same stage of training on both sides (fresh models)
        comm["step_no_comm_ms"] = modes["off"]["ms_per_step"]
        if modes[mode]["ms_per_step"] and modes["off"]["ms_per_step"]:
            comm["exposed_us"] = round((modes[mode]["ms_per_step"]
                                        - modes["off"]["ms_per_step"]) * 1e3, 1)
        print(f"[bench rank {rank}/{world}] chosen={mode!r} "
              f"allreduce_us={comm['allreduce_us']} "
              f"exposed_us={comm.get('exposed_us')} "
              f"step_ms={comm['step_ms']} per_rank_ms={comm['per_rank_ms']}",
              file=sys.stderr, flush=True)

    result = None
    if rank == 0:
        ms = timing["median_ms"]
        result = {
            "metric": metric_name(args.workload),
            "value": round(B * world / ms * 1e3, 1),
            "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 operands / f32 accumulate on the GEMM-shaped kernels "
                     "(K7, K8), f32 elsewhere" if args.bf16 else "f32",
            "data": "synthetic",
            "timing": timing,
            "config": {
                "workload": args.workload, "per_gpu_batch": B,
                "global_batch": B * world,
                "step": "forward + SCAE.loss + backward"
                        + (f" + RCCL all-reduce of the flat fp32 gradient "
                           f"buffer ({step.collective_mode})"
                           if collective else "")
                        + ("" if args.no_optimizer else " + RMSprop step"),
                "hip_graph": not args.no_graph,
                # the step computes the loss from the fused likelihood kernel;
                # the (B, M+1, C, H, W) transformed_templates / mixing_logits
                # of SCAE.forward's result are rendered on first access (bit
                # identical), not in every step -- extra_workloads has the
                # step that materialises them
                "lazy_render": lazy,
                "reconstruct_alternatives": bool(args.alternatives),
                "parallelism": f"dp{world}", "rccl_ranks": rccl_ranks,
                "final_loss": round(final_loss, 3),
                # the state of the part capsules the data-dependent kernels
                # (K1's backward) ran on: at the restored init / after the
                # last timed block
                "capsules_before": state0,
                "capsules_after": final_state,
                "data_note": "U[0,1) noise images, parameters put back to the "
                             "build's init before every timed block "
                             "(timing.state): every part capsule is live in "
                             "every timed step.  Left to train on noise the "
                             "capsules switch off within a few hundred steps "
                             "and K1's backward skips them (exact zeros): "
                             "extra_workloads' last leg times that collapsed "
                             "state (same_steps_on_uniform_noise) next to the "
                             "step on structured images",
            },
        }
        if no_opt is not None:
            result["step_without_optimizer"] = no_opt
        if comm is not None:
            result["comm"] = comm
        if not args.no_roofline:
            result["roofline"] = roofline(cfg, device, bf16=args.bf16)
            if in_step is not None:
                # the dominant kernel IN the step: its launches' durations between HIP events
                # on the step's own stream (round 5's review: the stand-alone loop flatters it)
                r = result["roofline"]
                pair = [l["us"] for l in in_step if "conv3x3_bwd_pair" in l["name"]
                        or "conv3x3_dgrad_bf16r" in l["name"]
                        or "conv3x3_wgrad_bf16r" in l["name"]]
                # (bf16-resident: a layer's data and weight gradient are two launches)
                if len(pair) in (r["launches_per_step"], 2 * r["launches_per_step"]):
                    fl_all = r["algorithmic_flops_per_launch"] * len(pair)
                    r["standalone"] = dict(achieved=r["achieved"], frac=r["frac"],
                                           us_per_launch=r["us_per_launch"],
                                           per_layer_us=r["per_layer_us"])
                    r["achieved"] = round(fl_all / (sum(pair) * 1e-6) / 1e12, 1)
                    r["frac"] = round(r["achieved"] / r["peak"], 4)
                    r["us_per_launch"] = round(sum(pair) / r["launches_per_step"], 2)
                    r["per_layer_us"] = pair
                    r["measured"] = ("in the step: a HIP event in front of and behind every "
                                     "launch of one real step on the step's stream "
                                     "(scae_launch_list_timeline; `standalone`: back-to-back "
                                     "launches of the kernel alone)")
                r["step_launches"] = in_step
            fl, by = step_algorithmic(cfg)
            ips = result["value"] / world        # per GPU
            result["roofline"]["step"] = {
                "flops_per_image": fl, "hot_bytes_per_image": by,
                "mfma_frac": round(ips * fl / 1e12 / MFMA_FP32_PEAK_TFLOPS, 4),
                "mfma_peak": MFMA_FP32_PEAK_TFLOPS,
                "hbm_frac": round(ips * by / 1e9 / HBM_PEAK_GBS, 5),
                "note": "whole step per GPU: images/s x SURVEY.md 8d's "
                        "algorithmic FLOPs (bytes) per image / fp32 MFMA "
                        "(HBM) peak"}
        if world == 1 and not args.force_spawn and not args.no_extra \
                and args.workload == "mnist_24_24_bs128" and not args.bf16 \
                and not args.no_graph and not args.no_optimizer \
                and lazy and not args.alternatives:
            del step
            torch.cuda.empty_cache()
            result["extra_workloads"] = extra_workloads(device)
            result["extra_workloads"].append(structured_leg(device))
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, args.cpu_steps)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
