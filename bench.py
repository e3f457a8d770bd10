#!/usr/bin/env python
"""AR-FPN + AR-RFF forward+backward throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by the driver through torch.distributed.run, one rank per GPU, RCCL over xGMI)

A step = one pass of the hot path over one batch of synthetic input, per GPU:
    5-level 256-ch pyramid from a 1333x800 image (padded 1344x800), B = 2 images  -> AR-FPN (WFPNDualSpatial)
    -> 3-RoI RoIAlign concat of R = 512*B RoIs -> AR-RFF head (MultiRoIsBBoxHead) -> backward of both
    -> gradient all-reduce(mean) over ranks.
Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact fp32 (ARFE_GEMM_F32=1)
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 MFMA; the default contractions issue 3 bf16 MFMAs per fp32 product
PEAK_HBM_GBS = 8000.0            # spec; ~6300 GB/s achievable by a float4 copy
ARFPN_HBM_BYTES_PER_IMG = 458.4e6   # fwd 2 x 91.67 MB + bwd 3 x 91.67 MB


def path_roofline(batch, rois_per_img, step_s):
    """SURVEY §8(d)'s whole-path figure: t_roof = sum over stages of max(bytes / 8 TB/s, flops / peak) from the table's
    ALGORITHMIC work per image (AR-FPN streaming stages 458.4 MB; NonLocal 20.4 + 49 GFLOP; AR-RFF convs + FCs
    103.5 + 207 GFLOP at 512 RoIs/img), achieved = t_roof / t_measured - against the exact-fp32 MFMA peak (what an fp32
    implementation is bounded by) and against the dense bf16 peak (the pipe the bf16x3 contractions run on)."""
    gflop = 69.4 + 310.5 * rois_per_img / 512.0
    t_hbm = ARFPN_HBM_BYTES_PER_IMG / (PEAK_HBM_GBS * 1e9)
    t32 = batch * (t_hbm + gflop * 1e9 / (PEAK_F32_MFMA_TFLOPS * 1e12))
    t16 = batch * (t_hbm + gflop * 1e9 / (PEAK_BF16_MFMA_TFLOPS * 1e12))
    return {"definition": "t_roof / t_measured, SURVEY 8(d) algorithmic bytes and flops per image",
            "algorithmic_gflop_per_img": round(gflop, 1), "algorithmic_hbm_bytes_per_img": ARFPN_HBM_BYTES_PER_IMG,
            "t_roof_ms_fp32_mfma": round(t32 * 1e3, 4), "achieved_vs_fp32_mfma": round(t32 / step_s, 4),
            "t_roof_ms_bf16_mfma": round(t16 * 1e3, 4), "achieved_vs_bf16_mfma": round(t16 / step_s, 4)}


def pmc_traffic(kernel_substr):
    """(HBM bytes per launch, source note) of the dominant kernel from the newest committed rocprofv3 PMC passes
    (FETCH_SIZE doubled as the gfx950 note in MI355X_MICROARCH.md prescribes, + WRITE_SIZE; tools/pmc_run.sh +
    tools/pmc_traffic.py on this same bench command).  PMC passes cannot run inside this process, so the figure is
    quoted from the committed profile and ONLY while the profile was taken on the kernel sources this run uses
    (`_meta.csrc_sha` == arfe_amd.build.sources_sha()); otherwise traffic is null and the note says why."""
    import glob
    from arfe_amd.build import sources_sha
    pmc_traffic.mfma = None
    paths = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_traffic.json')))
    if not paths:
        return None, "no committed PMC profile"
    path = paths[-1]
    name = os.path.relpath(path, ROOT)
    try:
        d = json.load(open(path))
        meta = d.get("_meta", {})
        if meta.get("csrc_sha") != sources_sha():
            return None, f"{name} was measured on other kernel sources (csrc sha {meta.get('csrc_sha')}): not quoted"
        for k, v in d["kernels"].items():
            if all(t in k for t in kernel_substr) and 'EpiConv, 1' not in k:
                pmc_traffic.mfma = v.get("mfma")
                return v["hbm_bytes_per_launch"], f"bytes/launch, PMC passes in {name} (csrc sha {meta['csrc_sha']})"
    except Exception as e:      # a malformed profile must not take the bench line down
        return None, f"{name}: {type(e).__name__}"
    return None, f"{name} has no such kernel"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=2, help='images per GPU (BASELINE config: bs=2)')
    ap.add_argument('--rois-per-img', type=int, default=512)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-iters', type=int, default=5)
    ap.add_argument('--no-fp16-leg', action='store_true', help='skip the second (fp16 mode) leg of the default run')
    ap.add_argument('--fp16', action='store_true',
                    help="the reference's fp16 mode (wrap_fp16_model): half storage at module boundaries, fp32 kernels")
    return ap.parse_args()


def build_modules(dev):
    import arfe_amd
    torch.manual_seed(1234)     # identical parameters on every rank
    neck = arfe_amd.build_neck(dict(type='WFPNDualSpatial', in_channels=256, num_levels=5))
    neck.init_weights()
    roi_head = arfe_amd.build_head(dict(
        type='StandardRoIHead',
        bbox_roi_extractor=dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=7, sample_num=0),
                                out_channels=256, featmap_strides=[4, 8, 16, 32, 64]),
        bbox_head=dict(type='MultiRoIsBBoxHead', in_channels=256, fc_out_channels=1024, roi_feat_size=7,
                       num_classes=80,
                       bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.],
                                       target_stds=[0.1, 0.1, 0.2, 0.2]),
                       reg_class_agnostic=False, loss_cls=dict(type='CrossEntropyLoss'), loss_bbox=dict(type='L1Loss'))))
    roi_head.init_weights()
    return neck.to(dev), roi_head.to(dev)


def host_cpu_share():
    """Cores this process may use: the scheduler affinity, cut by a cgroup CPU quota if one is set (os.cpu_count() reports the
    whole host on the GPU box; the 1-GPU lease is a 16-core share).  ARFE_CPU_THREADS overrides."""
    if os.environ.get('ARFE_CPU_THREADS'):
        return max(1, int(os.environ['ARFE_CPU_THREADS']))
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    elif n > 32:
        n = 16        # no affinity mask and no quota visible: stay on the 1-GPU box's documented 16-core share
    return n


def cpu_baseline(batch, rois_per_img, iters, neck_state, head_state, device_run=None):
    """The oracle (CPU restatement, kind='port') on the host cores, same synthetic workload.
    device_run: callable -> the device's outputs / gradients / decisions of the same step (oracle.parity.device_step); the
    oracle pass this leg runs anyway is then used as the checker too -> (cpu_baseline dict, parity dict or None)."""
    from oracle import arfe_oracle as O
    from oracle import parity as P
    threads = host_cpu_share()
    torch.set_num_threads(threads)
    O.set_roi_align_threads(threads)
    pn = {k: v.clone().requires_grad_(True) for k, v in neck_state.items()}
    ph = {k: v.clone().requires_grad_(True) for k, v in head_state.items()}
    xs = [x.requires_grad_(True) for x in O.synth_pyramid(batch, 256, seed=1234)]
    rois = O.synth_rois(batch, rois_per_img, seed=1234)
    g = torch.Generator().manual_seed(99)
    c1 = torch.randn(rois.shape[0], 81, generator=g)
    c2 = torch.randn(rois.shape[0], 320, generator=g)
    last = {}

    def step():
        for t in xs + list(pn.values()) + list(ph.values()):
            t.grad = None
        feats, parts_n = O.arfpn_forward(xs, pn, 2, return_parts=True)
        cls, reg, parts_h = O.rff_head(O.roi_extract3(feats, rois), ph, return_parts=True)
        torch.autograd.backward([cls, reg], [c1, c2])
        last.update(cls=cls, reg=reg, parts_n=parts_n, parts_h=parts_h)

    step()   # warm-up
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    base = {"value": round(batch / med, 4), "unit": "imgs/s", "cores": threads, "kind": "port",
            "sample": f"{iters} timed + 1 warm-up fwd+bwd passes of the same workload (B={batch}, R={rois.shape[0]}), "
                      f"median {med:.2f} s/step; oracle/arfe_oracle.py (torch-CPU fp32) + oracle C RoIAlign, {threads} threads each"}
    parity = None
    if device_run is not None:
        # the checker: this leg's last free-running pass + one pass under the device's ReLU / argmax choices (oracle/parity.py)
        free = dict(cls=last['cls'].detach(), reg=last['reg'].detach(), dxs=[x.grad for x in xs],
                    dneck={k: v.grad for k, v in pn.items()}, dhead={k: v.grad for k, v in ph.items()},
                    decisions=O.decisions_of(xs, last['parts_n'], last['parts_h'], 2))
        rep = P.compare(device_run(P), [x.detach() for x in xs], rois, {k: v.detach() for k, v in pn.items()},
                        {k: v.detach() for k, v in ph.items()}, c1, c2, free=free)
        parity = {k: (float(f"{v:.3e}") if isinstance(v, float) else v) for k, v in rep.items()}
    return base, parity


def self_launch(args):
    """`bench.py --gpus N` (N > 1) started WITHOUT a launcher: start the N ranks here as a child torch.distributed.run job
    (reference: tools/dist_train.sh:1-9) and relay rank 0's JSON line.  Runs before anything touches the GPU in this
    process; the child is a subprocess, never an exec."""
    import socket
    import subprocess
    backend = os.environ.get('ARFE_DIST_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()            # (does not initialise the GPU on this image)
    if backend == 'nccl' and ndev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs over RCCL, this node shows {ndev} "
              f"(ARFE_DIST_BACKEND=gloo rehearses the N > 1 path with ranks sharing GPUs)", file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(r.stdout)
    sys.stdout.flush()
    return r.returncode


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        # never measure fewer GPUs than the line claims
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` (it launches "
                 f"its own ranks) or through torch.distributed.run with --nproc-per-node equal to --gpus")
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # one rank per GPU; ARFE_DIST_BACKEND=gloo lets several ranks share one GPU to rehearse the N>1 code path
    backend = os.environ.get('ARFE_DIST_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count()
    dev = torch.device('cuda', dev_index)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    if world > 1:
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        # communicator set-up (lazy in RCCL) must not land in the timed region even with --warmup 0
        dist.all_reduce(torch.zeros(1, device=dev))
        torch.cuda.synchronize()

    from arfe_amd import functional as Fn
    from arfe_amd.dist import GradAllReducer
    from arfe_amd import synth as O          # synthetic inputs; oracle/ is imported by the cpu_baseline leg only

    def build_workload(fp16):
        """modules + synthetic inputs + the step closure of one precision mode"""
        neck, roi_head = build_modules(dev)
        neck_state = {k: v.detach().cpu().clone() for k, v in neck.state_dict().items()}
        head_state = {k: v.detach().cpu().clone() for k, v in roi_head.bbox_head.state_dict().items()}
        xs = [x.to(dev) for x in O.synth_pyramid(B, 256, seed=1234 + rank)]
        if fp16:                                        # SURVEY 8(d) fp16 variant: same tensors .half()
            import arfe_amd
            arfe_amd.wrap_fp16_model(neck)
            arfe_amd.wrap_fp16_model(roi_head)
            xs = [x.half() for x in xs]
        xs = [x.requires_grad_(True) for x in xs]
        rois = O.synth_rois(B, args.rois_per_img, seed=1234 + rank).to(dev)
        g = torch.Generator().manual_seed(99 + rank)
        cot_cls = torch.randn(R, 81, generator=g).to(dev)
        cot_reg = torch.randn(R, 320, generator=g).to(dev)
        if fp16:
            cot_cls, cot_reg = cot_cls.half(), cot_reg.half()
        # gradients become ready head-first, then the neck
        # fc1's weight gradient (51 of the head's 64 MB) exists 2 ms before the step ends: it goes on the wire from the kernel
        # that produces it instead of with the rest of the head's bucket
        reducer = GradAllReducer([list(roi_head.bbox_head.parameters()), list(neck.parameters())],
                                 early=[roi_head.bbox_head.shared_fcs[0].weight])

        def step():
            reducer.zero_grad()
            for x in xs:
                x.grad = None
            feats = neck(xs)
            res = roi_head._bbox_forward(feats, rois)
            torch.autograd.backward([res['cls_score'], res['bbox_pred']], [cot_cls, cot_reg])
            reducer.synchronize()
        return dict(step=step, neck=neck, roi_head=roi_head, xs=xs, rois=rois, cot_cls=cot_cls, cot_reg=cot_reg,
                    neck_state=neck_state, head_state=head_state, reducer=reducer)

    B, R = args.batch, args.batch * args.rois_per_img
    wl = build_workload(args.fp16)
    step, neck, roi_head, xs, rois = wl['step'], wl['neck'], wl['roi_head'], wl['xs'], wl['rois']
    cot_cls, cot_reg, neck_state, head_state = wl['cot_cls'], wl['cot_reg'], wl['neck_state'], wl['head_state']

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # HIP-event brackets inside the timed region: only the dominant kernel's launches (the `roofline` object); every
    # bracket is two event records on the launch stream, so the stage break-down is taken from extra steps afterwards
    Fn.probe_enable(True, only=('conv3x3_fwd', 'conv3x3_bwd_data'))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    probes = Fn.probe_results_ms()
    stage_steps = min(args.steps, 10)
    Fn.probe_enable(True)
    for _ in range(stage_steps):
        step()
    fence()
    stage_probes = Fn.probe_results_ms()
    Fn.probe_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # second leg (N = 1, default flags only): the SAME step in the reference's fp16 mode (BASELINE configs[4]'s arithmetic:
    # half storage, half-operand contractions), timed after and apart from the fp32 region above
    fp16_leg = None
    if world == 1 and not args.fp16 and not args.no_fp16_leg:
        del wl, step
        wl16 = build_workload(True)
        for _ in range(max(args.warmup // 2, 3)):
            wl16['step']()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            wl16['step']()
        fence()
        e16 = time.perf_counter() - t1
        fp16_leg = {"value": round(B * args.steps / e16, 3), "unit": "imgs/s", "ms_per_step": round(e16 / args.steps * 1e3, 3),
                    "vs_fp32_line": round(elapsed / e16, 4),
                    "dtype": "f16 storage (pyramid, RoI features, parameters), f16 MFMA products, fp32 accumulate",
                    "note": "wrap_fp16_model on the same modules and inputs; not the headline value"}
        del wl16

    if rank == 0:
        imgs = world * B * args.steps
        # dominant kernel: the implicit-GEMM 3x3 conv (conv3x3_stage_once_kernel<EpiConv>, conv_stage_once.hpp),
        # 3 forward + 3 data-gradient launches per step, each 2*C*C*9 * R*49 flop
        conv_ms = probes.get('conv3x3_fwd', []) + probes.get('conv3x3_bwd_data', [])
        conv_flop = 2.0 * 256 * 256 * 9 * R * 49
        avg_ms = sum(conv_ms) / max(len(conv_ms), 1)
        ach = conv_flop / (avg_ms * 1e-3) / 1e12 if conv_ms else 0.0
        fpn_keys = ['fpn_gather_gate_fwd', 'fpn_gather_fwd', 'fpn_gatemap_fwd', 'fpn_scatter_fwd', 'fpn_scatter_bwd',
                    'fpn_input_bwd', 'fpn_gate_wgrad']
        fpn_ms = sum(sum(stage_probes.get(k, [])) for k in fpn_keys) / stage_steps
        hbm_gbs = ARFPN_HBM_BYTES_PER_IMG * B / (fpn_ms * 1e-3) / 1e9 if fpn_ms > 0 else 0.0
        nl_ms = (sum(stage_probes.get('nonlocal_fwd', [])) + sum(stage_probes.get('nonlocal_bwd', []))) / stage_steps
        f32_mode = os.environ.get('ARFE_GEMM_F32', '0') == '1'
        peak = PEAK_F32_MFMA_TFLOPS if f32_mode else PEAK_BF16_MFMA_TFLOPS
        traffic, traffic_note = (None, "exact-fp32 mode: not profiled") if f32_mode else \
            pmc_traffic(("conv3x3_stage_once_kernel",))
        out = {
            "metric": "imgs/s AR-FPN+AR-RFF fwd+bwd, 5-level 256-ch pyramid @1333x800",
            "value": round(imgs / elapsed, 3), "unit": "imgs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f16 storage at module boundaries, " if args.fp16 else "") +
                     ("f32" if f32_mode else "f32 (dense contractions as bf16x3 split MFMA, fp32 accumulate)"),
            "data": "synthetic",
            "dist": {"backend": (dist.get_backend() if world > 1 else None), "world_size": (dist.get_world_size() if world > 1 else 1),
                     "launcher": os.environ.get('TORCHELASTIC_RUN_ID') is not None},
            "config": {"workload": "Faster R-CNN R50 AR-FPN + AR-RFF hot path (BASELINE configs[2]): WFPNDualSpatial "
                                   "(refine_level 2) + 3-RoI RoIAlign + MultiRoIsBBoxHead, fwd+bwd + grad all-reduce",
                       "per_gpu_batch": B, "global_batch": world * B, "rois_per_gpu": R, "channels": 256,
                       "pyramid": "200x336,100x168,50x84,25x42,13x21", "parallelism": f"image-dp{world}"},
            "roofline": {"bound": "mfma",
                         "kernel": ("gemm_f32_kernel<LoadRowMajorK,ConvGatherB,EpiConv>" if f32_mode else
                                    "conv3x3_stage_once_kernel<EpiConv> (256x224 tile, bf16x3 MFMA)") +
                                   " (3x3 conv 256->256 on [R,256,7,7], fwd + data-grad)",
                         "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(ach / peak, 4),
                         "traffic": traffic, "traffic_unit": traffic_note,
                         "mfma_pmc": getattr(pmc_traffic, 'mfma', None),
                         "algorithmic_bytes_per_launch": int(2 * R * 256 * 49 * 4 + 9 * 256 * 256 * 4),
                         "mfma_flop_per_algorithmic_flop": 1 if f32_mode else 3,
                         "frac_of_exact_fp32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                         "flop_per_launch": conv_flop, "avg_launch_ms": round(avg_ms, 4), "launches": len(conv_ms)},
            "roofline_hbm_stages": {"bound": "hbm", "stages": "AR-FPN gather+gate+scatter fwd/bwd (NonLocal excluded); event brackets on extra steps after the timed region",
                                    "achieved": round(hbm_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": round(hbm_gbs / PEAK_HBM_GBS, 4), "ms_per_step": round(fpn_ms, 4),
                                    "algorithmic_bytes_per_img": ARFPN_HBM_BYTES_PER_IMG},
            "roofline_path": path_roofline(B, args.rois_per_img, elapsed / args.steps),
            "stage_ms_per_step": {"arfpn_hbm_stages": round(fpn_ms, 4), "nonlocal_fwd_bwd": round(nl_ms, 4),
                                  "conv3x3_fwd_and_dgrad": round(sum(conv_ms) / args.steps, 4)},
        }
        if fp16_leg is not None:
            out["fp16_mode"] = fp16_leg
        if world == 1 and not args.no_cpu_baseline and not args.fp16:
            # (outside the timed region) the CPU leg doubles as the checker of this very workload
            def device_run(P):
                return P.device_step(neck, roi_head, xs, rois, cot_cls, cot_reg)
            out["cpu_baseline"], out["parity"] = cpu_baseline(B, args.rois_per_img, args.cpu_iters, neck_state, head_state,
                                                              device_run)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
