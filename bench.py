#!/usr/bin/env python
"""bench.py -- frames/sec of the CenterTrack per-frame inference hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one frame of every stream the rank owns through the whole hot path
(BASELINE.json north_star / SURVEY.md section 8d): frame already resident in HBM -> three
stems + DLA-34 + 16 DCNv2 nodes + heads (fp32, sigmoid fused) -> NMS/top-K/gather decode
-> one packed D2H -> host post-process -> track association (and, for N > 1, the RCCL
all-gather of the packed detections).  Workload at N=1: BASELINE.json configs[1] =
MOT17-half heads, DLA-34, 512x512, batch 1 (one stream), synthetic frames, seeded
random-init weights.  Weak scaling: every rank runs the same number of streams.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      the DCNv2 kernel (dcn_mfma_kernel): algorithmic flops (2*9*Cin*Cout*h*w per
                layer) and algorithmic bytes (4*(Cin*h*w + 27*h*w + Cout*h*w + 9*Cin*Cout
                + Cout)) of the 16 DCN layers / their summed launch time, measured with HIP
                events on the launch stream (graph replay of exactly these launches) in this process;
                traffic = HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/)
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's CPU path) timed on the
                host cores for a bounded sample of the same workload (reported baseline only)
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

PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 MFMA = vector peak
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', default='mot17_512', help='workload name (centertrack_amd.scenarios.CONFIGS)')
    ap.add_argument('--streams', type=int, default=0, help='streams (batch) per GPU; 0 = 1 (the headline config)')
    ap.add_argument('--height', type=int, default=0)
    ap.add_argument('--width', type=int, default=0)
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--cpu-frames', type=int, default=3)
    ap.add_argument('--hm-gain', type=float, default=11.0)
    ap.add_argument('--pcie', action='store_true',
                    help='also report the PCIe-inclusive rates (pinned host fp32 frames; raw u8 1080p frames with the '
                         'device-side pre-processing) under "pcie_inclusive" -- never the headline value')
    return ap.parse_args()


def kernel_pass(model, plan, reps=10):
    """Average duration of the DCN / conv launches of the plan: all launches of one kind are captured
    back-to-back in a HIP graph and the replay is timed with HIP events on the launch stream (device time,
    kernel boundaries included, host launch cost excluded -- the same thing rocprofv3's kernel trace sees)."""
    import ctypes
    from centertrack_amd import _lib
    lib = _lib.load()
    stats = {}
    model._run_plan(plan)
    torch.cuda.synchronize()
    for kind in ('dcn', 'conv'):
        launches = [l for l in plan['launches'] if l.fn == kind]

        def run():
            st = _lib.stream_ptr()
            for l in launches:
                if kind == 'dcn':
                    lib.ct_dcn_v2(ctypes.byref(l.args), st)
                else:
                    lib.ct_conv2d(ctypes.byref(l.args), st)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flops = bytes_ = 0.0
        for l in launches:
            d = l.args
            if kind == 'dcn':
                hw = d.N * d.H * d.W
                flops += 2.0 * 9 * d.Cin * d.Cout * hw
                bytes_ += 4.0 * (d.Cin * hw + d.Cout * hw + 9 * d.Cin * d.Cout + d.Cout)
                if d.fuse_offset:          # offset/mask conv computed in the same launch: its flops and weights
                    flops += 2.0 * 9 * d.Cin * 27 * hw
                    bytes_ += 4.0 * (9 * d.Cin * 27 + 27)
                else:                      # the 27-channel offset/mask map is read from HBM
                    bytes_ += 4.0 * 27 * hw
            else:
                pad = d.ks // 2
                ho = (d.H + 2 * pad - d.ks) // d.stride + 1
                wo = (d.W + 2 * pad - d.ks) // d.stride + 1
                flops += 2.0 * d.ks * d.ks * d.Cin * d.Cout * d.N * ho * wo
                bytes_ += 4.0 * (d.Cin * d.N * d.H * d.W + d.Cout * d.N * ho * wo + d.ks * d.ks * d.Cin * d.Cout)
        stats[kind] = dict(launches=len(launches), flops=flops, bytes=bytes_, ms=ms)
    return stats


def pmc_traffic():
    """HBM bytes per DCN launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, made by
    tools/collect_profiles.sh from FETCH_SIZE / WRITE_SIZE with the gfx950 corrections); None if absent."""
    p = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(p) as f:
            j = json.load(f)
        return j['dcn_mfma_kernel']['hbm_bytes_per_launch'], j.get('source', p)
    except Exception:
        return None, None


def cpu_baseline(cfg, heads, sd, frames_cpu, metas, opt_kw, nframes):
    """CPU oracle (port of the reference CPU path) on the same workload, bounded sample."""
    from oracle import detector as odet
    oopt = odet.default_opt(input_h=cfg['H'], input_w=cfg['W'], num_classes=heads['hm'], **opt_kw)
    det = odet.Detector(oopt, sd, heads)
    if oopt.flip_test:
        frames_cpu = [torch.cat((f, torch.flip(f, [3])), 0) for f in frames_cpu]
    det.run(frames_cpu[0], dict(metas[0]))                   # warm-up frame (thread pools, first-touch)
    t0 = time.time()
    for i in range(nframes):
        det.run(frames_cpu[(i + 1) % len(frames_cpu)], dict(metas[0]))
    dt = time.time() - t0
    return dict(value=round(nframes / dt, 4), unit='frames/s', cores=torch.get_num_threads(), kind='port',
                sample='%d frames of the N=1 workload (1 stream, %dx%d) through oracle/detector.py after 1 warm-up '
                       'frame; pure-PyTorch CPU restatement of the reference path incl. DCNv2' % (nframes, cfg['H'], cfg['W']))


def main():
    args = parse()
    from centertrack_amd import parallel
    rank, world, local = parallel.init_from_env()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d'
                             % (args.gpus, args.gpus))
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (the hot path has no CPU fallback)')
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)

    from centertrack_amd import scenarios as S
    from centertrack_amd import weights as W
    from centertrack_amd.detector import StreamDetector, default_opt
    from centertrack_amd.image import make_meta
    from centertrack_amd.model import DLASegHIP

    cfg = dict(S.CONFIGS[args.config])
    if args.height:
        cfg['H'] = args.height
    if args.width:
        cfg['W'] = args.width
    B = args.streams if args.streams > 0 else 1
    heads = S.HEAD_SETS[cfg['heads']]
    sd = W.make_synthetic_state_dict(heads, seed=317, hm_gain=args.hm_gain)
    if 'ltrb_amodal' in heads:
        sd['ltrb_amodal.2.bias'] = torch.tensor([-3.0, -3.0, 3.0, 3.0])
    opt_kw = dict(track_thresh=cfg['track_thresh'], pre_thresh=cfg['pre_thresh'], flip_test=cfg['flip'])
    opt = default_opt(heads, **opt_kw)
    model = DLASegHIP(heads)
    model.load_state_dict(sd)
    det = StreamDetector(opt, model=model, num_streams=B, use_graph=not args.no_graph)

    # synthetic stream: a fixed N(0,1) image scrolled by 4 px / frame, T distinct frames resident in HBM
    T = 8
    g = torch.Generator().manual_seed(317 + 7 + rank)
    base = torch.randn((B, 3, cfg['H'], cfg['W'] + 4 * T), generator=g, dtype=torch.float32)
    frames_cpu = [base[:, :, :, 4 * t:4 * t + cfg['W']].contiguous() for t in range(T)]
    frames = [f.to(device) for f in frames_cpu]
    meta = make_meta(cfg['H'], cfg['W'], cfg['H'] * 2, cfg['W'] * 2)
    metas = [meta] * B

    total_streams = B * world
    gathered = {}
    if world > 1:
        def gather(rows):
            gathered['rows'] = parallel.gather_detections(rows, total_streams, world, rank)
        det.gather_fn = gather
    ndet = 0
    for i in range(args.warmup):
        res = det.step(frames[i % T], metas)
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = det.step(frames[i % T], metas)
        ndet += sum(len(r) for r in res)
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    dt = parallel.max_over_ranks(time.perf_counter() - t0)
    fps = total_streams * args.steps / dt

    out = {
        'metric': 'frames/sec (DLA-34 + DCNv2 CenterTrack hot path: forward + decode + track association)',
        'value': round(fps, 2), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1000.0 * dt / args.steps, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s: %s heads, DLA-34, %dx%d, %d stream(s)/GPU, K=%d, flip_test=%s, seeded random-init '
                               'weights' % (args.config, cfg['heads'], cfg['H'], cfg['W'], B, opt.K, cfg['flip']),
                   'global_batch': total_streams, 'parallelism': 'streams sharded dp%d, all-gather of packed detections' % world,
                   'hip_graph': det._ctx['graph'] is not None, 'mean_detections_per_frame': round(ndet / max(1, args.steps * B), 1)},
        'fps_per_gpu': round(fps / world, 2),
    }
    if rank == 0:
        ctx = det._ctx
        # device-only time of one frame (graph replay or eager launches), HIP events on the launch stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            if ctx['graph'] is not None:
                ctx['graph'].replay()
            else:
                ctx['device_frame']()
        e1.record()
        torch.cuda.synchronize()
        dev_ms = e0.elapsed_time(e1) / reps
        out['device_ms_per_frame_batch'] = round(dev_ms, 4)
        if not args.no_roofline:
            st = kernel_pass(det.model, ctx['plan'])
            d = st['dcn']
            tf = d['flops'] / (d['ms'] * 1e-3) / 1e12
            gbs = d['bytes'] / (d['ms'] * 1e-3) / 1e9
            out['roofline'] = {'kernel': 'dcn_mfma_kernel (16 DCNv2 layers of one frame batch, incl. split-K reduce)',
                               'bound': 'mfma', 'achieved': round(tf, 3), 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': round(tf / PEAK_FP32_TFLOPS, 4), 'traffic': None, 'traffic_source': None,
                               'avg_launch_us': round(1000.0 * d['ms'] / d['launches'], 2),
                               'hbm': {'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                       'frac': round(gbs / PEAK_HBM_GBS, 4),
                                       'algorithmic_bytes_per_launch': round(d['bytes'] / d['launches'])}}
            c = st['conv']
            ctf = c['flops'] / (c['ms'] * 1e-3) / 1e12
            out['roofline_conv'] = {'kernel': 'conv_mfma_kernel (%d dense conv launches)' % c['launches'], 'bound': 'mfma',
                                    'achieved': round(ctf, 3), 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s',
                                    'frac': round(ctf / PEAK_FP32_TFLOPS, 4), 'total_ms': round(c['ms'], 4)}
            out['roofline']['total_ms'] = round(d['ms'], 4)
            if B == 1 and args.config == 'mot17_512' and not args.height and not args.width:
                tb, src = pmc_traffic()            # measured on this workload only
                if tb is not None:
                    out['roofline']['traffic'] = round(tb)
                    out['roofline']['traffic_source'] = src
        if args.pcie and world == 1:
            # the boundary of the reference hands over HOST frames (PrefetchDataset, test.py:22-51): same loop with the
            # H2D copy inside the step, and with raw u8 1080p frames warped / normalised on the device
            det2 = StreamDetector(opt, model=model, num_streams=B, use_graph=not args.no_graph)
            pinned = [f.pin_memory() for f in frames_cpu]
            raw = [np.random.RandomState(t).randint(0, 256, (1080, 1920 + 8, 3)).astype(np.uint8) for t in range(2)]
            rmeta = make_meta(cfg['H'], cfg['W'], 1080, 1920)
            modes = {'host_fp32_pinned_frames': lambda i: det2.step(pinned[i % T], metas),
                     'raw_u8_1080p_frames_device_preprocess':
                         lambda i: det2.step([raw[i & 1][:, 4 * (i % 3):4 * (i % 3) + 1920]] * B, [rmeta] * B)}
            out['pcie_inclusive'] = {}
            for name, fn in modes.items():
                det2.reset_tracking()
                for i in range(args.warmup):
                    fn(i)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(args.steps):
                    fn(i)
                torch.cuda.synchronize()
                out['pcie_inclusive'][name] = round(B * args.steps / (time.perf_counter() - t1), 2)
            out['pcie_inclusive']['unit'] = 'frames/s'
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(cfg, heads, sd, [f[0:1] for f in frames_cpu], metas, opt_kw,
                                                   args.cpu_frames)
            except Exception as e:  # the baseline is informational; never lose the GPU line
                out['cpu_baseline'] = {'value': None, 'unit': 'frames/s', 'cores': torch.get_num_threads(),
                                       'kind': 'port', 'sample': 'failed: %r' % (e,)}
        print(json.dumps(out))
        sys.stdout.flush()
    parallel.barrier()
    parallel.shutdown()


if __name__ == '__main__':
    main()
