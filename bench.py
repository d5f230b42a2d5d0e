#!/usr/bin/env python
"""bench.py -- frames/sec of the CenterTrack per-frame inference hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

The metric is SURVEY.md section 8(d)'s: the end-to-end ``Detector.run`` equivalent --
**H2D of a ready fp32 frame (pinned host memory, src/lib/detector.py:93-94)** -> three stems +
DLA-34 + 16 DCNv2 nodes + heads (fp32, sigmoid fused) -> NMS / top-K / gather decode -> one
packed D2H -> host post-process -> track association (and, for N > 1, the RCCL all-gather of
the packed detections).  The H2D is INSIDE the timed region -- every frame is uploaded, the upload
of frame t+1 being enqueued on a second stream while frame t is computed, as the reference's
pinned DataLoader + non_blocking copy does (test.py:74-76) -- and the rate with frames already
resident in HBM is reported next to it (``resident_frames_fps``), never as ``value``.

A "step" is one clip: ``frames_per_step`` consecutive frames of every stream the rank owns
(frames of one stream are sequential -- each needs the previous frame and the tracker state --
so a clip is the natural batch of synthetic input; the default of 64 / streams frames makes the
driver's 20 steps a > 1 s timed region).  ``value`` = streams x frames / time.  Workload at
N=1: BASELINE.json configs[1] = MOT17-half heads, DLA-34, 512x512, batch 1 (one stream),
synthetic frames, seeded random-init weights.  Weak scaling: every rank runs the same number
of streams.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      the DCNv2 kernel (dcn_mfma_kernel): algorithmic flops (2*9*Cin*Cout*h*w per
                layer, + the fused offset/mask conv's) and algorithmic bytes (4*(Cin*h*w
                + 27*h*w + Cout*h*w + 9*Cin*Cout + Cout)) of the 16 DCN layers / their summed launch
                time, measured with HIP events on the launch stream (graph replay of exactly these
                launches) in this process; ``frac_main`` = the same with section 8(d)'s formula
                (main contraction only); traffic = HBM bytes per launch from the committed
                rocprofv3 PMC passes (profiles/), against 8(d)'s bytes and against 8(d) + the IDAUp
                step the finishing launches carry; ``mfma_busy`` / ``waves_per_simd_avg`` of the
                dominant kernel from the committed SQ counter pass (profiles/pmc_mfma_busy.json)
  roofline_conv the dense conv launches (backbone + heads): ``frac`` = EXECUTED matrix-core flops
                (Winograd launches count 16/36 of the direct-convolution flops, the fused heads' 1x1
                layers -- VALU work -- not at all) / time / peak, never above 1; the
                direct-convolution rate is reported beside it as ``algorithmic_tflops``
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's CPU path) timed on the host
                cores: thread sweep, then >= 20 frames after 3 warm-ups at the best thread count
                (reported baseline only)
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
SWEEP_FRAMES = 5             # timed frames per thread count of the cpu_baseline sweep


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--frames-per-step', type=int, default=0,
                    help='consecutive frames of every stream in one step; 0 = max(4, 64 // streams)')
    ap.add_argument('--config', default='mot17_512', help='workload name (scenarios.CONFIGS)')
    ap.add_argument('--streams', type=int, default=0, help='streams (batch) per GPU; 0 = 1 (the headline config)')
    ap.add_argument('--height', type=int, default=0)
    ap.add_argument('--width', type=int, default=0)
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-resident', action='store_true', help='skip the extra resident-frames loop')
    ap.add_argument('--cpu-frames', type=int, default=20)
    ap.add_argument('--cpu-threads', default='8,16,32', help='thread counts of the CPU-baseline sweep')
    ap.add_argument('--hm-gain', type=float, default=11.0)
    ap.add_argument('--raw-u8', action='store_true',
                    help='also report the rate with raw u8 1080p frames handed to step() (u8 H2D + device-side '
                         'pre-processing) under "raw_u8_1080p_fps" -- never the headline value')
    ap.add_argument('--no-extra-configs', action='store_true',
                    help='skip the BASELINE configs 3-5 / 544x960 measurements the default invocation appends as "configs"')
    ap.add_argument('--extra-steps', type=int, default=6, help='timed steps of each appended configuration')
    ap.add_argument('--sparse-heads', action='store_true',
                    help='opt.sparse_heads: regression heads evaluated at the K decode winners only (opt-in product mode, '
                         'never the headline: the reference computes dense maps)')
    ap.add_argument('--no-box-probes', action='store_true', help='box_calibration without the latency / clock probes')
    return ap


def parse():
    return build_parser().parse_args()


def kernel_pass(model, plan, reps=10):
    """Average duration of the DCN / conv launches of the plan INSIDE the frame: the plan's launch list is cut where the kind of
    launch changes (stem | backbone convs | the DCN block | heads), every part is captured as its own HIP graph, and whole frames
    are replayed part by part with a HIP event between the parts -- device time of a part = kernel boundaries included, host launch
    cost excluded, and every part runs on what the part before it left in the caches, as in the real frame (the same thing
    rocprofv3's kernel trace of the frame sees; until round 6 each kind was replayed on its own, back to back, which made the DCN
    figure of the 4-stream configurations depend on the allocator's state by up to 10 % while the frame itself did not change)."""
    import ctypes
    from centertrack_amd import _lib
    lib = _lib.load()
    stats = {}

    def kind_of(l):
        if l.fn == 'dcn_group' or l.name.endswith('.offset'):
            return 'dcn'
        return 'conv' if l.fn in ('conv', 'heads') else 'other'
    parts = []
    for l in plan['launches']:
        k = kind_of(l)
        if not parts or parts[-1][0] != k:
            parts.append((k, []))
        parts[-1][1].append(l)
    model._run_plan(plan)
    torch.cuda.synchronize()
    graphs = []
    for k, ls in parts:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model._run_plan(plan, launches=ls)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with _lib.capture_guard(collect=False), torch.cuda.graph(g):
            model._run_plan(plan, launches=ls)
        graphs.append(g)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(parts) + 1)] for _ in range(reps + 1)]
    for r in range(reps + 1):                  # (frame 0 warms up)
        ev[r][0].record()
        for i, g in enumerate(graphs):
            g.replay()
            ev[r][i + 1].record()
    torch.cuda.synchronize()
    part_ms = {'dcn': 0.0, 'conv': 0.0, 'other': 0.0}
    for r in range(1, reps + 1):
        for i, (k, _) in enumerate(parts):
            part_ms[k] += ev[r][i].elapsed_time(ev[r][i + 1]) / reps
    for kind in ('dcn', 'conv'):
        launches = [l for l in plan['launches'] if kind_of(l) == kind]
        ms = part_ms[kind]
        flops = flops_main = flops_exec = bytes_ = bytes_up = 0.0
        nlayers = 0
        for l in launches:
            if l.fn == 'dcn_group':
                if not (l.args[2] & _lib.CT_DCN_MAIN):
                    continue
                for j in range(l.args[1]):
                    d = l.args[0][j]
                    nlayers += 1
                    hw = d.N * d.H * d.W
                    flops_main += 2.0 * 9 * d.Cin * d.Cout * hw
                    flops += 2.0 * 9 * d.Cin * d.Cout * hw + 2.0 * 9 * d.Cin * 27 * hw      # main + offset/mask conv
                    bytes_ += 4.0 * (d.Cin * hw + 27 * hw + d.Cout * hw + 9 * d.Cin * d.Cout + d.Cout)
                    if d.up_w:          # the IDAUp step the finishing launch of a `proj` layer carries (dla.py:543-545):
                        bytes_up += 4.0 * 2 * d.Cout * hw * d.up_f * d.up_f        # skip tensor read + result written
            elif l.fn == 'heads':        # conv3x3 64 -> 256 + conv1x1 256 -> c of every fused head
                d = l.args
                hw = d.N * d.H * d.W
                cs = sum(d.cout[i] for i in range(d.nheads))
                flops += 2.0 * hw * (9 * 64 * 256 * d.nheads + 256 * cs)
                # executed on the matrix cores: the 3x3 layers in Winograd F(2x2,3x3) form (16 multiplies per 36); the
                # 1x1 output layers are VALU work in the epilogue
                flops_exec += 2.0 * hw * 9 * 64 * 256 * d.nheads * 16.0 / 36.0
                bytes_ += 4.0 * (64 * hw + cs * hw + 9 * 64 * 256 * d.nheads)
            elif kind == 'conv':
                d = l.args
                pad = d.ks // 2
                ho = (d.H + 2 * pad - d.ks) // d.stride + 1
                wo = (d.W + 2 * pad - d.ks) // d.stride + 1
                f = 2.0 * d.ks * d.ks * d.Cin * d.Cout * d.N * ho * wo
                if d.proj_w_packed:             # Tree.project fused into the stride-2 launch: conv1x1 Cin -> Cout
                    f += 2.0 * d.Cin * d.Cout * d.N * ho * wo
                    bytes_ += 4.0 * (d.Cout * d.N * ho * wo + d.Cin * d.Cout)
                flops += f
                flops_exec += f * (16.0 / 36.0 if 201 <= int(d.algo) <= 211 else 1.0)      # Winograd launches
                bytes_ += 4.0 * (d.Cin * d.N * d.H * d.W + d.Cout * d.N * ho * wo + d.ks * d.ks * d.Cin * d.Cout)
        if kind == 'dcn':
            stats[kind] = dict(launches=len(launches), layers=nlayers, flops=flops, flops_main=flops_main, bytes=bytes_,
                               bytes_idaup=bytes_up, ms=ms)
            continue
        stats[kind] = dict(launches=len(launches), flops=flops, flops_main=flops_main, flops_exec=flops_exec, bytes=bytes_, ms=ms)
    return stats


def pmc_traffic():
    """HBM bytes per DCN layer (all dcn_* kernels of a frame / 16) from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, made by
    tools/collect_profiles.sh from FETCH_SIZE / WRITE_SIZE with the gfx950 corrections); None if absent."""
    p = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(p) as f:
            j = json.load(f)
        return j['dcn_mfma_kernel']['hbm_bytes_per_layer'], j.get('source', p)
    except Exception:
        return None, None


def pmc_busy():
    """MFMA-busy share and resident waves per SIMD per kernel class from the committed SQ counter pass
    (profiles/pmc_mfma_busy.json, tools/pmc_busy.py); None if absent."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_mfma_busy.json')) as f:
            return json.load(f)
    except Exception:
        return None


def profiled_dcn():
    """DCN device time per frame as the committed rocprofv3 kernel trace of this round saw it (profiles/dcn_profiled.json,
    written by tools/dcn_profiled.py from the `--kernel-trace --stats` table of the same bench command): lets a reader
    reproduce ``frac`` from profiles/ alone, next to the live HIP-event figure of this process."""
    p = os.path.join(ROOT, 'profiles', 'dcn_profiled.json')
    try:
        with open(p) as f:
            return json.load(f)
    except Exception:
        return None


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(cfg, heads, sd, frames_cpu, metas, opt_kw, nframes, sweep):
    """CPU oracle (port of the reference CPU path) on the same workload, bounded sample: a thread sweep (1 warm-up + 5
    timed frames per thread count), then ``nframes`` frames after 3 warm-ups at the best count."""
    from oracle import detector as odet
    oopt = odet.default_opt(input_h=cfg['H'], input_w=cfg['W'], num_classes=heads['hm'], **opt_kw)
    det = odet.Detector(oopt, sd, heads)
    if oopt.flip_test:
        frames_cpu = [torch.cat((f, torch.flip(f, [3])), 0) for f in frames_cpu]
    ncpu = os.cpu_count() or 1
    counts = sorted({min(max(1, t), ncpu) for t in sweep})
    saved = torch.get_num_threads()
    per_thread = {}

    def run(n, first):
        t0 = time.time()
        for i in range(n):
            det.run(frames_cpu[(first + i) % len(frames_cpu)], dict(metas[0]))
        return time.time() - t0

    try:
        for t in counts:
            torch.set_num_threads(t)
            det.reset_tracking()
            run(1, 0)
            per_thread[t] = round(SWEEP_FRAMES / run(SWEEP_FRAMES, 1), 3)
        best = max(per_thread, key=per_thread.get)
        torch.set_num_threads(best)
        det.reset_tracking()
        run(3, 0)
        fps = nframes / run(nframes, 3)
    finally:
        torch.set_num_threads(saved)
    return dict(value=round(fps, 4), unit='frames/s', cores=best, kind='port', cpu=cpu_model(), host_cpus=ncpu,
                thread_sweep_fps=per_thread,
                note='the sweep times %d frames per thread count and only picks `cores`: its figures scatter by +-25 %% between '
                     'runs on this host (turbo, page cache); `value` is the %d-frame measurement' % (SWEEP_FRAMES, nframes),
                sample='%d frames of the N=1 workload (1 stream, %dx%d) through oracle/detector.py after 3 warm-up ' % (nframes, cfg['H'], cfg['W']) +
                       'frames at the best thread count of the sweep (1 warm-up + %d timed frames per count); ' % SWEEP_FRAMES +
                       'pure-PyTorch CPU restatement of the reference path incl. DCNv2')


# BASELINE.json configs[2..4] at their per-GPU batch + the reference's own MOT input size (datasets/mot.py:15): measured by
# the DEFAULT invocation after the headline and appended to its line as "configs"  -- never `value`
# BASELINE configs 3-5 at their per-GPU batch, the reference's own MOT size, and the 32-stream end points of north_star's
# synthetic sweep (driver-witnessed instead of builder-only); (name, streams, timed steps or 0 = --extra-steps)
EXTRA_CONFIGS = (('kitti_1280x384', 4, 0), ('coco_512', 4, 0), ('nusc_800x448', 4, 0), ('mot17_544x960', 1, 0),
                 ('mot17_512', 32, 3), ('nusc_800x448', 32, 3))


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
    env = (rank, world, device)
    if args.sparse_heads:
        #  the opt-in mode is never the headline: `value` is the dense figure of the same workload, the sparse
        # figure goes under "sparse_heads" of the same line
        dense = argparse.Namespace(**vars(args))
        dense.sparse_heads = False
        out = measure(dense, env, headline=True)
        sp = argparse.Namespace(**vars(args))
        sp.no_resident = sp.no_cpu_baseline = sp.no_roofline = True
        try:
            o = measure(sp, env, headline=False)
            out['sparse_heads'] = {'fps': o['value'], 'device_ms_per_frame_batch': o.get('device_ms_per_frame_batch'),
                                   'launches_per_frame': o.get('launches_per_frame'), 'workload': o['config']['workload'],
                                   'note': 'opt.sparse_heads (opt-in, not the reference\'s computation graph); never `value`'}
        except Exception as e:
            out['sparse_heads'] = {'error': repr(e)}
    else:
        out = measure(args, env, headline=True)
    default_workload = (args.config == 'mot17_512' and args.streams <= 1 and not args.height and not args.width
                        and not args.no_graph and not args.sparse_heads)

    def sparse_line(name, streams):
        """the same workload with opt.sparse_heads (opt-in product mode; reported beside the dense figure, never as `value`)"""
        a = argparse.Namespace(**vars(args))
        a.config, a.streams, a.frames_per_step, a.sparse_heads = name, streams, 0, True
        a.steps, a.warmup = args.extra_steps, 2
        a.no_resident = a.no_cpu_baseline = a.no_roofline = True
        a.raw_u8 = False
        try:
            o = measure(a, env, headline=False)
            return {'fps': o['value'], 'device_ms_per_frame_batch': o['device_ms_per_frame_batch'],
                    'launches_per_frame': o['launches_per_frame'], 'active': bool(o['config'].get('sparse_heads')),
                    'mean_detections_per_frame': o['config']['mean_detections_per_frame']}
        except Exception as e:
            return {'error': repr(e)}

    if rank == 0 and world == 1 and default_workload and not args.no_extra_configs:
        out['sparse_heads'] = dict(sparse_line('mot17_512', 1), note=(
            'opt.sparse_heads (opt-in, not the reference\'s computation graph: its regression heads are dense maps that '
            'generic_decode reads at K pixels): the same workload with those heads evaluated at the K winners only; rows '
            'equal the dense path\'s (tests/test_hip_sparse_heads.py); never `value`'))
        out['configs'] = []
        for name, streams, nsteps in EXTRA_CONFIGS:
            a = argparse.Namespace(**vars(args))
            a.config, a.streams, a.frames_per_step = name, streams, 0
            a.steps, a.warmup = (nsteps or args.extra_steps), 2
            a.no_resident = a.no_cpu_baseline = True
            a.raw_u8 = False
            t0 = time.perf_counter()
            try:
                o = measure(a, env, headline=False)
                out['configs'].append({
                    'workload': o['config']['workload'].split(';')[0], 'streams_per_gpu': streams,
                    'fps': o['value'], 'steps': a.steps, 'frames_per_step': o['config']['frames_per_step'],
                    'timed_region_s': o['timed_region_s'], 'ms_per_frame_batch': o['ms_per_frame_batch'],
                    'device_ms_per_frame_batch': o['device_ms_per_frame_batch'], 'launches_per_frame': o['launches_per_frame'],
                    'mean_detections_per_frame': o['config']['mean_detections_per_frame'],
                    'plan_hash': o['plan_hash'], 'wall_s': round(time.perf_counter() - t0, 1)})
                if o.get('roofline') is not None:          # (absent with --no-roofline)
                    out['configs'][-1]['roofline'] = {
                        'kernel': 'dcn_mfma_kernel (all DCNv2 launches of a frame batch)', 'bound': 'mfma',
                        'achieved': o['roofline']['achieved'], 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': o['roofline']['frac'], 'total_ms': o['roofline']['total_ms']}
                if o.get('roofline_conv') is not None:
                    out['configs'][-1]['roofline_conv'] = {
                        'achieved': o['roofline_conv']['achieved'], 'frac': o['roofline_conv']['frac'],
                        'total_ms': o['roofline_conv']['total_ms'],
                        'algorithmic_tflops': o['roofline_conv']['algorithmic_tflops']}
                if streams <= 4:                           # (the opt-in mode beside the BASELINE batches only)
                    sp = sparse_line(name, streams)
                    if sp.get('active') or 'error' in sp:
                        out['configs'][-1]['sparse_heads'] = sp
            except Exception as e:       # the extra configurations never cost the headline line
                out['configs'].append({'workload': name, 'streams_per_gpu': streams, 'error': repr(e)})
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    parallel.barrier()
    parallel.shutdown()


def measure(args, env, headline=True):
    """one workload (args.config x args.streams): build the detector, warm up, time, and (rank 0) describe it"""
    from centertrack_amd import parallel
    rank, world, device = env

    import scenarios as S
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
    fps_step = args.frames_per_step if args.frames_per_step > 0 else max(4, 64 // B)
    heads = S.HEAD_SETS[cfg['heads']]
    sd = W.make_synthetic_state_dict(heads, seed=317, hm_gain=args.hm_gain)
    if 'ltrb_amodal' in heads:
        sd['ltrb_amodal.2.bias'] = torch.tensor([-3.0, -3.0, 3.0, 3.0])
    opt_kw = dict(track_thresh=cfg['track_thresh'], pre_thresh=cfg['pre_thresh'], flip_test=cfg['flip'])
    opt = default_opt(heads, sparse_heads=bool(getattr(args, 'sparse_heads', False)), **opt_kw)
    model = DLASegHIP(heads)
    model.load_state_dict(sd)
    det = StreamDetector(opt, model=model, num_streams=B, use_graph=not args.no_graph)

    # synthetic stream: a fixed N(0,1) image scrolled by 4 px / frame, T distinct frames in PINNED host memory (what
    # the reference's PrefetchDataset / DataLoader(pin_memory=True) hands to Detector.run, test.py:74-76)
    T = 8
    g = torch.Generator().manual_seed(317 + 7 + rank)
    base = torch.randn((B, 3, cfg['H'], cfg['W'] + 4 * T), generator=g, dtype=torch.float32)
    frames_cpu = [base[:, :, :, 4 * t:4 * t + cfg['W']].contiguous() for t in range(T)]
    pinned = [f.pin_memory() for f in frames_cpu]
    meta = make_meta(cfg['H'], cfg['W'], cfg['H'] * 2, cfg['W'] * 2)
    metas = [meta] * B

    total_streams = B * world
    gatherer = None
    last = {}
    K, F = parallel.build_plans_consistently(lambda: det_rows_shape(det, cfg))
    # every rank must replay the same fp32 summation orders: compare the plans before anything is timed
    plan_hash = parallel.check_same_plan(DLASegHIP.plan_signature(det._ctx['plan']))
    if world > 1 or parallel.group_active():        # (a torchrun job of one rank still exchanges through RCCL)
        gatherer = parallel.DetectionGatherer(total_streams, world, rank, K, F, device, overlap=True)
        score_col = [st for name, st, _ in det._ctx['decoder'].layout if name == 'scores'][0]

        def gather(rows):
            # the block of the PREVIOUS frame is complete by now (its collective ran beside the host work of that
            # frame): consume it -- count the detections of ALL streams -- then enqueue this frame's gather on the side
            # stream; the returned event keeps the next graph launch from overwriting the rows too early
            gatherer.consume(score_col, opt.out_thresh)
            last['rows'] = rows
            gatherer(rows)
            return gatherer.rows_free
        det.gather_fn = gather

    def timed(frame_of, steps, first=0):
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()
        if gatherer is not None:
            gatherer.consume(score_col, opt.out_thresh)      # (flush the block of the last untimed frame)
            gatherer.consumed_steps = gatherer.consumed_detections = 0
        t0 = time.perf_counter()
        nfr, ndet = parallel.run_steps(det, frame_of, metas, steps, fps_step, first)
        if gatherer is not None:
            gatherer.consume(score_col, opt.out_thresh)      # the last frame's block (host-synchronous)
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()
        return parallel.max_over_ranks(time.perf_counter() - t0), nfr, ndet

    host_frame = lambda t: pinned[t % T]
    parallel.run_steps(det, host_frame, metas, args.warmup, fps_step)
    dt, nfr, ndet = timed(host_frame, args.steps, args.warmup * fps_step)
    fps = total_streams * nfr / dt
    rccl_ranks = 1
    gathered = None
    if gatherer is not None:                         # outside the timed region: prove what the collective moved
        rccl_ranks = gatherer.verify(last['rows'])
        # the blocks consumed inside the timed loop carried every rank's detections: the count rank 0 made from them
        # equals the sum of what the ranks' own trackers were handed (decode rows above out_thresh)
        gathered = {'steps_consumed': gatherer.consumed_steps, 'detections_counted_from_blocks': gatherer.consumed_detections,
                    'sum_of_rank_local_results': parallel.sum_over_ranks(ndet)}
        gathered['equal'] = gathered['detections_counted_from_blocks'] == gathered['sum_of_rank_local_results']

    out = {
        'metric': 'frames/sec (DLA-34 + DCNv2 CenterTrack hot path: H2D of the frame + forward + decode + D2H + '
                  'track association)',
        'value': round(fps, 2), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1000.0 * dt / args.steps, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s: %s heads, DLA-34, %dx%d, %d stream(s)/GPU, K=%d, flip_test=%s, seeded random-init '
                               'weights; one step = %d consecutive frames of every stream, each frame uploaded from '
                               'pinned host memory inside the step (frame t+1 on a copy stream while frame t runs)' % (args.config, cfg['heads'], cfg['H'], cfg['W'], B,
                                                                      opt.K, cfg['flip'], fps_step),
                   'global_batch': total_streams, 'frames_per_step': fps_step,
                   'parallelism': 'streams sharded dp%d, all-gather of packed detections' % world,
                   'hip_graph': det._ctx['graph'] is not None,
                   'mean_detections_per_frame': round(ndet / max(1, nfr * B), 1)},
        'fps_per_gpu': round(fps / world, 2), 'ms_per_frame_batch': round(1000.0 * dt / max(1, nfr), 4),
        'timed_region_s': round(dt, 3), 'h2d_in_timed_region': True, 'h2d_overlaps_previous_frame': True, 'rccl_ranks': rccl_ranks,
        'process_group': (torch.distributed.get_backend() if parallel.group_active() else None),
        'plan_hash': plan_hash,
    }
    if det.sparse:
        out['config']['sparse_heads'] = True
        out['config']['workload'] += '; SPARSE HEADS (regression heads evaluated at the K decode winners only -- opt-in mode)'
    if gathered is not None:
        out['gathered'] = gathered
    clocks = None
    if not args.no_resident:
        frames = [f.to(device) for f in frames_cpu]
        dt2, nfr2, _ = timed(lambda t: frames[t % T], args.steps)
        out['resident_frames_fps'] = round(total_streams * nfr2 / dt2, 2)
        if headline:
            # clock levels under load: sampled in a SEPARATE, untimed pass of the same loop (the sampler thread
            # polls sysfs from this process and would perturb a timed loop); every rank runs the pass (it holds barriers),
            # rank 0 samples; a failure of the sampler never costs the line
            sampler = None
            if rank == 0:
                try:
                    from tools import box_calib
                    sampler = box_calib.ClockSampler()
                    sampler.start()
                except Exception as e:
                    sampler, clocks = None, {'error': repr(e)}
            timed(lambda t: frames[t % T], max(2, args.steps // 4))
            if sampler is not None:
                try:
                    clocks = sampler.summary()
                except Exception as e:
                    clocks = {'error': repr(e)}
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
        graph_ms = e0.elapsed_time(e1) / reps
        pre_ms = 0.0
        if ctx.get('partial') is not None:
            # split stem: the tracker-independent part of a frame (x / pre_img stem terms, mirrored half) is launched
            # behind the PREVIOUS frame's graph and runs while the host associates that frame -- device work of the
            # frame all the same, timed here on its own
            e0.record()
            for _ in range(reps):
                ctx['pre_stage'](0)
            e1.record()
            torch.cuda.synchronize()
            pre_ms = e0.elapsed_time(e1) / reps
        dev_ms = graph_ms + pre_ms
        out['device_ms_per_frame_batch'] = round(dev_ms, 4)
        out['device_ms_frame_graph'] = round(graph_ms, 4)
        out['device_ms_prestage'] = round(pre_ms, 4)
        # wall - graph: what the host adds between two frame graphs (association, launch); the pre-stage runs inside it
        out['host_gap_ms_per_frame_batch'] = round(1000.0 * dt / max(1, nfr) - graph_ms, 4)
        if headline:
            try:
                from tools import box_calib
                out['box_calibration'] = box_calib.box_calibration(device, probes=not args.no_box_probes)
                if clocks is not None:
                    out['box_calibration']['clocks_under_load'] = clocks
            except Exception as e:
                out['box_calibration'] = {'error': repr(e)}
        out['launches_per_frame'] = len(ctx['plan']['launches'])
        if not args.no_roofline:
            st = kernel_pass(det.model, ctx['plan'])
            d = st['dcn']
            tf = d['flops'] / (d['ms'] * 1e-3) / 1e12
            tf_main = d['flops_main'] / (d['ms'] * 1e-3) / 1e12
            gbs = d['bytes'] / (d['ms'] * 1e-3) / 1e9
            nl = max(1, d['layers'])
            out['roofline'] = {'kernel': 'dcn_mfma_kernel: the %d DCNv2 layers of one frame batch (grouped gather + contraction '
                                         'launches, their offset/mask convs, the finishing reduce / BN / ReLU / IDAUp launches: '
                                         '%d launches); unit of the per-launch figures = one layer' % (nl, d['launches']),
                               'bound': 'mfma', 'achieved': round(tf, 3), 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': round(tf / PEAK_FP32_TFLOPS, 4),
                               'frac_main': round(tf_main / PEAK_FP32_TFLOPS, 4),      # main contraction only
                               'traffic': None, 'traffic_source': None,
                               'method': 'HIP events between the parts of the frame -- stem | backbone | DCN block | heads, each a graph, whole frames '
                                         'replayed in order: in-frame device time of the DCN block, its kernel boundaries included',
                               'avg_launch_us': round(1000.0 * d['ms'] / nl, 2),
                               'hbm': {'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                       'frac': round(gbs / PEAK_HBM_GBS, 4),
                                       # 4 * (Cin*h*w + 27*h*w + Cout*h*w + 9*Cin*Cout + Cout) per layer
                                       'algorithmic_bytes_per_launch': round(d['bytes'] / nl),
                                       # the finishing launches of the 8 `proj` layers also carry the IDAUp step
                                       # (skip tensor read + result written, dla.py:543-545)
                                       'algorithmic_bytes_per_launch_with_idaup': round((d['bytes'] + d['bytes_idaup']) / nl)}}
            prof = profiled_dcn()
            if prof is not None and B == 1 and args.config == 'mot17_512' and not args.height and not args.width:
                out['roofline']['profiled'] = prof
            c = st['conv']
            ctf = c['flops_exec'] / (c['ms'] * 1e-3) / 1e12        # flops the matrix cores EXECUTE (Winograd: 16/36)
            out['roofline_conv'] = {'kernel': 'conv_mfma / conv_ksplit / wino_conv kernels (%d dense conv launches incl. the fused heads); achieved = '
                                              'executed matrix-core flops (Winograd launches: 16/36 of the direct-convolution flops; the fused '
                                              'heads\' 1x1 layers run on the VALU and are not counted)' % c['launches'], 'bound': 'mfma',
                                    'achieved': round(ctf, 3), 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s',
                                    'frac': round(ctf / PEAK_FP32_TFLOPS, 4), 'total_ms': round(c['ms'], 4),
                                    'algorithmic_tflops': round(c['flops'] / (c['ms'] * 1e-3) / 1e12, 3)}
            out['roofline']['total_ms'] = round(d['ms'], 4)
            if B == 1 and args.config == 'mot17_512' and not args.height and not args.width:
                tb, src = pmc_traffic()            # measured on this workload only
                if tb is not None:
                    out['roofline']['traffic'] = round(tb)
                    out['roofline']['traffic_source'] = src
                    out['roofline']['traffic_over_algorithmic'] = round(
                        tb / out['roofline']['hbm']['algorithmic_bytes_per_launch'], 2)
                    out['roofline']['traffic_over_algorithmic_with_idaup'] = round(
                        tb / out['roofline']['hbm']['algorithmic_bytes_per_launch_with_idaup'], 2)
                busy = pmc_busy()                   # measured on this workload only
                if busy is not None:
                    if busy.get('dcn_main'):
                        out['roofline']['mfma_busy'] = busy['dcn_main']['mfma_busy']
                        out['roofline']['waves_per_simd_avg'] = busy['dcn_main']['waves_per_simd_avg']
                        out['roofline']['pmc_kernel'] = busy['dcn_main']['kernel']
                        out['roofline']['pmc_source'] = busy.get('source')
                    if busy.get('conv'):
                        out['roofline_conv']['mfma_busy'] = busy['conv']['mfma_busy']
                        out['roofline_conv']['waves_per_simd_avg'] = busy['conv']['waves_per_simd_avg']
                        if busy.get('backbone_3x3'):
                            out['roofline_conv']['mfma_busy_backbone_3x3'] = busy['backbone_3x3']['mfma_busy']
        if args.raw_u8 and world == 1:
            # raw u8 1080p frames handed to step(): u8 H2D + warp / normalise on the device
            det2 = StreamDetector(opt, model=model, num_streams=B, use_graph=not args.no_graph)
            raw = [np.random.RandomState(t).randint(0, 256, (1080, 1920 + 8, 3)).astype(np.uint8) for t in range(2)]
            rmeta = make_meta(cfg['H'], cfg['W'], 1080, 1920)
            fn = lambda i: det2.step([raw[i & 1][:, 4 * (i % 3):4 * (i % 3) + 1920]] * B, [rmeta] * B)
            n = args.steps * fps_step
            for i in range(args.warmup * fps_step):
                fn(i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(n):
                fn(i)
            torch.cuda.synchronize()
            out['raw_u8_1080p_fps'] = round(B * n / (time.perf_counter() - t1), 2)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(cfg, heads, sd, [f[0:1] for f in frames_cpu], metas, opt_kw,
                                                   args.cpu_frames, [int(t) for t in args.cpu_threads.split(',')])
            except Exception as e:  # the baseline is informational; never lose the GPU line
                out['cpu_baseline'] = {'value': None, 'unit': 'frames/s', 'cores': torch.get_num_threads(),
                                       'kind': 'port', 'sample': 'failed: %r' % (e,)}
    return out


def det_rows_shape(det, cfg):
    """(K, F) of the packed decode rows of this detector (known once its context exists)"""
    ctx = det._context(cfg['H'], cfg['W'])
    return int(ctx['decoder'].out.shape[1]), int(ctx['decoder'].out.shape[2])


if __name__ == '__main__':
    main()
