#!/usr/bin/env python
"""What box did a measurement run on?  ``python tools/box_calib.py`` prints one JSON line; ``bench.py`` embeds the same
dict as ``box_calibration``.  The MI355X boxes of the pool run the one-stream frame 16 % apart while their fp32 MFMA
loops agree to 0.3 %; rounds 3/4 recorded only the MFMA rate and the device-to-device copy bandwidth, which did not
predict the class.  Round 5 adds probes of what a latency-bound launch actually waits for.  Over 47
calls every DATA-path probe read the same in both states; the one that differs is instruction fetch
(``launch_us.ifetch_64KB_code_256wg``: 39 us fast, 56 us slow -- DESIGN.md section 4):

  chase_ns       dependent-load latency of ONE lane through a random ring of 128-byte lines (ctp_chase): 16 KB
                 (the CU's L1), 1 MB (one XCD's L2), 64 MB (Infinity Cache), 2 GiB (HBM) and 1 MB of PINNED HOST memory
                 (the PCIe round trip the decode's row stores and the end-of-frame flag see)
  shader_mhz     s_memtime clocks / s_memrealtime ticks of those one-lane kernels: the shader clock a nearly idle chip runs
  stream_GBps    copy of 1 GiB at one wave per SIMD with one 16-byte load in flight per lane (the regime of the stem and
                 the decode) and at 8 workgroups per CU with four in flight (the bandwidth regime)
  launch_us      dependent kernel boundary inside a captured graph: 200 launches of 1 / 256 workgroups; ``ifetch_64KB_code_*``:
                 64 KB of straight-line code run once per wave by 256 / 1024 workgroups (cold instruction cache)
  cu_map         hardware ids (XCD / shader engine / CU) and start / end times of every workgroup of three probe launches
                 (ctp_cu_map): how many CUs each shader engine of each XCD holds on THIS chip (32 of 36 per XCD are
                 enabled; which ones differs), and whether a "two workgroups per CU" launch finishes in one round
  sysfs          clock levels (sclk / mclk / fclk / socclk: active level and the table), partition modes and power cap as
                 the amdgpu driver reports them for the device
  clocks_under_load (bench.py only) the active sclk / mclk / fclk levels sampled every 20 ms while the resident-frame
                 loop runs
"""
import ctypes
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _timed(fn, reps=1):
    import torch
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _ring(n_lines, device, pinned=False):
    """uint32 index of the next line at the start of every 128-byte line: one random cycle over all lines"""
    import torch
    g = torch.Generator(device='cpu').manual_seed(5)
    perm = torch.randperm(n_lines, generator=g, dtype=torch.int64)
    nxt = torch.empty(n_lines, dtype=torch.int32)
    nxt[perm] = torch.roll(perm, -1).to(torch.int32)
    if pinned:
        ring = torch.zeros(n_lines * 32, dtype=torch.int32).pin_memory()
        ring[::32] = nxt
        return ring
    ring = torch.zeros(n_lines * 32, dtype=torch.int32, device=device)
    ring[::32] = nxt.to(device)
    return ring


def chase(lib, st, device):
    import torch
    out = torch.zeros(3, dtype=torch.int64, device=device)
    res, mhz = {}, {}
    for name, lines, hops, pinned in (('l1_16KB', 128, 20000, False), ('l2_1MB', 8192, 20000, False),
                                      ('mall_64MB', 1 << 19, 20000, False), ('mall_128MB', 1 << 20, 20000, False),
                                      ('mall_192MB', 3 << 19, 20000, False), ('mall_240MB', 15 << 17, 20000, False),
                                      ('hbm_2GiB', 1 << 24, 20000, False),
                                      ('host_pinned_1MB', 8192, 3000, True)):
        ring = _ring(lines, device, pinned)
        if not pinned:
            ring.sum().item()                 # one streaming pass: the footprint is in whatever level can hold it
        best = None
        start = 0
        for rep in range(3):
            # every repetition CONTINUES the chain where the last one stopped: re-walking the same 20 000 lines (2.5 MB)
            # would find them in the L2 whatever the footprint (the first version of this probe did: 90 ns "HBM")
            lib.ctp_chase(ctypes.c_void_p(ring.data_ptr()), hops, start, ctypes.c_void_p(out.data_ptr()), st)
            torch.cuda.synchronize()
            start, ticks, clocks = (int(v) for v in out.tolist())
            ns = ticks * 10.0 / hops
            if best is None or ns < best[0]:
                best = (ns, clocks / max(1, ticks) * 100.0)
        res[name] = round(best[0], 1)
        mhz[name] = round(best[1])
        del ring
    # producer -> consumer: the ring was just WRITTEN by another kernel (a device copy), no read pass in between -- where
    # does a kernel find what its predecessor wrote (the situation of every activation tensor of a frame)?
    for name, lines in (('after_write_8MB', 1 << 16), ('after_write_64MB', 1 << 19)):
        src = _ring(lines, device)
        ring = torch.empty_like(src)
        best = None
        for rep in range(3):
            ring.copy_(src)                                       # the producer kernel
            lib.ctp_chase(ctypes.c_void_p(ring.data_ptr()), 20000, (rep * 7919) % lines, ctypes.c_void_p(out.data_ptr()), st)
            torch.cuda.synchronize()
            _, ticks, clocks = (int(v) for v in out.tolist())
            ns = ticks * 10.0 / 20000
            best = ns if best is None or ns < best else best
        res[name] = round(best, 1)
        del ring, src
    # ... and under LOAD: every lane of 256 / 1024 workgroups follows the ring from its own line -- Mlines/s of dependent random
    # 128-byte-line accesses (the single lane above sees an idle fabric; a slow lease serves the SAME traffic more slowly)
    for name, lines in (('mall_64MB', 1 << 19), ('hbm_2GiB', 1 << 24)):
        ring = _ring(lines, device)
        ring.sum().item()
        sink = torch.zeros(1024 * 256, dtype=torch.int32, device=device)
        for blocks in (256, 1024):
            hops = 2000
            ms = _timed(lambda: lib.ctp_chase_many(ctypes.c_void_p(ring.data_ptr()), hops, lines, blocks, ctypes.c_void_p(sink.data_ptr()), st), 2)
            res['loaded_%s_%dwg_Mlines_per_s' % (name, blocks)] = round(blocks * 256 * hops / ms / 1e3)
        del ring
    return res, mhz


def stream(lib, st, device):
    import torch
    n = 1 << 28
    a = torch.empty(n, device=device)
    b = torch.empty(n, device=device)
    res = {}
    for name, blocks, inflight in (('1GiB_1wave_per_simd_1load', 256, 1), ('1GiB_8wg_per_cu_4loads', 2048, 4)):
        ms = _timed(lambda: lib.ctp_stream(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), n * 4, blocks, inflight, st), 3)
        res[name] = round(2.0 * n * 4 / ms / 1e6)
    # the write side alone: the store round trip of one lane (a launch retires when its last store is acknowledged) and
    # store-only streaming at one wave per SIMD / eight workgroups per CU
    out = torch.zeros(2, dtype=torch.int64, device=device)
    hops = 20000
    best = None
    for rep in range(3):
        lib.ctp_write(ctypes.c_void_p(b.data_ptr()), n * 4, hops, 1, 0, ctypes.c_void_p(out.data_ptr()), st)
        torch.cuda.synchronize()
        ns = int(out[0].item()) * 10.0 / hops
        best = ns if best is None or ns < best else best
    res['store_ack_ns_one_lane'] = round(best, 1)
    for name, blocks in (('1GiB_fill_1wave_per_simd', 256), ('1GiB_fill_8wg_per_cu', 2048)):
        ms = _timed(lambda: lib.ctp_write(ctypes.c_void_p(b.data_ptr()), n * 4, 0, blocks, 1, None, st), 3)
        res[name] = round(n * 4 / ms / 1e6)
    ms = _timed(lambda: lib.ctp_write(ctypes.c_void_p(b.data_ptr()), (1 << 22) * 4, 0, 256, 1, None, st), 20)
    res['16MiB_fill_1wave_per_simd'] = round((1 << 22) * 4 / ms / 1e6)
    m = 1 << 22           # 16 MiB: L2 / Infinity-Cache resident
    ms = _timed(lambda: lib.ctp_stream(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), m * 4, 256, 1, st), 20)
    res['16MiB_1wave_per_simd_1load'] = round(2.0 * m * 4 / ms / 1e6)
    del a, b
    return res


def launches(lib, device):
    import torch
    from centertrack_amd import _lib
    buf = torch.zeros(256 * 256, device=device)
    res = {}
    # 64 KB of straight-line code run once per wave (cold instruction cache at every launch): us per launch, 256 / 1024 workgroups
    for blocks in (256, 1024):
        st0 = _lib.stream_ptr()
        ms = _timed(lambda: lib.ctp_ifetch(blocks, ctypes.c_void_p(buf.data_ptr()), st0), 10)
        res['ifetch_64KB_code_%dwg' % blocks] = round(ms * 1e3, 2)
    N = 200
    for blocks in (1, 256):
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            lib.ctp_launches(4, blocks, ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(side.cuda_stream))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            lib.ctp_launches(N, blocks, ctypes.c_void_p(buf.data_ptr()), _lib.stream_ptr())
        ms = _timed(g.replay, 5)
        res['graph_%dwg' % blocks] = round(ms * 1e3 / N, 2)
    return res


def cu_map(lib, st, device):
    """which CUs exist (per XCD and shader engine) and how a "two workgroups per CU" launch lands on them: 512 workgroups
    that can only sit two to a CU (64 KB of LDS each) and run 20 us each -- one round (20 us) if the dispatcher can give
    every CU exactly two, more if some shader engines hold fewer CUs than others"""
    import collections
    import torch
    res = {}
    for name, blocks, lds, ticks in (('512wg_2_per_cu_20us', 512, 64 * 1024, 2000), ('256wg_1_per_cu_20us', 256, 128 * 1024, 2000),
                                     ('1024wg_map', 1024, 16 * 1024, 200)):
        out = torch.zeros(4 * blocks, dtype=torch.int32, device=device)
        lib.ctp_cu_map(blocks, lds, ticks, ctypes.c_void_p(out.data_ptr()), st)      # (first launch: code upload)
        torch.cuda.synchronize()
        out.zero_()
        lib.ctp_cu_map(blocks, lds, ticks, ctypes.c_void_p(out.data_ptr()), st)
        torch.cuda.synchronize()
        v = out.cpu().numpy().astype('uint32').reshape(blocks, 4)
        hw, xcc = v[:, 0], v[:, 1] & 0xf
        cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
        key = [(int(x), int(e), int(h), int(c)) for x, e, h, c in zip(xcc, se, sh, cu)]
        per_cu = collections.Counter(key)
        per_se = collections.Counter((k[0], k[1], k[2]) for k in per_cu)              # CUs seen per (XCD, SE, SH)
        t0 = v[:, 2].astype('int64'); t1 = v[:, 3].astype('int64')
        span = int(((t1 - t0.min()) & 0xffffffff).max())
        d = {'span_us': round(span / 100.0, 1), 'distinct_cus': len(per_cu),
             'wgs_per_cu': dict(sorted(collections.Counter(per_cu.values()).items())),
             'late_starts': int((((t0 - t0.min()) & 0xffffffff) > ticks // 2).sum())}
        if name == '1024wg_map':
            d['cus_per_xcd'] = [sum(n for (x, e, h), n in per_se.items() if x == xi) for xi in range(8)]
            d['cus_per_engine'] = dict(sorted(collections.Counter(per_se.values()).items()))      # {CUs in an (XCD, SE, SH): how many such}
        res[name] = d
    return res


def xcd_stream(lib, st, device):
    """per-XCD memory rate: 256 / 2048 workgroups copy 4 MB / 512 KB of their own each (HBM-sized total: 1 GiB) and 256
    workgroups 64 KB each out of a 16 MiB L2 / Infinity-Cache resident buffer; reported: GB/s of the mean workgroup of every
    XCD and the slowest-to-fastest XCD ratio of the launch's critical path (the LAST workgroup end of each XCD)"""
    import torch
    res = {}
    a = torch.empty(1 << 28, device=device)
    b = torch.empty(1 << 28, device=device)
    for name, blocks, chunk, reps in (('1GiB_256wg', 256, 4 << 20, 2), ('1GiB_2048wg', 2048, 512 << 10, 2), ('16MiB_256wg', 256, 64 << 10, 6)):
        out = torch.zeros(4 * blocks, dtype=torch.int32, device=device)
        for _ in range(reps):
            lib.ctp_xcd_stream(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), chunk, blocks, ctypes.c_void_p(out.data_ptr()), st)
        torch.cuda.synchronize()
        v = out.cpu().numpy().astype('uint32').reshape(blocks, 4)
        xcc = v[:, 1] & 0xf
        t0 = v[:, 2].astype('int64'); t1 = v[:, 3].astype('int64')
        dur = ((t1 - t0) & 0xffffffff) / 100.0                     # us per workgroup
        end = ((t1 - t0.min()) & 0xffffffff) / 100.0
        rate = [round(2.0 * chunk / (dur[xcc == x].mean() * 1e3), 1) if (xcc == x).any() else None for x in range(8)]
        last = [round(float(end[xcc == x].max()), 1) if (xcc == x).any() else None for x in range(8)]
        res[name] = {'GBps_per_wg_by_xcd': rate, 'last_end_us_by_xcd': last,
                     'slowest_over_fastest_xcd': round(max(last) / max(1e-9, min(last)), 3)}
    del a, b
    return res


def _card_dir():
    """sysfs directory of THE device HIP runs on (by PCI address; a box holds eight cards and the lease is one of them:
    card0 is usually somebody else's, idle at 95 MHz); falls back to the first amdgpu device with clock tables"""
    try:
        import torch
        pr = torch.cuda.get_device_properties(torch.cuda.current_device())
        d = '/sys/bus/pci/devices/%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        if os.path.exists(os.path.join(d, 'pp_dpm_sclk')):
            return d
    except Exception:
        pass
    for d in sorted(glob.glob('/sys/class/drm/card*/device')):
        if os.path.exists(os.path.join(d, 'pp_dpm_sclk')):
            return d
    return None


def _read(path, limit=400):
    try:
        with open(path) as f:
            return f.read(limit).strip()
    except OSError:
        return None


def _active(table):
    """'0: 132Mhz\\n1: 2400Mhz *' -> 2400 (MHz of the starred level)"""
    if not table:
        return None
    for line in table.splitlines():
        if '*' in line:
            try:
                return int(''.join(c for c in line.split(':', 1)[1] if c.isdigit()))
            except (ValueError, IndexError):
                return None
    return None


def sysfs():
    d = _card_dir()
    if d is None:
        return {'error': 'no /sys/class/drm/card*/device/pp_dpm_sclk'}
    out = {'dir': d}
    for k in ('pp_dpm_sclk', 'pp_dpm_mclk', 'pp_dpm_fclk', 'pp_dpm_socclk'):
        t = _read(os.path.join(d, k))
        if t is not None:
            out[k] = t.replace('\n', ' | ')
    for k in ('current_compute_partition', 'current_memory_partition', 'power_dpm_force_performance_level',
              'gpu_busy_percent', 'mem_busy_percent'):
        t = _read(os.path.join(d, k), 80)
        if t is not None:
            out[k] = t
    for h in glob.glob(os.path.join(d, 'hwmon', 'hwmon*')):
        for k in ('power1_cap', 'power1_average', 'power1_input', 'freq1_input', 'freq2_input'):
            t = _read(os.path.join(h, k), 40)
            if t is not None:
                out[k] = t
        # every temperature sensor the card exposes (edge / junction / memory ...), milli-degrees C: a memory stack that
        # runs hot refreshes more often -- a candidate for the slow state that the clocks would not show
        for tp in sorted(glob.glob(os.path.join(h, 'temp*_input'))):
            label = _read(tp.replace('_input', '_label'), 20) or os.path.basename(tp)[:-6]
            t = _read(tp, 20)
            if t is not None:
                out['temp_' + label] = t
    return out


def node():
    """the HOST side of the box: the probe calls of round 5 alternated between a fast and a slow state with identical PCI
    addresses in both (profiles/r05_a_box_probes.jsonl); nothing below separated the states either (both host kernels seen
    were slow once), it is recorded so that a future difference is on file.  What the node says about itself:
    kernel, CPU, amdgpu driver version and the module parameters that change how the memory system is driven (retry
    faults / XNACK, page-table fragment size, scheduling policy), IOMMU groups, the ISA string HIP reports (xnack+/-)"""
    out = {}
    try:
        import platform
        out['kernel'] = platform.release()
    except Exception:
        pass
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    out['cpu'] = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        import torch
        out['gcn_arch'] = torch.cuda.get_device_properties(torch.cuda.current_device()).gcnArchName
    except Exception:
        pass
    out['amdgpu_version'] = _read('/sys/module/amdgpu/version', 60)
    prm = {}
    for k in ('noretry', 'vm_fragment_size', 'vm_block_size', 'vm_size', 'sched_policy', 'mes', 'hws_max_conc_proc', 'sdma_phase_quantum',
              'tmz', 'mtype_local', 'use_xgmi_p2p', 'pcie_p2p', 'aspm', 'runpm', 'ppfeaturemask', 'gpu_recovery', 'ras_enable'):
        v = _read('/sys/module/amdgpu/parameters/' + k, 40)
        if v is not None:
            prm[k] = v
    out['amdgpu_parameters'] = prm
    try:
        out['iommu_groups'] = len(os.listdir('/sys/kernel/iommu_groups'))
    except OSError:
        out['iommu_groups'] = None
    out['cmdline'] = (_read('/proc/cmdline', 300) or '')[:300]
    out['env'] = {k: os.environ[k] for k in ('HSA_XNACK', 'HSA_ENABLE_SDMA', 'HSA_ENABLE_IPC_MODE_LEGACY', 'GPU_MAX_HW_QUEUES') if k in os.environ}
    try:
        out['numa_nodes'] = len([d for d in os.listdir('/sys/devices/system/node') if d.startswith('node')])
        d = _card_dir()
        out['gpu_numa_node'] = _read(os.path.join(d, 'numa_node'), 10) if d else None
        out['pcie_link'] = {'speed': _read(os.path.join(d, 'current_link_speed'), 30), 'width': _read(os.path.join(d, 'current_link_width'), 10)} if d else None
        out['loadavg'] = _read('/proc/loadavg', 60)
    except OSError:
        pass
    return out


class ClockSampler(threading.Thread):
    """active sclk / mclk / fclk levels every `period` s while something else runs; summary() -> min / median / max MHz"""

    def __init__(self, period=0.02):
        threading.Thread.__init__(self, daemon=True)
        self.period, self.dir, self.stop_flag, self.samples = period, _card_dir(), threading.Event(), []

    def run(self):
        if self.dir is None:
            return
        while not self.stop_flag.is_set():
            self.samples.append(tuple(_active(_read(os.path.join(self.dir, k))) for k in ('pp_dpm_sclk', 'pp_dpm_mclk', 'pp_dpm_fclk')))
            time.sleep(self.period)

    def summary(self):
        self.stop_flag.set()
        self.join(1.0)
        out = {'samples': len(self.samples)}
        for i, k in enumerate(('sclk', 'mclk', 'fclk')):
            v = sorted(s[i] for s in self.samples if s[i] is not None)
            if v:
                out[k + '_mhz'] = {'min': v[0], 'median': v[len(v) // 2], 'max': v[-1]}
        return out


def box_calibration(device=None, probes=True):
    import torch
    from centertrack_amd import _lib
    from tools.micro import probes
    lib = probes.load()                   # (tools/micro/libct_probes.so: the probes are not part of the product ABI)
    plib = _lib.load()
    st = _lib.stream_ptr()
    device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
    out = torch.zeros(1 << 20, device=device)
    iters, blocks = 50000, 512
    ms = _timed(lambda: lib.ctp_mfma(blocks, iters, ctypes.c_void_p(out.data_ptr()), st))
    mfma = blocks * 4 * iters * 16 * 2.0 * 16 * 16 * 4 / ms / 1e9
    a = torch.empty(1 << 28, device=device)
    b = torch.empty(1 << 28, device=device)
    d2d = 2.0 * a.numel() * 4 / _timed(lambda: plib.ct_memcpy_async(b.data_ptr(), a.data_ptr(), a.numel() * 4, 0, st), 5) / 1e6
    n = 1 << 22
    d2d_small = 2.0 * n * 4 / _timed(lambda: plib.ct_memcpy_async(b.data_ptr(), a.data_ptr(), n * 4, 0, st), 50) / 1e6
    del a, b
    res = {'device': torch.cuda.get_device_name(device), 'mfma_f32_tflops': round(mfma, 1),
           'd2d_1GiB_GBps': round(d2d), 'd2d_16MiB_GBps': round(d2d_small)}
    if probes:
        for key, fn in (('chase', lambda: chase(lib, st, device)), ('stream_GBps', lambda: stream(lib, st, device)),
                        ('launch_us', lambda: launches(lib, device)), ('cu_map', lambda: cu_map(lib, st, device)), ('xcd_stream', lambda: xcd_stream(lib, st, device)),
                        ('sysfs', sysfs), ('node', node)):
            try:
                v = fn()
                if key == 'chase':
                    res['chase_ns'], res['shader_mhz'] = v
                else:
                    res[key] = v
            except Exception as e:       # a probe must never cost the bench line
                res[key] = {'error': repr(e)}
        try:
            res['state'] = fetch_state(res.get('launch_us'))
        except Exception as e:
            res['state'] = {'error': repr(e)}
    return res


IFETCH_FAST_US, IFETCH_SLOW_US = (38.8, 40.1), 56.5      # profiles/r05_a_box_probes.jsonl: nine fast leases, one slow one


def fetch_state(launch_us):
    """which of the pool's two per-lease states this process runs in, as far as the one probe that differs between them can
    tell: the instruction-fetch probe at one wave per SIMD.  The threshold is the midpoint of what was
    measured; a figure between the two clusters is reported as such, not forced into a class."""
    v = (launch_us or {}).get('ifetch_64KB_code_256wg') if isinstance(launch_us, dict) else None
    if not isinstance(v, (int, float)):
        return {'instruction_fetch': 'unknown'}
    lo, hi = IFETCH_FAST_US[1] * 1.05, IFETCH_SLOW_US * 0.92
    cls = 'fast' if v <= lo else 'slow' if v >= hi else 'between'
    return {'instruction_fetch': cls, 'ifetch_64KB_code_256wg_us': v, 'fast_leases_us': list(IFETCH_FAST_US), 'slow_lease_us': IFETCH_SLOW_US,
            'expect_device_ms_mot17_512': {'fast': [0.933, 0.954], 'slow': [1.074, 1.133]}.get(cls),
            'note': 'per-lease state of the box, not of the code: DESIGN.md section 4 "The box classes"'}


if __name__ == '__main__':
    print(json.dumps({'box_calibration': box_calibration()}))
