"""CPU: lane-level emulation of `dcn16_kernel` (csrc/dcn_mfma.hip, algo 41664) -- the kernel was written without a GPU at
hand, so its index maps are restated here thread by thread (gather assignment, swizzled LDS slots, which wave contracts
which 16-channel slab against which packed weight fragment, the MFMA operand / result lane layouts, the cross-wave
reduction) and the emulated workgroups are compared with the oracle's DCNv2.  This pins the DESIGN of the kernel, not its
compiled code: tests/test_hip_experimental.py does that on a GPU."""
import numpy as np
import pytest
import torch

BM, NKK, WN = 16, 4, 4
SLAB, BUF = BM * 16, NKK * BM * 16


def pack_weight(w):
    """ct_pack_conv_weight: p[((tap*Cin16 + c16)*NT + nt)*256 + g*64 + j*4 + e] = w[nt*16+j][c16*16 + 4g + e][tap]"""
    Cout, Cin, ks, _ = w.shape
    NT = (Cout + 15) // 16
    p = np.zeros((ks * ks, Cin // 16, NT, 4, 16, 4), np.float32)
    wf = w.reshape(Cout, Cin, ks * ks)
    for tap in range(ks * ks):
        for c16 in range(Cin // 16):
            for nt in range(NT):
                for g in range(4):
                    for j in range(16):
                        co = nt * 16 + j
                        if co < Cout:
                            p[tap, c16, nt, g, j, :] = wf[co, c16 * 16 + 4 * g:c16 * 16 + 4 * g + 4, tap]
    return p.reshape(-1), NT


def mfma_16x16x4(a_lane, b_lane, acc):
    """v_mfma_f32_16x16x4_f32: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; lane l holds
    D[i = 4 * (l >> 4) + r][j = l & 15] in element r"""
    A = a_lane.reshape(4, 16).T                # [i][k]
    B = b_lane.reshape(4, 16)                  # [k][j]
    D = A.astype(np.float32) @ B.astype(np.float32)      # [i][j]
    for l in range(64):
        for r in range(4):
            acc[l, r] += D[4 * (l >> 4) + r, l & 15]


def build_table(om, n, oy0, ox0, H, W, ldx):
    """dcn_build_table<16>: (pixel m, tap k) -> 4 corner offsets (floats into the image) + 4 weights (mask folded in)"""
    tab_off = np.zeros((BM * 9, 4), np.int64)
    tab_w = np.zeros((BM * 9, 4), np.float32)
    f = np.float32
    for it in range(BM * 9):
        m, k = it // 9, it % 9
        oy, ox = oy0 + (m >> 4), ox0 + (m & 15)
        if not (oy < H and ox < W):
            continue
        dy, dx, mk = om[n, oy, ox, 2 * k], om[n, oy, ox, 2 * k + 1], om[n, oy, ox, 18 + k]
        ys = f(oy - 1 + k // 3) + dy
        xs = f(ox - 1 + k % 3) + dx
        if ys > -1 and xs > -1 and ys < H and xs < W:
            yf, xf = np.floor(ys), np.floor(xs)
            y0, x0 = int(yf), int(xf)
            y1, x1 = y0 + 1, x0 + 1
            ly, lx = f(ys - yf), f(xs - xf)
            hy, hx = f(1) - ly, f(1) - lx
            for c, (ok, yy, xx, wgt) in enumerate([(y0 >= 0 and x0 >= 0, y0, x0, hy * hx * mk),
                                                    (y0 >= 0 and x1 <= W - 1, y0, x1, hy * lx * mk),
                                                    (y1 <= H - 1 and x0 >= 0, y1, x0, ly * hx * mk),
                                                    (y1 <= H - 1 and x1 <= W - 1, y1, x1, ly * lx * mk)]):
                if ok:
                    tab_off[it, c] = (yy * W + xx) * ldx
                    tab_w[it, c] = wgt
    return tab_off, tab_w


def workgroup(x_nhwc, om, wp, NT, N, H, W, Cin, n, ty, tx, cb, split, chunks_per_split):
    """one workgroup of dcn16_kernel: returns the [16 px][64 couts] partial tile of (image n, row ty, column block tx, cout
    block cb, K split `split`) as the four waves' reduced n-tiles"""
    ldx = Cin
    oy0, ox0 = ty, tx * 16
    nunits = (Cin // 32) >> 1
    c_begin = split * (chunks_per_split >> 1)
    c_end = min(nunits, c_begin + (chunks_per_split >> 1))
    xin = x_nhwc[n].reshape(-1)
    tab_off, tab_w = build_table(om, n, oy0, ox0, H, W, ldx)
    NCH16 = Cin >> 4
    slab_stride = NT << 8
    nt0 = cb * WN
    acc = np.zeros((4, WN, 64, 4), np.float32)          # [wave][nt][lane][r]
    lds = np.zeros(2 * BUF, np.float32)
    nsteps = (c_end - c_begin) * 9
    for s in range(nsteps):
        chunk, tap = c_begin + s // 9, s % 9
        buf = s & 1
        # gather + blend + LDS store of this step's A tile, thread by thread
        for tid in range(256):
            gm, gk, gq = tid >> 4, (tid >> 2) & 3, tid & 3
            lslot = gk * SLAB + gm * 16 + ((gq ^ ((gm >> 1) & 2)) << 2)
            base = chunk * (16 * NKK) + gk * 16 + gq * 4
            v = np.zeros(4, np.float32)
            for c in range(4):
                o = base + tab_off[gm * 9 + tap, c]
                v = v + tab_w[gm * 9 + tap, c] * xin[o:o + 4]
            lds[buf * BUF + lslot:buf * BUF + lslot + 4] = v
        # every wave: its slab of the tile against its slab of the weights, 4 n-tiles
        for wave in range(4):
            af = np.zeros((64, 4), np.float32)
            bq = np.zeros((WN, 64, 4), np.float32)
            for lane in range(64):
                li, lg = lane & 15, lane >> 4
                aoff = wave * SLAB + li * 16 + ((lg ^ ((li >> 1) & 2)) << 2)
                af[lane] = lds[buf * BUF + aoff:buf * BUF + aoff + 4]
                bp = (nt0 << 8) + (lane << 2) + (tap * NCH16 + chunk * NKK + wave) * slab_stride
                for nt in range(WN):
                    bq[nt, lane] = wp[bp + (nt << 8):bp + (nt << 8) + 4]
            for e in range(4):
                for nt in range(WN):
                    mfma_16x16x4(af[:, e], bq[nt, :, e], acc[wave, nt])
    # cross-wave reduction: red[(w * WN + nt) * 64 + lane]; wave w finishes n-tile w, summing the waves in order
    tile = np.zeros((16, 64), np.float32)
    for wave in range(4):
        s_ = acc[0, wave].copy()
        for w in range(1, 4):
            s_ = s_ + acc[w, wave]
        for lane in range(64):
            li, lg = lane & 15, lane >> 4
            for e in range(4):
                tile[lg * 4 + e, wave * 16 + li] = s_[lane, e]
    return tile


@pytest.mark.parametrize('H,W,Cin,Cout,splits,osc', [(3, 20, 64, 64, 1, 1.5), (2, 16, 128, 128, 2, 0.7)])
def test_emulated_16_pixel_workgroups_equal_the_oracle(H, W, Cin, Cout, splits, osc):
    from oracle import dcn_v2 as odcn
    g = torch.Generator().manual_seed(5)
    N = 1
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (Cin * 9) ** -0.5
    off = torch.randn(N, 18, H, W, generator=g) * osc
    mask = torch.sigmoid(torch.randn(N, 9, H, W, generator=g))
    want = odcn.dcn_v2_conv(x, off, mask, w, None).numpy()
    om = np.zeros((N, H, W, 32), np.float32)
    om[..., :18] = off.permute(0, 2, 3, 1).numpy()
    om[..., 18:27] = mask.permute(0, 2, 3, 1).numpy()
    x_nhwc = np.ascontiguousarray(x.permute(0, 2, 3, 1).numpy())
    wp, NT = pack_weight(w.numpy())
    nchunks = Cin // 32
    nunits = nchunks // 2
    cps = -(-nunits // splits) * 2                       # make_plan: chunksPerSplit = cdiv(nunits, splits) * upc
    assert -(-nchunks // cps) == splits
    got = np.zeros((N, Cout, H, W), np.float32)
    for ty in range(H):
        for tx in range((W + 15) // 16):
            for cb in range(Cout // 64):
                tile = sum(workgroup(x_nhwc, om, wp, NT, N, H, W, Cin, 0, ty, tx, cb, sp, cps) for sp in range(splits))
                for m in range(16):
                    if tx * 16 + m < W:
                        got[0, cb * 64:(cb + 1) * 64, ty, tx * 16 + m] = tile[m]
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=2e-5)


# ---- dcn32x_kernel (algo 53264): 32 pixels x 64 couts, contraction on v_mfma_f32_32x32x2_f32 ----------------------------

def mfma_32x32x2(a_lane, b_lane, acc):
    """v_mfma_f32_32x32x2_f32: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; register r of lane
    l holds D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31]   (guide: cdna_hip_programming.md, fragment layouts)"""
    A = a_lane.reshape(2, 32).T                # [i][k]
    B = b_lane.reshape(2, 32)                  # [k][j]
    D = A.astype(np.float32) @ B.astype(np.float32)
    for l in range(64):
        for r in range(16):
            acc[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]


def build_table32(om, n, oy0, ox0, H, W, ldx):
    """dcn_build_table<32>: pixel m of the tile is (row m >> 4, column m & 15)"""
    off = np.zeros((32 * 9, 4), np.int64)
    wgt = np.zeros((32 * 9, 4), np.float32)
    for row in range(2):
        if oy0 + row < H:
            o, w = build_table(om, n, oy0 + row, ox0, H, W, ldx)
            off[row * 144:(row + 1) * 144], wgt[row * 144:(row + 1) * 144] = o, w
    return off, wgt


def workgroup32x(x_nhwc, om, wp, NT, N, H, W, Cin, n, ty, tx, cb, split, chunks_per_split):
    BM32, SLAB32 = 32, 32 * 16
    BUF32 = 4 * SLAB32
    ldx = Cin
    oy0, ox0 = ty * 2, tx * 16
    nunits = (Cin // 32) >> 1
    c_begin = split * (chunks_per_split >> 1)
    c_end = min(nunits, c_begin + (chunks_per_split >> 1))
    xin = x_nhwc[n].reshape(-1)
    tab_off, tab_w = build_table32(om, n, oy0, ox0, H, W, ldx)
    NCH16 = Cin >> 4
    slab_stride = NT << 8
    acc = np.zeros((4, 64, 16), np.float32)             # [wave][lane][r]
    lds = np.zeros(2 * BUF32, np.float32)
    for s in range((c_end - c_begin) * 9):
        chunk, tap = c_begin + s // 9, s % 9
        buf = s & 1
        for tid in range(256):
            gm, gq, gk0 = tid >> 3, tid & 3, (tid >> 2) & 1
            lslot = gm * 16 + ((gq ^ ((gm >> 1) & 2)) << 2)
            base = chunk * 64 + gk0 * 16 + gq * 4
            for kk in range(2):
                v = np.zeros(4, np.float32)
                for c in range(4):
                    o = base + tab_off[gm * 9 + tap, c] + kk * 32
                    v = v + tab_w[gm * 9 + tap, c] * xin[o:o + 4]
                d = buf * BUF32 + (gk0 + kk * 2) * SLAB32 + lslot
                lds[d:d + 4] = v
        for wave in range(4):
            wn, wk = wave & 1, wave >> 1
            af = np.zeros((2, 2, 64, 4), np.float32)
            bq = np.zeros((2, 2, 64, 4), np.float32)
            for lane in range(64):
                j, kh = lane & 31, lane >> 5
                nt16 = cb * 4 + wn * 2 + (j >> 4)
                bbase = (nt16 << 8) + kh * 128 + ((j & 15) << 2)
                for sl in range(2):
                    bp = bbase + (tap * NCH16 + chunk * 4 + 2 * wk + sl) * slab_stride
                    bq[sl, 0, lane] = wp[bp:bp + 4]
                    bq[sl, 1, lane] = wp[bp + 64:bp + 68]
                    for qq in range(2):
                        aoff = j * 16 + (((2 * kh + qq) ^ ((j >> 1) & 2)) << 2)
                        a0 = buf * BUF32 + (2 * wk + sl) * SLAB32 + aoff
                        af[sl, qq, lane] = lds[a0:a0 + 4]
            for sl in range(2):
                for qq in range(2):
                    for e in range(4):
                        mfma_32x32x2(af[sl, qq, :, e], bq[sl, qq, :, e], acc[wave])
    tile = np.zeros((32, 64), np.float32)
    for wn in range(2):
        sm = acc[wn] + acc[2 + wn]                      # K half 0 + K half 1
        for lane in range(64):
            for r in range(16):
                m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                tile[m, wn * 32 + (lane & 31)] = sm[lane, r]
    return tile


@pytest.mark.parametrize('H,W,Cin,Cout,splits,osc', [(3, 20, 64, 64, 1, 1.5), (4, 16, 128, 128, 2, 0.7)])
def test_emulated_32x32x2_workgroups_equal_the_oracle(H, W, Cin, Cout, splits, osc):
    from oracle import dcn_v2 as odcn
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (Cin * 9) ** -0.5
    off = torch.randn(1, 18, H, W, generator=g) * osc
    mask = torch.sigmoid(torch.randn(1, 9, H, W, generator=g))
    want = odcn.dcn_v2_conv(x, off, mask, w, None).numpy()
    om = np.zeros((1, H, W, 32), np.float32)
    om[..., :18] = off.permute(0, 2, 3, 1).numpy()
    om[..., 18:27] = mask.permute(0, 2, 3, 1).numpy()
    x_nhwc = np.ascontiguousarray(x.permute(0, 2, 3, 1).numpy())
    wp, NT = pack_weight(w.numpy())
    nunits = Cin // 64
    cps = -(-nunits // splits) * 2
    got = np.zeros((1, Cout, H, W), np.float32)
    for ty in range((H + 1) // 2):
        for tx in range((W + 15) // 16):
            for cb in range(Cout // 64):
                tile = sum(workgroup32x(x_nhwc, om, wp, NT, 1, H, W, Cin, 0, ty, tx, cb, sp, cps) for sp in range(splits))
                for m in range(32):
                    oy, ox = ty * 2 + (m >> 4), tx * 16 + (m & 15)
                    if oy < H and ox < W:
                        got[0, cb * 64:(cb + 1) * 64, oy, ox] = tile[m]
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=2e-5)
