"""Exhaustive proof that the kernels' cheaper RGB forms equal the oracle's
definition on the WHOLE input domain: every (Y, Cb, Cr) byte
triple (16.7M), not a sample.  The kernel side is emulated in numpy float32
operation for operation (csrc/idct_kernels.hip: chroma_row::set + rgb_row);
v_cvt_pk_u8_f32 = round-to-nearest-even + saturate (probed on gfx950,
tools/probe_cvt.hip, profiles/r1_probe_cvt_pk_u8.txt)."""
import numpy as np

f = np.float32


def kernel_rgb(yc, u, v):
    """yc = clamped IDCT output (Y-128), u = Cb-128, v = Cr-128: float32 arrays."""
    y_tie = f(128.000244140625)
    y_half = f(128.5)
    r_ = (f(1.402) * v).astype(np.float32)
    r_ = (r_ + y_tie).astype(np.float32)
    g1 = (f(-0.34414) * u).astype(np.float32)
    g1 = (g1 + y_half).astype(np.float32)
    g2 = (f(-0.71414) * v).astype(np.float32)
    b_ = (f(1.772) * u).astype(np.float32)
    b_ = (b_ + y_tie).astype(np.float32)
    sat_rne = lambda c: np.clip(np.rint(c), 0, 255).astype(np.uint8)      # noqa: E731
    R = sat_rne((yc + r_).astype(np.float32))
    G = np.clip(np.floor(((yc + g1).astype(np.float32) + g2).astype(np.float32)), 0,
                255).astype(np.uint8)
    B = sat_rne((yc + b_).astype(np.float32))
    return R, G, B


def test_kernel_rgb_forms_equal_oracle_on_every_input(orc):
    import oracle
    info = oracle.Info()
    info.width, info.height, info.ncomps = 256, 256, 3
    for i in range(3):
        info.hblocks[i] = 32
        info.vblocks[i] = 32
    cb, cr = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))
    cb = np.ascontiguousarray(cb)
    cr = np.ascontiguousarray(cr)
    u = cb.astype(np.float32) - f(128)
    v = cr.astype(np.float32) - f(128)
    for Y in range(256):
        yp = np.full((256, 256), Y, np.uint8)
        want = orc.planes_to_rgb(info, [yp, cb, cr])
        R, G, B = kernel_rgb(f(Y - 128), u, v)
        assert np.array_equal(want[..., 0], R), Y
        assert np.array_equal(want[..., 1], G), Y
        assert np.array_equal(want[..., 2], B), Y
