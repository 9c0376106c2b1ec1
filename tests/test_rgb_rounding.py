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


# ---- the size of the definitional gap --------------------------------------
# The reference's pass 3 has no CPU statement: A.5 stands in for what a GL implementation makes of
# `vec3 rgb = yuvColor*vec3(y, u-128, v-128); color = rgb/255.0;` (res/unyuv.fs.glsl:12-16, 48-49)
# followed by the UNORM8 framebuffer conversion (clamp to [0,1], x*255 rounded to nearest).  The GLSL
# cannot run here, but its LITERAL float sequence can be evaluated over the whole input domain and
# held against A.5's `(int)(clamp(c,0,255) + 0.5f)`:
#   "div"  mat3*vec3 as three multiply-add steps left to right (the zero terms included), a real
#          division by 255.0, UNORM8 = floor(clamp(x,0,1)*255 + 0.5)
#   "fma"  the same with each multiply-add step fused (what a GPU compiler is free to emit)
#   "rcp"  the division replaced by a multiplication with fl(1/255) (also allowed by GLSL)
def shader_literal_rgb(Y, u, v, mode):
    def dot(c0, c1, c2):
        t = (f(c0) * Y).astype(np.float32)
        for c, x in ((c1, u), (c2, v)):
            if mode == "fma":
                t = (np.float64(f(c)) * x.astype(np.float64) + t.astype(np.float64)).astype(np.float32)
            else:
                t = (t + (f(c) * x).astype(np.float32)).astype(np.float32)
        return t

    def unorm8(c):
        x = (c * f(1.0 / 255.0)).astype(np.float32) if mode == "rcp" else (c / f(255.0)).astype(np.float32)
        x = np.clip(x, f(0), f(1)).astype(np.float32)
        return np.floor((x * f(255.0)).astype(np.float32) + f(0.5)).astype(np.int32).astype(np.uint8)
    return (unorm8(dot(1.0, 0.0, 1.402)), unorm8(dot(1.0, -0.34414, -0.71414)),
            unorm8(dot(1.0, 1.772, 0.0)))


def a5_rgb(Y, u, v):
    R = (Y + (f(1.402) * v).astype(np.float32)).astype(np.float32)
    G = ((Y + (f(-0.34414) * u).astype(np.float32)).astype(np.float32)
         + (f(-0.71414) * v).astype(np.float32)).astype(np.float32)
    B = (Y + (f(1.772) * u).astype(np.float32)).astype(np.float32)
    q = lambda c: (np.clip(c, 0, 255).astype(np.float32) + f(0.5)).astype(np.int32).astype(np.uint8)   # noqa: E731
    return q(R), q(G), q(B)


def rgb_gap_counts(orc=None):
    """{mode: (bytes that differ from A.5 per channel, largest difference)} over all 2^24 inputs.
    A.5 = the oracle's own orc_planes_to_rgb when an Oracle is handed in, else the numpy restatement."""
    cb, cr = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))
    cb, cr = np.ascontiguousarray(cb), np.ascontiguousarray(cr)
    u = cb.astype(np.float32) - f(128)
    v = cr.astype(np.float32) - f(128)
    info = None
    if orc is not None:
        import oracle
        info = oracle.Info()
        info.width, info.height, info.ncomps = 256, 256, 3
        for i in range(3):
            info.hblocks[i] = info.vblocks[i] = 32
    out = {}
    for mode in ("div", "fma", "rcp"):
        diff, worst = [0, 0, 0], 0
        for Y in range(256):
            want = a5_rgb(f(Y), u, v)
            if orc is not None:
                rgb = orc.planes_to_rgb(info, [np.full((256, 256), Y, np.uint8), cb, cr])
                assert all(np.array_equal(rgb[..., c], want[c]) for c in range(3)), Y   # (the restatement IS A.5)
            got = shader_literal_rgb(np.full_like(u, Y), u, v, mode)
            for c in range(3):
                d = np.abs(want[c].astype(np.int32) - got[c].astype(np.int32))
                diff[c] += int(np.count_nonzero(d))
                worst = max(worst, int(d.max()))
        out[mode] = (tuple(diff), worst)
    return out


def test_shader_literal_sequence_vs_a5_over_every_input(orc):
    """The count profiles/design_diary_r3_r5.md §4 quotes.  The shader's literal sequence — with or without fused
    multiply-adds — gives A.5's bytes on ALL 3 x 16 777 216 outputs; only an implementation that
    multiplies by fl(1/255) instead of dividing moves 21 G bytes (of 50 331 648) by one."""
    gap = rgb_gap_counts(orc)
    assert gap["div"] == ((0, 0, 0), 0)
    assert gap["fma"] == ((0, 0, 0), 0)
    assert gap["rcp"] == ((0, 21, 0), 1)
