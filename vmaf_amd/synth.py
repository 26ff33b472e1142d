"""Deterministic synthetic ref/dis luma clips (integer arithmetic only, so every numpy version and
every machine produces the same bytes).  Used by the tests, the golden-fixture generator and bench.py;
shape follows SURVEY.md 8(d): a band-limited panning texture with grain as ref, a blurred +
quantised + noisy copy as dis, so that the features land mid-range.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    """splitmix64 finaliser on a uint64 array."""
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x += np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def _noise(shape, seed, lo, hi):
    """Uniform integers in [lo, hi] as int32, a pure function of (shape, seed)."""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _mix(idx + np.uint64(seed) * np.uint64(0x632BE59BD9B4E019))
    return (lo + (h % np.uint64(hi - lo + 1)).astype(np.int64)).astype(np.int32).reshape(shape)


def _upsample8(coarse, h, w):
    """Integer bilinear x8 upsample of a coarse grid to (h, w)."""
    ys = np.arange(h)
    xs = np.arange(w)
    y0, fy = ys // 8, ys % 8
    x0, fx = xs // 8, xs % 8
    c = coarse.astype(np.int64)
    a = c[y0][:, x0]
    b = c[y0][:, x0 + 1]
    cc = c[y0 + 1][:, x0]
    d = c[y0 + 1][:, x0 + 1]
    fy = fy[:, None]
    fx = fx[None, :]
    top = a * (8 - fx) + b * fx
    bot = cc * (8 - fx) + d * fx
    return ((top * (8 - fy) + bot * fy + 32) // 64).astype(np.int32)


def _box3(img):
    p = np.pad(img.astype(np.int64), 1, mode="edge")
    s = sum(p[i:i + img.shape[0], j:j + img.shape[1]] for i in range(3) for j in range(3))
    return ((s + 4) // 9).astype(np.int32)


def make_clip(width, height, frames, seed=1234, bits=8, cuts=(), static=False, pan=2, grain=3, dis_noise=5, quant=6):
    """-> (ref, dis) uint8 [F,h,w] (bits=8) or uint16 10-bit [F,h,w] (bits=10).

    cuts: frame indices at which the scene changes (a new texture starts).  static=True: no pan, no
    per-frame grain (motion exactly 0 after frame 0)."""
    assert bits in (8, 10)
    ref = np.empty((frames, height, width), np.int32)
    dis = np.empty_like(ref)
    scene = 0
    canvas = None
    scene_start = 0
    for f in range(frames):
        if f == 0 or f in cuts:
            scene += 1
            scene_start = f
            span = width + pan * frames + 16
            coarse = _noise((height // 8 + 3, span // 8 + 3), seed * 1000 + scene, 16, 235)
            canvas = _upsample8(coarse, height, span)
        off = 0 if static else pan * (f - scene_start)
        tex = canvas[:, off:off + width]
        g_seed = seed * 7919 + (0 if static else f) + 1
        r = np.clip(tex + _noise((height, width), g_seed, -grain, grain), 0, 255)
        b = _box3(_box3(r))
        b = (b // quant) * quant + quant // 2
        d = np.clip(b + _noise((height, width), g_seed + 500009, -dis_noise, dis_noise), 0, 255)
        ref[f], dis[f] = r, d
    if bits == 8:
        return ref.astype(np.uint8), dis.astype(np.uint8)
    dr = _noise(ref.shape, seed + 11, 0, 3)
    dd = _noise(ref.shape, seed + 13, 0, 3)
    return (ref * 4 + dr).astype(np.uint16), (dis * 4 + dd).astype(np.uint16)


def checkerboard(width=1920, height=1080, block=10, frames=3, dx=0, dy=0, lo=16, hi=235):
    """The reference's test clips `checkerboard_<w>_<h>_<block>_<frames>_<dx>_<dy>.yuv` (python/test/resource/yuv; absent
    from the reference tree, .MISSING_LARGE_BLOBS:1-3), re-derived from the numbers its tests pin for them
    (python/test/feature_test.py:566-657, quality_runner_test.py:333-389): video-range squares of `block` pixels,
    luma = lo where floor((x - t - dx) / block) + floor((y - dy) / block) is even, hi elsewhere, i.e. the board moves
    right by one pixel per frame t and the "_dx_dy" variants are the same board displaced further.  With these
    parameters the reference's C code reproduces every pinned value of those tests to the printed digits
    (tests/golden/reference_pinned.json, tests/test_reference_pinned.py).  -> uint8 [frames, height, width]"""
    y, x = np.mgrid[0:height, 0:width]
    out = np.empty((frames, height, width), np.uint8)
    for t in range(frames):
        par = (np.floor_divide(x - t - dx, block) + np.floor_divide(y - dy, block)) & 1
        out[t] = np.where(par == 0, lo, hi)
    return out


def to_luma_f32(planes, bits=8):
    """What a read_frame callback delivers (feature/src/common/file_io.c:112-116,159-163):
    8-bit -> value, 10-bit -> value / 4.0, as fp32."""
    if bits == 8:
        return planes.astype(np.float32)
    return (planes.astype(np.float64) / 4.0).astype(np.float32)


def write_yuv(path, luma, fmt):
    """Write planes [F,h,w] as a raw YUV file of pixel format fmt with constant mid-grey chroma."""
    ten = fmt.endswith("10le")
    F, h, w = luma.shape
    n = h * w
    chroma = {"yuv420p": n // 2, "yuv422p": n, "yuv444p": 2 * n}[fmt.replace("10le", "")]
    with open(path, "wb") as f:
        for i in range(F):
            if ten:
                f.write(luma[i].astype("<u2").tobytes())
                f.write(np.full(chroma, 512, "<u2").tobytes())
            else:
                f.write(luma[i].astype(np.uint8).tobytes())
                f.write(np.full(chroma, 128, np.uint8).tobytes())
