"""Synthetic luma frames for the ME benchmark and the parity tests.

Frame t ("ref") is fixed-seed uniform noise low-passed with a 5x5 box and stretched to the
full sample range; frame t+1 ("cur") is the same texture translated by a known per-region
motion vector plus small seeded noise, so arg-mins are non-trivial and checkable.  Both are
returned as HM-style padded planes: int16 `Pel` samples, `margin` (=80) edge-replicated
samples on every side exactly like TComPicYuv (reference TLibCommon/TComPicYuv.cpp:91-92,
214-262).  Luma only: integer ME never reads chroma.
"""
import numpy as np

MARGIN = 80  # maxCUWidth(64) + 16, TComPicYuv.cpp:91-92


def _box5(a):
    """5x5 box filter, edge mode 'nearest', float64 in/out"""
    p = np.pad(a, 2, mode="edge")
    c = np.cumsum(np.cumsum(p, axis=0), axis=1)
    c = np.pad(c, ((1, 0), (1, 0)))
    h, w = a.shape
    return (c[5:5 + h, 5:5 + w] - c[0:h, 5:5 + w] - c[5:5 + h, 0:w] + c[0:h, 0:w]) / 25.0


def pad_plane(img, margin=MARGIN):
    """edge-replicate like TComPicYuv::extendPicBorder; returns C-contiguous int16"""
    return np.ascontiguousarray(np.pad(img, margin, mode="edge").astype(np.int16))


def make_pair(width, height, seed=1234, bit_depth=8, max_mv=12, region=128, noise_sigma=2.0, margin=MARGIN,
              shift=(0, 0), pad=0):
    """-> (cur_padded, ref_padded, true_mv[regions_y, regions_x, 2]) ; planes are
    (height+2*margin, width+2*margin) int16, sample (0,0) at [margin, margin].
    shift=(sx, sy) additionally translates `cur` as a whole (frame t of a synthetic sequence); give all frames
    of a sequence the same `pad` (>= the largest shift + max_mv + 2) so that they share one base texture."""
    rng = np.random.default_rng(seed)
    maxv = (1 << bit_depth) - 1
    g = max(max_mv + 2 + max(abs(shift[0]), abs(shift[1])), pad)
    base = _box5(rng.integers(0, 256, size=(height + 2 * g, width + 2 * g)).astype(np.float64))
    lo, hi = base.min(), base.max()
    base = np.clip(np.rint((base - lo) * (maxv / (hi - lo))), 0, maxv)
    ref = base[g:g + height, g:g + width]
    ry, rx = (height + region - 1) // region, (width + region - 1) // region
    mv = rng.integers(-max_mv, max_mv + 1, size=(ry, rx, 2))
    cur = np.empty_like(ref)
    for j in range(ry):
        for i in range(rx):
            y0, y1 = j * region, min((j + 1) * region, height)
            x0, x1 = i * region, min((i + 1) * region, width)
            dx, dy = int(mv[j, i, 0]) + shift[0], int(mv[j, i, 1]) + shift[1]
            # cur(x,y) = ref(x+dx, y+dy): the best integer MV of the region is (dx,dy)
            cur[y0:y1, x0:x1] = base[g + y0 + dy:g + y1 + dy, g + x0 + dx:g + x1 + dx]
    cur = np.clip(np.rint(cur + rng.normal(0.0, noise_sigma * (1 << (bit_depth - 8)), size=cur.shape)), 0, maxv)
    return pad_plane(cur, margin), pad_plane(ref, margin), mv


class Sequence:
    """Synthetic sequence for the open-loop sequence driver (tools/me_sequence.py, BASELINE config 4): picture t is ONE base
    texture translated by t * step, so the best 64x64 MV of the pair (cur, ref) is (ref - cur) * step wherever the window reaches
    it.  Pictures are unpadded (height, width) uint8 / uint16 arrays -- what a YUV file holds; padding happens on the device.
    Equal to make_pair(..., max_mv=0, noise_sigma=0, shift=(3t, 2t), pad=3 * n_frames + 4)[0] without its margin."""

    def __init__(self, width, height, n_frames, seed=777, bit_depth=8, step=(3, 2)):
        self.width, self.height, self.n_frames, self.bit_depth, self.step = width, height, n_frames, bit_depth, step
        rng = np.random.default_rng(seed)
        maxv = (1 << bit_depth) - 1
        g = self.g = max(step) * n_frames + 4
        base = _box5(rng.integers(0, 256, size=(height + 2 * g, width + 2 * g)).astype(np.float64))
        lo, hi = base.min(), base.max()
        self.base = np.clip(np.rint((base - lo) * (maxv / (hi - lo))), 0, maxv).astype(np.uint8 if bit_depth == 8 else np.uint16)

    def luma(self, t):
        g, (sx, sy) = self.g, self.step
        return np.ascontiguousarray(self.base[g + sy * t:g + sy * t + self.height, g + sx * t:g + sx * t + self.width])

    def read_into(self, t, out):
        g, (sx, sy) = self.g, self.step
        np.copyto(out, self.base[g + sy * t:g + sy * t + self.height, g + sx * t:g + sx * t + self.width])

    def padded(self, t, margin=MARGIN):
        """HM-style padded int16 plane of picture t (what the CPU oracle takes)"""
        return pad_plane(self.luma(t), margin)

    def write_yuv(self, path, chroma="420"):
        """planar file like HM's input: 8-bit samples, or 16-bit little-endian words above 8 bit (TVideoIOYuv.cpp:247)"""
        bps = 1 if self.bit_depth == 8 else 2
        n_c = {"400": 0, "420": self.width * self.height // 2}[chroma]
        grey = np.full(n_c, 1 << (self.bit_depth - 1), np.uint8 if bps == 1 else np.dtype("<u2")).tobytes()
        with open(path, "wb") as f:
            for t in range(self.n_frames):
                f.write(self.luma(t).astype(np.uint8 if bps == 1 else np.dtype("<u2")).tobytes())
                f.write(grey)


def random_predictors(n_ctu, seed, max_pel=16):
    """seeded quarter-pel AMVP predictors, |pred| <= max_pel pels"""
    rng = np.random.default_rng(seed)
    return rng.integers(-4 * max_pel, 4 * max_pel + 1, size=(n_ctu, 2)).astype(np.int16)
