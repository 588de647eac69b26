"""Frame feeder: planar YUV reader for the luma plane (the step before the ME path; reference
TLibVideoIO/TVideoIOYuv.cpp:680 `read`, :247 `readPlane`: 8-bit files hold one byte per sample,
higher bit depths two bytes little-endian; 4:2:0 chroma follows luma and is skipped here because
integer ME never reads it).  Padding to HM's 80-sample margin is done on the device by
hmme_plane_upload_* (TComPicYuv::extendPicBorder, TComPicYuv.cpp:214-262)."""
import numpy as np


def frame_bytes(width, height, file_bit_depth=8, chroma="420"):
    bps = 1 if file_bit_depth <= 8 else 2
    luma = width * height * bps
    c = {"400": 0, "420": luma // 2, "422": luma, "444": 2 * luma}[chroma]
    return luma + c


def read_luma(path, width, height, frame, file_bit_depth=8, chroma="420"):
    """-> (height, width) uint8 (8-bit files) or uint16 (16-bit little-endian files) luma of picture `frame`"""
    bps = 1 if file_bit_depth <= 8 else 2
    off = frame * frame_bytes(width, height, file_bit_depth, chroma)
    dt = np.uint8 if bps == 1 else np.dtype("<u2")
    a = np.fromfile(path, dtype=dt, count=width * height, offset=off)
    if a.size != width * height:
        raise ValueError(f"{path}: picture {frame} is beyond the end of the file")
    return a.reshape(height, width)


class LumaFile:
    """luma planes of a planar YUV file, read straight into caller-owned (page-locked) buffers: the feeder of the streaming
    sequence pipeline (hmme/sequence.py).  os.preadv releases the GIL, so a reader thread overlaps with the GPU work."""

    def __init__(self, path, width, height, file_bit_depth=8, chroma="420"):
        import os
        self.width, self.height, self.bit_depth = width, height, file_bit_depth
        self.bps = 1 if file_bit_depth <= 8 else 2
        self.frame_bytes = frame_bytes(width, height, file_bit_depth, chroma)
        self.fd = os.open(path, os.O_RDONLY)
        self.n_frames = os.fstat(self.fd).st_size // self.frame_bytes
        self.path = path

    def read_into(self, frame, out):
        """out: C-contiguous (height, width) uint8 / uint16 array (little-endian host)"""
        import os
        want = self.width * self.height * self.bps
        assert out.nbytes == want and out.flags["C_CONTIGUOUS"]
        if frame < 0 or frame >= self.n_frames:
            raise ValueError(f"{self.path}: picture {frame} is beyond the end of the file")
        mv = memoryview(out).cast("B")
        got, off = 0, frame * self.frame_bytes
        while got < want:
            n = os.preadv(self.fd, [mv[got:]], off + got)
            if n <= 0:
                raise ValueError(f"{self.path}: short read in picture {frame}")
            got += n

    def luma(self, frame):
        out = np.empty((self.height, self.width), np.uint8 if self.bps == 1 else np.uint16)
        self.read_into(frame, out)
        return out

    def close(self):
        import os
        if self.fd is not None:
            os.close(self.fd)
            self.fd = None


def write_luma_420(path, frames):
    """write 8-bit 4:2:0 pictures (luma given, chroma = 128): test helper"""
    with open(path, "wb") as f:
        for y in frames:
            y = np.ascontiguousarray(y, dtype=np.uint8)
            f.write(y.tobytes())
            f.write(np.full(y.size // 2, 128, np.uint8).tobytes())
