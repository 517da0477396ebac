"""Frame ingest without a codec library (row N3 of SURVEY §8(f), the part that can exist here).

The reference decodes with cv2.VideoCapture (FFmpeg underneath): sequential reads in the frame selectors
(backend/main.py:228-376) and random seeks per OCR task (backend/tools/subtitle_ocr.py:173-204).  Neither cv2 nor ffmpeg
exists on the build or the GPU box, so compressed video cannot be decoded by anything in this repository; what CAN be read
bit-exactly is uncompressed video: AVI with 24-bit BGR DIB frames ('DIB ' / BI_RGB — what `ffmpeg -c:v rawvideo -pix_fmt bgr24
out.avi` writes and cv2 reads back unchanged), and a stack of frames in a .npy file.  Both give the frame-source interface of
extractor.py: frame_count, fps, read(frame_no) (1-based random access, like CAP_PROP_POS_FRAMES = frame_no - 1 then read()),
frames() (decode order), pos_msec(frame_no) (what the SRT writer asks for, see AviBgr24Source.pos_msec).

Round 3: Motion-JPEG AVI (`ffmpeg -c:v mjpeg`, what many capture devices write) is the ONE compressed format that can be read here —
every frame is a stand-alone baseline JPEG, and Pillow (libjpeg-turbo; importable on both boxes) decodes it.  The reference decodes
the same stream with FFmpeg's own MJPEG decoder through cv2.VideoCapture: its IDCT / chroma up-sampling differ from libjpeg's by
+-1..2 grey levels on some pixels, so this source is NOT bit-identical to the reference's decode (the uncompressed sources are).
"""
import struct

import numpy as np


def write_avi_bgr24(path, frames, fps, riff_frames=None, dropped=()):
    """Uncompressed AVI (one 'vids' stream, BI_RGB 24 bit, bottom-up rows padded to 4 bytes, idx1 index).
    riff_frames: start a new `RIFF....AVIX` segment (OpenDML, what ffmpeg's muxer does after ~1 GiB) every that many frames;
    dropped: 0-based frame indices written as zero-length chunks (a dropped frame: the previous picture is shown again)."""
    frames = [np.asarray(f) for f in frames]
    h, w, _ = frames[0].shape
    stride = (w * 3 + 3) & ~3
    size = stride * h
    rate, scale = int(round(fps * 1000)), 1000

    def chunk(tag, data):
        return tag + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b"")

    def lst(tag, data):
        return b"LIST" + struct.pack("<I", len(data) + 4) + tag + data
    avih = struct.pack("<14I", int(1e6 / fps), size * int(fps + 1), 0, 0x10, len(frames), 0, 1, size, w, h, 0, 0, 0, 0)
    strh = b"vids" + b"DIB " + struct.pack("<IHHIIIIIIII4H", 0, 0, 0, 0, scale, rate, 0, len(frames), size, 0xFFFFFFFF, 0, 0, 0, w, h)
    strf = struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, size, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    segments, movi, idx = [], b"", b""
    for k, f in enumerate(frames):
        assert f.shape == (h, w, 3) and f.dtype == np.uint8
        if riff_frames and k and k % riff_frames == 0:
            segments.append(movi)
            movi = b""
        if k in dropped:
            data = b""
        else:
            rows = np.zeros((h, stride), np.uint8)
            rows[:, :w * 3] = f[::-1].reshape(h, w * 3)
            data = rows.tobytes()
        if not segments:
            idx += b"00db" + struct.pack("<III", 0x10, 4 + len(movi), len(data))       # idx1 covers the first RIFF only
        movi += chunk(b"00db", data)
    segments.append(movi)
    body = b"AVI " + hdrl + lst(b"movi", segments[0]) + chunk(b"idx1", idx)
    with open(path, "wb") as fp:
        fp.write(b"RIFF" + struct.pack("<I", len(body)) + body)
        for seg in segments[1:]:
            body = b"AVIX" + lst(b"movi", seg)
            fp.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def write_avi_mjpeg(path, frames, fps, quality=90):
    """Motion-JPEG AVI (one 'vids' stream, fourcc MJPG, every frame a baseline JPEG encoded by Pillow)."""
    import io
    from PIL import Image
    frames = [np.asarray(f) for f in frames]
    h, w, _ = frames[0].shape
    rate, scale = int(round(fps * 1000)), 1000
    blobs = []
    for f in frames:
        bio = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(f[:, :, ::-1])).save(bio, format="JPEG", quality=quality)
        blobs.append(bio.getvalue())
    big = max(len(b) for b in blobs)

    def chunk(tag, data):
        return tag + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b"")

    def lst(tag, data):
        return b"LIST" + struct.pack("<I", len(data) + 4) + tag + data
    avih = struct.pack("<14I", int(1e6 / fps), big * int(fps + 1), 0, 0x10, len(frames), 0, 1, big, w, h, 0, 0, 0, 0)
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII4H", 0, 0, 0, 0, scale, rate, 0, len(frames), big, 0xFFFFFFFF, 0, 0, 0, w, h)
    strf = struct.pack("<IiiHH4sIiiII", 40, w, h, 1, 24, b"MJPG", w * h * 3, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    movi, idx = b"", b""
    for b in blobs:
        idx += b"00dc" + struct.pack("<III", 0x10, 4 + len(movi), len(b))
        movi += chunk(b"00dc", b)
    body = b"AVI " + hdrl + lst(b"movi", movi) + chunk(b"idx1", idx)
    with open(path, "wb") as fp:
        fp.write(b"RIFF" + struct.pack("<I", len(body)) + body)


class AviBgr24Source:
    """Random-access reader of uncompressed BGR24 AVI (write_avi_bgr24 / ffmpeg rawvideo bgr24) and of Motion-JPEG AVI
    (write_avi_mjpeg / ffmpeg -c:v mjpeg; frames decoded by Pillow); any other codec is refused."""

    def __init__(self, path):
        self.path = path
        self._fp = open(path, "rb")
        head = self._fp.read(12)
        if head[:4] != b"RIFF" or head[8:12] != b"AVI ":
            raise ValueError(f"{path}: not a RIFF AVI file")
        self._offsets = []
        self.width = self.height = None
        self.fps = None
        self.mjpeg = False
        # every top-level RIFF chunk: the 'AVI ' one, then the OpenDML 'AVIX' segments a muxer opens about every GiB (170 frames
        # of 1080p bgr24) — a reader that stops after the first would silently see the first seconds of a clip only
        self._fp.seek(0, 2)
        size, pos = self._fp.tell(), 0
        while pos + 12 <= size:
            self._fp.seek(pos)
            tag, n, kind = struct.unpack("<4sI4s", self._fp.read(12))
            if tag != b"RIFF" or kind not in (b"AVI ", b"AVIX"):
                raise ValueError(f"{path}: {size - pos} bytes behind the last RIFF chunk are not an AVI segment ({tag!r} {kind!r})")
            self._walk(pos + 12, min(pos + 8 + n, size))
            pos += 8 + n + (n & 1)
        if self.width is None or self.fps is None:
            raise ValueError(f"{path}: no video stream header")
        self.frame_count = len(self._offsets)
        self._stride = (self.width * 3 + 3) & ~3

    def _walk(self, pos, end):
        fp = self._fp
        while pos + 8 <= end:
            fp.seek(pos)
            tag, n = struct.unpack("<4sI", fp.read(8))
            if tag == b"LIST":
                kind = fp.read(4)
                if kind in (b"hdrl", b"strl", b"movi"):
                    self._walk(pos + 12, pos + 8 + n)
            elif tag == b"strh":
                d = fp.read(n)
                if d[:4] == b"vids":
                    self.fourcc = d[4:8]
                    if d[4:8].upper() in (b"MJPG", b"JPEG"):
                        self.mjpeg = True
                    elif d[4:8] not in (b"DIB ", b"\0\0\0\0", b"RAW "):
                        raise ValueError(f"{self.path}: compressed video ({d[4:8]!r}) needs a codec; only uncompressed BGR24 and "
                                         "Motion-JPEG AVI can be read here")
                    scale, rate = struct.unpack("<II", d[20:28])
                    self.fps = rate / float(scale)
            elif tag == b"strf":
                d = fp.read(n)
                _sz, w, h, _planes, bits, comp = struct.unpack("<IiiHHI", d[:20])
                if getattr(self, "mjpeg", False):
                    self.width, self.height, self._bottom_up = w, abs(h), False
                elif bits != 24 or comp != 0:
                    raise ValueError(f"{self.path}: only BI_RGB 24-bit frames are supported (bits={bits}, compression={comp})")
                self.width, self.height, self._bottom_up = w, abs(h), h > 0
            elif tag in (b"00db", b"00dc"):
                self._offsets.append((pos + 8, n))
            pos += 8 + n + (n & 1)

    def read(self, frame_no):
        if not 1 <= frame_no <= self.frame_count:
            return None
        off, n = self._offsets[frame_no - 1]
        while n == 0 and frame_no > 1:          # zero-length chunk = dropped frame: the previous picture stays on screen
            frame_no -= 1
            off, n = self._offsets[frame_no - 1]
        if self.mjpeg:
            if n == 0:
                return None
            import io
            from PIL import Image
            self._fp.seek(off)
            img = Image.open(io.BytesIO(self._fp.read(n)))
            rgb = np.asarray(img.convert("RGB"))
            if rgb.shape[:2] != (self.height, self.width):
                raise ValueError(f"{self.path}: frame {frame_no} is {rgb.shape[1]}x{rgb.shape[0]}, the stream header says {self.width}x{self.height}")
            return np.ascontiguousarray(rgb[:, :, ::-1])            # BGR like cv2
        if n < self._stride * self.height:
            return None
        self._fp.seek(off)
        rows = np.frombuffer(self._fp.read(n), np.uint8)[:self._stride * self.height].reshape(self.height, self._stride)
        img = rows[:, :self.width * 3].reshape(self.height, self.width, 3)
        return np.ascontiguousarray(img[::-1] if self._bottom_up else img)

    def frames(self):
        for no in range(1, self.frame_count + 1):
            yield self.read(no)

    def pos_msec(self, frame_no):
        """main.py:738-743: cap.set(CAP_PROP_POS_FRAMES, frame_no); cap.read(); cap.get(CAP_PROP_POS_MSEC) — the time stamp of
        the frame with 0-based index frame_no (constant frame rate here); None when that read fails (past the end)."""
        return None if not 0 <= frame_no < self.frame_count else frame_no * 1000.0 / self.fps

    def close(self):
        self._fp.close()


class NpySource:
    """Frames stacked in one .npy file [N,H,W,3] uint8 (memory-mapped)."""

    def __init__(self, path, fps):
        self._a = np.load(path, mmap_mode="r")
        if self._a.ndim != 4 or self._a.shape[3] != 3 or self._a.dtype != np.uint8:
            raise ValueError(f"{path}: expected uint8 [N,H,W,3]")
        self.frame_count, self.fps = int(self._a.shape[0]), float(fps)

    def read(self, frame_no):
        return np.ascontiguousarray(self._a[frame_no - 1]) if 1 <= frame_no <= self.frame_count else None

    def frames(self):
        for no in range(1, self.frame_count + 1):
            yield self.read(no)

    pos_msec = None


def open_source(path, fps=None):
    if str(path).endswith(".npy"):
        if fps is None:
            raise ValueError("a .npy frame stack carries no frame rate: pass fps")
        return NpySource(path, fps)
    return AviBgr24Source(path)
